// gnuais_capi.hip -- host side of the C ABI declared in include/gnuais_hip.h.
//
// Owns the per-batch device state (FIR history, PLL phase, deframer state,
// frame ring) and sequences the kernels of one receiver_run() pass:
//   K1 fir_slice (+ history carry) -> K2 pll -> K2b hdlc_deframe -> K3 hdlc_crc
// K1 on the caller's stream, every later stage on an internal stream of its own, chained by
// events over four sets of hand-off buffers, so that the stages of consecutive calls overlap
// (DESIGN.md 4.6); `pipeline` = 0 runs them back to back on the caller's stream instead.
// No CPU implementation of the chain exists here: without a usable HIP device every entry
// point fails with GNUAIS_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/gnuais_hip.h"
#include "kernels.h"

using namespace gnuais;

namespace gnuais {
namespace scalar { hipError_t launch_fir_slice(const FirLaunch &a, hipStream_t stream); }
}

static thread_local std::string g_err;

static int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) return fail(GNUAIS_E_HIP, #expr, e_);            \
    } while (0)

// src/receiver.c:39-49, first half of the symmetric table (double literals that
// round to fp32 on assignment, exactly as the reference's static float array)
static const double k_tap_half[18] = {
    2.5959e-55, 2.9479e-49, 1.4741e-43, 3.2462e-38, 3.1480e-33, 1.3443e-28,
    2.5280e-24, 2.0934e-20, 7.6339e-17, 1.2259e-13, 8.6690e-11, 2.6996e-08,
    3.7020e-06, 2.2355e-04, 5.9448e-03, 6.9616e-02, 3.5899e-01, 8.1522e-01};

struct gnuais_batch {
    int device = 0;
    int N = 0, NT = 0, NE = 0, d = 0;
    uint32_t pllinc = 0;
    int max_len = 0, frame_cap = 0;
    int sgn_words = 0, bits_words = 0;
    std::vector<float> taps;
    float te[128] = {0};
    // device state
    // The FIR's carry (the last NT input rows) and the per-call peak buffers rotate over HB buffers: call i reads
    // hist[i % HB], writes hist[(i + 1) % HB], gathers its peaks in maxval[i % HB] and clears maxval[(i + 2) % HB].
    // (Two would do for FIR launches that run one after the other; four keep the writer of a buffer two calls away
    // from its readers.)
    static constexpr int HB = 4;
    int16_t *hist[HB] = {};
    int hist_cur = 0;
    // every hand-off buffer exists NBUF times; `nbuf` of them are in use (index = call % nbuf), so K1 can run up
    // to nbuf-1 calls ahead of the sequential stages
    static constexpr int NBUF = 8;
    // Depth in use.  The host waits for K3 of call i-nbuf before it launches the FIR of call i, so the pipeline is a
    // closed loop: period >= latency of a call / nbuf.  Round 4, C3, same box (profiles/r04_nbuf_3_vs_4.txt): depth 3
    // and 4 give the same steady state (0.518 ms: at 3 the loop's bound and the PLL stage's duration meet), 5-8 no
    // better (0.53-0.55), 2 starves (0.70); a short timed region ends sooner with fewer calls in flight to drain
    // (20 steps: 0.574 against 0.583), so 3.
    int nbuf = 3;
    int sets_alloc = 0;                         // hand-off sets that exist (>= nbuf)
    uint32_t *sgn[NBUF] = {};                   // K1 -> K2
    uint32_t *pll = nullptr, *lastbit = nullptr, *prev = nullptr;   // receiver.h:38-44, carried by K2
    int n_cu = 256;
    uint32_t *segbits[NBUF] = {};               // K2 -> K2b
    uint32_t *segcnt[NBUF] = {};
    int n_seg = 0, seg_words = 0;
    // stage pipeline: K1 on the caller's stream and one internal stream per later
    // kernel, so that the short-on-parallelism stages of call i overlap the FIR of
    // call i+1 (and each other).
    hipStream_t s_k[4] = {nullptr, nullptr, nullptr, nullptr};   // K2, (spare), K2b, K3 (entries of pool[])
    hipStream_t s_k_default[4] = {nullptr, nullptr, nullptr, nullptr};
    static constexpr int POOL = 12;
    hipStream_t pool[POOL] = {};                // candidates for gnuais_batch_autotune(): [0..3] the default
                                                // assignment, [0..7] high priority, [8..11] default priority
    hipEvent_t e_done[5][NBUF] = {};            // e_done[s][k]: stage s of the call using set k is done
                                                // (0 K1, 1 K2, 3 K2b, 4 K3)
    unsigned long long calls = 0, hdlc_calls = 0;   // run calls / K3 launches since the last drain
    bool pipeline = true;
    uint32_t *ctl = nullptr, *cand = nullptr;
    uint32_t *cand_first[NBUF] = {}, *cand_count[NBUF] = {};   // K2b -> K3
    uint32_t *frame_count = nullptr;
    int cand_K = 64;
    int32_t *counters = nullptr;
    int *maxval[HB] = {};                  // rotate with the history buffers
    int max_cur = 0, max_last = 0;
    gnuais_frame *frames = nullptr;
    float *d_taps = nullptr;
    MfmaTaps *d_mfma = nullptr;     // fir_sign_mfma.hip: the 48 central taps as integer Toeplitz operands (long tables)
    bool mfma_ok = false;
    MfmaTaps mfma_host;
    float mfma_eps_seen_u = 0.0f, mfma_eps_abs_u = 0.0f;     // its threshold in units of y' = floor(sum tq x / 256): per unit of max |x|, absolute
    int fir_mfma = 1;               // 1: long tables run their inner segments on the matrix pipe where the batch allows it
    // f1 on the device (gnuais_batch_drain_nmea): allocated on first use
    uint8_t *d_seq[2] = {nullptr, nullptr};
    char *d_text = nullptr;
    void *nmea_scratch = nullptr;
    char *d_msg = nullptr;          // gnuais_batch_drain_messages: lines, lengths, offsets, packed text
    uint32_t *d_word = nullptr;     // a few device words for counts read back by the drain-type calls
    // the vessel table carried on the device (gnuais_batch_vessel_table_*): one allocation, per-frame slot scratch
    void *vt = nullptr;
    uint32_t *vt_fslot = nullptr;
    uint32_t vt_slots = 0;
    int vt_capacity = 0;
    size_t d_msg_bytes = 0;
    size_t nmea_scratch_bytes = 0, d_text_bytes = 0;
    // gnuais_batch_stream_nmea: the frame ring exists NRING times (ring 0 is `frames` / `frame_count`
    // above until the first streaming call).  A ring is filled by K3; its formatter and the copy of its
    // text into pinned memory are queued behind that K3 at once, with every size taken on the device; the
    // text is handed out NRING - 1 calls later, which is the only thing the host ever waits for.  A call
    // is about 2.7 ms from its FIR to its text on the host (four chain stages, formatter, PCIe copy), so
    // about six of them have to be in flight for one to finish every 0.55 ms.
    static constexpr int NRING = 8;
    gnuais_frame *ring[NRING] = {};
    uint32_t *ring_count[NRING] = {};
    uint2 *ring_chunks[NRING] = {};             // K3's chunk table per ring (kernels.h: HdlcLaunch::chunks)
    int ring_runs[NRING] = {};                  // K3 launches into the ring since it became current
    int n_chunks = 0;
    int ring_cur = 0;
    bool streaming = false;
    hipStream_t s_post = nullptr, s_copy = nullptr, s_copy_own = nullptr;   // s_copy_own: the one created for it
    hipEvent_t e_fill[NRING] = {}, e_fmt[NRING] = {}, e_txt[NRING] = {};
    char *sd_text[NRING] = {};                  // device text per slot
    size_t sd_text_bytes[NRING] = {};
    char *sh_text[NRING] = {};                  // pinned host text per slot
    size_t sh_text_bytes[NRING] = {};
    uint32_t *sh_info = nullptr;                // pinned: [NRING][8]: format's 4 words, the ring's 4 counters
    int s_stage[NRING] = {};                    // 1: the slot's formatter is queued, its text not handed out yet
    size_t sh_text_want = 0;                    // pinned text buffers grow to this (learnt from the traffic)
    uint32_t *sd_info = nullptr;                // device: [NRING][8], what sh_info receives with the text
    bool copy_on_k3 = false;                    // experiment: the copy kernel on K3's stream as well
    int copy_wgs = 24;                          // waves of the device -> pinned copy (measured: 16 0.65, 24 0.62, 32 0.64, 64 0.79 ms per C3 step)
    uint8_t *sd_seq[2] = {nullptr, nullptr};    // per-channel sequence digit, carried on the device
    int sd_seq_cur = 0;
    unsigned long long stream_calls = 0;
    int16_t *stage_x = nullptr;
    size_t stage_bytes = 0;
    float *stage_f = nullptr;       // gnuais_batch_filter_host: the floats on their way out
    size_t stage_f_bytes = 0;
    // gnuais_batch_run_host_async: two pinned host buffers + two device buffers, one internal stream
    int16_t *pin[2] = {nullptr, nullptr}, *dev_in[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    hipStream_t s_io = nullptr;
    hipEvent_t e_in[2] = {nullptr, nullptr};    // the FIR of the call that used staging pair q is done
    hipEvent_t e_in_hook = nullptr;             // run_host_async -> run: record this right behind K1
    unsigned long long host_calls = 0;
    // options
    int fir_T = 512;
    int fir_map = 1;                            // K1s workgroup mapping (fir_slice.hip): XCD-contiguous channel groups
    int stage_mask = 0x1f;                      // experiments only: bit s = launch stage s
    int fir_variant = 3;            // 3 sign-exact slicer (default when the table allows);
                                    // 0 the exact ordered sum for every sample
    bool sign_ok = false;           // table is 32 symmetric effective taps: K1s applicable
    float sign_eps = 0.0f;
    float sign_eps_pk = 0.0f;       // the same for the packed transposed kernel's order of operations
    float sign_eps_seen = 0.0f, sign_eps_ahead = 0.0f;   // sign_eps split: what scales with the samples seen / what cannot
    bool pk40_ok = false;           // the table allows 40 central taps where sign_NC is 48 (fir_sign_pk.hip)
    float pk40_eps_pk = 0.0f, pk40_eps_seen = 0.0f, pk40_eps_ahead = 0.0f;
    // the packed kernel's bound per output position in its 16-row group: the group's later rows are under the running
    // maximum too, so an output that completes k rows before the group's end has k rows fewer "ahead" ([0..3]: k >= 6, 4, 2, 0)
    float pk_seen_k[2][4] = {}, pk_ahead_k[2][4] = {};      // [0]: 48 central taps, [1]: 40
    int fir_pk_taps = 0;            // 0: 40 where the table allows it; 48: never 40
    int fir_inloop = 1;             // 48-tap K1s: running window maximum in the loop (0: the per-segment pre-pass)
    int sign_NC = 12;               // central taps K1s evaluates
    int fir_flag2 = 1;              // the direct-form K1s gathers sign and threshold bit with one instruction per output (FL2)
    float sign_fscale = 0.0f;       //   the power of two its central taps are scaled by (0: the table does not allow it)
    int k0 = 0;                     // first effective tap
    int pll_variant = 0;            // 0: by channel count; 7 / 8 (kernels.h: PllLaunch::variant)
    int hdlc_lpw = 0;               // channels per wave in K2b; 0 = the variant's own default (16 event-driven, 64 bit-serial)
    int hdlc_variant = 1;           // 1: the event-driven deframer (hdlc_events.hip), 0: window by window (hdlc_crc.hip)
    bool timing = false;
    int timing_stride = 1;          // time every n-th call only: ten event records a call are not free
    // timing: a ring of per-call event sets so that kernel durations can be read back
    // for every call of a timed region, not just the last one
    static constexpr int TIMING_RING = 64;
    hipEvent_t evr[TIMING_RING][10] = {};  // 0,1 K1 | 2,6 K2 | 5,7 K2b | 9,4 K3
    unsigned long long timed_calls = 0;
    int last_k = 0;
    bool timed_last = false;
    hipStream_t last_stream = nullptr;
    int last_len = 0;
    // K3 on the deframer's stream: at ring lag 1 the two never overlap (deframer(i) -> K3(i) -> deframer(i+1)), so the two
    // cross-stream event waits per call in the loop that sets the period become stream order: 20-step 0.550 -> 0.544,
    // steady 0.527 -> 0.522 (three A/B pairs, profiles/r04_k3_on_the_deframers_stream.txt).  0 = a stream of its own.
    int k3_same = 1;
};

static int set_device(const gnuais_batch *b)
{
    HIP_TRY(hipSetDevice(b->device));
    return GNUAIS_OK;
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" {

const char *gnuais_last_error(void) { return g_err.c_str(); }
const char *gnuais_version(void) { return "gnuais-hip 0.1 (gfx950)"; }

int gnuais_default_taps(float *out36)
{
    if (!out36) return fail(GNUAIS_E_ARG, "gnuais_default_taps: NULL");
    for (int k = 0; k < 18; ++k) {
        out36[k] = (float) k_tap_half[k];
        out36[35 - k] = (float) k_tap_half[k];
    }
    return 36;
}

// hand-off sets [sets_alloc, n): sign words, bit packs and their counts, K2b -> K3 slot ranges (zeroed)
static hipError_t alloc_sets(gnuais_batch *b, int n)
{
    const size_t N = (size_t) b->N;
    for (int k = b->sets_alloc; k < n && k < gnuais_batch::NBUF; ++k) {
        struct { void **p; size_t bytes; } want[] = {
            {(void **) &b->sgn[k], sizeof(uint32_t) * sgn_words_alloc(b->sgn_words, b->N)},
            {(void **) &b->segbits[k], sizeof(uint32_t) * N * (size_t) b->n_seg * PACK_STRIDE},
            {(void **) &b->segcnt[k], sizeof(uint32_t) * N * (size_t) b->n_seg},
            {(void **) &b->cand_first[k], sizeof(uint32_t) * N},
            {(void **) &b->cand_count[k], sizeof(uint32_t) * N}};
        for (auto &w : want) {
            if (*w.p) continue;                 // left by an earlier attempt that failed further down this set
            hipError_t e = hipMalloc(w.p, w.bytes);
            if (e == hipSuccess) e = hipMemset(*w.p, 0, w.bytes);
            if (e != hipSuccess) {
                if (*w.p) (void) hipFree(*w.p);
                *w.p = nullptr;
                return e;
            }
        }
        b->sets_alloc = k + 1;
    }
    return hipSuccess;
}

void gnuais_batch_destroy(gnuais_batch *b)
{
    if (!b) return;
    (void) hipSetDevice(b->device);
    for (int q = 0; q < gnuais_batch::NBUF; ++q) {
        void *set[] = {b->sgn[q], b->segbits[q], b->segcnt[q], b->cand_first[q], b->cand_count[q]};
        for (void *p : set)
            if (p) (void) hipFree(p);
    }
    void *ptrs[] = {b->hist[0], b->hist[1], b->hist[2], b->hist[3], b->pll, b->lastbit, b->prev, b->ctl, b->cand,
                    b->frame_count, b->counters, b->maxval[0], b->maxval[1], b->maxval[2], b->maxval[3], b->frames, b->d_taps, b->d_mfma,
                    b->stage_x, b->d_seq[0], b->d_seq[1], b->d_text, b->nmea_scratch, b->d_msg, b->d_word, b->stage_f, b->vt, b->vt_fslot};
    for (void *p : ptrs)
        if (p) (void) hipFree(p);
    for (auto &set : b->evr)
        for (auto &e : set)
            if (e) (void) hipEventDestroy(e);
    for (auto &pair : b->e_done)
        for (auto &e : pair)
            if (e) (void) hipEventDestroy(e);
    for (auto &st : b->pool)
        if (st) (void) hipStreamDestroy(st);
    for (int q = 0; q < gnuais_batch::NRING; ++q) {
        if (q > 0 && b->ring[q]) (void) hipFree(b->ring[q]);          // ring 0 is frames / frame_count
        if (q > 0 && b->ring_count[q]) (void) hipFree(b->ring_count[q]);
        if (b->ring_chunks[q]) (void) hipFree(b->ring_chunks[q]);
        if (b->sd_text[q]) (void) hipFree(b->sd_text[q]);
        if (b->sh_text[q]) (void) hipHostFree(b->sh_text[q]);
        for (hipEvent_t e : {b->e_fill[q], b->e_fmt[q], b->e_txt[q]})
            if (e) (void) hipEventDestroy(e);
    }
    if (b->sh_info) (void) hipHostFree(b->sh_info);
    if (b->sd_info) (void) hipFree(b->sd_info);
    for (auto p : b->sd_seq)
        if (p) (void) hipFree(p);
    {   // the copy stream may have been replaced by a pool stream (autotune_delivery): destroy the one created for it
        hipStream_t st = b->s_copy_own ? b->s_copy_own : b->s_copy;
        if (st) (void) hipStreamDestroy(st);
    }
    for (int q = 0; q < 2; ++q) {
        if (b->pin[q]) (void) hipHostFree(b->pin[q]);
        if (b->dev_in[q]) (void) hipFree(b->dev_in[q]);
        if (b->e_in[q]) (void) hipEventDestroy(b->e_in[q]);
    }
    if (b->s_io) (void) hipStreamDestroy(b->s_io);
    delete b;
}

int gnuais_batch_create(gnuais_batch **out, int device, int n_channels, const float *taps,
                        int n_taps, unsigned pllinc, int max_len, int frame_capacity)
{
    if (!out) return fail(GNUAIS_E_ARG, "create: out is NULL");
    *out = nullptr;
    if (n_channels <= 0 || max_len <= 0) return fail(GNUAIS_E_ARG, "create: n_channels/max_len");
    if (taps && (n_taps <= 0 || n_taps > GNUAIS_MAX_TAPS)) return fail(GNUAIS_E_ARG, "create: n_taps");
    if (pllinc > 0xffffu) return fail(GNUAIS_E_ARG, "create: pllinc must be < 0x10000");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(GNUAIS_E_HIP, "create: no HIP device");
    if (device < 0 || device >= ndev) return fail(GNUAIS_E_ARG, "create: device index");
    HIP_TRY(hipSetDevice(device));
    {   // the PLL stage keeps its position lists and bit packs in LDS (PLL_NEED_LDS, about 116 KB per workgroup) and
        // is compiled for gfx950's wave and LDS sizes only
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess &&
            lds < pll_need_lds())
            return fail(GNUAIS_E_HIP, "create: the device offers too little LDS per workgroup (gfx950 / MI355X with "
                                      "160 KB is what this library is built for)");
    }
    HIP_TRY(pll_prepare_device());              // per device: the PLL stage's LDS reservation

    gnuais_batch *b = new gnuais_batch;
    b->device = device;
    if (hipDeviceGetAttribute(&b->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || b->n_cu <= 0)
        b->n_cu = 256;
    b->N = n_channels;
    if (taps) {
        b->taps.assign(taps, taps + n_taps);
    } else {
        b->taps.resize(36);
        gnuais_default_taps(b->taps.data());
    }
    b->NT = (int) b->taps.size();
    b->pllinc = pllinc ? pllinc : (0x10000u / 5u);     // receiver.c:69
    b->max_len = max_len;
    b->frame_cap = frame_capacity > 0 ? frame_capacity : std::max(4096, n_channels * 48);

    // trim exactly-zero taps at both ends (exact, see fir_slice.hip)
    int k0 = 0, k1 = b->NT - 1;
    while (k0 < b->NT - 1 && b->taps[k0] == 0.0f) ++k0;
    while (k1 > k0 && b->taps[k1] == 0.0f) --k1;
    b->NE = k1 - k0 + 1;
    b->d = b->NT - k0;
    b->k0 = k0;
    if (b->NE <= 128)
        for (int j = 0; j < b->NE; ++j) b->te[j] = b->taps[k0 + j];

    {
        // sign-exact slicer (fir_slice.hip K1s): error budget of the NC central taps against the
        // reference's ordered NE-term fp32 sum, for |x| <= 32768.  The smallest NC the kernel is
        // built for (12, 48) whose bound stays small enough is used: 12 for the reference table
        // (32 effective taps, bound 0.23), 48 for the 192 kHz table (126 taps, bound 0.87).  40 is
        // evaluated on the way as the packed kernel's alternative to 48 (fir_sign_pk.hip; the 192 kHz
        // table: bound 1.4, a sixth fewer multiply-adds).
        const int NE = b->NE;
        bool sym = NE <= 128;
        for (int j = 0; sym && j < NE; ++j) sym = memcmp(&b->te[j], &b->te[NE - 1 - j], 4) == 0;
        for (int NC : {12, 40, 48}) {
            if (!sym || b->sign_ok || NE < NC || (NE - NC) % 2) continue;
            const double u = 5.9604644775390625e-8, X = 32768.0;
            const int J0 = (NE - NC) / 2;
            // An ordered fp32 sum s_1 = fl(p_1), s_j = fl(s_{j-1} + fl(p_j)) of n products carries
            // product i with the factor (1+d_i) * prod_{j=max(i,2)..n} (1+e_j), |d|,|e| <= u: k_i = n
            // rounding factors for i = 1, n-i+2 for i >= 2 (Higham, Accuracy and Stability, sec. 4.2).
            // So |s_n - S| <= X * sum_i |t_i| * ((1+u)^k_i - 1): the late terms of the sum, and for a
            // bell-shaped table the big central ones are late enough, pass through few additions.
            // The reference adds in tap order (filter.h:40-49); K1s's accumulators take their NC
            // products in sample order, which is tap order from one edge of the centre to the other
            // (either edge: the table is symmetric).
            auto ordered = [&](int first, int n) {
                double e = 0;
                for (int i = 1; i <= n; ++i)
                    e += std::fabs((double) b->te[first + i - 1]) * (std::pow(1 + u, i == 1 ? n : n - i + 2) - 1);
                return e;
            };
            double sum_out = 0;
            for (int j = 0; j < NE; ++j)
                if (j < J0 || j >= J0 + NC) sum_out += std::fabs((double) b->te[j]);
            // + NE subnormal products, each off by at most 2^-150 (absolute)
            // NC = 12 is evaluated in direct form: s_q = x_a + x_b (exact: both are int16-valued), then
            // y = fl(t_0 s_0), y = fl(y + fl(t_q s_q)) for q = 1..NC/2-1, edge taps first; |s_q| <= 2X
            auto paired = [&]() {
                double e = 0;
                const int n = NC / 2;
                for (int i = 1; i <= n; ++i)
                    e += 2.0 * std::fabs((double) b->te[J0 + i - 1]) * (std::pow(1 + u, i == 1 ? n : n - i + 2) - 1);
                return e;
            };
            const double central = K1S_DIRECT(NC) ? paired() : ordered(J0, NC);
            const double bound = X * (ordered(0, NE) + central + sum_out) + 1e-30;
            if (NC == 40) {
                // the packed kernel only (running window maximum: its window behind a group is 96 rows)
                if (std::isfinite(bound) && bound < 2.0 && NC - 1 + J0 <= 96) {
                    double ahead = 0;
                    for (int i = J0 + NC + 1; i <= NE; ++i)
                        ahead += std::fabs((double) b->te[i - 1]) * (1.0 + (std::pow(1 + u, NE - i + 2) - 1));
                    b->pk40_ok = true;
                    b->pk40_eps_pk = (float) (bound * 1.1);
                    b->pk40_eps_ahead = (float) (X * ahead * 1.1 + 1e-30);
                    b->pk40_eps_seen = (float) ((bound - X * ahead) * 1.1);
                    for (int q = 0; q < 4; ++q) {
                        double ah = 0;
                        for (int i = J0 + NC + 1 + (6 - 2 * q); i <= NE; ++i)
                            ah += std::fabs((double) b->te[i - 1]) * (1.0 + (std::pow(1 + u, NE - i + 2) - 1));
                        b->pk_ahead_k[1][q] = (float) (X * ah * 1.1 + 1e-30);
                        b->pk_seen_k[1][q] = (float) ((bound - X * ah) * 1.1);
                    }
                }
                continue;
            }
            if (std::isfinite(bound) && bound < 2.0) {
                b->sign_eps = (float) (bound * 1.1);
                b->sign_eps_pk = (float) ((X * (ordered(0, NE) + ordered(J0, NC) + sum_out) + 1e-30) * 1.1);   // transposed sum (fir_sign_pk.hip)
                b->sign_NC = NC;
                b->sign_ok = true;
                // The same bound in two parts, for the kernel that scales it with the largest |x| it has SEEN (the
                // 48-tap instantiation, fir_slice.hip): the last J0 taps of a reference window multiply samples
                // that lie up to J0 rows beyond the newest one the central sum has loaded; their share of the
                // bound keeps X = 32768 (it is tiny: those taps are) and everything else scales with the maximum
                // over the rows behind.  bound = part_seen + part_ahead.
                double ahead = 0;
                for (int i = J0 + NC + 1; i <= NE; ++i)      // 1-based tap index, as in ordered()
                    ahead += std::fabs((double) b->te[i - 1]) * (1.0 + (std::pow(1 + u, NE - i + 2) - 1));
                b->sign_eps_ahead = (float) (X * ahead * 1.1 + 1e-30);
                b->sign_eps_seen = (float) ((bound - X * ahead) * 1.1);
                for (int q = 0; q < 4; ++q) {
                    double ah = 0;
                    for (int i = J0 + NC + 1 + (6 - 2 * q); i <= NE; ++i)
                        ah += std::fabs((double) b->te[i - 1]) * (1.0 + (std::pow(1 + u, NE - i + 2) - 1));
                    b->pk_ahead_k[0][q] = (float) (X * ah * 1.1 + 1e-30);
                    b->pk_seen_k[0][q] = (float) ((bound - X * ah) * 1.1);
                }
                if (NC == 48 && J0 <= 48) {
                    // fir_sign_mfma.hip: the central sum in exact integer arithmetic on quantised taps -- the bound is the
                    // reference's own rounding + the omitted taps + the quantisation, ALL of it per unit of the largest
                    // |x| in reach of a window (its running maximum covers the rows behind and ahead), + 1 for the floor
                    double S = 0, bq = 0;
                    if (fir_sign_mfma_taps(&b->te[J0], &b->mfma_host, &S, &bq)) {
                        const double rel = ordered(0, NE) + sum_out + bq;          // per unit of |x|, in units of y
                        if (X * rel < 2.0) {
                            b->mfma_ok = true;
                            b->mfma_eps_seen_u = (float) (rel * (S / 256.0) * 1.1);
                            b->mfma_eps_abs_u = 3.0f;
                        }
                    }
                }
                // FL2 (fir_sign_kernel): the direct form's central taps times k = 2 / P, P = the power of two at or above
                // eps.  k is a power of two >= 1, so every product, pre-add and partial sum of the scaled evaluation is
                // exactly k times the unscaled one (nothing overflows: |y'| <= 2 X sum|t| / eps < 1e9; an underflow the
                // unscaled sum has, the scaled one has at most as badly) and |y_c| < P  <=>  |y'| < 2  <=>  exponent
                // bit 7 of y' clear.  P >= eps: the band only widens.  Not taken when a central tap is subnormal or k
                // would leave [1, 2^60].
                auto flag_scale = [&](float eps, int nc) -> float {
                    if (!(eps > 0.0f) || !(eps <= 2.0f) || !K1S_DIRECT(nc)) return 0.0f;
                    int e = 0;
                    const float m = std::frexp(eps, &e);            // eps = m 2^e, m in [0.5, 1)
                    const float P = std::ldexp(1.0f, m == 0.5f ? e - 1 : e);
                    const float k = 2.0f / P;
                    if (!(k >= 1.0f) || !(k <= 1.152921504606846976e18f)) return 0.0f;
                    for (int j = 0; j < nc; ++j) {
                        const float t = b->te[(NE - nc) / 2 + j];
                        if (t != 0.0f && (!std::isnormal(t) || !std::isnormal(t * k))) return 0.0f;
                    }
                    return k;
                };
                if (NC == 12) {
                    b->sign_fscale = flag_scale(b->sign_eps, 12);
                }
            }
        }
    }
    b->sgn_words = (max_len + 31) / 32;
    // at most one slice per sample step of (pllinc + pllinc/16)/65536
    const uint64_t step = (uint64_t) b->pllinc + b->pllinc / 16;
    b->bits_words = (int) (((uint64_t) max_len * step / 65536 + 2 + 31) / 32) + 1;

    const size_t N = (size_t) b->N;
    hipError_t e = hipSuccess;
    b->n_seg = (b->sgn_words + SEG_WORDS - 1) / SEG_WORDS;
    b->cand_K = std::max(64, b->bits_words * 32 / 30 + 2);
    if (const char *v = getenv("GNUAIS_NBUF")) b->nbuf = std::min((int) gnuais_batch::NBUF, std::max(2, atoi(v)));
    {   // preflight: what this batch is about to allocate against what the device has free -- a batch that does not fit
        // says so with both figures at once instead of failing somewhere inside a dozen allocations
        const size_t per_set = sizeof(uint32_t) * (sgn_words_alloc(b->sgn_words, b->N) + N * (size_t) b->n_seg * (PACK_STRIDE + 1) + 2 * N);
        const size_t need = gnuais_batch::HB * (sizeof(int16_t) * N * b->NT + sizeof(int) * N) + (size_t) b->nbuf * per_set +
                            sizeof(uint32_t) * N * ((size_t) b->cand_K * CAND_WORDS + HDLC_CTL_WORDS + 6) +
                            sizeof(gnuais_frame) * (size_t) b->frame_cap;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b) {
            char msg[256];
            snprintf(msg, sizeof msg, "create: %d channels x %d samples need %.2f GB of device memory, device %d has %.2f GB free of %.2f",
                     b->N, b->max_len, need / 1e9, device, free_b / 1e9, total_b / 1e9);
            delete b;
            return fail(GNUAIS_E_HIP, msg);
        }
    }
    auto alloc = [&](void **p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
        if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
    };
    for (int q = 0; q < gnuais_batch::HB; ++q) alloc((void **) &b->hist[q], sizeof(int16_t) * N * b->NT);
    b->seg_words = (int) (((uint64_t) SEG_WORDS * 32 * step / 65536 + 2 + 31) / 32) + 1;
    if (b->seg_words > 16) {       // K2b keeps one segment pack (<= 16 words) in registers
        gnuais_batch_destroy(b);
        return fail(GNUAIS_E_ARG, "create: pllinc too large (more than one slice per ~4.6 samples)");
    }
    // the hand-off sets in use (`nbuf`; more are allocated when set_option raises it: a C5 set is 0.4 GB of sign words)
    if (e == hipSuccess) e = alloc_sets(b, b->nbuf);
    alloc((void **) &b->pll, sizeof(uint32_t) * N);
    alloc((void **) &b->lastbit, sizeof(uint32_t) * N);
    alloc((void **) &b->prev, sizeof(uint32_t) * N);
    alloc((void **) &b->ctl, sizeof(uint32_t) * N * HDLC_CTL_WORDS);
    // candidate ring, per channel and call.  The deframer cannot open frames faster than one per
    // 30 bits (16 alternating bits to leave ST_SKURR, protodec.c:1030-1043, six ones each for the
    // opening and the closing flag, a bit in ST_STOPSIGN), so this many slots hold whatever a call
    // can produce, adversarial bit streams included (real traffic: <= 38 frames per second)
    alloc((void **) &b->cand, sizeof(uint32_t) * N * (size_t) b->cand_K * CAND_WORDS);
    alloc((void **) &b->frame_count, sizeof(uint32_t) * 4);
    alloc((void **) &b->counters, sizeof(int32_t) * N * 3);
    for (int q = 0; q < gnuais_batch::HB; ++q) alloc((void **) &b->maxval[q], sizeof(int) * N);
    alloc((void **) &b->frames, sizeof(gnuais_frame) * (size_t) b->frame_cap);
    alloc((void **) &b->d_taps, sizeof(float) * b->NT);
    if (b->mfma_ok) {
        alloc((void **) &b->d_mfma, sizeof(MfmaTaps));
        if (e == hipSuccess) e = hipMemcpy(b->d_mfma, &b->mfma_host, sizeof(MfmaTaps), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess)
        e = hipMemcpy(b->d_taps, b->taps.data(), sizeof(float) * b->NT, hipMemcpyHostToDevice);
    for (auto &set : b->evr)
        for (auto &ev : set)
            if (e == hipSuccess) e = hipEventCreate(&ev);
    for (auto &pair : b->e_done)
        for (auto &ev : pair)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    {
        // the sequential stages are short on parallelism, long on latency: give them
        // dispatch priority over the FIR's tens of thousands of workgroups
        int lo = 0, hi = 0;
        if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        // Creation order matters (measured, scripts/alloc_experiment.py + scripts/queue_ids.py):
        // HIP hands out hardware queues in creation order and the queues are spread over the
        // compute pipes round-robin, so in a process with no other streams the 4th stream
        // created here shares its pipe with the caller's (FIR) queue.  With the PLL stage or
        // the deframer there a C3 call takes 0.81-0.83 ms, with K3 there 0.90-0.93 (K3 is the
        // stage whose completion the host waits on before it reuses a hand-off set), with the spare
        // stream there 0.87.  That was with four streams per batch; with the twelve of the pool below the picture turns
        // (round 6, four fresh processes, C3: the PLL stage on the fourth stream 0.65-0.67 ms per step, on the first 0.503 =
        // what gnuais_batch_autotune() finds, 0.498-0.505; profiles/r06_stream_assignment.txt).  Hence: the PLL stage first,
        // then the spare, K2b, K3.  GNUAIS_STREAM_ORDER (four digits, stage indices 0 = K2 .. 3 = K3 in creation order)
        // overrides, for processes that have created streams of their own before.
        const char *order = getenv("GNUAIS_STREAM_ORDER");
        if (!order || strlen(order) != 4) order = "0123";
        int made = 0;
        for (int q = 0; q < 4; ++q) {
            const int idx = (order[q] - '0') & 3;
            if (e == hipSuccess && !b->s_k[idx]) {
                e = hipStreamCreateWithPriority(&b->pool[made], hipStreamNonBlocking, hi);
                b->s_k[idx] = b->pool[made++];
            }
        }
        for (auto &st : b->s_k)
            if (e == hipSuccess && !st) {
                e = hipStreamCreateWithPriority(&b->pool[made], hipStreamNonBlocking, hi);
                st = b->pool[made++];
            }
        for (int q = 0; q < 4; ++q) b->s_k_default[q] = b->s_k[q];
        // spare candidates for gnuais_batch_autotune(): in a process that has created streams of its
        // own the default assignment can be 1.7x slower than the best one (0.86 vs 1.39-1.43 ms
        // per C3 call with two application streams), and there is no API to ask which queue a
        // stream got -- so the assignment can be measured instead
        for (; made < gnuais_batch::POOL; ++made)
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&b->pool[made], hipStreamNonBlocking, made < 8 ? hi : 0);
    }
    if (const char *v = getenv("GNUAIS_K3_SAME")) b->k3_same = atoi(v) != 0;
    if (const char *v = getenv("GNUAIS_PLL_VARIANT")) {      // the values set_option takes, nothing else
        const int pv = atoi(v);
        if (pv == 0 || pv == 7 || pv == 8) b->pll_variant = pv;
    }
    if (const char *v = getenv("GNUAIS_PIPELINE")) b->pipeline = atoi(v) != 0;
    if (const char *v = getenv("GNUAIS_HDLC_VARIANT")) b->hdlc_variant = atoi(v) != 0;
    if (const char *v = getenv("GNUAIS_HDLC_LPW")) b->hdlc_lpw = std::min(64, std::max(1, atoi(v)));
    if (e != hipSuccess) {
        gnuais_batch_destroy(b);
        return fail(GNUAIS_E_HIP, "create: device allocation", e);
    }
    if (const char *v = getenv("GNUAIS_FIR_VARIANT")) {
        const int fv = atoi(v);
        if (fv == 0 || fv == 3) b->fir_variant = fv;
    }
    if (const char *v = getenv("GNUAIS_FIR_FLAG2")) b->fir_flag2 = atoi(v) != 0;
    if (const char *v = getenv("GNUAIS_FIR_T")) b->fir_T = std::max(64, atoi(v) / 32 * 32);
    *out = b;
    int rc = gnuais_batch_reset(b);
    if (rc != GNUAIS_OK) {
        gnuais_batch_destroy(b);
        *out = nullptr;
    }
    return rc;
}

int gnuais_batch_reset(gnuais_batch *b)
{
    if (!b) return fail(GNUAIS_E_ARG, "reset: NULL batch");
    if (int rc = set_device(b)) return rc;
    const size_t N = (size_t) b->N;
    HIP_TRY(hipDeviceSynchronize());
    for (int q = 0; q < gnuais_batch::HB; ++q)
        HIP_TRY(hipMemset(b->hist[q], 0, sizeof(int16_t) * N * b->NT));   // filter.c:62
    b->hist_cur = 0;
    HIP_TRY(hipMemset(b->pll, 0, sizeof(uint32_t) * N));              // receiver.c:66-71
    HIP_TRY(hipMemset(b->lastbit, 0, sizeof(uint32_t) * N));
    HIP_TRY(hipMemset(b->prev, 0, sizeof(uint32_t) * N));
    for (int k = 0; k < b->sets_alloc; ++k)
        HIP_TRY(hipMemset(b->segcnt[k], 0, sizeof(uint32_t) * N * (size_t) b->n_seg));
    b->calls = 0;
    b->hdlc_calls = 0;
    HIP_TRY(hipMemset(b->counters, 0, sizeof(int32_t) * N * 3));      // protodec.c:62-64
    for (int q = 0; q < gnuais_batch::HB; ++q) HIP_TRY(hipMemset(b->maxval[q], 0, sizeof(int) * N));
    b->max_cur = 0;
    b->max_last = 0;
    HIP_TRY(hipMemset(b->frame_count, 0, sizeof(uint32_t) * 4));
    for (int q = 1; q < gnuais_batch::NRING; ++q)
        if (b->ring_count[q]) HIP_TRY(hipMemset(b->ring_count[q], 0, sizeof(uint32_t) * 4));
    for (auto p : b->sd_seq)
        if (p) HIP_TRY(hipMemset(p, 0, N));
    for (int q = 0; q < gnuais_batch::NRING; ++q) { b->s_stage[q] = 0; b->ring_runs[q] = 0; }
    b->ring_cur = 0;
    b->stream_calls = 0;
    HIP_TRY(launch_hdlc_reset(b->ctl, b->N, nullptr));                // protodec.c:87-100
    HIP_TRY(hipDeviceSynchronize());
    b->last_len = 0;
    return GNUAIS_OK;
}

int gnuais_batch_set_option(gnuais_batch *b, const char *name, int value)
{
    if (!b || !name) return fail(GNUAIS_E_ARG, "set_option: NULL");
    if (!strcmp(name, "stage_mask")) {          // timing experiments; results are invalid if != 0x1f
        b->stage_mask = value & 0x1f;
        return GNUAIS_OK;
    }
    if (!strcmp(name, "fir_T")) {
        if (value < 64 || value % 32) return fail(GNUAIS_E_ARG, "fir_T must be a multiple of 32, >= 64");
        b->fir_T = value;
    } else if (!strcmp(name, "nbuf")) {             // hand-off sets in use: the calls that may be in flight
        if (value < 2 || value > gnuais_batch::NBUF) return fail(GNUAIS_E_ARG, "nbuf must be 2..8");
        if (int rc = gnuais_batch_sync(b)) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(alloc_sets(b, value));
        b->nbuf = value;
    } else if (!strcmp(name, "streaming")) {
        // 0: leave the streamed delivery (gnuais_batch_stream_nmea / autotune_delivery switch it on): everything in
        // flight is flushed and DROPPED, K3 goes back to ring 0, the drain-type calls work again.  The rings, texts
        // and streams stay allocated for the next streaming call.
        if (value != 0) return fail(GNUAIS_E_ARG, "streaming can only be switched off here (stream_nmea switches it on)");
        if (b->streaming) {
            if (int rc = gnuais_batch_sync(b)) return rc;
            HIP_TRY(hipDeviceSynchronize());
            for (int q = 0; q < gnuais_batch::NRING; ++q) {
                if (b->ring_count[q]) HIP_TRY(hipMemset(b->ring_count[q], 0, sizeof(uint32_t) * 4));
                b->s_stage[q] = 0;
                b->ring_runs[q] = 0;
            }
            b->ring_cur = 0;
            b->stream_calls = 0;
            b->hdlc_calls = 0;
            b->streaming = false;
        }
    } else if (!strcmp(name, "fir_flag2")) {         // 0: |y| - eps and two alignbits per output (rounds 1-4)
        if (value != 0 && value != 1) return fail(GNUAIS_E_ARG, "fir_flag2: 0 or 1");
        b->fir_flag2 = value;
    } else if (!strcmp(name, "fir_pk_taps")) {       // long tables: 0 = 40 central taps where the table allows them, 48 = never 40
        if (value != 0 && value != 48) return fail(GNUAIS_E_ARG, "fir_pk_taps: 0 or 48");
        b->fir_pk_taps = value;
    } else if (!strcmp(name, "fir_mfma")) {          // long tables: 1 = inner segments on the matrix pipe (fir_sign_mfma.hip), 0 = the packed kernel throughout
        if (value != 0 && value != 1) return fail(GNUAIS_E_ARG, "fir_mfma: 0 or 1");
        b->fir_mfma = value;
    } else if (!strcmp(name, "fir_variant")) {
        if (value != 0 && value != 3) return fail(GNUAIS_E_ARG, "fir_variant must be 0 (the exact sum for every sample) or 3 (the sign-exact slicer)");
        b->fir_variant = value;
    } else if (!strcmp(name, "timing_stride")) {
        if (value < 1) return fail(GNUAIS_E_ARG, "timing_stride must be >= 1");
        b->timing_stride = value;
    } else if (!strcmp(name, "pipeline")) {
        b->pipeline = value != 0;

    } else if (!strcmp(name, "pll_variant")) {
        if (value != 0 && value != 7 && value != 8)
            return fail(GNUAIS_E_ARG, "pll_variant must be 0 (by channel count), 7 (time-parallel, pll_tp.hip) or 8 (pll_h3.hip)");
        b->pll_variant = value;
    } else if (!strcmp(name, "hdlc_variant")) {
        b->hdlc_variant = value != 0;
    } else if (!strcmp(name, "hdlc_lpw")) {
        if (value < 1 || value > 64) return fail(GNUAIS_E_ARG, "hdlc_lpw must be 1..64");
        b->hdlc_lpw = value;
    } else {
        return fail(GNUAIS_E_ARG, "set_option: unknown option");
    }
    return GNUAIS_OK;
}

static void fill_fir(const gnuais_batch *b, FirLaunch &f, const int16_t *x, int len, float *dump,
                     int k)
{
    memset(&f, 0, sizeof f);
    f.x = x;
    f.hist = b->hist[b->hist_cur];
    f.sgn = b->sgn[k];
    f.dump = dump;
    f.maxval = b->maxval[b->max_cur];
    f.hist_out = b->hist[(b->hist_cur + 1) % gnuais_batch::HB];
    f.maxval_next = b->maxval[(b->max_cur + 2) % gnuais_batch::HB];
    f.d_taps = b->d_taps;
    memcpy(f.te, b->te, sizeof f.te);
    f.N = b->N;
    f.L = len;
    f.T = b->fir_T;
    f.NT = b->NT;
    f.NE = b->NE;
    f.d = b->d;
    f.eps = b->sign_eps;
    f.eps_pk = b->sign_eps_pk;
    f.eps_seen = b->fir_inloop ? b->sign_eps_seen : 0.0f;
    f.eps_ahead = b->sign_eps_ahead;
    f.NC = b->sign_NC;
    if (b->sign_ok)
        for (int j = 0; j < b->sign_NC; ++j) f.ctaps[j] = b->te[(b->NE - b->sign_NC) / 2 + j];
    f.te_mem = b->d_taps + b->k0;
    f.map = b->fir_map;
}

static void fill_hdlc(const gnuais_batch *b, HdlcLaunch &h, int k)
{
    h.segbits = b->segbits[k]; h.segcnt = b->segcnt[k]; h.ctl = b->ctl; h.cand = b->cand;
    h.cand_first = b->cand_first[k]; h.cand_count = b->cand_count[k];
    h.counters = b->counters;
    h.frames = b->streaming ? b->ring[b->ring_cur] : b->frames;
    h.frame_count = b->streaming ? b->ring_count[b->ring_cur] : b->frame_count;
    h.frame_cap = (uint32_t) b->frame_cap; h.N = b->N; h.n_seg = b->n_seg;
    h.seg_words = b->seg_words; h.K = b->cand_K; h.K_call = b->cand_K;
    // event-driven deframer: 16 channels per wave finish a small batch soonest; where the chip is full anyway (a PLL
    // workgroup on every CU, four FIR waves per SIMD) 64 per wave -- a quarter of the waves -- cost the other stages
    // least: C3 steady state 0.492 against 0.525 ms per call (profiles/r05_pll_h3_in_the_pipeline.txt)
    // small batches: a lane's walk through its events is the stage's latency, and the fewer lanes share a wave the less
    // they wait for each other -- as few per wave as keep the launch at <= 512 waves (256 channels: 0.148 ms with one
    // lane per wave against 0.243 with 16; 1024 channels: 0.180 with two, 0.216 with one; time_pll_forms.py, round 5)
    int small_lpw = 1;
    while (small_lpw < 16 && (b->N + small_lpw - 1) / small_lpw > 512) small_lpw *= 2;
    if (b->N > PLL_TP_MAX_CHANNELS) small_lpw = 16;      // beside the lane-per-channel PLL more deframer waves cost more than they save (2048 channels: 0.346 against 0.336 ms per call)
    const int ev_lpw = 2 * ((b->N + 63) / 64) > (b->n_cu > 0 ? b->n_cu : 256) ? 64 : small_lpw;
    h.lanes_per_wave = b->hdlc_lpw ? b->hdlc_lpw : (b->hdlc_variant ? ev_lpw : 64);
    // the chunk table describes ONE launch; a second one into the same ring would overwrite it
    h.chunks = (b->streaming && b->ring_runs[b->ring_cur] == 0) ? b->ring_chunks[b->ring_cur] : nullptr;
}

// the stream K3 runs on (and everything that has to come behind the last K3)
static hipStream_t k3_stream(const gnuais_batch *b)
{
    // not while the batch is streaming: K3 then waits for the delivery side (a frame ring to come free), and on the
    // deframer's stream that wait would hold the next deframer launch too (0.89 against 0.62 ms per delivered step)
    return (b->k3_same && !b->streaming) ? b->s_k[2] : b->s_k[3];
}

// K1 + carry.  The specialised kernel updates the history and clears the next peak
// buffer itself; the generic fallback needs the two helper launches.
static int run_fir(gnuais_batch *b, const int16_t *x, int len, float *dump, hipStream_t s, int k)
{
    FirLaunch f;
    fill_fir(b, f, x, len, dump, k);
    if (b->fir_variant == 3 && b->sign_ok && !dump) {
        const int q = launch_fir_sign_quantum(f.NC);        // whole loop turns of the kernel's unrolled body
        f.T = std::min((f.T + q - 1) / q * q, 65280 / q * q);       // K1s notes open outputs as 16-bit offsets into the segment
        // 48 central taps (the 192 kHz table): the transposed sum on register pairs (fir_sign_pk.hip)
        if (f.NC == 48 && b->pk40_ok && b->fir_pk_taps != 48 && f.eps_seen > 0.0f) {
            f.NC = 40;
            f.eps_pk = b->pk40_eps_pk;
            f.eps_seen = b->pk40_eps_seen;
            f.eps_ahead = b->pk40_eps_ahead;
            for (int j = 0; j < 40; ++j) f.ctaps[j] = b->te[(b->NE - 40) / 2 + j];
        }
        for (int q = 0; q < 4; ++q) {
            f.eps_seen_k[q] = b->pk_seen_k[f.NC == 40][q];
            f.eps_ahead_k[q] = b->pk_ahead_k[f.NC == 40][q];
        }
        if ((f.NC == 48 || f.NC == 40) && f.NC - 1 + (f.NE - f.NC) / 2 <= 96 && f.eps_seen > 0.0f) {
            const int qp = launch_fir_sign_pk_quantum(f.NC);
            // 48 taps: a segment's warm-up is 47 pair steps' worth of samples; longer segments (there are plenty of
            // waves: 16384 x 192000 is 16000 segments of 3072) cut its share (round 4: 3072 against 1536, 4.29 against
            // 4.39 ms per C5 call in steady state, profiles/r04_c5_ring_and_segments.txt)
            // 40 taps: 1920 (FIR alone 3.34-3.40 ms against 3.43-3.49 at 3200 and 3.85 at 6400: the lists of open outputs a
            // segment settles at its end grow with it; profiles/r05_c5_forty_central_taps.txt)
            f.T = ((b->fir_T <= 768 ? (f.NC == 40 ? 1920 : 3072) : b->fir_T) + qp - 1) / qp * qp;
            f.T = std::min(f.T, 65280 / qp * qp);           // the kernel notes open outputs as 16-bit offsets into the segment
            // The matrix-pipe kernel takes everything but the call's head -- the outputs whose windows reach into the history
            // rows --, which stays the packed kernel's: the fewest whole packed loop turns (and whole 16-byte sign stores) that
            // cover d and dc + 64 rows (640 outputs for the 192 kHz table: one wave per 64 channels walks it alone, 0.1 ms for
            // 256 waves on 1024 SIMDs; round 5's whole first segment of 1920 took 0.3).  (Beside the matrix-pipe launch on a side
            // stream, between two events: 2.44 instead of 2.51 ms per C5 call, but one pipelined run in six came out with a frame
            // more or less -- not kept; profiles/r06_c5_matrix_pipe_k32.txt.)
            const int dc48 = f.d - (f.NE - 48) / 2;
            const int qh = qp % launch_fir_sign_mfma_quantum() == 0 ? qp : qp * launch_fir_sign_mfma_quantum();
            const int head = (std::max(dc48 + 64, f.d) + qh - 1) / qh * qh;
            const bool mfma = b->fir_mfma && b->mfma_ok && b->N % 64 == 0 && f.T % launch_fir_sign_mfma_quantum() == 0 && len > head &&
                              len >= b->NT && head <= 65280 &&
                              (unsigned long long) (f.T + f.NE + 512) * (unsigned long long) b->N * 2ull < 0x7fffffffull;
            if (mfma) {
                FirLaunch h = f, m = f;
                h.T = head;
                h.max_segments = 1;
                m.NC = 48;
                m.mfma = b->d_mfma;
                m.eps_seen = b->mfma_eps_seen_u;
                m.eps_ahead = b->mfma_eps_abs_u;
                HIP_TRY(launch_fir_sign_pk(h, s));
                HIP_TRY(launch_fir_sign_mfma(m, head, s));
            } else {
                HIP_TRY(launch_fir_sign_pk(f, s));
            }
            b->hist_cur = (b->hist_cur + 1) % gnuais_batch::HB;
            b->max_last = b->max_cur;
            b->max_cur = (b->max_cur + 1) % gnuais_batch::HB;
            return GNUAIS_OK;
        }
        if (f.NC == 12 && b->fir_flag2) f.fscale = b->sign_fscale;
        HIP_TRY(launch_fir_sign(f, s));
    } else if (b->NE != 32) {
        HIP_TRY(hipMemsetAsync(b->maxval[(b->max_cur + 2) % gnuais_batch::HB], 0, sizeof(int) * (size_t) b->N, s));
        HIP_TRY(launch_fir_generic(f, s));
        HIP_TRY(launch_fir_history(x, b->hist[b->hist_cur], b->hist[(b->hist_cur + 1) % gnuais_batch::HB], b->N, len,
                                   b->NT, s));
    } else {
        HIP_TRY(scalar::launch_fir_slice(f, s));
    }
    b->hist_cur = (b->hist_cur + 1) % gnuais_batch::HB;
    b->max_last = b->max_cur;
    b->max_cur = (b->max_cur + 1) % gnuais_batch::HB;
    return GNUAIS_OK;
}

static void fill_pll(const gnuais_batch *b, PllLaunch &p, int k, int len)
{
    p.sgn = b->sgn[k]; p.pll = b->pll; p.prev = b->prev;
    p.watchdog = (b->streaming ? b->ring_count[b->ring_cur] : b->frame_count) + 3; p.lastbit = b->lastbit;
    p.segbits = b->segbits[k]; p.segcnt = b->segcnt[k];
    p.N = b->N; p.L = len; p.n_seg = b->n_seg; p.seg_words = b->seg_words; p.pllinc = b->pllinc; p.variant = b->pll_variant;
    p.n_cu = b->n_cu;
}


// K2b, K3 of one call, each on its own stream (pipeline) or all on s0, after `after`
// (the event that says this call's PLL stage is done; null = stream order on s0).
static int run_tail(gnuais_batch *b, int k, int len, bool tm, hipEvent_t *ev,
                    hipStream_t s0, hipEvent_t after)
{
    const bool pl = b->pipeline;
    hipStream_t sC = pl ? b->s_k[2] : s0, sD = pl ? k3_stream(b) : s0;
    HdlcLaunch h;
    fill_hdlc(b, h, k);
    // K2b: needs segbits[k]; fills cand_first/count[k] (read by K3 of call - NBUF).  It also writes
    // the per-channel candidate ring that K3 of the PREVIOUS call may still be reading: slots are
    // reused after cand_K frame starts, which one call cannot exceed but two could
    if (pl && after) HIP_TRY(hipStreamWaitEvent(sC, after, 0));
    if (pl && b->calls >= 1 && sD != sC)
        HIP_TRY(hipStreamWaitEvent(sC, b->e_done[4][(k + b->nbuf - 1) % b->nbuf], 0));
    if (tm) HIP_TRY(hipEventRecord(ev[5], sC));
    if (b->stage_mask & 8) HIP_TRY(b->hdlc_variant ? launch_hdlc_events(h, sC) : launch_hdlc_deframe(h, sC));
    if (tm) HIP_TRY(hipEventRecord(ev[7], sC));
    if (pl) HIP_TRY(hipEventRecord(b->e_done[3][k], sC));
    // K3
    if (pl && sD != sC) HIP_TRY(hipStreamWaitEvent(sD, b->e_done[3][k], 0));
    if (tm) HIP_TRY(hipEventRecord(ev[9], sD));
    if (b->stage_mask & 16) HIP_TRY(launch_hdlc_crc(h, sD));
    if (b->streaming) b->ring_runs[b->ring_cur]++;
    b->hdlc_calls++;
    if (tm) HIP_TRY(hipEventRecord(ev[4], sD));
    if (pl) HIP_TRY(hipEventRecord(b->e_done[4][k], sD));
    return GNUAIS_OK;
}

int gnuais_batch_run(gnuais_batch *b, const int16_t *d_samples, int len, void *stream)
{
    if (!b || !d_samples) return fail(GNUAIS_E_ARG, "run: NULL argument");
    if (len <= 0 || len > b->max_len) return fail(GNUAIS_E_ARG, "run: len out of range (max_len)");
    if (int rc = set_device(b)) return rc;
    hipStream_t s0 = (hipStream_t) stream;
    const bool pl = b->pipeline;
    const int k = (int) (b->calls % (unsigned) b->nbuf);      // hand-off buffer set of this call
    const bool reuse = pl && b->calls >= (unsigned) b->nbuf;  // set k last used by call i-nbuf
    const bool tm = b->timing && (b->calls % (unsigned) b->timing_stride) == 0;
    hipEvent_t *ev = b->evr[b->timed_calls % gnuais_batch::TIMING_RING];

    {
        hipStream_t sA = pl ? b->s_k[0] : s0;
        // Hand-off set k was last used by call i-nbuf.  Its last user is that call's K3; wait for
        // it on the HOST (normally long done): five stream-wait packets per call, each ~20 us of
        // queue time on the stream it sits in, for a condition that is practically always true.
        // A caller that runs more than nbuf-1 calls ahead of the device blocks here.  (Round 4, measured once more at
        // depth 3, where this wait IS the loop: ONE wait packet on the caller's stream instead, the host held back only
        // by that call's FIR launch -- 0.583 against 0.542 ms per step, profiles/r04_k3_on_the_deframers_stream.txt.)
        if (reuse) HIP_TRY(hipEventSynchronize(b->e_done[4][k]));
        // K1 carries the FIR history and the peak buffers from call to call in stream order: a caller
        // that changes streams between calls gets the old stream drained first
        if (b->calls > 0 && s0 != b->last_stream) HIP_TRY(hipStreamSynchronize(b->last_stream));
        hipStream_t sF = s0;
        if (tm) HIP_TRY(hipEventRecord(ev[0], sF));
        if (b->stage_mask & 1)
            if (int rc = run_fir(b, d_samples, len, nullptr, sF, k)) return rc;
        if (tm) HIP_TRY(hipEventRecord(ev[1], sF));
        if (b->e_in_hook) {                     // run_host_async: its staging pair is free once K1 has read it --
            HIP_TRY(hipEventRecord(b->e_in_hook, sF));      // also when the whole chain runs on this one stream
            b->e_in_hook = nullptr;
        }
        if (pl) HIP_TRY(hipEventRecord(b->e_done[0][k], sF));
        // K2: this call's sign words -> bit packs segbits[k] (read by K2b of call i-nbuf); in order
        // across calls (it carries the receivers' pll / prev / lastbit)
        PllLaunch p;
        fill_pll(b, p, k, len);
        if (pl) HIP_TRY(hipStreamWaitEvent(sA, b->e_done[0][k], 0));
        if (tm) HIP_TRY(hipEventRecord(ev[2], sA));
        if (b->stage_mask & 2) HIP_TRY(launch_pll(p, sA));
        if (tm) HIP_TRY(hipEventRecord(ev[6], sA));
        if (pl) HIP_TRY(hipEventRecord(b->e_done[1][k], sA));
        if (int rc = run_tail(b, k, len, tm, ev, s0, pl ? b->e_done[1][k] : nullptr)) return rc;
    }

    b->timed_last = tm;
    if (tm) b->timed_calls++;
    b->last_stream = s0;
    b->last_len = len;
    b->last_k = k;
    b->calls++;
    return GNUAIS_OK;
}

int gnuais_batch_sync(gnuais_batch *b);
int gnuais_batch_reset(gnuais_batch *b);
int gnuais_batch_discard_frames(gnuais_batch *b, void *stream);

// Try the stage -> stream assignments greedily (PLL stage first, then K3, K2b, the spare; each on every
// free candidate stream), timing a few pipelined calls of the caller's own input each, and keep
// the fastest.  Resets the batch afterwards (the calls advanced every receiver's state).
int gnuais_batch_autotune(gnuais_batch *b, const int16_t *d_samples, int len, void *stream, float *ms_per_call)
{
    if (!b || !d_samples) return fail(GNUAIS_E_ARG, "autotune: NULL argument");
    if (!b->pipeline) {                         // one stream, nothing to assign
        if (ms_per_call) *ms_per_call = 0.0f;
        return GNUAIS_OK;
    }
    if (int rc = gnuais_batch_sync(b)) return rc;
    const bool timing = b->timing;
    b->timing = false;
    // the assignment is measured with K3 on a stream of its own (role 3 gets a queue that was timed: the streamed
    // delivery runs K3 there); outside the delivery loop K3 then shares the deframer's stream (k3_same)
    struct K3Own { gnuais_batch *b; int was; ~K3Own() { b->k3_same = was; } } k3_own{b, b->k3_same};
    b->k3_same = 0;
    auto measure = [&](double &ms, int meas = 10) -> int {
        const int warm = 4;
        for (int i = 0; i < warm + meas; ++i) {
            if (i == warm) {
                if (int rc = gnuais_batch_sync(b)) return rc;
                ms = -now_ms();
            }
            if (int rc = gnuais_batch_run(b, d_samples, len, stream)) return rc;
            if (int rc = gnuais_batch_discard_frames(b, stream)) return rc;
        }
        if (int rc = gnuais_batch_sync(b)) return rc;
        ms = (ms + now_ms()) / meas;
        return GNUAIS_OK;
    };
    const int order[4] = {0, 3, 2, 1};          // the PLL stage first, then K3, K2b, the spare
    double best_all = 0;
    auto search = [&](int chosen[4]) -> int {
        for (int q = 0; q < 4; ++q) chosen[q] = -1;
        for (int r = 0; r < 4; ++r) {
            const int role = order[r];
            double best = 1e30;
            int best_s = -1;
            for (int cand = 0; cand < gnuais_batch::POOL; ++cand) {
                bool used = false;
                for (int q = 0; q < r; ++q) used |= chosen[order[q]] == cand;
                if (used) continue;
                int trial[4];
                for (int q = 0; q < 4; ++q) trial[q] = chosen[q];
                trial[role] = cand;
                for (int q = r + 1; q < 4; ++q) {   // the roles not decided yet: any distinct free streams
                    for (int f = 0; f < gnuais_batch::POOL; ++f) {
                        bool taken = false;
                        for (int t = 0; t < 4; ++t) taken |= trial[t] == f;
                        if (!taken) { trial[order[q]] = f; break; }
                    }
                }
                for (int q = 0; q < 4; ++q) b->s_k[q] = b->pool[trial[q]];
                double ms = 0;
                if (int rc = measure(ms)) return rc;
                if (ms < best) { best = ms; best_s = cand; }
            }
            chosen[role] = best_s;
        }
        return GNUAIS_OK;
    };
    // Ten calls per trial are noisy (+-4 %) and a greedy search can follow the noise into a poor
    // assignment (seen: 0.65 instead of 0.53 ms per call in a third of the runs on one box).  So: two
    // independent searches, and the default assignment, in a longer head-to-head; the fastest stays.
    hipStream_t cand_set[3][4];
    int n_sets = 0;
    for (int q = 0; q < 4; ++q) cand_set[0][q] = b->s_k_default[q];
    n_sets = 1;
    for (int rep = 0; rep < 2; ++rep) {
        int chosen[4];
        if (int rc = search(chosen)) return rc;
        for (int q = 0; q < 4; ++q) cand_set[n_sets][q] = b->pool[chosen[q]];
        ++n_sets;
    }
    // The default assignment (creation order, see gnuais_batch_create) is the best one wherever the process has created no
    // streams of its own (round 6, three boxes: un-calibrated within 1 % of the best calibrated run) and a 40-call leg
    // still carries +-2 % of noise, so a searched assignment only replaces it when it is at least 3 % faster: what a
    // plain gnuais_batch_run() caller gets is then never worse than what this call leaves behind.
    int best_set = 0;
    best_all = 1e30;
    double ms_set[3] = {0, 0, 0};
    for (int k = 0; k < n_sets; ++k) {
        for (int q = 0; q < 4; ++q) b->s_k[q] = cand_set[k][q];
        if (int rc = measure(ms_set[k], 40)) return rc;
    }
    best_all = ms_set[0];
    for (int k = 1; k < n_sets; ++k)
        if (ms_set[k] < 0.97 * ms_set[0] && ms_set[k] < best_all) { best_all = ms_set[k]; best_set = k; }
    for (int q = 0; q < 4; ++q) b->s_k[q] = cand_set[best_set][q];
    b->timing = timing;
    if (ms_per_call) *ms_per_call = (float) best_all;
    return gnuais_batch_reset(b);
}

int gnuais_batch_sync(gnuais_batch *b)
{
    if (!b) return fail(GNUAIS_E_ARG, "sync: NULL batch");
    if (int rc = set_device(b)) return rc;
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    for (auto &st : b->s_k) HIP_TRY(hipStreamSynchronize(st));
    return GNUAIS_OK;
}

static int ensure_stage(gnuais_batch *b, size_t bytes)
{
    if (b->stage_bytes >= bytes) return GNUAIS_OK;
    if (b->stage_x) HIP_TRY(hipFree(b->stage_x));
    b->stage_x = nullptr;
    b->stage_bytes = 0;
    HIP_TRY(hipMalloc((void **) &b->stage_x, bytes));
    b->stage_bytes = bytes;
    return GNUAIS_OK;
}

// The delivery loop (run + stream_nmea) has one more stream to place: the copy kernel's.  Which hardware
// queue a stream shares is not queryable (see gnuais_batch_autotune); a copy stream that shares its queue with
// a stage serialises with it (0.62 against 1.5 ms per C3 call).  Times the loop with the copy kernel on each
// free pool stream and on the stream created for it, keeps the fastest, resets the batch (it stays in
// streaming mode).
int gnuais_batch_autotune_delivery(gnuais_batch *b, const int16_t *d_samples, int len, void *stream, float *ms_per_call)
{
    if (!b || !d_samples) return fail(GNUAIS_E_ARG, "autotune_delivery: NULL argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    const char *text = nullptr;
    size_t tl = 0;
    if (int rc = gnuais_batch_stream_nmea(b, &text, &tl, nullptr, nullptr)) return rc;      // rings, buffers, s_copy
    if (!b->pipeline) {
        if (ms_per_call) *ms_per_call = 0.0f;
        return gnuais_batch_reset(b);
    }
    const bool timing = b->timing;
    b->timing = false;
    auto quiesce = [&]() -> int {
        if (int rc = gnuais_batch_sync(b)) return rc;
        HIP_TRY(hipDeviceSynchronize());
        return GNUAIS_OK;
    };
    auto measure = [&](double &ms, int meas) -> int {
        const int warm = gnuais_batch::NRING + 2;
        for (int i = 0; i < warm + meas; ++i) {
            if (i == warm) {
                if (int rc = quiesce()) return rc;
                ms = -now_ms();
            }
            if (int rc = gnuais_batch_run(b, d_samples, len, stream)) return rc;
            const int rc = gnuais_batch_stream_nmea(b, &text, &tl, nullptr, nullptr);
            if (rc != GNUAIS_OK && rc != GNUAIS_E_OVERFLOW) return rc;
        }
        if (int rc = quiesce()) return rc;
        ms = (ms + now_ms()) / meas;
        return GNUAIS_OK;
    };
    hipStream_t own = b->s_copy_own ? b->s_copy_own : b->s_copy;
    b->s_copy_own = own;
    // twelve calls per candidate, then the two fastest again over thirty (a dozen calls are noisy)
    hipStream_t top[2] = {own, own};
    double top_ms[2] = {1e30, 1e30};
    for (int cand = -1; cand < gnuais_batch::POOL; ++cand) {
        hipStream_t st = cand < 0 ? own : b->pool[cand];
        bool used = false;
        for (int q = 0; q < 4; ++q) used |= b->s_k[q] == st;
        if (used || !st) continue;
        b->s_copy = st;
        double ms = 0;
        if (int rc = measure(ms, 12)) return rc;
        if (ms < top_ms[0]) { top_ms[1] = top_ms[0]; top[1] = top[0]; top_ms[0] = ms; top[0] = st; }
        else if (ms < top_ms[1]) { top_ms[1] = ms; top[1] = st; }
    }
    hipStream_t best_s = top[0];
    double best = 1e30;
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && top[1] == top[0]) break;
        b->s_copy = top[k];
        double ms = 0;
        if (int rc = measure(ms, 30)) return rc;
        if (ms < best) { best = ms; best_s = top[k]; }
    }
    b->s_copy = best_s;
    b->timing = timing;
    if (ms_per_call) *ms_per_call = (float) best;
    return gnuais_batch_reset(b);
}

int gnuais_batch_run_host(gnuais_batch *b, const int16_t *h_samples, int len)
{
    if (!b || !h_samples) return fail(GNUAIS_E_ARG, "run_host: NULL argument");
    if (len <= 0 || len > b->max_len) return fail(GNUAIS_E_ARG, "run_host: len out of range");
    if (int rc = set_device(b)) return rc;
    const size_t bytes = sizeof(int16_t) * (size_t) len * (size_t) b->N;
    if (int rc = ensure_stage(b, bytes)) return rc;
    HIP_TRY(hipMemcpy(b->stage_x, h_samples, bytes, hipMemcpyHostToDevice));
    if (int rc = gnuais_batch_run(b, b->stage_x, len, nullptr)) return rc;
    return gnuais_batch_sync(b);
}

int gnuais_batch_run_host_async(gnuais_batch *b, const int16_t *h_samples, int len)
{
    if (!b || !h_samples) return fail(GNUAIS_E_ARG, "run_host_async: NULL argument");
    if (len <= 0 || len > b->max_len) return fail(GNUAIS_E_ARG, "run_host_async: len out of range");
    if (int rc = set_device(b)) return rc;
    const size_t bytes = sizeof(int16_t) * (size_t) len * (size_t) b->N;
    if (!b->s_io) {
        HIP_TRY(hipStreamCreateWithFlags(&b->s_io, hipStreamNonBlocking));
        for (auto &e : b->e_in) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (b->pin_bytes < bytes) {                 // (re)size for the largest call seen
        HIP_TRY(hipStreamSynchronize(b->s_io));
        const size_t cap = sizeof(int16_t) * (size_t) b->max_len * (size_t) b->N;
        const size_t want = std::min(cap, std::max(bytes, (size_t) 1 << 20));
        for (int q = 0; q < 2; ++q) {
            if (b->pin[q]) HIP_TRY(hipHostFree(b->pin[q]));
            if (b->dev_in[q]) HIP_TRY(hipFree(b->dev_in[q]));
            b->pin[q] = b->dev_in[q] = nullptr;
        }
        b->pin_bytes = 0;
        for (int q = 0; q < 2; ++q) {
            HIP_TRY(hipHostMalloc((void **) &b->pin[q], want, hipHostMallocDefault));
            HIP_TRY(hipMalloc((void **) &b->dev_in[q], want));
        }
        b->pin_bytes = want;
        b->host_calls = 0;
    }
    const int q = (int) (b->host_calls & 1);
    // the pair was last used two calls ago: its transfer and the FIR that read it must be done
    if (b->host_calls >= 2) HIP_TRY(hipEventSynchronize(b->e_in[q]));
    memcpy(b->pin[q], h_samples, bytes);
    HIP_TRY(hipMemcpyAsync(b->dev_in[q], b->pin[q], bytes, hipMemcpyHostToDevice, b->s_io));
    b->e_in_hook = b->e_in[q];                  // recorded right behind K1 (run_chain): K1 is the input's only reader
    const int rc = gnuais_batch_run(b, b->dev_in[q], len, b->s_io);
    if (b->e_in_hook) {                         // the run failed before its K1: whatever s_io holds (the copy) guards the pair
        b->e_in_hook = nullptr;
        (void) hipEventRecord(b->e_in[q], b->s_io);
    }
    b->host_calls++;                            // also after a failed run: the copy from pin[q] may still be in flight
    return rc;
}

int gnuais_batch_filter(gnuais_batch *b, const int16_t *d_samples, int len, float *d_out,
                        void *stream)
{
    if (!b || !d_samples || !d_out) return fail(GNUAIS_E_ARG, "filter: NULL argument");
    if (len <= 0 || len > b->max_len) return fail(GNUAIS_E_ARG, "filter: len out of range");
    if (int rc = set_device(b)) return rc;
    hipStream_t s = (hipStream_t) stream;
    if (int rc = gnuais_batch_sync(b)) return rc;       // the sign-word scratch is shared
    if (int rc = run_fir(b, d_samples, len, d_out, s, (int) (b->calls % (unsigned) b->nbuf))) return rc;
    b->last_stream = s;
    b->timed_last = false;
    return GNUAIS_OK;
}

// filter_run_buf() from and to HOST memory, for plain-C callers (gnuais_amd/csrc/protodec_hip.c: the reference's
// filter.h names): h_samples int16 [len][n_channels], h_out float [len][n_channels]; synchronous
int gnuais_batch_filter_host(gnuais_batch *b, const int16_t *h_samples, int len, float *h_out)
{
    if (!b || !h_samples || !h_out) return fail(GNUAIS_E_ARG, "filter_host: NULL argument");
    if (len <= 0 || len > b->max_len) return fail(GNUAIS_E_ARG, "filter_host: len out of range");
    if (int rc = set_device(b)) return rc;
    const size_t n = (size_t) len * (size_t) b->N;
    if (b->stage_bytes < n * sizeof(int16_t)) {
        if (b->stage_x) HIP_TRY(hipFree(b->stage_x));
        b->stage_x = nullptr;
        b->stage_bytes = 0;
        HIP_TRY(hipMalloc((void **) &b->stage_x, n * sizeof(int16_t)));
        b->stage_bytes = n * sizeof(int16_t);
    }
    if (b->stage_f_bytes < n * sizeof(float)) {
        if (b->stage_f) HIP_TRY(hipFree(b->stage_f));
        b->stage_f = nullptr;
        b->stage_f_bytes = 0;
        HIP_TRY(hipMalloc((void **) &b->stage_f, n * sizeof(float)));
        b->stage_f_bytes = n * sizeof(float);
    }
    HIP_TRY(hipMemcpy(b->stage_x, h_samples, n * sizeof(int16_t), hipMemcpyHostToDevice));
    if (int rc = gnuais_batch_filter(b, b->stage_x, len, b->stage_f, nullptr)) return rc;
    if (int rc = gnuais_batch_sync(b)) return rc;
    HIP_TRY(hipMemcpy(h_out, b->stage_f, n * sizeof(float), hipMemcpyDeviceToHost));
    return GNUAIS_OK;
}

int gnuais_batch_decode_bits(gnuais_batch *b, const uint8_t *h_bits, int stride,
                             const int32_t *h_count)
{
    if (!b || !h_bits || !h_count || stride <= 0) return fail(GNUAIS_E_ARG, "decode_bits: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    const int N = b->N, segcap = b->seg_words * 32, chunk = segcap * b->n_seg;
    int maxc = 0;
    for (int c = 0; c < N; ++c) {
        if (h_count[c] < 0 || h_count[c] > stride) return fail(GNUAIS_E_ARG, "decode_bits: count");
        maxc = std::max(maxc, h_count[c]);
    }
    const size_t rowlen = (size_t) b->n_seg * PACK_STRIDE;
    std::vector<uint32_t> words(rowlen * N), cnt((size_t) b->n_seg * N);
    for (int pos = 0; pos < maxc; pos += chunk) {
        std::fill(words.begin(), words.end(), 0u);
        std::fill(cnt.begin(), cnt.end(), 0u);
        for (int c = 0; c < N; ++c) {
            const int n = std::max(0, std::min(chunk, h_count[c] - pos));
            const uint8_t *src = h_bits + (size_t) c * stride + pos;
            for (int k = 0; k < n; ++k) {
                const int seg = k / segcap, kk = k % segcap;
                if (src[k] & 1)
                    words[(size_t) c * rowlen + (size_t) seg * PACK_STRIDE + (kk >> 5)] |= 1u << (kk & 31);
            }
            for (int seg = 0; seg * segcap < n; ++seg)
                cnt[(size_t) c * b->n_seg + seg] = (uint32_t) std::min(segcap, n - seg * segcap);
        }
        HIP_TRY(hipMemcpy(b->segbits[0], words.data(), words.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->segcnt[0], cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice));
        HdlcLaunch h;
        fill_hdlc(b, h, 0);
        HIP_TRY(b->hdlc_variant ? launch_hdlc_events(h, nullptr) : launch_hdlc_deframe(h, nullptr));
        HIP_TRY(launch_hdlc_crc(h, nullptr));
        if (b->streaming) b->ring_runs[b->ring_cur]++;
        b->hdlc_calls++;
        HIP_TRY(hipDeviceSynchronize());
    }
    b->last_stream = nullptr;
    return GNUAIS_OK;
}

int gnuais_batch_last_bits(gnuais_batch *b, uint8_t *h_bits, int stride, int32_t *h_count)
{
    if (!b || !h_bits || !h_count || stride <= 0) return fail(GNUAIS_E_ARG, "last_bits: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    const int N = b->N;
    const size_t rowlen = (size_t) b->n_seg * PACK_STRIDE;
    std::vector<uint32_t> words(rowlen * N), cnt((size_t) b->n_seg * N);
    HIP_TRY(hipMemcpy(words.data(), b->segbits[b->last_k], words.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cnt.data(), b->segcnt[b->last_k], cnt.size() * 4, hipMemcpyDeviceToHost));
    const int nseg_used = ((b->last_len + 31) / 32 + SEG_WORDS - 1) / SEG_WORDS;
    for (int c = 0; c < N; ++c) {
        int n = 0;
        uint8_t *dst = h_bits + (size_t) c * stride;
        for (int seg = 0; seg < nseg_used; ++seg) {
            const int m = (int) std::min<uint32_t>(cnt[(size_t) c * b->n_seg + seg],
                                                   (uint32_t) b->seg_words * 32);
            if (n + m > stride) return fail(GNUAIS_E_ARG, "last_bits: stride too small");
            const uint32_t *w = &words[(size_t) c * rowlen + (size_t) seg * PACK_STRIDE];
            for (int k = 0; k < m; ++k) dst[n + k] = (w[k >> 5] >> (k & 31)) & 1u;
            n += m;
        }
        h_count[c] = n;
    }
    return GNUAIS_OK;
}

// device buffers of the post-stage (sorted records / text, rocPRIM scratch): allocated on first use
static int ensure_post_buffers(gnuais_batch *b, uint32_t have)
{
    const size_t need_text = (size_t) have * 164, need_scratch = nmea_scratch_bytes((int) have);
    if (b->d_text_bytes < need_text) {
        if (b->d_text) (void) hipFree(b->d_text);
        b->d_text = nullptr;
        b->d_text_bytes = 0;
        HIP_TRY(hipMalloc((void **) &b->d_text, need_text));
        b->d_text_bytes = need_text;
    }
    if (b->nmea_scratch_bytes < need_scratch) {
        if (b->nmea_scratch) (void) hipFree(b->nmea_scratch);
        b->nmea_scratch = nullptr;
        b->nmea_scratch_bytes = 0;
        HIP_TRY(hipMalloc(&b->nmea_scratch, need_scratch));
        b->nmea_scratch_bytes = need_scratch;
    }
    return GNUAIS_OK;
}

// drain: records and / or sentences of everything queued, consumed once
static int drain_impl(gnuais_batch *b, gnuais_frame *h_frames, int max_frames, int *n_frames,
                      uint8_t *seqnr, char *out, size_t out_cap, size_t *out_len, int *n_sentences)
{
    // a streaming batch spreads its frames over NRING rings that gnuais_batch_stream_nmea() consumes: ring 0 alone
    // would be a partial view, and clearing its counters would lose frames and error flags
    if (b->streaming) return fail(GNUAIS_E_STATE, "drain: the batch is streaming (gnuais_batch_stream_nmea); "
                                                  "set_option(\"streaming\", 0) leaves that mode");
    if (int rc = gnuais_batch_sync(b)) return rc;
    uint32_t cnt[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(cnt, b->frame_count, sizeof cnt, hipMemcpyDeviceToHost));
    const uint32_t have = std::min<uint32_t>(cnt[0], (uint32_t) b->frame_cap);
    const bool overflow = cnt[1] || cnt[0] > (uint32_t) b->frame_cap;
    const bool watchdog = cnt[3] != 0;          // a PLL-stage wave timed out waiting for its partner
    if (h_frames && (uint32_t) max_frames < have) return fail(GNUAIS_E_ARG, "drain: frame buffer too small");
    if (have) {
        const size_t N = (size_t) b->N;
        if (int rc = ensure_post_buffers(b, have)) return rc;
        if (seqnr) {
            // sentences first (the formatter sorts for itself and leaves the ring untouched)
            if (!b->d_seq[0]) {
                HIP_TRY(hipMalloc((void **) &b->d_seq[0], N));
                HIP_TRY(hipMalloc((void **) &b->d_seq[1], N));
            }
            HIP_TRY(hipMemcpy(b->d_seq[0], seqnr, N, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(b->d_seq[1], b->d_seq[0], N, hipMemcpyDeviceToDevice));
            uint32_t info[3] = {0, 0, 0};
            HIP_TRY(nmea_format(b->frames, (int) have, b->N, b->d_seq[0], b->d_seq[1], b->d_text, b->d_text_bytes,
                                b->nmea_scratch, b->nmea_scratch_bytes, info, nullptr));
            if (info[2]) return fail(GNUAIS_E_HIP, "drain: a frame record names a channel outside the batch");
            if ((size_t) info[0] > out_cap) {
                *out_len = info[0];
                return fail(GNUAIS_E_ARG, "drain: text buffer too small");
            }
            if (info[0]) HIP_TRY(hipMemcpy(out, b->d_text, info[0], hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(seqnr, b->d_seq[1], N, hipMemcpyDeviceToHost));
            *out_len = info[0];
            if (n_sentences) *n_sentences = (int) info[1];
        }
        if (h_frames) {
            // K3 appends the frames in pieces, in whatever order its blocks finish; the reference's
            // print order (channel, then time) is restored on the device -- radix sort of
            // (channel, end_bit), gather -- and the records cross PCIe once, straight into h_frames
            gnuais_frame *sorted = reinterpret_cast<gnuais_frame *>(b->d_text);
            HIP_TRY(frames_sort(b->frames, (int) have, sorted, b->nmea_scratch, b->nmea_scratch_bytes, nullptr));
            HIP_TRY(hipMemcpy(h_frames, sorted, sizeof(gnuais_frame) * have, hipMemcpyDeviceToHost));
        }
        if (n_frames) *n_frames = (int) have;
    }
    HIP_TRY(hipMemset(b->frame_count, 0, sizeof cnt));
    b->hdlc_calls = 0;
    if (watchdog)
        return fail(GNUAIS_E_HIP, "drain: the PLL stage's watchdog fired (device hung or badly oversubscribed); results are incomplete");
    if (overflow) return fail(GNUAIS_E_OVERFLOW, "drain: frame ring overflowed, frames were dropped");
    return GNUAIS_OK;
}

// Row f1 complete on the device: sentences AND stdout lines of everything queued, consumed once
int gnuais_batch_drain_messages(gnuais_batch *b, uint8_t *seqnr, const char *chanid, char *nmea, size_t nmea_cap,
                                size_t *nmea_len, int *n_sentences, char *text, size_t text_cap, size_t *text_len,
                                int *n_lines, int *n_frames)
{
    if (!b || !seqnr || !nmea_len || !text_len || (nmea_cap && !nmea) || (text_cap && !text))
        return fail(GNUAIS_E_ARG, "drain_messages: argument");
    if (b->streaming) return fail(GNUAIS_E_ARG, "drain_messages: the batch is streaming (gnuais_batch_stream_nmea)");
    *nmea_len = *text_len = 0;
    if (n_sentences) *n_sentences = 0;
    if (n_lines) *n_lines = 0;
    if (n_frames) *n_frames = 0;
    if (int rc = gnuais_batch_sync(b)) return rc;
    uint32_t cnt[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(cnt, b->frame_count, sizeof cnt, hipMemcpyDeviceToHost));
    const uint32_t have = std::min<uint32_t>(cnt[0], (uint32_t) b->frame_cap);
    const bool overflow = cnt[1] || cnt[0] > (uint32_t) b->frame_cap, watchdog = cnt[3] != 0;
    if (have) {
        const size_t N = (size_t) b->N, line = messages_line_bytes();
        if (nmea_cap < (size_t) have * 164 || text_cap < (size_t) have * line)
            return fail(GNUAIS_E_ARG, "drain_messages: buffers too small (164 / 512 bytes per pending frame always suffice)");
        if (int rc = ensure_post_buffers(b, have)) return rc;
        // lines at a fixed stride, their lengths and offsets, the packed text, two info words, the channel names
        const size_t need = (size_t) have * line * 2 + (size_t) have * 8 + 256 + N + 256;
        if (b->d_msg_bytes < need) {
            if (b->d_msg) (void) hipFree(b->d_msg);
            b->d_msg = nullptr;
            b->d_msg_bytes = 0;
            HIP_TRY(hipMalloc((void **) &b->d_msg, need + need / 4));
            b->d_msg_bytes = need + need / 4;
        }
        char *lines = b->d_msg, *packed = lines + (size_t) have * line;
        uint32_t *len = reinterpret_cast<uint32_t *>(packed + (size_t) have * line), *off = len + have;
        uint32_t *info2 = off + have;
        char *d_chanid = reinterpret_cast<char *>(info2 + 64);
        if (chanid) HIP_TRY(hipMemcpy(d_chanid, chanid, N, hipMemcpyHostToDevice));
        if (!b->d_seq[0]) {
            HIP_TRY(hipMalloc((void **) &b->d_seq[0], N));
            HIP_TRY(hipMalloc((void **) &b->d_seq[1], N));
        }
        HIP_TRY(hipMemcpy(b->d_seq[0], seqnr, N, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_seq[1], b->d_seq[0], N, hipMemcpyDeviceToDevice));
        uint32_t raw[4] = {0, 0, 0, 0}, inf[2] = {0, 0};
        HIP_TRY(nmea_format_enqueue(b->frames, (int) have, (int) have, b->N, b->d_seq[0], b->d_seq[1], b->d_text,
                                    b->d_text_bytes, b->nmea_scratch, b->nmea_scratch_bytes, raw, nullptr, 0, 0,
                                    nullptr, nullptr));
        HIP_TRY(messages_format_enqueue(b->frames, (int) have, b->N, b->d_seq[0], chanid ? d_chanid : nullptr,
                                        b->nmea_scratch, b->nmea_scratch_bytes, lines, len, off, packed,
                                        (size_t) have * line, info2, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(inf, info2, 8, hipMemcpyDeviceToHost));
        if (raw[3]) return fail(GNUAIS_E_HIP, "drain_messages: a frame record names a channel outside the batch");
        const size_t nl = (size_t) raw[0] + raw[1];
        if (nl) HIP_TRY(hipMemcpy(nmea, b->d_text, nl, hipMemcpyDeviceToHost));
        if (inf[0]) HIP_TRY(hipMemcpy(text, packed, inf[0], hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(seqnr, b->d_seq[1], N, hipMemcpyDeviceToHost));
        *nmea_len = nl;
        *text_len = inf[0];
        if (n_sentences) *n_sentences = (int) raw[2];
        if (n_lines) *n_lines = (int) inf[1];
        if (n_frames) *n_frames = (int) have;
    }
    HIP_TRY(hipMemset(b->frame_count, 0, sizeof cnt));
    b->hdlc_calls = 0;
    if (watchdog)
        return fail(GNUAIS_E_HIP, "drain_messages: the PLL stage's watchdog fired (device hung or badly oversubscribed)");
    if (overflow) return fail(GNUAIS_E_OVERFLOW, "drain_messages: frame ring overflowed, frames were dropped");
    return GNUAIS_OK;
}

// Row f3 on the device: what the queued frames do to the reference's position cache, folded per vessel.
// Does not consume the frames (call it before a drain).
int gnuais_batch_fold_vessels(gnuais_batch *b, gnuais_vessel *vessels, int cap, int *n_vessels)
{
    if (!b || !n_vessels || cap < 0 || (cap > 0 && !vessels)) return fail(GNUAIS_E_ARG, "fold_vessels: argument");
    if (b->streaming) return fail(GNUAIS_E_ARG, "fold_vessels: the batch is streaming (gnuais_batch_stream_nmea)");
    *n_vessels = 0;
    if (int rc = gnuais_batch_sync(b)) return rc;
    uint32_t cnt[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(cnt, b->frame_count, sizeof cnt, hipMemcpyDeviceToHost));
    const uint32_t have = std::min<uint32_t>(cnt[0], (uint32_t) b->frame_cap);
    if (!have) return GNUAIS_OK;
    if (int rc = ensure_post_buffers(b, have)) return rc;
    // at most one vessel per frame; the table shares the text buffer (164 bytes per frame >= 120)
    gnuais_vessel *d_tab = reinterpret_cast<gnuais_vessel *>(b->d_text);
    const int d_cap = (int) std::min<size_t>(b->d_text_bytes / sizeof(gnuais_vessel), (size_t) have);
    if (!b->d_word) HIP_TRY(hipMalloc((void **) &b->d_word, 16));
    HIP_TRY(vessels_fold_enqueue(b->frames, (int) have, b->nmea_scratch, b->nmea_scratch_bytes, d_tab, d_cap,
                                 b->d_word, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    uint32_t nv = 0;
    HIP_TRY(hipMemcpy(&nv, b->d_word, 4, hipMemcpyDeviceToHost));
    *n_vessels = (int) nv;
    if ((int) nv > cap) return fail(GNUAIS_E_OVERFLOW, "fold_vessels: table too small (*n_vessels entries needed)");
    if (nv) HIP_TRY(hipMemcpy(vessels, d_tab, sizeof(gnuais_vessel) * nv, hipMemcpyDeviceToHost));
    return GNUAIS_OK;
}

// ---- row f3, carried: the position cache kept on the device from batch to batch ------------------------------------
int gnuais_batch_vessel_table_enable(gnuais_batch *b, int capacity)
{
    if (!b || capacity < 1 || capacity > (1 << 24)) return fail(GNUAIS_E_ARG, "vessel_table_enable: capacity 1 .. 2^24");
    if (int rc = gnuais_batch_sync(b)) return rc;
    if (b->s_post) HIP_TRY(hipStreamSynchronize(b->s_post));
    uint32_t slots = 1024;
    while (slots < 2u * (uint32_t) capacity) slots <<= 1;        // at most half full: short probe sequences
    if (b->vt) HIP_TRY(hipFree(b->vt));
    b->vt = nullptr;
    b->vt_slots = 0;
    HIP_TRY(hipMalloc(&b->vt, vessel_table_bytes(slots)));
    HIP_TRY(hipMemset(b->vt, 0, vessel_table_bytes(slots)));
    if (!b->vt_fslot) HIP_TRY(hipMalloc((void **) &b->vt_fslot, sizeof(uint32_t) * (size_t) b->frame_cap));
    b->vt_slots = slots;
    b->vt_capacity = capacity;
    return GNUAIS_OK;
}

int gnuais_batch_vessel_table_clear(gnuais_batch *b)
{
    if (!b || !b->vt) return fail(GNUAIS_E_STATE, "vessel_table_clear: no table (gnuais_batch_vessel_table_enable)");
    if (int rc = gnuais_batch_sync(b)) return rc;
    if (b->s_post) HIP_TRY(hipStreamSynchronize(b->s_post));
    HIP_TRY(hipMemset(b->vt, 0, vessel_table_bytes(b->vt_slots)));
    return GNUAIS_OK;
}

// drain-type use: the queued frames into the table (they stay queued)
int gnuais_batch_vessel_table_update(gnuais_batch *b)
{
    if (!b || !b->vt) return fail(GNUAIS_E_STATE, "vessel_table_update: no table (gnuais_batch_vessel_table_enable)");
    if (b->streaming) return fail(GNUAIS_E_STATE, "vessel_table_update: a streaming batch updates its table by itself");
    if (int rc = gnuais_batch_sync(b)) return rc;             // a drain-type call: waits for the chain like the drains do
    hipStream_t s = b->pipeline ? k3_stream(b) : b->last_stream;
    HIP_TRY(vessel_table_update_enqueue(b->frames, b->frame_count, b->frame_cap, b->vt, b->vt_slots, b->vt_fslot, s));
    return GNUAIS_OK;
}

int gnuais_batch_vessel_table(gnuais_batch *b, gnuais_vessel *vessels, int cap, int *n_vessels)
{
    if (!b || !n_vessels || cap < 0 || (cap > 0 && !vessels)) return fail(GNUAIS_E_ARG, "vessel_table: argument");
    *n_vessels = 0;
    if (!b->vt) return fail(GNUAIS_E_STATE, "vessel_table: no table (gnuais_batch_vessel_table_enable)");
    if (int rc = set_device(b)) return rc;
    hipStream_t s = b->streaming ? b->s_post : (b->pipeline ? k3_stream(b) : b->last_stream);
    uint32_t info[4] = {0, 0, 0, 0};
    int n = 0;
    HIP_TRY(vessel_table_fetch(b->vt, b->vt_slots, vessels, cap, &n, info, s));
    *n_vessels = n;
    if (info[1] || (int) info[0] > b->vt_capacity) {
        char msg[160];
        snprintf(msg, sizeof msg, "vessel_table: more vessels than the table was enabled for (%u seen%s, capacity %d)",
                 info[0], info[1] ? ", some dropped" : "", b->vt_capacity);
        return fail(GNUAIS_E_OVERFLOW, msg);
    }
    if (n > cap) return fail(GNUAIS_E_OVERFLOW, "vessel_table: output too small (*n_vessels entries needed)");
    std::sort(vessels, vessels + n, [](const gnuais_vessel &x, const gnuais_vessel &y) { return x.mmsi < y.mmsi; });
    return GNUAIS_OK;
}

int gnuais_batch_drain_frames(gnuais_batch *b, gnuais_frame *h_out, int max, int *n_out)
{
    if (!b || !n_out || (max > 0 && !h_out)) return fail(GNUAIS_E_ARG, "drain_frames: argument");
    *n_out = 0;
    static gnuais_frame none;
    return drain_impl(b, h_out ? h_out : &none, max, n_out, nullptr, nullptr, 0, nullptr, nullptr);
}

int gnuais_batch_drain_nmea(gnuais_batch *b, uint8_t *seqnr, char *out, size_t out_cap, size_t *out_len,
                            int *n_sentences, int *n_frames)
{
    if (!b || !seqnr || !out_len || (out_cap > 0 && !out)) return fail(GNUAIS_E_ARG, "drain_nmea: argument");
    *out_len = 0;
    if (n_sentences) *n_sentences = 0;
    if (n_frames) *n_frames = 0;
    return drain_impl(b, nullptr, 0, n_frames, seqnr, out, out_cap, out_len, n_sentences);
}

int gnuais_batch_drain_frames_nmea(gnuais_batch *b, gnuais_frame *h_frames, int max, int *n_frames,
                                   uint8_t *seqnr, char *out, size_t out_cap, size_t *out_len, int *n_sentences)
{
    if (!b || !h_frames || !n_frames || !seqnr || !out_len || (out_cap > 0 && !out))
        return fail(GNUAIS_E_ARG, "drain_frames_nmea: argument");
    *n_frames = 0;
    *out_len = 0;
    if (n_sentences) *n_sentences = 0;
    return drain_impl(b, h_frames, max, n_frames, seqnr, out, out_cap, out_len, n_sentences);
}

// Streaming delivery (row f1 end to end).  Call after every gnuais_batch_run(): the frames of the
// runs since the previous call are taken off (K3 moves on to the next ring at once) and go through
// three later calls -- two calls on: count read + device formatter queued; then: text copy into pinned
// memory queued; then: text handed out -- so that no call waits for work queued in the same call.
int gnuais_batch_stream_nmea(gnuais_batch *b, const char **text, size_t *len, int *n_sentences, int *n_frames)
{
    if (!b || !text || !len) return fail(GNUAIS_E_ARG, "stream_nmea: argument");
    if (int rc = set_device(b)) return rc;
    *text = nullptr;
    *len = 0;
    if (n_sentences) *n_sentences = 0;
    if (n_frames) *n_frames = -1;               // nothing handed out yet
    constexpr int NR = gnuais_batch::NRING;
    const size_t N = (size_t) b->N;
    const size_t text_cap = (size_t) b->frame_cap * 164;        // a full ring of two-sentence frames
    if (!b->streaming) {                        // first use (or back from set_option("streaming", 0))
        if (int rc = gnuais_batch_sync(b)) return rc;
        // every object is created only if it does not exist yet: a first use that failed half way (e.g. the pinned
        // allocation) is repeated by the next call without leaking what the failed one had made
        b->ring[0] = b->frames;
        b->ring_count[0] = b->frame_count;
        for (int q = 1; q < NR; ++q) {
            if (!b->ring[q]) HIP_TRY(hipMalloc((void **) &b->ring[q], sizeof(gnuais_frame) * (size_t) b->frame_cap));
            if (!b->ring_count[q]) {
                HIP_TRY(hipMalloc((void **) &b->ring_count[q], sizeof(uint32_t) * 4));
                HIP_TRY(hipMemset(b->ring_count[q], 0, sizeof(uint32_t) * 4));
            }
        }
        b->n_chunks = k3_blocks(b->N) * k3_passes(b->cand_K);
        b->sh_text_want = std::max(b->sh_text_want, ((size_t) b->frame_cap * 32 + 65536) & ~(size_t) 15);
        for (int q = 0; q < NR; ++q) {
            if (!b->ring_chunks[q]) HIP_TRY(hipMalloc((void **) &b->ring_chunks[q], sizeof(uint2) * (size_t) b->n_chunks));
            b->ring_runs[q] = 2;                // whatever ring 0 holds by now came without a table
            if (!b->sd_text[q]) {
                HIP_TRY(hipMalloc((void **) &b->sd_text[q], text_cap));
                b->sd_text_bytes[q] = text_cap;
            }
            if (!b->e_fill[q]) HIP_TRY(hipEventCreateWithFlags(&b->e_fill[q], hipEventDisableTiming));
            if (!b->e_fmt[q]) HIP_TRY(hipEventCreateWithFlags(&b->e_fmt[q], hipEventDisableTiming));
            if (!b->e_txt[q]) HIP_TRY(hipEventCreateWithFlags(&b->e_txt[q], hipEventDisableTiming));
            // pinned text: a fifth of the worst case to begin with (single-sentence frames of average
            // length fill it to about a third); a slot whose text does not fit grows, see (3)
            if (!b->sh_text[q]) {
                HIP_TRY(hipHostMalloc((void **) &b->sh_text[q], b->sh_text_want, hipHostMallocDefault));
                b->sh_text_bytes[q] = b->sh_text_want;
            }
        }
        if (!b->sd_info) {
            HIP_TRY(hipMalloc((void **) &b->sd_info, sizeof(uint32_t) * 8 * NR));
            HIP_TRY(hipMemset(b->sd_info, 0, sizeof(uint32_t) * 8 * NR));
        }
        if (const char *v = getenv("GNUAIS_COPY_WGS")) b->copy_wgs = std::max(1, atoi(v));
        if (const char *v = getenv("GNUAIS_COPY_ON_K3")) b->copy_on_k3 = atoi(v) != 0;
        const size_t need_scratch = nmea_scratch_bytes(b->frame_cap, b->n_chunks);
        if (b->nmea_scratch_bytes < need_scratch) {
            if (b->nmea_scratch) HIP_TRY(hipFree(b->nmea_scratch));
            b->nmea_scratch = nullptr;
            b->nmea_scratch_bytes = 0;
            HIP_TRY(hipMalloc(&b->nmea_scratch, need_scratch));
            b->nmea_scratch_bytes = need_scratch;
        }
        if (!b->s_copy) {
            // The formatter's kernels go onto K3's stream: they are small, K3's stream is idle most of a call,
            // and every further stream is one more tenant for the few hardware queues (a formatter stream
            // that shares its queue with a stage serialises with it: 0.8 or 1.5 ms per call, by luck).
            // Only the copy, which lasts as long as PCIe needs, has a stream of its own.
            int lo = 0, hi = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIP_TRY(hipStreamCreateWithPriority(&b->s_copy, hipStreamNonBlocking, hi));
        }
        if (!b->sh_info) {
            HIP_TRY(hipHostMalloc((void **) &b->sh_info, sizeof(uint32_t) * 8 * NR, hipHostMallocDefault));
            memset(b->sh_info, 0, sizeof(uint32_t) * 8 * NR);
        }
        for (auto &p : b->sd_seq)
            if (!p) {
                HIP_TRY(hipMalloc((void **) &p, N));
                HIP_TRY(hipMemset(p, 0, N));
            }
        HIP_TRY(hipDeviceSynchronize());
        b->streaming = true;
    }
    hipStream_t sD = b->pipeline ? k3_stream(b) : b->last_stream;
    const int c = b->ring_cur;
    // (1) ring c: everything K3 has been asked to append so far
    HIP_TRY(hipEventRecord(b->e_fill[c], sD));
    const int runs = b->ring_runs[c];
    // (2) K3 moves on to the next ring; that ring's formatter (queued NRING - 1 calls ago) must be done
    const int nx = (c + 1) % NR;
    if (b->s_stage[nx]) HIP_TRY(hipStreamWaitEvent(sD, b->e_fmt[nx], 0));
    b->ring_cur = nx;
    b->ring_runs[nx] = 0;
    // (3) hand out the text of slot nx, queued NRING - 1 calls ago: the only wait of this call.  (Before (4): a slow-path copy below must not queue behind it.)
    int rc_late = GNUAIS_OK;
    if (b->s_stage[nx]) {
        HIP_TRY(hipEventSynchronize(b->e_txt[nx]));
        const uint32_t *info = b->sh_info + 8 * nx;
        const uint32_t have = std::min<uint32_t>(info[4], (uint32_t) b->frame_cap);
        const size_t n_text = (size_t) info[0] + info[1];
        if (n_text > b->sh_text_bytes[nx]) {
            // the pinned buffer was too small for this slot (the first calls, or traffic grew): make it larger
            // and fetch the text from the device copy, which stays intact until the slot is formatted again
            const size_t want = std::max(b->sh_text_want, (n_text + n_text / 2 + 65536) & ~(size_t) 15);
            b->sh_text_want = want;
            if (b->sh_text[nx]) HIP_TRY(hipHostFree(b->sh_text[nx]));
            b->sh_text[nx] = nullptr;
            b->sh_text_bytes[nx] = 0;
            HIP_TRY(hipHostMalloc((void **) &b->sh_text[nx], want, hipHostMallocDefault));
            b->sh_text_bytes[nx] = want;
            HIP_TRY(hipMemcpyAsync(b->sh_text[nx], b->sd_text[nx], n_text, hipMemcpyDeviceToHost, b->s_copy));
            HIP_TRY(hipStreamSynchronize(b->s_copy));
        }
        *text = b->sh_text[nx];
        *len = n_text;
        if (n_sentences) *n_sentences = (int) info[2];
        if (n_frames) *n_frames = (int) have;
        b->s_stage[nx] = 0;
        if (info[3]) rc_late = fail(GNUAIS_E_HIP, "stream_nmea: a frame record names a channel outside the batch");
        else if (info[7])
            rc_late = fail(GNUAIS_E_HIP, "stream_nmea: the PLL stage's watchdog fired (device hung or badly oversubscribed)");
        else if (info[5] || info[4] > (uint32_t) b->frame_cap)
            rc_late = fail(GNUAIS_E_OVERFLOW, "stream_nmea: frame ring overflowed, frames were dropped");
    }
    // (4) ring c: order, sequence digits, text, copy into pinned memory -- queued now, behind its K3, with
    // every size taken on the device
    b->s_post = sD;                             // behind the K3 launches that filled the ring, in stream order
    uint32_t *totals = nullptr;
    if (b->sh_text_bytes[c] < b->sh_text_want) {                // catch up with a buffer that had to grow (slot c is idle)
        if (b->sh_text[c]) HIP_TRY(hipHostFree(b->sh_text[c]));
        b->sh_text[c] = nullptr;
        b->sh_text_bytes[c] = 0;
        HIP_TRY(hipHostMalloc((void **) &b->sh_text[c], b->sh_text_want, hipHostMallocDefault));
        b->sh_text_bytes[c] = b->sh_text_want;
    }
    if (runs >= 1) {
        int n_host = -1;                        // one run: K3's chunk table is the order, the count stays on the device
        if (runs > 1) {                         // several runs share the ring: count on the host, radix sort
            uint32_t cnt[4];
            HIP_TRY(hipStreamSynchronize(b->s_post));
            HIP_TRY(hipMemcpy(cnt, b->ring_count[c], 16, hipMemcpyDeviceToHost));
            n_host = (int) std::min<uint32_t>(cnt[0], (uint32_t) b->frame_cap);
        }
        if (n_host != 0) {
            uint8_t *sin = b->sd_seq[b->sd_seq_cur], *sout = b->sd_seq[b->sd_seq_cur ^ 1];
            HIP_TRY(hipMemcpyAsync(sout, sin, N, hipMemcpyDeviceToDevice, b->s_post));
            HIP_TRY(nmea_format_enqueue(b->ring[c], n_host, b->frame_cap, b->N, sin, sout, b->sd_text[c],
                                        b->sd_text_bytes[c], b->nmea_scratch, b->nmea_scratch_bytes, nullptr,
                                        b->ring_chunks[c], b->n_chunks, k3_passes(b->cand_K), &totals, b->s_post));
            b->sd_seq_cur ^= 1;
        }
    }
    // the span's frames into the carried vessel table, behind the formatter and before the ring is handed back
    if (b->vt && runs >= 1)
        HIP_TRY(vessel_table_update_enqueue(b->ring[c], b->ring_count[c], b->frame_cap, b->vt, b->vt_slots, b->vt_fslot,
                                            b->s_post));
    HIP_TRY(nmea_slot_info_enqueue(totals, b->ring_count[c], b->sd_info + 8 * c, b->s_post));
    HIP_TRY(hipMemsetAsync(b->ring_count[c], 0, 16, b->s_post));
    HIP_TRY(hipEventRecord(b->e_fmt[c], b->s_post));          // the ring is free for K3 again
    // (5) the copy has a stream of its own: it runs at PCIe speed beside the next slot's formatter
    hipStream_t sc = b->copy_on_k3 ? b->s_post : b->s_copy;
    if (!b->copy_on_k3) HIP_TRY(hipStreamWaitEvent(sc, b->e_fmt[c], 0));
    HIP_TRY(nmea_text_copy_enqueue(b->sd_text[c], b->sd_info + 8 * c, b->sh_text[c], b->sh_text_bytes[c],
                                   b->sh_info + 8 * c, b->copy_wgs, sc));
    HIP_TRY(hipEventRecord(b->e_txt[c], sc));
    b->s_stage[c] = 1;
    b->hdlc_calls = 0;
    b->stream_calls++;
    return rc_late;
}

int gnuais_batch_pending_frames(gnuais_batch *b, int *n_out)
{
    if (!b || !n_out) return fail(GNUAIS_E_ARG, "pending_frames: argument");
    if (b->streaming) return fail(GNUAIS_E_STATE, "pending_frames: the batch is streaming (gnuais_batch_stream_nmea)");
    if (int rc = gnuais_batch_sync(b)) return rc;
    uint32_t cnt[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(cnt, b->frame_count, sizeof cnt, hipMemcpyDeviceToHost));
    *n_out = (int) std::min<uint32_t>(cnt[0], (uint32_t) b->frame_cap);
    return GNUAIS_OK;
}

int gnuais_batch_discard_frames(gnuais_batch *b, void *stream)
{
    if (!b) return fail(GNUAIS_E_ARG, "discard_frames: NULL batch");
    if (b->streaming) return fail(GNUAIS_E_STATE, "discard_frames: the batch is streaming (gnuais_batch_stream_nmea)");
    if (int rc = set_device(b)) return rc;
    hipStream_t s = b->pipeline ? k3_stream(b) : (hipStream_t) stream;   // behind the last K3
    HIP_TRY(hipMemsetAsync(b->frame_count, 0, sizeof(uint32_t) * 3, s));
    b->hdlc_calls = 0;
    return GNUAIS_OK;
}

int gnuais_batch_counters(gnuais_batch *b, gnuais_counters *h_out)
{
    if (!b || !h_out) return fail(GNUAIS_E_ARG, "counters: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    const int N = b->N;
    std::vector<int32_t> v((size_t) N * 3);
    HIP_TRY(hipMemcpy(v.data(), b->counters, v.size() * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < N; ++c) {
        h_out[c].receivedframes = v[c];
        h_out[c].lostframes = v[(size_t) N + c];
        h_out[c].lostframes2 = v[(size_t) 2 * N + c];
    }
    return GNUAIS_OK;
}

int gnuais_batch_total_received(gnuais_batch *b, long long *total)
{
    if (!b || !total) return fail(GNUAIS_E_ARG, "total_received: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    std::vector<int32_t> v((size_t) b->N);
    HIP_TRY(hipMemcpy(v.data(), b->counters, v.size() * 4, hipMemcpyDeviceToHost));
    long long t = 0;
    for (int32_t x : v) t += x;
    *total = t;
    return GNUAIS_OK;
}

int gnuais_batch_maxval(gnuais_batch *b, int16_t *h_out)
{
    if (!b || !h_out) return fail(GNUAIS_E_ARG, "maxval: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    std::vector<int> v((size_t) b->N);
    HIP_TRY(hipMemcpy(v.data(), b->maxval[b->max_last], v.size() * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < b->N; ++c) h_out[c] = (int16_t) v[c];
    return GNUAIS_OK;
}

int gnuais_batch_pll_state(gnuais_batch *b, gnuais_pll_state *h_out)
{
    if (!b || !h_out) return fail(GNUAIS_E_ARG, "pll_state: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    std::vector<uint32_t> v((size_t) b->N), lb((size_t) b->N), pv((size_t) b->N);
    HIP_TRY(hipMemcpy(v.data(), b->pll, v.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lb.data(), b->lastbit, lb.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pv.data(), b->prev, pv.size() * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < b->N; ++c) {
        h_out[c].pll = v[c] & 0xffffu;
        h_out[c].prev = pv[c] & 1;
        h_out[c].lastbit = lb[c] & 1;
    }
    return GNUAIS_OK;
}

int gnuais_batch_fsm_state(gnuais_batch *b, gnuais_fsm_state *h_out)
{
    if (!b || !h_out) return fail(GNUAIS_E_ARG, "fsm_state: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    std::vector<uint32_t> v((size_t) b->N);
    HIP_TRY(hipMemcpy(v.data(), b->ctl, v.size() * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < b->N; ++c) {
        const uint32_t w = v[c];
        h_out[c].state = w & 7;
        h_out[c].nstartsign = (w >> 3) & 15;
        h_out[c].antallpreamble = (w >> 7) & 15;
        h_out[c].antallenner = (w >> 11) & 7;
        h_out[c].bitstuff = (w >> 14) & 1;
        h_out[c].last = (w >> 15) & 1;
        h_out[c].bufferpos = (w >> 16) & 511;
    }
    return GNUAIS_OK;
}

// protodec_reset() (protodec.c:87-100) for every receiver's decoder: the machine back to ST_SKURR with its counts
// cleared, a frame in progress dropped; receivedframes / lostframes / lostframes2 stay, as in the reference
int gnuais_batch_protodec_reset(gnuais_batch *b)
{
    if (!b) return fail(GNUAIS_E_ARG, "protodec_reset: NULL batch");
    if (int rc = gnuais_batch_sync(b)) return rc;
    HIP_TRY(launch_hdlc_fsm_reset(b->ctl, b->N, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return GNUAIS_OK;
}

// d->buffer (protodec.h:52, written at protodec.c:1019): the stored bits -- one per byte, stuffed 0s dropped -- of the
// frame a channel's decoder is in (ST_DATA / ST_STOPSIGN) or, between frames, of the last frame that reached its stop
// bit; *n_bits = -1 when neither is on record (no frame yet, or the last one was given up at 449 bits).  Rebuilt on the host from
// the candidate record the deframer keeps (raw bits incl. stuffing) and its partial word.
int gnuais_batch_frame_bits(gnuais_batch *b, int channel, uint8_t *h_bits, int cap, int *n_bits)
{
    if (!b || !h_bits || !n_bits || channel < 0 || channel >= b->N || cap < 0)
        return fail(GNUAIS_E_ARG, "frame_bits: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    uint32_t w[HDLC_CTL_WORDS];
    for (int q = 0; q < HDLC_CTL_WORDS; ++q)
        HIP_TRY(hipMemcpy(&w[q], b->ctl + (size_t) q * (size_t) b->N + channel, 4, hipMemcpyDeviceToHost));
    *n_bits = -1;
    const uint32_t state = w[0] & 7, nstart = w[3];
    if (nstart == 0) return GNUAIS_OK;
    uint32_t rec[CAND_WORDS];
    HIP_TRY(hipMemcpy(rec, b->cand + ((size_t) channel * b->cand_K + (nstart - 1) % (uint32_t) b->cand_K) * CAND_WORDS,
                      sizeof rec, hipMemcpyDeviceToHost));
    int rawlen;
    const bool open = state == 4 || state == 5;             // ST_DATA, ST_STOPSIGN (protodec.h:33-34)
    if (open) {
        rawlen = (int) w[4];
        if ((rawlen >> 5) < CAND_WORDS - CAND_HDR) rec[CAND_HDR + (rawlen >> 5)] = w[1];   // the word being filled
    } else if (rec[0] >> 17) {                              // closed with a stop bit, good (CAND_VALID) or bad: its length is on record
        rawlen = (int) ((rec[0] >> 17) & 0x3ffu);
    } else {
        return GNUAIS_OK;                                   // given up at 449 bits (protodec.c:1024-1026), or overwritten
    }
    if (rawlen < 0 || rawlen > 32 * (CAND_WORDS - CAND_HDR)) return fail(GNUAIS_E_STATE, "frame_bits: record length");
    int n = 0, ones = 0;
    for (int i = 0; i < rawlen; ++i) {
        const uint8_t x = (rec[CAND_HDR + (i >> 5)] >> (i & 31)) & 1u;
        if (ones == 5) { ones = 0; continue; }              // the 0 after five 1s: protodec.c:1002-1006
        if (n < cap) h_bits[n] = x;
        ++n;
        ones = x ? ones + 1 : 0;
    }
    *n_bits = n;
    return GNUAIS_OK;
}

int gnuais_batch_history(gnuais_batch *b, int16_t *h_out)
{
    if (!b || !h_out) return fail(GNUAIS_E_ARG, "history: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    const int N = b->N, NT = b->NT;
    std::vector<int16_t> v((size_t) N * NT);
    HIP_TRY(hipMemcpy(v.data(), b->hist[b->hist_cur], v.size() * 2, hipMemcpyDeviceToHost));
    for (int c = 0; c < N; ++c)
        for (int k = 0; k < NT; ++k) h_out[(size_t) c * NT + k] = v[(size_t) k * N + c];
    return GNUAIS_OK;
}

int gnuais_batch_last_signs(gnuais_batch *b, uint8_t *h_out, int stride)
{
    if (!b || !h_out || stride < b->last_len) return fail(GNUAIS_E_ARG, "last_signs: argument");
    if (int rc = gnuais_batch_sync(b)) return rc;
    const int N = b->N, W = (b->last_len + 31) / 32;
    std::vector<uint32_t> w(sgn_words_alloc(W, N));
    if (W) HIP_TRY(hipMemcpy(w.data(), b->sgn[b->last_k], w.size() * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < N; ++c)
        for (int n = 0; n < b->last_len; ++n)
            h_out[(size_t) c * stride + n] = (w[sgn_index(n >> 5, N, c)] >> (31 - (n & 31))) & 1u;
    return GNUAIS_OK;
}

int gnuais_batch_info(const gnuais_batch *b, const char *name, double *value)
{
    if (!b || !name || !value) return fail(GNUAIS_E_ARG, "info: argument");
    if (!strcmp(name, "sign_exact")) *value = b->sign_ok && b->fir_variant == 3;
    else if (!strcmp(name, "sign_eps")) {            // of the kernel the options select; FL2: the power of two it works with
        const float fs = !b->fir_flag2 ? 0.0f : b->sign_NC == 12 ? b->sign_fscale : 0.0f;
        *value = fs > 0.0f ? 2.0f / fs : b->sign_eps;
    }
    else if (!strcmp(name, "sign_flag_scale")) *value = !b->fir_flag2 ? 0.0f : b->sign_NC == 12 ? b->sign_fscale : 0.0f;
    else if (!strcmp(name, "sign_eps_seen") || !strcmp(name, "sign_eps_ahead")) {
        const bool pk40 = b->sign_NC == 48 && b->pk40_ok && b->fir_pk_taps != 48 && b->fir_inloop;
        *value = name[9] == 's' ? (pk40 ? b->pk40_eps_seen : b->sign_eps_seen) : (pk40 ? b->pk40_eps_ahead : b->sign_eps_ahead);
    }
    else if (!strcmp(name, "sign_matrix_pipe")) *value = (b->fir_mfma && b->mfma_ok && b->N % 64 == 0 && b->sign_NC == 48 && b->fir_variant == 3) ? 1 : 0;
    else if (!strcmp(name, "sign_central_taps")) *value = (b->sign_NC == 48 && b->pk40_ok && b->fir_pk_taps != 48 && b->fir_inloop) ? 40 : b->sign_NC;
    else if (!strcmp(name, "first_effective_tap")) *value = b->k0;
    else if (!strcmp(name, "n_effective_taps")) *value = b->NE;
    else if (!strcmp(name, "compute_units")) *value = b->n_cu;
    else if (!strcmp(name, "device")) *value = b->device;
    else if (!strcmp(name, "segments")) *value = b->n_seg;
    else if (!strncmp(name, "stream_of_stage_", 16) && name[16] >= '0' && name[16] <= '3' && !name[17]) {
        // which of the batch's POOL candidate streams (creation order) serves stage 0 K2, 1 spare, 2 K2b, 3 K3 right now
        *value = -1;
        for (int q = 0; q < gnuais_batch::POOL; ++q)
            if (b->pool[q] == b->s_k[name[16] - '0']) *value = q;
    }
    else if (!strcmp(name, "stream_depth")) *value = gnuais_batch::NRING - 1;
    else return fail(GNUAIS_E_ARG, "info: unknown name");
    return GNUAIS_OK;
}

int gnuais_batch_n_channels(const gnuais_batch *b) { return b ? b->N : 0; }
int gnuais_batch_n_taps(const gnuais_batch *b) { return b ? b->NT : 0; }

int gnuais_batch_set_timing(gnuais_batch *b, int on)
{
    if (!b) return fail(GNUAIS_E_ARG, "set_timing: NULL batch");
    b->timing = on != 0;
    if (on) b->timed_calls = 0;
    return GNUAIS_OK;
}

// ms[0] K1 fir_slice  [1] K2 pll  [2] K2b hdlc_deframe  [3] K3 hdlc_crc
// [4] first event to last event of the call
static int timing_of(gnuais_batch *b, unsigned long long call, float *ms)
{
    hipEvent_t *ev = b->evr[call % gnuais_batch::TIMING_RING];
    HIP_TRY(hipEventElapsedTime(&ms[0], ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms[1], ev[2], ev[6]));
    HIP_TRY(hipEventElapsedTime(&ms[2], ev[5], ev[7]));
    HIP_TRY(hipEventElapsedTime(&ms[3], ev[9], ev[4]));
    HIP_TRY(hipEventElapsedTime(&ms[4], ev[0], ev[4]));
    return GNUAIS_OK;
}

int gnuais_batch_last_timing(gnuais_batch *b, float *ms5)
{
    if (!b || !ms5) return fail(GNUAIS_E_ARG, "last_timing: argument");
    if (!b->timed_calls) return fail(GNUAIS_E_STATE, "last_timing: no timed run");
    if (int rc = gnuais_batch_sync(b)) return rc;
    return timing_of(b, b->timed_calls - 1, ms5);
}

int gnuais_batch_mean_timing(gnuais_batch *b, float *ms5, int *n_calls)
{
    if (!b || !ms5 || !n_calls) return fail(GNUAIS_E_ARG, "mean_timing: argument");
    if (!b->timed_calls) return fail(GNUAIS_E_STATE, "mean_timing: no timed run");
    if (int rc = gnuais_batch_sync(b)) return rc;
    const unsigned long long n = std::min<unsigned long long>(b->timed_calls, gnuais_batch::TIMING_RING);
    double acc[5] = {0, 0, 0, 0, 0};
    for (unsigned long long i = 0; i < n; ++i) {
        float t[5];
        if (int rc = timing_of(b, b->timed_calls - 1 - i, t)) return rc;
        for (int q = 0; q < 5; ++q) acc[q] += t[q];
    }
    for (int q = 0; q < 5; ++q) ms5[q] = (float) (acc[q] / (double) n);
    *n_calls = (int) n;
    return GNUAIS_OK;
}

int gnuais_crc16_batch(int device, const uint8_t *h_data, int stride, const int32_t *h_len,
                       int n_msgs, uint16_t *h_crc)
{
    if (!h_data || !h_len || !h_crc || stride <= 0 || n_msgs <= 0)
        return fail(GNUAIS_E_ARG, "crc16_batch: argument");
    HIP_TRY(hipSetDevice(device));
    uint8_t *d_data = nullptr;
    int32_t *d_len = nullptr;
    uint16_t *d_crc = nullptr;
    hipError_t e = hipMalloc((void **) &d_data, (size_t) stride * n_msgs);
    if (e == hipSuccess) e = hipMalloc((void **) &d_len, sizeof(int32_t) * n_msgs);
    if (e == hipSuccess) e = hipMalloc((void **) &d_crc, sizeof(uint16_t) * n_msgs);
    if (e == hipSuccess) e = hipMemcpy(d_data, h_data, (size_t) stride * n_msgs, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_len, h_len, sizeof(int32_t) * n_msgs, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_crc16(d_data, stride, d_len, n_msgs, d_crc, nullptr);
    if (e == hipSuccess) e = hipMemcpy(h_crc, d_crc, sizeof(uint16_t) * n_msgs, hipMemcpyDeviceToHost);
    (void) hipFree(d_data);
    (void) hipFree(d_len);
    (void) hipFree(d_crc);
    if (e != hipSuccess) return fail(GNUAIS_E_HIP, "crc16_batch", e);
    return GNUAIS_OK;
}

int gnuais_crc16_bits(int device, const uint8_t *h_bits, int n_bytes, uint16_t *h_crc, uint8_t *h_msb, int n_out)
{
    if (!h_bits || !h_crc || n_bytes <= 0 || n_bytes > 64 || n_out < 0 || n_out > 8 * n_bytes || (n_out > 0 && !h_msb))
        return fail(GNUAIS_E_ARG, "crc16_bits: argument (1..64 bytes, n_out <= 8 * n_bytes)");
    HIP_TRY(hipSetDevice(device));
    // one allocation: [bits 512][msb 512][crc]
    uint8_t *d = nullptr;
    hipError_t e = hipMalloc((void **) &d, 512 + 512 + 16);
    if (e == hipSuccess) e = hipMemcpy(d, h_bits, (size_t) n_bytes * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launch_crc16_bits(d, n_bytes, reinterpret_cast<uint16_t *>(d + 1024), n_out ? d + 512 : nullptr, n_out, nullptr);
    if (e == hipSuccess) e = hipMemcpy(h_crc, d + 1024, sizeof(uint16_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && n_out) e = hipMemcpy(h_msb, d + 512, (size_t) n_out, hipMemcpyDeviceToHost);
    (void) hipFree(d);
    if (e != hipSuccess) return fail(GNUAIS_E_HIP, "crc16_bits", e);
    return GNUAIS_OK;
}

int gnuais_tile_channels(const int16_t *d_base, int n_base, int len, int16_t *d_out,
                         int n_channels, void *stream)
{
    if (!d_base || !d_out || n_base <= 0 || len <= 0 || n_channels <= 0)
        return fail(GNUAIS_E_ARG, "tile_channels: argument");
    HIP_TRY(launch_tile_channels(d_base, n_base, len, d_out, n_channels, (hipStream_t) stream));
    return GNUAIS_OK;
}

} // extern "C"
