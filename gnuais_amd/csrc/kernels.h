// kernels.h -- internal launch interface between the C-ABI host code
// (gnuais_capi.hip) and the gfx950 kernels.  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gnuais {

// ---- K1: FIR + slicer (fir_slice.hip) -------------------------------------
struct FirLaunch {
    const int16_t *x;      // [L][N] interleaved input
    const int16_t *hist;   // [NT][N] previous call's last NT samples, oldest first
    uint32_t *sgn;         // sign words, bit 31 = oldest sample; layout: sgn_index() below
    float *dump;           // optional [L][N] filter output
    int *maxval;           // [N] peak positive sample (atomicMax into a zeroed buffer)
    int16_t *hist_out;     // [NT][N] the other history buffer: written by the specialised kernel
    int *maxval_next;      // [N] the other peak buffer: cleared by the specialised kernel
    const float *d_taps;   // device copy of all NT taps (generic kernel)
    float te[64];          // trimmed taps (specialised kernel)
    int N, L, T;           // T: outputs per wave, multiple of 32
    int NT, NE, d;         // out[n] = sum_j te[j] * x[n - d + j], j < NE
    float eps;             // sign-exact slicer: |central sum| > eps certifies the sign
    float fscale = 0;      // K1s direct form: > 0 = a power of two the central taps are scaled by so that the certified distance
                           //   is |y'| >= 2.0 and one v_alignbit_b32 gathers sign and exponent bit (fir_sign_kernel FL2); eps is then unused
    float eps_pk = 0;      // fir_sign_pk.hip, 12 taps: the bound for the transposed fused sum (the direct form's is eps)
    float eps_seen = 0, eps_ahead = 0;     // 48-tap K1s with the running maximum (eps_seen > 0): eps = eps_seen * M / 32768 + eps_ahead
    float eps_seen_k[4] = {}, eps_ahead_k[4] = {};   // fir_sign_pk.hip, 40 / 48 taps: the same for an output that completes >= 6, 4, 2, 0 rows
                                                     //   before the end of its 16-row group (those rows are under the maximum M as well)
    int NC;                //   central taps used (12 or 48)
    float ctaps[48];       //   te[(NE-NC)/2 .. +NC)
    const float *te_mem;   //   the NE effective taps in device memory (exact re-evaluation)
    int map;               // K1s workgroup -> (channel group, segment) mapping, see fir_sign_kernel
    int max_segments = 0;  // fir_sign_pk.hip: > 0 = only the call's first segments (the rest is fir_sign_mfma.hip's)
    const struct MfmaTaps *mfma = nullptr;   // fir_sign_mfma.hip: the device copy of its integer taps; eps_seen / eps_ahead are then in ITS units
};
int launch_fir_sign_quantum(int NC);           // T must be a multiple of this
hipError_t launch_fir_sign(const FirLaunch &a, hipStream_t stream);
// K1s in transposed form on register pairs (fir_sign_pk.hip): NC 12 (32-tap table), 40 or 48
int launch_fir_sign_pk_quantum(int NC);
// K1s, 48 central taps as an exact integer Toeplitz product on the matrix pipe (fir_sign_mfma.hip): outputs first .. of a call
struct alignas(16) MfmaTaps { int a[3][3][64][4]; int k0; };   // [block of 32 window rows][tap digit 0..2][lane][16 bytes]: the A operands; 128 * sum of the integer taps
bool fir_sign_mfma_taps(const float *tc48, MfmaTaps *out, double *scale, double *bound_q);
int launch_fir_sign_mfma_quantum();
hipError_t launch_fir_sign_mfma(const FirLaunch &a, int first, hipStream_t stream);
hipError_t launch_fir_sign_pk(const FirLaunch &a, hipStream_t stream);
hipError_t launch_fir_generic(const FirLaunch &a, hipStream_t stream);
hipError_t launch_fir_history(const int16_t *x, const int16_t *hist_in, int16_t *hist_out,
                              int N, int L, int NT, hipStream_t stream);

// ---- K2: PLL clock recovery, slice + NRZI (pll_h3.hip, pll_tp.hip) ----------------
constexpr int PLL_LDS_BYTES = 81 * 1024;   // > half of a CU's LDS: one PLL workgroup per CU
constexpr int SEG_WORDS = 64;        // segment (one bit pack per channel): 64 sign words
constexpr int SEG_LEN = SEG_WORDS * 32;    //   = 2048 samples
constexpr int PACK_STRIDE = 16;      // words reserved per (channel, segment) bit pack: 64 bytes
// Sign words: word w (samples 32w .. 32w+31, bit 31 = oldest) of channel c.  Four consecutive
// words of a channel lie side by side, so that K2 fetches 128 samples of a channel with one
// 16-byte load and a wave's fetch is 1 KB of contiguous memory.
__host__ __device__ inline size_t sgn_index(int w, int N, int c)
{
    return ((size_t) (w >> 2) * (size_t) N + (size_t) c) * 4 + (size_t) (w & 3);
}
__host__ __device__ inline size_t sgn_words_alloc(int W, int N)   // words for W sign words per channel
{
    return (size_t) ((W + 3) / 4 + 1) * 4 * (size_t) N;
}
struct PllLaunch {
    const uint32_t *sgn;   // sgn_index() layout
    uint32_t *pll;         // [N] phase (receiver.h:40), carried
    uint32_t *prev;        // [N] sign of the last sample (receiver.h:44), carried
    uint32_t *lastbit;     // [N] level at the last slice (receiver.h:38), carried
    uint32_t *watchdog;    // one word: set to 1 if a wave of the launch gave up waiting for its partner
    uint32_t *segbits;     // [N][n_seg][PACK_STRIDE] recovered bits per segment, LSB first
    uint32_t *segcnt;      // [N][n_seg] bits in each pack
    int N, L, n_seg, seg_words;
    uint32_t pllinc;
    int n_cu;              // compute units of the batch's device
    int variant = 0;       // 0: by channel count; 7: the time-parallel form (pll_tp.hip); 8: one recurrence wave + three helpers (pll_h3.hip)
};
int pll_form_of(const PllLaunch &a);                                     // the form launch_pll() will take: 7 or 8
int pll_need_lds();                                                      // bytes of LDS a PLL workgroup cannot do without
hipError_t pll_prepare_device();                                         // once per device, after hipSetDevice
hipError_t launch_pll(const PllLaunch &a, hipStream_t stream);           // K2
// K2 in its time-parallel form (pll_tp.hip): a workgroup per channel, lanes = candidate phases; small batches
bool pll_tp_applicable(const PllLaunch &a);
hipError_t launch_pll_tp(const PllLaunch &a, hipStream_t stream);
constexpr int PLL_TP_MAX_CHANNELS = 1536; // launch_pll() takes the time-parallel form by itself up to this many channels (48 000 samples: 0.066 / 0.118 / 0.219 / 0.423 ms at 512 / 1024 / 2048 / 4096 against pll_h3's 0.32)

// ---- K2b: HDLC deframer, K3: CRC-16 + frame delivery (hdlc_crc.hip) ---------
constexpr int HDLC_CTL_WORDS = 6;
constexpr int HDLC_BUF_WORDS = 15;   // 449 bits max (protodec.c:1024)
constexpr int CAND_HDR = 2;          // [0] nbits | valid flag | raw length << 17 | end_bit[36:32] << 27, [1] end_bit[31:0]
constexpr int CAND_WORDS = 20;       // header + 17 raw words (449 frame bits + stuffing) = 80 bytes
struct HdlcLaunch {
    const uint32_t *segbits;  // as above (PACK_STRIDE words per pack)
    const uint32_t *segcnt;
    uint32_t *ctl;         // [HDLC_CTL_WORDS][N] control state
    uint32_t *cand;        // [N][K][CAND_WORDS] per-channel ring of candidate frames
    uint32_t *cand_first;  // [N] first slot closed in this call
    uint32_t *cand_count;  // [N] slots closed in this call
    int32_t *counters;     // [3][N] receivedframes, lostframes, lostframes2
    void *frames;          // gnuais_frame[frame_cap]
    uint32_t *frame_count; // [4]: frames appended, overflow flag, -, PLL watchdog
    uint32_t frame_cap;
    int N, n_seg, seg_words, K;   // K: slots of a channel's candidate ring
    int K_call = 0;        // most frame starts one call can have (<= K; 0: K); sizes the chunk table
    int lanes_per_wave;    // channels per wave in K2b (blockDim)
    uint2 *chunks = nullptr;   // optional [blocks of K3][k3_passes(K)]: where each pass of each K3 block put its
                           // frames in the ring (start, count).  (block, pass, position) is the reference's
                           // print order -- channel, then time -- so a ring that holds ONE call needs no sort
};
constexpr int K3_CH = 32;               // channels per K3 block
inline int k3_blocks(int N) { return (N + K3_CH - 1) / K3_CH; }
inline int k3_passes(int K) { return (K3_CH * K + 255) / 256; }
hipError_t launch_hdlc_deframe(const HdlcLaunch &a, hipStream_t stream); // K2b, window by window (hdlc_crc.hip)
hipError_t launch_hdlc_events(const HdlcLaunch &a, hipStream_t stream);  // K2b, event by event (hdlc_events.hip)
hipError_t launch_hdlc_crc(const HdlcLaunch &a, hipStream_t stream);     // K3
hipError_t launch_hdlc_reset(uint32_t *ctl, int N, hipStream_t stream);
hipError_t launch_hdlc_fsm_reset(uint32_t *ctl, int N, hipStream_t stream);   // protodec_reset() only: counters and time stay

// ---- utilities (util.hip) ---------------------------------------------------
hipError_t launch_tile_channels(const int16_t *base, int n_base, int len, int16_t *out,
                                int n_channels, hipStream_t stream);
hipError_t launch_crc16(const uint8_t *data, int stride, const int32_t *len, int n,
                        uint16_t *crc, hipStream_t stream);
// one frame's bits (a byte per bit) -> CRC + the bits byte-wise most-significant-first (util.hip)
hipError_t launch_crc16_bits(const uint8_t *bits, int n_bytes, uint16_t *crc, uint8_t *msb, int n_out,
                             hipStream_t stream);

// K1s evaluates the NC = 12 central taps in direct form with symmetric pre-adds (fir_slice.hip); the
// host's error bound for y_c follows the same order of operations (gnuais_capi.hip)
constexpr bool K1S_DIRECT(int nc) { return nc <= 12; }

// ---- f1 on the device (nmea_device.hip) ---------------------------------------
size_t nmea_scratch_bytes(int n_frames, int n_chunks = 0);
// frames: device gnuais_frame[n]; seq_in/seq_out: device u8[n_channels] (seq_out preloaded with
// seq_in); out: device text buffer.  h_info: [0] bytes written, [1] sentences, [2] != 0 if a
// frame named a channel >= n_channels.  Synchronises `s`.
// frames[n] (device) -> out[n] (device) in print order: channel, then end_bit
hipError_t frames_sort(const struct gnuais_frame *frames, int n, struct gnuais_frame *out, void *scratch,
                       size_t scratch_bytes, hipStream_t s);
// the device part only, queued without waiting; h_info4 (host, pinned): [0] + [1] bytes written,
// [2] sentences, [3] != 0 if a frame named a channel >= n_channels -- valid once `s` has got there
// n > 0: count known to the host, order by radix sort.  n < 0: the ring holds exactly one call; order and
// count (<= n_max) come from K3's chunk table (HdlcLaunch::chunks), nothing is read back.  *totals_dev:
// the device words h_info4 is copied from (h_info4 may be null).
hipError_t nmea_format_enqueue(const struct gnuais_frame *frames, int n, int n_max, int n_channels,
                               const uint8_t *seq_in, uint8_t *seq_out, char *out, size_t out_cap, void *scratch,
                               size_t scratch_bytes, uint32_t *h_info4, const uint2 *chunks, int n_chunks,
                               int chunk_passes, uint32_t **totals_dev, hipStream_t s);
// info8 (device): the formatter's four words (totals; null: nothing was formatted) + the ring's four counters
hipError_t nmea_slot_info_enqueue(const uint32_t *totals, const uint32_t *ring_count, uint32_t *info8_dev, hipStream_t s);
// device text -> pinned host text, length taken from info8 on the device; the info words follow it to the host
hipError_t nmea_text_copy_enqueue(const char *src, const uint32_t *info8_dev, char *dst_pinned, size_t dst_cap,
                                  uint32_t *info8_pinned, int workgroups, hipStream_t s);
// the stdout lines of protodec_getdata() for the same frames, right after nmea_format_enqueue(n > 0) on the same
// scratch and stream; lines: n * messages_line_bytes(), len / off: n words; info2 (device): [0] bytes, [1] lines
size_t messages_line_bytes();
hipError_t messages_format_enqueue(const struct gnuais_frame *frames, int n, int n_channels, const uint8_t *seq_in,
                                   const char *chanid_dev, void *scratch, size_t scratch_bytes, char *lines,
                                   uint32_t *len, uint32_t *off, char *out, size_t out_cap, uint32_t *info2,
                                   hipStream_t s);
// the batch's vessel table (gnuais_vessel per MMSI, sorted) folded on the device from the ring's frames
hipError_t vessels_fold_enqueue(const struct gnuais_frame *frames, int n, void *scratch, size_t scratch_bytes,
                                struct gnuais_vessel *out, int cap, uint32_t *count_dev, hipStream_t s);
// the vessel table carried on the device from batch to batch (one allocation of vessel_table_bytes(slots), zeroed;
// slots a power of two)
size_t vessel_table_bytes(uint32_t slots);
hipError_t vessel_table_update_enqueue(const struct gnuais_frame *frames, const uint32_t *count, int n_max, void *table,
                                       uint32_t slots, uint32_t *fslot, hipStream_t s);
hipError_t vessel_table_fetch(const void *table, uint32_t slots, struct gnuais_vessel *h_out, int max, int *n_out,
                              uint32_t h_info[4], hipStream_t s);
hipError_t nmea_format(const struct gnuais_frame *frames, int n, int n_channels, const uint8_t *seq_in,
                       uint8_t *seq_out, char *out, size_t out_cap, void *scratch, size_t scratch_bytes,
                       uint32_t *h_info, hipStream_t s);

} // namespace gnuais
