// fir_sign_pk.hip -- K1s in transposed form on register PAIRS (gfx950): v_pk_fma_f32 does two of the slicer's
// multiply-adds for the issue cost of one.
//
// Same contract as fir_sign_kernel (fir_slice.hip): the sign-exact slicer standing in for filter_run_buf() + the
// `out > 0` test of receiver_run() (gnuais src/filter.c:106-143, src/receiver.c:109-111,126); bit-identical sign
// words, peak and history carry.  What changes is how y_c, the sum over the NC central taps, is formed.
//
// Measured on this chip (scripts/ubench/valu_rate.hip, four waves per SIMD): v_fmac_f32 with the tap in an SGPR --
// what the scalar kernels issue -- costs 2.0 ns per wave-instruction per SIMD, v_pk_fma_f32 with the taps in an SGPR
// PAIR 2.1 ns for two multiply-adds, with or without op_sel swizzles on its operands.  The 48-tap kernel is bound by
// exactly that (48 x 2.0 ns + flags = 106 ns per sample and wave = the 5.1 ms it takes for C5), the 12-tap kernel
// spends half of its issue time there.
//
// Transposed form: sample x_i adds tc[q] * x_i to output o = i - q, q = 0 .. NC-1; the NC outputs in flight are a ring
// of NC accumulators, here NC/2 register pairs (slots a, a+1 with a even).  Samples are taken two at a time (x_P, x_P+1
// in one register pair, P even), each pair of samples is 2 x NC/2 packed instructions:
//   slot pair a, k = (P - a) mod NC (even):
//     acc[a:a+1] += { tc[k],   tc[k-1] } * { x_P,   x_P   }        "odd" tap pair  O(k)
//     acc[a:a+1] += { tc[k+1], tc[k]   } * { x_P+1, x_P+1 }        reversed "even" tap pair E(k) = { tc[k], tc[k+1] }
// The table is bitwise symmetric (tc[q] = tc[NC-1-q]), so E(NC-2-k) = reverse(E(k)) and O(NC-k) = reverse(O(k)): only
// NC/4 even and NC/4 + 1 odd pairs exist (NC + 2 SGPRs), the other half is the same registers with the halves
// swapped by op_sel; the sample broadcast is op_sel too.  After the first instruction group slot P+1 is complete (its
// last term was tc[NC-1] * x_P): it is read and cleared; after the second, slot P+2.  Every accumulator receives its
// NC products in sample order, one fused rounding each -- the arithmetic of the scalar transposed form, so the host's
// bound for it (ordered sum of NC fused terms, gnuais_capi.hip) carries over unchanged.
//
// Everything around the sum (typed loads, flags through v_alignbit_b32, the silence test, the exact ordered NE-tap
// re-evaluation of uncertified samples, the running window maximum that scales eps for long tables, peak, carry) is
// that of fir_sign_kernel's grouped branch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <utility>
#include "kernels.h"

namespace gnuais {

namespace {

typedef float pk_f2 __attribute__((ext_vector_type(2)));
typedef int pk_v4i __attribute__((ext_vector_type(4)));
extern "C" __device__ float pk_load_format_f32(pk_v4i rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.format.f32");

__device__ __forceinline__ int pk_load_sample(const int16_t *__restrict__ x, const int16_t *__restrict__ hist,
                                              int m, int N, int NT, int c)
{
    const int16_t *p = (m >= 0) ? (x + (size_t) m * (size_t) N + c) : (hist + (size_t) (NT + m) * (size_t) N + c);
    return (int) *p;
}

template <int I> struct PkIc { static constexpr int value = I; };
template <class F, int... Is>
__device__ __forceinline__ void pk_expand(F &&f, std::integer_sequence<int, Is...>) { (f(PkIc<Is>{}), ...); }
template <class F, int... Is>
__device__ __forceinline__ void pk_expand_void(F &&f, std::integer_sequence<int, Is...>) { (f(PkIc<Is>{}), ...); }
template <class F, int... Is>
__device__ __forceinline__ bool pk_expand_and(F &&f, std::integer_sequence<int, Is...>) { return (f(PkIc<Is>{}) && ...); }

constexpr int pk_gcd(int a, int b) { return b == 0 ? a : pk_gcd(b, a % b); }
constexpr int pk_unroll(int nc) { return 48 % nc == 0 ? 48 : nc * 16 / pk_gcd(nc, 16); }

template <int NC>
struct PkTaps {
    pk_f2 E[NC / 4];            // E[j] = { tc[2j], tc[2j+1] },   j < NC/4
    pk_f2 O[NC / 4 + 1];        // O[j] = { tc[2j], tc[2j-1] },   j <= NC/4, tc[-1] = tc[NC-1]
};
constexpr int PK_EXACT_BATCH = 6;   // loads in flight in the ordered sum (8 cost the 48-tap kernel a 145th register: 152 per wave instead of 144)
constexpr int PK_PEND = 8;          // noted outputs per lane (1 KB of LDS per wave); more than that are settled on the spot
template <int NES> struct PkExact { float te[NES > 0 ? NES : 1]; };
// INLOOP: eps = seen * M / 32768 + ahead with M the running maximum of |x| over the rows behind AND the rest of the
// output's own 16-row group; [0..3]: the output completes >= 6, 4, 2, 0 rows before the group's end (pair steps 0-4, 5,
// 6, 7) -- the taps that reach beyond those rows are priced with |x| = 32768 in `ahead` (gnuais_capi.hip)
struct PkEps { float seen[4], ahead[4]; };

constexpr int PK_WARM_BATCH = 8;    // warm-up rows loaded at a time
#include "fir_sign_pk_asm.inc"       // generated: the pair steps as fixed-register instruction streams (scripts/gen_fir_pk_asm.py)

// NES > 0: the table's NES effective taps travel in SGPRs for the exact re-evaluation (reference table: 32);
// NES == 0: they are read from te_mem.  INLOOP: eps follows a running window maximum (long tables).
template <int NES, int NC, bool INLOOP>
__device__ __forceinline__ void fir_sign_pk_body(
    const int16_t *__restrict__ x, const int16_t *__restrict__ hist,
    uint32_t *__restrict__ sgn, int *__restrict__ maxval,
    int16_t *__restrict__ hist_out, int *__restrict__ maxval_next,
    const float *__restrict__ te_mem, int N, int L, int T, int d, int NTaps, int NE_rt, float eps_up,
    PkEps ek, int map, PkTaps<NC> tp, PkExact<NES> ex)
{
    // the unrolled body holds whole turns of the accumulator ring AND whole 16-row groups: 48 phases for 12 and 48 taps, 80 for 40
    constexpr int GROUP = 16, UNROLL = 48 % NC == 0 ? 48 : NC * GROUP / pk_gcd(NC, GROUP), NG = UNROLL / GROUP, NP = NC / 2;
    static_assert(NC % 4 == 0 && UNROLL % NC == 0 && UNROLL % GROUP == 0 && UNROLL <= 96, "whole turns of the ring, whole groups");
    const int NE = NES > 0 ? NES : NE_rt;
    const int J0 = (NE - NC) / 2;
    const int lane = threadIdx.x;
    int bx = (int) blockIdx.x, by = (int) blockIdx.y;
    if (map == 1) {                             // XCD-contiguous channel groups, as in fir_sign_kernel
        const int G = (int) gridDim.x, id = by * G + bx, per = G >> 3;
        bx = (id & 7) * per + (id >> 3) % per;
        by = (id >> 3) / per;
    }
    const int cg = bx * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int t0 = by * T;                      // T is a multiple of 384 (whole unrolled turns, whole 16-byte sign stores)
    const int t1 = (t0 + T < L) ? t0 + T : L;
    if (t0 >= L) return;
    const int dc = d - J0;                      // y_c[n] = sum_q tc[q] * x[n - dc + q]
    const int m0 = t0 - dc;                     // local sample i <-> row m0 + i; output o completes with sample o + NC - 1

    // A segment whose every reference window lies inside this call's input reads the window through a typed buffer
    // descriptor based at its first row (fir_slice.hip: fir_sign_kernel has the reasoning): two VALU instructions
    // per tap instead of twenty.
    const uint32_t e_rowbytes = (uint32_t) N * 2u;
    const int e_row = t0 - d;
    const bool e_inner = e_row >= 0 && t1 - 1 - d + NE - 1 <= L - 1 &&
                         (unsigned long long) (t1 - t0 + NE) * e_rowbytes < 0x7fffffffull;
    const unsigned long long e_span = e_inner ? (unsigned long long) (L - e_row) * e_rowbytes : 0ull;
    const unsigned long long e_base = (unsigned long long) (x + (size_t) (e_inner ? e_row : 0) * (size_t) N);
    const pk_v4i rsrc_e = {(int) (e_base & 0xffffffffull), (int) ((e_base >> 32) & 0xffffull),
                           (int) (e_span > 0xffffffffull ? 0xffffffffull : e_span), 0x13004};    // R | SSCALED | 16
    auto exact_positive = [&](int n) __attribute__((always_inline)) -> bool {  // filter.h:40-49 order
        float sum = 0.0f;
        if (e_inner) {
            const int voff = (n - t0) * (int) e_rowbytes + c * 2;
            if constexpr (NES > 0) {
#pragma unroll
                for (int j0 = 0; j0 < NES; j0 += 8) {
                    float xs[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        xs[j] = pk_load_format_f32(rsrc_e, voff, (int) ((uint32_t) (j0 + j < NES ? j0 + j : NES - 1) * e_rowbytes), 0);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j0 + j < NES) sum = sum + ex.te[j0 + j] * xs[j];
                }
                return sum > 0.0f;
            }
            for (int j0 = 0; j0 < NE; j0 += PK_EXACT_BATCH) {
                float xs[PK_EXACT_BATCH];
#pragma unroll
                for (int j = 0; j < PK_EXACT_BATCH; ++j) {
                    const int jj = j0 + j < NE ? j0 + j : NE - 1;
                    xs[j] = pk_load_format_f32(rsrc_e, voff, (int) ((uint32_t) jj * e_rowbytes), 0);
                }
#pragma unroll
                for (int j = 0; j < PK_EXACT_BATCH; ++j)
                    if (j0 + j < NE) sum = sum + te_mem[j0 + j] * xs[j];
            }
            return sum > 0.0f;
        }
        if constexpr (NES > 0) {
#pragma unroll
            for (int j = 0; j < NES; ++j) {
                const float xs = (float) pk_load_sample(x, hist, n - d + j, N, NTaps, c);
                sum = sum + ex.te[j] * xs;
            }
            return sum > 0.0f;
        }
        for (int j0 = 0; j0 < NE; j0 += PK_EXACT_BATCH) {      // a few loads in flight: this rare path must not set the kernel's register count
            int xs[PK_EXACT_BATCH];
#pragma unroll
            for (int j = 0; j < PK_EXACT_BATCH; ++j) {
                const int jj = j0 + j < NE ? j0 + j : NE - 1;
                xs[j] = pk_load_sample(x, hist, n - d + jj, N, NTaps, c);
            }
#pragma unroll
            for (int j = 0; j < PK_EXACT_BATCH; ++j)
                if (j0 + j < NE) sum = sum + te_mem[j0 + j] * (float) xs[j];
        }
        return sum > 0.0f;
    };
    auto all_zero = [&](int m_first, int count) __attribute__((always_inline)) -> bool {
        uint32_t o = 0;
        for (int i = 0; i < count; ++i) {
            int m = m_first + i;
            m = m < -NTaps ? -NTaps : (m > L - 1 ? L - 1 : m);
            o |= (uint32_t) pk_load_sample(x, hist, m, N, NTaps, c);
        }
        return o == 0;
    };

    const int row0 = m0 > 0 ? m0 : 0;
    const uint32_t rowbytes = (uint32_t) N * 2u;
    const unsigned long long span = (unsigned long long) (L - row0) * rowbytes;
    const unsigned long long xbase = (unsigned long long) (x + (size_t) row0 * (size_t) N);
    const pk_v4i rsrc_f = {(int) (xbase & 0xffffffffull), (int) ((xbase >> 32) & 0xffffull),
                           (int) (span > 0xffffffffull ? 0xffffffffull : span), 0x13004};     // R | SSCALED | 16
    const int coff = c * 2;

    // The accumulator ring (and, for 48 taps, the odd tap pairs) sits in fixed registers above PK_VGPR_BASE, outside
    // the compiler's budget; everything that touches it is one of the generated instruction streams.
    (void) NP;
    pk_zero_ring<NC>(tp);
    // warm-up: samples i = 0 .. NC-2 (ring phase i + 1); nothing they complete is an output of this segment.  The loads
    // of up to eight steps go out together (the steps are volatile asm: a load issued between two of them would be
    // waited for there, one round trip per step)
    {
        constexpr int NW = (NC - 2) / 2;                        // full pair steps after the half step of sample 0
        pk_f2 w0;
        w0[0] = 0.0f;
        w0[1] = (float) pk_load_sample(x, hist, m0, N, NTaps, c);
        auto batch = [&](auto B0) __attribute__((always_inline)) {
            constexpr int s0 = decltype(B0)::value * PK_WARM_BATCH;
            constexpr int n = NW - s0 < PK_WARM_BATCH ? NW - s0 : PK_WARM_BATCH;
            pk_f2 w[n];
#pragma unroll
            for (int k = 0; k < n; ++k) {
                const int P = 2 + 2 * (s0 + k);                 // phases P, P+1 = samples P-1, P
                w[k][0] = (float) pk_load_sample(x, hist, m0 + P - 1, N, NTaps, c);
                w[k][1] = (float) pk_load_sample(x, hist, m0 + P, N, NTaps, c);
            }
            if constexpr (s0 == 0) {
                pk_warm<NC, 0, true>(w0, tp);                   // phase 0 is empty, phase 1 = sample 0
            }
            auto warm = [&](auto S) __attribute__((always_inline)) {
                constexpr int k = decltype(S)::value;
                constexpr int P = 2 + 2 * (s0 + k);
                pk_warm<NC, P, false>(w[k], tp);
            };
            pk_expand(warm, std::make_integer_sequence<int, n>{});
        };
        pk_expand(batch, std::make_integer_sequence<int, (NW + PK_WARM_BATCH - 1) / PK_WARM_BATCH>{});
    }

    int peakbits = 0;
    uint32_t neg = 0, amb = 0, zor = 0;
    bool zprev_known = false, zprev = false;
    float eps_w = eps_up, Mn = 0.0f;            // INLOOP: Mn = the running maximum / 32768, eps_w follows the pair step's class
    // maxima of |x|: [0] this turn of NG groups so far, [1] the turn before, ...: whole turns that hold the 96 rows behind a
    // group, which cover NC - 1 + J0 (48 taps: J0 <= 49, 40 taps: J0 <= 57)
    constexpr int NHIST = 1 + (96 + UNROLL - 1) / UNROLL;
    static_assert(NHIST == 3, "v_max3 of three maxima");
    float hmax[NHIST];
    if constexpr (INLOOP) {
        float Pm = 0.0f;
        const int plo = m0 - J0, phi = m0 + NC - 2;
        for (int mbase = plo; mbase <= phi; mbase += 16) {
            int v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int m = mbase + i <= phi ? mbase + i : phi;
                m = m < -NTaps ? -NTaps : m;
                v[i] = pk_load_sample(x, hist, m, N, NTaps, c);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) Pm = __builtin_fmaxf(Pm, __builtin_fabsf((float) v[i]));
        }
#pragma unroll
        for (int k = 0; k < NHIST; ++k) hmax[k] = Pm;
    }

    uint32_t wq[4] = {0u, 0u, 0u, 0u};
    __shared__ uint16_t pend[PK_PEND * 64];     // per lane: outputs (relative to t0; < 65536, the launcher caps T) whose sign the central sum left open
    int n_pend = 0;
    auto flush = [&](int obase) __attribute__((always_inline)) {               // one finished sign word: outputs obase .. obase+31
        const int mb = m0 + NC - 1 + obase;
        uint32_t w = ~neg;
        const int valid = t1 - (t0 + obase);
        if (valid < 32) {
            w &= ~0u << (32 - valid);
            amb &= ~0u << (32 - valid);
        }
        bool zc_known = false, zc = false;
        if (__popc(amb) >= 8) {                 // a silent stretch?  (fir_sign_kernel)
            zc = zor == 0;
            zc_known = true;
            const int before = J0 + NC - 1;
            if (zc && ((before <= 32 && zprev_known) ? zprev : all_zero(mb - before, before)) && all_zero(mb + 32, J0)) {
                w &= ~amb;
                amb = 0;
            }
        }
        zprev_known = zc_known;
        zprev = zc;
        // What is left needs the reference's ordered sum.  One lane doing that while 63 wait was 2 of the 48-tap
        // kernel's 36 instructions per sample (a quarter of a wave's words hold one such sample somewhere); instead the
        // lane notes the output, leaves the bit 0, and all lanes work their lists off TOGETHER -- the k-th noted output of
        // every lane at the same time -- after the segment's last word, or at once when a lane has more than its list
        // holds.  (The rows differ per lane, the channel does not: as many cache lines per load as lanes at work, the
        // same lines the one-at-a-time form fetched.)  A word that has left for memory gets its bit by an atomic OR (this
        // lane stored it: one location, one thread, program order); the up to four words still in registers are patched
        // there.  Same scheme as fir_sign_kernel's.
        w &= ~amb;
        const int slot = (obase >> 5) & 3;
        const bool last = t0 + obase + 32 >= t1;
        {
            while (amb && n_pend < PK_PEND) {
                const int pos = __clz((int) amb);
                amb &= ~(0x80000000u >> pos);
                pend[n_pend * 64 + lane] = (uint16_t) (obase + pos);
                ++n_pend;
            }
        }
        if (slot == 0) wq[0] = w; else if (slot == 1) wq[1] = w; else if (slot == 2) wq[2] = w; else wq[3] = w;
        if (__any(amb != 0) || (last && __any(n_pend != 0))) {
            const int held = (obase >> 5) - slot;                   // first word that is still in wq[]
            for (int k = 0;; ++k) {
                int o = -1;
                if (k < n_pend) {
                    o = (int) pend[k * 64 + lane];
                } else if (amb) {
                    const int pos = __clz((int) amb);
                    amb &= ~(0x80000000u >> pos);
                    o = obase + pos;
                }
                if (!__any(o >= 0)) break;
                if (o < 0 || !exact_positive(t0 + o)) continue;
                const uint32_t bit = 0x80000000u >> (o & 31);
                const int wi = (o >> 5) - held;
                if (wi < 0) {
                    if (live) atomicOr(sgn + sgn_index((t0 + o) >> 5, N, cg), bit);
                } else {
                    wq[0] |= wi == 0 ? bit : 0u;
                    wq[1] |= wi == 1 ? bit : 0u;
                    wq[2] |= wi == 2 ? bit : 0u;
                    wq[3] |= wi == 3 ? bit : 0u;
                }
            }
            n_pend = 0;
        }
        if (live && (slot == 3 || last)) {
            uint32_t *dst = sgn + sgn_index(((t0 + obase) >> 5) - slot, N, cg);
            if (slot == 3) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
            } else {
                dst[0] = wq[0];
                if (slot >= 1) dst[1] = wq[1];
                if (slot >= 2) dst[2] = wq[2];
            }
        }
        zor = 0;
    };

    const int ngroups = (t1 - t0 + 31) / 32 * 2;
    // rows mbx .. mbx + GROUP - 1 as floats; rows past the call are clamped (their outputs are masked), rows before it
    // come from the history
    auto load_group = [&](int mbx, pk_f2 *dst) __attribute__((always_inline)) {
        if ((mbx >= 0) && (mbx + GROUP - 1 < L)) {
            // the row in the SCALAR offset (the taps no longer occupy the SGPRs): no vector add per load
#pragma unroll
            for (int p = 0; p < GROUP; ++p)
                dst[p / 2][p % 2] = pk_load_format_f32(rsrc_f, coff, (int) ((uint32_t) (mbx - row0 + p) * rowbytes), 0);
        } else {
            // typed loads here as well (no 64-bit address arithmetic: this path must not set the register count):
            // the row is wave-uniform, so history or input is a scalar select of the descriptor
            const unsigned long long hbase = (unsigned long long) hist;
#pragma unroll
            for (int p = 0; p < GROUP; ++p) {
                const int m = mbx + p;
                const bool h = m < 0;
                const int mm = m < L ? m : L - 1;
                const pk_v4i r = {h ? (int) (hbase & 0xffffffffull) : rsrc_f[0], h ? (int) ((hbase >> 32) & 0xffffull) : rsrc_f[1],
                                  h ? (int) ((uint32_t) NTaps * rowbytes) : rsrc_f[2], 0x13004};
                dst[p / 2][p % 2] = pk_load_format_f32(r, coff, (int) ((uint32_t) (h ? NTaps + m : mm - row0) * rowbytes), 0);
            }
        }
    };
    constexpr bool PF = false;      // (the next group's rows requested a group ahead: level for the long tables, profiles/r04_c5_ring_and_segments.txt)
    pk_f2 xn[GROUP / 2];
    if constexpr (PF) load_group(m0 + NC - 1, xn);
    for (int b = 0; b * NG < ngroups; ++b) {
        auto group = [&](auto GI) __attribute__((always_inline)) -> bool {
            constexpr int g = decltype(GI)::value;
            const int gi = b * NG + g;
            if (gi >= ngroups) return false;
            const int gbase = gi * GROUP;                           // outputs gbase .. gbase + 15
            const int mb = m0 + NC - 1 + gbase;                     // row of the group's first phase
            const bool interior = (mb >= 0) && (mb + GROUP - 1 < L);
            pk_f2 xp[GROUP / 2];                                    // the group's rows as floats (exact), two to a register pair
            if constexpr (PF) {
            // this group's rows were asked for one group ago; the next group's go out now, ahead of this group's steps
#pragma unroll
            for (int p = 0; p < GROUP / 2; ++p) xp[p] = xn[p];
            if (gi + 1 < ngroups) load_group(mb + GROUP, xn);
            } else {
                load_group(mb, xp);
            }
            {   // filter.c:118-119 peak, on the float bit patterns
                int bp = 0;
                if (interior) {
#pragma unroll
                    for (int p = 0; p < GROUP; ++p) bp = __float_as_int(xp[p / 2][p % 2]) > bp ? __float_as_int(xp[p / 2][p % 2]) : bp;
                } else {
#pragma unroll
                    for (int p = 0; p < GROUP; ++p) {
                        const int m = mb + p;
                        const int v = (m >= 0 && m < L) ? __float_as_int(xp[p / 2][p % 2]) : 0;
                        bp = v > bp ? v : bp;
                    }
                }
                peakbits = bp > peakbits ? bp : peakbits;
            }
            float gmax_group = 0.0f;
            if constexpr (INLOOP) {
                float gm = 0.0f;
#pragma unroll
                for (int p = 0; p < GROUP; ++p) gm = __builtin_fmaxf(gm, __builtin_fabsf(xp[p / 2][p % 2]));
                // the window behind a group (NC - 1 + J0 <= 96 rows = six groups) as three maxima: this loop turn's groups
                // so far, the turn before, the turn before that -- six to eight groups, never fewer (a larger M only
                // widens the band); one v_max3 per group instead of a six-deep history that is shifted every time
                if constexpr (g == 0) {
                    hmax[2] = hmax[1];
                    hmax[1] = hmax[0];
                    hmax[0] = gm;
                } else {
                    hmax[0] = __builtin_fmaxf(hmax[0], gm);
                }
                const float M = __builtin_fmaxf(__builtin_fmaxf(hmax[0], hmax[1]), hmax[2]);
                gmax_group = gm;
                Mn = M * (1.0f / 32768.0f);
            }
            auto pairs = [&](auto S) __attribute__((always_inline)) {
                constexpr int s = decltype(S)::value;
                constexpr int P = (g * GROUP + 2 * s) % NC;
                if constexpr (INLOOP && (s == 0 || s >= 5)) eps_w = __builtin_fmaf(ek.seen[s <= 4 ? 0 : s - 4], Mn, ek.ahead[s <= 4 ? 0 : s - 4]);
                pk_step<NC, P>(neg, amb, xp[s], eps_w, tp);
            };
            pk_expand_void(pairs, std::make_integer_sequence<int, GROUP / 2>{});
            if constexpr (INLOOP) {
                zor |= __float_as_uint(gmax_group);                 // all sixteen are 0 exactly when their largest |x| is
            } else if (amb != 0) {
#pragma unroll
                for (int p = 0; p < GROUP; ++p) zor |= __float_as_uint(xp[p / 2][p % 2]);
            } else {
                zor |= 1u;
            }
            if (gi & 1) flush(gbase - 16);
            return true;
        };
        if (!pk_expand_and(group, std::make_integer_sequence<int, NG>{})) break;
    }

    int peak = (int) __int_as_float(peakbits);
    if (t1 == L) {                              // the last dc-NC+1 samples of the call
        const int shift = dc - NC + 1;
        for (int n = (L - shift > 0 ? L - shift : 0); n < L; ++n) {
            const int v = (int) x[(size_t) n * (size_t) N + c];
            peak = v > peak ? v : peak;
        }
    }
    if (live && peak > 0) atomicMax(&maxval[cg], peak);
    if (t1 == L && live) {                      // carry for the next call (filter.c:129-134 restated)
        for (int k = 0; k < NTaps; ++k) {
            const int m = L - NTaps + k;
            hist_out[(size_t) k * (size_t) N + cg] =
                (m >= 0) ? x[(size_t) m * (size_t) N + cg] : hist[(size_t) (NTaps + m) * (size_t) N + cg];
        }
        maxval_next[cg] = 0;
    }
}

// One kernel per instantiation.  NO amdgpu_num_vgpr attribute: with it the compiler treats every register above the budget
// as "reserved", IGNORES them in the asm statements' clobber lists (with a warning per statement) -- and, had the lists
// not been there at all, would allocate the wave only its own 44 / 64 registers while the streams write up to v155.
// Without it the clobber lists are honoured: the compiler knows every generated stream destroys the ring's registers, so
// no value of its own can sit there across one, and the kernel's register count covers them.  What a clobber cannot say
// is "these registers hold MY state between two asm statements": that the compiler's own code (a few dozen registers,
// allocated from v0 up) stays below the ring is checked on the ISA of every build, together with the allocated count
// (scripts/check_pk_registers.py, run by the Makefile; a violation fails the build).
#define PK_KERNEL(NAME, NCV, INL, BASE)                                                                              \
    __global__ __launch_bounds__(64) void NAME(                                                                       \
        const int16_t *__restrict__ x, const int16_t *__restrict__ hist, uint32_t *__restrict__ sgn,                 \
        int *__restrict__ maxval, int16_t *__restrict__ hist_out, int *__restrict__ maxval_next,                     \
        const float *__restrict__ te_mem, int N, int L, int T, int d, int NTaps, int NE_rt, float eps_up,            \
        PkEps ek, int map, PkTaps<NCV> tp, PkExact<0> ex)                                                            \
    {                                                                                                                 \
        fir_sign_pk_body<0, NCV, INL>(x, hist, sgn, maxval, hist_out, maxval_next, te_mem, N, L, T, d, NTaps, NE_rt,  \
                                      eps_up, ek, map, tp, ex);                                                       \
    }
PK_KERNEL(fir_sign_pk40_kernel, 40, true, PK40_VGPR_BUDGET)
PK_KERNEL(fir_sign_pk48_kernel, 48, true, PK48_VGPR_BUDGET)
#undef PK_KERNEL

} // namespace

// whole unrolled turns and whole 16-byte sign stores per segment: lcm(unrolled phases, 128)
int launch_fir_sign_pk_quantum(int NC) { return pk_unroll(NC) * 128 / pk_gcd(pk_unroll(NC), 128); }

namespace {

template <int NC, class K>
hipError_t pk_launch_long(K kern, const FirLaunch &a, dim3 grid, dim3 block, float eps_up, int map, hipStream_t stream)
{
    PkTaps<NC> tp;
    PkExact<0> ex;
    auto tc = [&](int q) { return a.ctaps[((q % NC) + NC) % NC]; };
    for (int j = 0; j < NC / 4; ++j) tp.E[j] = pk_f2{tc(2 * j), tc(2 * j + 1)};
    for (int j = 0; j <= NC / 4; ++j) tp.O[j] = pk_f2{tc(2 * j), tc(2 * j - 1)};
    ex.te[0] = 0.0f;
    PkEps ek;
    for (int q = 0; q < 4; ++q) {
        ek.seen[q] = a.eps_seen_k[q];
        ek.ahead[q] = a.eps_ahead_k[q];
        if (!(ek.seen[q] > 0.0f)) return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(kern, grid, block, 0, stream, a.x, a.hist, a.sgn, a.maxval, a.hist_out, a.maxval_next, a.te_mem, a.N, a.L,
                       a.T, a.d, a.NT, a.NE, eps_up, ek, map, tp, ex);
    return hipGetLastError();
}

} // namespace

// NC = 40 / 48: any symmetric table whose window behind a group, NC - 1 + (NE - NC) / 2 rows, is at most 96 rows long.
// (The reference's 32-tap table with 12 central taps in this form was slower than the direct form of fir_slice.hip --
// v_pk_fma_f32 issues at half the rate of v_fmac_f32 -- and left with round 5: profiles/r05_ubench_valu_op_rates.txt.)
hipError_t launch_fir_sign_pk(const FirLaunch &a, hipStream_t stream)
{
    if (a.dump || (a.NC != 40 && a.NC != 48) || a.T % launch_fir_sign_pk_quantum(a.NC) || a.NE < a.NC ||
        (a.NE - a.NC) % 2 || !a.te_mem || a.NC - 1 + (a.NE - a.NC) / 2 > 96 || a.eps_seen <= 0.0f)
        return hipErrorInvalidValue;
    dim3 grid((a.N + 63) / 64, (a.L + a.T - 1) / a.T), block(64);
    if (a.max_segments > 0 && (int) grid.y > a.max_segments) grid.y = a.max_segments;
    const float eps_up = __builtin_nextafterf(a.eps_pk > 0.0f ? a.eps_pk : a.eps, INFINITY);
    const int map = (a.map == 1 && grid.x % 8 == 0) ? 1 : 0;
    if (a.NC == 40) return pk_launch_long<40>(fir_sign_pk40_kernel, a, grid, block, eps_up, map, stream);
    return pk_launch_long<48>(fir_sign_pk48_kernel, a, grid, block, eps_up, map, stream);
}

} // namespace gnuais
