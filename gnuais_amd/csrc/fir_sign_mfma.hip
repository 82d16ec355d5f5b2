// fir_sign_mfma.hip -- K1s for LONG tables on the matrix pipe (gfx950): the sign-exact slicer standing in for
// filter_run_buf() + the `out > 0` test of receiver_run() (gnuais src/filter.c:106-143, src/receiver.c:109-111,126), 48
// central taps, as an EXACT INTEGER Toeplitz product on v_mfma_i32_32x32x32_i8 (gfx950's double-K int8 product: 32 window
// rows per instruction at the cost of the 16 rows of CDNA3's 32x32x16 form, which rounds 5 used).
//
// Same contract as fir_sign_pk.hip (bit-identical sign words, peak, history carry); what changes is how y_c, the sum over
// the 48 central taps, is formed -- and what it costs: the packed kernel issues 20-24 v_pk_fma_f32 per sample and is
// VALU-bound (3.4 ms per C5 call); here a step of 32 outputs x 64 channels is 36 matrix instructions and 370 vector
// instructions (11.5 per output row; what binds it is in the step loop's comment): 2.0 ms per C5 call.
//
//   y_c[n] = sum_q tc[q] * x[n - dc + q],  q < 48.   Taps as 24-bit integers tq = round(tc * S) (S a power of two, sum |tq| <
//   2^23), three signed int8 digits t2 t1 t0; samples as two int8 digits, x = 256 hs + l' + 128 (hs = x >> 8, l' = (x & 255)
//   - 128).  With Y = sum tq x (exact, |Y| < 2^38):
//       y' = A3 2^16 + A2 2^8 + A1 + (A0 >> 8) = floor(Y / 256),
//       A3 = sum t2 hs,  A2 = sum t2 l' + t1 hs,  A1 = sum t1 l' + t0 hs,  A0 = sum t0 l' + 128 sum tq
//   -- six matrix products per block of 32 window rows, int32 accumulators (A3 and A0 first; A3 2^8 and A0 >> 8 are what A2
//   and A1 accumulate on), no rounding anywhere: the only error against
//   the real central sum is the taps' quantisation (|tq / S - tc| <= 0.5 / S each), which the host adds to the
//   certification bound, and the floor (< 1 unit of 256 / S).
//   A step of 32 outputs reads window rows n0 - dc .. n0 - dc + 79: three blocks of 32 (the last 16 rows meet zero taps),
//   the first two are the previous step's last two.  A[i][k] = tq[32 b + k - i] (Toeplitz, constant), B[k][n] = a sample
//   digit of row k, channel n.  Which of a lane's 16 operand bytes is which k does not matter as long as A and B agree
//   (the product sums over k): byte j of half hh = lane / 32 is k = 16 hh + j in both.
//   A wave owns 64 channels as 32 PAIRS (lanes n and n + 32 hold the pair 2n, 2n+1 -- one dword per row -- for rows r0 + 16 hh
//   + j) and T outputs.
//
// The certification threshold follows a running maximum M of |x| over EVERY row a reference window of the step touches
// (behind AND ahead: the rows of the next step are loaded and converted one step early), so nothing is priced at full
// scale; a channel whose M is 0 has y = +0 exactly (bit 0, nothing to settle).  Outputs with |y'| below the threshold are
// noted per lane and settled with the reference's ordered sum (filter.h:40-49) lane-parallel, as in fir_sign_pk.hip.
//
// Only outputs whose windows lie inside the call's input run here (t0 >= first >= d, dc + 64); the call's head, with its
// history rows, is the packed kernel's (launch_fir_sign_pk with T = first, max_segments = 1), launched ahead of this one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>
#include "kernels.h"

namespace gnuais {

namespace {

typedef int mf_v16i __attribute__((ext_vector_type(16)));
typedef int mf_v4i __attribute__((ext_vector_type(4)));
extern "C" __device__ int mf_ld_b32(mf_v4i, int, int, int) __asm("llvm.amdgcn.raw.buffer.load.i32");
extern "C" __device__ float mf_load_format_f32(mf_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.f32");

constexpr int MF_NC = 48, MF_NB = 3, MF_BR = 16, MF_PEND = 8, MF_EXACT_BATCH = 8;   // MF_BR: rows of a block a lane holds

struct Blk {                 // one block of 32 window rows, this lane's 16 rows x 2 channels
    mf_v4i l[2], h[2];       // [set]: the sixteen l' / hs digits of the even / odd channel of the pair
};

// d[j] = row j: bytes (l' even, hs even, l' odd, hs odd)  ->  per digit the sixteen rows' bytes
__device__ __forceinline__ void mf_transpose16(const uint32_t *d, Blk &o)
{
    uint32_t e[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t d0 = d[4 * g], d1 = d[4 * g + 1], d2 = d[4 * g + 2], d3 = d[4 * g + 3];
        // v_perm_b32(a, b, sel): selector bytes 0-3 pick from b, 4-7 from a
        const uint32_t t0 = __builtin_amdgcn_perm(d1, d0, 0x05010400u);   // d0.b0 d1.b0 d0.b1 d1.b1
        const uint32_t t1 = __builtin_amdgcn_perm(d1, d0, 0x07030602u);   // d0.b2 d1.b2 d0.b3 d1.b3
        const uint32_t t2 = __builtin_amdgcn_perm(d3, d2, 0x05010400u);
        const uint32_t t3 = __builtin_amdgcn_perm(d3, d2, 0x07030602u);
        e[g][0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);             // b0 of rows 0..3
        e[g][1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);             // b1
        e[g][2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);             // b2
        e[g][3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);             // b3
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        o.l[0][g] = (int) e[g][0];
        o.h[0][g] = (int) e[g][1];
        o.l[1][g] = (int) e[g][2];
        o.h[1][g] = (int) e[g][3];
    }
}

__device__ __forceinline__ uint32_t mf_spread16(uint32_t p)     // nibbles n3 n2 n1 n0 -> 0 n3 0 n2 0 n1 0 n0
{
    p &= 0xffffu;
    p = (p | (p << 8)) & 0x00ff00ffu;
    return (p | (p << 4)) & 0x0f0f0f0fu;
}

typedef short mf_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mf_pk_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(mf_v2s, a), __builtin_bit_cast(mf_v2s, b)));
}
__device__ __forceinline__ uint32_t mf_pk_min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(mf_v2s, a), __builtin_bit_cast(mf_v2s, b)));
}

typedef unsigned short mf_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mf_pk_max_u(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(mf_v2u, a), __builtin_bit_cast(mf_v2u, b)));
}
// v_permlane32_swap_b32 (gfx950): the upper half of a changes places with the lower half of b -- afterwards a = (a.lo, b.lo),
// b = (a.hi, b.hi) (lower lanes | upper lanes).  A half-wave exchange in the VALU: no LDS trip, no lgkmcnt wait.
__device__ __forceinline__ void mf_swap32(uint32_t &a, uint32_t &b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

} // namespace

// One wave per (64 channels, segment of T outputs).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_sign_mfma_kernel(
    const int16_t *__restrict__ x, uint32_t *__restrict__ sgn, int *__restrict__ maxval, int16_t *__restrict__ hist_out,
    int *__restrict__ maxval_next, const float *__restrict__ te_mem, const MfmaTaps *__restrict__ cs, int N, int L, int T, int d,
    int NTaps, int NE, int first, float eps_seen_u, float eps_abs_u)
{
    const int lane = (int) threadIdx.x, n = lane & 31, hh = lane >> 5;
    const int t0 = first + (int) blockIdx.y * T;       // the call's first `first` outputs (windows into the history) are the packed kernel's
    if (t0 >= L) return;
    const int g = (int) blockIdx.x;
    const int t1 = (t0 + T < L) ? t0 + T : L;
    const int c = g * 64 + 2 * n + hh;                 // the channel whose words this lane assembles and stores
    const int J0 = (NE - MF_NC) / 2, dc = d - J0;      // y_c[o] = sum_q tc[q] * x[o - dc + q]
    const uint32_t rowb = (uint32_t) N * 2u;

    // the Toeplitz operands wait in LDS (in registers they are 30 of a wave's 256; the compiler keeps what fits)
    mf_v4i Ar[MF_NB * 3];
#pragma unroll
    for (int q = 0; q < MF_NB * 3; ++q) {
        Ar[q] = *reinterpret_cast<const mf_v4i *>(cs->a[q / 3][q % 3][lane]);
        asm volatile("" : "+v"(Ar[q]));                 // 36 registers that stay what they are (left to itself the compiler parks some in LDS and waits for them before every product)
    }
#define MF_A(b, dgt) Ar[(b) * 3 + (dgt)]
    const int K0 = cs->k0;                             // 128 * sum tq

    // the window rows through a descriptor based at the segment's first row (the whole input may exceed 4 GB); rows past
    // the call read as zero
    const int row0 = t0 - dc - 64;                     // first row of block 0 (>= 0: t0 >= first >= dc + 64, launcher)
    mf_v4i rs;
    {
        const unsigned long long p = (unsigned long long) (x + (size_t) row0 * (size_t) N);
        const unsigned long long span = (unsigned long long) (L - row0) * rowb;
        rs[0] = (int) (uint32_t) p;
        rs[1] = (int) (uint32_t) ((p >> 32) & 0xffffull);
        rs[2] = (int) (span > 0xffffffffull ? 0xffffffffu : (uint32_t) span);
        rs[3] = 0x00020000;
    }
    const int voff = (g * 64 + 2 * n) * 2 + hh * MF_BR * (int) rowb;
    auto load_block = [&](int r0, uint32_t *dd) __attribute__((always_inline)) {      // rows r0 + 16 hh + j (relative to row0)
#pragma unroll
        for (int j = 0; j < MF_BR; ++j) dd[j] = (uint32_t) mf_ld_b32(rs, voff, (int) ((uint32_t) (r0 + j) * rowb), 0);
    };
    // a block's digits and the packed maxima / minima of its 16 rows (both halves of the wave), even channel in the low half
    uint32_t peak_pk = 0;                              // packed running maximum of the samples seen (filter.c:118-119)
    // the packed largest |x| (unsigned 16 bit: 32768 fits) of a block's 32 rows, even channel in the low half; the
    // positive peak is kept per lane (its two halves meet after the loop)
    auto maxima = [&](const uint32_t *d0, uint32_t &am) __attribute__((always_inline)) {
        uint32_t mx = d0[0], mn = d0[0];
#pragma unroll
        for (int j = 1; j < MF_BR; ++j) {
            mx = mf_pk_max(mx, d0[j]);
            mn = mf_pk_min(mn, d0[j]);
        }
        peak_pk = mf_pk_max(peak_pk, mx);
        // max(mx, 0) and -min(mn, 0) as unsigned halves (-(-32768) wraps to 0x8000 = 32768: right as an unsigned value)
        const uint32_t up = mf_pk_max(mx, 0u);
        const uint32_t dn = __builtin_bit_cast(uint32_t, (mf_v2s) (__builtin_bit_cast(mf_v2s, 0u) - __builtin_bit_cast(mf_v2s, mf_pk_min(mn, 0u))));
        uint32_t a = mf_pk_max_u(up, dn), b = a;
        mf_swap32(a, b);                                // a = (own.lo, own.lo), b = (own.hi, own.hi): every lane sees both halves
        am = mf_pk_max_u(a, b);
    };
    auto digits = [&](const uint32_t *dd, Blk &o) __attribute__((always_inline)) {
        mf_transpose16(dd, o);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {                // low byte -> l' (offset binary -> two's complement), on the transposed words
            o.l[0][g4] ^= (int) 0x80808080u;
            o.l[1][g4] ^= (int) 0x80808080u;
        }
    };
    // exact re-evaluation of one output of ANY channel of the group (filter.h:40-49 order), window through a typed
    // descriptor (16-bit SSCALED) based at the segment's first reference row
    const int e_row = t0 - d;                          // >= 0 (launcher)
    mf_v4i rsrc_e;
    {
        const unsigned long long p = (unsigned long long) (x + (size_t) e_row * (size_t) N);
        const unsigned long long span = (unsigned long long) (L - e_row) * rowb;
        rsrc_e[0] = (int) (uint32_t) p;
        rsrc_e[1] = (int) (uint32_t) ((p >> 32) & 0xffffull);
        rsrc_e[2] = (int) (span > 0xffffffffull ? 0xffffffffu : (uint32_t) span);
        rsrc_e[3] = 0x13004;
    }
    auto exact_positive = [&](int o) __attribute__((always_inline)) -> bool {      // output t0 + o of channel c
        float sum = 0.0f;
        const int voe = o * (int) rowb + c * 2;
        for (int j0 = 0; j0 < NE; j0 += MF_EXACT_BATCH) {
            float xs[MF_EXACT_BATCH];
#pragma unroll
            for (int j = 0; j < MF_EXACT_BATCH; ++j) {
                const int jj = j0 + j < NE ? j0 + j : NE - 1;
                xs[j] = mf_load_format_f32(rsrc_e, voe, (int) ((uint32_t) jj * rowb), 0);
            }
#pragma unroll
            for (int j = 0; j < MF_EXACT_BATCH; ++j)
                if (j0 + j < NE) sum = sum + te_mem[j0 + j] * xs[j];
        }
        return sum > 0.0f;
    };

    // ---- prologue.  Block m = rows row0 + 32 m .. + 31.  Step k (outputs t0 + 32 k ..) multiplies blocks k + 2 .. k + 4 (its
    // central rows start at block k + 2); its reference windows reach from J0 <= 48 rows before that -- inside block k -- to
    // less than 127 rows past the first central row: inside block k + 5.  So: B[0..2] = blocks 2..4, the maxima of blocks
    // 0..4, and block 5 on its way (block k + 5 gives its maximum to step k and its digits to step k + 1).
    Blk B[MF_NB + 1];                                  // a ring: step k's blocks are B[k % 4], B[(k + 1) % 4], B[(k + 2) % 4]; the fourth slot takes block k + 5's digits
    uint32_t pam[6];                                   // packed largest |x| of blocks k .. k + 5 ([5] = the newest)
    {
        uint32_t r0[MF_BR];
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            load_block(32 * p, r0);
            maxima(r0, pam[p]);
            if (p >= 2) digits(r0, B[p - 2]);
        }
    }
    uint32_t raw[MF_BR];                               // block 5, on its way (two blocks in flight: 16 more registers, measured slower)
    load_block(160, raw);

    uint32_t wq[4] = {0u, 0u, 0u, 0u};
    __shared__ uint16_t pend[MF_PEND * 64];
    int n_pend = 0;
    auto flush = [&](int obase, uint32_t w, uint32_t amb) __attribute__((always_inline)) {     // outputs obase .. obase + 31 of channel c
        const int valid = t1 - (t0 + obase);
        if (valid < 32) {
            w &= ~0u << (32 - valid);
            amb &= ~0u << (32 - valid);
        }
        w &= ~amb;
        const int slot = (obase >> 5) & 3;
        const bool last = t0 + obase + 32 >= t1;
        while (amb && n_pend < MF_PEND) {
            const int pos = __clz((int) amb);
            amb &= ~(0x80000000u >> pos);
            pend[n_pend * 64 + lane] = (uint16_t) (obase + pos);
            ++n_pend;
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
                if (o < 0 || !exact_positive(o)) continue;
                const uint32_t bit = 0x80000000u >> (o & 31);
                const int wi = (o >> 5) - held;
                if (wi < 0) {
                    atomicOr(sgn + sgn_index((t0 + o) >> 5, N, c), bit);
                } else {
                    wq[0] |= wi == 0 ? bit : 0u;
                    wq[1] |= wi == 1 ? bit : 0u;
                    wq[2] |= wi == 2 ? bit : 0u;
                    wq[3] |= wi == 3 ? bit : 0u;
                }
            }
            n_pend = 0;
        }
        if (slot == 3 || last) {
            uint32_t *dst = sgn + sgn_index(((t0 + obase) >> 5) - slot, N, c);
            if (slot == 3) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
            } else {
                dst[0] = wq[0];
                if (slot >= 1) dst[1] = wq[1];
                if (slot >= 2) dst[2] = wq[2];
            }
        }
    };

    // ---- steps of 32 outputs.  Measured (profiles/r06_c5_matrix_pipe_k32.txt, rocprofv3 --pmc): the matrix pipe is busy 32
    // cycles per product = 0.74 ms per C5 call (42 % of the launch); the two waves of a SIMD together issue during 75 % of its
    // cycles -- the kernel is bound by its 370 vector instructions per step (96 to put the four digit levels together, 96 to
    // read sign and threshold, 80 for digits and maxima), 2.03 ms alone.  The compiler already runs the second set's products
    // beside the first set's flags (two accumulator pairs).  Round 5: one wave per SIMD software-pipelined by hand
    // (sched_group_barrier) and eight-wave workgroups alternating products / flags between s_barriers were slower;
    // round 6: three waves per SIMD (168 registers) spill, two blocks in flight are slower than one.
    // y' = ((A3 2^8 + A2) 2^8) + (A1 + (A0 >> 8)): the outer accumulators first (six products, alternating), their shifted values
    // are what the inner accumulators start from (twelve products, alternating) -- two accumulators per output instead of four,
    // three vector instructions per output to form y' instead of four; no product straight behind one on the same accumulator
    mf_v16i ksplat;
#pragma unroll
    for (int v = 0; v < 16; ++v) ksplat[v] = K0;
    asm volatile("" : "+v"(ksplat));                    // sixteen registers that stay what they are: the compiler must not rebuild them per step
    auto products = [&](auto RC, int s, mf_v16i &c2, mf_v16i &c1) __attribute__((always_inline)) {
        constexpr int R = decltype(RC)::value;
        mf_v16i a3 = {0}, a0 = ksplat;                  // (the first product reads ksplat as its C operand: no copy)
#pragma unroll
        for (int b = 0; b < MF_NB; ++b) {
            a3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(MF_A(b, 2), B[(R + b) & 3].h[s], a3, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(MF_A(b, 0), B[(R + b) & 3].l[s], a0, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            c2[v] = (int) (((uint32_t) a3[v] << 8) + 0x00800000u);     // + 2^23: y' below comes out as y + 2^31, bit 31 = (y >= 0)
            c1[v] = a0[v] >> 8;
        }
#pragma unroll
        for (int b = 0; b < MF_NB; ++b) {
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(MF_A(b, 2), B[(R + b) & 3].l[s], c2, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(MF_A(b, 1), B[(R + b) & 3].l[s], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(MF_A(b, 1), B[(R + b) & 3].h[s], c2, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(MF_A(b, 0), B[(R + b) & 3].h[s], c1, 0, 0, 0);
        }
    };
    // sixteen outputs' sign and threshold bits, spread to their places in the 32-output word, + the partner lane's half.
    // All in integers: yb = y + 2^31 (the bias rides in c2), so bit 31 IS "y >= 0", and |y| < E  <=>  (yb - (2^31 - E)) <u 2 E.
    // E = ceil(eps): for an integer |y|, |y| < eps <=> |y| < ceil(eps) -- the float form's set exactly.  Open outputs are rare
    // (1e-4), so the sixteen compares only feed ONE wave-wide "any" (v_cmp + s_or); the per-lane word of them is built when
    // there is one.  Per output: v_lshl_add, v_alignbit, v_sub, v_cmp (the float form: five, four of them VOP3).
    auto flags = [&](const mf_v16i &c2, const mf_v16i &c1, float eps, uint32_t &word, uint32_t &ambw)
        __attribute__((always_inline)) {
        const uint32_t negE = (uint32_t) -(int) __builtin_ceilf(eps);
        const uint32_t mid = 0x80000000u;
        uint32_t pos = 0, amb = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const uint32_t yb = ((uint32_t) c2[v] << 8) + (uint32_t) c1[v];                                // floor(Y / 256) + 2^31
            uint32_t t;
            asm("v_sad_u32 %0, %1, %2, %3" : "=v"(t) : "v"(yb), "s"(mid), "v"(negE));                     // |y| - E
            pos = __builtin_amdgcn_alignbit(pos, yb, 31);
            amb = __builtin_amdgcn_alignbit(amb, t, 31);
        }
        // this lane: outputs (v % 4) + 8 (v / 4) + 4 hh, v = 0 first = bit 15; the partner lane's half joins in the step
        word = mf_spread16(pos);
        ambw = mf_spread16(amb);
    };

    // One step.  R = step number mod 4 picks the ring slots at compile time (the loop below is unrolled four steps deep: no
    // block is ever moved).  Block k + 5 -- asked for a step ago -- gives its maximum and its digits right behind the first
    // set's products and its registers take the NEXT step's rows at once: those loads have the rest of the step to land.
    // The finished word of step k leaves in step k + 1, BEFORE that step's loads are issued: the memory counter is in order, so
    // a store issued behind the loads would be waited for with them at the next step's top (and the compiler, conservative
    // behind flush()'s branches, waits for everything outstanding).
    uint32_t held_w = 0, held_am = 0;
    auto step = [&](auto RC, const int orel) __attribute__((always_inline)) {     // outputs t0 + orel .. + 31
        constexpr int R = decltype(RC)::value;
        mf_v16i c2, c1;
        uint32_t word0 = 0, amb0 = 0, word1 = 0, amb1 = 0;
        int M0 = 0, M1 = 0;
        products(RC, 0, c2, c1);
        {
            maxima(raw, pam[5]);
            digits(raw, B[(R + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);          // the block's registers are free from here: nothing of it sinks below
            if (orel > 0) flush(orel - 32, held_w, held_am);
            // (no condition around the loads: rows past the end read as zero through the descriptor)
            load_block(orel + 160 + 32, raw);
            __builtin_amdgcn_sched_barrier(0);
            // the running maximum over blocks k .. k + 5: every row a reference window of this step's outputs touches
            uint32_t am = pam[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) am = mf_pk_max_u(am, pam[q]);
            M0 = (int) (am & 0xffffu);
            M1 = (int) (am >> 16);
            flags(c2, c1, __builtin_fmaf(eps_seen_u, (float) M0, eps_abs_u), word0, amb0);
        }
        products(RC, 1, c2, c1);
        flags(c2, c1, __builtin_fmaf(eps_seen_u, (float) M1, eps_abs_u), word1, amb1);
        // lane hh = 0 keeps the even channel's word (set 0), hh = 1 the odd one's (set 1); each needs its partner lane's half of
        // THAT set: one half-wave swap of (set 0, set 1) puts (own.lo, partner... ) where both find theirs -- after it word0 holds
        // the outputs 0-3 of every 8 (the lower lanes' rows) and word1 the outputs 4-7, of the set the lane keeps
        mf_swap32(word0, word1);
        mf_swap32(amb0, amb1);
        uint32_t w = (word0 << 4) | word1, am = (amb0 << 4) | amb1;
        // a channel without a nonzero sample in reach has y = +0 exactly: not positive (receiver.c:111), nothing to settle
        if ((hh == 0 ? M0 : M1) == 0) {
            w = 0;
            am = 0;
        }
        held_w = w;
        held_am = am;
#pragma unroll
        for (int q = 0; q < 5; ++q) pam[q] = pam[q + 1];
    };
    for (int n0 = t0; n0 < t1; n0 += 128) {
        step(std::integral_constant<int, 0>{}, n0 - t0);
        if (n0 + 32 < t1) step(std::integral_constant<int, 1>{}, n0 + 32 - t0);
        if (n0 + 64 < t1) step(std::integral_constant<int, 2>{}, n0 + 64 - t0);
        if (n0 + 96 < t1) step(std::integral_constant<int, 3>{}, n0 + 96 - t0);
    }
    flush(((t1 - t0 + 31) / 32 - 1) * 32, held_w, held_am);

    // ---- peak, carry
    peak_pk = mf_pk_max(peak_pk, (uint32_t) __shfl_xor((int) peak_pk, 32));        // the two halves' rows
    int peak = hh ? ((int) peak_pk >> 16) : (int) (short) (peak_pk & 0xffffu);
    if (t1 == L) {                                     // the call's last rows, which no window of this segment has read
        // (the loop's maxima reach row L - dc + 111 at least; dc = d - J0 grows with trailing zero taps, so the rescan
        // follows it -- 96 for the 192 kHz table, where 64 rows do)
        const int J0p = (NE - MF_NC) / 2, back = (d - J0p) > 64 ? (d - J0p) : 64;
        for (int m = (L - back > t0 ? L - back : t0); m < L; ++m) {
            const int v = (int) x[(size_t) m * (size_t) N + c];
            peak = v > peak ? v : peak;
        }
    }
    if (peak > 0) atomicMax(&maxval[c], peak);
    if (t1 == L) {                                     // carry for the next call (filter.c:129-134 restated); L >= NTaps (launcher)
        for (int k = 0; k < NTaps; ++k)
            hist_out[(size_t) k * (size_t) N + c] = x[(size_t) (L - NTaps + k) * (size_t) N + c];
        maxval_next[c] = 0;
    }
}

// 24-bit integer taps in three signed int8 digits, laid out as the A operands of the five blocks; false when the table
// does not fit (a tap too large for the scale).  bound_q: sum |tq / S - tc| (what the quantisation adds to the
// certification bound, per unit of |x|); scale: S.
bool fir_sign_mfma_taps(const float *tc48, MfmaTaps *out, double *scale, double *bound_q)
{
    double sabs = 0;
    for (int q = 0; q < MF_NC; ++q) sabs += fabs((double) tc48[q]);
    if (!(sabs > 0) || !isfinite(sabs)) return false;
    int e = 0;
    (void) frexp(8388608.0 / sabs * 0.999, &e);        // the largest power of two at or below 2^23 / sum |tc| (a power of two:
    const double S = ldexp(1.0, e - 1);                //  tq / S is then exact in double and fp32 alike)
    long tq[MF_NC], sumtq = 0, sumabs = 0;
    double bq = 0;
    for (int q = 0; q < MF_NC; ++q) {
        tq[q] = lround((double) tc48[q] * S);
        sumtq += tq[q];
        sumabs += labs(tq[q]);
        bq += fabs((double) tq[q] / S - (double) tc48[q]);
    }
    if (sumabs >= 8388608 - 64) return false;
    for (int b = 0; b < MF_NB; ++b)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31;
            uint32_t v[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            for (int j = 0; j < MF_BR; ++j) {
                const int k = MF_BR * (lane >> 5) + j, q = 32 * b + k - i;
                long t = (q >= 0 && q < MF_NC) ? tq[q] : 0;
                // signed digits: t = 65536 t2 + 256 t1 + t0, each in [-128, 127]
                const long t0 = ((t + 128) & 255) - 128; t = (t - t0) >> 8;
                const long t1 = ((t + 128) & 255) - 128; t = (t - t1) >> 8;
                const long t2 = t;
                if (t2 < -128 || t2 > 127) return false;
                v[0][j >> 2] |= (uint32_t) (uint8_t) (int8_t) t0 << (8 * (j & 3));
                v[1][j >> 2] |= (uint32_t) (uint8_t) (int8_t) t1 << (8 * (j & 3));
                v[2][j >> 2] |= (uint32_t) (uint8_t) (int8_t) t2 << (8 * (j & 3));
            }
            for (int dgt = 0; dgt < 3; ++dgt)
                for (int w = 0; w < 4; ++w) out->a[b][dgt][lane][w] = (int) v[dgt][w];
        }
    out->k0 = (int) (128 * sumtq);
    *scale = S;
    *bound_q = bq;
    return true;
}

int launch_fir_sign_mfma_quantum() { return 128; }

// outputs first .. L - 1 in segments of T (the packed kernel takes the call's first `first` outputs, whose windows reach
// into the history): a.mfma = the device copy of the taps, a.eps_seen / a.eps_ahead = the threshold in units of y' per unit of
// |x| / absolute
hipError_t launch_fir_sign_mfma(const FirLaunch &a, int first, hipStream_t stream)
{
    const int J0 = (a.NE - MF_NC) / 2, dc = a.d - J0;
    if (a.dump || a.NC != MF_NC || a.T % 128 || a.NE < MF_NC || (a.NE - MF_NC) % 2 || !a.te_mem || !a.mfma || a.N % 64 || first % 128 ||
        first < dc + 64 || first < a.d || a.L < a.NT || a.T > 65280 || !(a.eps_seen > 0.0f) || dc < 0 || J0 > 48 ||
        (unsigned long long) (a.T + a.NE + 512) * (unsigned long long) a.N * 2ull >= 0x7fffffffull)
        return hipErrorInvalidValue;
    const int segs = (a.L - first + a.T - 1) / a.T;
    if (segs <= 0) return hipSuccess;
    dim3 grid(a.N / 64, segs), block(64);
    hipLaunchKernelGGL(fir_sign_mfma_kernel, grid, block, 0, stream, a.x, a.sgn, a.maxval, a.hist_out, a.maxval_next, a.te_mem,
                       a.mfma, a.N, a.L, a.T, a.d, a.NT, a.NE, first, a.eps_seen, a.eps_ahead);
    return hipGetLastError();
}

} // namespace gnuais
