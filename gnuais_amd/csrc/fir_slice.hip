// fir_slice.hip -- K1: FIR low-pass + sign slicer for gfx950 (CDNA4).
//
// Stands in for filter_run_buf() (gnuais src/filter.c:106-143, MAC loop
// src/filter.h:40-49) and the `out > 0` slicer of receiver_run()
// (src/receiver.c:109-111,126) for a whole batch of channels.
//
// Mapping: one lane = one channel, one wavefront = 64 adjacent channels x one
// time segment.  The input is the reference's interleaved layout [L][N], so a
// wave's load of one sample time is 64 consecutive int16 = one 128-byte line:
// HBM traffic is coalesced along the channel axis and every sample is fetched
// once (plus a (NE-1)-sample warm-up per segment).
//
// Arithmetic: transposed-form FIR.  Sample x[m] contributes te[j]*x[m] to the
// output n = m + d - j; the NE partial sums in flight live in NE VGPRs per
// lane (a rotating register file, made static by unrolling NE phases).  For
// one output the products are added in ascending tap order with one rounding
// per multiply and one per add -- bit-identical to the reference's scalar
// mulss/addss sequence (no FMA contraction: build with -ffp-contract=off;
// fp32 subnormals stay enabled, taps 2 and 33 are subnormal).  Taps that are
// exactly 0.0f at either end of the table are trimmed on the host (x*0 = +-0
// and s + +-0 = s for every s reachable from +0): NE = 32 of the 36 reference
// taps, d = 34.
//
// Output: 1 bit per sample, the sign words (bit 31 = oldest of the 32; layout: sgn_index(), kernels.h:
// word w of channel c next to words w^1, w^2, w^3 of the same channel;
// samples of word w), optionally the fp32 filter output (parity dump), and the
// per-channel peak positive sample (filter_run_buf's return value).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <math.h>
#include "kernels.h"

namespace gnuais {

template <int NE>
struct FirTaps { float te[NE]; };

// A typed buffer load: descriptor (base, span, DST_SEL_X = R | NUM_FORMAT = SSCALED | DATA_FORMAT = 16) in SGPRs, the
// lane's byte offset in a VGPR, the row in the scalar offset -- the memory pipeline delivers the int16 sample as a float
// (exactly), with no 64-bit address arithmetic and no conversion in the VALU stream.
// clang has no builtin for llvm.amdgcn.raw.buffer.load.format; the intrinsic is reached by name
typedef int fir_v4i __attribute__((ext_vector_type(4)));
extern "C" __device__ float fir_load_format_f32(fir_v4i rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.format.f32");

__device__ __forceinline__ int load_sample(const int16_t *__restrict__ x,
                                           const int16_t *__restrict__ hist,
                                           int m, int N, int NT, int c)
{
    // m < 0: sample of an earlier run call -> history rows (NT rows, oldest first)
    const int16_t *p = (m >= 0) ? (x + (size_t) m * (size_t) N + c)
                                : (hist + (size_t) (NT + m) * (size_t) N + c);
    return (int) *p;
}

// the exact kernel (built with -fno-slp-vectorize: v_mul_f32 / v_add_f32, every product and every sum rounded as
// filter.h:40-49 rounds them)
namespace scalar {

// Zero-instruction ordering fence.  hipcc's DAG scheduler is free to hoist the
// multiplies (or sink the additions) of many samples of the unrolled block and
// then runs out of registers; a volatile asm that "uses and redefines" the
// accumulators pins each sample's work where the source puts it.
__device__ __forceinline__ void touch16(float *a)
{
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),
                      "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]),
                      "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}


// grid: x = channel group (64 channels), y = time segment of T outputs
// (T a multiple of 32).  block = 64 threads = one wave.
// SYM: the tap table is bitwise symmetric (te[j] == te[NE-1-j], true for the
// reference's Gaussian).  The rounded product te[j]*x is then the same float as
// te[NE-1-j]*x, so one multiply feeds two accumulators: NE/2 multiplies + NE
// additions per sample instead of NE + NE, with no change to any sum.
template <int NE, bool DUMP, bool SYM>
__global__ __launch_bounds__(64) void fir_slice_kernel(
    const int16_t *__restrict__ x, const int16_t *__restrict__ hist,
    uint32_t *__restrict__ sgn, float *__restrict__ dump, int *__restrict__ maxval,
    int16_t *__restrict__ hist_out, int *__restrict__ maxval_next,
    int N, int L, int T, int d, int NT, FirTaps<NE> taps)
{
    static_assert(NE == 32, "rotating accumulator file is sized for 32 effective taps");
    const int lane = threadIdx.x;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;          // clamp: tail lanes recompute channel N-1
    const bool live = cg < N;
    const int t0 = blockIdx.y * T;
    const int t1 = (t0 + T < L) ? t0 + T : L;
    if (t0 >= L) return;

    float acc[NE];
#pragma unroll
    for (int s = 0; s < NE; ++s) acc[s] = 0.0f;

    int peak = 0;
    int peakbits = 0;                           // the blocks' peak as float bits (see there)
    // local sample index i <-> m = t0 - d + i ; sample i feeds output o = i - j
    // (o = n - t0); output o is complete after sample i = o + NE - 1.
    const int m0 = t0 - d;

    // warm-up: samples i = 0 .. NE-2 (no output completes).  They are all
    // history rows for the first segment (m <= -(d-NE+2) < 0) and all input rows
    // for the others, so one uniform base pointer serves the whole loop.
    {
        const int16_t *wbase = (t0 == 0) ? hist + (size_t) (NT - d) * (size_t) N + c
                                         : x + (size_t) m0 * (size_t) N + c;
        int xw[NE - 1];
#pragma unroll
        for (int i = 0; i < NE - 1; ++i) xw[i] = (int) wbase[(size_t) i * (size_t) N];
#pragma unroll
        for (int i = 0; i < NE - 1; ++i) {
            const float xs = (float) xw[i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {      // o = i - j >= 0 only
                const int s = (i - j) & (NE - 1);
                // SYM: taps j and NE-1-j share one product (computed at the lower index)
                const int jm = (SYM && j >= NE / 2) ? NE - 1 - j : j;
                const float p = taps.te[jm] * xs;
                if (j == 0) acc[s] = p + 0.0f; else acc[s] = acc[s] + p;
            }
        }
    }

    // interior blocks read through a typed descriptor based at the segment's first interior row (as K1s does): every row
    // this segment reads lies within [row0, row0 + T + NE + 32) rows of it
    const int row0 = m0 + NE - 1 > 0 ? m0 + NE - 1 : 0;
    const uint32_t rowbytes = (uint32_t) N * 2u;
    const unsigned long long span = (unsigned long long) (L - row0) * rowbytes;
    const unsigned long long xbase = (unsigned long long) (x + (size_t) row0 * (size_t) N);
    const fir_v4i rsrc_f = {(int) (xbase & 0xffffffffull), (int) ((xbase >> 32) & 0xffffull),
                            (int) (span > 0xffffffffull ? 0xffffffffull : span), 0x13004};
    const bool typed_ok = (unsigned long long) (T + NE + 64) * rowbytes < 0x7fffffffull;   // the scalar offset is 32 bits
    const int coff = c * 2;

    const int nblk = (t1 - t0 + 31) >> 5;
    for (int b = 0; b < nblk; ++b) {
        const int obase = b * 32;               // outputs obase .. obase+31
        float xf[32];                           // the block's samples as floats (int16 values: exact)
        // the 32 samples of this block: i = NE-1+obase+p  ->  m = mb + p
        const int mb = m0 + NE - 1 + obase;
        const bool interior = (mb >= 0) && (mb + 31 < L) && typed_ok;
        if (interior) {
#pragma unroll
            for (int p = 0; p < 32; ++p)
                xf[p] = fir_load_format_f32(rsrc_f, coff, (int) ((uint32_t) (mb - row0 + p) * rowbytes), 0);
        } else {
            // first block of a call (3 history samples) or the last, partial block
            // (addresses clamped; those outputs are masked out below)
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                int m = mb + p;
                m = (m < L) ? m : L - 1;
                xf[p] = (float) load_sample(x, hist, m, N, NT, c);
            }
        }
        // filter.c:118-119: peak positive input sample of this call.  The block's
        // samples are n = mb .. mb+31; over all segments they cover [-shift, L-shift)
        // (shift = d-NE+1 trailing zero taps), so history samples (n < 0) are
        // masked here and the last `shift` samples are added after the loop.
        {   // (for values >= 0 the order of floats is the order of their bit patterns as signed integers, and negative
            // floats are negative integers: an integer max against 0 is the float max; converted back once, below)
            int bp = 0;
            if (interior) {
#pragma unroll
                for (int p = 0; p < 32; ++p) bp = __float_as_int(xf[p]) > bp ? __float_as_int(xf[p]) : bp;
            } else {
#pragma unroll
                for (int p = 0; p < 32; ++p) {
                    const int m = mb + p;
                    const int v = (m >= 0 && m < L) ? __float_as_int(xf[p]) : 0;
                    bp = v > bp ? v : bp;
                }
            }
            peakbits = bp > peakbits ? bp : peakbits;
        }
        uint32_t w = 0;
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const float xs = xf[p];
            if (SYM) {
#pragma unroll
                for (int j = 0; j < NE / 2; ++j) {
                    const float pr = taps.te[j] * xs;       // == te[NE-1-j] * xs bit for bit
                    const int s0 = (NE - 1 + p - j) & (NE - 1);
                    const int s1 = (p + j) & (NE - 1);      // slot of tap NE-1-j
                    if (j == 0) acc[s0] = pr + 0.0f; else acc[s0] = acc[s0] + pr;
                    acc[s1] = acc[s1] + pr;
                }
                touch16(acc);
                touch16(acc + 16);
            } else {
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int s = (NE - 1 + p - j) & (NE - 1);
                    const float pr = taps.te[j] * xs;
                    if (j == 0) acc[s] = pr + 0.0f; else acc[s] = acc[s] + pr;
                }
            }
            const float y = acc[p & (NE - 1)];  // output o = obase + p is complete
            w = (w << 1) | (y > 0.0f ? 1u : 0u);
            if (DUMP) {
                const int n = t0 + obase + p;
                if (live && n < t1) dump[(size_t) n * (size_t) N + cg] = y;
            }
        }
        // partial last word: outputs beyond t1 are garbage -> clear them; valid
        // bits stay left-aligned (bit 31 = oldest)
        const int valid = t1 - (t0 + obase);
        if (valid < 32) w &= ~0u << (32 - valid);
        if (live) sgn[sgn_index((t0 + obase) >> 5, N, cg)] = w;
    }

    peak = (int) __int_as_float(peakbits);      // (an int16 value as a float: exact)
    // the last segment also owns the final `shift` samples of the call
    if (t1 == L) {
        const int shift = d - NE + 1;
        for (int n = (L - shift > 0 ? L - shift : 0); n < L; ++n) {
            const int v = (int) x[(size_t) n * (size_t) N + c];
            peak = v > peak ? v : peak;
        }
    }
    if (live && peak > 0) atomicMax(&maxval[cg], peak);

    // The wave that owns the end of the call also leaves the carry for the next
    // one (filter.c:129-134 restated: the last NT input samples, oldest first, into
    // the other history buffer) and clears the peak buffer the next call will use,
    // so that one call is exactly one kernel on the caller's stream.
    if (t1 == L && live) {
        for (int k = 0; k < NT; ++k) {
            const int m = L - NT + k;
            hist_out[(size_t) k * (size_t) N + cg] =
                (m >= 0) ? x[(size_t) m * (size_t) N + cg] : hist[(size_t) (NT + m) * (size_t) N + cg];
        }
        maxval_next[cg] = 0;
    }
}

} // namespace scalar

// ---------------------------------------------------------------------------
// K1s -- sign-exact slicer (the default on the receive path).
//
// receiver_run() only ever looks at `out > 0` (receiver.c:111,126).  The sign of
// the reference's 32-term ordered fp32 sum y_ref can be decided from a much
// cheaper quantity whenever that quantity is far enough from zero:
//   y_c = the NC = 12 central taps only (te[10..21]; the 20 outer taps sum to
//         5.4e-8), evaluated in fp32 in transposed form, 6 shared products + 12
//         additions per sample;
//   |y_ref - S|   <= X * sum_i |te_i| * ((1+u)^k_i - 1)   k_i = roundings product i passes
//   |y_c   - S_c| <= X * sum_i |tc_i| * ((1+u)^k_i - 1)   through: n for i = 1, n-i+2 after
//   |S - S_c|     <= X * sum|te outside the centre|
// with u = 2^-24, X = 32768 (int16 input), S / S_c the exact real sums; subnormal
// products add at most 32 * 2^-150.  The host adds the three terms in double
// precision from the actual table (gnuais_capi.hip; 0.124 for the reference table) and
// passes eps = that * 1.1.  If |y_c| > eps then y_ref has the sign of y_c and is not
// zero; otherwise (about 1e-4 of the samples of a noisy channel, all of them in
// a silent one) the sample is re-evaluated with the exact ordered 32-tap sum from
// the input (L1/L2 hits).  The emitted sign words are therefore bit-identical to
// K1's; the fp32 filter output itself is only available from the exact kernel
// (gnuais_batch_filter).
//
// Same mapping as K1 (lane = channel, wave = 64 channels x one time segment), but
// 12 accumulators rotate, so 96 phases (three sign words) are unrolled.
constexpr int FIR_SIGN_PEND = 8;    // noted outputs per lane (1 KB of LDS per wave); more than that are settled on the spot
constexpr int FIR_SIGN_FENCE = 4;   // transposed form: the scheduling fence (touch12) after every fourth sample
// words per loop turn of the direct form: a lane stores its four sign words of 128 outputs as ONE 16-byte store
// (sgn_index() keeps them adjacent): a wave's store covers 1 KB densely instead of four stores of 4 bytes in every 16
// (PMC: 0.30 GB of write traffic per C3 call for 0.10 GB of sign words)
constexpr int FIR_DIRECT_UNROLL = 4;
// Main-loop loads are typed buffer loads (16-bit SSCALED descriptor): no VALU address adds, and the memory pipeline also
// does the int16 -> float conversion, exactly -- scripts/ubench/fmt_load.hip checks all 65536 values.  (Plain global loads,
// untyped buffer loads, the taps in vector registers, a software prefetch of the 48-tap groups: measured, git history.)
// zero-instruction fence (see touch16): bounds how many samples the scheduler interleaves,
// i.e. how many products are alive at once; without it the kernel needs 98 VGPRs (4 waves per
// SIMD) instead of <= 88 (5 waves)
__device__ __forceinline__ void touch12(float *a)
{
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),
                      "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]));
}

// NES > 0: the table has exactly NES effective taps and travels whole in SGPRs (`taps`, NT = NES);
// NES == 0: only the NC central taps do, the exact re-evaluation reads the NE taps from memory.
// INLOOP (48-tap instantiation): eps follows a running maximum of |x| kept in the loop instead of a pre-pass over the segment
// FL2 (direct form): the central taps arrive scaled by `fscale`, a power of two (exact), chosen by the host so that the
// certified distance becomes |y'| >= 2.0 -- bit 30 of the float, the top bit of its exponent.  ONE v_alignbit_b32 by 30
// then collects the sign AND that bit of every output (even and odd outputs in two words, pulled apart by four bit
// operations per 32 outputs) where the plain form needs |y| - eps and two alignbits: one instruction per output for the
// flags instead of three.  The band is the next power of two above eps (0.5 for 0.38 with the reference table).
template <int NES, int NC, int NT, bool INLOOP = false, bool FL2 = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NC == 48 ? 4 : 1))) void fir_sign_kernel(
    const int16_t *__restrict__ x, const int16_t *__restrict__ hist,
    uint32_t *__restrict__ sgn, int *__restrict__ maxval,
    int16_t *__restrict__ hist_out, int *__restrict__ maxval_next,
    const float *__restrict__ te_mem, int N, int L, int T, int d, int NTaps, int NE_rt, float eps_up,
    int map, FirTaps<NT> taps, float eps_seen, float eps_ahead, float fscale)
{
    static_assert(!FL2 || K1S_DIRECT(NC), "the two-bit flag gather belongs to the direct form");
    const int NE = NES > 0 ? NES : NE_rt;
    // the grid is channel groups x segments, a workgroup does one item.  (A grid of fewer, persistent workgroups looping
    // over the items was measured slower: the hardware's own placement of 24 000 short-lived workgroups balances the
    // launch better, and waves that never leave give the other stages no turn.)
    const int gx = (int) gridDim.x;
    const int item = (int) (blockIdx.y * gridDim.x + blockIdx.x);
    const int J0 = (NE - NC) / 2;               // first central tap (10 of 32 for the reference table)
    auto ctap = [&](int q) -> float { return NES > 0 ? taps.te[(NES - NC) / 2 + q] : taps.te[q]; };
    if constexpr (NC <= 12)
    // Claim 104 VGPRs although the fenced code needs 68: four waves per SIMD then leave 96
    // registers for waves of the stages that run beside us (the PLL stage needs 64).  At seven
    // waves per SIMD this kernel would fill the register file and they would wait for FIR
    // waves to retire before they could even be placed.  88 (five waves, 72 left: rounds 1-3) gives the same 20-call
    // figure and a steady state 5-7 % slower; 96, 112 and 128 are worse than both (profiles/r04_fir_claim.txt).
    asm volatile("" ::: "v103");
    static_assert((K1S_DIRECT(NC) || 96 % NC == 0) && NC % 2 == 0, "96 unrolled phases must hold whole turns of the accumulator ring");
    const int lane = threadIdx.x;
    // Workgroup -> (channel group, time segment).  The dispatcher deals consecutive workgroup ids
    // round-robin over the 8 XCDs.  map 0: id = segment * groups + group, so an XCD works on every
    // 8th channel group (its 128-byte pieces of a sample row are 1 KB apart); map 1: every XCD takes
    // a contiguous eighth of the channel groups (4 KB of each row for 16384 channels).
    int bx = item % gx, by = item / gx;
    if (map == 1) {
        const int G = gx, id = item, per = G >> 3;
        bx = (id & 7) * per + (id >> 3) % per;
        by = (id >> 3) / per;
    }
    const int cg = bx * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    // (shorter segments at the launch's end -- its tail is one wave lifetime, 70 us of 410, during which the occupancy
    // falls linearly to zero -- were measured level: profiles/r03_fir_wave_timeline_short_tail.txt)
    const int t0 = by * T;
    const int t1 = t0 + T < L ? t0 + T : L;
    if (t0 >= L) return;
    const int dc = d - J0;                      // y_c[n] = sum_q tc[q] * x[n - dc + q]

    float acc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) acc[q] = 0.0f;
    int peak = 0;
    int peakbits = 0;                           // 12-tap path: the peak as float bits (see there)
    const int m0 = t0 - dc;                     // local sample i <-> m = m0 + i, feeds output o = i - q

    // exact value of output n: filter.h:40-49 order, samples re-read from memory 32 at a time
    // (all loads of a chunk in flight together), taps from memory (uniform: scalar loads)
    // A segment whose every reference window lies inside this call's input (all but the call's first and last ones:
    // rows t0 - d .. t1 - 1 - d + NE - 1 within [0, L)) reads the window through a typed buffer descriptor based at its
    // first row: the lane's byte offset is ONE multiply, the tap's row goes into the scalar offset, and the memory
    // pipeline delivers the sample as a float -- two VALU instructions per tap (the product and the sum, rounded
    // separately as filter.h:40-49 rounds them) instead of twenty of address arithmetic, selects and conversion
    // around every load (650 per evaluation of the 32-tap table, a tenth of this kernel's instructions at C3).
    const uint32_t e_rowbytes = (uint32_t) N * 2u;
    const int e_row = t0 - d;
    const bool e_inner = e_row >= 0 && t1 - 1 - d + NE - 1 <= L - 1 &&
                         (unsigned long long) (t1 - t0 + NE) * e_rowbytes < 0x7fffffffull;
    const unsigned long long e_span = e_inner ? (unsigned long long) (L - e_row) * e_rowbytes : 0ull;
    const unsigned long long e_base = (unsigned long long) (x + (size_t) (e_inner ? e_row : 0) * (size_t) N);
    const fir_v4i rsrc_e = {(int) (e_base & 0xffffffffull), (int) ((e_base >> 32) & 0xffffull),
                            (int) (e_span > 0xffffffffull ? 0xffffffffull : e_span), 0x13004};   // R | SSCALED | 16
    auto exact_positive = [&](int n) -> bool {
        float sum = 0.0f;
        if (e_inner) {
            const int voff = (n - t0) * (int) e_rowbytes + c * 2;
            if constexpr (NES > 0) {
                float xs[NES];
#pragma unroll
                for (int j = 0; j < NES; ++j) xs[j] = fir_load_format_f32(rsrc_e, voff, (int) ((uint32_t) j * e_rowbytes), 0);
#pragma unroll
                for (int j = 0; j < NES; ++j) sum = sum + taps.te[j] * xs[j];
                return sum > 0.0f;
            }
            for (int j0 = 0; j0 < NE; j0 += 32) {
                float xs[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const int jj = j0 + j < NE ? j0 + j : NE - 1;
                    xs[j] = fir_load_format_f32(rsrc_e, voff, (int) ((uint32_t) jj * e_rowbytes), 0);
                }
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (j0 + j < NE) sum = sum + te_mem[j0 + j] * xs[j];
            }
            return sum > 0.0f;
        }
        if constexpr (NES > 0) {
#pragma unroll
            for (int j = 0; j < NES; ++j) {
                const float xs = (float) load_sample(x, hist, n - d + j, N, NTaps, c);
                sum = sum + taps.te[j] * xs;
            }
            return sum > 0.0f;
        }
        for (int j0 = 0; j0 < NE; j0 += 32) {
            int xs[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int jj = j0 + j < NE ? j0 + j : NE - 1;
                xs[j] = load_sample(x, hist, n - d + jj, N, NTaps, c);
            }
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (j0 + j < NE) sum = sum + te_mem[j0 + j] * (float) xs[j];
        }
        return sum > 0.0f;
    };

    // warm-up: samples i = 0 .. NC-2
    float tail[NC - 1];                         // direct form: the NC-1 samples before the current word
    {
        int xw[NC - 1];
#pragma unroll
        for (int i = 0; i < NC - 1; ++i) xw[i] = load_sample(x, hist, m0 + i, N, NTaps, c);
#pragma unroll
        for (int i = 0; i < NC - 1; ++i) tail[i] = (float) xw[i];
#pragma unroll
        for (int i = 0; i < NC - 1 && !K1S_DIRECT(NC); ++i) {
            const float xs = (float) xw[i];
#pragma unroll
            for (int q = 0; q <= i; ++q) {
                const int qm = q >= NC / 2 ? NC - 1 - q : q;
                acc[(i - q) % NC] = (q == 0 ? 0.0f : acc[(i - q) % NC]) + ctap(qm) * xs;
            }
        }
    }

    // Digital silence.  With every sample of a window 0 the reference's sum is exactly +0 and
    // `out > 0` is false, but y_c = 0 is "ambiguous", and 32 exact evaluations per word and
    // lane, one after the other, would make a silent channel ~50x slower than a live one.
    // When a word has many ambiguous samples the lane therefore checks whether all samples
    // its 32 windows can touch are 0: this word's 32 samples, the J0 + NC - 1 before them and the
    // J0 after them (21 and 10 for the reference table).  all_zero() re-reads from memory; it
    // only runs in that case.
    auto all_zero = [&](int m_first, int count) -> bool {
        uint32_t o = 0;
        for (int i = 0; i < count; ++i) {
            int m = m_first + i;
            m = m < -NTaps ? -NTaps : (m > L - 1 ? L - 1 : m);   // clamped samples are outside every window
            o |= (uint32_t) load_sample(x, hist, m, N, NTaps, c);
        }
        return o == 0;
    };
    bool zprev_known = false, zprev = false;    // was the previous word's block of 32 samples all 0?

    // Buffer descriptor for the interior loads of the main loop: every interior row this segment
    // reads lies in [row0, row0 + T + NC + 32), far less than the 4 GB a descriptor spans.
    const int row0 = m0 > 0 ? m0 : 0;
    const uint32_t rowbytes = (uint32_t) N * 2u;
    const unsigned long long span = (unsigned long long) (L - row0) * rowbytes;
    const int coff = c * 2;
    // a typed descriptor: DST_SEL_X = R, NUM_FORMAT = SSCALED, DATA_FORMAT = 16
    const unsigned long long xbase = (unsigned long long) (x + (size_t) row0 * (size_t) N);
    const fir_v4i rsrc_f = {(int) (xbase & 0xffffffffull), (int) ((xbase >> 32) & 0xffffull),
                            (int) (span > 0xffffffffull ? 0xffffffffull : span), 0x13004};

  if constexpr (NC % 32 != 16) {
        // words unrolled per loop turn: three for the transposed form (whole turns of its accumulator
        // ring); the direct form has no ring and takes FIR_DIRECT_UNROLL (the tail registers are renamed
        // inside the unrolled body, moved only at the back edge)
        constexpr int UW = K1S_DIRECT(NC) ? FIR_DIRECT_UNROLL : 3;
        const int nblk = (t1 - t0 + UW * 32 - 1) / (UW * 32);
        constexpr bool WIDE = K1S_DIRECT(NC) && (UW == 4 || UW == 2 || UW == 1);   // T % 128 == 0: t0's word index is 0 mod 4
        uint32_t wq[4] = {0u, 0u, 0u, 0u};                      // the sign words of 128 outputs, stored as one
        __shared__ uint16_t pend[FIR_SIGN_PEND * 64];           // per lane: outputs (relative to t0; the launcher caps T
        int n_pend = 0;                                         // below 65536) whose sign the central sum left open
        for (int b = 0; b < nblk; ++b) {
    #pragma unroll
            for (int w3 = 0; w3 < UW; ++w3) {
                const int obase = (b * UW + w3) * 32;           // outputs obase .. obase+31
                if (t0 + obase >= t1) break;
                float xf[32];                                   // the word's samples, as floats (exact)
                const int mb = m0 + NC - 1 + obase;             // sample of phase 0
                const bool interior = (mb >= 0) && (mb + 31 < L);
                if (interior) {
                    // buffer loads: descriptor (wave-uniform base = this segment's first row) in SGPRs, the
                    // row in the scalar offset, the lane's constant byte offset in the vector offset -- no
                    // per-load 64-bit VALU address add (a sixth of this kernel's issue cycles otherwise);
                    // typed, the load also delivers the sample as a float
    #pragma unroll
                    for (int p = 0; p < 32; ++p)
                        xf[p] = fir_load_format_f32(rsrc_f, coff, (int) ((uint32_t) (mb - row0 + p) * rowbytes), 0);
                } else {
    #pragma unroll
                    for (int p = 0; p < 32; ++p) {
                        int m = mb + p;
                        m = (m < L) ? m : L - 1;
                        xf[p] = (float) load_sample(x, hist, m, N, NTaps, c);
                    }
                }
                {   // filter.c:118-119 peak (same bookkeeping as K1, shift = dc - NC + 1).  For values >= 0
                    // the order of floats is the order of their bit patterns as signed integers, and
                    // negative floats are negative integers: an integer max against 0 is the float max
                    int bp = 0;
                    if (interior) {
    #pragma unroll
                        for (int p = 0; p < 32; ++p) bp = __float_as_int(xf[p]) > bp ? __float_as_int(xf[p]) : bp;
                    } else {
    #pragma unroll
                        for (int p = 0; p < 32; ++p) {
                            const int m = mb + p;
                            const int v = (m >= 0 && m < L) ? __float_as_int(xf[p]) : 0;
                            bp = v > bp ? v : bp;
                        }
                    }
                    peakbits = bp > peakbits ? bp : peakbits;
                }
                // Two flag bits per sample, each gathered with ONE v_alignbit_b32 (shift the word
                // left, take the new bit from another register's bit 31) instead of compare +
                // select + shift/or, which was a third of this kernel's issue time:
                //   neg  collects the sign bit of y_c: 0 for y_c > 0 and for +0.0 (ambiguous anyway);
                //   amb  collects the sign bit of |y_c| - eps_up, eps_up = nextafter(eps): set
                //        exactly when |y_c| <= eps (a - b is negative or -0 iff a < b).
                uint32_t neg = 0, amb = 0;
                uint32_t fe = 0, fo = 0;                        // FL2: (sign, exponent bit 7) of the even / the odd outputs
                float dtap[NC / 2];
    #pragma unroll
                for (int q = 0; q < NC / 2; ++q) dtap[q] = FL2 ? ctap(q) * fscale : ctap(q);
    #pragma unroll
                for (int p = 0; p < 32; ++p) {
                    const int P = w3 * 32 + p;                  // phase 0..95, P % 12 static
                    float y;
                    if constexpr (K1S_DIRECT(NC)) {
                        // direct form on the word's own registers: the window of output p is
                        // xf[p-NC+1 .. p] (older than the word: tail[]); symmetric taps share a product,
                        // and the sum of two int16-valued floats is exact: NC/2 adds, NC/2 muls,
                        // NC/2-1 adds -- one op less than the transposed form, no accumulator ring.
                        // Edge taps first, so the big central terms see the fewest roundings.
                        auto xb = [&](int k) -> float {     // the sample k steps before the newest
                            return p - k >= 0 ? xf[p - k >= 0 ? p - k : 0] : tail[p - k >= 0 ? 0 : NC - 1 + p - k];
                        };
                        // y_c only has to stay within the certified distance of the exact central sum,
                        // so its products are FUSED into the accumulation (one rounding per term instead
                        // of two: inside the host's bound, which is taken for mul + add)
                        y = dtap[0] * (xb(0) + xb(NC - 1));
    #pragma unroll
                        for (int q = 1; q < NC / 2; ++q) y = __builtin_fmaf(dtap[q], xb(q) + xb(NC - 1 - q), y);
                        (void) P;
                    } else {
                        const float xs = xf[p];
    #pragma unroll
                        for (int q = 0; q < NC / 2; ++q) {
                            // central taps q and NC-1-q are the same float; fused: see the direct form
                            const int s0 = (P + NC - 1 - q) % NC;
                            const int s1 = (P + q) % NC;
                            if (q == 0) acc[s0] = ctap(q) * xs; else acc[s0] = __builtin_fmaf(ctap(q), xs, acc[s0]);
                            acc[s1] = __builtin_fmaf(ctap(q), xs, acc[s1]);
                        }
                        y = acc[P % NC];                    // y_c of output obase + p
                    }
                    if constexpr (FL2) {
                        if (p & 1) fo = __builtin_amdgcn_alignbit(fo, __float_as_uint(y), 30);
                        else fe = __builtin_amdgcn_alignbit(fe, __float_as_uint(y), 30);
                    } else {
                    neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(y), 31);
                    amb = __builtin_amdgcn_alignbit(amb, __float_as_uint(__builtin_fabsf(y) - eps_up), 31);
                    }
                    if (p % FIR_SIGN_FENCE == FIR_SIGN_FENCE - 1 && !K1S_DIRECT(NC)) {
    #pragma unroll
                        for (int g = 0; g < NC; g += 12) touch12(acc + g);
                    }
                }
                if constexpr (K1S_DIRECT(NC)) {
    #pragma unroll
                    for (int k = 0; k < NC - 1; ++k) tail[k] = xf[32 - (NC - 1) + k];
                }
                if constexpr (FL2) {
                    // output 2k sits in bits 31-2k (sign), 30-2k (|y'| >= 2) of fe, output 2k+1 in the same bits of fo
                    neg = (fe & 0xaaaaaaaau) | ((fo >> 1) & 0x55555555u);
                    amb = ~(((fe << 1) & 0xaaaaaaaau) | (fo & 0x55555555u));
                }
                uint32_t w = ~neg;
                const int valid = t1 - (t0 + obase);
                if (valid < 32) {
                    w &= ~0u << (32 - valid);
                    amb &= ~0u << (32 - valid);
                }
                bool zc_known = false, zc = false;
                if (__popc(amb) >= 8) {                         // a silent stretch?
                    uint32_t o = 0;
    #pragma unroll
                    for (int p = 0; p < 32; ++p) o |= __float_as_uint(xf[p]);      // (float) 0 is +0.0: all bits clear
                    zc = o == 0;
                    zc_known = true;
                    const int before = J0 + NC - 1;
                    if (zc && ((before <= 32 && zprev_known) ? zprev : all_zero(mb - before, before)) &&
                        all_zero(mb + 32, J0)) {
                        w &= ~amb;                              // y_ref == +0 for every one of them
                        amb = 0;
                    }
                }
                zprev_known = zc_known;
                zprev = zc;
                // The samples whose sign y_c cannot certify need the reference's ordered sum.  One lane doing that while
                // 63 wait is what this used to cost (a sixth of a wave's words hold such a sample somewhere); instead the
                // lane notes the output, leaves its bit 0, and the lanes work their lists off TOGETHER -- the k-th noted
                // output of every lane at the same time -- after the segment's last word, or at once when a lane has more
                // than its list holds (a silent or clipped stretch).  The rows differ per lane, the channel does not: as
                // many cache lines per load as lanes at work, the lines the one-at-a-time form fetched.  A word that has
                // left for memory gets its bit by an atomic OR (this lane stored it: one location, one thread, program
                // order); the up to four words still in registers are patched there.
                w &= ~amb;
                const int wi = b * UW + w3;                     // word of the segment (wave-uniform; static mod 4 for UW = 4)
                const int slot = WIDE ? (wi & 3) : 0;
                const bool last = t0 + obase + 32 >= t1;        // the segment's last word
                if constexpr (WIDE) {
                    while (amb && n_pend < FIR_SIGN_PEND) {
                        const int pos = __clz((int) amb);
                        amb &= ~(0x80000000u >> pos);
                        pend[n_pend * 64 + lane] = (uint16_t) (obase + pos);
                        ++n_pend;
                    }
                }
                if constexpr (WIDE) {
                    if (slot == 0) wq[0] = w; else if (slot == 1) wq[1] = w; else if (slot == 2) wq[2] = w; else wq[3] = w;
                }
                if (__any(amb != 0) || (WIDE && last && __any(n_pend != 0))) {
                    const int held = wi - slot;                 // first word that is still in registers
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
                        const int rel = (o >> 5) - held;
                        if (rel < 0) {
                            if (live) atomicOr(sgn + sgn_index((t0 + o) >> 5, N, cg), bit);
                        } else if constexpr (WIDE) {
                            wq[0] |= rel == 0 ? bit : 0u;
                            wq[1] |= rel == 1 ? bit : 0u;
                            wq[2] |= rel == 2 ? bit : 0u;
                            wq[3] |= rel == 3 ? bit : 0u;
                        } else {
                            w |= bit;
                        }
                    }
                    n_pend = 0;
                }
                if constexpr (WIDE) {
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
                } else if (live) {
                    sgn[sgn_index((t0 + obase) >> 5, N, cg)] = w;
                }
            }
        }
  } else {
        // Unrolled phases must hold whole turns of the accumulator ring (the ring index of every
        // phase is static).  NC = 12: 96 phases = three sign words, one group of 32 samples per word.
        // NC = 48: 96 phases would be 80 KB of code, more than the 64 KB instruction cache two CUs
        // share (measured: the kernel then runs at half its issue rate); 48 phases = three groups of
        // 16 samples are 31 KB, and a word is flushed after every second group.
        constexpr int GROUP = 16;
        constexpr int UNROLL = NC;
        float vtap[NC / 2];
#pragma unroll
        for (int q = 0; q < NC / 2; ++q) vtap[q] = ctap(q);
        constexpr int NG = UNROLL / GROUP;
        static_assert(UNROLL % NC == 0 && UNROLL % GROUP == 0, "whole ring turns, whole groups");
        // Two flag bits per sample, each gathered with ONE v_alignbit_b32 (shift the word
        // left, take the new bit from another register's bit 31) instead of compare +
        // select + shift/or, which was a third of this kernel's issue time:
        //   neg  collects the sign bit of y_c: 0 for y_c > 0 and for +0.0 (ambiguous anyway);
        //   amb  collects the sign bit of |y_c| - eps_up, eps_up = nextafter(eps): set
        //        exactly when |y_c| <= eps (a - b is negative or -0 iff a < b).
        uint32_t neg = 0, amb = 0, zor = 0;

        // The error bound behind eps is linear in the largest |x| a window holds and was taken for
        // 32768.  With 126 taps a third of this kernel's time went into exact re-evaluations
        // (126 taps each, ~5e-4 of the samples, most of them in the noise between messages where
        // |x| stays small), so this instantiation first takes the largest |x| of every sample its
        // segment's windows can touch and scales eps with it: eps_w = eps * M / 32768 (+ the absolute
        // slack for subnormal products).  One more pass over the segment's input (second read
        // mostly from L2 / Infinity Cache), 1 % more VALU work.
        float eps_w;
        // INLOOP: the maximum over a window that moves with the loop instead.  The reference windows of the outputs that
        // complete in a group of 16 rows starting at row mbg reach from mbg - (NC - 1) - J0 back to mbg + 15 + J0
        // ahead.  What lies behind (and the group itself) is covered by the maxima of this group and of the NHIST
        // groups before it; the J0 rows ahead have not been loaded yet: their taps are the table's last J0, and the
        // host has priced those with |x| = 32768 into eps_ahead (a few percent of the whole for a bell-shaped
        // table).  eps_w = eps_seen * M / 32768 + eps_ahead, per group.  No second pass over the segment (the
        // pre-pass read it twice and was a quarter of a wave's lifetime in load latency), and a loud message only
        // raises eps while it is inside the window, not for its whole segment.
        constexpr int NHIST = 6;                                // 96 rows behind the group: >= NC - 1 + J0 for J0 <= 49
        float hmax[NHIST];
        if constexpr (INLOOP) {
            float P = 0.0f;                                     // rows m0 - J0 .. m0 + NC - 2: everything before the first group
            const int plo = m0 - J0, phi = m0 + NC - 2;
            for (int mbase = plo; mbase <= phi; mbase += 16) {
                int v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    int m = mbase + i <= phi ? mbase + i : phi;
                    m = m < -NTaps ? -NTaps : m;
                    v[i] = load_sample(x, hist, m, N, NTaps, c);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) P = __builtin_fmaxf(P, __builtin_fabsf((float) v[i]));
            }
#pragma unroll
            for (int k = 0; k < NHIST; ++k) hmax[k] = P;
            eps_w = 0.0f;
        } else {
            const int mlo = t0 - d;                             // oldest sample of output t0's window
            const int mhi = t1 - 1 - d + NE - 1;                // newest sample of output t1-1's window
            int M = 0;
            for (int mbase = mlo; mbase <= mhi; mbase += 16) {
                int v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = mbase + i <= mhi ? mbase + i : mhi;
                    v[i] = load_sample(x, hist, m, N, NTaps, c);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int a = v[i] < 0 ? -v[i] : v[i];
                    M = a > M ? a : M;
                }
            }
            eps_w = eps_up * ((float) M * (1.0f / 32768.0f)) + 1e-30f;
        }

        uint32_t wq[4] = {0u, 0u, 0u, 0u};
        // one finished sign word: outputs obase .. obase+31 (the flags of the newest 32 samples)
        auto flush = [&](int obase) {
            const int mb = m0 + NC - 1 + obase;                 // sample of the word's first phase
            uint32_t w = ~neg;
            const int valid = t1 - (t0 + obase);
            if (valid < 32) {
                w &= ~0u << (32 - valid);
                amb &= ~0u << (32 - valid);
            }
            bool zc_known = false, zc = false;
            if (__popc(amb) >= 8) {                             // a silent stretch?
                zc = zor == 0;
                zc_known = true;
                const int before = J0 + NC - 1;
                if (zc && ((before <= 32 && zprev_known) ? zprev : all_zero(mb - before, before)) &&
                    all_zero(mb + 32, J0)) {
                    w &= ~amb;                                  // y_ref == +0 for every one of them
                    amb = 0;
                }
            }
            zprev_known = zc_known;
            zprev = zc;
            // the samples whose sign y_c cannot certify: exact ordered sum
            while (amb) {
                const int pos = __clz((int) amb);
                const uint32_t bit = 0x80000000u >> pos;
                amb &= ~bit;
                if (exact_positive(t0 + obase + pos)) w |= bit; else w &= ~bit;
            }
            {   // the sign words of 128 outputs leave as one 16-byte store (T % 128 == 0: t0's word index is 0 mod 4)
                const int slot = (obase >> 5) & 3;
                if (slot == 0) wq[0] = w; else if (slot == 1) wq[1] = w; else if (slot == 2) wq[2] = w; else wq[3] = w;
                const bool last = t0 + obase + 32 >= t1;        // the segment's last word
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
            }
            zor = 0;
        };

        // whole words, also for a last word that ends inside its first half: flush()'s silence test
        // needs all 32 samples of the word in zor (the phases past t1 are masked out)
        const int ngroups = (t1 - t0 + 31) / 32 * 2;
        // rows mbx .. mbx + GROUP - 1 of the lane's channel as floats
        auto load_group48 = [&](int mbx, float *dst) {
            if ((mbx >= 0) && (mbx + GROUP - 1 < L)) {
                // typed buffer loads as in the 12-tap path, but with the row in the VECTOR offset (one
                // two-operand add per load): this instantiation has no SGPRs to spare for 16 row offsets
                // -- with them it lost 6.8 vs 6.6 ms -- and still sheds the 64-bit address add and the
                // int -> float convert
                const int voff0 = coff + (int) ((uint32_t) (mbx - row0) * rowbytes);
    #pragma unroll
                for (int p = 0; p < GROUP; ++p)
                    dst[p] = fir_load_format_f32(rsrc_f, voff0 + (int) ((uint32_t) p * rowbytes), 0, 0);
            } else {
    #pragma unroll
                for (int p = 0; p < GROUP; ++p) {
                    int m = mbx + p;
                    m = (m < L) ? m : L - 1;
                    dst[p] = (float) load_sample(x, hist, m, N, NTaps, c);
                }
            }
        };
        for (int b = 0; b * NG < ngroups; ++b) {
    #pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int gi = b * NG + g;                      // group index within the segment
                const int gbase = gi * GROUP;                   // outputs gbase .. gbase+GROUP-1
                if (gi >= ngroups) break;
                float xf[GROUP];                                // the group's samples as floats (exact)
                const int mb = m0 + NC - 1 + gbase;             // sample of the group's first phase
                const bool interior = (mb >= 0) && (mb + GROUP - 1 < L);
                load_group48(mb, xf);
                {   // filter.c:118-119 peak, on the float bit patterns (see the 12-tap path)
                    int bp = 0;
                    if (interior) {
    #pragma unroll
                        for (int p = 0; p < GROUP; ++p) bp = __float_as_int(xf[p]) > bp ? __float_as_int(xf[p]) : bp;
                    } else {
    #pragma unroll
                        for (int p = 0; p < GROUP; ++p) {
                            const int m = mb + p;
                            const int v = (m >= 0 && m < L) ? __float_as_int(xf[p]) : 0;
                            bp = v > bp ? v : bp;
                        }
                    }
                    peakbits = bp > peakbits ? bp : peakbits;
                }
                if constexpr (INLOOP) {
                    float gm = 0.0f;
    #pragma unroll
                    for (int p = 0; p < GROUP; ++p) gm = __builtin_fmaxf(gm, __builtin_fabsf(xf[p]));
                    float M = gm;
    #pragma unroll
                    for (int k = 0; k < NHIST; ++k) M = __builtin_fmaxf(M, hmax[k]);
    #pragma unroll
                    for (int k = 0; k + 1 < NHIST; ++k) hmax[k] = hmax[k + 1];
                    hmax[NHIST - 1] = gm;
                    eps_w = __builtin_fmaf(eps_seen, M * (1.0f / 32768.0f), eps_ahead);
                }
    #pragma unroll
                for (int p = 0; p < GROUP; ++p) {
                    const int P = g * GROUP + p;                // phase within the unrolled block, P % NC static
                    const float xs = xf[p];
    #pragma unroll
                    for (int q = 0; q < NC / 2; ++q) {
                        // central taps q and NC-1-q are the same float.  y_c only has to stay within the
                        // certified distance of the exact central sum, so the products are FUSED into the
                        // accumulation: two v_fmac instead of a multiply and two adds, and one rounding per
                        // term instead of two (inside the host's bound, which is taken for mul + add)
                        const int s0 = (P + NC - 1 - q) % NC;
                        const int s1 = (P + q) % NC;
                        if (q == 0) acc[s0] = vtap[q] * xs; else acc[s0] = __builtin_fmaf(vtap[q], xs, acc[s0]);
                        acc[s1] = __builtin_fmaf(vtap[q], xs, acc[s1]);
                    }
                    const float y = acc[P % NC];                // y_c of output gbase + p
                    neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(y), 31);
                    amb = __builtin_amdgcn_alignbit(amb, __float_as_uint(__builtin_fabsf(y) - eps_w), 31);
                    if (p % FIR_SIGN_FENCE == FIR_SIGN_FENCE - 1) {
    #pragma unroll
                        for (int gg = 0; gg < NC; gg += 12) touch12(acc + gg);
                    }
                }
                // zor = OR of the word's samples, for the silence test in flush(), which only looks at
                // it when the word has >= 8 ambiguous samples.  A first half without any cannot belong
                // to a silent word, whatever its samples are.
                if (amb != 0) {
    #pragma unroll
                    for (int p = 0; p < GROUP; ++p) zor |= __float_as_uint(xf[p]);
                } else {
                    zor |= 1u;
                }
                if (gi & 1) flush(gbase - 16);
            }
        }
  }

    {
        const int pk = (int) __int_as_float(peakbits);
        peak = pk > peak ? pk : peak;
    }
    if (t1 == L) {                              // the last dc-NC+1 samples of the call
        const int shift = dc - NC + 1;
        for (int n = (L - shift > 0 ? L - shift : 0); n < L; ++n) {
            const int v = (int) x[(size_t) n * (size_t) N + c];
            peak = v > peak ? v : peak;
        }
    }
    if (live && peak > 0) atomicMax(&maxval[cg], peak);
    if (t1 == L && live) {                      // carry for the next call, as in K1
        for (int k = 0; k < NTaps; ++k) {
            const int m = L - NTaps + k;
            hist_out[(size_t) k * (size_t) N + cg] =
                (m >= 0) ? x[(size_t) m * (size_t) N + cg] : hist[(size_t) (NTaps + m) * (size_t) N + cg];
        }
        maxval_next[cg] = 0;
    }
}

int launch_fir_sign_quantum(int NC)
{
    // the 12-tap kernel stores four sign words at once: segments start on a multiple of 128 outputs
    // (the 48-tap one too, and its unrolled body is three words: 384)
    return NC <= 12 ? 128 : 384;
}

hipError_t launch_fir_sign(const FirLaunch &a, hipStream_t stream)
{
    if (a.dump || a.T % launch_fir_sign_quantum(a.NC) || !a.te_mem || (a.NC != 12 && a.NC != 48) || a.NE < a.NC || (a.NE - a.NC) % 2)
        return hipErrorInvalidValue;
    if (a.T > 65280) return hipErrorInvalidValue;       // the kernel notes open outputs as 16-bit offsets into the segment
    dim3 grid((a.N + 63) / 64, (a.L + a.T - 1) / a.T), block(64);
    const float eps_up = __builtin_nextafterf(a.eps, INFINITY);
    const int map = (a.map == 1 && grid.x % 8 == 0) ? 1 : 0;
    if (a.NC == 12 && a.NE == 32) {
        FirTaps<32> t;
        for (int j = 0; j < 32; ++j) t.te[j] = a.te[j];
        if (a.fscale > 0.0f)
            hipLaunchKernelGGL((fir_sign_kernel<32, 12, 32, false, true>), grid, block, 0, stream, a.x, a.hist, a.sgn, a.maxval,
                               a.hist_out, a.maxval_next, a.te_mem, a.N, a.L, a.T, a.d, a.NT, a.NE, eps_up, map, t, a.eps_seen, a.eps_ahead, a.fscale);
        else
        hipLaunchKernelGGL((fir_sign_kernel<32, 12, 32>), grid, block, 0, stream, a.x, a.hist, a.sgn, a.maxval,
                           a.hist_out, a.maxval_next, a.te_mem, a.N, a.L, a.T, a.d, a.NT, a.NE, eps_up, map, t, a.eps_seen, a.eps_ahead, a.fscale);
    } else if (a.NC == 12) {
        FirTaps<12> t;
        for (int j = 0; j < 12; ++j) t.te[j] = a.ctaps[j];
        if (a.fscale > 0.0f)
            hipLaunchKernelGGL((fir_sign_kernel<0, 12, 12, false, true>), grid, block, 0, stream, a.x, a.hist, a.sgn, a.maxval,
                               a.hist_out, a.maxval_next, a.te_mem, a.N, a.L, a.T, a.d, a.NT, a.NE, eps_up, map, t, a.eps_seen, a.eps_ahead, a.fscale);
        else
        hipLaunchKernelGGL((fir_sign_kernel<0, 12, 12>), grid, block, 0, stream, a.x, a.hist, a.sgn, a.maxval,
                           a.hist_out, a.maxval_next, a.te_mem, a.N, a.L, a.T, a.d, a.NT, a.NE, eps_up, map, t, a.eps_seen, a.eps_ahead, a.fscale);
    } else {
        FirTaps<48> t;
        for (int j = 0; j < 48; ++j) t.te[j] = a.ctaps[j];
        if (a.eps_seen > 0.0f && a.NE - a.NC <= 98)
            hipLaunchKernelGGL((fir_sign_kernel<0, 48, 48, true>), grid, block, 0, stream, a.x, a.hist, a.sgn, a.maxval,
                               a.hist_out, a.maxval_next, a.te_mem, a.N, a.L, a.T, a.d, a.NT, a.NE, eps_up, map, t, a.eps_seen, a.eps_ahead, a.fscale);
        else
        hipLaunchKernelGGL((fir_sign_kernel<0, 48, 48>), grid, block, 0, stream, a.x, a.hist, a.sgn, a.maxval,
                           a.hist_out, a.maxval_next, a.te_mem, a.N, a.L, a.T, a.d, a.NT, a.NE, eps_up, map, t, a.eps_seen, a.eps_ahead, a.fscale);
    }
    return hipGetLastError();
}

// Fallback for any other tap count (e.g. the 144-tap 192 kHz table): direct
// form, window re-read from L1/L2 per output.  Correct, not fast; bit-identical
// summation order.  One lane = one channel, grid.y = time segments.
__global__ __launch_bounds__(64) void fir_slice_generic_kernel(
    const int16_t *__restrict__ x, const int16_t *__restrict__ hist,
    uint32_t *__restrict__ sgn, float *__restrict__ dump, int *__restrict__ maxval,
    const float *__restrict__ taps, int N, int L, int T, int NT)
{
    const int lane = threadIdx.x;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int t0 = blockIdx.y * T;
    const int t1 = (t0 + T < L) ? t0 + T : L;
    if (t0 >= L) return;
    int peak = 0;
    for (int nb = t0; nb < t1; nb += 32) {
        uint32_t w = 0;
        for (int p = 0; p < 32; ++p) {
            const int n = nb + p;
            float y = 0.0f;
            if (n < t1) {
                float sum = 0.0f;
                for (int k = 0; k < NT; ++k) {
                    const float xs = (float) load_sample(x, hist, n - NT + k, N, NT, c);
                    sum = sum + xs * taps[k];
                }
                y = sum;
                const int v = (int) x[(size_t) n * (size_t) N + c];
                peak = v > peak ? v : peak;
                if (dump && live) dump[(size_t) n * (size_t) N + cg] = y;
            }
            w = (w << 1) | ((n < t1 && y > 0.0f) ? 1u : 0u);
        }
        if (live) sgn[sgn_index(nb >> 5, N, cg)] = w;
    }
    if (live && peak > 0) atomicMax(&maxval[cg], peak);
}

// filter.c:129-134 ring wrap, restated: after a call the carry is the last NT
// input samples, oldest first.  hist_out and hist_in are distinct buffers.
__global__ void fir_history_kernel(const int16_t *__restrict__ x,
                                   const int16_t *__restrict__ hist_in,
                                   int16_t *__restrict__ hist_out, int N, int L, int NT)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (c >= N) return;
    const int m = L - NT + k;                   // sample index of new row k
    hist_out[(size_t) k * N + c] = (m >= 0) ? x[(size_t) m * N + c]
                                            : hist_in[(size_t) (NT + m) * N + c];
}

// ---------------------------------------------------------------------------

hipError_t launch_fir_generic(const FirLaunch &a, hipStream_t stream)
{
    dim3 grid((a.N + 63) / 64, (a.L + a.T - 1) / a.T), block(64);
    hipLaunchKernelGGL(fir_slice_generic_kernel, grid, block, 0, stream, a.x, a.hist, a.sgn,
                       a.dump, a.maxval, a.d_taps, a.N, a.L, a.T, a.NT);
    return hipGetLastError();
}

hipError_t launch_fir_history(const int16_t *x, const int16_t *hist_in, int16_t *hist_out,
                              int N, int L, int NT, hipStream_t stream)
{
    dim3 grid((N + 255) / 256, NT), block(256);
    hipLaunchKernelGGL(fir_history_kernel, grid, block, 0, stream, x, hist_in, hist_out, N, L, NT);
    return hipGetLastError();
}

namespace scalar {
// NE == 32 only (the reference table after trimming); anything else goes to
// launch_fir_generic.
hipError_t launch_fir_slice(const FirLaunch &a, hipStream_t stream)
{
    if (a.NE != 32) return hipErrorInvalidValue;
    dim3 grid((a.N + 63) / 64, (a.L + a.T - 1) / a.T), block(64);
    FirTaps<32> t;
    bool sym = true;
    for (int j = 0; j < 32; ++j) {
        t.te[j] = a.te[j];
        uint32_t u0, u1;
        memcpy(&u0, &a.te[j], 4);
        memcpy(&u1, &a.te[31 - j], 4);
        sym = sym && (u0 == u1);
    }
#define FIR_LAUNCH(D, S)                                                                     \
    hipLaunchKernelGGL((fir_slice_kernel<32, D, S>), grid, block, 0, stream, a.x, a.hist, a.sgn, \
                       a.dump, a.maxval, a.hist_out, a.maxval_next, a.N, a.L, a.T, a.d, a.NT, t)
    if (a.dump) { if (sym) FIR_LAUNCH(true, true); else FIR_LAUNCH(true, false); }
    else        { if (sym) FIR_LAUNCH(false, true); else FIR_LAUNCH(false, false); }
#undef FIR_LAUNCH
    return hipGetLastError();
}
} // namespace scalar

} // namespace gnuais
