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
// Output: 1 bit per sample, the sign words sgn[w][c] (bit 31 = oldest of the 32
// samples of word w), optionally the fp32 filter output (parity dump), and the
// per-channel peak positive sample (filter_run_buf's return value).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

#ifndef FIR_VARIANT
#define FIR_VARIANT scalar      // Makefile builds this file twice: scalar (-fno-slp-vectorize,
#endif                          // v_mul_f32/v_add_f32) and packed (v_pk_mul_f32/v_pk_add_f32)

namespace gnuais {

template <int NE>
struct FirTaps { float te[NE]; };

__device__ __forceinline__ int load_sample(const int16_t *__restrict__ x,
                                           const int16_t *__restrict__ hist,
                                           int m, int N, int NT, int c)
{
    // m < 0: sample of an earlier run call -> history rows (NT rows, oldest first)
    const int16_t *p = (m >= 0) ? (x + (size_t) m * (size_t) N + c)
                                : (hist + (size_t) (NT + m) * (size_t) N + c);
    return (int) *p;
}

namespace FIR_VARIANT {

// grid: x = channel group (64 channels), y = time segment of T outputs
// (T a multiple of 32).  block = 64 threads = one wave.
template <int NE, bool DUMP>
__global__ __launch_bounds__(64) void fir_slice_kernel(
    const int16_t *__restrict__ x, const int16_t *__restrict__ hist,
    uint32_t *__restrict__ sgn, float *__restrict__ dump, int *__restrict__ maxval,
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
                const float p = taps.te[j] * xs;
                if (j == 0) acc[s] = p + 0.0f; else acc[s] = acc[s] + p;
            }
        }
    }

    const int nblk = (t1 - t0 + 31) >> 5;
    for (int b = 0; b < nblk; ++b) {
        const int obase = b * 32;               // outputs obase .. obase+31
        int xi[32];
        // the 32 samples of this block: i = NE-1+obase+p  ->  m = mb + p
        const int mb = m0 + NE - 1 + obase;
        const bool interior = (mb >= 0) && (mb + 31 < L);
        if (interior) {
            const int16_t *row = x + (size_t) mb * (size_t) N + c;
#pragma unroll
            for (int p = 0; p < 32; ++p) xi[p] = (int) row[(size_t) p * (size_t) N];
        } else {
            // first block of a call (3 history samples) or the last, partial block
            // (addresses clamped; those outputs are masked out below)
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                int m = mb + p;
                m = (m < L) ? m : L - 1;
                xi[p] = load_sample(x, hist, m, N, NT, c);
            }
        }
        // filter.c:118-119: peak positive input sample of this call.  The block's
        // samples are n = mb .. mb+31; over all segments they cover [-shift, L-shift)
        // (shift = d-NE+1 trailing zero taps), so history samples (n < 0) are
        // masked here and the last `shift` samples are added after the loop.
        {
            int bp = 0;
            if (interior) {
#pragma unroll
                for (int p = 0; p < 32; ++p) bp = xi[p] > bp ? xi[p] : bp;
            } else {
#pragma unroll
                for (int p = 0; p < 32; ++p) {
                    const int m = mb + p;
                    const int v = (m >= 0 && m < L) ? xi[p] : 0;
                    bp = v > bp ? v : bp;
                }
            }
            peak = bp > peak ? bp : peak;
        }
        uint32_t w = 0;
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const float xs = (float) xi[p];
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int s = (NE - 1 + p - j) & (NE - 1);
                const float pr = taps.te[j] * xs;
                if (j == 0) acc[s] = pr + 0.0f; else acc[s] = acc[s] + pr;
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
        if (live) sgn[(size_t) ((t0 + obase) >> 5) * (size_t) N + cg] = w;
    }

    // the last segment also owns the final `shift` samples of the call
    if (t1 == L) {
        const int shift = d - NE + 1;
        for (int n = (L - shift > 0 ? L - shift : 0); n < L; ++n) {
            const int v = (int) x[(size_t) n * (size_t) N + c];
            peak = v > peak ? v : peak;
        }
    }
    if (live && peak > 0) atomicMax(&maxval[cg], peak);
}

} // namespace FIR_VARIANT

#ifdef FIR_PRIMARY
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
        if (live) sgn[(size_t) (nb >> 5) * (size_t) N + cg] = w;
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

#endif // FIR_PRIMARY

namespace FIR_VARIANT {
// NE == 32 only (the reference table after trimming); anything else goes to
// launch_fir_generic.
hipError_t launch_fir_slice(const FirLaunch &a, hipStream_t stream)
{
    if (a.NE != 32) return hipErrorInvalidValue;
    dim3 grid((a.N + 63) / 64, (a.L + a.T - 1) / a.T), block(64);
    FirTaps<32> t;
    for (int j = 0; j < 32; ++j) t.te[j] = a.te[j];
    if (a.dump)
        hipLaunchKernelGGL((fir_slice_kernel<32, true>), grid, block, 0, stream, a.x, a.hist,
                           a.sgn, a.dump, a.maxval, a.N, a.L, a.T, a.d, a.NT, t);
    else
        hipLaunchKernelGGL((fir_slice_kernel<32, false>), grid, block, 0, stream, a.x, a.hist,
                           a.sgn, a.dump, a.maxval, a.N, a.L, a.T, a.d, a.NT, t);
    return hipGetLastError();
}
} // namespace FIR_VARIANT

} // namespace gnuais
