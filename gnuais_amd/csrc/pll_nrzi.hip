// pll_nrzi.hip -- K2a: bit-clock recovery PLL + NRZI decode for gfx950.
//
// Stands in for the per-sample loop of receiver_run(), gnuais
// src/receiver.c:109-135, for a whole batch of channels.
//
// The loop is a nonlinear recurrence in time (the nudge direction depends on
// the current phase), so time stays sequential per channel; the batch axis is
// the parallel one: one lane = one channel, one wave = 64 adjacent channels
// reading K1's sign words sgn[w][c] (coalesced, 256 B per wave per 32 samples).
//
// The 16-bit phase of the reference lives in the top half of a 32-bit register
// (P = pll << 16), so `pll > 0xffff; pll &= 0xffff` (receiver.c:124,133) is the
// carry-out of one 32-bit add and `pll < 0x8000` (receiver.c:114) is the sign
// bit.  The nudge never carries by itself (pll < 0x8000 -> +q stays < 0x10000,
// pll >= 0x8000 -> -q stays > 0), so folding nudge and increment into one add
// leaves the overflow test unchanged.
//
// Output: the recovered (NRZI-decoded) bits, packed LSB first per channel:
// bit k of channel c = bits[k/32][c] >> (k%32) & 1; nbits[c] = count.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gnuais {

__global__ __launch_bounds__(64) void pll_nrzi_kernel(
    const uint32_t *__restrict__ sgn, uint32_t *__restrict__ pllst,
    uint32_t *__restrict__ bits, uint32_t *__restrict__ nbits,
    int N, int L, int bits_words, uint32_t pllinc)
{
    const int cg = blockIdx.x * 64 + threadIdx.x;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;

    const uint32_t st = pllst[c];
    uint32_t P = (st & 0xffffu) << 16;            // receiver.h:40 pll, scaled
    uint32_t prev = (st >> 16) & 1u;              // receiver.h:44
    uint32_t last = (st >> 17) & 1u;              // receiver.h:38 lastbit
    const uint32_t INC = pllinc << 16;            // receiver.c:122
    const uint32_t Q = (pllinc / 16u) << 16;      // receiver.c:84,115,117
    const uint32_t Kp = INC + Q, Km = INC - Q;

    uint32_t outw = 0, outn = 0, wr = 0;
    const int W = (L + 31) >> 5;

    // The sign words of a channel are N*4 bytes apart, so every lane's load is its
    // own 4-byte gather; with one wave per SIMD nothing else hides the HBM latency.
    // Keep PF words in flight: the group for step g+1 is requested before the
    // group of step g is consumed.
    constexpr int PF = 8;
    uint32_t nxt[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) nxt[q] = (q < W) ? sgn[(size_t) q * (size_t) N + c] : 0u;

    for (int w0 = 0; w0 < W; w0 += PF) {
        uint32_t grp[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) grp[q] = nxt[q];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int wn = w0 + PF + q;
            nxt[q] = (wn < W) ? sgn[(size_t) wn * (size_t) N + c] : 0u;
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int w = w0 + q;
            if (w >= W) break;
            const uint32_t S = grp[q];                          // bit 31 = oldest
            const int nv = (L - w * 32 < 32) ? L - w * 32 : 32;
            // transition word: bit (31-i) = s_i ^ s_{i-1}   (receiver.c:113)
            uint32_t D = S ^ ((S >> 1) | (prev << 31));
            uint32_t O = 0;                                     // overflow (slice) marks
            if (nv == 32) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const bool t = (int32_t) D < 0;
                    D <<= 1;
                    const uint32_t Kt = ((int32_t) P < 0) ? Km : Kp; // receiver.c:114-118
                    const uint32_t K = t ? Kt : INC;
                    const uint32_t Pn = P + K;                    // receiver.c:122
                    O = (O << 1) | (Pn < P ? 1u : 0u);            // receiver.c:124
                    P = Pn;
                }
                prev = S & 1u;
            } else {
                for (int i = 0; i < nv; ++i) {
                    const bool t = (int32_t) D < 0;
                    D <<= 1;
                    const uint32_t Kt = ((int32_t) P < 0) ? Km : Kp;
                    const uint32_t K = t ? Kt : INC;
                    const uint32_t Pn = P + K;
                    O = (O << 1) | (Pn < P ? 1u : 0u);
                    P = Pn;
                }
                O <<= (32 - nv);                                  // left-align like S
                prev = (S >> (32 - nv)) & 1u;
            }
            // slice + NRZI at every overflow, oldest first (receiver.c:126-132)
            while (O) {
                const int pos = __clz((int) O);
                const uint32_t level = (S >> (31 - pos)) & 1u;
                O &= ~(0x80000000u >> pos);
                const uint32_t b = (level ^ last) ^ 1u;
                last = level;
                outw |= b << outn;
                if (++outn == 32) {
                    if (live && (int) wr < bits_words) bits[(size_t) wr * (size_t) N + cg] = outw;
                    ++wr;
                    outw = 0;
                    outn = 0;
                }
            }
        }
    }
    if (outn && live && (int) wr < bits_words) bits[(size_t) wr * (size_t) N + cg] = outw;
    if (live) {
        nbits[cg] = wr * 32 + outn;
        pllst[cg] = (P >> 16) | (prev << 16) | (last << 17);
    }
}

hipError_t launch_pll_nrzi(const PllLaunch &a, hipStream_t stream)
{
    dim3 grid((a.N + 63) / 64), block(64);
    hipLaunchKernelGGL(pll_nrzi_kernel, grid, block, 0, stream, a.sgn, a.pll, a.bits, a.nbits,
                       a.N, a.L, a.bits_words, a.pllinc);
    return hipGetLastError();
}

} // namespace gnuais
