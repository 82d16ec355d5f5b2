// pll_nrzi.hip -- K2a: bit-clock recovery PLL, and K2x: slice + NRZI decode,
// for gfx950.
//
// Together they stand in for the per-sample loop of receiver_run(), gnuais
// src/receiver.c:109-135, for a whole batch of channels.
//
// K2a (pll_core_kernel).  The phase update is a nonlinear recurrence in time
// (the nudge direction depends on the current phase), so time stays sequential
// per channel and the batch axis is the parallel one: one lane = one channel,
// one wave = 64 adjacent channels reading K1's sign words sgn[w][c] (coalesced).
// A wave that is alone on its SIMD issues roughly one instruction every ~5
// cycles whatever it is, so the whole game is instructions per sample.  The
// kernel therefore keeps ONLY the recurrence: 6 VALU instructions per sample,
//   v_bfe_i32   tm = -(transition at this sample)          receiver.c:113
//   v_ashrrev   um = -(pll >= 0x8000)                      receiver.c:114
//   v_bfi       k  = um ? INC-Q : INC+Q                    receiver.c:115-117
//   v_bfi       k  = tm ? k : INC
//   v_add_co    P += k            (carry = `pll > 0xffff`)  receiver.c:122-124
//   v_addc      O  = 2*O + carry  (slice marks of the word)
// hand-scheduled (the carry is consumed two instructions later, which is the
// gfx950 wait-state requirement for a VALU-written VCC).  The 16-bit phase of
// the reference lives in the top half of a 32-bit register (P = pll << 16), so
// `pll &= 0xffff` (receiver.c:133) is the natural wrap of the add and
// `pll < 0x8000` is the sign bit.  The nudge never carries by itself (pll <
// 0x8000 -> +q stays < 0x10000; pll >= 0x8000 -> -q stays > 0), so folding nudge
// and increment into one add leaves the overflow test unchanged.
//
// K2x (nrzi_extract_kernel).  Everything that is NOT a recurrence runs in
// parallel over (channel, 2048-sample segment): at every slice mark take the
// level (receiver.c:126), NRZI-decode against the previous slice's level
// (receiver.c:128-132; for the first mark of a segment that level is found by
// looking back through the preceding words) and pack the bits.  Output: one
// pack of seg_words words + a bit count per (channel, segment); bit k of a
// pack is at word k/32, bit k%32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include "kernels.h"

namespace gnuais {

#define PLL_HEAD(sh)                                                                      \
    "v_bfe_i32 %[tm], %[D], " #sh ", 1\n\t"                                               \
    "v_ashrrev_i32 %[um], 31, %[P]\n\t"
#define PLL_TAIL                                                                          \
    "v_bfi_b32 %[k], %[um], %[Km], %[Kp]\n\t"                                             \
    "v_bfi_b32 %[k], %[tm], %[k], %[INC]\n\t"                                             \
    "v_add_co_u32 %[P], vcc, %[P], %[k]\n\t"
#define PLL_CARRY "v_addc_co_u32 %[O], vcc, %[O], %[O], vcc\n\t"
#define PLL_STEP(sh) PLL_HEAD(sh) PLL_CARRY PLL_TAIL

// one full word (32 samples), oldest sample = bit 31 of D
__device__ __forceinline__ void pll_word(uint32_t D, uint32_t &P, uint32_t &O, uint32_t Kp,
                                         uint32_t Km, uint32_t INC)
{
    uint32_t tm, um, k;
    asm volatile(
        PLL_HEAD(31) PLL_TAIL
        PLL_STEP(30) PLL_STEP(29) PLL_STEP(28) PLL_STEP(27) PLL_STEP(26) PLL_STEP(25)
        PLL_STEP(24) PLL_STEP(23) PLL_STEP(22) PLL_STEP(21) PLL_STEP(20) PLL_STEP(19)
        PLL_STEP(18) PLL_STEP(17) PLL_STEP(16) PLL_STEP(15) PLL_STEP(14) PLL_STEP(13)
        PLL_STEP(12) PLL_STEP(11) PLL_STEP(10) PLL_STEP(9) PLL_STEP(8) PLL_STEP(7)
        PLL_STEP(6) PLL_STEP(5) PLL_STEP(4) PLL_STEP(3) PLL_STEP(2) PLL_STEP(1)
        PLL_STEP(0)
        "s_nop 1\n\t"
        PLL_CARRY
        : [P] "+v"(P), [O] "+v"(O), [tm] "=&v"(tm), [um] "=&v"(um), [k] "=&v"(k)
        : [D] "v"(D), [Kp] "v"(Kp), [Km] "v"(Km), [INC] "v"(INC)
        : "vcc");
}

// One workgroup = 64 channels (group blockIdx.x) through the whole call, as TWO waves:
//   wave 0  the recurrence.  It touches only LDS: beside a FIR that keeps the CU's vector
//           memory pipeline full, every global load / store this wave issued cost it
//           microseconds at ISSUE (measured: 0.60 ms without memory instructions, 1.3 ms
//           with loads or stores, whatever the prefetch distance);
//   wave 1  the mover.  Streams the sign words into an LDS ring PLL_RING words ahead
//           (~50 us of lead) and drains the slice-mark words from a second ring, in
//           batches.  Its own stalls are absorbed by the rings.
// The rings are handed over through three monotonic LDS counters (words loaded, words
// computed, words stored); the LDS unit serves DS instructions in order, so a counter
// written after the data (release) is seen after the data (acquire).
// The launch asks for PLL_LDS_BYTES of LDS, more than half a CU's 160 KB, so the dispatcher
// places at most ONE of these workgroups per CU and two chains never share a SIMD (that
// doubles both; seen when the FIR's grid grew, scripts/ubench/overlap.hip).
constexpr int PLL_RING = 128;        // words per ring (in and out): 2 x 32 KB
constexpr int PLL_BATCH = 16;        // words the mover handles per touch of the memory pipeline
static_assert(2 * PLL_RING * 64 * 4 + 64 <= PLL_LDS_BYTES, "rings must fit the LDS we reserve");
static_assert(PLL_BATCH <= PLL_PAD_ROWS, "the mover reads whole batches past the last word");

__global__ __launch_bounds__(128) void pll_core_kernel(
    const uint32_t *__restrict__ sgn, uint32_t *__restrict__ ovf, uint32_t *__restrict__ pllst,
    uint32_t *__restrict__ watchdog, int N, int L, uint32_t pllinc)
{
    extern __shared__ uint32_t lds[];
    uint32_t *rin = lds, *rout = lds + PLL_RING * 64, *flag = lds + 2 * PLL_RING * 64;
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int W = (L + 31) >> 5;
    const int Wfull = L >> 5;                     // words with all 32 samples valid
    if (threadIdx.x < 3) flag[threadIdx.x] = 0;   // 0 loaded, 1 computed, 2 stored
    __syncthreads();
    const unsigned long long t_start = wall_clock64();
    // nothing here may spin forever: a wave that waits longer than this gives up
    // (200 ms; the two waves of a workgroup normally hand over every few microseconds) and says
    // so in *watchdog, which the host turns into an error when the frames are drained
    auto expired = [&]() {
        if (wall_clock64() - t_start <= 20000000ull) return false;
        if (lane == 0) atomicOr(watchdog, 1u);
        return true;
    };

    if (role == 1) {                              // ---- the mover ----
        // Loads are software-pipelined: batch k+1 is in flight while batch k is written to the
        // ring, so one memory round trip (several microseconds beside the FIR) is paid per
        // batch of PLL_BATCH words only once, not in series with the LDS writes and the stores.
        const uint32_t *__restrict__ src = sgn + c;
        int w_issue = 0, w_in = 0, w_out = 0;     // words: loads issued / in the ring / stored
        uint32_t va[PLL_BATCH], vb[PLL_BATCH];
        bool a_pending = false;
        while (w_out < W && !expired()) {
            bool moved = false;
            const int done = (int) __hip_atomic_load(flag + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool can_b = w_issue < W && w_issue + PLL_BATCH - done <= PLL_RING;
            if (can_b) {
#pragma unroll
                for (int q = 0; q < PLL_BATCH; ++q) vb[q] = src[(size_t) (w_issue + q) * (size_t) N];
            }
            if (a_pending) {
#pragma unroll
                for (int q = 0; q < PLL_BATCH; ++q) rin[((w_in + q) % PLL_RING) * 64 + lane] = va[q];
                w_in = w_in + PLL_BATCH < W ? w_in + PLL_BATCH : W;
                __hip_atomic_store(flag + 0, (uint32_t) w_in, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                a_pending = false;
                moved = true;
            }
            if (can_b) {
#pragma unroll
                for (int q = 0; q < PLL_BATCH; ++q) va[q] = vb[q];
                a_pending = true;
                w_issue += PLL_BATCH;
                moved = true;
            }
            if (done - w_out >= PLL_BATCH || (done == W && done > w_out)) {
                const int n = done - w_out < PLL_BATCH ? done - w_out : PLL_BATCH;
#pragma unroll
                for (int q = 0; q < PLL_BATCH; ++q)
                    if (q < n) {
                        const uint32_t O = rout[((w_out + q) % PLL_RING) * 64 + lane];
                        if (live) ovf[(size_t) (w_out + q) * (size_t) N + cg] = O;
                    }
                w_out += n;
                __hip_atomic_store(flag + 2, (uint32_t) w_out, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                moved = true;
            }
            if (!moved) __builtin_amdgcn_s_sleep(16);
        }
        return;
    }

    // ---- the recurrence ----
    // a long dependent chain: when it shares a SIMD with other waves it must win every
    // issue slot it can use
    __builtin_amdgcn_s_setprio(3);
    const uint32_t st = pllst[c];
    uint32_t P = (st & 0xffffu) << 16;            // receiver.h:40 pll, scaled
    uint32_t prev = (st >> 16) & 1u;              // receiver.h:44
    const uint32_t INC = pllinc << 16;            // receiver.c:122
    const uint32_t Q = (pllinc / 16u) << 16;      // receiver.c:84,115,117
    const uint32_t Kp = INC + Q, Km = INC - Q;
    constexpr int PF = PLL_PAD;

    for (int w0 = 0; w0 < Wfull; w0 += PF) {
        const int w1 = w0 + PF < Wfull ? w0 + PF : Wfull;
        while ((int) __hip_atomic_load(flag + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < w1 && !expired())
            __builtin_amdgcn_s_sleep(1);
        while (w1 - (int) __hip_atomic_load(flag + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) > PLL_RING && !expired())
            __builtin_amdgcn_s_sleep(1);
        uint32_t cur[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) cur[q] = rin[((w0 + q) % PLL_RING) * 64 + lane];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            if (w0 + q < Wfull) {
                const uint32_t S = cur[q];                          // bit 31 = oldest
                const uint32_t D = S ^ ((S >> 1) | (prev << 31));   // receiver.c:113
                prev = S & 1u;
                uint32_t O = 0;
                pll_word(D, P, O, Kp, Km, INC);
                rout[((w0 + q) % PLL_RING) * 64 + lane] = O;
            }
        }
        __hip_atomic_store(flag + 1, (uint32_t) w1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (Wfull < W) {                              // last, partial word (L % 32 samples)
        const int nv = L - Wfull * 32;
        while ((int) __hip_atomic_load(flag + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < W && !expired())
            __builtin_amdgcn_s_sleep(1);
        while (W - (int) __hip_atomic_load(flag + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) > PLL_RING && !expired())
            __builtin_amdgcn_s_sleep(1);
        const uint32_t S = rin[(Wfull % PLL_RING) * 64 + lane];
        uint32_t D = S ^ ((S >> 1) | (prev << 31));
        uint32_t O = 0;
        for (int i = 0; i < nv; ++i) {
            const bool t = (int32_t) D < 0;
            D <<= 1;
            const uint32_t Kt = ((int32_t) P < 0) ? Km : Kp;
            const uint32_t K = t ? Kt : INC;
            const uint32_t Pn = P + K;
            O = (O << 1) | (Pn < P ? 1u : 0u);
            P = Pn;
        }
        O <<= (32 - nv);                          // left-align like S
        prev = (S >> (32 - nv)) & 1u;
        rout[(Wfull % PLL_RING) * 64 + lane] = O;
        __hip_atomic_store(flag + 1, (uint32_t) W, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (live) pllst[cg] = (P >> 16) | (prev << 16);
}

// grid.x = channel group (64 channels), grid.y = segment of SEG_WORDS words
__global__ __launch_bounds__(64) void nrzi_extract_kernel(
    const uint32_t *__restrict__ sgn, const uint32_t *__restrict__ ovf,
    const uint32_t *__restrict__ lastbit, uint32_t *__restrict__ segbits,
    uint32_t *__restrict__ segcnt, int N, int L, int n_seg, int seg_words)
{
    const int cg = blockIdx.x * 64 + threadIdx.x;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int seg = blockIdx.y;
    const int W = (L + 31) >> 5;
    const int w0 = seg * SEG_WORDS;
    const int w1 = (w0 + SEG_WORDS < W) ? w0 + SEG_WORDS : W;
    if (w0 >= W) {
        if (live) segcnt[(size_t) cg * n_seg + seg] = 0;
        return;
    }
    // level at the last slice before this segment (receiver.h:38 lastbit): the
    // latest mark is the lowest set bit of the nearest earlier non-empty word
    uint32_t last = lastbit[c];
    for (int w = w0 - 1; w >= 0; --w) {
        const uint32_t O = ovf[(size_t) w * (size_t) N + c];
        if (O) {
            const uint32_t S = sgn[(size_t) w * (size_t) N + c];
            last = (S >> (__ffs((int) O) - 1)) & 1u;
            break;
        }
    }
    uint32_t *__restrict__ out = segbits + ((size_t) c * n_seg + seg) * (size_t) seg_words;
    uint32_t outw = 0, outn = 0, wr = 0;
    for (int w = w0; w < w1; ++w) {
        const uint32_t S = sgn[(size_t) w * (size_t) N + c];
        uint32_t O = ovf[(size_t) w * (size_t) N + c];
        while (O) {                                             // oldest mark first
            const int pos = __clz((int) O);
            const uint32_t level = (S >> (31 - pos)) & 1u;      // receiver.c:126
            O &= ~(0x80000000u >> pos);
            const uint32_t b = (level ^ last) ^ 1u;             // receiver.c:128
            last = level;                                       // receiver.c:132
            outw |= b << outn;
            if (++outn == 32) {
                if (live && (int) wr < seg_words) out[wr] = outw;
                ++wr;
                outw = 0;
                outn = 0;
            }
        }
    }
    if (outn && live && (int) wr < seg_words) out[wr] = outw;
    if (live) segcnt[(size_t) cg * n_seg + seg] = wr * 32 + outn;
}

// after K2x: carry the level of the call's last slice into the next call.  Runs
// after K2x in the same stream (K2x reads lastbit[]).
__global__ void nrzi_lastbit_kernel(const uint32_t *__restrict__ sgn,
                                    const uint32_t *__restrict__ ovf,
                                    uint32_t *__restrict__ lastbit, int N, int L)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int W = (L + 31) >> 5;
    for (int w = W - 1; w >= 0; --w) {
        const uint32_t O = ovf[(size_t) w * (size_t) N + c];
        if (O) {
            const uint32_t S = sgn[(size_t) w * (size_t) N + c];
            lastbit[c] = (S >> (__ffs((int) O) - 1)) & 1u;
            return;
        }
    }
}

hipError_t launch_pll_core(const PllLaunch &a, hipStream_t stream)
{
    // one wave per CU while the channel groups fit one round; beyond that share the CUs evenly
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
        hipError_t e = hipFuncSetAttribute((const void *) pll_core_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PLL_LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    const int groups = (a.N + 63) / 64, per_cu = (groups + n_cu - 1) / n_cu;
    const int need = 2 * PLL_RING * 64 * 4 + 64;
    const int lds = per_cu <= 1 ? PLL_LDS_BYTES : std::max(need, (160 * 1024 / per_cu) & ~1023);
    hipLaunchKernelGGL(pll_core_kernel, dim3(groups), dim3(128), lds, stream, a.sgn, a.ovf,
                       a.pll, a.watchdog, a.N, a.L, a.pllinc);
    return hipGetLastError();
}

hipError_t launch_nrzi_extract(const PllLaunch &a, hipStream_t stream)
{
    hipLaunchKernelGGL(nrzi_extract_kernel, dim3((a.N + 63) / 64, a.n_seg), dim3(64), 0, stream,
                       a.sgn, a.ovf, a.lastbit, a.segbits, a.segcnt, a.N, a.L, a.n_seg, a.seg_words);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nrzi_lastbit_kernel, dim3((a.N + 255) / 256), dim3(256), 0, stream, a.sgn,
                       a.ovf, a.lastbit, a.N, a.L);
    return hipGetLastError();
}

} // namespace gnuais
