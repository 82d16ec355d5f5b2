// pll_common.h -- small pieces the PLL stage's kernels share (pll_h3.hip, pll_tp.hip): a wave maximum, the LDS hand-over
// primitives, the pack geometry and the byte table of transition positions.  (gnuais src/receiver.c:109-135.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gnuais {

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const uint32_t u = (uint32_t) __shfl_xor((int) v, o);
        v = u > v ? u : v;
    }
    return v;
}

constexpr int PLL_PACKW = PACK_STRIDE + 1;   // words per lane and pack buffer: the pack + its bit count
constexpr int PLL_LUT_BYTES = 2048;

// The hand-over counters live in LDS and guard LDS data only.  The LDS unit executes a wave's DS
// instructions in order, so "data, then counter" on the producer side and "counter, then data" on
// the consumer side is all the ordering needed; a C++ release / acquire here would also wait for
// every global load and store the wave has in flight (s_waitcnt vmcnt(0)) -- which is exactly what
// the scanner's load queue must not do.
__device__ __forceinline__ void lds_flag_store(uint32_t *f, uint32_t v)
{
    asm volatile("" ::: "memory");
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint32_t lds_flag_load(uint32_t *f)
{
    asm volatile("" ::: "memory");
    const uint32_t v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}

__host__ __device__ inline int n_seg_cap(int L)
{
    return (((L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
}

// the byte table: entry v = the positions of v's set bits, MSB (oldest sample) first, one per byte
__device__ __forceinline__ void pll_fill_lut(uint64_t *lut, int tid, int n_threads)
{
    for (int v = tid; v < 256; v += n_threads) {
        uint64_t e = 0;
        int n = 0;
        for (int b = 7; b >= 0; --b)
            if (v & (1 << b)) e |= (uint64_t) (7 - b) << (8 * n++);
        lut[v] = e;
    }
}

} // namespace gnuais
