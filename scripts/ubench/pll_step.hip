// Cycles per PLL step for one wave alone on its SIMD (one 64-thread workgroup per CU): which part of
// pll_phase_kernel's step costs what.  hipcc --offload-arch=gfx950 -O3 -o pll_step.bin pll_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REC   "v_add_u32 %[U], %[X], %[T]\n\tv_bfe_i32 %[um], %[U], 22, 1\n\tv_xad_u32 %[X], %[Q], %[um], %[X]\n\t"
#define BITS  "v_lshrrev_b32 %[m], 23, %[U]\n\tv_lshrrev_b32 %[U], 28, %[U]\n\tv_lshlrev_b32_e64 %[m], %[m], 1\n\tv_lshl_add_u32 %[U], %[U], 8, %[pb]\n\t"
#define XOR   "ds_xor_b32 %[U], %[m]\n\t"
#define WR    "ds_write_b32 %[U], %[m]\n\t"
#define ACC   "v_xor_b32 %[acc], %[acc], %[m]\n\t"

template <int V>
__global__ __launch_bounds__(64) void k(unsigned *out, unsigned long long *cyc, int iters, unsigned T0)
{
    __shared__ unsigned pack[17 * 64];
    for (int i = threadIdx.x; i < 17 * 64; i += 64) pack[i] = 0;
    __syncthreads();
    unsigned X = threadIdx.x * 12345u, U, um, m, acc = 0, T = T0 + threadIdx.x * 77u;
    const unsigned Q = 819u << 7, pb = threadIdx.x * 4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define STEP4(body) asm volatile(body body body body : [X] "+v"(X), [U] "=&v"(U), [um] "=&v"(um), [m] "=&v"(m), [acc] "+v"(acc) : [T] "v"(T), [Q] "s"(Q), [pb] "v"(pb) : "memory")
        if (V == 0) STEP4(REC);
        if (V == 1) STEP4(REC BITS);
        if (V == 2) STEP4(REC BITS XOR);
        if (V == 3) STEP4(REC BITS WR);
        if (V == 4) STEP4(REC BITS ACC);
        if (V == 5) { STEP4(REC BITS XOR); STEP4(REC BITS XOR); asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); }
        if (V == 6) { STEP4(REC BITS XOR); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = X + acc + pack[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char *name, unsigned *out, unsigned long long *cyc, int blocks)
{
    const int iters = 20000;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1234567u);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double) h[i];
    printf("%-34s %6.1f s_memtime ticks per step\n", name, s / blocks / iters / 4.0);
}

int main()
{
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, 4 * 64 * 1024);
    hipMalloc(&cyc, 8 * 1024);
    const int blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("recurrence (3 VALU)", out, cyc, blocks);
        run<1>("+ bit/word address (7 VALU)", out, cyc, blocks);
        run<2>("+ ds_xor_b32", out, cyc, blocks);
        run<3>("+ ds_write_b32 instead", out, cyc, blocks);
        run<4>("+ v_xor into a register instead", out, cyc, blocks);
        run<5>("full step x8, lgkmcnt(5) (=2 rows)", out, cyc, blocks);
        run<6>("full step, lgkmcnt(4) per 4 steps", out, cyc, blocks);
    }
    return 0;
}
