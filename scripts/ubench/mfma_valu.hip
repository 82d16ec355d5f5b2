// Microbenchmark: throughput of v_mfma_f32_4x4x1_16b_f32, v_add_f32, v_mul_f32,
// v_pk_add_f32 and of MFMA/VALU mixes on gfx950, per SIMD cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0)
{
    float acc[16];
    f32x4 pr[8];
    f32x2 pk[8];
    for (int i = 0; i < 16; ++i) acc[i] = a0 + i;
    for (int i = 0; i < 8; ++i) { pr[i] = {0.f, 0.f, 0.f, 0.f}; pk[i] = {a0, b0}; }
    float a = a0 + threadIdx.x, b = b0;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 8 independent MFMA
#pragma unroll
            for (int g = 0; g < 8; ++g) pr[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, 0);
            asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(pr[7]));
            asm volatile("" : "+v"(a));
        } else if (MODE == 1) {   // 32 independent v_add
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = acc[i] + b;
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
        } else if (MODE == 2) {   // 8 MFMA + 32 v_add interleaved (1 : 4)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                pr[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[(4 * g + q) & 15] = acc[(4 * g + q) & 15] + b;
            }
            asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(pr[7]));
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
            asm volatile("" : "+v"(a));
        } else if (MODE == 3) {   // 16 v_pk_add_f32 (32 lane-adds)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) pk[i] = pk[i] + f32x2{b, b};
            asm volatile("" : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]), "+v"(pk[4]), "+v"(pk[5]), "+v"(pk[6]), "+v"(pk[7]));
        } else if (MODE == 4) {   // 16 mul + 16 add (scalar FIR mix)
#pragma unroll
            for (int i = 0; i < 16; ++i) { float p = a * acc[(i + 5) & 15]; asm volatile("" : "+v"(p)); acc[i] = acc[i] + p; }
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += pr[i][0] + pr[i][3] + pk[i][0] + pk[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(int blocks, int iters, float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f, 1e-3f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    const int iters = 20000;
    const char *names[] = {"8 mfma_4x4x1", "32 v_add", "8 mfma + 32 v_add", "16 v_pk_add", "16 mul+16 add"};
    for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD: blocks = 256 CUs * wps (256 thr = 4 waves = 1/SIMD)
        const int blocks = 256 * wps;
        double ms[5] = {run<0>(blocks, iters, d), run<1>(blocks, iters, d), run<2>(blocks, iters, d),
                        run<3>(blocks, iters, d), run<4>(blocks, iters, d)};
        for (int m = 0; m < 5; ++m) {
            // per SIMD: wps waves * iters iterations; report ns per iteration per SIMD-resident wave set
            printf("waves/SIMD=%d  %-20s %8.3f ms  -> %.1f ns per iteration-per-wave (x%d waves)\n", wps,
                   names[m], ms[m], ms[m] * 1e6 / iters / wps, wps);
        }
    }
    return 0;
}
