// Clock ticks per transition of the PLL recurrence's row for ONE wave alone on its SIMD, as a function of the ORDER of
// its instructions: a lone wave issues an independent VALU instruction every ~4.5 ticks and a dependent one every ~9,
// so a row written step after step (chain, then the step's other work) leaves most issue slots behind a dependent
// instruction empty.  Positions come out of real LDS strips (sorted bytes), results go back to LDS as in the kernels.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pll_rows_sched.bin pll_rows_sched.hip && /tmp/pll_rows_sched.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define MUL(k)  "v_mul_u32_u24_sdwa %[T" #k "], %[K7], %[E] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t"
#define MULF(k) "v_mul_u32_u24_sdwa %[T" #k "], %[K7], %[F] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t"
#define CMPX(k) "v_cmpx_lt_i32 vcc, " #k ", %[rem]\n\t"
#define ADD(k)  "v_add_u32 %[U], %[X], %[T" #k "]\n\t"
#define BFE     "v_bfe_i32 %[um], %[U], 22, 1\n\t"
#define XAD     "v_xad_u32 %[X], %[Q], %[um], %[X]\n\t"
#define WS(k)   "v_lshrrev_b32_sdwa %[W], %[c23], %[U] dst_sel:BYTE_" #k " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"

// six-wave R as it is in pll_nrzi.hip: step after step
#define STEP6(k) CMPX(k) MUL(k) ADD(k) BFE XAD WS(k)
#define ROW6_PLAIN "v_or_b32 %[X], 0x7f, %[X]\n\t" STEP6(0) STEP6(1) STEP6(2) STEP6(3)
// the same instructions, chain instructions alternating with the others: add_k, (mul_k+1), bfe_k, (ws_k), xad_k, (cmpx_k+1)
#define ROW6_ILP                                           \
    "v_or_b32 %[X], 0x7f, %[X]\n\t"                        \
    CMPX(0) MUL(0)                                         \
    ADD(0) MUL(1) BFE WS(0) XAD CMPX(1)                    \
    ADD(1) MUL(2) BFE WS(1) XAD CMPX(2)                    \
    ADD(2) MUL(3) BFE WS(2) XAD CMPX(3)                    \
    ADD(3) "s_nop 0\n\t" BFE WS(3) XAD
// all four products first (they only need E), then the chain with the byte writes in its shadow
#define ROW6_ILP2                                          \
    "v_or_b32 %[X], 0x7f, %[X]\n\t"                        \
    MUL(0) MUL(1) MUL(2) MUL(3) CMPX(0)                    \
    ADD(0) BFE XAD CMPX(1) WS(0)                           \
    ADD(1) BFE XAD CMPX(2) WS(1)                           \
    ADD(2) BFE XAD CMPX(3) WS(2)                           \
    ADD(3) BFE XAD WS(3)

// fused row (three-wave form): the recurrence toggles its own bits
#define TG1     "v_lshrrev_b32 %[m], 23, %[U]\n\t"
#define TG2     "v_lshrrev_b32 %[a], 28, %[U]\n\t"
#define TG3     "v_lshlrev_b32_e64 %[m], %[m], 1\n\t"
#define TG4     "v_lshl_add_u32 %[a], %[a], 8, %[pb]\n\t"
#define TGX     "ds_xor_b32 %[a], %[m]\n\t"
#define STEP10(k) CMPX(k) MUL(k) ADD(k) BFE TG1 XAD TG2 TG3 TG4 TGX
#define ROW10_PLAIN "v_or_b32 %[X], 0x7f, %[X]\n\t" STEP10(0) STEP10(1) STEP10(2) STEP10(3)
// interleaved: the toggle of step k in the shadow of step k+1's chain; ds_xor_k before cmpx_k+1 (it needs EXEC_k),
// U alternates between two registers so that step k's toggle can still read it
#define ADDB(k) "v_add_u32 %[V], %[X], %[T" #k "]\n\t"
#define BFEB    "v_bfe_i32 %[um], %[V], 22, 1\n\t"
#define TG1B    "v_lshrrev_b32 %[m], 23, %[V]\n\t"
#define TG2B    "v_lshrrev_b32 %[a], 28, %[V]\n\t"
#define ROW10_ILP                                                          \
    "v_or_b32 %[X], 0x7f, %[X]\n\t"                                        \
    CMPX(0) MUL(0) MUL(1)                                                  \
    ADD(0) MUL(2) BFE MUL(3) XAD                                           \
    ADDB(1) TG1 TG2 BFEB TG3 TG4 TGX CMPX(1) XAD                           \
    ADD(2) TG1B TG2B BFE TG3 TG4 TGX CMPX(2) XAD                           \
    ADDB(3) TG1 TG2 BFEB TG3 TG4 TGX CMPX(3) XAD                           \
    TG1B TG2B TG3 TG4 TGX

template <int V>
__global__ __launch_bounds__(64) void k(unsigned *out, unsigned long long *cyc, int iters)
{
    __shared__ unsigned strip[64 * 67 + 17 * 64];
    for (int i = threadIdx.x; i < 64 * 67; i += 64) strip[i] = 0x30201000u + 0x40404040u * (i & 3) + 0x01010101u * (i % 13);
    for (int i = threadIdx.x; i < 17 * 64; i += 64) strip[64 * 67 + i] = 0;
    __syncthreads();
    unsigned X = threadIdx.x * 12345u, U = 0, Vv = 0, um = 0, T0 = 0, T1 = 0, T2 = 0, T3 = 0, E, F, W = 0, m = 0, a = 0;
    unsigned ad = threadIdx.x * 268u, c23 = 23, pb = 64 * 67 * 4 + threadIdx.x * 4;
    int rem = 1 << 30;
    const unsigned Q = 819u << 7, K7 = 13107u << 7;
    unsigned long long sv;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(E), "=v"(F) : "v"(ad) : "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define RUN(ROW, TAIL)                                                                                               \
        asm volatile("s_mov_b64 %[sv], exec\n\t" ROW "s_mov_b64 exec, %[sv]\n\t" TAIL                                    \
                     "v_subrev_u32 %[rem], 4, %[rem]\n\t"                                                               \
                     : [X] "+v"(X), [U] "+v"(U), [V] "+v"(Vv), [um] "+v"(um), [T0] "+v"(T0), [T1] "+v"(T1), [T2] "+v"(T2),   \
                       [T3] "+v"(T3), [E] "+v"(E), [F] "+v"(F), [W] "+v"(W), [m] "+v"(m), [a] "+v"(a), [ad] "+v"(ad),     \
                       [rem] "+v"(rem), [sv] "=&s"(sv)                                                                   \
                     : [Q] "s"(Q), [K7] "v"(K7), [c23] "v"(c23), [pb] "v"(pb)                                           \
                     : "vcc", "memory")
#define TAIL6  "ds_write_b32 %[ad], %[W]\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32 %[E], %[F]\n\tds_read_b32 %[F], %[ad] offset:8\n\t"
#define TAIL10 "s_waitcnt lgkmcnt(4)\n\tv_mov_b32 %[E], %[F]\n\tds_read_b32 %[F], %[ad] offset:8\n\t"
        if (V == 0) RUN(ROW6_PLAIN, TAIL6);
        if (V == 1) RUN(ROW6_ILP, TAIL6);
        if (V == 2) RUN(ROW6_ILP2, TAIL6);
        if (V == 3) RUN(ROW10_PLAIN, TAIL10);
        if (V == 4) RUN(ROW10_ILP, TAIL10);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = X + W + E + T0 + U + um + Vv + m + a + strip[64 * 67 + threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char *name, unsigned *out, unsigned long long *cyc, int blocks)
{
    const int iters = 20000;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    unsigned ho[64];
    hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost);
    unsigned chk = 0;
    for (int i = 0; i < 64; ++i) chk = chk * 31u + ho[i];
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double) h[i];
    printf("%-72s %6.1f clock ticks per transition   (check %08x)\n", name, s / blocks / iters / 4.0, chk);
}

int main()
{
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, 4 * 64 * 1024);
    hipMalloc(&cyc, 8 * 1024);
    const int blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("six-wave recurrence row (6 VALU per step), step after step", out, cyc, blocks);
        run<1>("  the same, chain and shadow instructions alternating", out, cyc, blocks);
        run<2>("  the same, products first, byte writes in the chain's shadow", out, cyc, blocks);
        run<3>("three-wave row (9 VALU + ds_xor per step), step after step", out, cyc, blocks);
        run<4>("  the same, step k's toggle in the shadow of step k+1's chain", out, cyc, blocks);
    }
    return 0;
}
