// Cycles per transition of the six-wave PLL's recurrence row (pll_nrzi.hip: PLL_STEP x 4 + the row's own
// bookkeeping) for one wave alone on its SIMD, and of variants that leave one ingredient out: what do the
// v_cmpx (EXEC narrowing), the SDWA forms and the LDS traffic cost?
//   hipcc --offload-arch=gfx950 -O3 -o pll_step6.bin pll_step6.hip && ./pll_step6.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CMPX(k) "v_cmpx_lt_i32 vcc, " #k ", %[rem]\n\t"
#define MULS(k) "v_mul_u32_u24_sdwa %[T], %[K7], %[E] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t"
#define MULP(k) "v_mul_u32_u24 %[T], %[K7], %[E]\n\t"
#define REC     "v_add_u32 %[U], %[X], %[T]\n\tv_bfe_i32 %[um], %[U], 22, 1\n\tv_xad_u32 %[X], %[Q], %[um], %[X]\n\t"
#define WS(k)   "v_lshrrev_b32_sdwa %[W], %[c23], %[U] dst_sel:BYTE_" #k " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
#define WP(k)   "v_lshrrev_b32 %[W], 23, %[U]\n\t"
// the nudge masked instead of the lane: Qv = rem > k ? Q : 0 (v_cmp + v_cndmask, off the chain), EXEC untouched
#define CND(k)  "v_cmp_lt_i32 vcc, " #k ", %[rem]\n\tv_cndmask_b32 %[Qv], 0, %[Qr], vcc\n\t"
#define RECV    "v_add_u32 %[U], %[X], %[T]\n\tv_bfe_i32 %[um], %[U], 22, 1\n\tv_xad_u32 %[X], %[Qv], %[um], %[X]\n\t"

#define STEP_A(k) CMPX(k) MULS(k) REC WS(k)
#define STEP_B(k) MULS(k) REC WS(k)
#define STEP_C(k) CMPX(k) MULS(k) REC WP(k)
#define STEP_D(k) CMPX(k) MULP(k) REC WS(k)
#define STEP_E(k) MULP(k) REC WP(k)
#define STEP_F(k) REC
#define STEP_G(k) CND(k) MULS(k) RECV WS(k)

#define ROW(S, LDS)                                                                                    \
    asm volatile("s_mov_b64 %[sv], exec\n\t"                                                           \
                 "v_or_b32 %[X], 0x7f, %[X]\n\t" S(0) S(1) S(2) S(3) "s_mov_b64 exec, %[sv]\n\t"        \
                 LDS                                                                                   \
                 "v_subrev_u32 %[rem], 4, %[rem]\n\t"                                                  \
                 : [X] "+v"(X), [U] "=&v"(U), [um] "=&v"(um), [T] "=&v"(T), [E] "+v"(E), [F] "+v"(F), [W] "+v"(W),    \
                   [ad] "+v"(ad), [rem] "+v"(rem), [sv] "=&s"(sv), [Qv] "=&v"(Qv)                         \
                 : [Q] "s"(Q), [K7] "v"(K7), [c23] "v"(c23), [Qr] "v"(Qr)                                 \
                 : "vcc", "memory")
#define LDS_ON  "ds_write_b32 %[ad], %[W]\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32 %[E], %[F]\n\tds_read_b32 %[F], %[ad] offset:8\n\t"
#define LDS_OFF "v_mov_b32 %[E], %[F]\n\t"

template <int V>
__global__ __launch_bounds__(64) void k(unsigned *out, unsigned long long *cyc, int iters)
{
    __shared__ unsigned strip[64 * 67];
    for (int i = threadIdx.x; i < 64 * 67; i += 64) strip[i] = (i * 2654435761u) & 0xfcfcfcfcu;
    __syncthreads();
    unsigned X = threadIdx.x * 12345u, U, um, T, E = 0x40302010u, F = 0x80706050u, W = 0, Qv;
    unsigned ad = threadIdx.x * 268u, c23 = 23;
    int rem = 1 << 30;
    const unsigned Q = 819u << 7, K7 = 13107u << 7, Qr = Q;
    unsigned long long sv;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(F) : "v"(ad) : "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) ROW(STEP_A, LDS_ON);
        if (V == 1) ROW(STEP_A, LDS_OFF);
        if (V == 2) ROW(STEP_B, LDS_OFF);
        if (V == 3) ROW(STEP_C, LDS_OFF);
        if (V == 4) ROW(STEP_D, LDS_OFF);
        if (V == 5) ROW(STEP_E, LDS_OFF);
        if (V == 6) ROW(STEP_F, LDS_OFF);
        if (V == 7) ROW(STEP_G, LDS_OFF);
        if (V == 8) ROW(STEP_G, LDS_ON);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = X + W + E + T + U + um;
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
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double) h[i];
    printf("%-64s %6.1f clock ticks per transition\n", name, s / blocks / iters / 4.0);
}

int main()
{
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, 4 * 64 * 1024);
    hipMalloc(&cyc, 8 * 1024);
    const int blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("A: the row as it is (cmpx, mul sdwa, 3 dep., lshr sdwa) + LDS", out, cyc, blocks);
        run<1>("A without the row's ds_write / ds_read", out, cyc, blocks);
        run<2>("B: A without v_cmpx", out, cyc, blocks);
        run<3>("C: A with a plain v_lshrrev instead of the sdwa byte write", out, cyc, blocks);
        run<4>("D: A with a plain v_mul_u32_u24", out, cyc, blocks);
        run<5>("E: no cmpx, no sdwa (5 VALU)", out, cyc, blocks);
        run<6>("F: the three dependent instructions alone", out, cyc, blocks);
        run<7>("G: v_cmp + v_cndmask on the nudge instead of v_cmpx (7 VALU)", out, cyc, blocks);
        run<8>("G + LDS", out, cyc, blocks);
    }
    return 0;
}
