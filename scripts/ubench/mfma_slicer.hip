// The certified 12-tap sum of K1s as an exact integer Toeplitz product on the matrix pipe -- structure and speed test.
//   y_c[n] = sum_q ct[q] * x[n - 24 + q]   (fir_slice.hip: dc = 24, NC = 12 for the reference table)
// taps as 24-bit integers tq = round(ct * S), three int8 digits t2 t1 t0; samples as two int8 digits (hs, l' = lo ^ 0x80):
//   y' = (A3 << 16) + (A2 << 8) + A1 + K,  A3 = sum t2 hs,  A2 = sum t2 l' + t1 hs,  A1 = sum t1 l' + t0 hs
// (the t0 l' term and the tap rounding are inside the bound).  v_mfma_i32_32x32x16_i8: 32 outputs x 32 channels x 16 window
// rows; a step of 32 outputs needs rows n0-24 .. n0+23 = three blocks of 16, of which the first is the previous step's last.
// A wave owns 64 channels (lane n, n+32 hold the channel pair 2n, 2n+1: one dword per row) and T outputs.
// Prints the launch time at C3's shape and checks the signs of the first channels against a double-precision sum.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_slicer.bin mfma_slicer.hip && ./mfma_slicer.bin [N L T]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
extern "C" __device__ int ld_b32(v4i, int, int, int) __asm("llvm.amdgcn.raw.buffer.load.i32");

constexpr int DC = 24, NC = 12;
#ifndef DROP
#define DROP 0     // speed experiments: 1 / 2 / 3 leave out 2 / 4 / 5 of the five products (wrong signs)
#endif
#ifndef NOPERM
#define NOPERM 0
#endif
#ifndef STAGE
#define STAGE 4
#endif
#ifndef PFD
#define PFD 2
#endif

struct Consts {
    long a[3][3][64];      // [block][digit 0..2][lane]: the Toeplitz slice's eight bytes of this lane
    int K;                 // 128 * sum tq / 256, rounded
    float eps;             // in units of y'
};

__device__ __forceinline__ void transpose8(const uint32_t *d, long &l_even, long &h_even, long &l_odd, long &h_odd)
{
    // d[j] = row j: bytes (l' even, hs even, l' odd, hs odd)  ->  per digit the eight rows' bytes
    uint32_t e[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t d0 = d[4 * g], d1 = d[4 * g + 1], d2 = d[4 * g + 2], d3 = d[4 * g + 3];
#if NOPERM        // speed experiment: the same number of cheap instructions instead of the eight v_perm_b32
#define PERM(a, b, sel) (((a) + (b)) ^ (sel))
#else
#define PERM(a, b, sel) __builtin_amdgcn_perm(a, b, sel)
#endif
        // v_perm_b32(a, b, sel): selector bytes 0-3 pick from b, 4-7 from a
        const uint32_t t0 = PERM(d1, d0, 0x05010400u);   // d0.b0 d1.b0 d0.b1 d1.b1
        const uint32_t t1 = PERM(d1, d0, 0x07030602u);   // d0.b2 d1.b2 d0.b3 d1.b3
        const uint32_t t2 = PERM(d3, d2, 0x05010400u);
        const uint32_t t3 = PERM(d3, d2, 0x07030602u);
        e[g][0] = PERM(t2, t0, 0x05040100u);             // b0 of rows 0..3
        e[g][1] = PERM(t2, t0, 0x07060302u);             // b1
        e[g][2] = PERM(t3, t1, 0x05040100u);             // b2
        e[g][3] = PERM(t3, t1, 0x07060302u);             // b3
    }
    l_even = (long) (((unsigned long) e[1][0] << 32) | e[0][0]);
    h_even = (long) (((unsigned long) e[1][1] << 32) | e[0][1]);
    l_odd = (long) (((unsigned long) e[1][2] << 32) | e[0][2]);
    h_odd = (long) (((unsigned long) e[1][3] << 32) | e[0][3]);
}

struct Blk { long l[2], h[2]; };    // [set]: even / odd channel of the pair

__device__ __forceinline__ uint32_t spread16(uint32_t p)     // nibbles n3 n2 n1 n0 -> 0 n3 0 n2 0 n1 0 n0
{
    p &= 0xffffu;
    p = (p | (p << 8)) & 0x00ff00ffu;
    return (p | (p << 4)) & 0x0f0f0f0fu;
}

template <int PF>
__global__ __launch_bounds__(64) void slicer(const int16_t *__restrict__ x, uint32_t *__restrict__ sgn, uint32_t *__restrict__ ambout,
                                              const Consts *__restrict__ cs, int N, int L, int T)
{
    const int lane = threadIdx.x, n = lane & 31, hh = lane >> 5;
    const int g = blockIdx.x, t0 = 32 + blockIdx.y * T;
    int t1 = t0 + T;
    if (t1 > L) t1 = L;
    if (t0 >= L) return;
    long A[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int dgt = 0; dgt < 3; ++dgt) A[b][dgt] = cs->a[b][dgt][lane];
    const int K = cs->K;
    const float eps = cs->eps;
    v4i rs;
    {
        const unsigned long p = (unsigned long) x;
        rs[0] = (int) (uint32_t) p;
        rs[1] = (int) (uint32_t) (p >> 32);
        rs[2] = (int) ((size_t) L * N * 2 > 0xffffffffu ? 0xffffffffu : (uint32_t) ((size_t) L * N * 2));
        rs[3] = 0x00020000;
    }
    const int voff = (g * 64 + 2 * n) * 2 + hh * 8 * N * 2;
    const int rowb = N * 2;
    auto load_block = [&](int r0, uint32_t *d) {      // rows r0 + 8 hh + j
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = (uint32_t) ld_b32(rs, voff, (r0 + j) * rowb, 0);
    };
    auto digits = [&](const uint32_t *d, Blk &o) {
        uint32_t f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = d[j] ^ 0x00800080u;
        transpose8(f, o.l[0], o.h[0], o.l[1], o.h[1]);
    };
    Blk B0, B1, B2;
    {
        uint32_t r[8];
        load_block(t0 - DC, r);
        digits(r, B0);
    }
    uint32_t wq[4] = {0, 0, 0, 0}, amball = 0;
    auto step = [&](int n0) {
        uint32_t word[2], ambw[2];
#if STAGE < 2                       // speed experiments: 0 = loads only, 1 = + digits, 2 = + MFMA, 3 = + flags, 4 = all
        {
            uint32_t acc = (uint32_t) B1.l[0] ^ (uint32_t) B2.h[1] ^ (uint32_t) (B1.h[0] >> 32) ^ (uint32_t) (B2.l[1] >> 32) ^ (uint32_t) B1.l[1] ^ (uint32_t) B2.l[0] ^ (uint32_t) B1.h[1] ^ (uint32_t) B2.h[0];
            amball ^= acc;
            B0 = B2;
            return;
        }
#endif
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            v16i a3 = {0}, a2 = {0}, a1;
#pragma unroll
            for (int v = 0; v < 16; ++v) a1[v] = K;
            const Blk *bl[3] = {&B0, &B1, &B2};
#pragma unroll
            for (int b = 0; b < 3; ++b) {
#if DROP < 3
                a3 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[b][2], bl[b]->h[s], a3, 0, 0, 0);
#else
                a3[b] += (int) bl[b]->h[s];
#endif
#if DROP < 2
                a2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[b][2], bl[b]->l[s], a2, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[b][1], bl[b]->h[s], a2, 0, 0, 0);
#else
                a2[b] += (int) bl[b]->l[s];
#endif
#if DROP < 1
                a1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[b][1], bl[b]->l[s], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[b][0], bl[b]->h[s], a1, 0, 0, 0);
#else
                a1[b] += (int) (bl[b]->l[s] >> 32);
#endif
            }
            uint32_t neg = 0, amb = 0;
#if STAGE < 3
            amball ^= (uint32_t) (a3[0] ^ a2[5] ^ a1[9] ^ a3[15] ^ a2[12] ^ a1[3]);
            word[s] = ambw[s] = 0;
            continue;
#endif
            // the asm below reads MFMA results: the compiler does not count the wait states for it
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" : "+v"(a3), "+v"(a2), "+v"(a1));
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                int t, y;              // ((a3 << 8) + a2 << 8) + a1: the compiler makes two shifts and an add3 of it
                asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(a3[v]), "v"(a2[v]));
                asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(y) : "v"(t), "v"(a1[v]));
                const float yf = (float) y;
                neg = __builtin_amdgcn_alignbit(neg, (uint32_t) y, 31);
                amb = __builtin_amdgcn_alignbit(amb, __float_as_uint(__builtin_fabsf(yf) - eps), 31);
            }
#if STAGE < 4
            amball ^= neg ^ amb;
            word[s] = ambw[s] = 0;
            continue;
#endif
            // this lane: outputs (v % 4) + 8 (v / 4) + 4 hh, v = 0 first = bit 15
            const uint32_t sn = spread16(~neg), sa = spread16(amb);
            const uint32_t pn = (uint32_t) __shfl_xor((int) sn, 32), pa = (uint32_t) __shfl_xor((int) sa, 32);
            word[s] = hh == 0 ? (sn << 4) | pn : (pn << 4) | sn;
            ambw[s] = hh == 0 ? (sa << 4) | pa : (pa << 4) | sa;
        }
        B0 = B2;
#if STAGE < 4
        return;
#endif
        // lane hh = 0 keeps the even channel's word, hh = 1 the odd one's
        const uint32_t w = hh == 0 ? word[0] : word[1];
        amball |= hh == 0 ? ambw[0] : ambw[1];
        const int wi = (n0 >> 5) & 3;
        if (wi == 0) wq[0] = w; else if (wi == 1) wq[1] = w; else if (wi == 2) wq[2] = w; else wq[3] = w;
        if (wi == 3 || n0 + 32 >= t1) {
            const int c = g * 64 + 2 * n + hh;
            uint32_t *dst = sgn + ((size_t) (n0 >> 7) * (size_t) N + (size_t) c) * 4;
            if (wi == 3) *reinterpret_cast<uint4 *>(dst) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
            else for (int q = 0; q <= wi; ++q) dst[q] = wq[q];
        }
    };
    uint32_t raw[PF][2][8];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        load_block(t0 + 32 * p - DC + 16, raw[p][0]);
        load_block(t0 + 32 * p - DC + 32, raw[p][1]);
    }
    for (int n0 = t0; n0 < t1; n0 += 32 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            {   // (no condition around the loads: the compiler's vmcnt bookkeeping gives up at a conditional load and
                // waits for everything; rows past the end read as zero through the descriptor)
#if STAGE >= 1
                digits(raw[p][0], B1);
                digits(raw[p][1], B2);
#else
                B1.l[0] = raw[p][0][0] ^ raw[p][0][1] ^ raw[p][0][2] ^ raw[p][0][3] ^ raw[p][0][4] ^ raw[p][0][5] ^ raw[p][0][6] ^ raw[p][0][7];
                B2.l[0] = raw[p][1][0] ^ raw[p][1][1] ^ raw[p][1][2] ^ raw[p][1][3] ^ raw[p][1][4] ^ raw[p][1][5] ^ raw[p][1][6] ^ raw[p][1][7];
                B1.h[0] = B2.h[0] = 0;
                B1.l[1] = B1.h[1] = B2.l[1] = B2.h[1] = 0;
#endif
                // the two new blocks of the step PF steps ahead
                load_block(n0 + 32 * (p + PF) - DC + 16, raw[p][0]);
                load_block(n0 + 32 * (p + PF) - DC + 32, raw[p][1]);
                step(n0 + 32 * p);
            }
        }
    }
    if (amball) atomicOr(ambout + (g * 64 + 2 * n + hh), amball);
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 16384, L = argc > 2 ? atoi(argv[2]) : 48000, T = argc > 3 ? atoi(argv[3]) : 512;
    // the reference table's twelve central taps (filter.c:57 coeffs, the middle of the 32 effective ones)
    const float ct[12] = {3.70199996e-06f, 2.23550000e-04f, 5.94480010e-03f, 6.96159974e-02f, 3.58990014e-01f, 8.15219998e-01f,
                          8.15219998e-01f, 3.58990014e-01f, 6.96159974e-02f, 5.94480010e-03f, 2.23550000e-04f, 3.70199996e-06f};
    double sabs = 0;
    for (float t : ct) sabs += fabs(t);
    const double S = 8388608.0 / sabs * 0.999;
    long tq[12], sumtq = 0;
    for (int q = 0; q < 12; ++q) { tq[q] = lround(ct[q] * S); sumtq += tq[q]; }
    Consts hc;
    for (int b = 0; b < 3; ++b)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31;
            unsigned long v[3] = {0, 0, 0};
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * (lane >> 5) + j, q = 16 * b + k - i;
                long t = (q >= 0 && q < 12) ? tq[q] : 0;
                // signed digits: t = 65536 t2 + 256 t1 + t0, each in [-128, 127]
                long t0 = ((t + 128) & 255) - 128; t = (t - t0) >> 8;
                long t1 = ((t + 128) & 255) - 128; t = (t - t1) >> 8;
                long t2 = t;
                v[0] |= (unsigned long) (uint8_t) (int8_t) t0 << (8 * j);
                v[1] |= (unsigned long) (uint8_t) (int8_t) t1 << (8 * j);
                v[2] |= (unsigned long) (uint8_t) (int8_t) t2 << (8 * j);
            }
            for (int dgt = 0; dgt < 3; ++dgt) hc.a[b][dgt][lane] = (long) v[dgt];
        }
    hc.K = (int) lround(128.0 * (double) sumtq / 256.0);
    hc.eps = (float) (0.4 * S / 256.0);
    std::vector<int16_t> hx((size_t) 4096 * N);
    uint32_t seed = 12345;
    for (auto &v : hx) { seed = seed * 1664525u + 1013904223u; v = (int16_t) ((int) (seed >> 16) - 32768) / 2; }
    int16_t *x; uint32_t *sgn, *amb; Consts *dc;
    CHECK(hipMalloc(&x, (size_t) N * L * 2));
    for (size_t off = 0; off < (size_t) L; off += 4096) {
        const size_t rows = off + 4096 <= (size_t) L ? 4096 : (size_t) L - off;
        CHECK(hipMemcpy(x + off * N, hx.data(), rows * N * 2, hipMemcpyHostToDevice));
    }
    const size_t words = (size_t) ((L + 127) / 128 + 1) * 4 * N;
    CHECK(hipMalloc(&sgn, words * 4)); CHECK(hipMemset(sgn, 0, words * 4));
    CHECK(hipMalloc(&amb, (size_t) N * 4)); CHECK(hipMemset(amb, 0, (size_t) N * 4));
    CHECK(hipMalloc(&dc, sizeof hc)); CHECK(hipMemcpy(dc, &hc, sizeof hc, hipMemcpyHostToDevice));
    const dim3 grid(N / 64, (L - 32 + T - 1) / T);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((slicer<PFD>), grid, dim3(64), 0, 0, x, sgn, amb, dc, N, L, T);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("N %d L %d T %d: %.3f ms per launch  (%.2f TB/s of input)\n", N, L, T, ms / 10, (double) N * L * 2 / (ms / 10 * 1e-3) / 1e12);
    }
    // signs of the first 64 channels, outputs 32 .. 4000, against the double sum
    std::vector<uint32_t> hs(words);
    CHECK(hipMemcpy(hs.data(), sgn, words * 4, hipMemcpyDeviceToHost));
    long bad = 0, checked = 0, small = 0;
    for (int c = 0; c < 64; ++c)
        for (int o = 32; o < 4000; ++o) {
            double y = 0;
            for (int q = 0; q < 12; ++q) y += (double) ct[q] * hx[(size_t) (o - DC + q) * N + c];
            if (fabs(y) < 1.0) { ++small; continue; }
            const uint32_t w = hs[((size_t) (o >> 7) * N + c) * 4 + ((o >> 5) & 3)];
            const int bit = (w >> (31 - (o & 31))) & 1;
            ++checked;
            if (bit != (y > 0)) { if (bad < 8) printf("mismatch c %d o %d y %.3f bit %d\n", c, o, y, bit); ++bad; }
        }
    printf("signs: %ld checked, %ld wrong, %ld skipped (|y| < 1)\n", checked, bad, small);
    return 0;
}
