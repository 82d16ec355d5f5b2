// fir_sign_wide.hip -- K1s with CPL adjacent channels per lane (gfx950).
//
// Same contract as fir_sign_kernel<32, 12, 32> in fir_slice.hip (the sign-exact slicer that stands
// in for filter_run_buf() + the `out > 0` test of receiver_run(); gnuais src/filter.c:106-143,
// src/receiver.c:109-111,126): bit-identical sign words, peak, history carry.  What changes is the
// shape of the memory requests.  With one channel per lane a wave's load of one sample row is 64
// int16 = ONE 128-byte request, the smallest the vector memory pipeline issues, and the kernel
// stood at 4.6 TB/s with every other limiter ruled out.  Here a lane owns CPL = 2 (or 4) ADJACENT
// channels: one typed buffer load (16_16 / 16_16_16_16 SSCALED) brings the lane's 4 / 8 bytes of a
// row as 2 / 4 floats, a wave covers 128 / 256 channels, and a request is 256 / 512 bytes -- half /
// a quarter of the vector-memory instructions for the same bytes.  The arithmetic per channel is
// untouched (12 central taps in direct form, symmetric pre-adds, fused accumulation, |y_c| > eps
// certifies the sign, exact ordered 32-tap sum otherwise), the two channels' outputs are computed
// side by side so that a loaded row dies for both at once.
//
// Sign words keep sgn_index(): the four words of 128 outputs of one channel are 16 bytes, the CPL
// channels of a lane are adjacent, so a lane stores CPL x 16 contiguous bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <utility>
#include "kernels.h"

namespace gnuais {

namespace {

#ifndef WIDE_FENCE
#define WIDE_FENCE 1
#endif
constexpr int NC = 12;      // central taps
constexpr int NES = 32;     // effective taps of the reference table
struct WTaps { float te[NES]; };

typedef int w_v4i __attribute__((ext_vector_type(4)));
typedef float w_v2f __attribute__((ext_vector_type(2)));
typedef float w_v4f __attribute__((ext_vector_type(4)));
// clang has no builtin for llvm.amdgcn.raw.buffer.load.format; the intrinsics are reached by name
extern "C" __device__ float w_load_format_f32(w_v4i rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.format.f32");
extern "C" __device__ w_v2f w_load_format_v2f32(w_v4i rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.format.v2f32");
extern "C" __device__ w_v4f w_load_format_v4f32(w_v4i rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");

__device__ __forceinline__ int w_load_sample(const int16_t *__restrict__ x, const int16_t *__restrict__ hist,
                                             int m, int N, int NT, int c)
{
    const int16_t *p = (m >= 0) ? (x + (size_t) m * (size_t) N + c) : (hist + (size_t) (NT + m) * (size_t) N + c);
    return (int) *p;
}

template <int I> struct WIc { static constexpr int value = I; };
template <class F, int... Is>
__device__ __forceinline__ bool w_expand(F &&f, std::integer_sequence<int, Is...>) { return (f(WIc<Is>{}) && ...); }

template <int CPL> __device__ __forceinline__ void w_touch(uint32_t *a, uint32_t *b);
template <> __device__ __forceinline__ void w_touch<2>(uint32_t *a, uint32_t *b)
{
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
}
template <> __device__ __forceinline__ void w_touch<4>(uint32_t *a, uint32_t *b)
{
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}

template <int CPL> struct RowVec;
template <> struct RowVec<2> {
    typedef w_v2f type;
    static __device__ __forceinline__ type load(w_v4i r, int v, int s) { return w_load_format_v2f32(r, v, s, 0); }
    static constexpr int word3 = 4 | (5 << 3) | (3 << 12) | (5 << 15);               // R,G | SSCALED | 16_16
};
template <> struct RowVec<4> {
    typedef w_v4f type;
    static __device__ __forceinline__ type load(w_v4i r, int v, int s) { return w_load_format_v4f32(r, v, s, 0); }
    static constexpr int word3 = 4 | (5 << 3) | (6 << 6) | (7 << 9) | (3 << 12) | (12 << 15);   // RGBA | SSCALED | 16_16_16_16
};

// Raw (untyped) row loads: the lane's CPL int16 stay packed in CPL / 2 registers while they are in flight
// -- half the registers per byte of a typed load that delivers floats -- and are converted with one SDWA
// v_cvt_f32_i32 per sample (sign-extending word select) when their group comes up.
__device__ __forceinline__ float w_cvt_lo(uint32_t q)
{
    float r;
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(q));
    return r;
}
__device__ __forceinline__ float w_cvt_hi(uint32_t q)
{
    float r;
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(q));
    return r;
}
typedef uint32_t w_v2u __attribute__((ext_vector_type(2)));
template <int CPL> struct RawRow;
template <> struct RawRow<2> {
    typedef uint32_t type;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int v, int s) { return __builtin_amdgcn_raw_buffer_load_b32(r, v, s, 0); }
    static __device__ __forceinline__ w_v2f unpack(type q) { w_v2f f; f[0] = w_cvt_lo(q); f[1] = w_cvt_hi(q); return f; }
};
template <> struct RawRow<4> {
    typedef w_v2u type;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int v, int s) { return __builtin_amdgcn_raw_buffer_load_b64(r, v, s, 0); }
    static __device__ __forceinline__ w_v4f unpack(type q)
    {
        w_v4f f; f[0] = w_cvt_lo(q[0]); f[1] = w_cvt_hi(q[0]); f[2] = w_cvt_lo(q[1]); f[3] = w_cvt_hi(q[1]); return f;
    }
};

// CPL channels per lane; PK: the central sum on <CPL x float> vectors (v_pk_add_f32 / v_pk_fma_f32);
// G: rows per group (a group's loads are issued together, a scheduling fence follows every group, so at most
// NC - 1 + G rows -- + G with PF = 1, the next group's loads issued before this group's arithmetic -- are alive).
// PF = 2: raw loads, two groups ahead: while group g is worked on, the loads of groups g+1 and g+2 are in
// flight (16..32 rows x 64 lanes x 2 CPL bytes per wave), held packed (CPL / 2 registers per row).  The memory
// system answers in ~3.4 us when it is saturated (80 KB in flight per CU at 6 TB/s): what a wave has in flight
// only while it waits is not enough to keep it saturated once there is arithmetic between the loads.
template <int CPL, bool PK, int G, int PF, int DBG = 0, int FENCE = WIDE_FENCE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((CPL == 2 && PF == 2) ? 4 : 1))) void fir_sign_wide_kernel(
    const int16_t *__restrict__ x, const int16_t *__restrict__ hist,
    uint32_t *__restrict__ sgn, int *__restrict__ maxval,
    int16_t *__restrict__ hist_out, int *__restrict__ maxval_next,
    int N, int L, int T, int d, int NTaps, float eps_up, int map, WTaps taps, unsigned long long *stamps, int n_big, int T2)
{
    typedef typename RowVec<CPL>::type rowv;
    const int wave_id = (int) (blockIdx.y * gridDim.x + blockIdx.x);
    if (stamps && threadIdx.x == 0) stamps[2 * wave_id] = wall_clock64();
    static_assert(32 % G == 0 && G >= NC - 1 || G == 8, "groups tile a word");
    constexpr int J0 = (NES - NC) / 2;
    constexpr int GPW = 32 / G;                 // groups per sign word
    const int lane = threadIdx.x;
    int bx = (int) blockIdx.x, by = (int) blockIdx.y;
    if (map == 1) {                             // XCD-contiguous channel groups, as in fir_sign_kernel
        const int Gx = (int) gridDim.x, id = by * Gx + bx, per = Gx >> 3;
        bx = (id & 7) * per + (id >> 3) % per;
        by = (id >> 3) / per;
    }
    const int cg = (bx * 64 + lane) * CPL;      // the lane's first channel; N % CPL == 0: all live or none
    const bool live = cg < N;
    const int c = live ? cg : N - CPL;
    const int t0 = by < n_big ? by * T : n_big * T + (by - n_big) * T2;     // two segment lengths: see fir_sign_kernel
    const int t1e = t0 + (by < n_big ? T : T2);
    const int t1 = t1e < L ? t1e : L;
    if (t0 >= L) return;
    const int dc = d - J0;                      // y_c[n] = sum_q tc[q] * x[n - dc + q]
    const int m0 = t0 - dc;
    auto ctap = [&](int q) -> float { return taps.te[J0 + q]; };

    // Descriptors: the rows of this segment (from row0 on: far less than the 4 GB a descriptor spans) and the
    // history rows, each as a CPL-channel row format and as a single-channel one (exact re-evaluation).  The
    // lane's byte offset goes into the vector offset, the row into the scalar offset: no 64-bit address
    // arithmetic anywhere, also not on the rare paths (their address registers would otherwise set the
    // kernel's register count).
    const int row0 = m0 - J0 > 0 ? m0 - J0 : 0;   // oldest row anything of this segment reads from x
    const uint32_t rowbytes = (uint32_t) N * 2u;
    const unsigned long long span = (unsigned long long) (L - row0) * rowbytes;
    const unsigned long long xbase = (unsigned long long) (x + (size_t) row0 * (size_t) N);
    const unsigned long long hbase = (unsigned long long) hist;
    const int xspan = (int) (span > 0xffffffffull ? 0xffffffffull : span);
    const int hspan = (int) ((uint32_t) NTaps * rowbytes);
    const w_v4i rsrc = {(int) (xbase & 0xffffffffull), (int) ((xbase >> 32) & 0xffffull), xspan, RowVec<CPL>::word3};
    const w_v4i rsrc_h = {(int) (hbase & 0xffffffffull), (int) ((hbase >> 32) & 0xffffull), hspan, RowVec<CPL>::word3};
    const w_v4i rs1 = {rsrc[0], rsrc[1], xspan, 0x13004};       // R | SSCALED | 16
    const w_v4i rs1_h = {rsrc_h[0], rsrc_h[1], hspan, 0x13004};
    const int coff = c * 2;
    // row m (>= -NTaps) of the lane's channels as floats; rows past the call are clamped (whatever needs them
    // is masked), rows before it come from the history
    auto load_row = [&](int m) -> rowv {            // m is wave-uniform: scalar selects, no branch
        const bool h = m < 0;
        const int mm = m < L ? m : L - 1;
        const w_v4i r = {h ? rsrc_h[0] : rsrc[0], h ? rsrc_h[1] : rsrc[1], h ? hspan : xspan, RowVec<CPL>::word3};
        return RowVec<CPL>::load(r, coff, (int) ((uint32_t) (h ? NTaps + m : mm - row0) * rowbytes));
    };
    // exact value of output n (per lane) of channel c + j: filter.h:40-49 order; eight loads in flight at a
    // time.  The window's rows go into the vector offset (n differs from lane to lane).
    auto exact_positive = [&](int n, int j) -> bool {
        float sum = 0.0f;
        const int mw = n - d;                       // the window's oldest row: >= m0 - J0 >= row0 when it is >= 0
        if (mw >= 0) {
            const int voff = coff + 2 * j + (int) ((uint32_t) (mw - row0) * rowbytes);
#pragma unroll
            for (int k0 = 0; k0 < NES; k0 += 8) {
                float xs[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) xs[k] = w_load_format_f32(rs1, voff + (int) ((uint32_t) (k0 + k) * rowbytes), 0, 0);
#pragma unroll
                for (int k = 0; k < 8; ++k) sum = sum + taps.te[k0 + k] * xs[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {                                    // the call's first outputs: part of the window is history
#pragma unroll
            for (int k0 = 0; k0 < NES; k0 += 8) {
                float xs[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) xs[k] = (float) w_load_sample(x, hist, mw + k0 + k, N, NTaps, c + j);
#pragma unroll
                for (int k = 0; k < 8; ++k) sum = sum + taps.te[k0 + k] * xs[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return sum > 0.0f;
    };
    auto all_zero = [&](int m_first, int count, int j) -> bool {
        uint32_t o = 0;
        for (int i = 0; i < count; ++i) {
            int m = m_first + i;
            m = m < -NTaps ? -NTaps : (m > L - 1 ? L - 1 : m);
            o |= (uint32_t) w_load_sample(x, hist, m, N, NTaps, c + j);
        }
        return o == 0;
    };

    // rows mbg .. mbg+G-1
    auto load_group = [&](int mbg, rowv *dst) {
        if (mbg >= 0 && mbg + G - 1 < L) {
#pragma unroll
            for (int p = 0; p < G; ++p)
                dst[p] = RowVec<CPL>::load(rsrc, coff, (int) ((uint32_t) (mbg - row0 + p) * rowbytes));
        } else {
#pragma unroll
            for (int p = 0; p < G; ++p) dst[p] = load_row(mbg + p);
        }
    };

    typedef typename RawRow<CPL>::type pkv;
    const int16_t *xrow0 = x + (size_t) row0 * (size_t) N;
    auto load_group_raw = [&](int mbg, pkv *dst) {
        if (mbg >= 0 && mbg + G - 1 < L) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t *>(xrow0), 0, xspan, 0x00020000);
#pragma unroll
            for (int p = 0; p < G; ++p) dst[p] = RawRow<CPL>::load(rr, coff, (int) ((uint32_t) (mbg - row0 + p) * rowbytes));
        } else {
#pragma unroll
            for (int p = 0; p < G; ++p) {
                const int m = mbg + p;
                const bool h = m < 0;
                const int mm = m < L ? m : L - 1;
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<int16_t *>(h ? hist : xrow0), 0, h ? hspan : xspan, 0x00020000);
                dst[p] = RawRow<CPL>::load(rr, coff, (int) ((uint32_t) (h ? NTaps + m : mm - row0) * rowbytes));
            }
        }
    };

    rowv tail[NC - 1];                          // the NC-1 rows before the current group
#pragma unroll
    for (int i = 0; i < NC - 1; ++i) tail[i] = load_row(m0 + i);

    bool zprev_known[CPL], zprev[CPL];
    int peakbits[CPL];
    uint32_t neg[CPL], amb[CPL], zor[CPL];
    uint32_t wq[CPL][4];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        zprev_known[j] = false; zprev[j] = false; peakbits[j] = 0;
        neg[j] = amb[j] = zor[j] = 0u;
        wq[j][0] = wq[j][1] = wq[j][2] = wq[j][3] = 0u;
    }

    const int nwords = (t1 - t0 + 31) / 32;
    const int ngroups = nwords * GPW;
    const int nblk = (nwords + 3) / 4;
    rowv cur[G], nxt[G];
    pkv pkd[2][G];                              // PF = 2: the packed rows of the next two groups
    if constexpr (PF == 1) load_group(m0 + NC - 1, cur);
    if constexpr (PF == 2) {
        load_group_raw(m0 + NC - 1, pkd[0]);
        load_group_raw(m0 + NC - 1 + G, pkd[1]);
    }
    // One loop turn = four sign words = 4 * GPW groups, written out by template expansion (the group index
    // within the turn is a compile-time constant: the registers of the window rotate by renaming).  hipcc's
    // unroller refuses a body of this size, so the expansion does not go through `#pragma unroll`.
    for (int b = 0; b < nblk; ++b) {
        auto group = [&](auto GI) -> bool {
            constexpr int gi = decltype(GI)::value;
            const int g = b * 4 * GPW + gi;             // group of the segment: outputs g*G .. g*G+G-1
            if (g >= ngroups) return false;
            const int mbg = m0 + NC - 1 + g * G;        // the group's first row
            if constexpr (PF == 2) {
                constexpr int h = gi & 1;               // 4 * GPW is even: the parity of g is the parity of gi
#pragma unroll
                for (int p = 0; p < G; ++p) cur[p] = RawRow<CPL>::unpack(pkd[h][p]);
                if (g + 2 < ngroups) load_group_raw(mbg + 2 * G, pkd[h]);
            } else if constexpr (PF == 1) {
                if (g + 1 < ngroups) load_group(mbg + G, nxt);
            } else {
                load_group(mbg, cur);
            }
            // filter.c:118-119 peak on the float bit patterns (see fir_sign_kernel)
            const bool interior = mbg >= 0 && mbg + G - 1 < L;
#pragma unroll
            for (int j = 0; j < CPL && !(DBG & 8); ++j) {
                int bp = 0;
                if (interior) {
#pragma unroll
                    for (int p = 0; p < G; ++p) bp = __float_as_int(cur[p][j]) > bp ? __float_as_int(cur[p][j]) : bp;
                } else {
#pragma unroll
                    for (int p = 0; p < G; ++p) {
                        const int m = mbg + p;
                        const int v = (m >= 0 && m < L) ? __float_as_int(cur[p][j]) : 0;
                        bp = v > bp ? v : bp;
                    }
                }
                peakbits[j] = bp > peakbits[j] ? bp : peakbits[j];
            }
#pragma unroll
            for (int p = 0; p < G; ++p) {
                auto xb = [&](int k) -> rowv {      // the row k steps before the newest
                    return p - k >= 0 ? cur[p - k >= 0 ? p - k : 0] : tail[p - k >= 0 ? 0 : NC - 1 + p - k];
                };
                rowv y;
                if constexpr (DBG & 4) {
                    y = xb(0) + xb(NC - 1);
                } else if constexpr (PK) {
                    rowv t0v;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) t0v[j] = ctap(0);
                    y = t0v * (xb(0) + xb(NC - 1));
#pragma unroll
                    for (int q = 1; q < NC / 2; ++q) {
                        rowv tq;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) tq[j] = ctap(q);
                        y = __builtin_elementwise_fma(tq, xb(q) + xb(NC - 1 - q), y);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        float yy = ctap(0) * (xb(0)[j] + xb(NC - 1)[j]);
#pragma unroll
                        for (int q = 1; q < NC / 2; ++q) yy = __builtin_fmaf(ctap(q), xb(q)[j] + xb(NC - 1 - q)[j], yy);
                        y[j] = yy;
                    }
                }
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    neg[j] = __builtin_amdgcn_alignbit(neg[j], __float_as_uint(y[j]), 31);
                    amb[j] = __builtin_amdgcn_alignbit(amb[j], __float_as_uint(__builtin_fabsf(y[j]) - eps_up), 31);
                }
                // hipcc's scheduler otherwise hoists the pre-adds of every output of the group (one live
                // temporary each): a fence after every FENCE-th output keeps an output's work together
                if (p % FENCE == FENCE - 1) {
                    w_touch<CPL>(neg, amb);             // both channels' flags of this output now, not after the group
                    if (p != G - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            // zor = OR of the word's samples for the silence test, which only looks at it when the word has
            // >= 8 ambiguous samples; a part of a word without any cannot belong to a silent word
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                if (amb[j] != 0) {
#pragma unroll
                    for (int p = 0; p < G; ++p) zor[j] |= __float_as_uint(cur[p][j]);
                } else {
                    zor[j] |= 1u;
                }
            }
            // the window moves on
            if constexpr (G >= NC - 1) {
#pragma unroll
                for (int k = 0; k < NC - 1; ++k) tail[k] = cur[G - (NC - 1) + k];
            } else {
#pragma unroll
                for (int k = 0; k < NC - 1 - G; ++k) tail[k] = tail[k + G];
#pragma unroll
                for (int k = 0; k < G; ++k) tail[NC - 1 - G + k] = cur[k];
            }

            if ((gi + 1) % GPW == 0) {                  // a sign word is complete
                const int obase = (g / GPW) * 32;
                const int mb = m0 + NC - 1 + obase;
                const int valid = t1 - (t0 + obase);
                const int slot = (gi / GPW) & 3;
                const bool last = t0 + obase + 32 >= t1;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    uint32_t w = ~neg[j];
                    uint32_t a = amb[j];
                    if (valid < 32) {
                        w &= ~0u << (32 - valid);
                        a &= ~0u << (32 - valid);
                    }
                    bool zc_known = false, zc = false;
                    if constexpr (DBG & 1) a = 0;
                    if (__popc(a) >= 8) {                       // a silent stretch?  (see fir_sign_kernel)
                        zc = zor[j] == 0;
                        zc_known = true;
                        constexpr int before = J0 + NC - 1;
                        if (zc && (zprev_known[j] ? zprev[j] : all_zero(mb - before, before, j)) &&
                            all_zero(mb + 32, J0, j)) {
                            w &= ~a;
                            a = 0;
                        }
                    }
                    zprev_known[j] = zc_known;
                    zprev[j] = zc;
                    while (a) {
                        const int pos = __clz((int) a);
                        const uint32_t bit = 0x80000000u >> pos;
                        a &= ~bit;
                        if (exact_positive(t0 + obase + pos, j)) w |= bit; else w &= ~bit;
                    }
                    if (slot == 0) wq[j][0] = w; else if (slot == 1) wq[j][1] = w; else if (slot == 2) wq[j][2] = w; else wq[j][3] = w;
                    neg[j] = amb[j] = zor[j] = 0u;
                }
                if (live && (slot == 3 || last) && (!(DBG & 2) || wq[0][0] == 0x12345u)) {
                    uint32_t *dst = sgn + sgn_index(((t0 + obase) >> 5) - slot, N, cg);   // channel cg + j: + 4 * j words
                    if (slot == 3) {
#pragma unroll
                        for (int j = 0; j < CPL; ++j)
                            reinterpret_cast<uint4 *>(dst)[j] = make_uint4(wq[j][0], wq[j][1], wq[j][2], wq[j][3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < CPL; ++j) {
                            dst[4 * j] = wq[j][0];
                            if (slot >= 1) dst[4 * j + 1] = wq[j][1];
                            if (slot >= 2) dst[4 * j + 2] = wq[j][2];
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF == 1) {
#pragma unroll
                for (int p = 0; p < G; ++p) cur[p] = nxt[p];
            }
            return true;
        };
        if (!w_expand(group, std::make_integer_sequence<int, 4 * GPW>{})) break;
    }

    if constexpr (DBG & 16) {
        if (peakbits[0] == 0x7fffffff) maxval[cg] = 1;
        return;
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        int peak = (int) __int_as_float(peakbits[j]);
        if (t1 == L) {                              // the last dc-NC+1 samples of the call
            const int shift = dc - NC + 1;
            for (int n = (L - shift > 0 ? L - shift : 0); n < L; ++n) {
                const int v = (int) x[(size_t) n * (size_t) N + c + j];
                peak = v > peak ? v : peak;
            }
        }
        if (live && peak > 0) atomicMax(&maxval[cg + j], peak);
    }
    if (t1 == L && live) {                          // carry for the next call (filter.c:129-134 restated)
        for (int k = 0; k < NTaps; ++k) {
            const int m = L - NTaps + k;
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                hist_out[(size_t) k * (size_t) N + cg + j] =
                    (m >= 0) ? x[(size_t) m * (size_t) N + cg + j] : hist[(size_t) (NTaps + m) * (size_t) N + cg + j];
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) maxval_next[cg + j] = 0;
    }
    if (stamps && threadIdx.x == 0) stamps[2 * wave_id + 1] = wall_clock64();
}

} // namespace

// cpl: 2 or 4 channels per lane (N % cpl == 0, the input 2 * cpl-byte aligned); form: bit 0 packed fp32
// central sum, bits 1-2 prefetch (1: typed, one group ahead; 2: raw, two groups ahead), bits 4.. rows per group (8 or 16; 0 = 16)
hipError_t launch_fir_sign_wide(const FirLaunch &a, int cpl, int form, hipStream_t stream)
{
    if (a.dump || a.T % 128 || a.NC != 12 || a.NE != 32 || (cpl != 2 && cpl != 4) || a.N % cpl ||
        ((uintptr_t) a.x & (uintptr_t) (2 * cpl - 1)))
        return hipErrorInvalidValue;
    dim3 grid((a.N / cpl + 63) / 64, (a.L + a.T - 1) / a.T), block(64);
    int n_big = 1 << 30, T2 = a.T;
    if (a.T2 > 0 && a.T2 < a.T && a.T2 % 128 == 0 && a.n_big < (int) grid.y) {
        n_big = a.n_big > 0 ? a.n_big : 0;
        T2 = a.T2;
        grid.y = n_big + (a.L - n_big * a.T + T2 - 1) / T2;
    }
    const float eps_up = __builtin_nextafterf(a.eps, INFINITY);
    const int map = (a.map == 1 && grid.x % 8 == 0) ? 1 : 0;
    WTaps t;
    for (int j = 0; j < 32; ++j) t.te[j] = a.te[j];
    const int pk = form & 1, pf = (form >> 1) & 3, g = (form >> 4) ? (form >> 4) : 16;
#define WIDE_LAUNCH(C, P, GG, F)                                                                                   \
    hipLaunchKernelGGL((fir_sign_wide_kernel<C, P, GG, F>), grid, block, a.lds_pad, stream, a.x, a.hist, a.sgn, a.maxval, \
                       a.hist_out, a.maxval_next, a.N, a.L, a.T, a.d, a.NT, eps_up, map, t, a.stamps, n_big, T2)
#define WIDE_PICK(C, GG)                                                  \
    do {                                                                  \
        if (pf == 2)      { if (pk) WIDE_LAUNCH(C, true, GG, 2); else WIDE_LAUNCH(C, false, GG, 2); }   \
        else if (pf == 1) { if (pk) WIDE_LAUNCH(C, true, GG, 1); else WIDE_LAUNCH(C, false, GG, 1); }   \
        else              { if (pk) WIDE_LAUNCH(C, true, GG, 0); else WIDE_LAUNCH(C, false, GG, 0); }   \
    } while (0)
#ifdef WIDE_DEBUG_FORMS
    // elimination experiments (results are wrong): a.dbg bit 0 no exact path, 1 no sign stores, 2 no central sum,
    // 3 no peak, 4 no epilogue
#define WIDE_DBG(D) case D: hipLaunchKernelGGL((fir_sign_wide_kernel<2, true, 16, 2, D>), grid, block, a.lds_pad, stream, a.x, a.hist, a.sgn, \
                                               a.maxval, a.hist_out, a.maxval_next, a.N, a.L, a.T, a.d, a.NT, eps_up, map, t, a.stamps, n_big, T2); return hipGetLastError();
    switch (a.dbg) { WIDE_DBG(1) WIDE_DBG(2) WIDE_DBG(4) WIDE_DBG(8) WIDE_DBG(16) WIDE_DBG(5) WIDE_DBG(31) WIDE_DBG(27) default: break; }
#undef WIDE_DBG
#endif
    if (cpl == 2) {
        if (g == 8) WIDE_PICK(2, 8); else WIDE_PICK(2, 16);
    } else {                                    // four channels per lane: too many registers for the typed prefetch
        if (pf == 2)  { if (pk) WIDE_LAUNCH(4, true, 8, 2); else WIDE_LAUNCH(4, false, 8, 2); }
        else          { if (pk) WIDE_LAUNCH(4, true, 8, 0); else WIDE_LAUNCH(4, false, 8, 0); }
    }
#undef WIDE_PICK
#undef WIDE_LAUNCH
    return hipGetLastError();
}

} // namespace gnuais
