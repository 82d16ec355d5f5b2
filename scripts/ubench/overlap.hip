// overlap.hip -- what happens to a 1-wave-per-SIMD dependent VALU chain (the PLL stage)
// when a VALU-saturating kernel (the FIR stage) runs beside it: issue competition,
// placement, or clock?  Each chain wave reports its core-clock / wall-clock ratio and HW_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned simd_slot()
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 15u) << 10) | (((hw >> 8) & 0xffu) << 2) | ((hw >> 4) & 3u);
}
__device__ __forceinline__ void chain_body(int bx, unsigned long long *rec, int iters, unsigned k, int prio, unsigned *busy)
{
    const unsigned slot = simd_slot();
    if (busy && threadIdx.x == 0) {
        __hip_atomic_store(busy + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(busy + 16000, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // epoch
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    unsigned P = threadIdx.x * 977u, O = 0, D = bx * 2654435761u;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        unsigned tm, um, kk;
#define STEP(sh) \
        "v_bfe_i32 %[tm], %[D], " #sh ", 1\n\tv_ashrrev_i32 %[um], 31, %[P]\n\t" \
        "v_addc_co_u32 %[O], vcc, %[O], %[O], vcc\n\tv_bfi_b32 %[kk], %[um], %[Km], %[Kp]\n\t" \
        "v_bfi_b32 %[kk], %[tm], %[kk], %[INC]\n\tv_add_co_u32 %[P], vcc, %[P], %[kk]\n\t"
        asm volatile(STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
                     STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
                     : [P] "+v"(P), [O] "+v"(O), [tm] "=&v"(tm), [um] "=&v"(um), [kk] "=&v"(kk)
                     : [D] "v"(D), [Kp] "v"(k + 7), [Km] "v"(k - 7), [INC] "v"(k) : "vcc");
        D = D * 5 + 1;
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        rec[bx * 4 + 0] = c1 - c0;
        rec[bx * 4 + 1] = w1 - w0;
        rec[bx * 4 + 2] = ((unsigned long long) xcc << 32) | hw;
        rec[bx * 4 + 3] = w0;
        if (busy) __hip_atomic_store(busy + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (P + O == 12345) rec[0] = 0;
}

// FIR-like: per "sample" 6 mul + 12 add on 12 rotating accumulators, ~100 us per wave
__device__ __forceinline__ void hog_body(int bx, float *out, int iters, float b, unsigned long long *hrec,
                                          unsigned *work, const unsigned *busy, int n_items, unsigned *stats, unsigned epoch)
{
    asm volatile("" ::: "v95");          // >= 96 VGPRs: 5 waves per SIMD like the FIR kernel, never all 8 slots
    unsigned long long w0 = wall_clock64();
    unsigned bid = bx;
    if (work) {
        const unsigned slot = simd_slot();
        if (epoch) {
            const unsigned long long tg = wall_clock64();
            while (__hip_atomic_load(busy + 16000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch &&
                   wall_clock64() - tg < 100000ull) {
                for (int q = 0; q < 3; ++q) __builtin_amdgcn_s_sleep(127);
            }
        }
        if (__hip_atomic_load(busy + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (threadIdx.x == 0) atomicAdd(stats + 0, 1u);
            do {
                if (__hip_atomic_load(work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned) n_items) {
                    if (threadIdx.x == 0) atomicAdd(stats + 1, 1u);
                    return;
                }
                for (int q = 0; q < 8; ++q) __builtin_amdgcn_s_sleep(127);
            } while (__hip_atomic_load(busy + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        unsigned idx = 0;
        if (threadIdx.x == 0) idx = atomicAdd(work, 1u);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= (unsigned) n_items) return;
        bid = idx;
        w0 = wall_clock64();
    }
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = b * q;
    float x = b + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 12; ++p) {
            x = x * 1.0001f;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float pr = (0.1f + q) * x;
                acc[(p + q) % 12] += pr;
                acc[(p + 11 - q) % 12] += pr;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int q = 0; q < 12; ++q) s += acc[q];
    out[bid * 64 + threadIdx.x] = s;
    if (hrec && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hrec[bid * 3 + 0] = w0;
        hrec[bid * 3 + 1] = wall_clock64();
        hrec[bid * 3 + 2] = ((unsigned long long) xcc << 32) | hw;
    }
}

__global__ __launch_bounds__(64) void chain(unsigned long long *rec, int iters, unsigned k, int prio, unsigned *busy)
{
    chain_body(blockIdx.x, rec, iters, k, prio, busy);
}
__global__ __launch_bounds__(64) void hog(float *out, int iters, float b, unsigned long long *hrec,
                                          unsigned *work, const unsigned *busy, int n_items, unsigned *stats, unsigned epoch)
{
    hog_body(blockIdx.x, out, iters, b, hrec, work, busy, n_items, stats, epoch);
}
// one launch = the chain of step r-1 (first n_chain workgroups) + the hog of step r
__global__ __launch_bounds__(64) void fused(unsigned long long *rec, int chain_iters, int n_chain, float *out, int hog_iters,
                                            unsigned long long *hrec, unsigned *work, unsigned *busy, int n_items,
                                            unsigned *stats, unsigned epoch)
{
    if ((int) blockIdx.x < n_chain) chain_body(blockIdx.x, rec, chain_iters, 0x33330000u, 1, busy);
    else hog_body(blockIdx.x - n_chain, out, hog_iters, 1.0f, hrec, work, busy, n_items, stats, epoch);
}

int main()
{
    unsigned long long *rec, *hrec; float *out; unsigned *work, *busy, *stats;
    const int R = 12;
    CK(hipMalloc(&rec, 16 * 256 * 4 * 8)); CK(hipMalloc(&out, 65536 * 64 * 4)); CK(hipMalloc(&hrec, 65536 * 3 * 8));
    CK(hipMalloc(&work, 64 * 4)); CK(hipMalloc(&busy, 16384 * 4)); CK(hipMalloc(&stats, 64 * 4));
    hipStream_t sa, sb; int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    hipEvent_t e0[R], e1[R], f0[R], f1[R], hd[R];
    for (int r = 0; r < R; ++r) {
        CK(hipEventCreate(&e0[r])); CK(hipEventCreate(&e1[r])); CK(hipEventCreate(&f0[r])); CK(hipEventCreate(&f1[r]));
        CK(hipEventCreateWithFlags(&hd[r], hipEventDisableTiming));
    }
    const int chain_iters = 3000;
    const int chain_lds = 81 * 1024;   // one chain wave per CU (160 KB LDS per CU)
    CK(hipFuncSetAttribute((const void *) chain, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds));
    const int hog_iters = 56, n_items = 20480;
    std::vector<unsigned long long> h(1024), hr((size_t) n_items * 3);
    for (int r = 0; r < R; ++r) {
        CK(hipEventRecord(f0[r], sa));
        hipLaunchKernelGGL(hog, dim3(n_items), dim3(64), 0, sa, out, hog_iters, 1.0f, hrec, nullptr, busy, n_items, stats, 0u);
        CK(hipEventRecord(f1[r], sa));
    }
    CK(hipDeviceSynchronize());
    { float mh; CK(hipEventElapsedTime(&mh, f0[R - 1], f1[R - 1])); printf("hog alone (static grid, back to back): %.3f ms\n", mh); }
    for (int mode = 0; mode < 5; ++mode) {       // 0 static grid, 1 claim (no park), 2 claim + park
        CK(hipMemset(work, 0, 64 * 4)); CK(hipMemset(busy, 0, 16384 * 4)); CK(hipMemset(stats, 0, 64 * 4));
        CK(hipMemset(hrec, 0, (size_t) n_items * 24));
        CK(hipDeviceSynchronize());
        if (mode == 4) {
            for (int r = 0; r <= R; ++r) {     // launch r: hog r (r < R) + chain r-1 (r > 0)
                const int rr = r < R ? r : R - 1;
                if (r < R) CK(hipEventRecord(f0[rr], sa));
                if (r == R - 0 && r > 0) CK(hipEventRecord(e0[R - 1], sa));
                hipLaunchKernelGGL(fused, dim3((r > 0 ? 256 : 0) + (r < R ? n_items + 2048 : 0)), dim3(64), 0, sa,
                                   rec + (r > 0 ? r - 1 : 0) * 1024, chain_iters, r > 0 ? 256 : 0, out, hog_iters, hrec,
                                   work + rr, busy, r < R ? n_items : 0, stats, r > 0 ? 256u * r : 0u);
                if (r < R) CK(hipEventRecord(f1[rr], sa));
                if (r == R) CK(hipEventRecord(e1[R - 1], sa));
            }
            CK(hipDeviceSynchronize());
            float total; CK(hipEventElapsedTime(&total, f0[0], e1[R - 1]));
            printf("mode 4 (fused launch): %.3f ms per step over %d steps\n", total / R, R);
            for (int r = R - 3; r < R; ++r) { float mh; CK(hipEventElapsedTime(&mh, f0[r], f1[r])); printf("   launch %d: %.3f ms\n", r, mh); }
            unsigned st[4]; CK(hipMemcpy(st, stats, 16, hipMemcpyDeviceToHost));
            printf("   parked %u (of which left without work %u) over all steps\n", st[0], st[1]);
        } else
        for (int r = 0; r < R; ++r) {
            CK(hipEventRecord(f0[r], sa));
            if (mode == 0)
                hipLaunchKernelGGL(hog, dim3(n_items), dim3(64), 0, sa, out, hog_iters, 1.0f, hrec, nullptr, busy, n_items, stats, 0u);
            else
                hipLaunchKernelGGL(hog, dim3(n_items + 2048), dim3(64), 0, sa, out, hog_iters, 1.0f, hrec, work + r,
                                   mode >= 2 ? busy : busy + 16383, n_items, stats, mode == 3 ? 256u * r : 0u);
            CK(hipEventRecord(f1[r], sa));
            CK(hipEventRecord(hd[r], sa));
            CK(hipStreamWaitEvent(sb, hd[r], 0));
            CK(hipEventRecord(e0[r], sb));
            hipLaunchKernelGGL(chain, dim3(256), dim3(64), chain_lds, sb, rec + r * 1024, chain_iters, 0x33330000u, 1, mode >= 2 ? busy : nullptr);
            CK(hipEventRecord(e1[r], sb));
        }
        CK(hipDeviceSynchronize());
        float total; CK(hipEventElapsedTime(&total, f0[0], e1[R - 1]));
        if (mode < 4) printf("mode %d: %.3f ms per step over %d steps\n", mode, total / R, R);
        if (mode < 4) for (int r = R - 3; r < R; ++r) {
            float mh, mc, gap = 0; CK(hipEventElapsedTime(&mh, f0[r], f1[r])); CK(hipEventElapsedTime(&mc, e0[r], e1[r]));
            if (r) CK(hipEventElapsedTime(&gap, f1[r - 1], f0[r]));
            printf("   step %d: hog %.3f ms (gap before %.3f), chain %.3f ms\n", r, mh, gap, mc);
        }
        unsigned st[4]; CK(hipMemcpy(st, stats, 16, hipMemcpyDeviceToHost));
        if (mode < 4) printf("   parked %u (of which left without work %u) over all steps\n", st[0], st[1]);
        for (int r = R - 4; r < R; ++r) {
            CK(hipMemcpy(h.data(), rec + r * 1024, 1024 * 8, hipMemcpyDeviceToHost));
            std::vector<double> us, st; std::vector<unsigned long long> cid;
            for (int i = 0; i < 256; ++i) { us.push_back(h[i * 4 + 1] / 100.0); st.push_back(h[i * 4 + 3] / 100.0); cid.push_back(h[i * 4 + 2] & 0xffffffff0000ff30ull); }
            std::sort(us.begin(), us.end()); std::sort(cid.begin(), cid.end()); std::sort(st.begin(), st.end());
            int dup = 0; for (int i = 1; i < 256; ++i) dup += cid[i] == cid[i - 1];
            printf("   chain %d: wave us min %.0f med %.0f max %.0f | start spread: med-min %.0f max-min %.0f us | waves sharing a SIMD %d\n",
                   r, us[0], us[128], us[255], st[128] - st[0], st[255] - st[0], dup);
        }
    }
    return 0;
}
