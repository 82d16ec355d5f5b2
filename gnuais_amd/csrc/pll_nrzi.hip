// pll_nrzi.hip -- K2, six-wave form: bit-clock recovery PLL, slice and NRZI decode for gfx950 (and the
// choice between this form and the three-wave one of pll_nrzi3.hip, launch_pll() at the end).
//
// Stands in for the per-sample loop of receiver_run(), gnuais src/receiver.c:109-135, for a whole
// batch of channels.
//
// The reference touches the phase on every sample, but only a sign change of the filter output
// (a "transition", receiver.c:113) makes it do anything that is not linear:
//
//     transition at sample t :  pll += (pll < 0x8000) ? +pllinc/16 : -pllinc/16     receiver.c:114-117
//     every sample           :  pll += pllinc;  overflow -> slice, pll &= 0xffff    receiver.c:122-133
//
// Write the phase without the `& 0xffff`: U(t) = pll0 + t * pllinc + K(t), K = the nudges so far.
// U only grows (pllinc > pllinc/16), a nudge never crosses a multiple of 2^16 (pll < 0x8000 -> +q
// stays below 0x10000, pll >= 0x8000 -> -q stays above 0), and pllinc + q < 2^16, so the slices are
// exactly the times U crosses a multiple of 2^16: slice number m happens at the first sample whose
// increment takes U to (m + 1) * 2^16 or beyond, and floor(U / 2^16) slices have happened before
// sample t.  The nudge at a transition needs U mod 2^16 there, nothing else.
//
// The bit the reference emits at a slice is 1 if the level (sign of the filter output) is the same
// as at the previous slice, 0 if it differs (receiver.c:126-132), i.e. NOT the parity of the
// transitions since the previous slice.  A transition at sample t (the level seen by a slice AT t is
// already the new one) therefore toggles exactly one bit of the output: number floor(U(t) / 2^16).
// So:
//     bits = ~( XOR over the transitions of  1 << floor(U(t_j) / 2^16) )
//
// That turns 48 000 dependent steps per channel and call into ~10 000 (one per transition; the max
// over the 64 channels of a wave, re-synchronised every 256 samples).  One kernel, one workgroup
// per 64 channels, five waves that hand work to each other through LDS (pll_kernel below): a
// scanner turns sign words into lists of transition positions, the recurrence walks them and leaves
// the number of the bit each transition toggles, two togglers apply those, a writer takes the
// finished bit packs to HBM.
//
// Output: one pack of <= PACK_STRIDE words + a bit count per (channel, 2048-sample segment); bit k
// of a pack is at word k/32, bit k%32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
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

// Unit of hand-over: a "block" = 256 samples = two of the four-word pieces K1 stores side by side.
//   scanner    (wave 1) loads a block's 32 bytes of sign bits per lane (PLL_AHEAD blocks in flight),
//              forms the transition bits D = S ^ (S >> 1) (receiver.c:113) and expands them BYTE BY
//              BYTE through a 256-entry table in LDS -- the positions of a byte's set bits, eight to
//              an 8-byte entry -- appending each entry with one unaligned ds_write_b64 to the lane's
//              strip of the block's slot and advancing the cursor by the byte's popcount.  Seven
//              instructions per 8 samples whatever the data; a loop over the set bits costs twelve
//              per transition and runs max-over-lanes times.
//   recurrence (wave 0) takes a slot when it is complete: rows = the longest lane's transitions,
//              four to a ds_read_b32, lanes masked off step by step where their list has ended
//              (v_cmpx).  Six instructions per transition: it only advances the phase and writes
//              the number of the slice the transition toggles -- relative to the block's first
//              sample, so that it fits a byte -- over the position it has just read (one
//              ds_write_b32 per row).  It touches only LDS: beside a FIR that keeps the CU's vector
//              memory pipeline full, every global access of this wave costs it microseconds.
//   togglers   (waves 3, 4: even / odd blocks) XOR 1 << bit into the segment's pack in LDS for every
//              entry of a block the recurrence has finished (six instructions + a ds_xor_b32 each):
//              this was two thirds of the recurrence's work when it did it itself.
//   writer     (wave 2) takes a finished segment's pack out of LDS: complement, trim to the bit
//              count, carry the toggles that fall on a later segment's first slice, 64 bytes per
//              lane to HBM; clears the buffer for the segment after next.
// Monotonic LDS counters hand things over: blocks scanned / consumed, segments finished / written.
// The launch asks for more than half a CU's 160 KB of LDS, so the dispatcher places at most ONE of
// these workgroups per CU and two chains never share a SIMD.
constexpr int BLK_QUADS = 2;         // 16-byte pieces (four sign words) per block
constexpr int BLK_LEN = 128 * BLK_QUADS;   // samples per block: positions fit a byte
constexpr int SEG_BLKS = SEG_LEN / BLK_LEN;
constexpr int PLL_STRIP = BLK_LEN + 12;    // bytes per lane and slot: the positions + an 8-byte store's overhang +
                                     // the recurrence's read-ahead; 67 dwords (odd): lanes hit different banks
#ifndef PLL_SLOTS_N
#define PLL_SLOTS_N 6
#endif
constexpr int PLL_SLOTS = PLL_SLOTS_N;   // block slots between scanner, recurrence and togglers
#ifndef SCAN_EXP
#define SCAN_EXP 0
#endif
#ifndef PLL_SCAN_PRIO
#define PLL_SCAN_PRIO 3     // the scanner beside five FIR waves on its SIMD: in-pipeline PLL 0.52 -> 0.50 ms (C3), period -1 %
#endif
#ifndef PLL_AHEAD_N
#define PLL_AHEAD_N 2
#endif
constexpr int PLL_AHEAD = PLL_AHEAD_N;         // blocks of sign words the scanner has in flight
constexpr int PLL_SLOT_BYTES = 64 * PLL_STRIP + 64 * 4 + 64 + 64 * 4;     // strips, counts, rows, slices before the block
constexpr int PLL_PACKW = PACK_STRIDE + 1;   // words per lane and pack buffer: the pack + its bit count
constexpr int PLL_LUT_BYTES = 2048;
constexpr int PLL_FLAG_WORDS = 16 + 2 * 64 + 4 * 64;  // counters, the sign before / after the call per lane, bit counts of four segments
constexpr int PLL_NEED_LDS = PLL_LUT_BYTES + PLL_SLOTS * PLL_SLOT_BYTES + 2 * PLL_PACKW * 64 * 4 + PLL_FLAG_WORDS * 4;
static_assert(SEG_LEN % BLK_LEN == 0, "segments are whole blocks");
static_assert(BLK_LEN <= 256 && (PLL_STRIP / 4) % 2 == 1 && PLL_STRIP % 4 == 0, "byte positions, odd dword stride");

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

// One transition at position p (byte k of the list word E) of the current block.
// X = (pll0 + K + p0 * pllinc) * 2^7 + spare with p0 = the block's first sample and the slices before
// it taken out, T = p * pllinc * 2^7, so U = X + T is the unwrapped phase * 2^7 counted from the
// block's first sample: bit 22 is `pll >= 0x8000` (receiver.c:114), bits 31:23 the number of the slice
// the transition toggles, counted from the block's first sample (< 64: it fits byte k of W).
//     um = -(pll >= 0x8000);  X = (Q ^ um) + X   is  X + Q  or  X - Q - 1:
// the -1 is taken from the seven spare bits, which are set to all ones every four steps.
// A lane takes part while its list lasts: `rem` = its transitions from this row of four on (signed:
// it keeps counting down after the list has ended).
#define PLL_STEP(k)                                                                       \
    "v_cmpx_lt_i32 vcc, " #k ", %[rem]\n\t"                                                \
    "v_mul_u32_u24_sdwa %[T], %[K7], %[E] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t" \
    "v_add_u32 %[U], %[X], %[T]\n\t"                                                      \
    "v_bfe_i32 %[um], %[U], 22, 1\n\t"                                                    \
    "v_xad_u32 %[X], %[Q], %[um], %[X]\n\t"                                               \
    "v_lshrrev_b32_sdwa %[W], %[c23], %[U] dst_sel:BYTE_" #k " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"

// `ng` rows-of-four of one block: this lane's list starts at LDS byte address `ad`, `cnt` entries.
// v_cmpx narrows EXEC step by step inside a row; every lane writes its row word back (bytes past the
// end of its list are never looked at).  LDS operations complete in order: the wait lets the row's
// write stay in flight and counts the read-ahead issued before it.
__device__ __forceinline__ void pll_rows(uint32_t &X, uint32_t cnt, uint32_t ad, uint32_t ng,
                                         uint32_t Q, uint32_t K7)
{
    uint32_t U, um, T, E, F, W = 0, rem = cnt, c23 = 23;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b32 %[E], %[ad]\n\t"
        "ds_read_b32 %[F], %[ad] offset:4\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "1:\n\t"
        "v_or_b32 %[X], 0x7f, %[X]\n\t"
        PLL_STEP(0) PLL_STEP(1) PLL_STEP(2) PLL_STEP(3)
        "s_mov_b64 exec, %[sv]\n\t"
        "ds_write_b32 %[ad], %[W]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_mov_b32 %[E], %[F]\n\t"
        "ds_read_b32 %[F], %[ad] offset:8\n\t"
        "v_add_u32 %[ad], 4, %[ad]\n\t"
        "v_subrev_u32 %[rem], 4, %[rem]\n\t"
        "s_sub_u32 %[ng], %[ng], 1\n\t"
        "s_cmp_lg_u32 %[ng], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [X] "+v"(X), [U] "=&v"(U), [um] "=&v"(um), [T] "=&v"(T), [E] "=&v"(E), [F] "=&v"(F), [W] "+v"(W),
          [ad] "+v"(ad), [rem] "+v"(rem), [ng] "+s"(ng), [sv] "=&s"(sv)
        : [Q] "s"(Q), [K7] "v"(K7), [c23] "v"(c23)
        : "vcc", "scc", "memory");
}

// The same rows, second pass: entry = slice number relative to the block; `base` = slices of the
// segment before the block.  Bit base + entry of the lane's pack (word stride 64 dwords from `pb`)
// is toggled.
#define PLL_TOGGLE(k)                                                                     \
    "v_cmpx_lt_i32 vcc, " #k ", %[rem]\n\t"                                                \
    "v_add_u32_sdwa %[b], %[base], %[E] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t" \
    "v_lshlrev_b32_e64 %[m], %[b], 1\n\t"                                                 \
    "v_and_b32 %[b], 0x1e0, %[b]\n\t"                                                     \
    "v_lshl_add_u32 %[b], %[b], 3, %[pb]\n\t"                                             \
    "ds_xor_b32 %[b], %[m]\n\t"

__device__ __forceinline__ void pll_toggle_rows(uint32_t cnt, uint32_t ad, uint32_t ng, uint32_t base, uint32_t pb)
{
    uint32_t b, m, E, F, rem = cnt;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b32 %[E], %[ad]\n\t"
        "ds_read_b32 %[F], %[ad] offset:4\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "1:\n\t"
        PLL_TOGGLE(0) PLL_TOGGLE(1) PLL_TOGGLE(2) PLL_TOGGLE(3)
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_mov_b32 %[E], %[F]\n\t"
        "ds_read_b32 %[F], %[ad] offset:8\n\t"
        "v_add_u32 %[ad], 4, %[ad]\n\t"
        "v_subrev_u32 %[rem], 4, %[rem]\n\t"
        "s_sub_u32 %[ng], %[ng], 1\n\t"
        "s_cmp_lg_u32 %[ng], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [b] "=&v"(b), [m] "=&v"(m), [E] "=&v"(E), [F] "=&v"(F), [ad] "+v"(ad), [rem] "+v"(rem), [ng] "+s"(ng),
          [sv] "=&s"(sv)
        : [base] "v"(base), [pb] "v"(pb)
        : "vcc", "scc", "memory");
}

__host__ __device__ inline int n_seg_cap(int L)
{
    return (((L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
}

// LDS map (dynamic, from address 0: this kernel has no static LDS and the asm above relies on it):
//   [0, 2048)                       lut: positions of the set bits of a byte, MSB (oldest sample) first
//   PLL_SLOTS x PLL_SLOT_BYTES      block slots: 64 strips of PLL_STRIP bytes, cnt[64], rows
//   2 x PLL_PACKW x 64 words        pack buffers (toggle words per lane)
//   PLL_FLAG_WORDS                  hand-over counters; sign before / after the call per lane; the bit
//                                   counts of four segments per lane
// NSC scanners and NTG togglers: 2 (even / odd blocks) or 1 (every block) of each; six waves down to four
template <int NSC, int NTG>
__global__ __launch_bounds__(64 * (2 + NSC + NTG)) __attribute__((amdgpu_waves_per_eu(7, 8))) void pll_kernel(
    const uint4 *__restrict__ sgn4, uint32_t *__restrict__ pllst, uint32_t *__restrict__ prevst,
    uint32_t *__restrict__ lastbit, uint32_t *__restrict__ segbits, uint32_t *__restrict__ segcnt,
    uint32_t *__restrict__ watchdog, int N, int L, int n_seg_alloc, uint32_t pllinc,
    uint32_t *__restrict__ started, uint32_t stamp)
{
    extern __shared__ uint8_t lds[];
    uint64_t *lut = reinterpret_cast<uint64_t *>(lds);
    uint8_t *slots = lds + PLL_LUT_BYTES;
    uint32_t *pack = reinterpret_cast<uint32_t *>(slots + PLL_SLOTS * PLL_SLOT_BYTES);    // [2][PLL_PACKW][64]
    uint32_t *flag = pack + 2 * PLL_PACKW * 64;
    uint32_t *sign0 = flag + 16, *sign1 = flag + 16 + 64;      // level before the call's first / at its last sample
    uint32_t *nbuf = flag + 16 + 128;                          // [4][64]: bits of segment s at (s & 3)
    const int lane = threadIdx.x & 63;
    // wave -> role: 0 recurrence, 1 scanner, 2 writer, 3 toggler, then the second toggler (4) and scanner (5) if any
    const int wv = threadIdx.x >> 6, role = wv < 4 ? wv : (wv == 4 && NTG == 2 ? 4 : 5);
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int n_seg = n_seg_cap(L);
    const int n_blk = (L + BLK_LEN - 1) / BLK_LEN;
    // flag[0] blocks scanned, [1] blocks the recurrence has finished, [2] segments whose bit count it has
    // published, [3] packs written, [4] scanners done, [5] / [6] the next block the even / odd toggler takes,
    // [7] / [8] blocks scanned by the even / odd scanner (last block + 1)
    if (threadIdx.x < 16) flag[threadIdx.x] = (NTG == 2 && threadIdx.x == 6) ? 1u : 0u;
    // the launch's LAST workgroup is running (workgroups are placed in order): tell the host, which holds the next FIR
    // launch back after a cold start until this stage has its place (gnuais_capi.hip: cold start)
    if (started && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(started, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (role == 0) sign0[lane] = prevst[c] & 1u;               // receiver.h:44 prev, before the scanner rewrites it
    for (int v = threadIdx.x; v < 256; v += 64 * (2 + NSC + NTG)) {
        uint64_t e = 0;
        int n = 0;
        for (int b = 7; b >= 0; --b)
            if (v & (1 << b)) e |= (uint64_t) (7 - b) << (8 * n++);
        lut[v] = e;
    }
    for (int q = threadIdx.x; q < 2 * PLL_PACKW * 64; q += 64 * (2 + NSC + NTG)) pack[q] = 0;
    __syncthreads();
    const unsigned long long t_start = wall_clock64();
    // nothing here may spin forever: a wave that waits longer than this gives up (200 ms; the waves
    // of a workgroup normally hand over every few microseconds) and says so in *watchdog, which
    // the host turns into an error when the frames are drained
    auto expired = [&]() {
        if (wall_clock64() - t_start <= 20000000ull) return false;
        if (lane == 0) atomicOr(watchdog, 1u);
        return true;
    };

    if (role == 1 || role == 5) {                 // ---- the scanners: even / odd blocks ----
#if PLL_SCAN_PRIO
        __builtin_amdgcn_s_setprio(PLL_SCAN_PRIO);
#endif
        const int w = role == 1 ? 0 : 1;                           // (role 5 exists with NSC == 2 only)
        const uint4 *__restrict__ src = sgn4 + c;                  // piece i of this lane: src[i * N]
        // a block's lists depend on the sign before its first sample only: the newest bit of the block before
        auto last_word = [&](int b) -> uint32_t {                  // word 8 b - 1 (any valid word when there is none)
            if (NSC == 1) return 0;                                // a lone scanner carries the bit itself
            const int quad = (b >= 1 && b < n_blk) ? b * BLK_QUADS - 1 : 0;
            return reinterpret_cast<const uint32_t *>(src + (size_t) quad * (size_t) N)[3];
        };
        const int n_own = (n_blk - w + NSC - 1) / NSC;             // blocks w, w + NSC, ...
        uint4 q[PLL_AHEAD][BLK_QUADS];
        uint32_t pw[PLL_AHEAD];
#pragma unroll
        for (int j = 0; j < PLL_AHEAD; ++j) {
            const int b = j < n_own ? w + NSC * j : w;
#pragma unroll
            for (int h = 0; h < BLK_QUADS; ++h) q[j][h] = src[(size_t) ((b < n_blk ? b : 0) * BLK_QUADS + h) * (size_t) N];
            pw[j] = last_word(b);
        }
        int seen = 0;
        bool dead = false;
        uint32_t prev = sign0[lane];
        for (int i0 = 0; i0 < n_own && !dead; i0 += PLL_AHEAD) {
#pragma unroll
            for (int j = 0; j < PLL_AHEAD; ++j) {
                const int i = i0 + j, b = w + NSC * i;
                uint32_t S[4 * BLK_QUADS];
#pragma unroll
                for (int h = 0; h < BLK_QUADS; ++h) {
                    S[4 * h] = q[j][h].x; S[4 * h + 1] = q[j][h].y; S[4 * h + 2] = q[j][h].z; S[4 * h + 3] = q[j][h].w;
                }
                const uint32_t pword = pw[j];
                {   // loads are unconditional (past the end: an early block again), so that the compiler
                    // counts them and waits for exactly the oldest
                    const int nb = b + NSC * PLL_AHEAD;
                    const int lb = nb < n_blk ? nb : (w < n_blk ? w : 0);
#pragma unroll
                    for (int h = 0; h < BLK_QUADS; ++h) q[j][h] = src[(size_t) (lb * BLK_QUADS + h) * (size_t) N];
                    pw[j] = last_word(lb);
                }
                if (i < n_own && !dead) {
                    while (b - seen >= PLL_SLOTS && !dead) {       // slot b % PLL_SLOTS still in use?
                        const int t0 = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 5));
                        const int t1 = NTG == 2 ? __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 6)) : t0;
                        seen = t0 < t1 ? t0 : t1;              // every block before this one has been toggled
                        if (b - seen >= PLL_SLOTS) {
                            if (expired()) dead = true;
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                    if (!dead) {
                        if (NSC == 2) prev = b == 0 ? sign0[lane] : (pword & 1u);
                        uint8_t *slot = slots + (b % PLL_SLOTS) * PLL_SLOT_BYTES;
                        uint32_t cur = (uint32_t) (slot - lds) + (uint32_t) (lane * PLL_STRIP);   // LDS address
                        const uint32_t cur0 = cur;
                        const int nv = L - b * BLK_LEN;            // valid samples of this block (>= 1)
#pragma unroll
                        for (int w8 = 0; w8 < 4 * BLK_QUADS; ++w8) {
                            const int k = nv - 32 * w8;            // valid samples of this word
                            uint32_t d = S[w8] ^ ((S[w8] >> 1) | (prev << 31));      // receiver.c:113
                            if (k <= 0) {
                                d = 0;
                            } else if (k < 32) {
                                d &= ~0u << (32 - k);
                                prev = (S[w8] >> (32 - k)) & 1u;
                            } else {
                                prev = S[w8] & 1u;
                            }
                            uint64_t ent[4];
#pragma unroll
                            for (int y = 0; y < 4; ++y) ent[y] = lut[(d >> (24 - 8 * y)) & 0xffu];
#pragma unroll
                            for (int y = 0; y < 4; ++y) {
                                const uint32_t base = 0x01010101u * (uint32_t) (32 * w8 + 8 * y);
                                const uint64_t e = ent[y] + (((uint64_t) base << 32) | base);
#if SCAN_EXP            // timing experiment only (wrong lists): what the unaligned stores cost
                                asm volatile("ds_write_b64 %0, %1" :: "v"(cur & ~7u), "v"(e) : "memory");
#else
                                asm volatile("ds_write_b64 %0, %1" :: "v"(cur), "v"(e) : "memory");   // any byte address
#endif
                                cur += (uint32_t) __popc((d >> (24 - 8 * y)) & 0xffu);
                            }
                        }
                        const uint32_t cnt = cur - cur0;
                        reinterpret_cast<uint32_t *>(slot + 64 * PLL_STRIP)[lane] = cnt;
                        const uint32_t ng = wave_max((cnt + 3u) >> 2);
                        if (lane == 0) reinterpret_cast<uint32_t *>(slot + 64 * PLL_STRIP + 256)[0] = ng;
                        lds_flag_store(flag + 7 + w, (uint32_t) (b + 1));
                    }
                }
            }
        }
        if (((n_blk - 1) % NSC) == w) {                              // this scanner had the call's last block
            sign1[lane] = prev;
            lds_flag_store(flag + 4, 1u);
            if (live && !dead) prevst[cg] = prev;
        }
        return;
    }

    if (role == 2) {                              // ---- the writer ----
        // A transition after a segment's last slice toggles the first bit of the next segment that
        // has one (receiver.c:128), or of a later call: `par` carries that parity from pack to
        // pack.  Between calls it is the level at the last slice (receiver.h:38 lastbit) XOR the
        // level of the last sample (receiver.h:44 prev).
        uint32_t par = (lastbit[c] ^ sign0[lane]) & 1u;
        for (int s = 0; s < n_seg; ++s) {
            // the segment's bit count is published and every one of its blocks has been toggled
            const int b1 = (s + 1) * SEG_BLKS < n_blk ? (s + 1) * SEG_BLKS : n_blk;
            for (;;) {
                const int pub = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 2));
                const int t0 = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 5));
                const int t1 = NTG == 2 ? __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 6)) : t0;
                if (pub >= s + 1 && t0 >= b1 && t1 >= b1) break;
                if (expired()) return;
                __builtin_amdgcn_s_sleep(16);
            }
            uint32_t *pk = pack + (s & 1) * PLL_PACKW * 64 + lane;
            const uint32_t nb = nbuf[(s & 3) * 64 + lane];        // slices of the segment = bits of the pack
            uint32_t out[PACK_STRIDE], pd = 0;
#pragma unroll
            for (int w = 0; w < PACK_STRIDE; ++w) {
                const uint32_t tg = pk[w * 64];
                pk[w * 64] = 0;
                const int k = (int) nb - 32 * w;                  // valid bits of this word
                out[w] = ~tg & (k >= 32 ? ~0u : k > 0 ? (1u << k) - 1u : 0u);
                if (k >= 0 && k < 32) pd = (tg >> k) & 1u;        // toggles that fall on the NEXT slice
            }
            lds_flag_store(flag + 3, (uint32_t) (s + 1));
            if (nb) {
                out[0] ^= par;
                par = pd;
            } else {
                par ^= pd;
            }
            if (live) {
                uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(segbits + ((size_t) cg * n_seg_alloc + s) * PACK_STRIDE);
#pragma unroll
                for (int k = 0; k < PACK_STRIDE / 4; ++k)
                    dst[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
                segcnt[(size_t) cg * n_seg_alloc + s] = nb;
            }
        }
        while (lds_flag_load(flag + 4) == 0)
            if (expired()) return;
        if (live) {
            for (int s = n_seg; s < n_seg_alloc; ++s) segcnt[(size_t) cg * n_seg_alloc + s] = 0;
            lastbit[cg] = (sign1[lane] ^ par) & 1u;
        }
        return;
    }

    if (role >= 3) {                              // ---- the togglers: even / odd blocks ----
        const int w = role - 3;
        int done = 0, drained = 0;
        for (int b = w; b < n_blk; b += NTG) {
            const int s = b / SEG_BLKS;
            while (drained < s - 1) {                          // pack buffer s & 1 was segment s-2's
                drained = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 3));
                if (drained < s - 1) {
                    if (expired()) return;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            while (done < b + 1) {
                done = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 1));
                if (done < b + 1) {
                    if (expired()) return;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const uint8_t *slot = slots + (b % PLL_SLOTS) * PLL_SLOT_BYTES;
            const uint32_t cnt = reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP)[lane];
            const uint32_t ng = (uint32_t) __builtin_amdgcn_readfirstlane(
                (int) reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP + 256)[0]);
            const uint32_t base = reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP + 320)[lane];
            const uint32_t pb = (uint32_t) (reinterpret_cast<uint8_t *>(pack) - lds) +
                                (uint32_t) (((s & 1) * PLL_PACKW * 64 + lane) * 4);   // LDS address of this lane's pack word 0
            if (ng) pll_toggle_rows(cnt, (uint32_t) (slot - lds) + (uint32_t) (lane * PLL_STRIP), ng, base, pb);
            lds_flag_store(flag + 5 + w, (uint32_t) (b + NTG));
        }
        lds_flag_store(flag + 5 + w, 0x7fffffffu);
        return;
    }

    // ---- the recurrence ----
    // a long dependent chain: when it shares a SIMD with other waves it must win every issue slot
    // it can use
    __builtin_amdgcn_s_setprio(3);
    uint32_t X = ((pllst[c] & 0xffffu) << 7) | 0x7fu;          // receiver.h:40 pll, scaled; spare bits set
    const uint32_t Q = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((pllinc / 16u) << 7));   // receiver.c:84,115,117
    const uint32_t K7 = pllinc << 7;              // p * pllinc * 2^7 < 2^32 (create refuses pllinc > 14426)
    int seen = 0;
    bool dead = false;
    for (int s = 0; s < n_seg && !dead; ++s) {
        uint32_t segbase = 0;                                  // slices of this segment before the current block
        const int b1 = (s + 1) * SEG_BLKS < n_blk ? (s + 1) * SEG_BLKS : n_blk;
        for (int b = s * SEG_BLKS; b < b1 && !dead; ++b) {
            seen = 0;
            while (seen < b + 1 && !dead) {                    // scanned by the scanner of its parity
                seen = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 7 + (b % NSC)));
                if (seen < b + 1) {
                    if (expired()) dead = true;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (dead) break;
            uint8_t *slot = slots + (b % PLL_SLOTS) * PLL_SLOT_BYTES;
            const uint32_t cnt = reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP)[lane];
            const uint32_t ng = (uint32_t) __builtin_amdgcn_readfirstlane(
                (int) reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP + 256)[0]);
            segbase += X >> 23;                                // slice numbers inside the block start at 0
            X &= 0x007fffffu;
            reinterpret_cast<uint32_t *>(slot + 64 * PLL_STRIP + 320)[lane] = segbase;
            if (ng) pll_rows(X, cnt, (uint32_t) (slot - lds) + (uint32_t) (lane * PLL_STRIP), ng, Q, K7);
            lds_flag_store(flag + 1, (uint32_t) (b + 1));
            const int blen = L - b * BLK_LEN < BLK_LEN ? L - b * BLK_LEN : BLK_LEN;
            X += (uint32_t) blen * K7;                         // to the next block's first sample
        }
        if (dead) break;
        X |= 0x7fu;
        nbuf[(s & 3) * 64 + lane] = segbase + (X >> 23);       // slices so far = bits of the segment
        X = (X & 0x007fff80u) | 0x7fu;                         // receiver.c:133 pll &= 0xffff
        lds_flag_store(flag + 2, (uint32_t) (s + 1));
    }
    if (live && !dead) pllst[cg] = (X >> 7) & 0xffffu;
}

hipError_t pll3_prepare_device();
hipError_t launch_pll3(const PllLaunch &a, hipStream_t stream);

int pll_need_lds() { return PLL_NEED_LDS; }

hipError_t pll_prepare_device()
{
    const hipError_t e = pll3_prepare_device();
    if (e != hipSuccess) return e;
    const void *forms[] = {(const void *) pll_kernel<1, 1>, (const void *) pll_kernel<2, 1>, (const void *) pll_kernel<1, 2>,
                           (const void *) pll_kernel<2, 2>};
    for (const void *f : forms) {
        const hipError_t ef = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ef != hipSuccess) return ef;
    }
    return hipSuccess;
}

// Which form: the six-wave workgroup finishes a call a quarter sooner (0.31 against 0.41 ms per 48 000 samples)
// but issues a third more instructions doing so.  Where the batch leaves the chip half empty -- fewer channel
// groups than half the CUs -- the PLL stage's latency is what a call takes (BASELINE C2: 256 channels, 4 groups);
// where every CU holds a PLL workgroup beside five FIR waves per SIMD, the instructions are what it costs (C3: the
// pipelined call takes 0.535 ms with three waves, 0.56-0.59 with six).
hipError_t launch_pll(const PllLaunch &a, hipStream_t stream)
{
    // small batches: the time-parallel form (pll_tp.hip), where this implementation of it applies
    if ((a.variant == 7 || (a.variant == 0 && a.N <= PLL_TP_MAX_CHANNELS)) && pll_tp_applicable(a)) return launch_pll_tp(a, stream);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int groups = (a.N + 63) / 64;
    const int variant = a.variant == 3 || a.variant == 4 || a.variant == 6 || a.variant == 32 || a.variant == 51 || a.variant == 52
                            ? a.variant : (2 * groups <= n_cu ? 6 : 3);
    if (variant == 3 || variant == 32) return launch_pll3(a, stream);
    // one workgroup per CU while the channel groups fit one round; beyond that share the CUs evenly
    const int per_cu = (groups + n_cu - 1) / n_cu;
    const int lds = per_cu <= 1 ? std::max(PLL_NEED_LDS, PLL_LDS_BYTES)
                                : std::max(PLL_NEED_LDS, (160 * 1024 / per_cu) & ~1023);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#define PLL_FORM(NSC, NTG)                                                                                     \
    hipLaunchKernelGGL((pll_kernel<NSC, NTG>), dim3(groups), dim3(64 * (2 + NSC + NTG)), lds, stream,             \
                       (const uint4 *) a.sgn, a.pll, a.prev, a.lastbit, a.segbits, a.segcnt, a.watchdog, a.N, a.L, \
                       a.n_seg, a.pllinc, a.started, a.stamp)
    if (variant == 4) PLL_FORM(1, 1);
    else if (variant == 51) PLL_FORM(2, 1);          // measurement forms: which of the two is short of a wave
    else if (variant == 52) PLL_FORM(1, 2);
    else PLL_FORM(2, 2);
#undef PLL_FORM
    return hipGetLastError();
}

} // namespace gnuais
