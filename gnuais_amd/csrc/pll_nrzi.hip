// pll_nrzi.hip -- K2, the forms with togglers of their own (four to six waves per 64 channels): bit-clock recovery PLL,
// slice and NRZI decode for gfx950 (gnuais src/receiver.c:109-135) -- and launch_pll(), which picks a form.
//
// What is computed, and why a transition toggles exactly one output bit (bits = ~XOR_j 1 << floor(U(t_j) / 2^16)), is
// derived in pll_nrzi3.hip, the form in which the recurrence wave toggles the bits itself.  Here the recurrence only
// advances the phase and leaves, per transition, the NUMBER of the slice it toggles; togglers apply those: fewer
// instructions on the critical wave, a third more in total -- for batches that leave the chip half empty.  What the forms
// share (block geometry, LDS hand-over, the scanners, the way a pack leaves) is in pll_common.h; the time-parallel form
// for small batches is pll_tp.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include "kernels.h"
#include "pll_common.h"      // geometry, LDS hand-over primitives, the scanner and the writer's pack handling: shared with pll_nrzi3.hip

namespace gnuais {

#ifdef PLL6_BUDGET
// Measurement build only (EXTRA=-DPLL6_BUDGET; scripts/pll_wave_budget.py 6): clock ticks per workgroup --
//   0 recurrence: total   1 ... waiting for a scanner   3 ... in the rows   4 rows of four   5 blocks
//   8 first toggler: total   9 ... waiting (pack buffer, recurrence)   10 ... in its rows
__device__ unsigned long long pll6_budget[4096 * 16];
#define BUDGET(i, v) do { if (lane == 0 && blockIdx.x < 4096) pll6_budget[blockIdx.x * 16 + (i)] = (v); } while (0)
#define TICK() ((unsigned long long) clock64())
#else
#define BUDGET(i, v) do { } while (0)
#define TICK() 0ull
#pragma clang diagnostic ignored "-Wunused-variable"
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#endif
#ifndef PLL_TOG_PRIO
#define PLL_TOG_PRIO 0
#endif

// Waves of a workgroup (pll_nrzi3.hip describes the scanner, the recurrence's rows and the writer):
//   recurrence (wave 0) six instructions per transition: it only advances the phase and writes the number of the slice
//              the transition toggles -- relative to the block's first sample, so that it fits a byte -- over the
//              position it has just read (one ds_write_b32 per row).  It touches only LDS.
//   scanners   (wave 1, and 5 with NSC = 2: even / odd blocks), writer (wave 2): pll_common.h
//   togglers   (wave 3, and 4 with NTG = 2: even / odd blocks) XOR 1 << bit into the segment's pack in LDS for every
//              entry of a block the recurrence has finished (six instructions + a ds_xor_b32 each): this was two thirds
//              of the recurrence's work when it did it itself.
#ifndef PLL_SLOTS_N
#define PLL_SLOTS_N 6
#endif
constexpr int PLL_SLOTS = PLL_SLOTS_N;   // block slots between scanner, recurrence and togglers
constexpr int PLL_SLOT_BYTES = 64 * PLL_STRIP + 64 * 4 + 64 + 64 * 4;     // strips, counts, rows, slices before the block
constexpr int PLL_FLAG_WORDS = 16 + 2 * 64 + 4 * 64;  // counters, the sign before / after the call per lane, bit counts of four segments
constexpr int PLL_NEED_LDS = PLL_LUT_BYTES + PLL_SLOTS * PLL_SLOT_BYTES + 2 * PLL_PACKW * 64 * 4 + PLL_FLAG_WORDS * 4;

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
    pll_fill_lut(lut, (int) threadIdx.x, 64 * (2 + NSC + NTG));
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
        uint32_t prev;
        const bool ok = pll_scan_blocks<NSC, PLL_SLOTS, PLL_SLOT_BYTES>(
            w, sgn4, c, N, L, n_blk, lds, slots, lut, flag + 7 + w, sign0, prev, lane,
            [&]() {                                                // every block before this one has been toggled
                const int t0 = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 5));
                const int t1 = NTG == 2 ? __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 6)) : t0;
                return t0 < t1 ? t0 : t1;
            },
            expired);
        const bool dead = !ok;
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
            uint32_t out[PACK_STRIDE], pd;
            pll_pack_out(pk, nb, out, pd);
            lds_flag_store(flag + 3, (uint32_t) (s + 1));
            pll_pack_store(out, nb, pd, par, live, segbits, segcnt, (size_t) cg, n_seg_alloc, s);
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
#if PLL_TOG_PRIO
        __builtin_amdgcn_s_setprio(PLL_TOG_PRIO);
#endif
        unsigned long long tg_wait = 0, tg_rows = 0;
        const unsigned long long tg_t0 = TICK();
        for (int b = w; b < n_blk; b += NTG) {
            const int s = b / SEG_BLKS;
            const unsigned long long q0 = TICK();
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
            const unsigned long long q1 = TICK();
            tg_wait += q1 - q0;
            const uint8_t *slot = slots + (b % PLL_SLOTS) * PLL_SLOT_BYTES;
            const uint32_t cnt = reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP)[lane];
            const uint32_t ng = (uint32_t) __builtin_amdgcn_readfirstlane(
                (int) reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP + 256)[0]);
            const uint32_t base = reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP + 320)[lane];
            const uint32_t pb = (uint32_t) (reinterpret_cast<uint8_t *>(pack) - lds) +
                                (uint32_t) (((s & 1) * PLL_PACKW * 64 + lane) * 4);   // LDS address of this lane's pack word 0
            if (ng) pll_toggle_rows(cnt, (uint32_t) (slot - lds) + (uint32_t) (lane * PLL_STRIP), ng, base, pb);
            lds_flag_store(flag + 5 + w, (uint32_t) (b + NTG));
            tg_rows += TICK() - q1;
        }
        lds_flag_store(flag + 5 + w, 0x7fffffffu);
        if (w == 0) { BUDGET(8, TICK() - tg_t0); BUDGET(9, tg_wait); BUDGET(10, tg_rows); }
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
    unsigned long long rc_wscan = 0, rc_rows = 0, rc_nrows = 0, rc_nblk = 0;
    const unsigned long long rc_t0 = TICK();
    for (int s = 0; s < n_seg && !dead; ++s) {
        uint32_t segbase = 0;                                  // slices of this segment before the current block
        const int b1 = (s + 1) * SEG_BLKS < n_blk ? (s + 1) * SEG_BLKS : n_blk;
        for (int b = s * SEG_BLKS; b < b1 && !dead; ++b) {
            seen = 0;
            const unsigned long long s0 = TICK();
            while (seen < b + 1 && !dead) {                    // scanned by the scanner of its parity
                seen = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 7 + (b % NSC)));
                if (seen < b + 1) {
                    if (expired()) dead = true;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (dead) break;
            const unsigned long long s1 = TICK();
            rc_wscan += s1 - s0;
            uint8_t *slot = slots + (b % PLL_SLOTS) * PLL_SLOT_BYTES;
            const uint32_t cnt = reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP)[lane];
            const uint32_t ng = (uint32_t) __builtin_amdgcn_readfirstlane(
                (int) reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP + 256)[0]);
            segbase += X >> 23;                                // slice numbers inside the block start at 0
            X &= 0x007fffffu;
            reinterpret_cast<uint32_t *>(slot + 64 * PLL_STRIP + 320)[lane] = segbase;
            if (ng) pll_rows(X, cnt, (uint32_t) (slot - lds) + (uint32_t) (lane * PLL_STRIP), ng, Q, K7);
            lds_flag_store(flag + 1, (uint32_t) (b + 1));
#ifdef PLL6_BUDGET
            rc_rows += TICK() - s1;
            rc_nrows += ng;
            rc_nblk += 1;
#endif
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
    BUDGET(0, TICK() - rc_t0); BUDGET(1, rc_wscan); BUDGET(3, rc_rows); BUDGET(4, rc_nrows); BUDGET(5, rc_nblk);
}

#ifdef PLL6_BUDGET
extern "C" int gnuais_debug_pll6_budget(unsigned long long *out, int n_wg)
{
    if (n_wg > 4096) n_wg = 4096;
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(pll6_budget), sizeof(unsigned long long) * 16 * (size_t) n_wg);
}
#endif

hipError_t pll3_prepare_device();
hipError_t launch_pll3(const PllLaunch &a, hipStream_t stream);

int pll_need_lds() { return PLL_NEED_LDS; }

hipError_t pll_prepare_device()
{
    const hipError_t e = pll3_prepare_device();
    if (e != hipSuccess) return e;
    const hipError_t eh = pll_h3_prepare_device();
    if (eh != hipSuccess) return eh;
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
int pll_form_of(const PllLaunch &a)
{
    // small batches: the time-parallel form (pll_tp.hip), where this implementation of it applies
    if ((a.variant == 7 || (a.variant == 0 && a.N <= PLL_TP_MAX_CHANNELS)) && pll_tp_applicable(a)) return 7;
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int groups = (a.N + 63) / 64;
    return a.variant == 3 || a.variant == 4 || a.variant == 6 || a.variant == 8 || a.variant == 32 || a.variant == 51 || a.variant == 52
               ? a.variant : (2 * groups <= n_cu ? 6 : 8);
}

hipError_t launch_pll(const PllLaunch &a, hipStream_t stream)
{
    const int variant = pll_form_of(a);
    if (variant == 7) return launch_pll_tp(a, stream);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int groups = (a.N + 63) / 64;
    if (variant == 3 || variant == 32) return launch_pll3(a, stream);
    if (variant == 8) return launch_pll_h3(a, stream);
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
