// pll_nrzi3.hip -- K2, three-wave form: bit-clock recovery PLL, slice and NRZI decode for gfx950.
//
// The form the full pipeline runs (16 384 channels: every CU has one of these workgroups beside five FIR waves
// per SIMD, and what a stage costs there is the instructions it issues): one scanner, a recurrence that toggles
// its own bits (ten instructions per transition), a writer.  pll_nrzi.hip is the six-wave form -- fewer
// instructions on the critical wave, a third more in total -- for batches that leave the chip half empty.
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
// per 64 channels, three waves that hand work to each other through LDS (pll3_kernel below): a
// scanner turns sign words into lists of transition positions, the recurrence walks them, a writer
// takes the finished bit packs to HBM.
//
// Output: one pack of <= PACK_STRIDE words + a bit count per (channel, 2048-sample segment); bit k
// of a pack is at word k/32, bit k%32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include "kernels.h"
#include "pll_common.h"      // geometry, LDS hand-over primitives, the scanner and the writer's pack handling: shared with pll_nrzi.hip

namespace gnuais {

#ifdef PLL3_BUDGET
// Measurement build only (make EXTRA=-DPLL3_BUDGET; scripts/pll_wave_budget.py): where the recurrence wave and the
// scanner of every workgroup spend their clock ticks.  16 counters per workgroup:
//   0 recurrence: ticks from the start barrier to its end   1 ... waiting for the scanner (block not published)
//   2 ... waiting for the writer (pack buffer not drained)   3 ... inside the rows (pll3_rows)
//   4 rows of four walked   5 blocks   6 transitions of the workgroup's lanes, summed   7 largest lane total
//   8 scanner: ticks from the start barrier to its end   9 ... waiting for a free slot   10 ... expanding blocks
//   11 writer: ticks to its end
__device__ unsigned long long pll3_budget[4096 * 16];
#define BUDGET(i, v) do { if (lane == 0 && blockIdx.x < 4096) pll3_budget[blockIdx.x * 16 + (i)] = (v); } while (0)
#define TICK() ((unsigned long long) clock64())
#else
#define BUDGET(i, v) do { } while (0)
#define TICK() 0ull
#pragma clang diagnostic ignored "-Wunused-variable"
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#endif

// Unit of hand-over: a "block" = 256 samples = two of the four-word pieces K1 stores side by side.
//   scanner    (wave 1) loads a block's 32 bytes of sign bits per lane (PLL_AHEAD blocks in flight),
//              forms the transition bits D = S ^ (S >> 1) (receiver.c:113) and expands them BYTE BY
//              BYTE through a 256-entry table in LDS -- the positions of a byte's set bits, eight to
//              an 8-byte entry -- appending each entry with one unaligned ds_write_b64 to the lane's
//              strip of the block's slot and advancing the cursor by the byte's popcount.  Seven
//              instructions per 8 samples whatever the data; a loop over the set bits costs twelve
//              per transition and runs max-over-lanes times.
//   recurrence (wave 0) takes a slot when it is complete: rows = the longest lane's transitions,
//              four to a ds_read_b32, lanes masked off row by row where their list has ended
//              (v_cmpx).  It touches only LDS: beside a FIR that keeps the CU's vector memory
//              pipeline full, every global access of this wave costs it microseconds.
//   writer     (wave 2) takes a finished segment's pack out of LDS: complement, trim to the bit
//              count, carry the toggles that fall on a later segment's first slice, 64 bytes per
//              lane to HBM; clears the buffer for the segment after next.
// Monotonic LDS counters hand things over: blocks scanned / consumed, segments finished / written.
// The launch asks for more than half a CU's 160 KB of LDS, so the dispatcher places at most ONE of
// these workgroups per CU and two chains never share a SIMD.
constexpr int PLL_SLOTS = 4;         // block slots between scanner and recurrence
#ifndef PLL3_LONE_SCANNER
#define PLL3_LONE_SCANNER 1
#endif
constexpr int PLL_SLOT_BYTES = 64 * PLL_STRIP + 64 * 4 + 64;     // strips, counts, rows
constexpr int PLL_FLAG_WORDS = 16 + 2 * 64;  // counters, then the sign before / after the call per lane
constexpr int PLL_NEED_LDS = PLL_LUT_BYTES + PLL_SLOTS * PLL_SLOT_BYTES + 2 * PLL_PACKW * 64 * 4 + PLL_FLAG_WORDS * 4;

// One transition at position p (byte k of the list word E) of the current block.
// X = (pll0 + K + block start * pllinc) * 2^7 + spare, T = p * pllinc * 2^7, so U = X + T is the
// unwrapped phase * 2^7: bit 22 is `pll >= 0x8000` (receiver.c:114), bits 31:23 the number of the
// slice the transition toggles (word = bits 31:28, bit = 27:23: a pack has 16 words).
//     um = -(pll >= 0x8000);  X = (Q ^ um) + X   is  X + Q  or  X - Q - 1:
// the -1 is taken from the seven spare bits, which are set to all ones every four steps.
// A lane takes part while its list lasts: `rem` = its transitions from this row of four on.
#define PLL_STEP(k)                                                                       \
    "v_cmpx_lt_u32 vcc, " #k ", %[rem]\n\t"                                                \
    "v_mul_u32_u24_sdwa %[T], %[K7], %[E] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t" \
    "v_add_u32 %[U], %[X], %[T]\n\t"                                                      \
    "v_bfe_i32 %[um], %[U], 22, 1\n\t"                                                    \
    "v_lshrrev_b32 %[m], 23, %[U]\n\t"                                                    \
    "v_xad_u32 %[X], %[Q], %[um], %[X]\n\t"                                               \
    "v_lshrrev_b32 %[U], 28, %[U]\n\t"                                                    \
    "v_lshlrev_b32_e64 %[m], %[m], 1\n\t"                                                 \
    "v_lshl_add_u32 %[U], %[U], 8, %[pb]\n\t"                                             \
    "ds_xor_b32 %[U], %[m]\n\t"

// `ng` rows-of-four of one block: this lane's list starts at LDS byte address `ad`, `cnt` entries.
// v_cmpx narrows EXEC, monotonically (a lane that leaves never has rem >= 4 again: its rem is no
// longer updated); EXEC is restored on exit.  LDS operations complete in order: the waits count
// the toggles and the read-ahead issued after the list word they are for.
__device__ __forceinline__ void pll3_rows(uint32_t &X, uint32_t cnt, uint32_t ad, uint32_t ng,
                                         uint32_t Q, uint32_t K7, uint32_t pb)
{
    uint32_t U, um, m, T, E, F, rem = cnt;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b32 %[E], %[ad]\n\t"
        "ds_read_b32 %[F], %[ad] offset:4\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "1:\n\t"
        "v_or_b32 %[X], 0x7f, %[X]\n\t"
        PLL_STEP(0) PLL_STEP(1) PLL_STEP(2) PLL_STEP(3)
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_mov_b32 %[E], %[F]\n\t"
        "ds_read_b32 %[F], %[ad] offset:8\n\t"
        "v_add_u32 %[ad], 4, %[ad]\n\t"
        "v_subrev_u32 %[rem], 4, %[rem]\n\t"
        "s_sub_u32 %[ng], %[ng], 1\n\t"
        "s_cmp_lg_u32 %[ng], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [X] "+v"(X), [U] "=&v"(U), [um] "=&v"(um), [m] "=&v"(m), [T] "=&v"(T), [E] "=&v"(E), [F] "=&v"(F),
          [ad] "+v"(ad), [rem] "+v"(rem), [ng] "+s"(ng), [sv] "=&s"(sv)
        : [Q] "s"(Q), [K7] "v"(K7), [pb] "v"(pb)
        : "vcc", "scc", "memory");
}

// LDS map (dynamic, from address 0: this kernel has no static LDS and the asm above relies on it):
//   [0, 2048)                       lut: positions of the set bits of a byte, MSB (oldest sample) first
//   PLL_SLOTS x PLL_SLOT_BYTES      block slots: 64 strips of PLL_STRIP bytes, cnt[64], rows
//   2 x PLL_PACKW x 64 words        pack buffers (toggle words + bit count per lane)
//   PLL_FLAG_WORDS                  hand-over counters; sign before / after the call per lane
// NSC scanners: 1 (three waves) or 2 (four waves: even / odd blocks, six block slots instead of four)
template <int NSC>
__global__ __launch_bounds__(64 * (2 + NSC)) __attribute__((amdgpu_waves_per_eu(7, 8))) void pll3_kernel(
    const uint4 *__restrict__ sgn4, uint32_t *__restrict__ pllst, uint32_t *__restrict__ prevst,
    uint32_t *__restrict__ lastbit, uint32_t *__restrict__ segbits, uint32_t *__restrict__ segcnt,
    uint32_t *__restrict__ watchdog, int N, int L, int n_seg_alloc, uint32_t pllinc,
    uint32_t *__restrict__ started, uint32_t stamp, uint32_t *__restrict__ progress, uint32_t progress_base)
{
    extern __shared__ uint8_t lds[];
    uint64_t *lut = reinterpret_cast<uint64_t *>(lds);
    uint8_t *slots = lds + PLL_LUT_BYTES;
    constexpr int SLOTS = NSC == 2 ? PLL_SLOTS + 2 : PLL_SLOTS;
    uint32_t *pack = reinterpret_cast<uint32_t *>(slots + SLOTS * PLL_SLOT_BYTES);    // [2][PLL_PACKW][64]
    uint32_t *flag = pack + 2 * PLL_PACKW * 64;
    uint32_t *sign0 = flag + 16, *sign1 = flag + 16 + 64;      // level before the call's first / at its last sample
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int n_seg = n_seg_cap(L);
    const int n_blk = (L + BLK_LEN - 1) / BLK_LEN;
    // flag[1] blocks consumed, [2] segments finished, [3] packs written, [4] scanners done, [5] / [6] blocks scanned by
    // the first / second scanner (last block + 1)
    if (threadIdx.x < 16) flag[threadIdx.x] = 0;
    // the launch's LAST workgroup is running (workgroups are placed in order): tell the host, which holds the next FIR
    // launch back after a cold start until this stage has its place (gnuais_capi.hip: cold start)
    if (started && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(started, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (role == 0) sign0[lane] = prevst[c] & 1u;               // receiver.h:44 prev, before the scanner rewrites it
    pll_fill_lut(lut, (int) threadIdx.x, 64 * (2 + NSC));
    for (int q = threadIdx.x; q < 2 * PLL_PACKW * 64; q += 64 * (2 + NSC)) pack[q] = 0;
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

    if (NSC == 1 && PLL3_LONE_SCANNER && role == 1) {   // ---- the lone scanner: every block, the sign carried in a register ----
#if PLL_SCAN_PRIO
        __builtin_amdgcn_s_setprio(PLL_SCAN_PRIO);
#endif
        const uint4 *__restrict__ src = sgn4 + c;                  // piece i of this lane: src[i * N]
        uint32_t prev = sign0[lane];
        uint4 q[PLL_AHEAD][BLK_QUADS];
#pragma unroll
        for (int j = 0; j < PLL_AHEAD; ++j)
#pragma unroll
            for (int h = 0; h < BLK_QUADS; ++h)
                q[j][h] = src[(size_t) ((j < n_blk ? j : 0) * BLK_QUADS + h) * (size_t) N];
        int seen = 0;
        bool dead = false;
        unsigned long long sc_wait = 0, sc_exp = 0;
        const unsigned long long sc_t0 = TICK();
        for (int b0 = 0; b0 < n_blk && !dead; b0 += PLL_AHEAD) {
#pragma unroll
            for (int j = 0; j < PLL_AHEAD; ++j) {
                const int b = b0 + j;
                uint32_t S[4 * BLK_QUADS];
#pragma unroll
                for (int h = 0; h < BLK_QUADS; ++h) {
                    S[4 * h] = q[j][h].x; S[4 * h + 1] = q[j][h].y; S[4 * h + 2] = q[j][h].z; S[4 * h + 3] = q[j][h].w;
                }
                {   // loads are unconditional (past the end: block 0 again), so that the compiler
                    // counts them and waits for exactly the oldest
                    const int nb = b + PLL_AHEAD;
#pragma unroll
                    for (int h = 0; h < BLK_QUADS; ++h)
                        q[j][h] = src[(size_t) ((nb < n_blk ? nb : 0) * BLK_QUADS + h) * (size_t) N];
                }
                if (b < n_blk && !dead) {
                    const unsigned long long w0 = TICK();
                    while (b - seen >= SLOTS && !dead) {       // slot b % SLOTS still in use?
                        seen = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 1));
                        if (b - seen >= SLOTS) {
                            if (expired()) dead = true;
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                    const unsigned long long w1 = TICK();
                    sc_wait += w1 - w0;
                    if (!dead) {
                        pll_expand_block(S, prev, L - b * BLK_LEN, lds, slots + (b % SLOTS) * PLL_SLOT_BYTES, lut, lane);
                        lds_flag_store(flag + 5, (uint32_t) (b + 1));
                    }
                    sc_exp += TICK() - w1;
                }
            }
        }
        sign1[lane] = prev;
        lds_flag_store(flag + 4, 1u);
        if (live && !dead) prevst[cg] = prev;
        BUDGET(8, TICK() - sc_t0); BUDGET(9, sc_wait); BUDGET(10, sc_exp);
        return;
    }

    if ((NSC == 2 || !PLL3_LONE_SCANNER) && (role == 1 || role == 3)) {   // ---- two scanners: even / odd blocks (or the same code for one) ----
#if PLL_SCAN_PRIO
        __builtin_amdgcn_s_setprio(PLL_SCAN_PRIO);
#endif
        const int w = role == 1 ? 0 : 1;                           // (role 3 exists with NSC == 2 only)
        uint32_t prev;
        const bool ok = pll_scan_blocks<NSC, SLOTS, PLL_SLOT_BYTES>(
            w, sgn4, c, N, L, n_blk, lds, slots, lut, flag + 5 + w, sign0, prev, lane,
            [&]() { return __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 1)); },      // blocks consumed
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
        int seen = 0;
        const unsigned long long wr_t0 = TICK();
        for (int s = 0; s < n_seg; ++s) {
            while (seen < s + 1) {
                seen = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 2));
                if (seen < s + 1) {
                    if (expired()) return;
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            uint32_t *pk = pack + (s & 1) * PLL_PACKW * 64 + lane;
            const uint32_t nb = pk[PACK_STRIDE * 64];             // slices of the segment = bits of the pack
            uint32_t out[PACK_STRIDE], pd;
            pll_pack_out(pk, nb, out, pd);
            lds_flag_store(flag + 3, (uint32_t) (s + 1));
            pll_pack_store(out, nb, pd, par, live, segbits, segcnt, (size_t) cg, n_seg_alloc, s, progress != nullptr);
            if (progress) {
                // the segment is K2b's from here on (it runs beside this launch): every lane's write-through stores
                // acknowledged, then the group's counter
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    __hip_atomic_store(progress + blockIdx.x, progress_base + (uint32_t) s + 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        while (lds_flag_load(flag + 4) == 0)
            if (expired()) return;
        if (live) {
            for (int s = n_seg; s < n_seg_alloc; ++s) {
                if (progress) __hip_atomic_store(segcnt + (size_t) cg * n_seg_alloc + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else segcnt[(size_t) cg * n_seg_alloc + s] = 0;
            }
            lastbit[cg] = (sign1[lane] ^ par) & 1u;
        }
        if (progress) {                                        // the empty segments behind a short call
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(progress + blockIdx.x, progress_base + (uint32_t) n_seg_alloc, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        BUDGET(11, TICK() - wr_t0);
        return;
    }

    // ---- the recurrence ----
    // a long dependent chain: when it shares a SIMD with other waves it must win every issue slot
    // it can use
    __builtin_amdgcn_s_setprio(3);
    uint32_t X = ((pllst[c] & 0xffffu) << 7) | 0x7fu;          // receiver.h:40 pll, scaled; spare bits set
    const uint32_t Q = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((pllinc / 16u) << 7));   // receiver.c:84,115,117
    const uint32_t K7 = pllinc << 7;              // p * pllinc * 2^7 < 2^32 (create refuses pllinc > 14426)
    int seen = 0, drained = 0;
    bool dead = false;
    unsigned long long rc_wscan = 0, rc_wdrain = 0, rc_rows = 0, rc_nrows = 0, rc_nblk = 0;
    uint32_t rc_mine = 0;
    const unsigned long long rc_t0 = TICK();
    for (int s = 0; s < n_seg && !dead; ++s) {
        const unsigned long long d0 = TICK();
        while (drained < s - 1 && !dead) {                     // pack buffer s & 1 was segment s-2's
            drained = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 3));
            if (drained < s - 1) {
                if (expired()) dead = true;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        rc_wdrain += TICK() - d0;
        const uint32_t pb = (uint32_t) (reinterpret_cast<uint8_t *>(pack) - lds) +
                            (uint32_t) (((s & 1) * PLL_PACKW * 64 + lane) * 4);   // LDS address of this lane's pack word 0
        const int b1 = (s + 1) * SEG_BLKS < n_blk ? (s + 1) * SEG_BLKS : n_blk;
        for (int b = s * SEG_BLKS; b < b1 && !dead; ++b) {
            if (NSC == 2) seen = 0;                            // (the other scanner's counter: nothing known yet)
            const unsigned long long s0 = TICK();
            while (seen < b + 1 && !dead) {
                seen = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 5 + (b % NSC)));
                if (seen < b + 1) {
                    if (expired()) dead = true;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const unsigned long long s1 = TICK();
            rc_wscan += s1 - s0;
            if (dead) break;
            const uint8_t *slot = slots + (b % SLOTS) * PLL_SLOT_BYTES;
            const uint32_t cnt = reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP)[lane];
            const uint32_t ng = (uint32_t) __builtin_amdgcn_readfirstlane(
                (int) reinterpret_cast<const uint32_t *>(slot + 64 * PLL_STRIP + 256)[0]);
            if (ng)
                pll3_rows(X, cnt, (uint32_t) (slot - lds) + (uint32_t) (lane * PLL_STRIP), ng, Q, K7, pb);
            lds_flag_store(flag + 1, (uint32_t) (b + 1));
#ifdef PLL3_BUDGET
            rc_rows += TICK() - s1;
            rc_nrows += ng;
            rc_nblk += 1;
            rc_mine += cnt;
#endif
            const int blen = L - b * BLK_LEN < BLK_LEN ? L - b * BLK_LEN : BLK_LEN;
            X += (uint32_t) blen * K7;                         // to the next block's first sample
        }
        if (dead) break;
        X |= 0x7fu;
        pack[(s & 1) * PLL_PACKW * 64 + PACK_STRIDE * 64 + lane] = X >> 23;      // slices so far = bits
        X = (X & 0x007fff80u) | 0x7fu;                         // receiver.c:133 pll &= 0xffff
        lds_flag_store(flag + 2, (uint32_t) (s + 1));
    }
    if (live && !dead) pllst[cg] = (X >> 7) & 0xffffu;
#ifdef PLL3_BUDGET
    {
        uint32_t tot = rc_mine, mx = wave_max(rc_mine);
        for (int o = 32; o; o >>= 1) tot += (uint32_t) __shfl_xor((int) tot, o);
        BUDGET(0, TICK() - rc_t0); BUDGET(1, rc_wscan); BUDGET(2, rc_wdrain); BUDGET(3, rc_rows);
        BUDGET(4, rc_nrows); BUDGET(5, rc_nblk); BUDGET(6, (unsigned long long) tot); BUDGET(7, (unsigned long long) mx);
    }
#endif
}

#ifdef PLL3_BUDGET
extern "C" int gnuais_debug_pll_budget(unsigned long long *out, int n_wg)
{
    if (n_wg > 4096) n_wg = 4096;
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(pll3_budget), sizeof(unsigned long long) * 16 * (size_t) n_wg);
}
#endif

hipError_t pll3_prepare_device()
{
    const hipError_t e = hipFuncSetAttribute((const void *) pll3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *) pll3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_pll3(const PllLaunch &a, hipStream_t stream)
{
    // one workgroup per CU while the channel groups fit one round; beyond that share the CUs evenly
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int groups = (a.N + 63) / 64, per_cu = (groups + n_cu - 1) / n_cu;
    const int nsc = a.variant == 32 ? 2 : 1;
    const int need = PLL_NEED_LDS + (nsc == 2 ? 2 * PLL_SLOT_BYTES : 0);
    const int lds = per_cu <= 1 ? std::max(need, PLL_LDS_BYTES) : std::max(need, (160 * 1024 / per_cu) & ~1023);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (nsc == 2)
        hipLaunchKernelGGL(pll3_kernel<2>, dim3(groups), dim3(64 * 4), lds, stream, (const uint4 *) a.sgn, a.pll,
                           a.prev, a.lastbit, a.segbits, a.segcnt, a.watchdog, a.N, a.L, a.n_seg, a.pllinc, a.started, a.stamp, a.progress, a.progress_base);
    else
        hipLaunchKernelGGL(pll3_kernel<1>, dim3(groups), dim3(64 * 3), lds, stream, (const uint4 *) a.sgn, a.pll,
                           a.prev, a.lastbit, a.segbits, a.segcnt, a.watchdog, a.N, a.L, a.n_seg, a.pllinc, a.started, a.stamp, a.progress, a.progress_base);
    return hipGetLastError();
}

} // namespace gnuais
