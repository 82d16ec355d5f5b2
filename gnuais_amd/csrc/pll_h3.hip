// pll_h3.hip -- K2, the form the full pipeline runs from round 5 on: bit-clock recovery PLL, slice and NRZI decode for
// gfx950 (gnuais src/receiver.c:109-135) as ONE recurrence wave and THREE uniform helper waves per 64 channels.
//
// What is computed, and why a transition toggles exactly one output bit (bits = ~XOR_j 1 << floor(U(t_j) / 2^16)), is
// derived in pll_nrzi3.hip.  What this form is about is measured in profiles/r05_pll_wave_budget.txt:
//
//   * a lone wave issues ONE instruction per ~6.5 clock ticks, whatever the dependencies between them (reordering a row
//     for instruction-level parallelism changes nothing, profiles/r05_ubench_pll_rows_sched.txt): the recurrence wave
//     of the three-wave form spends 92 % of the launch inside its rows, 77 ticks per transition for twelve
//     instructions.  Its time is its instruction count; everything that is not the recurrence has to leave that wave.
//   * beside the FIR (four waves of 104 registers per SIMD, 96 left) a workgroup is placed at once only if it asks for
//     one wave of <= 96 registers per SIMD: the six-wave form (pll_nrzi.hip: two waves on two of the SIMDs) finishes a
//     call in 0.31 ms alone and needs 0.58-0.63 inside the pipeline, waiting for FIR waves to retire in pairs.
//
// Hence FOUR waves, one per SIMD:
//   recurrence (wave 0)  six instructions per transition (pll_nrzi.hip's row): advances the phase and leaves, per
//                        transition, the number of the slice it toggles (relative to the block: a byte, in place).
//   helpers (waves 1-3)  helper h owns blocks h, h + 3, ...: it SCANS a block (sign words -> transition positions, byte by
//                        byte through the table: pll_common.h), and later, when the recurrence has walked it, TOGGLES
//                        the block's bits into the segment's pack; helper 0 also takes finished packs to HBM.  Each
//                        helper has two block slots of its own (scan b + 3 into one while b waits for the recurrence
//                        in the other), so no slot is ever handed from one helper to another.
// A block is 128 samples (one 16-byte piece of sign words per lane): six slots of 64 x 140 bytes fit the 81 KB this
// stage may take of a CU's LDS beside K3's and the deframer's buffers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include "kernels.h"
#include "pll_common.h"      // LDS hand-over primitives, the byte table, the writer's way out of a pack

namespace gnuais {

constexpr int H3_BLK = 128;                    // samples per block: one uint4 of sign words
constexpr int H3_WORDS = H3_BLK / 32;
constexpr int H3_STRIP = H3_BLK + 12;          // bytes per lane and slot (positions + an 8-byte store's overhang + read-ahead)
constexpr int H3_SEG_BLKS = SEG_LEN / H3_BLK;  // 16
constexpr int H3_NH = 3;                       // helpers
constexpr int H3_SLOTS = 2 * H3_NH;            // slot of block b: b % H3_SLOTS (helper b % 3 owns slots h and h + 3)
constexpr int H3_NPACK = 4;                    // pack buffers (segment s in buffer s & 3)
// slot: 64 strips, cnt[64], base[64] (slices of the segment before the block), rows (one word + padding)
constexpr int H3_OFF_CNT = 64 * H3_STRIP, H3_OFF_BASE = H3_OFF_CNT + 256, H3_OFF_NG = H3_OFF_BASE + 256;
constexpr int H3_SLOT_BYTES = H3_OFF_NG + 64;
constexpr int H3_FLAG_WORDS = 32 + 2 * 64 + 4 * 64;   // counters, sign before / after the call, bit counts of four segments
constexpr int H3_NEED_LDS = PLL_LUT_BYTES + H3_SLOTS * H3_SLOT_BYTES + H3_NPACK * PLL_PACKW * 64 * 4 + H3_FLAG_WORDS * 4;
static_assert((H3_STRIP / 4) % 2 == 1 && H3_STRIP % 4 == 0, "odd dword stride: lanes hit different banks");
static_assert(SEG_LEN % H3_BLK == 0 && H3_NEED_LDS <= PLL_LDS_BYTES, "segments are whole blocks; the stage's LDS share");

// hand-over counters (all monotonic)
enum { F_RDONE = 1, F_SEGPUB = 2, F_WRITTEN = 3, F_LAST = 4, F_SCAN = 8 /* +h */, F_TOG = 12 /* +h */ };

#ifdef PLLH3_BUDGET
// Measurement build only (EXTRA=-DPLLH3_BUDGET; scripts/pll_wave_budget.py h3): clock ticks per workgroup --
//   0 recurrence: total   1 ... waiting for a helper's scan   3 ... in the rows   4 rows of four   5 blocks
//   8 helper 0: total   9 ... scanning   10 ... waiting for the recurrence   11 ... toggling   12 ... writing packs
__device__ unsigned long long pllh3_budget[4096 * 16];
#define BUDGET(i, v) do { if (lane == 0 && blockIdx.x < 4096) pllh3_budget[blockIdx.x * 16 + (i)] = (v); } while (0)
#define TICK() ((unsigned long long) clock64())
#else
#define BUDGET(i, v) do { } while (0)
#define TICK() 0ull
#pragma clang diagnostic ignored "-Wunused-variable"
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#endif

// One block's transition lists (pll_common.h: pll_expand_block, for a block of H3_WORDS sign words): D = S ^ (S >> 1)
// (receiver.c:113) expanded byte by byte through the table, one unaligned ds_write_b64 per byte into the lane's strip.
// `prev` = the sign before the block, updated to the sign of its last valid sample; nv = valid samples.
__device__ __forceinline__ void h3_expand_block(const uint32_t (&S)[H3_WORDS], uint32_t &prev, int nv, uint8_t *lds,
                                                uint8_t *slot, const uint64_t *lut, int lane)
{
    uint32_t cur = (uint32_t) (slot - lds) + (uint32_t) (lane * H3_STRIP);   // LDS address
    const uint32_t cur0 = cur;
#pragma unroll
    for (int w = 0; w < H3_WORDS; ++w) {
        const int k = nv - 32 * w;             // valid samples of this word
        uint32_t d = S[w] ^ ((S[w] >> 1) | (prev << 31));      // receiver.c:113
        if (k <= 0) {
            d = 0;
        } else if (k < 32) {
            d &= ~0u << (32 - k);
            prev = (S[w] >> (32 - k)) & 1u;
        } else {
            prev = S[w] & 1u;
        }
        uint64_t ent[4];
#pragma unroll
        for (int y = 0; y < 4; ++y)
            ent[y] = lut[(d >> (24 - 8 * y)) & 0xffu];
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const uint32_t base = 0x01010101u * (uint32_t) (32 * w + 8 * y);
            const uint64_t e = ent[y] + (((uint64_t) base << 32) | base);
            asm volatile("ds_write_b64 %0, %1" :: "v"(cur), "v"(e) : "memory");   // any byte address
            cur += (uint32_t) __popc((d >> (24 - 8 * y)) & 0xffu);
        }
    }
    const uint32_t cnt = cur - cur0;
    reinterpret_cast<uint32_t *>(slot + H3_OFF_CNT)[lane] = cnt;
    const uint32_t ng = wave_max((cnt + 3u) >> 2);
    if (lane == 0) reinterpret_cast<uint32_t *>(slot + H3_OFF_NG)[0] = ng;
}

// One transition at position p (byte k of the list word E) of the current block -- the six-wave form's step
// (pll_nrzi.hip): X = (pll0 + K + p0 * pllinc) * 2^7 + spare with p0 = the block's first sample and the slices before it
// taken out, T = p * pllinc * 2^7, U = X + T: bit 22 is `pll >= 0x8000` (receiver.c:114), bits 31:23 the number of the
// slice the transition toggles, counted from the block's first sample (it fits byte k of W).
//     um = -(pll >= 0x8000);  X = (Q ^ um) + X   is  X + Q  or  X - Q - 1:
// the -1 is taken from the seven spare bits, which are set to all ones every four steps.
#define H3_STEP(k, E, kk)                                                                 \
    "v_cmpx_lt_i32 vcc, " #kk ", %[rem]\n\t"                                              \
    "v_mul_u32_u24_sdwa %[T], %[K7], %[" #E "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t" \
    "v_add_u32 %[U], %[X], %[T]\n\t"                                                      \
    "v_bfe_i32 %[um], %[U], 22, 1\n\t"                                                    \
    "v_xad_u32 %[X], %[Q], %[um], %[X]\n\t"                                               \
    "v_lshrrev_b32_sdwa %[" #E "], %[c23], %[U] dst_sel:BYTE_" #k " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"

// `ng` rows-of-four of one block, two rows to a loop turn: this lane's list starts at LDS byte address `ad`, `cnt`
// entries.  v_cmpx narrows EXEC monotonically (a lane whose list has ended never comes back; EXEC is restored for the
// row's write-back and read-ahead, which every lane does); the slice numbers go back over the positions they came from,
// in the register the row was read into.  The list words of the next two rows are read one turn ahead; LDS operations
// complete in order, which is what the waits count.
__device__ __forceinline__ void h3_rows(uint32_t &X, uint32_t cnt, uint32_t ad, uint32_t ng, uint32_t Q, uint32_t K7)
{
    uint32_t U, um, T, E, F, c23 = 23;
    int32_t rem = (int32_t) cnt;
    uint32_t np = (ng + 1u) >> 1;                  // turns of two rows
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b32 %[E], %[ad]\n\t"
        "ds_read_b32 %[F], %[ad] offset:4\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_or_b32 %[X], 0x7f, %[X]\n\t"
        H3_STEP(0, E, 0) H3_STEP(1, E, 1) H3_STEP(2, E, 2) H3_STEP(3, E, 3)
        "v_or_b32 %[X], 0x7f, %[X]\n\t"
        H3_STEP(0, F, 4) H3_STEP(1, F, 5) H3_STEP(2, F, 6) H3_STEP(3, F, 7)
        "s_mov_b64 exec, %[sv]\n\t"
        "ds_write_b32 %[ad], %[E]\n\t"
        "ds_write_b32 %[ad], %[F] offset:4\n\t"
        "ds_read_b32 %[E], %[ad] offset:8\n\t"
        "ds_read_b32 %[F], %[ad] offset:12\n\t"
        "v_add_u32 %[ad], 8, %[ad]\n\t"
        "v_subrev_u32 %[rem], 8, %[rem]\n\t"
        "s_sub_u32 %[np], %[np], 1\n\t"
        "s_cmp_lg_u32 %[np], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [X] "+v"(X), [U] "=&v"(U), [um] "=&v"(um), [T] "=&v"(T), [E] "=&v"(E), [F] "=&v"(F), [ad] "+v"(ad),
          [rem] "+v"(rem), [np] "+s"(np), [sv] "=&s"(sv)
        : [Q] "s"(Q), [K7] "v"(K7), [c23] "v"(c23)
        : "vcc", "scc", "memory");
}

// The same rows, second pass (a helper): entry = slice number relative to the block; `base` = slices of the segment
// before the block.  Bit base + entry of the lane's pack (word stride 64 dwords from `pb`) is toggled.
#define H3_TOGGLE(k, E, kk)                                                               \
    "v_cmpx_lt_i32 vcc, " #kk ", %[rem]\n\t"                                              \
    "v_add_u32_sdwa %[b], %[base], %[" #E "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t" \
    "v_lshlrev_b32_e64 %[m], %[b], 1\n\t"                                                 \
    "v_and_b32 %[b], 0x1e0, %[b]\n\t"                                                     \
    "v_lshl_add_u32 %[b], %[b], 3, %[pb]\n\t"                                             \
    "ds_xor_b32 %[b], %[m]\n\t"

__device__ __forceinline__ void h3_toggle_rows(uint32_t cnt, uint32_t ad, uint32_t ng, uint32_t base, uint32_t pb)
{
    uint32_t b, m, E, F;
    int32_t rem = (int32_t) cnt;
    uint32_t np = (ng + 1u) >> 1;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b32 %[E], %[ad]\n\t"
        "ds_read_b32 %[F], %[ad] offset:4\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        H3_TOGGLE(0, E, 0) H3_TOGGLE(1, E, 1) H3_TOGGLE(2, E, 2) H3_TOGGLE(3, E, 3)
        H3_TOGGLE(0, F, 4) H3_TOGGLE(1, F, 5) H3_TOGGLE(2, F, 6) H3_TOGGLE(3, F, 7)
        "s_mov_b64 exec, %[sv]\n\t"
        "ds_read_b32 %[E], %[ad] offset:8\n\t"
        "ds_read_b32 %[F], %[ad] offset:12\n\t"
        "v_add_u32 %[ad], 8, %[ad]\n\t"
        "v_subrev_u32 %[rem], 8, %[rem]\n\t"
        "s_sub_u32 %[np], %[np], 1\n\t"
        "s_cmp_lg_u32 %[np], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [b] "=&v"(b), [m] "=&v"(m), [E] "=&v"(E), [F] "=&v"(F), [ad] "+v"(ad), [rem] "+v"(rem), [np] "+s"(np),
          [sv] "=&s"(sv)
        : [base] "v"(base), [pb] "v"(pb)
        : "vcc", "scc", "memory");
}

// A finished pack leaves (helper 0): complement the toggle words, trim to the segment's nb bits, clear the buffer, take
// the toggle that fell on the NEXT slice (pll_common.h: pll_pack_out / pll_pack_store, four words at a time so that the
// helper's other roles keep their registers).  `par` = the parity carried from pack to pack (receiver.c:128: a transition
// after a segment's last slice toggles the first bit of the next segment that has one, or of a later call); returns it.
__device__ __forceinline__ uint32_t h3_write_pack(uint32_t *pk, uint32_t nb, uint32_t par, bool live,
                                                  uint32_t *__restrict__ dst, uint32_t *__restrict__ dcnt)
{
    uint32_t pd = 0;
#pragma unroll 1
    for (int g = 0; g < PACK_STRIDE / 4; ++g) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = 4 * g + j;
            const uint32_t tg = pk[w * 64];
            pk[w * 64] = 0;
            const int k = (int) nb - 32 * w;                  // valid bits of this word
            o[j] = ~tg & (k >= 32 ? ~0u : k > 0 ? (1u << k) - 1u : 0u);
            if (k >= 0 && k < 32) pd = (tg >> k) & 1u;        // toggles that fall on the NEXT slice
        }
        if (g == 0 && nb) o[0] ^= par;
        if (live) reinterpret_cast<uint4 *>(dst)[g] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    if (live) *dcnt = nb;
    return nb ? pd : (par ^ pd);
}

// segment s is published by the recurrence and every block of it has been toggled
__device__ __forceinline__ bool h3_seg_ready(uint32_t *flag, int s, int n_blk)
{
    if (__builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + F_SEGPUB)) < s + 1) return false;
    const int b1 = (s + 1) * H3_SEG_BLKS < n_blk ? (s + 1) * H3_SEG_BLKS : n_blk;
    bool ok = true;
#pragma unroll
    for (int q = 0; q < H3_NH; ++q) ok = ok && __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + F_TOG + q)) >= b1;
    return ok;
}

// LDS map (dynamic, from address 0: this kernel has no static LDS and the asm above relies on it):
//   [0, 2048)                       lut: positions of the set bits of a byte, MSB (oldest sample) first
//   H3_SLOTS x H3_SLOT_BYTES        block slots: 64 strips, cnt[64], base[64], rows
//   H3_NPACK x PLL_PACKW x 64 words pack buffers (toggle words per lane)
//   H3_FLAG_WORDS                   hand-over counters; sign before / after the call per lane; bit counts of four segments
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void pll_h3_kernel(
    const uint4 *__restrict__ sgn4, uint32_t *__restrict__ pllst, uint32_t *__restrict__ prevst,
    uint32_t *__restrict__ lastbit, uint32_t *__restrict__ segbits, uint32_t *__restrict__ segcnt,
    uint32_t *__restrict__ watchdog, int N, int L, int n_seg_alloc, uint32_t pllinc)
{
    extern __shared__ uint8_t lds[];
    uint64_t *lut = reinterpret_cast<uint64_t *>(lds);
    uint8_t *slots = lds + PLL_LUT_BYTES;
    uint32_t *pack = reinterpret_cast<uint32_t *>(slots + H3_SLOTS * H3_SLOT_BYTES);    // [H3_NPACK][PLL_PACKW][64]
    uint32_t *flag = pack + H3_NPACK * PLL_PACKW * 64;
    uint32_t *sign0 = flag + 32, *sign1 = flag + 32 + 64;      // level before the call's first / at its last sample
    uint32_t *nbuf = flag + 32 + 128;                          // [4][64]: bits of segment s at (s & 3)
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;    // 0: the recurrence; 1..3: helper role - 1
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int n_seg = n_seg_cap(L);
    const int n_blk = (L + H3_BLK - 1) / H3_BLK;
    if (threadIdx.x < 32) flag[threadIdx.x] = (threadIdx.x >= F_SCAN && threadIdx.x < F_SCAN + H3_NH)  ? threadIdx.x - F_SCAN
                                              : (threadIdx.x >= F_TOG && threadIdx.x < F_TOG + H3_NH) ? threadIdx.x - F_TOG : 0u;
    if (role == 0) sign0[lane] = prevst[c] & 1u;               // receiver.h:44 prev, before a helper rewrites it
    pll_fill_lut(lut, (int) threadIdx.x, 256);
    for (int q = threadIdx.x; q < H3_NPACK * PLL_PACKW * 64; q += 256) pack[q] = 0;
    __syncthreads();
    const unsigned long long t_start = wall_clock64();
    // nothing here may spin forever: a wave that waits longer than this gives up (200 ms; the waves of a workgroup
    // normally hand over every few microseconds) and says so in *watchdog, which the host turns into an error
    auto expired = [=]() {
        if (wall_clock64() - t_start <= 20000000ull) return false;
        if (lane == 0) atomicOr(watchdog, 1u);
        return true;
    };
#define peek(f) __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + (f)))
    __builtin_amdgcn_s_setprio(3);     // every wave of this workgroup is on the call's critical path

    if (role >= 1) {                   // ---- a helper: scans and toggles blocks h, h + 3, ...; helper 0 writes the packs ----
        const int h = role - 1;
        const uint4 *__restrict__ src = sgn4 + c;                  // piece b of this lane: src[b * N]
        // the writer's state (helper 0): a transition after a segment's last slice toggles the first bit of the next
        // segment that has one (receiver.c:128), or of a later call: `par` carries that parity from pack to pack.
        // Between calls it is the level at the last slice (receiver.h:38 lastbit) XOR the level of the last sample.
        uint32_t par = h == 0 ? (lastbit[c] ^ sign0[lane]) & 1u : 0u;
        int wseg = 0;
#define H3_SEG_READY(s) h3_seg_ready(flag, (s), n_blk)
#define H3_LOAD_BLOCK(b)           /* unconditional, clamped: the compiler counts them */                        \
    do {                                                                                                        \
        const int bb = (b) < n_blk ? (b) : 0;                                                                   \
        q = src[(size_t) bb * (size_t) N];                                                                      \
        pw = reinterpret_cast<const uint32_t *>(src + (size_t) (bb > 0 ? bb - 1 : 0) * (size_t) N)[3];          \
    } while (0)
        uint4 q;
        uint32_t pw;
        H3_LOAD_BLOCK(h);
        bool dead = false;
        unsigned long long hb_scan = 0, hb_wait = 0, hb_tog = 0, hb_wr = 0;
        const unsigned long long hb_t0 = TICK();
        for (int bs = h; !dead; bs += H3_NH) {
            const int bt = bs - H3_NH;                             // scan bs, then toggle the block scanned one turn ago
            if (bt >= n_blk && !(h == 0 && wseg < n_seg)) break;
            if (h == 0) {                                          // finished packs leave, in order (the only place)
                const unsigned long long t3 = TICK();
                while (wseg < n_seg && H3_SEG_READY(wseg)) {
                    uint32_t *pk = pack + (wseg & (H3_NPACK - 1)) * PLL_PACKW * 64 + lane;
                    const uint32_t nb = nbuf[(wseg & 3) * 64 + lane];          // slices of the segment = bits of the pack
                    par = h3_write_pack(pk, nb, par, live, segbits + ((size_t) cg * n_seg_alloc + wseg) * PACK_STRIDE,
                                        segcnt + (size_t) cg * n_seg_alloc + wseg);
                    lds_flag_store(flag + F_WRITTEN, (uint32_t) (++wseg));
                }
                hb_wr += TICK() - t3;
                if (bt >= n_blk) {                                 // nothing left but the call's last packs: wait for them
                    if (expired()) dead = true;
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
            }
            const unsigned long long t0 = TICK();
            if (bs < n_blk) {
                uint32_t S[H3_WORDS] = {q.x, q.y, q.z, q.w};
                uint32_t prev = bs == 0 ? sign0[lane] : (pw & 1u);
                H3_LOAD_BLOCK(bs + H3_NH);                         // the next own block's words, in flight while this one is expanded
                h3_expand_block(S, prev, L - bs * H3_BLK, lds, slots + (bs % H3_SLOTS) * H3_SLOT_BYTES, lut, lane);
                lds_flag_store(flag + F_SCAN + h, (uint32_t) (bs + H3_NH));
                if (bs == n_blk - 1) {                             // the call's last block: the level after the call
                    sign1[lane] = prev;
                    lds_flag_store(flag + F_LAST, 1u);
                    if (live) prevst[cg] = prev;
                }
            }
            const unsigned long long t1 = TICK();
            hb_scan += t1 - t0;
            if (bt >= 0 && bt < n_blk) {
                const int s = bt / H3_SEG_BLKS;
                // pack buffer s & 3 was segment s - 4's: written long ago (its blocks were toggled sixty blocks back, and
                // helper 0 looks after the packs at every turn) unless something is badly wrong
                while (peek(F_WRITTEN) < s - (H3_NPACK - 1) && !dead) {
                    if (expired()) dead = true;
                    __builtin_amdgcn_s_sleep(2);
                }
                while (peek(F_RDONE) < bt + 1 && !dead) {          // walked by the recurrence
                    if (expired()) dead = true;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (dead) break;
                const unsigned long long t2 = TICK();
                hb_wait += t2 - t1;
                const uint8_t *slot = slots + (bt % H3_SLOTS) * H3_SLOT_BYTES;
                const uint32_t cnt = reinterpret_cast<const uint32_t *>(slot + H3_OFF_CNT)[lane];
                const uint32_t ng = (uint32_t) __builtin_amdgcn_readfirstlane(
                    (int) reinterpret_cast<const uint32_t *>(slot + H3_OFF_NG)[0]);
                const uint32_t base = reinterpret_cast<const uint32_t *>(slot + H3_OFF_BASE)[lane];
                const uint32_t pb = (uint32_t) (reinterpret_cast<uint8_t *>(pack) - lds) +
                                    (uint32_t) (((s & (H3_NPACK - 1)) * PLL_PACKW * 64 + lane) * 4);   // this lane's pack word 0
                if (ng) h3_toggle_rows(cnt, (uint32_t) (slot - lds) + (uint32_t) (lane * H3_STRIP), ng, base, pb);
                lds_flag_store(flag + F_TOG + h, (uint32_t) (bt + H3_NH));
                hb_tog += TICK() - t2;
            }
        }
        lds_flag_store(flag + F_TOG + h, 0x7fffffffu);
        if (h == 0) {
            while (lds_flag_load(flag + F_LAST) == 0 && !dead)
                if (expired()) dead = true;
            if (live && !dead) {
                for (int s = n_seg; s < n_seg_alloc; ++s) segcnt[(size_t) cg * n_seg_alloc + s] = 0;
                lastbit[cg] = (sign1[lane] ^ par) & 1u;
            }
            BUDGET(8, TICK() - hb_t0); BUDGET(9, hb_scan); BUDGET(10, hb_wait); BUDGET(11, hb_tog); BUDGET(12, hb_wr);
        }
        return;
    }

    // ---- the recurrence ----
    uint32_t X = ((pllst[c] & 0xffffu) << 7) | 0x7fu;          // receiver.h:40 pll, scaled; spare bits set
    const uint32_t Q = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((pllinc / 16u) << 7));   // receiver.c:84,115,117
    const uint32_t K7 = pllinc << 7;              // p * pllinc * 2^7 < 2^32 (create refuses pllinc > 14426)
    bool dead = false;
    unsigned long long rc_wscan = 0, rc_rows = 0, rc_nrows = 0, rc_nblk = 0;
    const unsigned long long rc_t0 = TICK();
    for (int s = 0; s < n_seg && !dead; ++s) {
        uint32_t segbase = 0;                                  // slices of this segment before the current block
        const int b1 = (s + 1) * H3_SEG_BLKS < n_blk ? (s + 1) * H3_SEG_BLKS : n_blk;
        for (int b = s * H3_SEG_BLKS; b < b1 && !dead; ++b) {
            const unsigned long long s0 = TICK();
            while (peek(F_SCAN + b % H3_NH) < b + 1 && !dead) {    // scanned by its helper
                if (expired()) dead = true;
                __builtin_amdgcn_s_sleep(1);
            }
            if (dead) break;
            const unsigned long long s1 = TICK();
            rc_wscan += s1 - s0;
            uint8_t *slot = slots + (b % H3_SLOTS) * H3_SLOT_BYTES;
            const uint32_t cnt = reinterpret_cast<const uint32_t *>(slot + H3_OFF_CNT)[lane];
            const uint32_t ng = (uint32_t) __builtin_amdgcn_readfirstlane(
                (int) reinterpret_cast<const uint32_t *>(slot + H3_OFF_NG)[0]);
            segbase += X >> 23;                                // slice numbers inside the block start at 0
            X &= 0x007fffffu;
            reinterpret_cast<uint32_t *>(slot + H3_OFF_BASE)[lane] = segbase;
            if (ng) h3_rows(X, cnt, (uint32_t) (slot - lds) + (uint32_t) (lane * H3_STRIP), ng, Q, K7);
            lds_flag_store(flag + F_RDONE, (uint32_t) (b + 1));
#ifdef PLLH3_BUDGET
            rc_rows += TICK() - s1;
            rc_nrows += ng;
            rc_nblk += 1;
#endif
            const int blen = L - b * H3_BLK < H3_BLK ? L - b * H3_BLK : H3_BLK;
            X += (uint32_t) blen * K7;                         // to the next block's first sample
        }
        if (dead) break;
        X |= 0x7fu;
        nbuf[(s & 3) * 64 + lane] = segbase + (X >> 23);       // slices so far = bits of the segment
        X = (X & 0x007fff80u) | 0x7fu;                         // receiver.c:133 pll &= 0xffff
        lds_flag_store(flag + F_SEGPUB, (uint32_t) (s + 1));
    }
    if (live && !dead) pllst[cg] = (X >> 7) & 0xffffu;
    BUDGET(0, TICK() - rc_t0); BUDGET(1, rc_wscan); BUDGET(3, rc_rows); BUDGET(4, rc_nrows); BUDGET(5, rc_nblk);
}

#ifdef PLLH3_BUDGET
extern "C" int gnuais_debug_pllh3_budget(unsigned long long *out, int n_wg)
{
    if (n_wg > 4096) n_wg = 4096;
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(pllh3_budget), sizeof(unsigned long long) * 16 * (size_t) n_wg);
}
#endif

hipError_t pll_h3_prepare_device()
{
    return hipFuncSetAttribute((const void *) pll_h3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_pll_h3(const PllLaunch &a, hipStream_t stream)
{
    // one workgroup per CU while the channel groups fit one round; beyond that share the CUs evenly
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int groups = (a.N + 63) / 64, per_cu = (groups + n_cu - 1) / n_cu;
    const int lds = per_cu <= 1 ? std::max(H3_NEED_LDS, PLL_LDS_BYTES) : std::max(H3_NEED_LDS, (160 * 1024 / per_cu) & ~1023);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pll_h3_kernel, dim3(groups), dim3(256), lds, stream, (const uint4 *) a.sgn, a.pll, a.prev, a.lastbit,
                       a.segbits, a.segcnt, a.watchdog, a.N, a.L, a.n_seg, a.pllinc);
    return hipGetLastError();
}

} // namespace gnuais
