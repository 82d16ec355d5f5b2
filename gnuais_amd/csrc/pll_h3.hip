// pll_h3.hip -- K2: bit-clock recovery PLL, slice and NRZI decode for gfx950, ONE recurrence wave and THREE helper waves
// per 64 channels.  Stands in for the per-sample loop of receiver_run(), gnuais src/receiver.c:109-135, for a whole
// batch of channels.
//
// The reference touches the phase on every sample, but only a sign change of the filter output (a "transition",
// receiver.c:113) makes it do anything that is not linear:
//
//     transition at sample t :  pll += (pll < 0x8000) ? +pllinc/16 : -pllinc/16     receiver.c:114-117
//     every sample           :  pll += pllinc;  overflow -> slice, pll &= 0xffff    receiver.c:122-133
//
// Write the phase without the `& 0xffff`: U(t) = pll0 + t * pllinc + K(t), K = the nudges so far.  U only grows
// (pllinc > pllinc/16), a nudge never crosses a multiple of 2^16 (pll < 0x8000 -> +q stays below 0x10000,
// pll >= 0x8000 -> -q stays above 0), and pllinc + q < 2^16, so the slices are exactly the times U crosses a multiple
// of 2^16: floor(U / 2^16) slices have happened before sample t, and the nudge at a transition needs U mod 2^16 there,
// nothing else.  The bit the reference emits at a slice is 1 if the level (the sign of the filter output) is the same
// as at the previous slice, 0 if it differs (receiver.c:126-132), i.e. NOT the parity of the transitions since the
// previous slice.  A transition at sample t (the level a slice AT t sees is already the new one) therefore toggles
// exactly one bit of the output: number floor(U(t) / 2^16).  So
//
//     bits = ~( XOR over the transitions of  1 << floor(U(t_j) / 2^16) )
//
// which turns 48 000 dependent steps per channel and call into ~13 000: one per transition of the busiest of a wave's
// 64 channels, re-synchronised every 128 samples (the busiest single channel has 8 000; lanes that ran ahead freely
// would need rings of thousands of entries: profiles/r05_pll_wave_budget.txt).
// Output: one pack of <= PACK_STRIDE words + a bit count per (channel, 2048-sample segment); bit k of a pack is at
// word k/32, bit k%32.
//
// How it is laid out follows from three measurements (profiles/r05_pll_wave_budget.txt, r05_ubench_pll_rows_sched.txt,
// r05_pll_h3_in_the_pipeline.txt):
//
//   * a lone wave issues ONE instruction per ~6.5 clock ticks, whatever the dependencies between them (reordering a row
//     for instruction-level parallelism changes nothing): the recurrence wave of round 4's three-wave form spent 92 % of
//     the launch inside its rows, 77 ticks per transition for twelve instructions.  Its time is its instruction count;
//     everything that is not the recurrence has to leave that wave.
//   * beside the FIR (four waves of 104 registers per SIMD, 96 left) a workgroup is placed at once only if it asks for
//     one wave of <= 96 registers per SIMD: round 4's six-wave form (two waves on two of the SIMDs) finished a call in
//     0.31 ms alone and needed 0.58-0.63 inside the pipeline, waiting for FIR waves to retire in pairs.
//   * a block of 128 samples holds at most 30 slices (create refuses pllinc > 14426), so the toggles of a block fit ONE
//     32-bit mask per lane: with the phase scaled by 2^8 the slice number of a transition is byte 3 of U, and
//     `v_lshlrev_b32_sdwa bit, U.byte3, 1` + `v_xor` toggle it in a REGISTER -- two instructions instead of the five
//     and the LDS atomic of the three-wave form, or the byte written back and a second wave's seven of the six-wave one.
//
// Hence FOUR waves, one per SIMD, 56 registers each:
//   recurrence (wave 0)  seven instructions per transition: phase, nudge, and the toggle into the block's mask; the mask
//                        goes into the segment's pack once per block (two ds_xor).  Nothing is written back per row.
//   helpers (waves 1-3)  helper h SCANS blocks h, h + 3, ... (sign words -> transition positions, byte by byte through the
//                        table: pll_common.h) into two block slots of its own, alternately, as far ahead as the
//                        recurrence has freed them; helper 0 also takes finished packs to HBM.
// A block is 128 samples (one 16-byte piece of sign words per lane): six slots of 64 x 140 bytes fit the 81 KB this
// stage may take of a CU's LDS beside K3's and the deframer's buffers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include "kernels.h"
#include "pll_common.h"      // LDS hand-over primitives, the byte table, the pack geometry

namespace gnuais {

constexpr int H3_BLK = 128;                    // samples per block: one uint4 of sign words
constexpr int H3_WORDS = H3_BLK / 32;
constexpr int H3_STRIP = H3_BLK + 12;          // bytes per lane and slot (positions + an 8-byte store's overhang + read-ahead)
constexpr int H3_SEG_BLKS = SEG_LEN / H3_BLK;  // 16
constexpr int H3_NH = 3;                       // helpers (a wave each; two: 0.347 against 0.326 ms alone, level in the pipeline)
constexpr int H3_SLOTS = 2 * H3_NH;            // slot of block b: b % H3_SLOTS (helper b % 3 owns slots h and h + 3)
constexpr int H3_NPACK = 4;                    // pack buffers (segment s in buffer s & 3)
// slot: 64 strips, cnt[64], rows (one word + padding)
constexpr int H3_OFF_CNT = 64 * H3_STRIP, H3_OFF_NG = H3_OFF_CNT + 256;
constexpr int H3_SLOT_BYTES = H3_OFF_NG + 64;
constexpr int H3_FLAG_WORDS = 32 + 2 * 64 + 4 * 64;   // counters, sign before / after the call, bit counts of four segments
constexpr int H3_NEED_LDS = PLL_LUT_BYTES + H3_SLOTS * H3_SLOT_BYTES + H3_NPACK * PLL_PACKW * 64 * 4 + H3_FLAG_WORDS * 4;
static_assert((H3_STRIP / 4) % 2 == 1 && H3_STRIP % 4 == 0, "odd dword stride: lanes hit different banks");
static_assert(SEG_LEN % H3_BLK == 0 && H3_NEED_LDS <= PLL_LDS_BYTES, "segments are whole blocks; the stage's LDS share");

// hand-over counters (all monotonic)
enum { F_RDONE = 1, F_SEGPUB = 2, F_WRITTEN = 3, F_LAST = 4, F_SCAN = 8 /* +h */ };

// (A measurement build of this kernel -- clock ticks per phase of the recurrence wave and of helper 0, on both clocks --
// produced profiles/r05_pll_wave_budget.txt; the instrumentation is in git history, commit b1fc8e6.)

// One block's transition lists: D = S ^ (S >> 1)
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

// One transition at position p (byte k of the list word E) of the current block.
// X = (pll0 + K + p0 * pllinc) * 2^8 + spare with p0 = the block's first sample and the slices before it taken out,
// T = p * pllinc * 2^8, U = X + T: bit 23 is `pll >= 0x8000` (receiver.c:114), BYTE 3 the number of the slice the
// transition toggles, counted from the block's first sample (<= 30).
//     um = -(pll >= 0x8000);  X = (Q ^ um) + X   is  X + Q  or  X - Q - 1   (receiver.c:114-117):
// the -1 is taken from the eight spare bits, which are set to all ones every four steps.
//     M ^= 1 << U.byte3   toggles the slice in the block's mask (v_lshlrev_b32 takes the low five bits of its count).
#define H3_STEP(k, E, kk)                                                                 \
    "v_cmpx_lt_i32 vcc, " #kk ", %[rem]\n\t"                                              \
    "v_mul_u32_u24_sdwa %[T], %[K8], %[" #E "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #k "\n\t" \
    "v_add_u32 %[U], %[X], %[T]\n\t"                                                      \
    "v_bfe_i32 %[um], %[U], 23, 1\n\t"                                                    \
    "v_xad_u32 %[X], %[Q], %[um], %[X]\n\t"                                               \
    "v_lshlrev_b32_sdwa %[T], %[U], %[one] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_xor_b32 %[M], %[M], %[T]\n\t"

// `ng` rows-of-four of one block, two rows to a loop turn: this lane's list starts at LDS byte address `ad`, `cnt`
// entries.  v_cmpx narrows EXEC monotonically (a lane whose list has ended never comes back; EXEC is restored at the
// end).  The list words of the next two rows are read one turn ahead (every lane reads: the address is the lane's own).
// Returns the block's toggle mask.
__device__ __forceinline__ uint32_t h3_rows(uint32_t &X, uint32_t cnt, uint32_t ad, uint32_t ng, uint32_t Q, uint32_t K8)
{
    uint32_t U, um, T, E, F, G, H, M = 0, one = 1;
    int32_t rem = (int32_t) cnt;
    uint32_t np = (ng + 1u) >> 1;                  // turns of two rows
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b32 %[E], %[ad]\n\t"
        "ds_read_b32 %[F], %[ad] offset:4\n\t"
        "1:\n\t"
        "ds_read_b32 %[G], %[ad] offset:8\n\t"
        "ds_read_b32 %[H], %[ad] offset:12\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_or_b32 %[X], 0xff, %[X]\n\t"
        H3_STEP(0, E, 0) H3_STEP(1, E, 1) H3_STEP(2, E, 2) H3_STEP(3, E, 3)
        "v_or_b32 %[X], 0xff, %[X]\n\t"
        H3_STEP(0, F, 4) H3_STEP(1, F, 5) H3_STEP(2, F, 6) H3_STEP(3, F, 7)
        "s_sub_u32 %[np], %[np], 1\n\t"
        "s_cmp_eq_u32 %[np], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "ds_read_b32 %[E], %[ad] offset:16\n\t"
        "ds_read_b32 %[F], %[ad] offset:20\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_or_b32 %[X], 0xff, %[X]\n\t"
        H3_STEP(0, G, 8) H3_STEP(1, G, 9) H3_STEP(2, G, 10) H3_STEP(3, G, 11)
        "v_or_b32 %[X], 0xff, %[X]\n\t"
        H3_STEP(0, H, 12) H3_STEP(1, H, 13) H3_STEP(2, H, 14) H3_STEP(3, H, 15)
        "v_add_u32 %[ad], 16, %[ad]\n\t"
        "v_subrev_u32 %[rem], 16, %[rem]\n\t"
        "s_sub_u32 %[np], %[np], 1\n\t"
        "s_cmp_lg_u32 %[np], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "2:\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [X] "+v"(X), [U] "=&v"(U), [um] "=&v"(um), [T] "=&v"(T), [E] "=&v"(E), [F] "=&v"(F), [G] "=&v"(G), [H] "=&v"(H),
          [M] "+v"(M), [ad] "+v"(ad), [rem] "+v"(rem), [np] "+s"(np), [sv] "=&s"(sv)
        : [Q] "s"(Q), [K8] "v"(K8), [one] "v"(one)
        : "vcc", "scc", "memory");
    return M;
}

// A finished pack leaves (helper 0): complement the toggle words, trim to the segment's nb bits, clear the buffer, take
// the toggle that fell on the NEXT slice (four words at a time: the helper's other role keeps its registers).  `par` = the parity carried from pack to pack (receiver.c:128: a transition
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

// LDS map (dynamic, from address 0: this kernel has no static LDS and the asm above relies on it):
//   [0, 2048)                       lut: positions of the set bits of a byte, MSB (oldest sample) first
//   H3_SLOTS x H3_SLOT_BYTES        block slots: 64 strips, cnt[64], rows
//   H3_NPACK x PLL_PACKW x 64 words pack buffers (toggle words per lane)
//   H3_FLAG_WORDS                   hand-over counters; sign before / after the call per lane; bit counts of four segments
__global__ __launch_bounds__(64 * (1 + H3_NH)) __attribute__((amdgpu_waves_per_eu(4, 8))) void pll_h3_kernel(
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
    if (threadIdx.x < 32) flag[threadIdx.x] = (threadIdx.x >= F_SCAN && threadIdx.x < F_SCAN + H3_NH) ? threadIdx.x - F_SCAN : 0u;
    if (role == 0) sign0[lane] = prevst[c] & 1u;               // receiver.h:44 prev, before a helper rewrites it
    pll_fill_lut(lut, (int) threadIdx.x, 64 * (1 + H3_NH));
    for (int q = threadIdx.x; q < H3_NPACK * PLL_PACKW * 64; q += 64 * (1 + H3_NH)) pack[q] = 0;
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

    if (role >= 1) {                   // ---- a helper: scans blocks h, h + 3, ...; helper 0 writes the packs ----
        const int h = role - 1;
        const uint4 *__restrict__ src = sgn4 + c;                  // piece b of this lane: src[b * N]
        // the writer's state (helper 0): a transition after a segment's last slice toggles the first bit of the next
        // segment that has one (receiver.c:128), or of a later call: `par` carries that parity from pack to pack.
        // Between calls it is the level at the last slice (receiver.h:38 lastbit) XOR the level of the last sample.
        uint32_t par = h == 0 ? (lastbit[c] ^ sign0[lane]) & 1u : 0u;
        int wseg = 0;
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
        for (int bs = h; !dead; bs += H3_NH) {
            if (bs >= n_blk && !(h == 0 && wseg < n_seg)) break;
            if (h == 0) {                                          // finished packs leave, in order (the only place)
                while (wseg < n_seg && peek(F_SEGPUB) >= wseg + 1) {
                    uint32_t *pk = pack + (wseg & (H3_NPACK - 1)) * PLL_PACKW * 64 + lane;
                    const uint32_t nb = nbuf[(wseg & 3) * 64 + lane];          // slices of the segment = bits of the pack
                    par = h3_write_pack(pk, nb, par, live, segbits + ((size_t) cg * n_seg_alloc + wseg) * PACK_STRIDE,
                                        segcnt + (size_t) cg * n_seg_alloc + wseg);
                    lds_flag_store(flag + F_WRITTEN, (uint32_t) (++wseg));
                }
                if (bs >= n_blk) {                                 // nothing left but the call's last packs: wait for them
                    if (expired()) dead = true;
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
            }
            while (peek(F_RDONE) < bs - (H3_SLOTS - 1) && !dead) { // slot bs % H3_SLOTS was block bs - H3_SLOTS's: walked?
                if (expired()) dead = true;
                __builtin_amdgcn_s_sleep(8);                       // (a block takes the recurrence ~2000 ticks)
            }
            if (dead) break;
            {
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
        }
        if (h == 0) {
            while (lds_flag_load(flag + F_LAST) == 0 && !dead)
                if (expired()) dead = true;
            if (live && !dead) {
                for (int s = n_seg; s < n_seg_alloc; ++s) segcnt[(size_t) cg * n_seg_alloc + s] = 0;
                lastbit[cg] = (sign1[lane] ^ par) & 1u;
            }
        }
        return;
    }

    // ---- the recurrence ----
    uint32_t X = ((pllst[c] & 0xffffu) << 8) | 0xffu;          // receiver.h:40 pll, scaled; spare bits set
    const uint32_t Q = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((pllinc / 16u) << 8));   // receiver.c:84,115,117
    const uint32_t K8 = pllinc << 8;              // p * pllinc * 2^8 < 2^32 and pllinc * 2^8 < 2^24 (create refuses pllinc > 14426)
    bool dead = false;
    for (int s = 0; s < n_seg && !dead; ++s) {
        while (peek(F_WRITTEN) < s - (H3_NPACK - 1) && !dead) {    // pack buffer s & 3 was segment s - 4's: long written
            if (expired()) dead = true;
            __builtin_amdgcn_s_sleep(2);
        }
        uint32_t segbase = 0;                                  // slices of this segment before the current block
        uint32_t *pk = pack + (s & (H3_NPACK - 1)) * PLL_PACKW * 64 + lane;   // this lane's pack word 0 (word stride 64)
        const int b1 = (s + 1) * H3_SEG_BLKS < n_blk ? (s + 1) * H3_SEG_BLKS : n_blk;
        for (int b = s * H3_SEG_BLKS; b < b1 && !dead; ++b) {
            // the helper's counter, the lane's count and the block's rows in ONE trip to the LDS: the three reads are
            // executed in this order, so a counter that says "scanned" vouches for the two values read behind it
            const uint8_t *slot = slots + (b % H3_SLOTS) * H3_SLOT_BYTES;
            uint32_t cnt, ng;
            for (;;) {
                uint32_t f, g;
                asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(f), "=&v"(cnt), "=&v"(g)
                             : "v"((uint32_t) (reinterpret_cast<uint8_t *>(flag + F_SCAN + b % H3_NH) - lds)),
                               "v"((uint32_t) (slot + H3_OFF_CNT - lds) + (uint32_t) lane * 4u),
                               "v"((uint32_t) (slot + H3_OFF_NG - lds))
                             : "memory");
                ng = (uint32_t) __builtin_amdgcn_readfirstlane((int) g);
                if (__builtin_amdgcn_readfirstlane((int) f) >= b + 1) break;       // scanned by its helper
                if (expired()) { dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (dead) break;
            segbase += X >> 24;                                // slice numbers inside the block start at 0
            X &= 0x00ffffffu;
            if (ng) {
                const uint32_t M = h3_rows(X, cnt, (uint32_t) (slot - lds) + (uint32_t) (lane * H3_STRIP), ng, Q, K8);
                lds_flag_store(flag + F_RDONE, (uint32_t) (b + 1));        // the slot is its helper's again
                // bit i of M = the toggle of slice segbase + i of the segment: into the pack (two words; <= 465 < 512 bits)
                const uint32_t sh = segbase & 31u, w = segbase >> 5;
                atomicXor(pk + w * 64, M << sh);                           // (this lane's own words: a plain read-modify-write
                atomicXor(pk + w * 64 + 64, (M >> 1) >> (31u - sh));       //  would do; ds_xor is the one-instruction form)
            } else {
                lds_flag_store(flag + F_RDONE, (uint32_t) (b + 1));
            }
            const int blen = L - b * H3_BLK < H3_BLK ? L - b * H3_BLK : H3_BLK;
            X += (uint32_t) blen * K8;                         // to the next block's first sample
        }
        if (dead) break;
        X |= 0xffu;
        nbuf[(s & 3) * 64 + lane] = segbase + (X >> 24);       // slices so far = bits of the segment
        X = (X & 0x00ffff00u) | 0xffu;                         // receiver.c:133 pll &= 0xffff
        lds_flag_store(flag + F_SEGPUB, (uint32_t) (s + 1));
    }
    if (live && !dead) pllst[cg] = (X >> 8) & 0xffffu;
}

int pll_need_lds() { return H3_NEED_LDS; }

hipError_t pll_prepare_device()
{
    return hipFuncSetAttribute((const void *) pll_h3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// Which form: the time-parallel one (pll_tp.hip: a workgroup per channel, 64 x the instructions, a fifth of the latency)
// where the batch leaves the chip nearly empty and that implementation applies; this one otherwise.
int pll_form_of(const PllLaunch &a)
{
    if ((a.variant == 7 || (a.variant == 0 && a.N <= PLL_TP_MAX_CHANNELS)) && pll_tp_applicable(a)) return 7;
    return 8;
}

hipError_t launch_pll(const PllLaunch &a, hipStream_t stream)
{
    if (pll_form_of(a) == 7) return launch_pll_tp(a, stream);
    // one workgroup per CU while the channel groups fit one round; beyond that share the CUs evenly
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int groups = (a.N + 63) / 64, per_cu = (groups + n_cu - 1) / n_cu;
    const int lds = per_cu <= 1 ? std::max(H3_NEED_LDS, PLL_LDS_BYTES) : std::max(H3_NEED_LDS, (160 * 1024 / per_cu) & ~1023);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pll_h3_kernel, dim3(groups), dim3(64 * (1 + H3_NH)), lds, stream, (const uint4 *) a.sgn, a.pll, a.prev, a.lastbit,
                       a.segbits, a.segcnt, a.watchdog, a.N, a.L, a.n_seg, a.pllinc);
    return hipGetLastError();
}

} // namespace gnuais
