// pll_common.h -- what the lane-per-channel forms of K2 share (pll_nrzi3.hip: three / four waves, the recurrence
// toggles its own bits; pll_nrzi.hip: four to six waves, togglers of their own): the block / strip geometry, the LDS
// hand-over primitives, the byte table, the scanner that takes every NSC-th block, and the writer's way out of a pack.
// One copy, so that a change to any of it is made once.  (gnuais src/receiver.c:109-135; the derivation is in
// pll_nrzi3.hip's header.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

constexpr int BLK_QUADS = 2;         // 16-byte pieces (four sign words) per block
constexpr int BLK_LEN = 128 * BLK_QUADS;   // samples per block: positions fit a byte
constexpr int SEG_BLKS = SEG_LEN / BLK_LEN;
constexpr int PLL_STRIP = BLK_LEN + 12;    // bytes per lane and slot: the positions + an 8-byte store's overhang +
                                     // the recurrence's read-ahead; 67 dwords (odd): lanes hit different banks
#ifndef PLL_SCAN_PRIO
#define PLL_SCAN_PRIO 3     // the scanner beside five FIR waves on its SIMD: in-pipeline PLL 0.52 -> 0.50 ms (C3), period -1 %
#endif
#ifndef PLL_AHEAD_N
#define PLL_AHEAD_N 2
#endif
constexpr int PLL_AHEAD = PLL_AHEAD_N;         // blocks of sign words the scanner has in flight
constexpr int PLL_PACKW = PACK_STRIDE + 1;   // words per lane and pack buffer: the pack + its bit count
constexpr int PLL_LUT_BYTES = 2048;
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

__host__ __device__ inline int n_seg_cap(int L)
{
    return (((L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
}

// the byte table: entry v = the positions of v's set bits, MSB (oldest sample) first, one per byte
__device__ __forceinline__ void pll_fill_lut(uint64_t *lut, int tid, int n_threads)
{
    for (int v = tid; v < 256; v += n_threads) {
        uint64_t e = 0;
        int n = 0;
        for (int b = 7; b >= 0; --b)
            if (v & (1 << b)) e |= (uint64_t) (7 - b) << (8 * n++);
        lut[v] = e;
    }
}

// One block's transition lists: the block's 4 * BLK_QUADS sign words S (this lane's channel), `prev` = the sign before
// the block (updated to the sign of its last valid sample), nv = valid samples.  D = S ^ (S >> 1) (receiver.c:113) is
// expanded BYTE BY BYTE through the table -- eight positions to an 8-byte entry -- with one unaligned ds_write_b64 per
// byte into the lane's strip of the slot, the cursor advancing by the byte's popcount.  Leaves the lane's count and
// the wave's number of rows of four in the slot.
__device__ __forceinline__ void pll_expand_block(const uint32_t (&S)[4 * BLK_QUADS], uint32_t &prev, int nv, uint8_t *lds,
                                                 uint8_t *slot, const uint64_t *lut, int lane)
{
    uint32_t cur = (uint32_t) (slot - lds) + (uint32_t) (lane * PLL_STRIP);   // LDS address
    const uint32_t cur0 = cur;
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
        for (int y = 0; y < 4; ++y)
            ent[y] = lut[(d >> (24 - 8 * y)) & 0xffu];
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const uint32_t base = 0x01010101u * (uint32_t) (32 * w8 + 8 * y);
            const uint64_t e = ent[y] + (((uint64_t) base << 32) | base);
            asm volatile("ds_write_b64 %0, %1" :: "v"(cur), "v"(e) : "memory");   // any byte address
            cur += (uint32_t) __popc((d >> (24 - 8 * y)) & 0xffu);
        }
    }
    const uint32_t cnt = cur - cur0;
    reinterpret_cast<uint32_t *>(slot + 64 * PLL_STRIP)[lane] = cnt;
    const uint32_t ng = wave_max((cnt + 3u) >> 2);
    if (lane == 0) reinterpret_cast<uint32_t *>(slot + 64 * PLL_STRIP + 256)[0] = ng;
}

// The scanner that takes blocks w, w + NSC, ... (NSC = 2: even / odd blocks; NSC = 1: every block, the sign carried in
// a register): PLL_AHEAD blocks of sign words in flight, slot b % SLOTS filled when `consumed()` says the block that used
// it is through, `published` raised to b + 1 behind it.  Returns false if it gave up (expired()).
template <int NSC, int SLOTS, int SLOT_BYTES, class Consumed, class Expired>
__device__ __forceinline__ bool pll_scan_blocks(int w, const uint4 *__restrict__ sgn4, int c, int N, int L, int n_blk,
                                                uint8_t *lds, uint8_t *slots, const uint64_t *lut, uint32_t *published,
                                                const uint32_t *sign0, uint32_t &prev, int lane, Consumed consumed,
                                                Expired expired)
{
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
    prev = sign0[lane];
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
                while (b - seen >= SLOTS && !dead) {       // slot b % SLOTS still in use?
                    seen = consumed();
                    if (b - seen >= SLOTS) {
                        if (expired()) dead = true;
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                if (!dead) {
                    if (NSC == 2) prev = b == 0 ? sign0[lane] : (pword & 1u);
                    pll_expand_block(S, prev, L - b * BLK_LEN, lds, slots + (b % SLOTS) * SLOT_BYTES, lut, lane);
                    lds_flag_store(published, (uint32_t) (b + 1));
                }
            }
        }
    }
    return !dead;
}

// The writer's way out of a finished pack: complement the toggle words, trim to the segment's nb bits, take the toggle
// that fell on the NEXT slice (pd), clear the buffer; then the parity carried from pack to pack (receiver.c:128: a
// transition after a segment's last slice toggles the first bit of the next segment that has one, or of a later call).
__device__ __forceinline__ void pll_pack_out(uint32_t *pk, uint32_t nb, uint32_t (&out)[PACK_STRIDE], uint32_t &pd)
{
    pd = 0;
#pragma unroll
    for (int w = 0; w < PACK_STRIDE; ++w) {
        const uint32_t tg = pk[w * 64];
        pk[w * 64] = 0;
        const int k = (int) nb - 32 * w;                  // valid bits of this word
        out[w] = ~tg & (k >= 32 ? ~0u : k > 0 ? (1u << k) - 1u : 0u);
        if (k >= 0 && k < 32) pd = (tg >> k) & 1u;        // toggles that fall on the NEXT slice
    }
}
// through == true: the pack is read by ANOTHER launch while this one still runs (K2b fed segment by segment): the words
// leave as agent-scope stores (written through to where every XCD sees them), so that no cache has to be flushed to
// publish them -- an agent-scope release fence writes back the XCD's whole L2, and the deframer's matching acquire
// invalidates one, once per segment and workgroup: measured, that alone took the pipeline from 0.57 to 0.85 ms per call.
__device__ __forceinline__ void pll_pack_store(uint32_t (&out)[PACK_STRIDE], uint32_t nb, uint32_t pd, uint32_t &par,
                                               bool live, uint32_t *__restrict__ segbits, uint32_t *__restrict__ segcnt,
                                               size_t cg, int n_seg_alloc, int s, bool through = false)
{
    if (nb) {
        out[0] ^= par;
        par = pd;
    } else {
        par ^= pd;
    }
    if (live && through) {
        uint32_t *dst = segbits + (cg * n_seg_alloc + s) * PACK_STRIDE;
#pragma unroll
        for (int k = 0; k < PACK_STRIDE; ++k) __hip_atomic_store(dst + k, out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(segcnt + cg * n_seg_alloc + s, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (live) {
        uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(segbits + (cg * n_seg_alloc + s) * PACK_STRIDE);
#pragma unroll
        for (int k = 0; k < PACK_STRIDE / 4; ++k)
            dst[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
        segcnt[cg * n_seg_alloc + s] = nb;
    }
}

} // namespace gnuais
