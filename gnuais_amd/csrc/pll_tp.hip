// pll_tp.hip -- K2, TIME-PARALLEL form: bit-clock recovery PLL, slice and NRZI decode (gnuais src/receiver.c:109-135)
// for batches too small to fill the chip with one lane per channel (BASELINE C2: 256 channels = 4 workgroups of the
// lane-per-channel kernels on 256 CUs, 0.31 ms of pure recurrence latency per 48 000 samples).
//
// The recurrence.  With the phase unwrapped, U(t) = pll0 + t * pllinc + q * c(t), q = pllinc / 16 and c(t) the NET count
// of nudges before sample t, the reference's loop does exactly two things:
//   * at a sign change at sample t (receiver.c:113-118):   c += 1 if bit 15 of U(t) is clear, else c -= 1;
//   * it slices whenever U crosses a multiple of 2^16 (:124): floor(U / 2^16) slices lie before sample t, and a sign
//     change at t toggles output bit number floor(U(t) / 2^16) of   bits = ~XOR_j (1 << floor(U(t_j) / 2^16))
//     (pll_h3.hip has the derivation).
// Everything nonlinear is the one integer c, and c moves by +-1 per transition with a KNOWN parity (the number of
// transitions so far).  That is what makes an exact block-parallel form affordable:
//
//   pass 1  one WAVE per (channel, block of 256 samples), LANE = CANDIDATE: the 64 lanes run the block from the 64
//           values c_in = centre + 2 (k - 32) of the right parity.  The transitions of the block are the same for every
//           lane, so the scan of the sign words is scalar code and a step is four vector instructions.  Result: a table
//           c_out[block][k] in LDS -- the block's map on a window of +-64 around the centre.
//   walk    one lane goes through the blocks of a chunk in order: c <- table[block][(c - lo) / 2], 60 dependent LDS reads
//           instead of 15 000 dependent steps.  A value that has left the window (the phase diffuses by one nudge per
//           transition in noise) makes the walker run that block itself, exactly, from the true value; the next chunk is
//           centred on the value the walk ended with, so that this stays rare.
//   pass 3  LANE = BLOCK: every block runs once more from its now known c_in, per-lane bit scan, and XORs its toggles
//           into the segment's bit pack in LDS; then the packs leave as the lane-per-channel kernels write them
//           (complement, trim, parity carried from segment to segment and call to call: pll_h3.hip's pack writer).
//
// Exact for every input: pass 1 computes the true block map on its window, the walk composes maps or falls back to the
// map's definition, pass 3 is the reference's loop restricted to one block.  Work is 64 x the serial kernel's vector
// instructions (that is the price of the candidates), which a small batch has room for: 256 channels x 7 500
// transitions x 64 lanes = 2 M wave-steps on 1024 SIMDs.  launch_pll() takes this form for small batches only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gnuais {

namespace {

constexpr int TP_BLK = 256;                 // samples per block = 8 sign words = two 16-byte pieces of sgn_index()
constexpr int TP_SEG_BLKS = SEG_LEN / TP_BLK;
constexpr int TP_CHUNK = 64;                // blocks per chunk: one table in LDS, one walk, one re-centring
constexpr int TP_WAVES = 16;                // waves of a workgroup (one channel)
constexpr int TP_HALF = 32;                 // candidates on either side of the centre (in steps of 2)

// the 256 transition bits of block b of channel c as four 64-bit words, oldest sample in bit 63 of d[0]; `carry` = the
// sign of the sample before the block; returns the sign of the block's last valid sample.  Wave-uniform arguments:
// the loads are scalar loads.
// UNIFORM: every lane asks for the same block; the words are then moved to scalar registers whatever kind of load the
// compiler chose (after the kernel's first store to global memory it no longer uses scalar loads by itself), so that
// the scan of the transitions that follows is scalar code.
template <bool UNIFORM>
__device__ __forceinline__ uint32_t tp_block_bits(const uint32_t *__restrict__ sgn, int N, int c, int b, int L,
                                                  uint32_t carry, uint64_t d[4])
{
    const uint4 p0 = *reinterpret_cast<const uint4 *>(sgn + ((size_t) (2 * b) * (size_t) N + (size_t) c) * 4);
    const uint4 p1 = *reinterpret_cast<const uint4 *>(sgn + ((size_t) (2 * b + 1) * (size_t) N + (size_t) c) * 4);
    uint32_t S[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    if constexpr (UNIFORM) {
#pragma unroll
        for (int w = 0; w < 8; ++w) S[w] = (uint32_t) __builtin_amdgcn_readfirstlane((int) S[w]);
    }
    const int nv = L - b * TP_BLK;          // valid samples of this block (>= 1)
    uint32_t prev = carry & 1u, D[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int k = nv - 32 * w;          // valid samples of this word
        uint32_t x = S[w] ^ ((S[w] >> 1) | (prev << 31));       // receiver.c:113 curr != prev
        if (k <= 0) {
            x = 0;
        } else if (k < 32) {
            x &= ~0u << (32 - k);
            prev = (S[w] >> (32 - k)) & 1u;
        } else {
            prev = S[w] & 1u;
        }
        D[w] = x;
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) d[w] = ((uint64_t) D[2 * w] << 32) | D[2 * w + 1];
    return prev;
}

// one block from c_in (per lane), transitions d[] (uniform): returns c_out.  base = pll0 + (first sample of the block) *
// pllinc, all mod 2^32 (U < 2^32 is checked by the launcher).  receiver.c:113-118.
// The lane carries A = base + q * c, so that the phase at the transition at position p of the block is ONE
// v_mad_u32_u24: U = p * pllinc + A, with p in a scalar register.  Sixteen waves of a workgroup share their CU's one
// scalar ALU, and the bit scan already costs it five instructions a transition (find, shift, clear, loop): the
// position's product and sum are therefore kept OFF it (`inc_v` is pllinc in a vector register; the empty asm hides from
// the compiler that it is uniform).
__device__ __forceinline__ int32_t tp_run_block(uint64_t d[4], uint32_t base, uint32_t pllinc, uint32_t q, int32_t c_in)
{
    uint32_t inc_v = pllinc;
    asm volatile("" : "+v"(inc_v));
    const uint32_t A0 = base + q * (uint32_t) c_in;
    uint32_t A = A0, U, um;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        // positions inside word w count from its first sample: A carries the word's offset while the word lasts
        // (the scan's words and q must BE in scalar registers for the asm below, not merely uniform)
        uint64_t m = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (d[w] >> 32)) << 32) |
                     (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) d[w]), t;
        int p;
        const uint32_t q_s = (uint32_t) __builtin_amdgcn_readfirstlane((int) q);
        asm volatile(
            "s_cmp_eq_u64 %[m], 0\n\t"
            "s_cbranch_scc1 2f\n\t"
            "1:\n\t"
            "s_flbit_i32_b64 %[p], %[m]\n\t"                  // the oldest transition left: bit 63 - p
            "s_lshr_b64 %[t], %[top], %[p]\n\t"
            "s_andn2_b64 %[m], %[m], %[t]\n\t"
            "v_mad_u32_u24 %[U], %[p], %[inc], %[A]\n\t"      // the phase there (receiver.c:114 reads its bit 15)
            "v_bfe_i32 %[um], %[U], 15, 1\n\t"                // -1: pll >= 0x8000
            "s_cmp_lg_u64 %[m], 0\n\t"
            "v_xad_u32 %[A], %[q], %[um], %[A]\n\t"           // A + q, or A - q - 1 ...
            "v_sub_u32 %[A], %[A], %[um]\n\t"                 // ... + 1
            "s_cbranch_scc1 1b\n\t"
            "2:\n\t"
            : [m] "+s"(m), [A] "+v"(A), [U] "=&v"(U), [um] "=&v"(um), [p] "=&s"(p), [t] "=&s"(t)
            : [top] "s"(0x8000000000000000ull), [inc] "v"(inc_v), [q] "s"(q_s)
            : "scc");
        A += 64u * pllinc;
    }
    // q * (c_out - c_in) = A - A0 - 256 * pllinc, |c_out - c_in| <= 256: one exact division per block
    return c_in + (int32_t) (A - A0 - 256u * pllinc) / (int32_t) q;
}

__device__ __forceinline__ int tp_popc(const uint64_t d[4])
{
    return __popcll(d[0]) + __popcll(d[1]) + __popcll(d[2]) + __popcll(d[3]);
}

} // namespace


// One workgroup per channel.  LDS: the chunk's table, per-block transition counts / true c_in / last signs, the bit packs.
__global__ __launch_bounds__(64 * TP_WAVES) void pll_tp_kernel(
    const uint32_t *__restrict__ sgn, uint32_t *__restrict__ pllst, uint32_t *__restrict__ prevst,
    uint32_t *__restrict__ lastbit, uint32_t *__restrict__ segbits, uint32_t *__restrict__ segcnt,
    int N, int L, int n_seg_alloc, uint32_t pllinc)
{
    extern __shared__ uint32_t tp_lds[];
    const int n_blk = (L + TP_BLK - 1) / TP_BLK;
    const int n_seg = (((L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
    int32_t *tab = reinterpret_cast<int32_t *>(tp_lds);       // [TP_CHUNK][64] c_out of candidate k
    uint32_t *cnt = tp_lds + TP_CHUNK * 64;                   // [n_blk] transitions before the block (after the scan)
    int32_t *cin = reinterpret_cast<int32_t *>(cnt + n_blk);  // [n_blk + 1] the true c at the block's first sample
    uint32_t *lastsign = cnt + 2 * n_blk + 1;                 // [n_blk] sign of the block's last valid sample
    uint32_t *pack = lastsign + n_blk;                        // [n_seg][PACK_STRIDE] toggle words
    uint32_t *wsum = pack + n_seg * PACK_STRIDE;              // [TP_WAVES] the scan's per-wave totals
    const int c = (int) blockIdx.x;                           // the channel
    // (readfirstlane: the wave index IS uniform, but only this tells the compiler, and everything "scalar" below hangs on it)
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), lane = (int) (threadIdx.x & 63);
    const uint32_t q = pllinc / 16u;                          // receiver.c:84,115,117
    const uint32_t pll0 = pllst[c] & 0xffffu;                 // receiver.h:40
    const uint32_t prev0 = prevst[c] & 1u;                    // receiver.h:44

    for (int i = (int) threadIdx.x; i < n_seg * PACK_STRIDE; i += 64 * TP_WAVES) pack[i] = 0;

    // ---- pre-pass, lane = block: the number of transitions and the sign of the block's last sample (a block's
    // transition bits need the sign before it: the newest bit of the word before), then an exclusive scan of the counts
    // over the blocks (their parity is the parity of c): wave scans + the waves' totals.  n_blk <= 1024 (launcher).
    {
        const int b = (int) threadIdx.x;
        uint32_t v = 0;
        if (b < n_blk) {
            uint32_t carry = prev0;
            if (b > 0) carry = sgn[sgn_index(8 * b - 1, N, c)] & 1u;         // bit 0 = newest sample of the word before
            uint64_t d[4];
            lastsign[b] = tp_block_bits<false>(sgn, N, c, b, L, carry, d);
            v = (uint32_t) tp_popc(d);
        }
        uint32_t inc = v;                                     // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = (uint32_t) __shfl_up((int) inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (b < n_blk) cnt[b] = before + inc - v;
        if (threadIdx.x == 0) cin[0] = 0;                     // c = 0 at the call's first sample, by definition
    }
    __syncthreads();

    // ---- chunks: pass 1 (all waves), walk (one wave)
    for (int b0 = 0; b0 < n_blk; b0 += TP_CHUNK) {
        const int b1 = b0 + TP_CHUNK < n_blk ? b0 + TP_CHUNK : n_blk;
        const int32_t cc = cin[b0];                           // the true c at the chunk's first sample: the centre
        const uint32_t cnt0 = cnt[b0];
        for (int b = b0 + wave; b < b1; b += TP_WAVES) {
            // (LDS reads land in vector registers: readfirstlane keeps the block's bit scan scalar)
            const uint32_t carry = (uint32_t) __builtin_amdgcn_readfirstlane((int) (b > 0 ? lastsign[b - 1] : prev0));
            uint64_t d[4];
            (void) tp_block_bits<true>(sgn, N, c, b, L, carry, d);
            // candidates of the block's parity: c_in = centre + (transitions between the chunk's start and the block) % 2
            // + 2 (lane - 32)
            const int32_t odd = (int32_t) ((cnt[b] - cnt0) & 1u);
            const uint32_t base = pll0 + (uint32_t) (b * TP_BLK) * pllinc;
            tab[(b - b0) * 64 + lane] = tp_run_block(d, base, pllinc, q, cc + odd + 2 * (lane - TP_HALF));
        }
        __syncthreads();
        if (wave == 0) {
            // the walk: uniform over the wave (every lane carries the same value); one dependent LDS read per block --
            // the blocks' parities come as a ballot, lane i = block b0 + i
            const unsigned long long oddmask = __ballot(b0 + lane < b1 && ((cnt[b0 + lane < b1 ? b0 + lane : b0] - cnt0) & 1u));
            int32_t cur = cc, res = 0;                         // res: lane i collects c at the END of block b0 + i
            for (int i = 0; i < b1 - b0; ++i) {
                const int32_t odd = (int32_t) ((oddmask >> i) & 1ull);
                const int32_t off = cur - (cc + odd - 2 * TP_HALF);             // = 2 k for candidate k
                if (off >= 0 && off < 128 && !(off & 1)) {
                    cur = tab[i * 64 + (off >> 1)];
                } else {
                    // outside the window: the block itself, from the true value (all lanes alike)
                    const int b = b0 + i;
                    const uint32_t carry = (uint32_t) __builtin_amdgcn_readfirstlane((int) (b > 0 ? lastsign[b - 1] : prev0));
                    uint64_t d[4];
                    (void) tp_block_bits<true>(sgn, N, c, b, L, carry, d);
                    cur = tp_run_block(d, pll0 + (uint32_t) (b * TP_BLK) * pllinc, pllinc, q, cur);
                }
                cur = __builtin_amdgcn_readfirstlane(cur);
                res = lane == i ? cur : res;
            }
            if (b0 + lane < b1) cin[b0 + lane + 1] = res;
        }
        __syncthreads();
    }

    // ---- pass 3: lane = block.  Toggle bit floor(U(t) / 2^16) - (slices before the segment) of the segment's pack for
    // every transition (receiver.c:124-132 restated, see the header).
    for (int b = (int) threadIdx.x; b < n_blk; b += 64 * TP_WAVES) {
        const int s = b / TP_SEG_BLKS;
        // slices before the segment's first sample: its unwrapped phase there >> 16
        const uint32_t Useg = pll0 + (uint32_t) (s * SEG_LEN) * pllinc + q * (uint32_t) cin[s * TP_SEG_BLKS];
        const uint32_t sig0 = Useg >> 16;
        const uint32_t carry = b > 0 ? lastsign[b - 1] : prev0;
        uint64_t d[4];
        (void) tp_block_bits<false>(sgn, N, c, b, L, carry, d);
        uint32_t Kq = q * (uint32_t) cin[b];
        const uint32_t base = pll0 + (uint32_t) (b * TP_BLK) * pllinc;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint64_t m = d[w];
            while (m) {
                const int pos = __builtin_clzll(m);
                m &= ~(0x8000000000000000ull >> pos);
                const uint32_t U = base + (uint32_t) (64 * w + pos) * pllinc + Kq;
                const uint32_t idx = (U >> 16) - sig0;                             // < 512: a pack has PACK_STRIDE words
                atomicXor(&pack[s * PACK_STRIDE + (idx >> 5)], 1u << (idx & 31u));
                Kq = ((U >> 15) & 1u) ? Kq - q : Kq + q;
            }
        }
    }
    __syncthreads();

    // ---- the packs leave: one lane per segment forms its words, lane 0 carries the parity from segment to segment
    // (pll_h3.hip's pack writer: a transition after a segment's last slice toggles the first bit of the next segment that
    // has one, or of a later call)
    uint32_t *nbs = reinterpret_cast<uint32_t *>(tab);        // [n_seg] bits of the segment; [n_seg .. 2 n_seg) its pending toggle
    const uint32_t Kend = q * (uint32_t) cin[n_blk];
    for (int s = (int) threadIdx.x; s < n_seg; s += 64 * TP_WAVES) {
        const uint32_t U0 = pll0 + (uint32_t) (s * SEG_LEN) * pllinc + q * (uint32_t) cin[s * TP_SEG_BLKS];
        const int e = (s + 1) * SEG_LEN < L ? (s + 1) * SEG_LEN : L;
        const uint32_t U1 = pll0 + (uint32_t) e * pllinc +
                            ((s + 1) * TP_SEG_BLKS < n_blk ? q * (uint32_t) cin[(s + 1) * TP_SEG_BLKS] : Kend);
        const uint32_t nb = (U1 >> 16) - (U0 >> 16);
        nbs[s] = nb;
        uint32_t pd = 0;
#pragma unroll
        for (int w = 0; w < PACK_STRIDE; ++w) {
            const uint32_t tg = pack[s * PACK_STRIDE + w];
            const int k = (int) nb - 32 * w;
            pack[s * PACK_STRIDE + w] = ~tg & (k >= 32 ? ~0u : k > 0 ? (1u << k) - 1u : 0u);
            if (k >= 0 && k < 32) pd = (tg >> k) & 1u;
        }
        nbs[n_seg + s] = pd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t par = (lastbit[c] ^ prev0) & 1u;
        for (int s = 0; s < n_seg; ++s) {
            const uint32_t nb = nbs[s], pd = nbs[n_seg + s];
            if (nb) {
                pack[s * PACK_STRIDE] ^= par;
                par = pd;
            } else {
                par ^= pd;
            }
        }
        const uint32_t Uend = pll0 + (uint32_t) L * pllinc + Kend;
        pllst[c] = Uend & 0xffffu;                            // receiver.c:133
        prevst[c] = lastsign[n_blk - 1];
        lastbit[c] = (lastsign[n_blk - 1] ^ par) & 1u;
    }
    __syncthreads();
    for (int i = (int) threadIdx.x; i < n_seg * PACK_STRIDE; i += 64 * TP_WAVES)
        segbits[((size_t) c * n_seg_alloc + (size_t) (i / PACK_STRIDE)) * PACK_STRIDE + (size_t) (i % PACK_STRIDE)] = pack[i];
    for (int s = (int) threadIdx.x; s < n_seg_alloc; s += 64 * TP_WAVES)
        segcnt[(size_t) c * n_seg_alloc + s] = s < n_seg ? nbs[s] : 0u;
}


// The time-parallel form is exact for any input; these are the limits of THIS implementation: whole 32-bit unwrapped
// phase (U = pll0 + L * pllinc + q * c < 2^32 with |c| <= L), its LDS tables, and a call long enough to be worth it.
bool pll_tp_applicable(const PllLaunch &a)
{
    const int n_blk = (a.L + TP_BLK - 1) / TP_BLK;
    const size_t lds = sizeof(uint32_t) * ((size_t) TP_CHUNK * 64 + 3 * (size_t) n_blk + 1 + (size_t) a.n_seg * PACK_STRIDE + TP_WAVES);
    const unsigned long long reach = 65536ull + (unsigned long long) a.L * a.pllinc + (unsigned long long) a.L * (a.pllinc / 16u);
    return a.L >= TP_BLK && n_blk <= 64 * TP_WAVES && reach < 0x80000000ull && lds <= 150 * 1024 && a.pllinc >= 16 &&
           2 * a.n_seg <= TP_CHUNK * 64;
}

hipError_t launch_pll_tp(const PllLaunch &a, hipStream_t stream)
{
    if (!pll_tp_applicable(a)) return hipErrorInvalidValue;
    const int n_blk = (a.L + TP_BLK - 1) / TP_BLK;
    const int n_seg = (((a.L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
    const size_t lds = sizeof(uint32_t) * ((size_t) TP_CHUNK * 64 + 3 * (size_t) n_blk + 1 + (size_t) n_seg * PACK_STRIDE + TP_WAVES);
    static bool raised[64] = {};
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void *) pll_tp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised[dev] = true;
    }
    hipLaunchKernelGGL(pll_tp_kernel, dim3(a.N), dim3(64 * TP_WAVES), lds, stream, a.sgn, a.pll, a.prev, a.lastbit,
                       a.segbits, a.segcnt, a.N, a.L, a.n_seg, a.pllinc);
    return hipGetLastError();
}

} // namespace gnuais
