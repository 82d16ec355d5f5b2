// pll_tp.hip -- K2, TIME-PARALLEL form: bit-clock recovery PLL, slice and NRZI decode (gnuais src/receiver.c:109-135)
// for batches too small to fill the chip with one lane per channel (BASELINE C2: 256 channels = 4 workgroups of the
// lane-per-channel kernel on 256 CUs, 0.31 ms of pure recurrence latency per 48 000 samples).
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
//   counts  LANE = BLOCK of 256 samples: the block's transition bits (oldest sample in bit 0) go to LDS once, with
//           their number (a scan over the blocks gives every block's parity) and the block's last sign.
//   pass 1  one WAVE per block, LANE = CANDIDATE: the 64 lanes run the block from the 64 values
//           c_in = centre + 2 (k - 32) of the right parity.  The transitions of the block are the same for every lane,
//           so the scan of the bits is scalar code and a step is four vector instructions.  Result: a table in LDS --
//           the block's map on a window of +-64 around the centre, stored as the NEXT block's table offset.
//   walk    one wave goes through the blocks of a chunk in order, v <- table[block][v]: one dependent LDS read per
//           block instead of 80 dependent steps.  A value that leaves the window (c diffuses by one nudge per transition
//           in noise and steps by up to 40 when a burst is acquired) ENDS the chunk there: the next chunk starts at
//           that block, centred on its true value, so the walk always advances by at least one block per chunk.
//   pass 3  LANE = BLOCK: every block runs once more from its now known c_in, per-lane bit scan, and XORs its toggles
//           into the segment's bit pack in LDS; then the packs leave as the lane-per-channel kernel writes them
//           (complement, trim, parity carried from segment to segment and call to call: pll_h3.hip's pack writer).
//
// Exact for every input: pass 1 computes the true block map on its window, the walk composes maps inside their windows
// only, pass 3 is the reference's loop restricted to one block.  Work is 64 x the serial kernel's vector instructions
// (that is the price of the candidates), which a small batch has room for: 256 channels x 7 500 transitions x 64
// lanes = 2 M wave-steps on 1024 SIMDs.  launch_pll() takes this form for small batches only.
// Round 5 (profiles/r05_pll_tp_budget.txt): the launch was 0.125-0.145 ms where a workgroup took 0.059 -- the walk ran
// every block whose value had left the window BY ITSELF (1.35 us each), 20-60 blocks in the 18 of 256 channels whose
// count stepped twice in a chunk of 64 blocks, and the launch ended with them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"

namespace gnuais {

namespace {

constexpr int TP_BLK = 256;                 // samples per block = 8 sign words = two 16-byte pieces of sgn_index()
constexpr int TP_SEG_BLKS = SEG_LEN / TP_BLK;
constexpr int TP_WAVES = 16;                // waves of a workgroup (one channel)
constexpr int TP_HALF = 32;                 // candidates on either side of the centre (in steps of 2)
constexpr int TP_CHUNK = 32;                // blocks per chunk: one table, one walk, one re-centring (16 and 64 measured: level / slower)

// the 256 transition bits of block b of channel c as four 64-bit words, OLDEST sample in bit 0 of d[0]; `carry` = the
// sign of the sample before the block; returns the sign of the block's last valid sample.  receiver.c:113 curr != prev.
__device__ __forceinline__ uint32_t tp_block_bits(const uint32_t *__restrict__ sgn, int N, int c, int b, int L,
                                                  uint32_t carry, uint32_t R[8])
{
    const uint4 p0 = *reinterpret_cast<const uint4 *>(sgn + ((size_t) (2 * b) * (size_t) N + (size_t) c) * 4);
    const uint4 p1 = *reinterpret_cast<const uint4 *>(sgn + ((size_t) (2 * b + 1) * (size_t) N + (size_t) c) * 4);
    const uint32_t S[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const int nv = L - b * TP_BLK;          // valid samples of this block (>= 1)
    uint32_t prev = carry & 1u;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int k = nv - 32 * w;          // valid samples of this word
        uint32_t x = S[w] ^ ((S[w] >> 1) | (prev << 31));       // bit 31 = the word's oldest sample
        if (k <= 0) {
            x = 0;
        } else if (k < 32) {
            x &= ~0u << (32 - k);
            prev = (S[w] >> (32 - k)) & 1u;
        } else {
            prev = S[w] & 1u;
        }
        R[w] = __brev(x);                   // bit j = sample 32 w + j of the block
    }
    return prev;
}

// one block from c_in (per lane), transition bits m[] (uniform, oldest sample in bit 0): returns q * (c_out - c_in) + 256 *
// pllinc.  base = pll0 + (first sample of the block) * pllinc, all mod 2^32 (U < 2^31 is checked by the launcher).
// receiver.c:113-118.  The lane carries A = base + q * c, so that the phase at the transition at position p of the
// block is ONE v_mad_u32_u24: U = p * pllinc + A, with p in a scalar register; bit 15 of U is the sign of its low half
// (a 16-bit compare), which picks +q or -q.  Four vector instructions, of which three are two-operand ones (issued at
// twice the rate of the three-operand forms: profiles/r05_ubench_valu_op_rates.txt), and four scalar ones a
// transition: sixteen waves share their CU's scalar unit, which is why the position's product is kept OFF it
// (`inc_v` is pllinc in a vector register; the empty asm hides from the compiler that it is uniform).
__device__ __forceinline__ uint32_t tp_run_block(const uint32_t Rw[8], uint32_t base, uint32_t pllinc, uint32_t q, int32_t c_in)
{
    uint32_t inc_v = pllinc, qp = q, qm = 0u - q;
    asm volatile("" : "+v"(inc_v), "+v"(qp), "+v"(qm));
    const uint32_t A0 = base + q * (uint32_t) c_in;
    uint32_t A = A0, U, T;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        // positions inside word w count from its first sample: A carries the word's offset while the word lasts
        // (the scan's words must BE in scalar registers for the asm below, not merely uniform)
        uint64_t m = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) Rw[2 * w + 1]) << 32) |
                     (uint32_t) __builtin_amdgcn_readfirstlane((int) Rw[2 * w]);
        int p;
        asm volatile(
            "s_cmp_eq_u64 %[m], 0\n\t"
            "s_cbranch_scc1 2f\n\t"
            "1:\n\t"
            "s_ff1_i32_b64 %[p], %[m]\n\t"                    // the oldest transition left
            "s_bitset0_b64 %[m], %[p]\n\t"
            "v_mad_u32_u24 %[U], %[p], %[inc], %[A]\n\t"      // the phase there (receiver.c:114 reads its bit 15)
            "v_cmp_gt_i16 vcc, 0, %[U]\n\t"                   // pll >= 0x8000
            "s_cmp_lg_u64 %[m], 0\n\t"
            "v_cndmask_b32 %[T], %[qp], %[qm], vcc\n\t"       // + q (:115) or - q (:117)
            "v_add_u32 %[A], %[A], %[T]\n\t"
            "s_cbranch_scc1 1b\n\t"
            "2:\n\t"
            : [m] "+s"(m), [A] "+v"(A), [U] "=&v"(U), [T] "=&v"(T), [p] "=&s"(p)
            : [inc] "v"(inc_v), [qp] "v"(qp), [qm] "v"(qm)
            : "scc", "vcc");
        A += 64u * pllinc;
    }
    return A - A0;
}

} // namespace

// (A measurement build -- 100 MHz stamps of a workgroup's phases -- produced profiles/r05_pll_tp_budget.txt; the
// instrumentation is in git history, commit b1fc8e6.)

// One workgroup per channel.  LDS: the chunk's table, the blocks' transition bits, per-block transition counts / true
// c_in / last signs, the bit packs.  CHUNK: blocks per chunk (one table, one walk, one re-centring), at most 64.
template <int CHUNK>
__global__ __launch_bounds__(64 * TP_WAVES) void pll_tp_kernel(
    const uint32_t *__restrict__ sgn, uint32_t *__restrict__ pllst, uint32_t *__restrict__ prevst,
    uint32_t *__restrict__ lastbit, uint32_t *__restrict__ segbits, uint32_t *__restrict__ segcnt,
    int N, int L, int n_seg_alloc, uint32_t pllinc)
{
    static_assert(CHUNK >= 4 && CHUNK <= 64, "one wave walks a chunk");
    extern __shared__ uint32_t tp_lds[];
    const int n_blk = (L + TP_BLK - 1) / TP_BLK;
    const int n_seg = (((L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
    int32_t *tab = reinterpret_cast<int32_t *>(tp_lds);       // [64][64] (CHUNK rows in use) table offset of c_out in the NEXT block's row
    uint32_t *bits = tp_lds + 64 * 64;                        // [n_blk][8] transition bits, oldest sample in bit 0 of word 0
    uint32_t *cnt = bits + 8 * n_blk;                         // [n_blk + 1] transitions before the block (after the scan)
    int32_t *cin = reinterpret_cast<int32_t *>(cnt + n_blk + 1);   // [n_blk + 1] the true c at the block's first sample
    uint32_t *lastsign = cnt + 2 * n_blk + 2;                 // [n_blk] sign of the block's last valid sample
    uint32_t *pack = lastsign + n_blk;                        // [n_seg][PACK_STRIDE] toggle words
    uint32_t *wsum = pack + n_seg * PACK_STRIDE;              // [TP_WAVES] the scan's per-wave totals
    uint32_t *ctl = wsum + TP_WAVES;                          // [1] the block the next chunk starts at
    const int c = (int) blockIdx.x;                           // the channel
    // (readfirstlane: the wave index IS uniform, but only this tells the compiler, and everything "scalar" below hangs on it)
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), lane = (int) (threadIdx.x & 63);
    const uint32_t q = pllinc / 16u;                          // receiver.c:84,115,117
    const float rq = 1.0f / (float) q;
    const uint32_t pll0 = pllst[c] & 0xffffu;                 // receiver.h:40
    const uint32_t prev0 = prevst[c] & 1u;                    // receiver.h:44

    for (int i = (int) threadIdx.x; i < n_seg * PACK_STRIDE; i += 64 * TP_WAVES) pack[i] = 0;

    // ---- counts, lane = block: the transition bits (a block's bits need the sign before it: the newest bit of the word
    // before), their number and the sign of the block's last sample; then an exclusive scan of the numbers over the
    // blocks (their parity is the parity of c): wave scans + the waves' totals.  n_blk <= 1024 (launcher).
    {
        const int b = (int) threadIdx.x;
        uint32_t v = 0;
        if (b < n_blk) {
            uint32_t carry = prev0;
            if (b > 0) carry = sgn[sgn_index(8 * b - 1, N, c)] & 1u;         // bit 0 = newest sample of the word before
            uint32_t R[8];
            lastsign[b] = tp_block_bits(sgn, N, c, b, L, carry, R);
            *reinterpret_cast<uint4 *>(bits + 8 * b) = make_uint4(R[0], R[1], R[2], R[3]);
            *reinterpret_cast<uint4 *>(bits + 8 * b + 4) = make_uint4(R[4], R[5], R[6], R[7]);
#pragma unroll
            for (int w = 0; w < 8; ++w) v += (uint32_t) __popc(R[w]);
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
        if (b == n_blk - 1) cnt[n_blk] = before + inc;
        if (threadIdx.x == 0) cin[0] = 0;                     // c = 0 at the call's first sample, by definition
    }
    __syncthreads();

    // ---- chunks: pass 1 (all waves), walk (one wave).  Block b's candidates are lo(b) + 2 k, lo(b) = centre + (transitions
    // between the chunk's first sample and the block) % 2 - 64: the parity c has there.
    int b0 = 0;
    while (b0 < n_blk) {
        const int b1 = b0 + CHUNK < n_blk ? b0 + CHUNK : n_blk;
        const int32_t cc = cin[b0];                           // the true c at the chunk's first sample: the centre
        const uint32_t cnt0 = cnt[b0];
        for (int b = b0 + wave; b < b1; b += TP_WAVES) {
            const uint4 r0 = *reinterpret_cast<const uint4 *>(bits + 8 * b), r1 = *reinterpret_cast<const uint4 *>(bits + 8 * b + 4);
            const uint32_t Rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const int32_t lo = cc + (int32_t) ((cnt[b] - cnt0) & 1u) - 2 * TP_HALF;
            const int32_t lo_next = cc + (int32_t) ((cnt[b + 1] - cnt0) & 1u) - 2 * TP_HALF;
            const int32_t c_in = lo + 2 * lane;
            const uint32_t base = pll0 + (uint32_t) (b * TP_BLK) * pllinc;
            // q * (c_out - c_in) = (what the lane's A moved by) - 256 * pllinc, |c_out - c_in| <= 256: exact in fp32
            const int32_t moved = (int32_t) (tp_run_block(Rw, base, pllinc, q, c_in) - 256u * pllinc);
            const int32_t c_out = c_in + __float2int_rn((float) moved * rq);
            tab[(b - b0) * 64 + lane] = 2 * (c_out - lo_next);          // byte offset in the next row when inside [0, 256)
        }
        __syncthreads();
        if (wave == 0) {
            // the walk: uniform over the wave (every lane carries the same value), one dependent LDS read per block;
            // lane i collects the value at the END of block b0 + i
            int32_t v = 4 * TP_HALF, res = 0;                 // c = cc is candidate 32 of the chunk's first block
            int i = 0;
            for (; i < b1 - b0; ++i) {
                if ((uint32_t) v >= 256u) break;              // outside the window: the next chunk starts at this block
                v = tab[i * 64 + (v >> 2)];
                res = lane == i ? v : res;
            }
            // blocks b0 .. b0 + i - 1 are done (i >= 1): c at the first sample of block b0 + j + 1 = lo(b0 + j + 1) + res / 2
            if (lane < i)
                cin[b0 + lane + 1] = cc + (int32_t) ((cnt[b0 + lane + 1] - cnt0) & 1u) - 2 * TP_HALF + (res >> 1);
            if (lane == 0) ctl[0] = (uint32_t) (b0 + i);
        }
        __syncthreads();
        b0 = __builtin_amdgcn_readfirstlane((int) ctl[0]);
    }

    // ---- pass 3: lane = block.  Toggle bit floor(U(t) / 2^16) - (slices before the segment) of the segment's pack for
    // every transition (receiver.c:124-132 restated, see the header).
    for (int b = (int) threadIdx.x; b < n_blk; b += 64 * TP_WAVES) {
        const int s = b / TP_SEG_BLKS;
        // slices before the segment's first sample: its unwrapped phase there >> 16
        const uint32_t Useg = pll0 + (uint32_t) (s * SEG_LEN) * pllinc + q * (uint32_t) cin[s * TP_SEG_BLKS];
        const uint32_t sig0 = Useg >> 16;
        const uint4 r0 = *reinterpret_cast<const uint4 *>(bits + 8 * b), r1 = *reinterpret_cast<const uint4 *>(bits + 8 * b + 4);
        const uint32_t Rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        uint32_t Kq = q * (uint32_t) cin[b];
        const uint32_t base = pll0 + (uint32_t) (b * TP_BLK) * pllinc;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t m = Rw[w];
            while (m) {
                const int pos = __builtin_ctz(m);
                m &= m - 1u;
                const uint32_t U = base + (uint32_t) (32 * w + pos) * pllinc + Kq;
                const uint32_t idx = (U >> 16) - sig0;                             // < 512: a pack has PACK_STRIDE words
                atomicXor(&pack[s * PACK_STRIDE + (idx >> 5)], 1u << (idx & 31u));
                Kq = ((U >> 15) & 1u) ? Kq - q : Kq + q;
            }
        }
    }
    __syncthreads();

    // ---- the packs leave: one lane per segment forms its words, then the parity is carried from segment to segment
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
    if (wave == 0) {
        // lane = segment, 64 at a time: which segments have bits and which hand a toggle on, as two masks; the carried
        // parity then runs along the masks in scalar code
        uint32_t par = (lastbit[c] ^ prev0) & 1u;
        for (int s0 = 0; s0 < n_seg; s0 += 64) {
            const int s = s0 + lane;
            const unsigned long long has = __ballot(s < n_seg && nbs[s < n_seg ? s : 0] != 0u);
            const unsigned long long pdm = __ballot(s < n_seg && nbs[n_seg + (s < n_seg ? s : 0)] != 0u);
            unsigned long long flip = 0;                      // segments whose first bit is toggled
            const int n = n_seg - s0 < 64 ? n_seg - s0 : 64;
            for (int j = 0; j < n; ++j) {
                const uint32_t pd = (uint32_t) (pdm >> j) & 1u;
                if ((has >> j) & 1ull) {
                    flip |= (unsigned long long) par << j;
                    par = pd;
                } else {
                    par ^= pd;
                }
            }
            if (s < n_seg && ((flip >> lane) & 1ull)) pack[s * PACK_STRIDE] ^= 1u;
        }
        if (lane == 0) {
            const uint32_t Uend = pll0 + (uint32_t) L * pllinc + Kend;
            pllst[c] = Uend & 0xffffu;                        // receiver.c:133
            prevst[c] = lastsign[n_blk - 1];
            lastbit[c] = (lastsign[n_blk - 1] ^ par) & 1u;
        }
    }
    __syncthreads();
    for (int i = (int) threadIdx.x; i < n_seg * PACK_STRIDE; i += 64 * TP_WAVES)
        segbits[((size_t) c * n_seg_alloc + (size_t) (i / PACK_STRIDE)) * PACK_STRIDE + (size_t) (i % PACK_STRIDE)] = pack[i];
    for (int s = (int) threadIdx.x; s < n_seg_alloc; s += 64 * TP_WAVES)
        segcnt[(size_t) c * n_seg_alloc + s] = s < n_seg ? nbs[s] : 0u;
}


namespace {

size_t tp_lds_bytes(int n_blk, int n_seg)
{
    return sizeof(uint32_t) * ((size_t) 64 * 64 + 8 * (size_t) n_blk + 3 * (size_t) n_blk + 2 + (size_t) n_seg * PACK_STRIDE + TP_WAVES + 4);
}

} // namespace

// The time-parallel form is exact for any input; these are the limits of THIS implementation: whole 31-bit unwrapped
// phase (U = pll0 + L * pllinc + q * c < 2^31 with |c| <= L), 256 q < 2^24 (a block's q * (c_out - c_in) is divided in
// fp32), its LDS tables, and a call long enough to be worth it.
bool pll_tp_applicable(const PllLaunch &a)
{
    const int n_blk = (a.L + TP_BLK - 1) / TP_BLK;
    const unsigned long long reach = 65536ull + (unsigned long long) a.L * a.pllinc + (unsigned long long) a.L * (a.pllinc / 16u);
    return a.L >= TP_BLK && n_blk <= 64 * TP_WAVES && reach < 0x80000000ull && tp_lds_bytes(n_blk, a.n_seg) <= 150 * 1024 &&
           a.pllinc >= 16 && a.pllinc < (1u << 20) && 2 * a.n_seg <= 64 * 64;
}

hipError_t launch_pll_tp(const PllLaunch &a, hipStream_t stream)
{
    if (!pll_tp_applicable(a)) return hipErrorInvalidValue;
    const int n_blk = (a.L + TP_BLK - 1) / TP_BLK;
    const int n_seg = (((a.L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
    const size_t lds = tp_lds_bytes(n_blk, n_seg);
    auto kern = pll_tp_kernel<TP_CHUNK>;
    static bool raised[64] = {};
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.N), dim3(64 * TP_WAVES), lds, stream, a.sgn, a.pll, a.prev, a.lastbit,
                       a.segbits, a.segcnt, a.N, a.L, a.n_seg, a.pllinc);
    return hipGetLastError();
}

} // namespace gnuais
