// hdlc_events.hip -- K2b, event-driven: the HDLC deframer of protodec_decode() (gnuais
// src/protodec.c:988-1122) for a whole batch of channels, one lane per channel.
//
// The reference is a five-state machine fed one bit at a time, and hdlc_crc.hip restates it as
// closed-form steps over 32-bit windows, one mode (hunting / flag matching / inside a frame) per
// step.  A wave executes the union of what its lanes do, so with 64 channels in three different
// modes every step ran all three bodies.  Here the data-dependent work is done once per bit pack
// for all lanes alike -- bitmaps of where anything can happen at all -- and the machine then JUMPS
// from event to event: one loop turn takes a lane from hunting through the flag into the frame
// and on to its closing flag when all of that lies inside the pack.
//
// Bitmaps of a pack (bit k of a bitmap <-> bit k of the pack; x[-1] = the bit before the pack):
//   NA  x[k] == x[k-1]                              -- where an alternating run breaks
//   CB  x[k] == 0 and x[k-14..k] all alternate      -- "antallpreamble > 14 && in == 0"
//       (protodec.c:1036) for a count that started inside the pack
//   E6  x[k-5..k] are all 1                         -- the sixth 1: the frame's closing flag
//       (protodec.c:996-1001 after :1011-1015)
//   SF  x[k] == 0 after exactly five 1s             -- a stuffed 0, not stored (:1002-1006)
// They are exact because the machine's `last` is always the previous bit (protodec.c:1119) and
// because a frame starts after a 0 and never contains six 1s in a row; what the machine carries
// into the pack (antallpreamble for a hunt in progress, the 1s counted so far inside a frame) is
// folded in where it matters.  Everything the bitmaps cannot decide in closed form -- a flag that
// straddles the pack's end, the 449-bit overflow -- goes bit by bit through a literal transcription
// of the reference (slow_bit), so that the state handed to the next pack is the reference's in
// every case.
//
// Candidate records, counters and the state words are exactly those of hdlc_deframe_kernel
// (hdlc_crc.hip); K3 does not know which of the two ran.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gnuais {

namespace {

enum { ST_SKURR = 1, ST_PREAMBLE = 2, ST_STARTSIGN = 3, ST_DATA = 4, ST_STOPSIGN = 5 };
constexpr uint32_t CAND_VALID = 0x10000u;
constexpr int EW = PACK_STRIDE;                 // words per pack
constexpr int NONE = 1 << 20;                   // "no such position"

__device__ __forceinline__ uint32_t lowmask(int k) { return k >= 32 ? ~0u : k <= 0 ? 0u : ((1u << k) - 1u); }
__device__ __forceinline__ int ctz32(uint32_t v) { return v ? __ffs((int) v) - 1 : 32; }
__device__ __forceinline__ int clz32(uint32_t v) { return v ? __clz((int) v) : 32; }
// max over the workgroup's lanes (one wave, possibly fewer than 64 lanes)
__device__ __forceinline__ int wave_max_i(int v, int lanes, int lane)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const int u = __shfl_xor(v, o);
        v = ((lane ^ o) < lanes && u > v) ? u : v;
    }
    return v;
}

} // namespace

// LDS per lane (stride = lanes): X[EW + 2] (X[0] = the word before the pack, X[EW + 1] = 0),
// NA[EW], CB[EW], E6[EW], SF[EW], PS[EW + 1] (stuffed 0s before word q)
constexpr int EV_ROWS = (EW + 2) + 4 * EW + (EW + 1);

// (the kernel takes the registers it wants -- 120; held to 96 / 80 / 72 it spilt and lost: profiles/r04_deframer_bitmaps_and_stage_masks.txt)
// TPB: lanes per workgroup when the launcher knows them (8 / 16 / 32 / 64: every LDS row offset is then an immediate of
// its ds instruction); 0: taken from blockDim
template <int TPB>
__global__ __launch_bounds__(64) void hdlc_events_kernel(
    const uint32_t *__restrict__ segbits, const uint32_t *__restrict__ segcnt,
    uint32_t *__restrict__ ctl, uint32_t *__restrict__ cand, uint32_t *__restrict__ cand_first,
    uint32_t *__restrict__ cand_count, int32_t *__restrict__ counters,
    uint32_t *__restrict__ flags, int N, int n_seg, int seg_words, int K)
{
    __builtin_amdgcn_s_setprio(3);      // latency-bound chain: take every issue slot it can use (priority 0 / 1: level, profiles/r05_pll_h3_in_the_pipeline.txt)
    extern __shared__ uint32_t lds[];
    const int tpb = TPB > 0 ? TPB : (int) blockDim.x, tx = (int) threadIdx.x;
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = cg < N;
    const size_t c = (size_t) (live ? cg : N - 1), n_ = (size_t) N;
    uint32_t *const Xa = lds + tx, *const NAa = Xa + (EW + 2) * tpb, *const CBa = NAa + EW * tpb,
                   *const E6a = CBa + EW * tpb, *const SFa = E6a + EW * tpb, *const PSa = SFa + EW * tpb;
#define XW(q) Xa[((q) + 1) * tpb]               /* pack word q, q = -1 .. EW */

    const uint32_t c0 = ctl[c];
    int state = c0 & 7, nstartsign = (c0 >> 3) & 15, antallpreamble = (c0 >> 7) & 15;
    int antallenner = (c0 >> 11) & 7, bitstuff = (c0 >> 14) & 1;
    uint32_t last = (c0 >> 15) & 1;
    int bufferpos = (c0 >> 16) & 511;
    uint32_t cur = ctl[n_ + c];
    const uint32_t seen0 = ctl[2 * n_ + c], seenhi0 = ctl[5 * n_ + c];
    uint32_t nstart = ctl[3 * n_ + c];
    int rawpos = (int) ctl[4 * n_ + c];
    int lost2 = 0;

    const bool open0 = (state == ST_DATA || state == ST_STOPSIGN);
    const uint32_t first = nstart - (open0 ? 1u : 0u);      // first slot this call may close
    uint32_t *rec = cand + ((size_t) c * K + (open0 ? (nstart - 1) % (uint32_t) K : 0u)) * CAND_WORDS;
    bool rec_ok = open0;
    uint32_t seenbase = seen0;          // bits fed before the current pack

#define HDLC_RESET()                                                                    \
    do { state = ST_SKURR; nstartsign = 0; antallpreamble = 0; antallenner = 0;        \
         last = 0; bitstuff = 0; bufferpos = 0; } while (0)
    // protodec.c:1076-1082: the 0 after the flag's six 1s
#define ENTER_DATA()                                                                    \
    do { state = ST_DATA; nstartsign = 1; antallenner = 0; antallpreamble = 0;         \
         bufferpos = 0; cur = 0; rawpos = 0; bitstuff = 0; last = 0;                    \
         rec_ok = (nstart - first) < (uint32_t) K;                                      \
         rec = cand + ((size_t) c * K + nstart % (uint32_t) K) * CAND_WORDS;            \
         if (rec_ok) rec[0] = 0; else flags[1] = 1;                                     \
         ++nstart; } while (0)
    // one raw bit / a run of raw bits into the open frame's record
#define RAW_APPEND(v_, n_bits_)                                                         \
    do { const uint32_t sb_ = (v_); const int nb_ = (n_bits_), bsh_ = rawpos & 31;      \
         const uint64_t t64_ = (uint64_t) sb_ << bsh_;                                  \
         cur |= (uint32_t) t64_;                                                        \
         if (bsh_ + nb_ >= 32) {                                                        \
             if (rec_ok) rec[CAND_HDR + (rawpos >> 5)] = cur;                           \
             cur = (uint32_t) (t64_ >> 32);                                             \
         }                                                                              \
         rawpos += nb_; } while (0)
    // protodec.c:1095-1115
#define STOP_BIT(x_, at_)                                                               \
    do { const int nb_ = bufferpos - 6 - 16;                                            \
         if ((x_) == 0 && nb_ > 0) {                                                    \
             if (rec_ok) {                                                              \
                 const uint32_t eb_ = seenbase + (uint32_t) (at_);                      \
                 const uint32_t ebhi_ = (seenhi0 + (eb_ < seen0 ? 1u : 0u)) & 31u;      \
                 rec[CAND_HDR + (rawpos >> 5)] = cur;                                   \
                 rec[1] = eb_;                                                          \
                 rec[0] = (uint32_t) nb_ | CAND_VALID | ((uint32_t) rawpos << 17) | (ebhi_ << 27); \
             }                                                                          \
         } else {                                                                       \
             ++lost2;                                                                   \
             if (rec_ok) {      /* not a candidate (no CAND_VALID); its bits stay on record for d->buffer */ \
                 rec[CAND_HDR + (rawpos >> 5)] = cur;                                   \
                 rec[0] = (uint32_t) rawpos << 17;                                      \
             }                                                                          \
         }                                                                              \
         HDLC_RESET();                                                                  \
         last = (x_); } while (0)

    // the reference, one bit (protodec.c:993-1120); x = in[i], at = its position in the pack
    // nstartsign and bufferpos go in and out BY VALUE (the inner block's locals shadow the captures, so the macros
    // below work on copies): captured by reference, these two stayed in private memory through every optimisation
    // pass -- private_seg_size 12, scratch loads and stores inside every event turn, each a trip through the vector
    // memory pipeline that the FIR beside this kernel keeps full (0.266 -> 0.248 ms alone, 0.495 -> 0.455 in the
    // pipeline).  The Makefile fails the build if this kernel's descriptor asks for scratch again.
    auto slow_bit = [&](uint32_t x, int at) {
        int ns_v = nstartsign, bp_v = bufferpos;
        {
        int nstartsign = ns_v, bufferpos = bp_v;
        switch (state) {
        case ST_DATA:
            if (bitstuff) {
                if (x == 1) {
                    state = ST_STOPSIGN;
                } else {
                    RAW_APPEND(0u, 1);                  // the stuffed 0 stays in the raw record
                }
                bitstuff = 0;
            } else {
                if (x == last && x == 1) {
                    if (++antallenner == 4) { bitstuff = 1; antallenner = 0; }
                } else {
                    antallenner = 0;
                }
                RAW_APPEND(x, 1);
                if (++bufferpos >= 449) {
                    if (rec_ok) rec[0] = 0;
                    HDLC_RESET();
                }
            }
            break;
        case ST_SKURR:
            antallpreamble = (x != last) ? (antallpreamble < 15 ? antallpreamble + 1 : 15) : 0;
            if (antallpreamble > 14 && x == 0) { state = ST_PREAMBLE; antallpreamble = 0; }
            break;
        case ST_PREAMBLE:
            if (x != last && nstartsign == 0) {
                antallpreamble = antallpreamble < 15 ? antallpreamble + 1 : 15;
            } else if (x == 1) {
                if (nstartsign == 0) nstartsign = 3;
                else if (nstartsign == 5) { nstartsign = 6; antallpreamble = 0; state = ST_STARTSIGN; }
                else ++nstartsign;
            } else {
                if (nstartsign == 0) nstartsign = 1; else HDLC_RESET();
            }
            break;
        case ST_STARTSIGN:
            if (nstartsign >= 7) {
                if (x == 0) { ENTER_DATA(); nstartsign = 0; } else HDLC_RESET();
            } else if (x == 0) {
                HDLC_RESET();
            }
            ++nstartsign;
            if (nstartsign > 15) nstartsign = 15;
            break;
        case ST_STOPSIGN:
            STOP_BIT(x, at);
            break;
        default:
            HDLC_RESET();
            break;
        }
        ns_v = nstartsign; bp_v = bufferpos;
        }
        nstartsign = ns_v; bufferpos = bp_v;
        last = x;
    };

    const uint32_t *__restrict__ rows = segbits + c * (size_t) n_seg * (size_t) PACK_STRIDE;
    uint32_t pf[EW];
    int pf_cnt = 0;
    bool pf_ready = false;                      // pf / pf_cnt hold the pack of the segment the loop is about to take
    auto fetch = [&](int seg) {
        const uint32_t *__restrict__ row = rows + (size_t) seg * (size_t) PACK_STRIDE;
        {
            pf_cnt = live ? (int) segcnt[c * (size_t) n_seg + seg] : 0;
#pragma unroll
            for (int q = 0; q < EW / 4; ++q) {
                const uint4 v = reinterpret_cast<const uint4 *>(row)[q];
                pf[4 * q] = v.x; pf[4 * q + 1] = v.y; pf[4 * q + 2] = v.z; pf[4 * q + 3] = v.w;
            }
        }
        pf_ready = true;
    };

    for (int seg = 0; seg < n_seg; ++seg) {
        if (!pf_ready) fetch(seg);
        int tile_end = pf_cnt;
        if (tile_end > seg_words * 32) tile_end = seg_words * 32;

        // ---- the pack's bitmaps, the same code for every lane ------------------------------------
        // the word before the pack, as far as it matters: inside a frame the 1s counted so far
        // (protodec.c:1011-1015: antallenner + 1 of them, five when bitstuff is up), else `last`
        const int m_in = (state == ST_DATA) ? (bitstuff ? 5 : (last ? antallenner + 1 : 0)) : 0;
        uint32_t prevw = (state == ST_DATA) ? (m_in ? ~0u << (32 - m_in) : 0u) : (last << 31);
        uint32_t ps = 0, nzNA = 0, nzCB = 0, nzE6 = 0;
        XW(-1) = prevw;
#pragma unroll
        for (int q = 0; q < EW; ++q) XW(q) = pf[q];
        pf_ready = false;
        if (seg + 1 < n_seg) fetch(seg + 1);    // the next pack is on its way while this one is walked
        const int nwq = __builtin_amdgcn_readfirstlane((int) wave_max_i((tile_end + 31) >> 5, tpb, tx));
        // Every "bit i of (v << d) is bit i-d of v" below takes the bits that come in from the word before out of THAT
        // word's value of the same quantity (one v_alignbit_b32 each), instead of carrying the chain through 64-bit
        // shifts and ANDs of {this word, the word before}: the same bits -- no term reaches further back than the word
        // before, and the word before the pack has nothing but zeros below it in either form -- for 45 instead of
        // 118 VALU instructions per word, three quarters of what this kernel issues.
        auto shl_in = [](uint32_t v, uint32_t below, int n) -> uint32_t {    // (v << n) | (below >> (32 - n)), 0 < n < 32
            return __builtin_amdgcn_alignbit(v, below, 32 - n);
        };
        uint32_t pA = 0, pa2 = 0, pa4 = 0, pa8 = 0;                          // nothing alternates before the pack (hunt0 has that)
        uint32_t pu2 = prevw & (prevw << 1);
        uint32_t pt = (pu2 & (pu2 << 2)) & (prevw << 4);
#pragma unroll
        for (int q = 0; q < EW; ++q) {
            if (q >= nwq) {                 // beyond every lane's bits: empty
                XW(q) = 0; NAa[q * tpb] = 0; CBa[q * tpb] = 0; E6a[q * tpb] = 0; SFa[q * tpb] = 0; PSa[q * tpb] = ps;
                continue;
            }
            const uint32_t vm = lowmask(tile_end - 32 * q);
            const uint32_t xw = XW(q) & vm;
            const uint32_t A = (xw ^ shl_in(xw, prevw, 1)) & vm;                // x[k] != x[k-1]
            const uint32_t a2 = A & shl_in(A, pA, 1), a4 = a2 & shl_in(a2, pa2, 2), a8 = a4 & shl_in(a4, pa4, 4);
            const uint32_t a15 = a8 & shl_in(a8, pa8, 7);                       // alternations at k-14 .. k
            const uint32_t u2 = xw & shl_in(xw, prevw, 1), u4 = u2 & shl_in(u2, pu2, 2);     // 1s at k-1..k, k-3..k
            const uint32_t t5 = u4 & shl_in(xw, prevw, 4);                      // 1s at k-4..k
            const uint32_t na = ~A & vm, cb = a15 & ~xw & vm;
            const uint32_t e6 = u4 & shl_in(u2, pu2, 4) & vm;                   // 1s at k-5..k
            const uint32_t sf = shl_in(t5, pt, 1) & ~xw & vm;                   // 1s at k-5..k-1, 0 at k
            XW(q) = xw;
            NAa[q * tpb] = na;
            CBa[q * tpb] = cb;
            E6a[q * tpb] = e6;
            SFa[q * tpb] = sf;
            PSa[q * tpb] = ps;
            ps += (uint32_t) __popc(sf);
            nzNA |= (na < 1u ? na : 1u) << q;
            nzCB |= (cb < 1u ? cb : 1u) << q;
            nzE6 |= (e6 < 1u ? e6 : 1u) << q;
            pA = A; pa2 = a2; pa4 = a4; pa8 = a8; pu2 = u2; pt = t5;
            prevw = xw;
        }
        XW(EW) = 0;
        PSa[EW * tpb] = ps;

        // first set bit of bitmap `arr` (words in LDS, non-empty words in `nz`) at position >= p
        auto first_set = [&](const uint32_t *arr, uint32_t nz, int p) -> int {
            if (p >= tile_end) return NONE;
            const int q = p >> 5;
            const uint32_t m = arr[q * tpb] & (~0u << (p & 31));
            if (m) return 32 * q + ctz32(m);
            const uint32_t rest = q >= 31 ? 0u : nz & (~0u << (q + 1));
            if (!rest) return NONE;
            const int q2 = ctz32(rest);
            return 32 * q2 + ctz32(arr[q2 * tpb]);
        };
        auto bit_at = [&](int p) -> uint32_t { return (XW(p >> 5) >> (p & 31)) & 1u; };       // p >= -1 ... careful: p >= 0
        auto window = [&](int p) -> uint32_t {              // bits p .. p+31 of the pack (0 beyond its end)
            const int q = p >> 5, sh = p & 31;
            const uint32_t lo = XW(q), hi = XW(q + 1 <= EW ? q + 1 : EW);
            return sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
        };
        // stuffed 0s among the bits [a, b)
        auto stuffed = [&](int a, int b) -> int {
            if (b <= a) return 0;
            const int qa = a >> 5, qb = b >> 5;
            int n = (int) PSa[qb * tpb] - (int) PSa[qa * tpb];
            n -= __popc(SFa[qa * tpb] & lowmask(a & 31));
            if (qb < EW) n += __popc(SFa[qb * tpb] & lowmask(b & 31));
            return n;
        };

        int pos = 0;
        int rs = -1;                    // position of the last reset inside this pack, -1: none yet
        bool hunt0 = true;              // a hunt that began before the pack: antallpreamble counts from there
        bool slow = false;              // this lane goes bit by bit until its state changes

        // ---- hunting: protodec.c:1030-1043 ---------------------------------------------------------
        // (a lane in ST_SKURR at `pos`; runs at the top of a turn and again at its bottom: a lane whose frame has just
        // closed -- or whose flag count has just been reset -- goes on hunting in the SAME turn instead of holding the
        // whole wave for one more: a pack with f frames on its busiest lane takes f turns, not f + 1)
        auto hunt = [&]() {
            int trig = NONE;
            if (rs < 0 && hunt0) {
                // the count carried into the pack keeps running while the bits alternate from the
                // pack's first bit on: alternations at 0 .. t0-1; bit k brings it to ap0 + k + 1
                const int t0 = first_set(NAa, nzNA, 0);
                int k0 = 14 - antallpreamble;
                if (k0 < 0) k0 = 0;
                const int lim = t0 < tile_end ? t0 : tile_end;
                if (k0 < lim && bit_at(k0)) ++k0;           // the bits alternate: the next one is a 0
                if (k0 < lim) trig = k0;
                const int cb = first_set(CBa, nzCB, 0);
                if (cb < trig) trig = cb;
            } else {
                // a count that started after the reset at rs: 15 alternations at rs+1 ..
                const int from = rs + 15 > pos ? rs + 15 : pos;
                trig = first_set(CBa, nzCB, from);
            }
            if (trig < tile_end) {
                state = ST_PREAMBLE;
                antallpreamble = 0;
                last = 0;
                pos = trig + 1;
            } else {
                // to the pack's end: the alternations that end there, counted from the reset / the
                // carried count
                int lastna = -1;                                   // last position with x[k] == x[k-1]
                for (uint32_t nz = nzNA; nz;) {
                    const int q = 31 - clz32(nz);
                    const uint32_t m = NAa[q * tpb];
                    lastna = 32 * q + 31 - clz32(m);
                    break;
                }
                int ap;
                if (rs < 0 && hunt0) ap = lastna < 0 ? antallpreamble + tile_end : tile_end - 1 - lastna;
                else {
                    const int since = tile_end - 1 - rs, run = tile_end - 1 - lastna;
                    ap = since < run ? since : run;
                }
                antallpreamble = ap > 15 ? 15 : ap;
                last = bit_at(tile_end - 1);
                pos = tile_end;
            }
        };

        while (__any(pos < tile_end)) {
            if (pos < tile_end) {
                if (state == ST_SKURR && !slow) hunt();
                // ---- the training sequence goes on: protodec.c:1047-1048 --------------------------
                if (state == ST_PREAMBLE && nstartsign == 0 && pos < tile_end && !slow) {
                    const int j = first_set(NAa, nzNA, pos);
                    if (j >= tile_end) {
                        const int ap = antallpreamble + (tile_end - pos);
                        antallpreamble = ap > 15 ? 15 : ap;
                        last = bit_at(tile_end - 1);
                        pos = tile_end;
                    } else {
                        const int ap = antallpreamble + (j - pos);
                        antallpreamble = ap > 15 ? 15 : ap;
                        const uint32_t xj = bit_at(j);
                        nstartsign = xj ? 3 : 1;                // protodec.c:1052-1053 / 1065-1066
                        last = xj;
                        pos = j + 1;
                    }
                }
                // ---- counting the flag's 1s: protodec.c:1050-1093 --------------------------------
                // from PREAMBLE with nstartsign = k >= 1 (or STARTSIGN, k = 6, 7) the machine needs
                // R = 7 - k more 1s and then a 0 to enter ST_DATA; a 0 among the first R-1 of them resets
                // from PREAMBLE (nstartsign 0), a 0 at the R-th or a 1 after them resets from STARTSIGN
                // (nstartsign 1 after its trailing ++)
                if (((state == ST_PREAMBLE && nstartsign != 0) || state == ST_STARTSIGN) && pos < tile_end && !slow) {
                    const uint32_t W = window(pos);
                    const int avail = tile_end - pos < 32 ? tile_end - pos : 32;
                    const int R = 7 - nstartsign;
                    int Lr = ctz32(~W);
                    if (Lr > avail) Lr = avail;
                    if (R < 0) {
                        slow = true;                            // nstartsign > 7: never seen, the reference's way
                    } else if (Lr >= R + 1) {                   // a 1 where the 0 had to be
                        HDLC_RESET(); nstartsign = 1; last = 1;
                        rs = pos + R; hunt0 = false;
                        pos += R + 1;
                    } else if (Lr < avail) {                    // the run ends on a visible 0
                        if (Lr == R) {
                            ENTER_DATA();
                            pos += Lr + 1;
                        } else if (Lr == R - 1) {               // 0 in ST_STARTSIGN (nstartsign 6)
                            HDLC_RESET(); nstartsign = 1; last = 0;
                            rs = pos + Lr; hunt0 = false;
                            pos += Lr + 1;
                        } else {                                // 0 while still in ST_PREAMBLE
                            HDLC_RESET(); last = 0;
                            rs = pos + Lr; hunt0 = false;
                            pos += Lr + 1;
                        }
                    } else {                                    // only 1s left in this pack
                        nstartsign += Lr;
                        if (nstartsign >= 6) { state = ST_STARTSIGN; antallpreamble = 0; }
                        last = 1;
                        pos += Lr;
                    }
                }
                // ---- inside a frame: protodec.c:995-1028 -------------------------------------------
                if (state == ST_DATA && pos < tile_end && !slow) {
                    const int e6 = first_set(E6a, nzE6, pos);
                    const int e = e6 < tile_end ? e6 : tile_end;    // raw bits pos .. e-1 go into the record
                    const int stored = (e - pos) - stuffed(pos, e);
                    if (bufferpos + stored >= 449) {
                        // protodec.c:1024-1026: the frame is given up at the bit that brings bufferpos to
                        // 449 -- the `need`-th stored (not stuffed) bit from here.  Rare (a false start in
                        // noise that finds no closing flag), but bit by bit it would hold the whole wave up
                        // for hundreds of turns: find the word, then the bit.
                        int need = 449 - bufferpos, p = pos, t = pos;
                        for (;;) {
                            const int q = p >> 5;
                            const int we = (32 * (q + 1) < e) ? 32 * (q + 1) : e;
                            uint32_t m = ~SFa[q * tpb] & (~0u << (p & 31)) & lowmask(we - 32 * q);   // stored bits p .. we-1
                            const int n = __popc(m);
                            if (n >= need) {
                                for (int k = 1; k < need; ++k) m &= m - 1;
                                t = 32 * q + ctz32(m);
                                break;
                            }
                            need -= n;
                            p = we;
                        }
                        if (rec_ok) rec[0] = 0;
                        const uint32_t xt = bit_at(t);
                        HDLC_RESET();
                        last = xt;
                        rs = t; hunt0 = false;
                        pos = t + 1;
                    } else {
                        for (int p = pos; p < e;) {
                            const int sh = p & 31;
                            const int take = (32 - sh < e - p) ? 32 - sh : e - p;
                            RAW_APPEND((XW(p >> 5) >> sh) & lowmask(take), take);
                            p += take;
                        }
                        bufferpos += stored;
                        if (e6 < tile_end) {                    // the sixth 1
                            state = ST_STOPSIGN;
                            antallenner = 0;
                            bitstuff = 0;
                            last = 1;
                            pos = e6 + 1;
                        } else {
                            // the pack ends inside the frame: the 1s counted so far (at most five, or the
                            // sixth would have been found), looking back into the word before the pack
                            // when the pack is that short
                            int r;
                            if (tile_end >= 8) {
                                r = clz32(~(window(tile_end - 8) << 24));
                            } else {
                                r = clz32(~((XW(0) << (32 - tile_end)) | (XW(-1) >> tile_end)));
                            }
                            last = bit_at(tile_end - 1);
                            if (r >= 5) { bitstuff = 1; antallenner = 0; }
                            else { bitstuff = 0; antallenner = r > 0 ? r - 1 : 0; }
                            pos = tile_end;
                        }
                    }
                }
                // ---- the bit after the closing flag: protodec.c:1095-1115 --------------------------
                if (state == ST_STOPSIGN && pos < tile_end && !slow) {
                    const uint32_t x = bit_at(pos);
                    STOP_BIT(x, pos);
                    rs = pos; hunt0 = false;
                    pos += 1;
                }
                // ---- anything else, and whatever asked for it: the reference, bit by bit ---------
                if (pos < tile_end && (slow || state < ST_SKURR || state > ST_STOPSIGN)) {
                    const int st0 = state;
                    const uint32_t x = bit_at(pos);
                    slow_bit(x, pos);
                    if (state == ST_SKURR && st0 != ST_SKURR) { rs = pos; hunt0 = false; }
                    if (state != st0) slow = false;
                    pos += 1;
                }
                if (state == ST_SKURR && !slow && pos < tile_end) hunt();
            }
        }
        seenbase += (uint32_t) tile_end;
    }
#undef XW
#undef HDLC_RESET
#undef ENTER_DATA
#undef RAW_APPEND
#undef STOP_BIT

    if (live) {
        ctl[c] = (uint32_t) state | ((uint32_t) nstartsign << 3) | ((uint32_t) antallpreamble << 7) |
                 ((uint32_t) antallenner << 11) | ((uint32_t) bitstuff << 14) | (last << 15) |
                 ((uint32_t) bufferpos << 16);
        ctl[n_ + c] = cur;
        ctl[2 * n_ + c] = seenbase;
        ctl[5 * n_ + c] = seenhi0 + (seenbase < seen0 ? 1u : 0u);
        ctl[3 * n_ + c] = nstart;
        ctl[4 * n_ + c] = (uint32_t) rawpos;
        const bool open1 = (state == ST_DATA || state == ST_STOPSIGN);
        const uint32_t limit = nstart - (open1 ? 1u : 0u);
        cand_first[c] = first;
        uint32_t cnt = limit - first;
        cand_count[c] = cnt > (uint32_t) K ? (uint32_t) K : cnt;
        if (lost2) counters[2 * n_ + c] += lost2;
    }
}

hipError_t launch_hdlc_events(const HdlcLaunch &a, hipStream_t stream)
{
    if (a.seg_words > PACK_STRIDE) return hipErrorInvalidValue;
    const int lpw = a.lanes_per_wave >= 1 && a.lanes_per_wave <= 64 ? a.lanes_per_wave : 64;
    const dim3 grid((a.N + lpw - 1) / lpw), block(lpw);
    const size_t lds = EV_ROWS * lpw * sizeof(uint32_t);
#define EV_LAUNCH(T)                                                                                                  \
    hipLaunchKernelGGL(hdlc_events_kernel<T>, grid, block, lds, stream, a.segbits, a.segcnt, a.ctl, a.cand, a.cand_first, \
                       a.cand_count, a.counters, a.frame_count, a.N, a.n_seg, a.seg_words, a.K)
    switch (lpw) {
    case 8: EV_LAUNCH(8); break;
    case 16: EV_LAUNCH(16); break;
    case 32: EV_LAUNCH(32); break;
    case 64: EV_LAUNCH(64); break;
    default: EV_LAUNCH(0); break;
    }
#undef EV_LAUNCH
    return hipGetLastError();
}

} // namespace gnuais
