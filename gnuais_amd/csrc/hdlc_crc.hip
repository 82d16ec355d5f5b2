// hdlc_crc.hip -- K2b: HDLC deframer (flag hunt + bit-unstuffing FSM) and
// K3: CRC-16 check + frame delivery, for gfx950.
//
// Stands in for protodec_decode() (gnuais src/protodec.c:988-1122), for
// protodec_reset() (src/protodec.c:87-100) and for protodec_calculate_crc() /
// protodec_sdlc_crc() (src/protodec.c:106-167), for a whole batch of channels.
//
// K2b -- one lane = one channel walking its own recovered bit stream (K2's
// per-segment bit packs) at its own pace.  A wave that is alone on its SIMD issues
// about one instruction per 5 cycles, so what counts is instructions per channel,
// i.e. steps x instructions per step.  The five-state machine of the reference is
// reproduced exactly -- same states, same counters, same quirks (SURVEY.md
// appendix A.7-A.9: `last` rewritten after every state, `nstartsign++` even after
// a reset in ST_STARTSIGN, reset at bufferpos >= 449) -- but not bit by bit:
//   ST_SKURR : "15 alternations then a 0" is a run-length test on x ^ (x << 1),
//              up to 32 bits per step;
//   ST_DATA  : up to 32 RAW bits per step.  Only two things matter while a frame is
//              open: where it ends (five 1s followed by a sixth) and how many bits
//              have been stored (bufferpos, for the 449 limit and the frame
//              length); stuffed 0s are counted with a popcount and left in place.
//              The raw bits go straight into the frame's candidate record in HBM
//              (fire-and-forget stores, no read-back); a frame that is still open
//              at the end of a call continues in the same record at the next call;
//   ST_PREAMBLE / ST_STARTSIGN / ST_STOPSIGN last ~20 bits per frame and run bit
//              by bit, but inside one step.
// Lanes are not in lock-step on the bit index, so a wave needs
// max-over-lanes(steps) iterations, not the sum.
//
// K3 -- the candidates closed in this call, shared evenly over threads: remove the
// stuffed 0s (the bit after every five 1s), then the reference's CRC over
// n/8 + 2 bytes (bits packed LSB first, protodec.c:138-143), good frame <=>
// 0x0f47 after the final complement (protodec.c:166).  Taking both out of the
// sequential walk costs nothing in exactness and turns per-bit work into fully
// parallel work.  Good frames are appended to the frame ring as 64-byte records
// and counted in receivedframes, bad ones in lostframes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gnuais {

enum { ST_SKURR = 1, ST_PREAMBLE = 2, ST_STARTSIGN = 3, ST_DATA = 4, ST_STOPSIGN = 5 };

// ctl[0]: state[2:0] nstartsign[6:3] antallpreamble[10:7] antallenner[13:11]
//         bitstuff[14] last[15] bufferpos[24:16]
// ctl[1]: partially filled word of the raw frame record
// ctl[2]: bits fed since reset, low word; ctl[5]: high word (a frame carries 5 of its bits, see below)
// ctl[3]: ST_DATA entries since reset (candidate slots handed out)
// ctl[4]: raw bits in the open frame record
constexpr uint32_t CAND_VALID = 0x10000u;
constexpr int PACK_MAX = PACK_STRIDE;   // words per segment pack held in registers/LDS

__global__ void hdlc_reset_kernel(uint32_t *__restrict__ ctl, int N)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    ctl[c] = ST_SKURR;                  // protodec.c:89-99
    ctl[(size_t) N + c] = 0;
    ctl[(size_t) 2 * N + c] = 0;
    ctl[(size_t) 3 * N + c] = 0;
    ctl[(size_t) 4 * N + c] = 0;
    ctl[(size_t) 5 * N + c] = 0;
}

// protodec_reset() alone (protodec.c:87-100): the machine's fields, the frame in progress; bits seen and frames started
// keep counting (they order a channel's frames in time)
__global__ void hdlc_fsm_reset_kernel(uint32_t *__restrict__ ctl, int N)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    ctl[c] = ST_SKURR;
    ctl[(size_t) N + c] = 0;
    ctl[(size_t) 4 * N + c] = 0;
}

__device__ __forceinline__ uint32_t lowmask(int k)      // k in [0, 32]
{
    return k >= 32 ? ~0u : ((1u << k) - 1u);
}
__device__ __forceinline__ int ctz32(uint32_t v)        // 32 for v == 0
{
    return v ? __ffs((int) v) - 1 : 32;
}
__device__ __forceinline__ int clz32(uint32_t v)        // 32 for v == 0
{
    return v ? __clz((int) v) : 32;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void hdlc_deframe_kernel(
    const uint32_t *__restrict__ segbits, const uint32_t *__restrict__ segcnt,
    uint32_t *__restrict__ ctl, uint32_t *__restrict__ cand, uint32_t *__restrict__ cand_first,
    uint32_t *__restrict__ cand_count, int32_t *__restrict__ counters,
    uint32_t *__restrict__ flags, int N, int n_seg, int seg_words, int K)
{
    __builtin_amdgcn_s_setprio(3);      // latency-bound chain: take every issue slot it can use
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = cg < N;
    const size_t c = (size_t) (live ? cg : N - 1), n_ = (size_t) N;

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

    uint32_t seenbase = seen0;          // bits fed before the current segment

#define HDLC_RESET()                                                                    \
    do { state = ST_SKURR; nstartsign = 0; antallpreamble = 0; antallenner = 0;        \
         last = 0; bitstuff = 0; bufferpos = 0; } while (0)

    // K2 hands over one bit pack per 2048-sample segment; the word-parallel steps
    // accept any window length, so a pack boundary is just a short window.  A step's
    // window must not wait on HBM/L2 (~1-2 us per dependent load, several hundred
    // steps per channel): the pack of the NEXT segment is fetched into registers
    // while the current one is walked out of LDS.
    extern __shared__ uint32_t lds_pack[];                    // [PACK_MAX + 1][blockDim.x]
    const int tpb = (int) blockDim.x, tx = (int) threadIdx.x;
    const uint32_t *__restrict__ rows = segbits + c * (size_t) n_seg * (size_t) PACK_STRIDE;
    uint32_t pf[PACK_MAX];
    int pf_cnt = live ? (int) segcnt[c * (size_t) n_seg] : 0;
    // a pack is PACK_STRIDE words, 64-byte aligned, zero beyond its bits: four 16-byte loads
#pragma unroll
    for (int q = 0; q < PACK_MAX / 4; ++q) {
        const uint4 v = reinterpret_cast<const uint4 *>(rows)[q];
        pf[4 * q] = v.x; pf[4 * q + 1] = v.y; pf[4 * q + 2] = v.z; pf[4 * q + 3] = v.w;
    }

    for (int seg = 0; seg < n_seg; ++seg) {
        int tile_end = pf_cnt;
        if (tile_end > seg_words * 32) tile_end = seg_words * 32;
#pragma unroll
        for (int q = 0; q < PACK_MAX; ++q) lds_pack[q * tpb + tx] = pf[q];
        lds_pack[PACK_MAX * tpb + tx] = 0;
        if (seg + 1 < n_seg) {
            const uint32_t *__restrict__ nrow = rows + (size_t) (seg + 1) * (size_t) PACK_STRIDE;
            pf_cnt = live ? (int) segcnt[c * (size_t) n_seg + seg + 1] : 0;
#pragma unroll
            for (int q = 0; q < PACK_MAX / 4; ++q) {
                const uint4 v = reinterpret_cast<const uint4 *>(nrow)[q];
                pf[4 * q] = v.x; pf[4 * q + 1] = v.y; pf[4 * q + 2] = v.z; pf[4 * q + 3] = v.w;
            }
        }
        int pos = 0;
        // A wave executes the union of what its lanes do, so the walk is batched by
        // mode: all lanes hunting for a preamble advance together, then all lanes
        // matching a flag, then all lanes inside a frame.  Every step is closed-form
        // (no per-bit loop anywhere), and an iteration runs exactly one mode's code.
#define FETCH_WINDOW()                                                                          \
        const int lw = pos >> 5, sh = pos & 31;                                                 \
        const uint32_t lo = lds_pack[lw * tpb + tx], hi = lds_pack[(lw + 1) * tpb + tx];        \
        const uint32_t W = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo; /* bit j = x[pos+j] */  \
        const int nv = (tile_end - pos < 32) ? tile_end - pos : 32

        while (__any(pos < tile_end)) {
            // ---- ST_SKURR: protodec.c:1030-1043, up to 32 bits per step -------------
            while (__any(pos < tile_end && state == ST_SKURR)) {
                if (pos < tile_end && state == ST_SKURR) {
                    FETCH_WINDOW();
                    const uint32_t vm = lowmask(nv);
                    const uint32_t A = W ^ ((W << 1) | last);      // bit j: x_j != x_{j-1}
                    const uint32_t B2 = A & (A << 1), B4 = B2 & (B2 << 2), B8 = B4 & (B4 << 4);
                    const uint32_t B15 = B8 & (B8 << 7);           // 15 alternations inside the window
                    const int t = ctz32(~A);                        // alternation run from the window start
                    const int need = 14 - antallpreamble;           // run continues the carried count
                    const uint32_t C = lowmask(t) & ~lowmask(need > 0 ? need : 0);
                    const uint32_t TR = (B15 | C) & ~W & vm;        // antallpreamble > 14 && x == 0
                    if (TR) {
                        pos += ctz32(TR) + 1;
                        state = ST_PREAMBLE;
                        antallpreamble = 0;
                        last = 0;
                    } else {
                        pos += nv;
                        last = (W >> (nv - 1)) & 1u;
                        const int ap = (t >= nv) ? antallpreamble + nv : clz32(~(A << (32 - nv)));
                        antallpreamble = ap > 15 ? 15 : ap;
                    }
                }
            }
            // ---- ST_PREAMBLE / ST_STARTSIGN / ST_STOPSIGN ------------------------------
            // protodec.c:1045-1115.  After the alternating training bits the machine only
            // counts 1s: from PREAMBLE with nstartsign = k >= 1 (or STARTSIGN, k = 6, 7) it
            // needs R = 7 - k more 1s and then a 0 to enter ST_DATA; a 0 among the first
            // R-1 of them resets from PREAMBLE (nstartsign 0), a 0 at the R-th or a 1 after
            // them resets from STARTSIGN (nstartsign 1 after its trailing ++).
            while (__any(pos < tile_end && state != ST_SKURR && state != ST_DATA)) {
                if (pos < tile_end && state != ST_SKURR && state != ST_DATA) {
                    FETCH_WINDOW();
                    int k = 0;
                    if (state == ST_STOPSIGN) {                 // protodec.c:1095-1115
                        const uint32_t x = W & 1u;
                        const int nb = bufferpos - 6 - 16;
                        if (x == 0 && nb > 0) {
                            if (rec_ok) {
                                rec[CAND_HDR + (rawpos >> 5)] = cur;
                                // when the frame ended: bits fed so far, 37 of them (32 in the record's
                                // end_bit, 5 in its flags byte): 9600 bit/s wrap that after 165 days
                                const uint32_t eb = seenbase + (uint32_t) pos;
                                const uint32_t ebhi = (seenhi0 + (eb < seen0 ? 1u : 0u)) & 31u;
                                rec[1] = eb;
                                rec[0] = (uint32_t) nb | CAND_VALID | ((uint32_t) rawpos << 17) | (ebhi << 27);
                            }
                        } else {
                            ++lost2;                            // protodec.c:1112
                            if (rec_ok) {                       // not a candidate (no CAND_VALID); its bits stay on record for d->buffer
                                rec[CAND_HDR + (rawpos >> 5)] = cur;
                                rec[0] = (uint32_t) rawpos << 17;
                            }
                        }
                        HDLC_RESET();
                        last = x;                               // protodec.c:1119
                        k = 1;
                    } else if (state != ST_PREAMBLE && state != ST_STARTSIGN) {
                        HDLC_RESET();                           // not a state: as the reference's
                        last = W & 1u;                          // switch would fall through
                        k = 1;
                    } else {
                        if (state == ST_PREAMBLE && nstartsign == 0) {
                            // alternating part of the training sequence, all at once
                            const uint32_t A = W ^ ((W << 1) | last);
                            int t = ctz32(~A);
                            if (t > nv) t = nv;
                            if (t > 0) {
                                antallpreamble = antallpreamble + t > 15 ? 15 : antallpreamble + t;
                                last = (W >> (t - 1)) & 1u;
                                k = t;
                            }
                            if (k < nv) {                       // first repeated bit
                                const uint32_t x0 = (W >> k) & 1u;
                                nstartsign = x0 ? 3 : 1;        // protodec.c:1052-1053 / 1065-1066
                                last = x0;
                                ++k;
                            }
                        }
                        if (k < nv && nstartsign != 0) {
                            const int R = 7 - nstartsign;       // 1s still needed, then a 0
                            const int avail = nv - k;
                            int L = ctz32(~(W >> k));           // run of 1s from bit k
                            if (L > avail) L = avail;
                            if (L >= R + 1) {                   // a 1 where the 0 had to be
                                HDLC_RESET(); nstartsign = 1; last = 1; k += R + 1;
                            } else if (L < avail) {             // the run ends on a visible 0
                                if (L == R) {                   // protodec.c:1076-1082: ST_DATA
                                    state = ST_DATA; nstartsign = 1; antallenner = 0; antallpreamble = 0;
                                    bufferpos = 0; cur = 0; rawpos = 0; bitstuff = 0; last = 0;
                                    rec_ok = (nstart - first) < (uint32_t) K;
                                    rec = cand + ((size_t) c * K + nstart % (uint32_t) K) * CAND_WORDS;
                                    if (rec_ok) rec[0] = 0; else flags[1] = 1;
                                    ++nstart;
                                } else if (L == R - 1) {        // 0 in ST_STARTSIGN (nstartsign 6)
                                    HDLC_RESET(); nstartsign = 1; last = 0;
                                } else {                        // 0 while still in ST_PREAMBLE
                                    HDLC_RESET(); last = 0;
                                }
                                k += L + 1;
                            } else {                            // only 1s left in this window
                                nstartsign += L;
                                if (nstartsign >= 6) { state = ST_STARTSIGN; antallpreamble = 0; }
                                last = 1;
                                k += L;
                            }
                        }
                    }
                    pos += k;
                }
            }
            // ---- ST_DATA: protodec.c:995-1028, up to 32 raw bits per step ----------------
            while (__any(pos < tile_end && state == ST_DATA)) {
                if (pos < tile_end && state == ST_DATA) {
                    FETCH_WINDOW();
                    if (bitstuff) {                             // protodec.c:996-1007
                        const uint32_t x = W & 1u;
                        if (x) {
                            state = ST_STOPSIGN;                // sixth 1: closing flag (or abort)
                        } else {                                // stuffed 0: stays in the raw record
                            if ((rawpos & 31) == 31) {
                                if (rec_ok) rec[CAND_HDR + (rawpos >> 5)] = cur;
                                cur = 0;
                            }
                            ++rawpos;
                        }
                        bitstuff = 0;
                        last = x;
                        pos += 1;
                    } else {                                    // protodec.c:1008-1027
                        // m = run of 1s ending at the previous bit (antallenner = m-1 when last = 1)
                        const int m = last ? antallenner + 1 : 0;
                        const int room = 449 - bufferpos;       // protodec.c:1024
                        int nw = nv;
                        if (nw > 32 - m) nw = 32 - m;
                        if (nw > room) nw = room;               // bufferpos can reach 449 only at
                                                                // the last bit of this step
                        const int nvE = nw + m;
                        const uint32_t vmE = lowmask(nvE);
                        // E: the carried 1s, then the window; bit j = raw bit j-m
                        const uint32_t E = ((W << m) | lowmask(m)) & vmE;
                        const uint32_t X1 = E & (E >> 1), X2 = X1 & (X1 >> 2);
                        const uint32_t R5 = X2 & (E >> 4);      // five 1s starting at bit j
                        const uint32_t S5 = R5 & ~(E << 1);     // ... that begin a run
                        const uint32_t P5 = S5 << 4;            // position of the run's fifth 1
                        const uint32_t nx = E >> 1, kn = vmE >> 1;
                        const uint32_t END5 = P5 & nx & kn;     // followed by a sixth 1: closing flag
                        const uint32_t STF5 = P5 & ~nx & kn;    // followed by a 0: stuffing, dropped
                        const uint32_t PND5 = P5 & ~kn;         // fifth 1 is the last bit seen
                        int nraw, stored;
                        if (END5) {
                            const int pe = ctz32(END5);
                            nraw = pe + 1 - m;                  // raw bits up to the fifth 1
                            stored = nraw - __popc(STF5 & lowmask(pe));
                        } else {
                            nraw = nw;
                            stored = nw - __popc(STF5);
                        }
                        {   // raw record [rawpos .. rawpos+nraw) = x[pos .. pos+nraw)
                            const uint32_t sb = W & lowmask(nraw);
                            const int bsh = rawpos & 31;
                            const uint64_t t64 = (uint64_t) sb << bsh;
                            cur |= (uint32_t) t64;
                            if (bsh + nraw >= 32) {
                                if (rec_ok) rec[CAND_HDR + (rawpos >> 5)] = cur;
                                cur = (uint32_t) (t64 >> 32);
                            }
                            rawpos += nraw;
                        }
                        bufferpos += stored;
                        if (END5) {
                            pos += nraw + 1;                    // ... and the sixth 1
                            state = ST_STOPSIGN;
                            antallenner = 0;
                            last = 1;
                        } else {
                            pos += nraw;
                            const uint32_t xl = (W >> (nraw - 1)) & 1u;
                            if (bufferpos >= 449) {             // give up the frame
                                if (rec_ok) rec[0] = 0;
                                HDLC_RESET();
                                last = xl;
                            } else if (PND5) {                  // next bit is stuffing or flag
                                bitstuff = 1;
                                antallenner = 0;
                                last = 1;
                            } else {
                                const int r = clz32(~(E << (32 - nvE)));   // run of 1s at the end
                                last = xl;
                                antallenner = r > 0 ? r - 1 : 0;
                            }
                        }
                    }
                }
            }
        }
#undef FETCH_WINDOW
        seenbase += (uint32_t) tile_end;
    }
#undef HDLC_RESET

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

// K3: CRC check + delivery of the candidates closed in this call.  One block of 256
// threads owns K3_CH adjacent channels: their candidate counts are prefix-summed in
// LDS and the candidates, enumerated channel-major then in time, are taken 256 at
// a time (one per thread).  CRC-16/X-25 by bytes through a 256-entry table built
// in LDS from the bitwise definition (protodec.c:106-118).  The good frames of a
// pass are compacted in order (ballot ranks) and the block reserves one contiguous
// piece of the frame ring for them.  Pieces land in whatever order the blocks finish;
// the drain restores the reference's print order with a device sort on (channel, end_bit).
__global__ __launch_bounds__(256) void hdlc_crc_kernel(
    const uint32_t *__restrict__ cand, const uint32_t *__restrict__ cand_first,
    const uint32_t *__restrict__ cand_count, int32_t *__restrict__ counters,
    uint32_t *__restrict__ frames, uint32_t *__restrict__ flags, uint32_t frame_cap, int N, int K,
    uint2 *__restrict__ chunks, int n_pass)
{
    __shared__ uint32_t tab[256];
    __shared__ uint32_t pre[K3_CH + 1];
    // The two per-thread buffers are DYNAMIC shared memory (K3_DYN_LDS bytes at launch).  As static arrays (36 KB) they
    // made the compiler derive "at most four waves per SIMD" from the workgroup's LDS and then pad the kernel's register
    // request up to that occupancy: 104 VGPRs per wave in the descriptor for 59 in use -- beside four FIR waves and a
    // PLL wave that is the difference between a wave that fits a SIMD's free registers and one that waits for a FIR
    // wave to retire.
    extern __shared__ uint32_t k3_dyn[];
    uint32_t (*const stage)[HDLC_BUF_WORDS + 1] = reinterpret_cast<uint32_t (*)[HDLC_BUF_WORDS + 1]>(k3_dyn);   // unstuffed frame bits per thread
    uint32_t (*const rawlds)[CAND_WORDS - CAND_HDR + 1] =                                                     // raw record per thread
        reinterpret_cast<uint32_t (*)[CAND_WORDS - CAND_HDR + 1]>(k3_dyn + 256 * (HDLC_BUF_WORDS + 1));
    __shared__ uint32_t wave_cnt[4];
    __shared__ uint32_t pass_base;
    const int tid = threadIdx.x;
    const int c_own = blockIdx.x * K3_CH + tid;

    {   // table entry tid: eight LFSR steps on the byte value
        uint32_t t = (uint32_t) tid;
#pragma unroll
        for (int k = 0; k < 8; ++k) t = (t >> 1) ^ ((t & 1u) ? 0x8408u : 0u);
        tab[tid] = t;
    }
    // prefix sums of the per-channel candidate counts (K3_CH is small: serial scan)
    if (tid < K3_CH) pre[tid + 1] = (c_own < N) ? cand_count[c_own] : 0u;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        pre[0] = 0;
        for (int q = 1; q <= K3_CH; ++q) { run += pre[q]; pre[q] = run; }
    }
    __syncthreads();
    const uint32_t total = pre[K3_CH];
    const size_t n_ = (size_t) N;

    int pass = 0;
    for (uint32_t i0 = 0; i0 < total; i0 += 256, ++pass) {
        const uint32_t i = i0 + (uint32_t) tid;
        bool good = false;
        uint32_t out[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) out[q] = 0;
        if (i < total) {
            // channel of candidate i: largest k with pre[k] <= i
            int lo = 0, hi = K3_CH - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pre[mid] <= i) lo = mid; else hi = mid - 1;
            }
            const int c = blockIdx.x * K3_CH + lo;
            const uint32_t j = i - pre[lo];
            const uint32_t slot = (cand_first[c] + j) % (uint32_t) K;
            const uint32_t *rec = cand + ((size_t) c * K + slot) * CAND_WORDS;
            const uint32_t hdr = rec[0];
            if (hdr & CAND_VALID) {                     // else: abandoned before its closing flag
                const int n = (int) (hdr & 0xffffu);
                const int rawlen = (int) ((hdr >> 17) & 0x3ffu);
                const int nbytes = n >> 3, buflen = nbytes + 2; // protodec.c:133-134
                // protodec.c:1008-1023 on the raw bits: store every bit except the one
                // that follows five 1s (it is a stuffed 0 inside a frame)
                {
                    // whole record first (one latency), parked in LDS so that the bit
                    // loop below can stay rolled (small code: this kernel shares the
                    // instruction cache with the FIR and the sequential kernels)
                    uint32_t raw[CAND_WORDS - CAND_HDR];
#pragma unroll
                    for (int q = 0; q < CAND_WORDS - CAND_HDR; ++q) raw[q] = rec[CAND_HDR + q];
#pragma unroll
                    for (int q = 0; q < CAND_WORDS - CAND_HDR; ++q) rawlds[tid][q] = raw[q];
                    // A word at a time.  Inside a candidate a run of 1s is at most five long (a sixth
                    // closed the frame), so the bits to drop are exactly the 0s that follow five 1s:
                    // bit i of `five` = raw[i-1] & ... & raw[i-5], across the word boundary through the
                    // previous word.  The few such bits of a word are cut out one by one, the rest is
                    // appended to the output through a 64-bit window.
                    unsigned long long acc = 0;
                    uint32_t prevw = 0;
                    int fill = 0, ow = 0;
                    const int nwords = (rawlen + 31) >> 5;
#pragma unroll 1
                    for (int q = 0; q < nwords; ++q) {
                        uint32_t rw = rawlds[tid][q];
                        const int nvq = rawlen - 32 * q < 32 ? rawlen - 32 * q : 32;
                        const unsigned long long cc = ((unsigned long long) rw << 32) | prevw;
                        const uint32_t five = (uint32_t) (cc >> 31) & (uint32_t) (cc >> 30) & (uint32_t) (cc >> 29) &
                                              (uint32_t) (cc >> 28) & (uint32_t) (cc >> 27);
                        uint32_t stf = five & ~rw & (nvq >= 32 ? ~0u : (1u << nvq) - 1u);
                        prevw = rw;
                        int nout = nvq;
                        while (stf) {                               // highest first: lower positions stay valid
                            const int pz = 31 - __clz((int) stf);
                            stf &= ~(1u << pz);
                            rw = (rw & ((1u << pz) - 1u)) | ((pz >= 31 ? 0u : rw >> (pz + 1)) << pz);
                            --nout;
                        }
                        if (nout < 32) rw &= (1u << nout) - 1u;
                        acc |= (unsigned long long) rw << fill;
                        fill += nout;
                        if (fill >= 32) {
                            if (ow <= HDLC_BUF_WORDS) stage[tid][ow] = (uint32_t) acc;
                            ++ow;
                            acc >>= 32;
                            fill -= 32;
                        }
                    }
                    if (ow <= HDLC_BUF_WORDS) stage[tid][ow] = (uint32_t) acc;
                    for (int q = ow + 1; q <= HDLC_BUF_WORDS; ++q) stage[tid][q] = 0;
                }
                uint32_t w[HDLC_BUF_WORDS];
#pragma unroll
                for (int q = 0; q < HDLC_BUF_WORDS; ++q) w[q] = stage[tid][q];
                uint32_t crc = 0xffffu;
#pragma unroll
                for (int q = 0; q < HDLC_BUF_WORDS; ++q) {
#pragma unroll
                    for (int bq = 0; bq < 4; ++bq) {
                        if (q * 4 + bq < buflen) {
                            const uint32_t byte = (w[q] >> (8 * bq)) & 0xffu;
                            crc = (crc >> 8) ^ tab[(crc ^ byte) & 0xffu];
                        }
                    }
                }
                if (crc == 0xf0b8u) {                   // ~crc == 0x0f47, protodec.c:166
                    good = true;
                    atomicAdd(&counters[c], 1);         // protodec.c:1103
                    out[0] = (uint32_t) c;
                    out[1] = rec[1];
#pragma unroll
                    for (int q = 0; q < 14; ++q) {
                        uint32_t v = 0;
                        if (q * 4 < nbytes) {
                            v = w[q];
                            const int keep = nbytes - q * 4;
                            if (keep < 4) v &= (1u << (8 * keep)) - 1u;
                        }
                        // flags: bit 0 CRC ok, bits 5:1 = bits 36:32 of end_bit
                        if (q == 13) v = (v & 0xffu) | ((1u | ((hdr >> 27) << 1)) << 8) | ((uint32_t) n << 16);
                        out[2 + q] = v;
                    }
                } else {
                    atomicAdd(&counters[n_ + c], 1);    // protodec.c:1107
                }
            }
        }
        // ordered compaction of this pass's good frames
        const unsigned long long m = __ballot(good);
        const int wave = tid >> 6, lane = tid & 63;
        const uint32_t below = (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = (uint32_t) __popcll(m);
        __syncthreads();
        uint32_t woff = 0, npass = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < wave) woff += wave_cnt[q];
            npass += wave_cnt[q];
        }
        if (tid == 0) {
            const uint32_t base = npass ? atomicAdd(&flags[0], npass) : 0u;
            pass_base = base;
            // candidates are walked channel by channel, each channel's in time order, and the compaction
            // keeps that order: (block, pass, position in the pass) is the reference's print order
            if (chunks && pass < n_pass)
                chunks[(size_t) blockIdx.x * n_pass + pass] =
                    make_uint2(base, base >= frame_cap ? 0u : (npass < frame_cap - base ? npass : frame_cap - base));
        }
        __syncthreads();
        if (good) {
            const uint32_t idx = pass_base + woff + below;
            if (idx < frame_cap) {
                uint4 *dst = reinterpret_cast<uint4 *>(frames + (size_t) idx * 16);
                dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
                dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
                dst[2] = make_uint4(out[8], out[9], out[10], out[11]);
                dst[3] = make_uint4(out[12], out[13], out[14], out[15]);
            } else {
                flags[1] = 1;                           // ring full: frame dropped, still counted
            }
        }
        __syncthreads();
    }
    if (chunks)
        for (int q = pass + tid; q < n_pass; q += 256) chunks[(size_t) blockIdx.x * n_pass + q] = make_uint2(0u, 0u);
}

hipError_t launch_hdlc_deframe(const HdlcLaunch &a, hipStream_t stream)
{
    const int lpw = a.lanes_per_wave > 0 ? a.lanes_per_wave : 64;
    if (a.seg_words > PACK_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hdlc_deframe_kernel, dim3((a.N + lpw - 1) / lpw), dim3(lpw),
                       (PACK_MAX + 1) * lpw * sizeof(uint32_t), stream,
                       a.segbits, a.segcnt, a.ctl, a.cand, a.cand_first, a.cand_count, a.counters,
                       a.frame_count, a.N, a.n_seg, a.seg_words, a.K);
    return hipGetLastError();
}

hipError_t launch_hdlc_crc(const HdlcLaunch &a, hipStream_t stream)
{
    constexpr size_t K3_DYN_LDS = 256 * ((HDLC_BUF_WORDS + 1) + (CAND_WORDS - CAND_HDR + 1)) * sizeof(uint32_t);
    hipLaunchKernelGGL(hdlc_crc_kernel, dim3((unsigned) ((a.N + K3_CH - 1) / K3_CH)), dim3(256), K3_DYN_LDS,
                       stream, a.cand, a.cand_first, a.cand_count, a.counters,
                       (uint32_t *) a.frames, a.frame_count, a.frame_cap, a.N, a.K, a.chunks, k3_passes(a.K_call > 0 ? a.K_call : a.K));
    return hipGetLastError();
}

hipError_t launch_hdlc_fsm_reset(uint32_t *ctl, int N, hipStream_t stream)
{
    hipLaunchKernelGGL(hdlc_fsm_reset_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, ctl, N);
    return hipGetLastError();
}

hipError_t launch_hdlc_reset(uint32_t *ctl, int N, hipStream_t stream)
{
    hipLaunchKernelGGL(hdlc_reset_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, ctl, N);
    return hipGetLastError();
}

} // namespace gnuais
