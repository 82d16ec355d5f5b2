// pll_nrzi.hip -- K2t / K2a: bit-clock recovery PLL, slice and NRZI decode for gfx950.
//
// Together they stand in for the per-sample loop of receiver_run(), gnuais
// src/receiver.c:109-135, for a whole batch of channels.
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
// over the 64 channels of a wave, re-synchronised every 2048 samples), eight instructions each:
//
//   K2t  pll_edges_kernel (parallel over channel x 2048-sample segment): sign words -> the list of
//        transition positions of the segment;
//   K2a  pll_phase_kernel (sequential in time, lane = channel): walks the lists with U scaled by
//        2^7 -- seven spare low bits, see PLL_STEP -- toggling bits of the segment's pack in LDS;
//        nrzi_carry_kernel then carries the toggles that fall on a later segment's first slice
//        across the segment boundaries and into the next call.
//
// Output: one pack of <= PACK_STRIDE words + a bit count per (channel, segment); bit k of a pack
// is at word k/32, bit k%32.
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

// ---- K2t ---------------------------------------------------------------------------------------
// grid.x = channel group (64 channels), grid.y = segment.  Per lane the positions (0 .. 2047) of the
// segment's transitions in time order, eight 16-bit entries to a 16-byte "pair" (two groups of four):
//   pair 0            header: .x = count, .z/.w = the count % 4 entries that do not fill a group
//                     (the sequential kernel takes them one by one at the end)
//   pair 1 + j        entries 8j .. 8j+7 (whole groups only; a trailing half pair holds one group)
// en4p[segment][group] = pairs the longest lane of the wave needs: what K2a streams.
// A lane appends to its own strip of LDS (one ds_write_b16 and an add per transition); whole pairs
// leave for HBM whenever a strip might fill up.
#ifdef K2T_NOSTORE
#define K2T_NOSTORE_V 1
#else
#define K2T_NOSTORE_V 0
#endif
constexpr int EDGE_STW = 37;          // words per lane strip: 64 entries + the flush's look-ahead (odd: no bank conflicts)
constexpr int EDGE_FLUSH = 32;        // flush when a strip could overflow during the next word
#ifndef EDGE_BATCH_N
#define EDGE_BATCH_N 16
#endif
constexpr int EDGE_BATCH = EDGE_BATCH_N; // sign-word rows per load batch (SEG_WORDS is a multiple)

__global__ __launch_bounds__(64) void pll_edges_kernel(
    const uint32_t *__restrict__ sgn, uint4 *__restrict__ edges, uint32_t *__restrict__ en4p,
    const uint32_t *__restrict__ prev_in, uint32_t *__restrict__ prev_out,
    uint32_t *__restrict__ prev0, int N, int L)
{
    __shared__ uint32_t stage[64 * EDGE_STW];
    const int lane = threadIdx.x;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int s = blockIdx.y;
    const int W = (L + 31) >> 5;
    const int w0 = s * SEG_WORDS;
    const int w1 = (w0 + SEG_WORDS < W) ? w0 + SEG_WORDS : W;
    // sign of the last sample before the segment (receiver.h:44 prev)
    uint32_t prev = (s == 0) ? (prev_in[c] & 1u) : (sgn[(size_t) (w0 - 1) * (size_t) N + c] & 1u);
    if (s == 0 && live) prev0[cg] = prev;
    uint4 *__restrict__ seg = edges + (size_t) s * EDGE_PAIRS * (size_t) N;   // pair p of lane c: seg[p * N + c]
    uint32_t *__restrict__ strip = stage + lane * EDGE_STW;
    uint16_t *__restrict__ strip16 = reinterpret_cast<uint16_t *>(strip);

    int fill = 0, rowbase = 0;                    // entries in the strip / pairs already in HBM
    auto flush = [&]() {
        const int rows = fill >> 3;
        for (int r = 0; __any(r < rows); ++r)
            if (r < rows && live && !K2T_NOSTORE_V)
                seg[(uint32_t) (1 + rowbase + r) * (uint32_t) N + (uint32_t) c] =
                    make_uint4(strip[4 * r], strip[4 * r + 1], strip[4 * r + 2], strip[4 * r + 3]);
        const uint32_t a = strip[4 * rows], b = strip[4 * rows + 1], d = strip[4 * rows + 2], e = strip[4 * rows + 3];
        strip[0] = a; strip[1] = b; strip[2] = d; strip[3] = e;
        rowbase += rows;
        fill &= 7;
    };
    // the sign words of the next EDGE_BATCH rows are in flight while the current ones are scanned
    // (rows past W are the buffer's spare rows)
    uint32_t Sn[EDGE_BATCH];
#pragma unroll
    for (int q = 0; q < EDGE_BATCH; ++q) Sn[q] = sgn[(size_t) (w0 + q) * (size_t) N + c];
    for (int wb = w0; wb < w1; wb += EDGE_BATCH) {
        uint32_t Sv[EDGE_BATCH];
#pragma unroll
        for (int q = 0; q < EDGE_BATCH; ++q) Sv[q] = Sn[q];
        if (wb + EDGE_BATCH < w1) {
#pragma unroll
            for (int q = 0; q < EDGE_BATCH; ++q) Sn[q] = sgn[(size_t) (wb + EDGE_BATCH + q) * (size_t) N + c];
        }
#pragma unroll
        for (int q = 0; q < EDGE_BATCH; ++q) {
            const int w = wb + q;
            if (w < w1) {
                const uint32_t S = Sv[q];
                const int nv = L - w * 32;        // < 32 only in the call's last word
                uint32_t D = S ^ ((S >> 1) | (prev << 31));           // receiver.c:113
                if (nv < 32) {
                    D &= ~0u << (32 - nv);
                    prev = (S >> (32 - nv)) & 1u;
                } else {
                    prev = S & 1u;
                }
                if (__any(fill > EDGE_FLUSH)) flush();
                const uint32_t tb = (uint32_t) (w - w0) * 32u;
                while (D) {
                    const uint32_t pos = (uint32_t) __clz((int) D);
                    D ^= 0x80000000u >> pos;
#ifndef K2T_NOAPPEND
                    strip16[fill] = (uint16_t) (tb | pos);
#endif
                    ++fill;
                }
            }
        }
    }
    flush();
    // the last, partial pair: k entries in strip[0..3]
    const int k = fill;
    const uint32_t keep = k & 3;                  // entries of the partial group
    uint32_t e0 = strip[0], e1 = strip[1], e2 = strip[2], e3 = strip[3];
    const int cnt = rowbase * 8 + k;
    if (k >= 4 && live) seg[(uint32_t) (1 + rowbase) * (uint32_t) N + (uint32_t) c] = make_uint4(e0, e1, 0, 0);
    uint32_t t0 = k >= 4 ? e2 : e0, t1 = k >= 4 ? e3 : e1;
    if (keep < 3) t1 = 0;
    if (keep == 1) t0 &= 0xffffu;
    if (keep == 0) t0 = 0;
    if (live) seg[c] = make_uint4((uint32_t) cnt, 0, t0, t1);
    const uint32_t n4p = wave_max((uint32_t) (((cnt >> 2) + 1) >> 1));
    if (lane == 0) en4p[(size_t) s * gridDim.x + blockIdx.x] = n4p;
    if (w1 == W && live) prev_out[cg] = prev;
}

// ---- K2a ---------------------------------------------------------------------------------------
// One workgroup = 64 channels (group blockIdx.x) through the whole call, as 2 + PLL_MOVERS waves:
//   wave 0  the recurrence.  It touches only LDS: beside a FIR that keeps the CU's vector memory
//           pipeline full, every global load this wave issued cost it microseconds (round 1: 0.60 ms
//           without memory instructions, 1.3 ms with them, whatever the prefetch distance);
//   movers  stream the lists into an LDS ring, each position t expanded to what the recurrence
//           adds, t * pllinc * 2^7.  A load instruction waits ~0.45 us to be ACCEPTED by the CU's
//           memory pipeline while the FIR's bursts queue in front of it, however many loads the wave
//           has in flight (16 or 32: the same 0.70 ms; without the loads 0.41), and a call is
//           ~1300 loads of 1 KB -- so several waves take turns (batch b belongs to mover b % M) and
//           wait in parallel;
//   writer  takes a finished segment's pack out of LDS: complement, trim to the bit count,
//           64 bytes per lane to HBM, clears the buffer for the segment after next.
// Ring unit = one "group": four consecutive list rows, 16 bytes per lane, so that the recurrence
// fetches four steps with one ds_read_b128.  Stream per segment: header group, a spare group (keeps
// every block even), then 2 * n4p groups.  Monotonic LDS counters hand things over: batches each
// mover has put in the ring / groups consumed, segments finished / packs written out (the two pack buffers).
// The launch asks for more than half a CU's 160 KB of LDS, so the dispatcher places at most ONE of
// these workgroups per CU and two chains never share a SIMD.
constexpr int PLL_RING = 72;         // groups in the ring (even)
constexpr int PLL_SPARE = 4;         // groups after the ring the recurrence may read ahead into
#ifndef PLL_QUEUE_N
#define PLL_QUEUE_N 8
#endif
constexpr int PLL_QUEUE = PLL_QUEUE_N; // loads the mover keeps in flight (divides 64)
#ifndef PLL_MOVERS_N
#define PLL_MOVERS_N 3
#endif
constexpr int PLL_MOVERS = PLL_MOVERS_N;   // mover waves; batch b belongs to mover b % PLL_MOVERS
constexpr int PLL_WAVES = PLL_MOVERS + 2;
constexpr int PLL_BLOCK = 16;        // groups the recurrence takes per hand-over (even); 64 steps, see PLL_STEP
constexpr int PLL_PACKW = PACK_STRIDE + 1;   // words per lane and pack buffer: the pack + its bit count
static_assert(PLL_RING % 2 == 0 && PLL_BLOCK % 2 == 0 && PLL_BLOCK <= PLL_RING, "even blocks");
static_assert(PLL_BLOCK * 4 < 127, "a block's steps must not exhaust the seven spare bits");

// The hand-over counters live in LDS and guard LDS data only.  The LDS unit executes a wave's DS
// instructions in order, so "data, then counter" on the producer side and "counter, then data" on
// the consumer side is all the ordering needed; a C++ release / acquire here would also wait for
// every global load and store the wave has in flight (s_waitcnt vmcnt(0)) -- which is exactly what
// the mover's load queue must not do.
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

// One transition.  X = (pll0 + K) * 2^7 + spare, T = t * pllinc * 2^7, so U = X + T is the
// unwrapped phase * 2^7: bit 22 is `pll >= 0x8000` (receiver.c:114), bits 31:23 the number of the
// slice the transition toggles (word = bits 31:28, bit = 27:23: a pack has 16 words).
//     um = -(pll >= 0x8000);  X = (Q ^ um) + X   is  X + Q  or  X - Q - 1:
// the -1 is taken from the seven spare bits, which are set to all ones at least every 64 steps.
#define PLL_STEP(r)                                                                       \
    "v_add_u32 %[U], %[X], " r "\n\t"                                                     \
    "v_bfe_i32 %[um], %[U], 22, 1\n\t"                                                    \
    "v_lshrrev_b32 %[m], 23, %[U]\n\t"                                                    \
    "v_xad_u32 %[X], %[Q], %[um], %[X]\n\t"                                               \
    "v_lshrrev_b32 %[U], 28, %[U]\n\t"                                                    \
    "v_lshlrev_b32_e64 %[m], %[m], 1\n\t"                                                 \
    "v_lshl_add_u32 %[U], %[U], 8, %[pb]\n\t"                                             \
    "ds_xor_b32 %[U], %[m]\n\t"

// `ng` (even) groups starting at LDS byte address `ad` (this lane's 16 bytes of the first group),
// first row index `i`.  A lane takes part in a group while i < cnt4 (its rows in whole groups):
// v_cmpx narrows EXEC, monotonically within a segment; EXEC is restored on exit.  LDS operations
// complete in order: five are issued after the group a wait is for (four toggles, one read ahead).
__device__ __forceinline__ void pll_groups(uint32_t &X, uint32_t cnt4, uint32_t ad, uint32_t i,
                                           uint32_t ng, uint32_t Q, uint32_t pb)
{
    uint32_t U, um, m;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b128 v[40:43], %[ad]\n\t"
        "ds_read_b128 v[44:47], %[ad] offset:1024\n\t"
        "v_or_b32 %[X], 0x7f, %[X]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "1:\n\t"
        "v_cmpx_lt_u32 vcc, %[i], %[c4]\n\t"
        PLL_STEP("v40") PLL_STEP("v41") PLL_STEP("v42") PLL_STEP("v43")
        "ds_read_b128 v[40:43], %[ad] offset:2048\n\t"
        "s_add_u32 %[i], %[i], 4\n\t"
        "v_cmpx_lt_u32 vcc, %[i], %[c4]\n\t"
        "s_waitcnt lgkmcnt(5)\n\t"
        PLL_STEP("v44") PLL_STEP("v45") PLL_STEP("v46") PLL_STEP("v47")
        "ds_read_b128 v[44:47], %[ad] offset:3072\n\t"
        "s_add_u32 %[i], %[i], 4\n\t"
        "v_add_u32 %[ad], 0x800, %[ad]\n\t"
        "s_sub_u32 %[ng], %[ng], 2\n\t"
        "s_cmp_lg_u32 %[ng], 0\n\t"
        "s_waitcnt lgkmcnt(5)\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [X] "+v"(X), [U] "=&v"(U), [um] "=&v"(um), [m] "=&v"(m), [ad] "+v"(ad), [i] "+s"(i),
          [ng] "+s"(ng), [sv] "=&s"(sv)
        : [c4] "v"(cnt4), [Q] "s"(Q), [pb] "v"(pb)
        : "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
}

__host__ __device__ inline int n_seg_cap(int L)
{
    return (((L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
}

__global__ __launch_bounds__(64 * PLL_WAVES) __attribute__((amdgpu_waves_per_eu(7, 8))) void pll_phase_kernel(
    const uint4 *__restrict__ edges, const uint32_t *__restrict__ en4p, uint32_t *__restrict__ pllst,
    uint32_t *__restrict__ segbits, uint32_t *__restrict__ segcnt, uint32_t *__restrict__ pend,
    uint32_t *__restrict__ watchdog, int N, int L, int n_seg_alloc, uint32_t pllinc)
{
    extern __shared__ uint4 lds4[];
    uint4 *ring = lds4;                                                    // [PLL_RING + PLL_SPARE][64]
    uint32_t *pack = reinterpret_cast<uint32_t *>(lds4 + (PLL_RING + PLL_SPARE) * 64);   // [2][PLL_PACKW][64]
    uint32_t *flag = pack + 2 * PLL_PACKW * 64;
    uint32_t *tbl = flag + 16;                                             // [n_seg] pairs per segment
    uint32_t *pre = tbl + ((n_seg_cap(L) + 15) & ~15);                     // [n_seg + 1] batches before each segment
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int n_seg = n_seg_cap(L);
    // flag[1] groups consumed, [2] segments finished, [3] packs written, [4 + j] batches mover j has put in the ring
    if (threadIdx.x < 16) flag[threadIdx.x] = 0;
    for (int s = threadIdx.x; s < n_seg; s += 64 * PLL_WAVES) tbl[s] = en4p[(size_t) s * gridDim.x + blockIdx.x];
    for (int q = threadIdx.x; q < 2 * PLL_PACKW * 64; q += 64 * PLL_WAVES) pack[q] = 0;
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

    if (role >= 1 && role <= PLL_MOVERS) {        // ---- a mover ----
        const int mj = role - 1;                  // this mover's batches: mj, mj + PLL_MOVERS, ...
        // The stream is a flat sequence of "batches" (16 bytes per lane: a segment's header pair or
        // one list pair).  Loads are issued unconditionally (past the end: a harmless reload of row 0)
        // from a row number that comes out of a lane of `rows` (v_readlane), and only LDS traffic is
        // conditional: straight-line code in which the compiler can count the loads in flight and
        // wait for exactly the oldest (s_waitcnt vmcnt(PLL_QUEUE - 1)), not for all of them.
        const uint4 *__restrict__ src = edges + c;
        const uint32_t K7 = pllinc << 7;          // T = t * pllinc * 2^7 < 2^32 (create refuses pllinc > 14426)
        // pre[s] = batches before segment s, pre[n_seg] = all of them
        int total = 0;
        for (int s0 = 0; s0 < n_seg; s0 += 64) {
            const int s = s0 + lane;
            const int v = s < n_seg ? 1 + (int) tbl[s] : 0;
            int incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            if (s < n_seg) pre[s] = (uint32_t) (total + incl - v);
            total += __shfl(incl, 63);
        }
        total = __builtin_amdgcn_readfirstlane(total);
        if (lane == 0) pre[n_seg] = (uint32_t) total;
        // row of this mover's batch number `base + lane` in the edges array (bit 31: header pair)
        auto rows_of = [&](int base) -> uint32_t {
            const int bl = mj + (base + lane) * PLL_MOVERS;
            if (bl >= total) return 0u;
            int lo = 0, hi = n_seg - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((int) pre[mid] <= bl) lo = mid; else hi = mid - 1;
            }
            const int pr = bl - (int) pre[lo];
            return (uint32_t) (lo * EDGE_PAIRS + pr) | (pr == 0 ? 0x80000000u : 0u);
        };
#ifdef PLL_NOLOAD       /* timing experiment only: no list traffic, results are garbage */
#define PLL_LOAD(dst, mem, r) dst = make_uint4(((r) >> 31) ? 64u : 0x00200010u, 0x00400030u, 0x00600050u, 0x00800070u)
#else
#define PLL_LOAD(dst, mem, r) dst = mem
#endif
        uint4 slot[PLL_QUEUE];
        uint32_t hdr[PLL_QUEUE];                  // != 0: slot holds a segment's header pair
        uint32_t rows = rows_of(0);
        const int mine = total > mj ? (total - mj + PLL_MOVERS - 1) / PLL_MOVERS : 0;   // batches of this mover
        int wpos = 2 * mj, seen = 0, b = 0;       // ring position of the next batch / consumption last read / own batches written
        bool dead = false;
#define PLL_ISSUE(q, idx)                                                                     \
        do {                                                                                  \
            const uint32_t r_ = (uint32_t) __builtin_amdgcn_readlane((int) rows, (idx) & 63); \
            hdr[q] = r_ >> 31;                                                                \
            PLL_LOAD(slot[q], src[(size_t) (r_ & 0x7fffffffu) * (size_t) N], r_);             \
        } while (0)
#define PLL_LO(x) __umul24((x) & 0xffffu, K7)
#define PLL_HI(x) __umul24((x) >> 16, K7)
#define PLL_PUT(q)                                                                            \
        do {                                                                                  \
            const uint4 v_ = slot[q];                                                         \
            wait_space(2);                                                                    \
            const int r_ = wpos % PLL_RING;           /* even */                              \
            const bool h_ = hdr[q] != 0;                                                      \
            ring[r_ * 64 + lane] = make_uint4(h_ ? v_.x : PLL_LO(v_.x), h_ ? v_.y : PLL_HI(v_.x),  \
                                              h_ ? v_.z : PLL_LO(v_.y), h_ ? v_.w : PLL_HI(v_.y)); \
            ring[(r_ + 1) * 64 + lane] = make_uint4(PLL_LO(v_.z), PLL_HI(v_.z), PLL_LO(v_.w), PLL_HI(v_.w)); \
            wpos += 2 * PLL_MOVERS;                                                           \
            ++b;                                                                              \
            lds_flag_store(flag + 4 + mj, (uint32_t) b);                                      \
        } while (0)
        auto wait_space = [&](int groups) {
            while (wpos + groups - seen > PLL_RING && !dead) {
                seen = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 1));
                if (wpos + groups - seen > PLL_RING) {
                    if (expired()) dead = true;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
        };
#pragma unroll
        for (int q = 0; q < PLL_QUEUE; ++q) PLL_ISSUE(q, q);
        while (mine - b >= PLL_QUEUE && !dead) {          // whole rounds
            const int ib = b + PLL_QUEUE;                 // batches this round issues: ib .. ib + PLL_QUEUE - 1
            if ((ib & 63) == 0) rows = rows_of(ib);
#pragma unroll
            for (int q = 0; q < PLL_QUEUE; ++q) {
                PLL_PUT(q);
                PLL_ISSUE(q, ib + q);
            }
        }
        if (b < mine && !dead) {                          // the last, partial round
            const int b0 = b;
#pragma unroll
            for (int q = 0; q < PLL_QUEUE; ++q)
                if (b0 + q < mine && !dead) PLL_PUT(q);
        }
#undef PLL_ISSUE
#undef PLL_PUT
#undef PLL_LO
#undef PLL_HI
        return;
    }

    if (role == PLL_MOVERS + 1) {                 // ---- the writer ----
        int seen = 0;
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
            if (live) {
                uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(segbits + ((size_t) cg * n_seg_alloc + s) * PACK_STRIDE);
#pragma unroll
                for (int k = 0; k < PACK_STRIDE / 4; ++k)
                    dst[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
                segcnt[(size_t) cg * n_seg_alloc + s] = nb;
                pend[(size_t) s * (size_t) N + cg] = pd;
            }
        }
        if (live)
            for (int s = n_seg; s < n_seg_alloc; ++s) segcnt[(size_t) cg * n_seg_alloc + s] = 0;
        return;
    }

    // ---- the recurrence ----
    // a long dependent chain: when it shares a SIMD with other waves it must win every issue slot
    // it can use
    __builtin_amdgcn_s_setprio(3);
    uint32_t X = ((pllst[c] & 0xffffu) << 7) | 0x7fu;          // receiver.h:40 pll, scaled; spare bits set
    const uint32_t Q = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((pllinc / 16u) << 7));   // receiver.c:84,115,117
    const uint32_t K7 = pllinc << 7;
    int rpos = 0, seen = 0, drained = 0;
    bool dead = false;
    auto wait_loaded = [&](int target) {
        while (seen < target && !dead) {
            int lo = 0x7fffffff;                  // every batch below mover j's next one is in the ring
#pragma unroll
            for (int j = 0; j < PLL_MOVERS; ++j) {
                const int nj = j + PLL_MOVERS * __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 4 + j));
                lo = nj < lo ? nj : lo;
            }
            seen = 2 * lo;
            if (seen < target) {
                if (expired()) dead = true;
                __builtin_amdgcn_s_sleep(1);
            }
        }
    };
    // one transition outside the block loop (the count % 4 entries at a segment's end)
    auto step = [&](uint32_t T, uint32_t pb) {
        const uint32_t U = X + T;
        const uint32_t um = (uint32_t) ((int32_t) (U << 9) >> 31);
        X = (Q ^ um) + X;
        __hip_atomic_fetch_xor(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(pack) + ((U >> 28) << 8) + pb),
                               1u << ((U >> 23) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int s = 0; s < n_seg && !dead; ++s) {
        const int ng = 2 * __builtin_amdgcn_readfirstlane((int) tbl[s]);
        const int seg_len = (L - s * SEG_LEN < SEG_LEN) ? L - s * SEG_LEN : SEG_LEN;
        wait_loaded(rpos + 2);
        while (drained < s - 1 && !dead) {                     // pack buffer s & 1 was segment s-2's
            drained = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 3));
            if (drained < s - 1) {
                if (expired()) dead = true;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (dead) break;
        const uint4 ht = ring[(rpos % PLL_RING) * 64 + lane];
        rpos += 2;
        const uint32_t cnt = ht.x & 0xffffu;
        const uint32_t cnt4 = cnt & ~3u, rem = cnt & 3u;
        const uint32_t pb = (uint32_t) (((s & 1) * PLL_PACKW * 64 + lane) * 4);   // byte offset of this lane's pack word 0
        const uint32_t pba = pb + (uint32_t) (reinterpret_cast<char *>(pack) - reinterpret_cast<char *>(lds4));
        int done = 0;
        while (done < ng) {
            const int r = rpos % PLL_RING;
            int blk = ng - done;
            if (blk > PLL_RING - r) blk = PLL_RING - r;
            if (blk > PLL_BLOCK) blk = PLL_BLOCK;
            wait_loaded(rpos + blk);
            if (dead) break;
            pll_groups(X, cnt4, (uint32_t) ((r * 64 + lane) * 16),
                       (uint32_t) __builtin_amdgcn_readfirstlane(done * 4),
                       (uint32_t) __builtin_amdgcn_readfirstlane(blk), Q, pba);
            rpos += blk;
            done += blk;
            lds_flag_store(flag + 1, (uint32_t) rpos);
        }
        if (dead) break;
        X |= 0x7fu;
        if (rem > 0) step(__umul24(ht.z & 0xffffu, K7), pb);
        if (rem > 1) step(__umul24(ht.z >> 16, K7), pb);
        if (rem > 2) step(__umul24(ht.w & 0xffffu, K7), pb);
        const uint32_t Uend = X + (uint32_t) seg_len * K7;     // before the next segment's first sample
        pack[(s & 1) * PLL_PACKW * 64 + PACK_STRIDE * 64 + lane] = Uend >> 23;   // slices so far = bits
        X = (Uend & 0x007fff80u) | 0x7fu;                      // receiver.c:133 pll &= 0xffff
        lds_flag_store(flag + 1, (uint32_t) rpos);
        lds_flag_store(flag + 2, (uint32_t) (s + 1));
    }
    if (live && !dead) pllst[cg] = (X >> 7) & 0xffffu;
}

// after K2a, same stream: a transition after a segment's last slice toggles the first bit of the
// next segment that has one (receiver.c:128), or of a later call: the level at the last slice
// (receiver.h:38 lastbit) is the level of the call's last sample XOR that pending parity.
__global__ void nrzi_carry_kernel(const uint32_t *__restrict__ sgn, uint32_t *__restrict__ segbits,
                                  const uint32_t *__restrict__ segcnt, const uint32_t *__restrict__ pend,
                                  const uint32_t *__restrict__ prev0, uint32_t *__restrict__ lastbit,
                                  int N, int L, int n_seg)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int W = (L + 31) >> 5;
    const int used = (W + SEG_WORDS - 1) / SEG_WORDS;
    uint32_t p = (lastbit[c] ^ prev0[c]) & 1u;      // transitions since the last slice, mod 2
    for (int s = 0; s < used; ++s) {
        const uint32_t pd = pend[(size_t) s * (size_t) N + c] & 1u;
        if (segcnt[(size_t) c * n_seg + s]) {
            if (p) segbits[((size_t) c * n_seg + s) * PACK_STRIDE] ^= 1u;
            p = pd;
        } else {
            p ^= pd;
        }
    }
    const int nv = L - (W - 1) * 32;                // samples in the last word, left-aligned
    const uint32_t lastsign = (sgn[(size_t) (W - 1) * (size_t) N + c] >> (32 - nv)) & 1u;
    lastbit[c] = lastsign ^ p;
}

hipError_t pll_prepare_device()
{
    return hipFuncSetAttribute((const void *) pll_phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024);
}

static int pll_lds_bytes(int n_seg)
{
    return (PLL_RING + PLL_SPARE) * 64 * 16 + 2 * PLL_PACKW * 64 * 4 + 64 +
           (((n_seg + 15) & ~15) + ((n_seg + 16) & ~15)) * 4;
}

hipError_t launch_pll_edges(const PllLaunch &a, hipStream_t stream)
{
    const int used = n_seg_cap(a.L);
    hipLaunchKernelGGL(pll_edges_kernel, dim3((a.N + 63) / 64, used), dim3(64), 0, stream, a.sgn,
                       (uint4 *) a.edges, a.en4p, a.prev_in, a.prev_out, a.prev0, a.N, a.L);
    return hipGetLastError();
}

hipError_t launch_pll_phase(const PllLaunch &a, hipStream_t stream)
{
    // one workgroup per CU while the channel groups fit one round; beyond that share the CUs evenly
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int groups = (a.N + 63) / 64, per_cu = (groups + n_cu - 1) / n_cu;
    const int need = pll_lds_bytes(n_seg_cap(a.L));
    const int lds = per_cu <= 1 ? std::max(need, PLL_LDS_BYTES) : std::max(need, (160 * 1024 / per_cu) & ~1023);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pll_phase_kernel, dim3(groups), dim3(64 * PLL_WAVES), lds, stream, (const uint4 *) a.edges,
                       a.en4p, a.pll, a.segbits, a.segcnt, a.pend, a.watchdog, a.N, a.L, a.n_seg, a.pllinc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nrzi_carry_kernel, dim3((a.N + 255) / 256), dim3(256), 0, stream, a.sgn, a.segbits,
                       a.segcnt, a.pend, a.prev0, a.lastbit, a.N, a.L, a.n_seg);
    return hipGetLastError();
}

} // namespace gnuais
