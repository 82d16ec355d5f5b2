// pll_nrzi.hip -- K2t / K2a / K2x: bit-clock recovery PLL, slice and NRZI decode for gfx950.
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
// Between two transitions g samples apart the phase just advances by g * pllinc (mod 2^16), and
// the number of slices in that stretch is the number of times it wrapped,
// floor((pll + g * pllinc) / 2^16) -- the nudge can never wrap by itself (pll < 0x8000 -> +q stays
// below 0x10000, pll >= 0x8000 -> -q stays above 0) and pllinc + q < 2^16, so no sample wraps
// twice.  The level the slicer sees (receiver.c:126) is constant between transitions, so the NRZI
// bits of a stretch are: the first slice compares the stretch's level with the level at the
// previous slice (receiver.c:128), every further slice of the same stretch gives a 1.
//
// That turns 48 000 dependent steps per channel and call into ~10 000 (one per transition, the
// max over the 64 channels of a wave, re-synchronised every 2048 samples), and everything else
// into work that is parallel over (channel, 2048-sample segment):
//
//   K2t  pll_edges_kernel   (parallel)  sign words -> per (channel, segment) the list of
//        A_j = (gap_j * pllinc) mod 2^16, gap_j = samples since the previous transition (or since
//        the segment start), plus the advance from the last transition to the segment end;
//   K2a  pll_phase_kernel   (sequential in time, lane = channel) walks the lists: three VALU
//        instructions per transition,
//            Y = X + A;  um = Y >> 31 (arithmetic);  X = (Q ^ um) + Y
//        with the 16-bit phase in the top half of X.  (Q ^ um) + Y is Y + Q or Y - Q - 1: the -1
//        lands in the low half, which starts at 0x8000 and is rewritten after every segment, so
//        it never borrows from the phase.  It records the phase at every segment start;
//   K2x  nrzi_bits_kernel   (parallel) replays each segment from its recorded start phase --
//        the same recurrence, now independent per segment -- counts the wraps of every stretch and
//        packs the NRZI bits; nrzi_carry_kernel then carries the level of the last slice across
//        segment boundaries (it decides the first bit of a pack) and into the next call.
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
// grid.x = channel group (64 channels), grid.y = segment.  Per lane the transitions of the
// segment in time order, eight 16-bit entries to a 16-byte "pair" (two groups of four):
//   pair 0            header: .x = count | A_end << 16, .z/.w = the count % 4 entries that do not
//                     fill a group (the sequential kernel takes them one by one at the end)
//   pair 1 + j        entries 8j .. 8j+7 (whole groups only; a trailing half pair holds one group)
// en4p[segment][group] = pairs the longest lane of the wave needs: what K2a streams.
__global__ __launch_bounds__(64) void pll_edges_kernel(
    const uint32_t *__restrict__ sgn, uint4 *__restrict__ edges, uint32_t *__restrict__ en4p,
    const uint32_t *__restrict__ prev_in, uint32_t *__restrict__ prev_out,
    uint32_t *__restrict__ prev0, int N, int L, uint32_t pllinc)
{
    const int lane = threadIdx.x;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int s = blockIdx.y;
    const int W = (L + 31) >> 5;
    const int w0 = s * SEG_WORDS;
    const int w1 = (w0 + SEG_WORDS < W) ? w0 + SEG_WORDS : W;
    const int seg_len = (L - s * SEG_LEN < SEG_LEN) ? L - s * SEG_LEN : SEG_LEN;
    // sign of the last sample before the segment (receiver.h:44 prev)
    uint32_t prev = (s == 0) ? (prev_in[c] & 1u) : (sgn[(size_t) (w0 - 1) * (size_t) N + c] & 1u);
    if (s == 0 && live) prev0[cg] = prev;
    uint4 *__restrict__ seg = edges + (size_t) s * EDGE_PAIRS * (size_t) N + c;

    uint4 acc = make_uint4(0, 0, 0, 0);           // shift register: newest entry enters at the top
    int cnt = 0, tprev = 0;
    for (int wb = w0; wb < w1; wb += 8) {
        uint32_t Sv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)               // rows past W are the buffer's spare rows
            Sv[q] = sgn[(size_t) (wb + q) * (size_t) N + c];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
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
                const int tb = (w - w0) * 32;
                while (D) {
                    const int pos = __clz((int) D);
                    D &= ~(0x80000000u >> pos);
                    const int t = tb + pos;
                    const uint32_t A = ((uint32_t) (t - tprev) * pllinc) & 0xffffu;
                    tprev = t;
                    acc.x = __builtin_amdgcn_alignbit(acc.y, acc.x, 16);
                    acc.y = __builtin_amdgcn_alignbit(acc.z, acc.y, 16);
                    acc.z = __builtin_amdgcn_alignbit(acc.w, acc.z, 16);
                    acc.w = __builtin_amdgcn_alignbit(A, acc.w, 16);
                    ++cnt;
                    if ((cnt & 7) == 0 && live) seg[(size_t) (cnt >> 3) * (size_t) N] = acc;
                }
            }
        }
    }
    // the last, partial pair: bring its k entries down to slots 0..k-1
    const int k = cnt & 7;
    for (int q = k; q < 8 && k; ++q) {
        acc.x = __builtin_amdgcn_alignbit(acc.y, acc.x, 16);
        acc.y = __builtin_amdgcn_alignbit(acc.z, acc.y, 16);
        acc.z = __builtin_amdgcn_alignbit(acc.w, acc.z, 16);
        acc.w >>= 16;
    }
    if (!k) acc = make_uint4(0, 0, 0, 0);
    if (k >= 4 && live) seg[(size_t) (1 + (cnt >> 3)) * (size_t) N] = make_uint4(acc.x, acc.y, 0, 0);
    const uint32_t aend = ((uint32_t) (seg_len - tprev) * pllinc) & 0xffffu;
    if (live)
        seg[0] = make_uint4((uint32_t) cnt | (aend << 16), 0, k >= 4 ? acc.z : acc.x, k >= 4 ? acc.w : acc.y);
    const uint32_t n4p = wave_max((uint32_t) (((cnt >> 2) + 1) >> 1));
    if (lane == 0) en4p[(size_t) s * gridDim.x + blockIdx.x] = n4p;
    if (w1 == W && live) prev_out[cg] = prev;
}

// ---- K2a ---------------------------------------------------------------------------------------
// One workgroup = 64 channels (group blockIdx.x) through the whole call, as TWO waves:
//   wave 0  the recurrence.  It reads only LDS (and stores one word per segment): beside a FIR
//           that keeps the CU's vector memory pipeline full, every global load this wave issued
//           cost it microseconds at ISSUE (measured in round 1: 0.60 ms without memory
//           instructions, 1.3 ms with them, whatever the prefetch distance);
//   wave 1  the mover.  Streams the lists into an LDS ring, expanded to what the recurrence adds
//           (A << 16), PLL_QUEUE loads of 1 KB in flight.
// Ring unit = one "group": four consecutive list rows, 16 bytes per lane, so that the recurrence
// fetches four steps with one ds_read_b128.  Stream per segment: header group, a spare group (keeps
// every block even), then 2 * n4p groups.  Two monotonic LDS counters hand the ring over (groups
// loaded, groups consumed); the LDS unit serves DS instructions in order, so a counter written
// after the data (release) is seen after the data (acquire).
// The launch asks for more than half a CU's 160 KB of LDS, so the dispatcher places at most ONE of
// these workgroups per CU and two chains never share a SIMD.
constexpr int PLL_RING = 72;         // groups in the ring (even)
constexpr int PLL_SPARE = 4;         // groups after the ring the recurrence may read ahead into
constexpr int PLL_QUEUE = 16;        // loads the mover keeps in flight (divides 64)
constexpr int PLL_BLOCK = 32;        // groups the recurrence takes per hand-over (even)
static_assert(PLL_RING % 2 == 0 && PLL_BLOCK % 2 == 0 && PLL_BLOCK <= PLL_RING, "even blocks");

// The hand-over counters live in LDS and guard LDS data only.  The LDS unit executes a wave's DS
// instructions in order, so "data, then counter" on the producer side and "counter, then data" on
// the consumer side is all the ordering needed; a C++ release / acquire here would also wait for
// every global load and store the wave has in flight (s_waitcnt vmcnt(0)) -- which is exactly what
// the mover's load queue and the recurrence's fire-and-forget stores must not do.
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

// four steps of the recurrence on the group in v[R:R+3]
#define PLL_STEP(r)                                                                       \
    "v_add_u32 %[Y], %[X], " r "\n\t"                                                     \
    "v_ashrrev_i32 %[um], 31, %[Y]\n\t"                                                   \
    "v_xad_u32 %[X], %[Q], %[um], %[Y]\n\t"

// `ng` (even) groups starting at LDS byte address `ad` (this lane's 16 bytes of the first group),
// first row index `i`.  A lane takes part in a group while i < cnt4 (its rows in whole groups):
// v_cmpx narrows EXEC, monotonically within a segment; EXEC is restored on exit.
__device__ __forceinline__ void pll_groups(uint32_t &X, uint32_t cnt4, uint32_t ad, uint32_t i,
                                           uint32_t ng, uint32_t Q)
{
    uint32_t Y, um;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b128 v[40:43], %[ad]\n\t"
        "ds_read_b128 v[44:47], %[ad] offset:1024\n\t"
        "1:\n\t"
        "v_cmpx_lt_u32 vcc, %[i], %[c4]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        PLL_STEP("v40") PLL_STEP("v41") PLL_STEP("v42") PLL_STEP("v43")
        "ds_read_b128 v[40:43], %[ad] offset:2048\n\t"
        "s_add_u32 %[i], %[i], 4\n\t"
        "v_cmpx_lt_u32 vcc, %[i], %[c4]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        PLL_STEP("v44") PLL_STEP("v45") PLL_STEP("v46") PLL_STEP("v47")
        "ds_read_b128 v[44:47], %[ad] offset:3072\n\t"
        "s_add_u32 %[i], %[i], 4\n\t"
        "v_add_u32 %[ad], 0x800, %[ad]\n\t"
        "s_sub_u32 %[ng], %[ng], 2\n\t"
        "s_cmp_lg_u32 %[ng], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [X] "+v"(X), [Y] "=&v"(Y), [um] "=&v"(um), [ad] "+v"(ad), [i] "+s"(i), [ng] "+s"(ng),
          [sv] "=&s"(sv)
        : [c4] "v"(cnt4), [Q] "s"(Q)
        : "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
}

__device__ __forceinline__ uint32_t pll_step(uint32_t X, uint32_t A, uint32_t Q)
{
    const uint32_t Y = X + A;
    const uint32_t um = (uint32_t) ((int32_t) Y >> 31);
    return (Q ^ um) + Y;
}

__host__ __device__ inline int n_seg_cap(int L)
{
    return (((L + 31) >> 5) + SEG_WORDS - 1) / SEG_WORDS;
}

__global__ __launch_bounds__(128) void pll_phase_kernel(
    const uint4 *__restrict__ edges, const uint32_t *__restrict__ en4p, uint32_t *__restrict__ xs,
    uint32_t *__restrict__ pllst, uint32_t *__restrict__ watchdog, int N, int L, uint32_t pllinc)
{
    extern __shared__ uint4 lds4[];
    uint4 *ring = lds4;                                                    // [PLL_RING + PLL_SPARE][64]
    uint32_t *flag = reinterpret_cast<uint32_t *>(lds4 + (PLL_RING + PLL_SPARE) * 64);
    uint32_t *tbl = flag + 16;                                             // [n_seg] pairs per segment
    uint32_t *pre = tbl + ((n_seg_cap(L) + 15) & ~15);                     // [n_seg + 1] batches before each segment
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int W = (L + 31) >> 5;
    const int n_seg = (W + SEG_WORDS - 1) / SEG_WORDS;
    if (threadIdx.x < 2) flag[threadIdx.x] = 0;       // 0 groups loaded, 1 groups consumed
    for (int s = threadIdx.x; s < n_seg; s += 128) tbl[s] = en4p[(size_t) s * gridDim.x + blockIdx.x];
    __syncthreads();
    const unsigned long long t_start = wall_clock64();
    // nothing here may spin forever: a wave that waits longer than this gives up (200 ms; the two
    // waves of a workgroup normally hand over every few microseconds) and says so in *watchdog,
    // which the host turns into an error when the frames are drained
    auto expired = [&]() {
        if (wall_clock64() - t_start <= 20000000ull) return false;
        if (lane == 0) atomicOr(watchdog, 1u);
        return true;
    };

    if (role == 1) {                              // ---- the mover ----
        // The stream is a flat sequence of "batches" (16 bytes per lane: a segment's header pair or
        // one list pair).  Loads are issued unconditionally (past the end: a harmless reload of row 0)
        // from a row number that comes out of a lane of `rows` (v_readlane), and only LDS traffic is
        // conditional: straight-line code in which the compiler can count the loads in flight and
        // wait for exactly the oldest (s_waitcnt vmcnt(PLL_QUEUE - 1)), not for all of them.
        const uint4 *__restrict__ src = edges + c;
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
        // row of batch `base + lane` in the edges array (bit 31: header pair)
        auto rows_of = [&](int base) -> uint32_t {
            const int bl = base + lane;
            if (bl >= total) return 0u;
            int lo = 0, hi = n_seg - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((int) pre[mid] <= bl) lo = mid; else hi = mid - 1;
            }
            const int pr = bl - (int) pre[lo];
            return (uint32_t) (lo * EDGE_PAIRS + pr) | (pr == 0 ? 0x80000000u : 0u);
        };
        uint4 slot[PLL_QUEUE];
        uint32_t hdr[PLL_QUEUE];                  // != 0: slot holds a segment's header pair
        uint32_t rows = rows_of(0);
        int wpos = 0, seen = 0, b = 0;            // groups written / consumption last read / batches written
        bool dead = false;
#define PLL_ISSUE(q, idx)                                                                     \
        do {                                                                                  \
            const uint32_t r_ = (uint32_t) __builtin_amdgcn_readlane((int) rows, (idx) & 63); \
            hdr[q] = r_ >> 31;                                                                \
            slot[q] = src[(size_t) (r_ & 0x7fffffffu) * (size_t) N];                          \
        } while (0)
#define PLL_PUT(q)                                                                            \
        do {                                                                                  \
            const uint4 v_ = slot[q];                                                         \
            const int r_ = wpos % PLL_RING;           /* even */                              \
            const bool h_ = hdr[q] != 0;                                                      \
            ring[r_ * 64 + lane] = make_uint4(h_ ? v_.x : v_.x << 16, h_ ? v_.y : v_.x & 0xffff0000u,   \
                                              h_ ? v_.z : v_.y << 16, h_ ? v_.w : v_.y & 0xffff0000u);  \
            ring[(r_ + 1) * 64 + lane] = make_uint4(v_.z << 16, v_.z & 0xffff0000u, v_.w << 16, v_.w & 0xffff0000u); \
            wpos += 2;                                                                        \
            lds_flag_store(flag + 0, (uint32_t) wpos);                                        \
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
        while (total - b >= PLL_QUEUE && !dead) {         // whole rounds
            wait_space(2 * PLL_QUEUE);
            if (dead) break;
            const int ib = b + PLL_QUEUE;                 // batches this round issues: ib .. ib + PLL_QUEUE - 1
            if ((ib & 63) == 0) rows = rows_of(ib);
#pragma unroll
            for (int q = 0; q < PLL_QUEUE; ++q) {
                PLL_PUT(q);
                PLL_ISSUE(q, ib + q);
            }
            b += PLL_QUEUE;
        }
        if (b < total && !dead) {                         // the last, partial round
            wait_space(2 * (total - b));
#pragma unroll
            for (int q = 0; q < PLL_QUEUE; ++q)
                if (b + q < total && !dead) PLL_PUT(q);
        }
#undef PLL_ISSUE
#undef PLL_PUT
        return;
    }

    // ---- the recurrence ----
    // a long dependent chain: when it shares a SIMD with other waves it must win every issue slot
    // it can use
    __builtin_amdgcn_s_setprio(3);
    uint32_t X = ((pllst[c] & 0xffffu) << 16) | 0x8000u;       // receiver.h:40 pll in the top half
    const uint32_t Q = (uint32_t) __builtin_amdgcn_readfirstlane((int) ((pllinc / 16u) << 16));   // receiver.c:84,115,117
    int rpos = 0, seen = 0;
    bool dead = false;
    auto wait_loaded = [&](int target) {
        while (seen < target && !dead) {
            seen = __builtin_amdgcn_readfirstlane((int) lds_flag_load(flag + 0));
            if (seen < target) {
                if (expired()) dead = true;
                __builtin_amdgcn_s_sleep(1);
            }
        }
    };
    for (int s = 0; s < n_seg && !dead; ++s) {
        const int ng = 2 * __builtin_amdgcn_readfirstlane((int) tbl[s]);
        wait_loaded(rpos + 2);
        if (dead) break;
        const uint4 ht = ring[(rpos % PLL_RING) * 64 + lane];
        rpos += 2;
        const uint32_t cnt = ht.x & 0xffffu, aend = ht.x & 0xffff0000u;
        const uint32_t cnt4 = cnt & ~3u, rem = cnt & 3u;
        if (live) xs[(size_t) s * (size_t) N + cg] = X >> 16;  // phase before the segment's first sample
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
                       (uint32_t) __builtin_amdgcn_readfirstlane(blk), Q);
            rpos += blk;
            done += blk;
            lds_flag_store(flag + 1, (uint32_t) rpos);
        }
        if (rem > 0) X = pll_step(X, ht.z << 16, Q);
        if (rem > 1) X = pll_step(X, ht.z & 0xffff0000u, Q);
        if (rem > 2) X = pll_step(X, ht.w << 16, Q);
        X += aend;                                             // last transition -> segment end
        X = (X & 0xffff0000u) | 0x8000u;
        lds_flag_store(flag + 1, (uint32_t) rpos);
    }
    if (live && !dead) pllst[cg] = X >> 16;
}

// ---- K2x ---------------------------------------------------------------------------------------
// grid.x = channel group (64 channels), grid.y = segment of SEG_WORDS words.  Replays the segment
// from xs[segment][channel] with the phase in 16 bits and the wrap count kept: for every stretch
// between transitions `cm` slices at the stretch's level.  The first bit of the pack is written as
// if the level at the previous slice were 0; nrzi_carry_kernel corrects it.
__global__ __launch_bounds__(64) void nrzi_bits_kernel(
    const uint32_t *__restrict__ sgn, const uint32_t *__restrict__ xs, const uint32_t *__restrict__ prev0,
    uint32_t *__restrict__ segbits, uint32_t *__restrict__ segcnt, uint32_t *__restrict__ seglast,
    int N, int L, int n_seg, uint32_t pllinc)
{
    __shared__ uint32_t pack[PACK_STRIDE][64];
    const int lane = threadIdx.x;
    const int cg = blockIdx.x * 64 + lane;
    const int c = cg < N ? cg : N - 1;
    const bool live = cg < N;
    const int s = blockIdx.y;
    const int W = (L + 31) >> 5;
    const int w0 = s * SEG_WORDS;
    const int w1 = (w0 + SEG_WORDS < W) ? w0 + SEG_WORDS : W;
    if (w0 >= W) {
        if (live) {
            segcnt[(size_t) cg * n_seg + s] = 0;
            seglast[(size_t) s * (size_t) N + cg] = 2;
        }
        return;
    }
    const int seg_len = (L - s * SEG_LEN < SEG_LEN) ? L - s * SEG_LEN : SEG_LEN;
    uint32_t prev = (s == 0) ? (prev0[c] & 1u) : (sgn[(size_t) (w0 - 1) * (size_t) N + c] & 1u);
    uint32_t X = xs[(size_t) s * (size_t) N + c] & 0xffffu;
    const uint32_t q = pllinc / 16u;                           // receiver.c:84
    uint32_t level = prev;          // sign of the filter output in the current stretch
    uint32_t ll = 0;                // level at the previous slice (receiver.h:38 lastbit), see above
    uint32_t lastmark = 2;          // level at the segment's last slice, 2 = no slice
    uint32_t outw = 0;
    int outn = 0, wr = 0, tprev = 0;
#pragma unroll
    for (int k = 0; k < PACK_STRIDE; ++k) pack[k][lane] = 0;

    // cm slices at `level`: receiver.c:126-132
#define NRZI_EMIT(cm_)                                                                        \
    do {                                                                                      \
        uint32_t rem_ = (cm_);                                                                \
        if (rem_) {                                                                           \
            uint32_t clr_ = (level ^ ll) & 1u;      /* first bit = !(level ^ lastbit) */      \
            ll = level;                                                                       \
            lastmark = level;                                                                 \
            while (rem_) {                                                                    \
                const uint32_t k_ = rem_ < (uint32_t) (32 - outn) ? rem_ : (uint32_t) (32 - outn); \
                const uint32_t m_ = (k_ >= 32 ? ~0u : ((1u << k_) - 1u)) & ~clr_;             \
                clr_ = 0;                                                                     \
                outw |= m_ << outn;                                                           \
                outn += (int) k_;                                                             \
                rem_ -= k_;                                                                   \
                if (outn == 32) {                                                             \
                    if (wr < PACK_STRIDE) pack[wr][lane] = outw;                              \
                    ++wr;                                                                     \
                    outw = 0;                                                                 \
                    outn = 0;                                                                 \
                }                                                                             \
            }                                                                                 \
        }                                                                                     \
    } while (0)

    for (int wb = w0; wb < w1; wb += 8) {
        uint32_t Sv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) Sv[k] = sgn[(size_t) (wb + k) * (size_t) N + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int w = wb + k;
            if (w < w1) {
                const uint32_t S = Sv[k];
                const int nv = L - w * 32;
                uint32_t D = S ^ ((S >> 1) | (prev << 31));           // receiver.c:113
                if (nv < 32) {
                    D &= ~0u << (32 - nv);
                    prev = (S >> (32 - nv)) & 1u;
                } else {
                    prev = S & 1u;
                }
                const int tb = (w - w0) * 32;
                while (D) {
                    const int pos = __clz((int) D);
                    D &= ~(0x80000000u >> pos);
                    const int t = tb + pos;
                    const uint32_t phi = X + (uint32_t) (t - tprev) * pllinc;   // receiver.c:122
                    tprev = t;
                    X = phi & 0xffffu;                                          // receiver.c:133
                    NRZI_EMIT(phi >> 16);
                    level ^= 1u;
                    X = (X & 0x8000u) ? X - q : X + q;                          // receiver.c:114-117
                }
            }
        }
    }
    {
        const uint32_t phi = X + (uint32_t) (seg_len - tprev) * pllinc;
        NRZI_EMIT(phi >> 16);
    }
#undef NRZI_EMIT
    if (outn && wr < PACK_STRIDE) pack[wr][lane] = outw;
    if (live) {
        uint4 *__restrict__ out = reinterpret_cast<uint4 *>(segbits + ((size_t) cg * n_seg + s) * PACK_STRIDE);
#pragma unroll
        for (int k = 0; k < PACK_STRIDE / 4; ++k)
            out[k] = make_uint4(pack[4 * k][lane], pack[4 * k + 1][lane], pack[4 * k + 2][lane], pack[4 * k + 3][lane]);
        segcnt[(size_t) cg * n_seg + s] = (uint32_t) (wr * 32 + outn);
        seglast[(size_t) s * (size_t) N + cg] = lastmark;
    }
}

// after K2x, same stream: the level at the last slice before each pack decides the pack's first
// bit (receiver.c:128), and the level of the call's last slice is carried into the next call
// (receiver.h:38 lastbit)
__global__ void nrzi_carry_kernel(uint32_t *__restrict__ segbits, const uint32_t *__restrict__ segcnt,
                                  const uint32_t *__restrict__ seglast, uint32_t *__restrict__ lastbit,
                                  int N, int L, int n_seg)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int W = (L + 31) >> 5;
    const int used = (W + SEG_WORDS - 1) / SEG_WORDS;
    uint32_t carry = lastbit[c] & 1u;
    for (int s = 0; s < used; ++s) {
        const uint32_t lm = seglast[(size_t) s * (size_t) N + c];
        if (lm < 2u) {                              // the pack has at least one bit
            if (carry) segbits[((size_t) c * n_seg + s) * PACK_STRIDE] ^= 1u;
            carry = lm;
        }
    }
    lastbit[c] = carry;
}

hipError_t pll_prepare_device()
{
    return hipFuncSetAttribute((const void *) pll_phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024);
}

static int pll_lds_bytes(int n_seg)
{
    return (PLL_RING + PLL_SPARE) * 64 * 16 + 64 + (((n_seg + 15) & ~15) + ((n_seg + 16) & ~15)) * 4;
}

hipError_t launch_pll_edges(const PllLaunch &a, hipStream_t stream)
{
    const int W = (a.L + 31) >> 5, used = (W + SEG_WORDS - 1) / SEG_WORDS;
    hipLaunchKernelGGL(pll_edges_kernel, dim3((a.N + 63) / 64, used), dim3(64), 0, stream, a.sgn,
                       (uint4 *) a.edges, a.en4p, a.prev_in, a.prev_out, a.prev0, a.N, a.L, a.pllinc);
    return hipGetLastError();
}

hipError_t launch_pll_phase(const PllLaunch &a, hipStream_t stream)
{
    // one workgroup per CU while the channel groups fit one round; beyond that share the CUs evenly
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int groups = (a.N + 63) / 64, per_cu = (groups + n_cu - 1) / n_cu;
    const int W = (a.L + 31) >> 5, used = (W + SEG_WORDS - 1) / SEG_WORDS;
    const int need = pll_lds_bytes(used);
    const int lds = per_cu <= 1 ? std::max(need, PLL_LDS_BYTES) : std::max(need, (160 * 1024 / per_cu) & ~1023);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pll_phase_kernel, dim3(groups), dim3(128), lds, stream, (const uint4 *) a.edges,
                       a.en4p, a.xs, a.pll, a.watchdog, a.N, a.L, a.pllinc);
    return hipGetLastError();
}

hipError_t launch_nrzi_bits(const PllLaunch &a, hipStream_t stream)
{
    hipLaunchKernelGGL(nrzi_bits_kernel, dim3((a.N + 63) / 64, a.n_seg), dim3(64), 0, stream, a.sgn, a.xs,
                       a.prev0, a.segbits, a.segcnt, a.seglast, a.N, a.L, a.n_seg, a.pllinc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nrzi_carry_kernel, dim3((a.N + 255) / 256), dim3(256), 0, stream, a.segbits,
                       a.segcnt, a.seglast, a.lastbit, a.N, a.L, a.n_seg);
    return hipGetLastError();
}

} // namespace gnuais
