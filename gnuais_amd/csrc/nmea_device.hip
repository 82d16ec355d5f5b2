// nmea_device.hip -- row f1 on the device: !AIVDM sentences of the frames in the HBM frame ring,
// byte for byte what protodec_generate_nmea() hands to serial_write() (src/protodec.c:780-894,
// called from protodec_getdata() :896-926), in the reference's print order (per channel, time).
//
// The host formatter (nmea.cpp) does 1.6e7 frames/s on sixteen threads; the chain delivers 3.6e8.
// This is per-frame byte work with one piece of per-channel state, the rolling sequence digit
// (d->seqnr, :922-926: +1 per accepted frame, 9 -> 0), so it maps onto sort + scan + gather:
//   1. key = channel << 37 | time stamp (end_bit + 5 bits of the flags byte) for every frame of the ring (the ring holds K3's chunks in
//      whatever order the blocks finished); radix sort (rocPRIM) gives the reference's order;
//   2. per frame, in that order: accepted (AIS type 1..24, :899-900), bytes it will print
//      (nchars + 21 per sentence: 14 of header, 5 of ",f*hh", CR LF), head-of-channel flag;
//   3. ONE scan over (bytes, accepted, head position) with (+, +, max): every frame knows where its
//      text starts and how many accepted frames of its channel came before it, hence its sequence
//      digit (seq0[channel] + that) mod 10;
//   4. a workgroup builds the text of 256 consecutive frames in LDS (one thread per frame) and stores
//      it 16 bytes per lane; the last frame of a channel leaves the channel's new digit.
// When the ring holds exactly one call (gnuais_batch_stream_nmea) step 1 is not a sort at all: K3
// walks its candidates channel by channel in time order, so (K3 block, pass, position) already IS
// the print order, and a scan over K3's chunk table (a few thousand entries) turns a print position
// into a ring index.
// HBM-bound byte shuffling: 64 B read + ~50 B written per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "../../include/gnuais_hip.h"
#include "kernels.h"

namespace gnuais {

namespace {

constexpr int CHARS_PER_SENTENCE = 61;      // protodec.c:793
constexpr int KEY_CH_SHIFT = 37;            // sort key = channel << 37 | 37-bit time stamp
constexpr unsigned MAX_TYPE = 24;           // cfg.h:48 MAX_AIS_PACKET_TYPE

struct FrameView {
    uint32_t w[16];                         // the 64-byte record
    __device__ uint32_t channel() const { return w[0]; }
    __device__ uint32_t end_bit() const { return w[1]; }
    __device__ int nbits() const { return (int) (w[15] >> 16); }
    __device__ unsigned byte(int k) const { return (w[2 + (k >> 2)] >> (8 * (k & 3))) & 0xffu; }   // payload[k]
    // the i-th six-bit group of d->rbuffer, MSB first; only the nbits / 8 whole bytes are ever filled in
    // (protodec.c:133,150-162), everything beyond is 0
    __device__ unsigned six(int i) const
    {
        const int bit = 6 * i, k = bit >> 3, sh = bit & 7;
        const unsigned two = (byte(k) << 8) | (k + 1 < 53 ? byte(k + 1) : 0u);
        unsigned v = (two >> (10 - sh)) & 63u;
        const int valid = (nbits() & ~7) - bit;
        if (valid < 6) v = valid <= 0 ? 0u : (v & ~((1u << (6 - valid)) - 1u));
        return v;
    }
};

__device__ __forceinline__ FrameView load_frame(const gnuais_frame *frames, uint32_t i)
{
    FrameView f;
    const uint4 *p = reinterpret_cast<const uint4 *>(frames + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = p[q];
        f.w[4 * q] = v.x; f.w[4 * q + 1] = v.y; f.w[4 * q + 2] = v.z; f.w[4 * q + 3] = v.w;
    }
    return f;
}

__global__ __launch_bounds__(256) void nmea_keys_kernel(const gnuais_frame *__restrict__ frames, int n,
                                                        uint64_t *__restrict__ keys, uint32_t *__restrict__ idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // print order: channel, then the 37-bit time stamp (end_bit + 5 bits of the flags byte)
    const uint2 h = *reinterpret_cast<const uint2 *>(frames + i);
    const uint32_t hi = (reinterpret_cast<const uint32_t *>(frames + i)[15] >> 9) & 31u;
    keys[i] = ((uint64_t) h.x << KEY_CH_SHIFT) | ((uint64_t) hi << 32) | h.y;
    idx[i] = (uint32_t) i;
}

// sentence geometry of one accepted frame
struct Geo { int fill, nchars, parts; };
__device__ __forceinline__ Geo geometry(int nbits)
{
    Geo g;
    g.fill = (6 - nbits % 6) % 6;                                   // protodec.c:909-915
    g.nchars = (nbits + g.fill) / 6;
    g.parts = g.nchars <= CHARS_PER_SENTENCE ? 1 : (g.nchars + CHARS_PER_SENTENCE - 1) / CHARS_PER_SENTENCE;
    return g;
}

// Per frame, in print order: bytes of text, accepted or not, head-of-channel position -- one struct so
// that ONE scan yields the text offset, the accepted frames before it and the head of its channel.
struct Tri { uint32_t bytes, acc, head; };
struct TriOp {
    __device__ __host__ Tri operator()(const Tri &a, const Tri &b) const
    {
        return Tri{a.bytes + b.bytes, a.acc + b.acc, a.head > b.head ? a.head : b.head};
    }
};

// Exclusive scan of the chunk table's counts, one workgroup: chunk_off[k] = frames in print order before
// chunk k, chunk_off[E] = all of them.  The table is [K3 blocks][P passes] with only the first few passes of
// a block in use: a thread sums one block's row, the row totals are scanned, the thread writes its row.
__global__ __launch_bounds__(256) void chunk_scan_kernel(const uint2 *__restrict__ chunks, int n_blocks, int P,
                                                         uint32_t *__restrict__ chunk_off, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 4) totals[tid] = 0;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 256) {
        const int b = base + tid;
        const uint2 *row = chunks + (size_t) b * P;
        uint32_t v = 0;
        if (b < n_blocks)
            for (int q = 0; q < P; ++q) v += row[q].y;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = carry_s;
        for (int q = 0; q < wave; ++q) before += wsum[q];
        if (b < n_blocks) {
            uint32_t run = before + inc - v;
            for (int q = 0; q < P; ++q) {
                chunk_off[(size_t) b * P + q] = run;
                run += row[q].y;
            }
        }
        __syncthreads();
        if (tid == 255) carry_s = before + inc;
        __syncthreads();
    }
    if (tid == 0) chunk_off[(size_t) n_blocks * P] = carry_s;
}

// ring index of the frame at print position j, from the scanned chunk table
__device__ __forceinline__ uint32_t chunk_lookup(const uint2 *__restrict__ chunks, const uint32_t *__restrict__ chunk_off,
                                                 int E, uint32_t j)
{
    int lo = 0, hi = E - 1;                     // largest k with chunk_off[k] <= j: the chunk that holds j
    while (lo < hi) {                           // (empty chunks share their successor's offset and lose)
        const int mid = (lo + hi + 1) >> 1;
        if (chunk_off[mid] <= j) lo = mid; else hi = mid - 1;
    }
    return chunks[lo].x + (j - chunk_off[lo]);
}

// order: in (sorted ring indices) when chunks == nullptr, else out (computed from the chunk table)
__global__ __launch_bounds__(256) void nmea_meta_kernel(const gnuais_frame *__restrict__ frames,
                                                        uint32_t *__restrict__ order, int n,
                                                        const uint2 *__restrict__ chunks,
                                                        const uint32_t *__restrict__ chunk_off, int E,
                                                        Tri *__restrict__ tri, uint32_t *__restrict__ chan, int n_max)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (chunks) n = (int) chunk_off[E];         // the ring's count lives on the device; the grid covers n_max
    if (j >= n) {
        if (j < n_max) { tri[j] = Tri{0u, 0u, 0u}; chan[j] = ~0u; }
        return;
    }
    uint32_t r, rprev = 0;
    if (chunks) {
        r = chunk_lookup(chunks, chunk_off, E, (uint32_t) j);
        if (j > 0) rprev = chunk_lookup(chunks, chunk_off, E, (uint32_t) j - 1u);
        order[j] = r;
    } else {
        r = order[j];
        if (j > 0) rprev = order[j - 1];
    }
    const uint32_t *f = reinterpret_cast<const uint32_t *>(frames + r);
    const uint32_t ch = f[0];
    const unsigned first = f[2] & 0xffu;                                            // payload[0]
    const int nbits = (int) (f[15] >> 16);
    const unsigned type = nbits >= 6 ? first >> 2 : (nbits > 0 ? (first >> 2) & ~((1u << (6 - nbits)) - 1u) : 0u);
    const bool ok = type >= 1 && type <= MAX_TYPE;
    const Geo g = geometry(nbits);
    Tri t;
    t.bytes = ok ? (uint32_t) (g.nchars + 21 * g.parts) : 0u;
    t.acc = ok ? 1u : 0u;
    t.head = (j > 0 && reinterpret_cast<const uint32_t *>(frames + rprev)[0] != ch) ? (uint32_t) j : 0u;
    tri[j] = t;
    chan[j] = ch;
}

__device__ __forceinline__ char armor(unsigned v) { return (char) (v < 40 ? v + 48 : v + 56); }   // protodec.c:826-830
__device__ __forceinline__ char hexdigit(unsigned v) { return (char) (v < 10 ? '0' + v : 'A' + v - 10); }

constexpr int MAX_FRAME_TEXT = 164;         // two sentences of 82 bytes: 449 bits are at most 75 characters
constexpr int NMEA_WRITE_TEXT = 256 * MAX_FRAME_TEXT + 32;             // nmea_write_kernel's LDS: the text of 256 frames ...
constexpr int NMEA_WRITE_LDS_BYTES = NMEA_WRITE_TEXT + 256 * 17 * 4;   // ... and their records, a row of 17 words per thread
static_assert(NMEA_WRITE_TEXT % 16 == 0, "the frame rows behind the text are dword-aligned");

// 256 consecutive frames of the print order per workgroup: their text is one contiguous piece of the
// output, built in LDS (a thread writes its own sentences byte by byte) and then stored 16 bytes per lane.
__global__ __launch_bounds__(256) void nmea_write_kernel(
    const gnuais_frame *__restrict__ frames, const uint32_t *__restrict__ order, const uint32_t *__restrict__ chan,
    const Tri *__restrict__ tri, const Tri *__restrict__ scan, int n, int n_channels,
    const uint8_t *__restrict__ seq_in, uint8_t *__restrict__ seq_out, char *__restrict__ out,
    unsigned long long out_cap, uint32_t *__restrict__ totals /* [0] offset of the last frame's text, [1] its
    length, [2] sentences, [3] bad channel */, const uint32_t *__restrict__ n_dev)
{
    // dynamic shared memory (NMEA_WRITE_LDS_BYTES at launch): as a static array the 42 KB made the compiler pad this
    // kernel's register request from 32 to 136 per wave (the occupancy it derives from static LDS), and the kernel runs
    // inside the delivery loop beside the chain's stages
    extern __shared__ __attribute__((aligned(16))) char nmea_write_dyn[];
    char *const buf = nmea_write_dyn;
    __shared__ uint32_t s_base, s_end, s_sent;
    const int tid = threadIdx.x;
    if (n_dev) n = (int) *n_dev;
    if ((int) blockIdx.x * 256 >= n) return;
    const int j = blockIdx.x * 256 + tid;
    const bool in = j < n;
    Tri me = {0, 0, 0}, inc = {0, 0, 0};
    uint32_t ch = 0;
    if (in) { me = tri[j]; inc = scan[j]; ch = chan[j]; }
    const uint32_t off = inc.bytes - me.bytes;
    if (tid == 0) { s_base = off; s_sent = 0; }
    if (in && (tid == 255 || j == n - 1)) s_end = inc.bytes;
    __syncthreads();
    const uint32_t base = s_base, shift = base & 15u;
    int parts_written = 0;
    if (in) {
        if (ch >= (uint32_t) n_channels) {
            atomicOr(&totals[3], 1u);
        } else {
            const bool ok = me.bytes != 0;
            const uint32_t h = inc.head;                                  // position of this channel's first frame
            const uint32_t before = (inc.acc - me.acc) - (scan[h].acc - tri[h].acc);   // accepted frames of the channel before j
            const bool last = (j + 1 == n) || chan[j + 1] != ch;
            if (last) seq_out[ch] = (uint8_t) ((seq_in[ch] + before + (ok ? 1u : 0u)) % 10u);
            if (ok) {
                // the frame's 64 bytes go to this thread's row of LDS (17 words apart: no two lanes of a wave share a
                // bank): six() indexes the payload with a run-time byte offset, which on a register copy means
                // scratch memory -- 75 dependent scratch loads per frame, inside the delivery loop
                uint32_t *const fr = reinterpret_cast<uint32_t *>(buf + NMEA_WRITE_TEXT) + tid * 17;
                {
                    const uint4 *p = reinterpret_cast<const uint4 *>(frames + order[j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint4 v = p[q];
                        fr[4 * q] = v.x; fr[4 * q + 1] = v.y; fr[4 * q + 2] = v.z; fr[4 * q + 3] = v.w;
                    }
                }
                const int f_nbits = (int) (fr[15] >> 16);
                auto f_byte = [&](int k) -> unsigned { return (fr[2 + (k >> 2)] >> (8 * (k & 3))) & 0xffu; };
                auto f_six = [&](int i) -> unsigned {          // FrameView::six() on the LDS copy
                    const int bit = 6 * i, k = bit >> 3, sh = bit & 7;
                    const unsigned two = (f_byte(k) << 8) | (k + 1 < 53 ? f_byte(k + 1) : 0u);
                    unsigned v = (two >> (10 - sh)) & 63u;
                    const int valid = (f_nbits & ~7) - bit;
                    if (valid < 6) v = valid <= 0 ? 0u : (v & ~((1u << (6 - valid)) - 1u));
                    return v;
                };
                const Geo g = geometry(f_nbits);
                const char seq = (char) ('0' + (seq_in[ch] + before) % 10u);
                char *o = buf + shift + (off - base);
                int done = 0;
                for (int part = 1; part <= g.parts; ++part) {
                    int k = 0;
                    unsigned x = 0;
                    auto put = [&](char c) { o[k++] = c; x ^= (unsigned char) c; };
                    o[k++] = '!';
                    put('A'); put('I'); put('V'); put('D'); put('M'); put(',');
                    put((char) ('0' + g.parts)); put(',');
                    put((char) ('0' + part)); put(',');
                    if (g.parts > 1) {                                  // protodec.c:847-849: no channel letter
                        put(seq); put(','); put(',');
                    } else {                                            // :857-859: always 'A'
                        put(','); put('A'); put(',');
                    }
                    for (int i = 0; i < CHARS_PER_SENTENCE && done < g.nchars; ++i, ++done) put(armor(f_six(done)));
                    put(',');
                    put((char) ((g.parts > 1 && part == g.parts) ? '0' + g.fill : '0'));
                    o[k++] = '*'; o[k++] = hexdigit(x >> 4); o[k++] = hexdigit(x & 15u);      // :864-869
                    o[k++] = '\r'; o[k++] = '\n';
                    o += k;
                }
                parts_written = g.parts;
            }
        }
        if (j == n - 1) { totals[0] = off; totals[1] = me.bytes; }
    }
    if (parts_written) atomicAdd(&s_sent, (uint32_t) parts_written);
    __syncthreads();
    if (tid == 0 && s_sent) atomicAdd(&totals[2], s_sent);
    // the piece [base, end) of the output = buf[shift, shift + end - base): whole 16-byte units where they
    // lie inside it, single bytes at its two ragged ends (the neighbours' pieces share those units)
    const uint32_t total = s_end - base;
    const unsigned long long g0 = (unsigned long long) base - shift;
    const uint32_t units = (shift + total + 15u) >> 4;
    for (uint32_t u = tid; u < units; u += 256) {
        const uint32_t lo = u * 16u, hi = lo + 16u;
        if (lo >= shift && hi <= shift + total && g0 + hi <= out_cap) {
            *reinterpret_cast<uint4 *>(out + g0 + lo) = *reinterpret_cast<const uint4 *>(buf + lo);
        } else {
            for (uint32_t q = lo; q < hi; ++q)
                if (q >= shift && q < shift + total && g0 + q < out_cap) out[g0 + q] = buf[q];
        }
    }
}

// ---- the stdout line of protodec_getdata() (src/protodec.c:934-985), on the device ------------------
// "ch <id> type <t> mmsi <9 digits>:" + the fields of the per-type decoders (:357-776) + " (!<last
// sentence>)\n" -- the host formatter (nmea.cpp: describe()) restated for one thread per frame.  The only
// part that is not integer formatting is printf's %.Nf of a double: done exactly (the double's
// mantissa times 10^N as a 128-bit integer, shifted, rounded half to even on the exact value -- what
// glibc prints).

constexpr int MSG_LINE_MAX = 512;               // longest line: type 6 / 8 with the weather report, < 460 bytes

struct LineOut {
    char *p;
    int n;
    __device__ void ch(char c) { if (n < MSG_LINE_MAX) p[n] = c; ++n; }
    __device__ void str(const char *t) { while (*t) ch(*t++); }
    // %ld / %d, and %0<width>ld
    __device__ void dec(long long v, int width = 0)
    {
        char t[24];
        int k = 0;
        const bool neg = v < 0;
        unsigned long long u = neg ? 0ull - (unsigned long long) v : (unsigned long long) v;
        if (u <= 0xffffffffull) {           // nearly always: 32-bit division by a constant is a multiply
            uint32_t w = (uint32_t) u;
            do { t[k++] = (char) ('0' + w % 10u); w /= 10u; } while (w);
        } else {
            do { t[k++] = (char) ('0' + u % 10ull); u /= 10ull; } while (u);
        }
        if (neg) ch('-');
        for (int i = k + (neg ? 1 : 0); i < width; ++i) ch('0');
        while (k) ch(t[--k]);
    }
    // printf("%.<prec>f", d) for |d| < 2^40, prec <= 6
    __device__ void fixed(double d, int prec)
    {
        const unsigned long long bits = (unsigned long long) __double_as_longlong(d);
        const bool neg = (bits >> 63) != 0;
        const int ex = (int) ((bits >> 52) & 0x7ffull);
        unsigned long long m = bits & 0xfffffffffffffull;
        int e2;
        if (ex == 0) { e2 = -1074; } else { m |= 1ull << 52; e2 = ex - 1075; }      // |d| = m * 2^e2
        unsigned long long S = 1;
        for (int i = 0; i < prec; ++i) S *= 10ull;
        // P = m * S < 2^73, as hi:lo
        const unsigned long long a = (m & 0xffffffffull) * S, bq = (m >> 32) * S;
        unsigned long long lo = a + (bq << 32), hi = (bq >> 32) + (lo < a ? 1ull : 0ull);
        unsigned long long q;
        if (e2 >= 0) {                       // an integer already (never for the fields printed here)
            q = e2 < 11 ? lo << e2 : ~0ull;
        } else {
            const int sh = -e2;              // q = round_half_even(P / 2^sh)
            unsigned long long rem_hi, rem_lo, half_hi, half_lo;
            if (sh >= 128) {
                q = 0; rem_hi = hi; rem_lo = lo; half_hi = ~0ull; half_lo = ~0ull;     // P < 2^73 << half
            } else if (sh >= 64) {
                const int s2 = sh - 64;
                q = s2 ? hi >> s2 : hi;
                rem_hi = s2 ? hi & ((1ull << s2) - 1ull) : 0ull;
                rem_lo = lo;
                half_hi = s2 ? 1ull << (s2 - 1) : 0ull;
                half_lo = s2 ? 0ull : 1ull << 63;
            } else {
                q = (lo >> sh) | (hi << (64 - sh));          // hi < 2^9 and sh >= 1: no bits lost for P < 2^73, sh >= 10
                rem_hi = 0; rem_lo = lo & ((1ull << sh) - 1ull);
                half_hi = 0; half_lo = 1ull << (sh - 1);
            }
            const bool above = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
            const bool tie = rem_hi == half_hi && rem_lo == half_lo;
            if (above || (tie && (q & 1ull))) ++q;
        }
        if (neg) ch('-');
        unsigned long long ip, fr;
        if (q <= 0xffffffffull) {           // the fields printed here: 32-bit arithmetic
            const uint32_t q32 = (uint32_t) q, s32 = (uint32_t) S;
            ip = q32 / s32; fr = q32 % s32;
        } else {
            ip = q / S; fr = q % S;
        }
        dec((long long) ip);
        if (prec) {
            ch('.');
            char t[8];
            uint32_t f32 = (uint32_t) fr;                      // < 10^6
            for (int i = prec - 1; i >= 0; --i) { t[i] = (char) ('0' + f32 % 10u); f32 /= 10u; }
            for (int i = 0; i < prec; ++i) ch(t[i]);
        }
    }
};

// the frame's payload as the reference's d->rbuffer: bit k, 0 beyond the frame's whole bytes
struct BitsView {
    const FrameView &f;
    int whole;                               // bytes that count (protodec.c:133,150-162)
    __device__ explicit BitsView(const FrameView &fv) : f(fv)
    {
        const int n = fv.nbits();
        whole = (n < 8 * 53 ? n : 8 * 53) >> 3;
    }
    __device__ unsigned byte_at(int k) const { return k < whole ? f.byte(k) : 0u; }
    // `count` (<= 32) bits from `pos`, MSB first (protodec_henten, protodec.c:205-214)
    __device__ unsigned long long get(int pos, int count) const
    {
        if (pos >= 8 * 64) return 0;
        const int k = pos >> 3;
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) v = (v << 8) | byte_at(k + i);
        return (v >> (40 - (pos & 7) - count)) & ((1ull << count) - 1ull);
    }
    __device__ int sget(int pos, int count) const
    {
        unsigned long long v = get(pos, count);
        if ((v >> (count - 1)) & 1ull) v |= ~0ull << count;
        return (int) v;
    }
    // `n` six-bit characters from `pos`, trailing blanks dropped (protodec.c:173-203)
    __device__ void text(int pos, int n, LineOut &o) const
    {
        int last = -1;
        for (int k = 0; k < n; ++k)
            if (get(pos + 6 * k, 6) != 32ull && get(pos + 6 * k, 6) != 0ull) last = k;
        for (int k = 0; k <= last; ++k) {
            const int v = (int) get(pos + 6 * k, 6);
            o.ch(v >= 1 && v <= 31 ? (char) (v + 64) : (v >= 32 ? (char) v : ' '));
        }
    }
};

__device__ const char *ifm_name_dev(int fi)    // appid_ifm, protodec.c:220-273
{
    switch (fi) {
    case 0: return "text-telegram"; case 1: return "application-ack"; case 2: return "iai-fi-capab-interrogation";
    case 3: return "iai-capabi-interrogation"; case 4: return "capability-reply"; case 11: return "tide-weather";
    case 16: return "vts-targets"; case 17: return "ship-waypoints"; case 18: return "advice-of-waypoints";
    case 19: return "extended-ship-data"; case 20: return "berthing-data"; case 21: return "weather-obs-report";
    case 22: return "area-notice-bc"; case 23: return "area-notice-addr"; case 24: return "extended-ship-static";
    case 25: return "dangerous-cargo-info"; case 26: return "environmental"; case 27: return "route-info-bc";
    case 28: return "route-info-addr"; case 29: return "text-description-bc"; case 30: return "text-description-addr";
    case 40: return "persons-on-board";
    default: return "unknown";
    }
}

__device__ void binary_payload_dev(const BitsView &b, int fi, int at, LineOut &o)   // protodec_msg_bin :338-350
{
    if (fi == 40) {                              // protodec_msg_40 :279-285
        o.str(" persons-on-board "); o.dec((int) b.get(at, 13));
    } else if (fi == 11) {                       // protodec_msg_11 :287-336, its offsets as they are
        const int lat = (int) b.get(at, 24), lon = (int) b.get(at + 24, 25);
        const int wind = (int) b.get(at + 40, 7), gust = (int) b.get(at + 47, 7);
        const int wdir = (int) b.get(at + 54, 9), gdir = (int) b.get(at + 63, 9);
        const int temp = (int) b.get(at + 72, 11), hum = (int) b.get(at + 83, 7);
        const int dew = (int) b.get(at + 90, 10), pres = (int) b.get(at + 100, 9) + 800;
        const int tend = (int) b.get(at + 109, 2), vis = (int) b.get(at + 111, 8);
        const int level = (int) b.get(at + 119, 9), wave = (int) b.get(at + 124, 8);
        const int wtemp = (int) b.get(at + 128, 10);
        o.str(" lat "); o.fixed((double) (float) lat / 60000.0, 6);
        o.str(" lon "); o.fixed((double) (float) lon / 60000.0, 6);
        o.str(" wind_speed "); o.dec(wind); o.str("kt wind_gust "); o.dec(gust);
        o.str("kt wind_dir "); o.dec(wdir); o.str(" wind_gust_dir "); o.dec(gdir);
        o.str(" air_temp "); o.fixed((double) (float) temp / 10.0 - 60.0, 1);
        o.str("C rel_humid "); o.dec(hum);
        o.str("% dew_point "); o.fixed((double) (float) dew / 10.0 - 20.0, 1);
        o.str("C pressure "); o.dec(pres); o.str(" pressure_tend "); o.dec(tend);
        o.str(" visib "); o.fixed((double) (float) vis / 10.0, 1);
        o.str("NM water_level "); o.fixed((double) (float) level / 10.0 - 10.0, 1);
        o.str("m wave_height "); o.fixed((double) (float) wave / 10.0, 1);
        o.str("m water_temp "); o.fixed((double) (float) wtemp / 10.0 - 10.0, 1);
        o.ch('C');
    }
}

__device__ void position_fields_dev(int lat, int lon, unsigned course, unsigned sog, int rot, int navstat,
                                    unsigned heading, LineOut &o)
{
    o.str(" lat "); o.fixed((double) (float) lat / 600000.0, 6);
    o.str(" lon "); o.fixed((double) (float) lon / 600000.0, 6);
    o.str(" course "); o.fixed((double) (float) (unsigned short) course / 10.0, 0);
    o.str(" speed "); o.fixed((double) (float) (unsigned short) sog / 10.0, 1);
    o.str(" rateofturn "); o.dec(rot);
    o.str(" navstat "); o.dec(navstat);
    o.str(" heading "); o.dec((int) (unsigned short) heading);
}

// the fields the reference prints for one frame (the switch at protodec.c:936-982)
__device__ void describe_dev(const BitsView &b, unsigned type, int padded_len, LineOut &o)
{
    switch (type) {
    case 1: case 2: case 3:                      // protodec_pos :357-402
        position_fields_dev(b.sget(89, 27), b.sget(61, 28), (unsigned) b.get(116, 12), (unsigned) b.get(50, 10),
                            (int) (signed char) b.get(40, 8), (int) (signed char) b.get(38, 2),
                            (unsigned) b.get(128, 9), o);
        break;
    case 4: {                                    // protodec_4 :404-441
        const float lon = (float) ((double) (float) b.sget(79, 28) / 10000.0 / 60.0);
        const float lat = (float) ((double) (float) b.sget(107, 27) / 10000.0 / 60.0);
        o.str(" date "); o.dec((long long) b.get(40, 12)); o.ch('-'); o.dec((long long) b.get(52, 4)); o.ch('-');
        o.dec((long long) b.get(56, 5));
        o.str(" time "); o.dec((long long) b.get(61, 5), 2); o.ch(':'); o.dec((long long) b.get(66, 6), 2); o.ch(':');
        o.dec((long long) b.get(72, 6), 2);
        o.str(" lat "); o.fixed((double) lat, 6); o.str(" lon "); o.fixed((double) lon, 6);
        break;
    }
    case 5: {                                    // protodec_5 :443-519
        const unsigned A = (unsigned) b.get(240, 9), B = (unsigned) b.get(249, 9);
        const unsigned char C = (unsigned char) b.get(258, 6), D = (unsigned char) b.get(264, 6);
        const unsigned char draught = (unsigned char) b.get(294, 8);
        o.str(" name \""); b.text(112, 20, o); o.str("\" destination \""); b.text(302, 20, o);
        o.str("\" type "); o.dec((int) b.get(232, 8)); o.str(" length "); o.dec((int) (A + B));
        o.str(" width "); o.dec(C + D); o.str(" draught "); o.fixed((double) (float) draught / 10.0, 1);
        break;
    }
    case 6: {                                    // protodec_6 :525-542
        const int dac = (int) b.get(72, 10), fi = (int) b.get(82, 6);
        o.str(" dst_mmsi "); o.dec((long long) b.get(40, 30), 9); o.str(" seq "); o.dec((int) b.get(38, 2));
        o.str(" retransmitted "); o.dec((int) b.get(70, 1)); o.str(" appid "); o.dec((int) b.get(72, 16));
        o.str(" app_dac "); o.dec(dac); o.str(" app_fi "); o.dec(fi);
        if (dac == 1) {
            o.ch('('); o.str(ifm_name_dev(fi)); o.ch(')');
            binary_payload_dev(b, fi, 88, o);
        }
        break;
    }
    case 7: case 13: {                           // protodec_7_13 :549-567
        int pos = 40;
        o.str(" buflen "); o.dec(padded_len); o.str(" pos+32 "); o.dec(pos + 32);
        for (int i = 0; i < 4 && pos + 32 <= padded_len; pos += 32, ++i) {
            o.str(" ack "); o.dec(i + 1); o.str(" (to "); o.dec((long long) b.get(pos, 30), 9); o.str(" seq ");
            o.dec((int) b.get(pos + 30, 2)); o.ch(')');
        }
        break;
    }
    case 8: {                                    // protodec_8 :573-584
        const int dac = (int) b.get(40, 10), fi = (int) b.get(50, 6);
        o.str(" appid "); o.dec((int) b.get(40, 16)); o.str(" app_dac "); o.dec(dac); o.str(" app_fi "); o.dec(fi);
        if (dac == 1) {
            o.ch('('); o.str(ifm_name_dev(fi)); o.ch(')');
            binary_payload_dev(b, fi, 56, o);
        }
        break;
    }
    case 18:                                     // protodec_18 :586-635 (no turn rate / status in class B)
        position_fields_dev(b.sget(85, 27), b.sget(57, 28), (unsigned) b.get(112, 12), (unsigned) b.get(46, 10), 0, 15,
                            (unsigned) b.get(124, 9), o);
        break;
    case 19: {                                   // protodec_19 :637-684
        const unsigned A = (unsigned) b.get(271, 9), B = (unsigned) b.get(280, 9);
        const unsigned char C = (unsigned char) b.get(289, 6), D = (unsigned char) b.get(295, 6);
        o.str(" name \""); b.text(143, 20, o); o.str("\" type "); o.dec((int) b.get(263, 8));
        o.str(" length "); o.dec((int) (A + B)); o.str("  width "); o.dec(C + D);
        break;
    }
    case 20: {                                   // protodec_20 :686-705
        int pos = 40;
        for (int i = 0; i < 4 && pos + 30 < padded_len; pos += 30, ++i) {
            o.str(" reserve "); o.dec(i + 1); o.str(" (ofs "); o.dec((int) b.get(pos, 12)); o.str(" slots ");
            o.dec((int) b.get(pos + 12, 4)); o.str(" timeout "); o.dec((int) b.get(pos + 16, 3)); o.str(" incr ");
            o.dec((int) b.get(pos + 19, 11)); o.ch(')');
        }
        break;
    }
    case 24: {                                   // protodec_24 :707-776
        const int part = (int) b.get(38, 2);
        if (part == 0) { o.str(" name \""); b.text(40, 20, o); o.ch('"'); }
        if (part == 1) {
            const unsigned A = (unsigned) b.get(132, 9), B = (unsigned) b.get(141, 9);
            const unsigned char C = (unsigned char) b.get(150, 6), D = (unsigned char) b.get(156, 6);
            o.str(" callsign \""); b.text(90, 6, o); o.str("\" type "); o.dec((int) b.get(40, 8));
            o.str(" length "); o.dec((int) (A + B)); o.str(" width "); o.dec(C + D);
        }
        break;
    }
    default:
        break;
    }
}

// one thread per frame of the print order: its line into lines[j * MSG_LINE_MAX ..], its length into len[j]
// (0 for frames protodec_getdata() does not accept)
__global__ __launch_bounds__(128) void message_lines_kernel(
    const gnuais_frame *__restrict__ frames, const uint32_t *__restrict__ order, const uint32_t *__restrict__ chan,
    const Tri *__restrict__ tri, const Tri *__restrict__ scan, int n, int n_channels,
    const uint8_t *__restrict__ seq_in, const char *__restrict__ chanid, char *__restrict__ lines,
    uint32_t *__restrict__ len)
{
    const int j = blockIdx.x * 128 + threadIdx.x;
    if (j >= n) return;
    const Tri me = tri[j];
    const uint32_t ch = chan[j];
    if (me.bytes == 0 || ch >= (uint32_t) n_channels) { len[j] = 0; return; }
    const Tri inc = scan[j];
    const uint32_t h = inc.head;
    const uint32_t before = (inc.acc - me.acc) - (scan[h].acc - tri[h].acc);
    const FrameView f = load_frame(frames, order[j]);
    const BitsView b(f);
    const Geo g = geometry(f.nbits());
    LineOut o{lines + (size_t) j * MSG_LINE_MAX, 0};
    const unsigned type = (unsigned) b.get(0, 6);
    o.str("ch "); o.ch(chanid ? chanid[ch] : (char) ('A' + ch % 26u));
    o.str(" type "); o.dec((int) type); o.str(" mmsi "); o.dec((long long) b.get(8, 30), 9); o.ch(':');
    describe_dev(b, type, f.nbits() + g.fill, o);
    // protodec.c:934, 984: only the last sentence is shown
    o.str(" (");
    {
        const int part = g.parts;
        const char seq = (char) ('0' + (seq_in[ch] + before) % 10u);
        const int n0 = o.n;
        o.ch('!');
        o.str("AIVDM,"); o.ch((char) ('0' + g.parts)); o.ch(','); o.ch((char) ('0' + part)); o.ch(',');
        if (g.parts > 1) { o.ch(seq); o.ch(','); o.ch(','); } else { o.ch(','); o.ch('A'); o.ch(','); }
        for (int done = (part - 1) * CHARS_PER_SENTENCE; done < g.nchars && done < part * CHARS_PER_SENTENCE; ++done)
            o.ch(armor(f.six(done)));
        o.ch(',');
        o.ch((char) ((g.parts > 1) ? '0' + g.fill : '0'));
        unsigned x = 0;
        for (int i = n0 + 1; i < o.n && i < MSG_LINE_MAX; ++i) x ^= (unsigned char) o.p[i];
        o.ch('*'); o.ch(hexdigit(x >> 4)); o.ch(hexdigit(x & 15u));
    }
    o.ch(')'); o.ch('\n');
    len[j] = (uint32_t) (o.n < MSG_LINE_MAX ? o.n : MSG_LINE_MAX);
}

// the lines packed back to back: a workgroup takes 64 consecutive lines, gathers them in LDS at their packed
// offsets (dword loads from the fixed-stride scratch, byte stores into LDS) and stores the piece 16 bytes per lane
constexpr int PACK_FRAMES = 64;
__global__ __launch_bounds__(256) void message_pack_kernel(const char *__restrict__ lines, const uint32_t *__restrict__ len,
                                                           const uint32_t *__restrict__ off, int n, char *__restrict__ out,
                                                           unsigned long long out_cap, uint32_t *__restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) char message_pack_dyn[];     // MESSAGE_PACK_LDS bytes at launch (see nmea_write_kernel)
    char *const buf = message_pack_dyn;
    __shared__ uint32_t s_lines;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j0 = blockIdx.x * PACK_FRAMES;
    const int j1 = j0 + PACK_FRAMES < n ? j0 + PACK_FRAMES : n;
    const uint32_t base = off[j0], shift = base & 15u;
    const uint32_t end = off[j1 - 1] + len[j1 - 1];
    if (tid == 0) s_lines = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (int j = j0 + wave; j < j1; j += 4) {
        const uint32_t l = len[j], o = off[j] - base + shift;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(lines + (size_t) j * MSG_LINE_MAX);
        for (uint32_t q = lane; q * 4u < l; q += 64) {
            const uint32_t v = src[q];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (q * 4u + k < l) buf[o + q * 4u + k] = (char) (v >> (8 * k));
        }
        if (lane == 0 && l) ++mine;
    }
    if (mine) atomicAdd(&s_lines, mine);
    __syncthreads();
    if (tid == 0) {
        if (s_lines) atomicAdd(&info[1], s_lines);
        if (j1 == n) info[0] = end;
    }
    const uint32_t total = end - base;
    const unsigned long long g0 = (unsigned long long) base - shift;
    const uint32_t units = (shift + total + 15u) >> 4;
    for (uint32_t u = tid; u < units; u += 256) {
        const uint32_t lo = u * 16u, hi = lo + 16u;
        if (lo >= shift && hi <= shift + total && g0 + hi <= out_cap) {
            *reinterpret_cast<uint4 *>(out + g0 + lo) = *reinterpret_cast<const uint4 *>(buf + lo);
        } else {
            for (uint32_t q = lo; q < hi; ++q)
                if (q >= shift && q < shift + total && g0 + q < out_cap) out[g0 + q] = buf[q];
        }
    }
}

// ---- the position cache of one batch, folded on the device (cache.c:163-384; nmea.cpp: fold()) --------
// Every frame that reaches a cache_*() call of the per-type decoders sets whole groups of a vessel's
// entry (position / static data / name + destination / call sign / persons on board); what the entry
// holds after the batch is, per group, what the LAST such frame in arrival order wrote.  Arrival order
// is the print order, so: key = mmsi << 20 | print position, radix sort, one thread per vessel walking
// its (short) run backwards until every group it can still find has been found.
enum { VG_POS = 1, VG_STATIC = 2, VG_NAME = 4, VG_CALL = 8, VG_PERSONS = 16 };
constexpr uint64_t VKEY_NONE = ~0ull;

__device__ __forceinline__ unsigned vessel_groups(const BitsView &b, unsigned type)
{
    switch (type) {
    case 1: case 2: case 3: case 4: case 18: return VG_POS;
    case 5: return VG_CALL | VG_NAME | VG_STATIC;
    case 19: return VG_NAME | VG_STATIC;
    case 24: return b.get(38, 2) == 0 ? VG_NAME : b.get(38, 2) == 1 ? (VG_CALL | VG_STATIC) : 0u;
    case 6: return (b.get(72, 10) == 1 && b.get(82, 6) == 40) ? VG_PERSONS : 0u;
    case 8: return (b.get(40, 10) == 1 && b.get(50, 6) == 40) ? VG_PERSONS : 0u;
    default: return 0u;
    }
}

// j = position in print order; frames that touch no cache entry sort to the end
__global__ __launch_bounds__(256) void vessel_keys_kernel(const gnuais_frame *__restrict__ frames,
                                                          const uint32_t *__restrict__ order, int n,
                                                          uint64_t *__restrict__ keys, uint32_t *__restrict__ val)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const FrameView f = load_frame(frames, order[j]);
    const BitsView b(f);
    const unsigned type = (unsigned) b.get(0, 6);
    const unsigned g = (type >= 1 && type <= MAX_TYPE) ? vessel_groups(b, type) : 0u;
    keys[j] = g ? ((uint64_t) b.get(8, 30) << 20) | (uint64_t) j : VKEY_NONE;
    val[j] = order[j] ;
}

__global__ __launch_bounds__(256) void vessel_heads_kernel(const uint64_t *__restrict__ keys, int n,
                                                           uint32_t *__restrict__ head)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    head[i] = (k != VKEY_NONE && (i == 0 || (keys[i - 1] >> 20) != (k >> 20))) ? 1u : 0u;
}

__device__ void vessel_text(const BitsView &b, int pos, int nchar, char *dst, int cap)
{
    char t[24];
    LineOut o{t, 0};
    b.text(pos, nchar, o);
    const int n = o.n < cap - 1 ? o.n : cap - 1;
    for (int i = 0; i < cap; ++i) dst[i] = i < n ? t[i] : 0;
}
__device__ void vessel_literal(const char *s, char *dst, int cap)
{
    int n = 0;
    while (s[n] && n < cap - 1) { dst[n] = s[n]; ++n; }
    for (int i = n; i < cap; ++i) dst[i] = 0;
}

// what the frame's cache_*() calls write, for the groups in `todo` (the frame's own groups or a subset)
__device__ void vessel_write_groups(gnuais_vessel &v, const BitsView &b, unsigned type, unsigned todo)
{
    if (todo & VG_POS) {                       // cache_position :204-229
        long latitude, longitude;
        int navstat = 0, hdg = 0;
        unsigned course = 0, sog = 0;
        if (type == 4) {
            latitude = b.sget(107, 27); longitude = b.sget(79, 28);
        } else if (type == 18) {
            latitude = b.sget(85, 27); longitude = b.sget(57, 28); navstat = 15;
            hdg = (int) b.get(124, 9); course = (unsigned) b.get(112, 12); sog = (unsigned) b.get(46, 10);
        } else {
            latitude = b.sget(89, 27); longitude = b.sget(61, 28); navstat = (int) (signed char) b.get(38, 2);
            hdg = (int) b.get(128, 9); course = (unsigned) b.get(116, 12); sog = (unsigned) b.get(50, 10);
        }
        v.set |= GNUAIS_V_POSITION;
        v.lat = (float) ((double) (float) latitude / 600000.0);
        v.lon = (float) ((double) (float) longitude / 600000.0);
        v.hdg = hdg;
        v.course = (float) ((double) (float) (unsigned short) course / 10.0);
        v.sog = (float) ((double) (float) (unsigned short) sog / 10.0);
        v.navstat = navstat;
    }
    if (todo & VG_CALL) {
        v.set |= GNUAIS_V_CALLSIGN;
        vessel_text(b, type == 5 ? 70 : 90, 6, v.callsign, (int) sizeof v.callsign);
    }
    if (todo & VG_NAME) {
        v.set |= GNUAIS_V_DATA | GNUAIS_V_NAME;
        vessel_text(b, type == 5 ? 112 : type == 19 ? 143 : 40, 20, v.name, (int) sizeof v.name);
        if (type == 5) vessel_text(b, 302, 20, v.destination, (int) sizeof v.destination);
        else vessel_literal("CLASS B", v.destination, (int) sizeof v.destination);
    }
    if (todo & VG_STATIC) {
        v.set |= GNUAIS_V_DATA | GNUAIS_V_STATIC;
        if (type == 5) {
            const unsigned char draught = (unsigned char) b.get(294, 8);
            v.imo = (int) b.get(40, 30); v.shiptype = (int) b.get(232, 8);
            v.A = (int) b.get(240, 9); v.B = (int) b.get(249, 9);
            v.C = (unsigned char) b.get(258, 6); v.D = (unsigned char) b.get(264, 6);
            v.draught = (float) ((double) draught / 10.0);
        } else if (type == 19) {
            v.imo = 0; v.shiptype = (int) b.get(263, 8); v.A = (int) b.get(271, 9); v.B = (int) b.get(280, 9);
            v.C = (unsigned char) b.get(289, 6); v.D = (unsigned char) b.get(295, 6); v.draught = 0;
        } else {
            v.imo = 0; v.shiptype = (int) b.get(40, 8); v.A = (int) b.get(132, 9); v.B = (int) b.get(141, 9);
            v.C = (unsigned char) b.get(150, 6); v.D = (unsigned char) b.get(156, 6); v.draught = 0;
        }
    }
    if (todo & VG_PERSONS) {                   // protodec_msg_40 :279-285
        v.set |= GNUAIS_V_PERSONS;
        v.persons_on_board = (int) b.get(type == 6 ? 88 : 56, 13);
    }
}

__global__ __launch_bounds__(128) void vessel_fold_kernel(const gnuais_frame *__restrict__ frames,
                                                          const uint64_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ val,
                                                          const uint32_t *__restrict__ head,
                                                          const uint32_t *__restrict__ hidx, int n,
                                                          gnuais_vessel *__restrict__ out, int cap,
                                                          uint32_t *__restrict__ count)
{
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n || !head[i]) return;
    const uint32_t slot = hidx[i];                 // vessels before this one
    atomicMax(count, slot + 1u);
    if ((int) slot >= cap) return;
    const uint64_t mm = keys[i] >> 20;
    int e = i;
    while (e + 1 < n && (keys[e + 1] >> 20) == mm) ++e;        // the vessel's run: [i, e]
    gnuais_vessel v;
    {
        uint32_t *z = reinterpret_cast<uint32_t *>(&v);
        for (unsigned q = 0; q < sizeof v / 4; ++q) z[q] = 0;
    }
    v.mmsi = (int32_t) mm;                         // cache_get's new entry, cache.c:175-196
    v.hdg = -1; v.course = -1; v.sog = -1; v.shiptype = -1; v.imo = -1; v.navstat = -1;
    v.A = v.B = v.C = v.D = -1;
    v.persons_on_board = -1;
    unsigned found = 0;
    for (int k = e; k >= i && found != 31u; --k) {
        const FrameView f = load_frame(frames, val[k]);
        const BitsView b(f);
        const unsigned type = (unsigned) b.get(0, 6);
        const unsigned todo = vessel_groups(b, type) & ~found;
        if (!todo) continue;
        found |= todo;
        vessel_write_groups(v, b, type, todo);
    }
    out[slot] = v;
}

// ---- the position cache CARRIED on the device from batch to batch (cache.c:163-384 across calls) --------
// An open-addressing table keyed by MMSI (slot key = mmsi + 1, 0 = free; linear probing, nothing is ever
// removed -- the reference's cache only expires entries by age, from another thread).  A span of frames is
// folded into it without any sort: what an entry holds afterwards is, per group, what the LAST frame of the
// span in arrival order wrote, and arrival order inside a span is the print order (channel, then the 37-bit
// time stamp), which every record carries.  So: pass 1, every frame finds or creates its vessel's slot and
// raises the slot's stamp of each group it writes to its own (atomicMax); pass 2, the frame whose stamp
// stands writes the group (distinct groups = distinct fields, one writer each) and clears the stamp.
constexpr uint32_t VT_NONE = 0xffffffffu;

__device__ __forceinline__ void vessel_fresh(gnuais_vessel &v, int32_t mmsi)      // cache_get's new entry, cache.c:175-196
{
    uint32_t *z = reinterpret_cast<uint32_t *>(&v);
    for (unsigned q = 0; q < sizeof v / 4; ++q) z[q] = 0;
    v.mmsi = mmsi;
    v.hdg = -1; v.course = -1; v.sog = -1; v.shiptype = -1; v.imo = -1; v.navstat = -1;
    v.A = v.B = v.C = v.D = -1;
    v.persons_on_board = -1;
}

__device__ __forceinline__ unsigned long long vessel_stamp(const gnuais_frame *f)
{
    const uint2 h = *reinterpret_cast<const uint2 *>(f);
    const uint32_t hi = (reinterpret_cast<const uint32_t *>(f)[15] >> 9) & 31u;
    return (((unsigned long long) h.x << 37) | ((unsigned long long) hi << 32) | h.y) + 1ull;
}

// info: [0] slots in use, [1] a frame found no slot (table full), [2] frames seen
__global__ __launch_bounds__(256) void vtable_stamp_kernel(const gnuais_frame *__restrict__ frames,
                                                           const uint32_t *__restrict__ count, uint32_t n_max,
                                                           uint32_t *__restrict__ keys, uint32_t mask,
                                                           gnuais_vessel *__restrict__ ent,
                                                           unsigned long long *__restrict__ stamps,
                                                           uint32_t *__restrict__ fslot, uint32_t *__restrict__ info)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = count[0] < n_max ? count[0] : n_max;
    if (i >= n) return;
    if (i == 0) atomicAdd(info + 2, n);
    const FrameView f = load_frame(frames, i);
    const BitsView b(f);
    const unsigned type = (unsigned) b.get(0, 6);
    const unsigned g = (type >= 1 && type <= MAX_TYPE) ? vessel_groups(b, type) : 0u;
    uint32_t slot = VT_NONE;
    if (g) {
        const uint32_t mmsi = (uint32_t) b.get(8, 30), want = mmsi + 1u;
        uint32_t h = (mmsi * 2654435761u) >> 7 & mask;
        for (uint32_t probes = 0; probes <= mask; ++probes, h = (h + 1u) & mask) {
            uint32_t k = __hip_atomic_load(keys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == 0u) {
                k = atomicCAS(keys + h, 0u, want);
                if (k == 0u) {                     // this thread made the entry: cache_get's defaults
                    gnuais_vessel v;
                    vessel_fresh(v, (int32_t) mmsi);
                    ent[h] = v;
                    atomicAdd(info + 0, 1u);
                    k = want;
                }
            }
            if (k == want) { slot = h; break; }
        }
        if (slot == VT_NONE) atomicOr(info + 1, 1u);
        else {
            const unsigned long long st = vessel_stamp(frames + i);
            for (unsigned gi = 0; gi < 5; ++gi)
                if (g >> gi & 1u) atomicMax(stamps + (size_t) slot * 5 + gi, st);
        }
    }
    fslot[i] = slot;
}

__global__ __launch_bounds__(256) void vtable_apply_kernel(const gnuais_frame *__restrict__ frames,
                                                           const uint32_t *__restrict__ count, uint32_t n_max,
                                                           gnuais_vessel *__restrict__ ent,
                                                           unsigned long long *__restrict__ stamps,
                                                           const uint32_t *__restrict__ fslot)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = count[0] < n_max ? count[0] : n_max;
    if (i >= n) return;
    const uint32_t slot = fslot[i];
    if (slot == VT_NONE) return;
    const unsigned long long st = vessel_stamp(frames + i);
    unsigned win = 0;
    for (unsigned gi = 0; gi < 5; ++gi)
        if (stamps[(size_t) slot * 5 + gi] == st) win |= 1u << gi;
    if (!win) return;
    const FrameView f = load_frame(frames, i);
    const BitsView b(f);
    const unsigned type = (unsigned) b.get(0, 6);
    win &= vessel_groups(b, type);                 // (a stamp is the frame's own: this only guards the reads below)
    gnuais_vessel t;
    t.set = 0;
    vessel_write_groups(t, b, type, win);
    gnuais_vessel *e = ent + slot;
    if (win & VG_POS) {
        e->lat = t.lat; e->lon = t.lon; e->hdg = t.hdg; e->course = t.course; e->sog = t.sog; e->navstat = t.navstat;
    }
    if (win & VG_CALL)
        for (unsigned q = 0; q < sizeof t.callsign; ++q) e->callsign[q] = t.callsign[q];
    if (win & VG_NAME)
        for (unsigned q = 0; q < sizeof t.name; ++q) { e->name[q] = t.name[q]; e->destination[q] = t.destination[q]; }
    if (win & VG_STATIC) {
        e->imo = t.imo; e->shiptype = t.shiptype; e->A = t.A; e->B = t.B; e->C = t.C; e->D = t.D; e->draught = t.draught;
    }
    if (win & VG_PERSONS) e->persons_on_board = t.persons_on_board;
    atomicOr(&e->set, t.set);
    for (unsigned gi = 0; gi < 5; ++gi)
        if (win >> gi & 1u) stamps[(size_t) slot * 5 + gi] = 0ull;
}

// records gathered into sorted order: one thread moves one 16-byte quarter of a record
__global__ __launch_bounds__(256) void frames_gather_kernel(const gnuais_frame *__restrict__ frames,
                                                            const uint32_t *__restrict__ order, int n,
                                                            gnuais_frame *__restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int j = t >> 2, q = t & 3;
    if (j >= n) return;
    reinterpret_cast<uint4 *>(out + j)[q] = reinterpret_cast<const uint4 *>(frames + order[j])[q];
}

struct MaxOp {
    __device__ __host__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

} // namespace

size_t nmea_scratch_bytes(int n, int n_chunks)
{
    const size_t m = (size_t) (n > 0 ? n : 1);
    size_t sort_tmp = 0, scan_tmp = 0;
    (void) rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint64_t *) nullptr, (uint64_t *) nullptr,
                                     (uint32_t *) nullptr, (uint32_t *) nullptr, m, 0, 64, (hipStream_t) 0);
    (void) rocprim::inclusive_scan(nullptr, scan_tmp, (Tri *) nullptr, (Tri *) nullptr, m, TriOp(), (hipStream_t) 0);
    size_t scan_tmp2 = 0;
    (void) rocprim::exclusive_scan(nullptr, scan_tmp2, (uint32_t *) nullptr, (uint32_t *) nullptr, 0u, m,
                                   rocprim::plus<uint32_t>(), (hipStream_t) 0);
    size_t tmp = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    tmp = tmp > scan_tmp2 ? tmp : scan_tmp2;
    // keys x2, idx x2, chan, tri x2, chunk offsets, totals, rocPRIM temp (256-byte slots)
    return 2 * 8 * m + 3 * 4 * m + 2 * sizeof(Tri) * m + 4 * ((size_t) n_chunks + 1) + 256 * 12 + tmp + 64;
}

// The ring's records in the reference's print order (channel, then time = end_bit), on the device:
// what gnuais_batch_drain_frames() copies out.
hipError_t frames_sort(const gnuais_frame *frames, int n, gnuais_frame *out, void *scratch, size_t scratch_bytes,
                       hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    if (scratch_bytes < nmea_scratch_bytes(n)) return hipErrorInvalidValue;
    const size_t m = (size_t) n;
    char *p = static_cast<char *>(scratch);
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) / 256 * 256; return (void *) q; };
    uint64_t *keys = (uint64_t *) take(8 * m), *keys2 = (uint64_t *) take(8 * m);
    uint32_t *idx = (uint32_t *) take(4 * m), *idx2 = (uint32_t *) take(4 * m);
    void *tmp = p;
    size_t t = scratch_bytes - (size_t) (p - static_cast<char *>(scratch));
    hipLaunchKernelGGL(nmea_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, frames, n, keys, idx);
    hipError_t e = rocprim::radix_sort_pairs(tmp, t, keys, keys2, idx, idx2, m, 0, 61, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(frames_gather_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, s, frames, idx2, n, out);
    return hipGetLastError();
}

// The slot's eight info words, kept on the device: the formatter's four (offset and length of the last
// frame's text, sentences, bad-channel flag; zero when nothing was formatted) and the ring's four counters.
__global__ void slot_info_kernel(const uint32_t *__restrict__ totals, const uint32_t *__restrict__ ring_count,
                                 uint32_t *__restrict__ info8)
{
    if (threadIdx.x < 4) {
        info8[threadIdx.x] = totals ? totals[threadIdx.x] : 0u;
        info8[4 + threadIdx.x] = ring_count[threadIdx.x];
    }
}

// device text -> pinned host memory with the length taken on the device, plus the info words: nothing on
// the host has to know a size before it hands the text out.  Few workgroups on purpose: the copy moves at
// PCIe speed whatever its width, and every CU it occupies is a CU whose other waves queue behind its stores.
__global__ __launch_bounds__(64) void text_copy_kernel(const char *__restrict__ src, const uint32_t *__restrict__ info_dev,
                                                       char *__restrict__ dst, unsigned long long dst_cap,
                                                       uint32_t *__restrict__ info)
{
    const unsigned long long len = (unsigned long long) info_dev[0] + info_dev[1];
    const unsigned long long take = len < dst_cap ? len : dst_cap;          // both buffers are 16-byte multiples
    const unsigned long long units = (take + 15ull) >> 4;
    for (unsigned long long u = (unsigned long long) blockIdx.x * 64ull + threadIdx.x; u < units;
         u += (unsigned long long) gridDim.x * 64ull)
        reinterpret_cast<uint4 *>(dst)[u] = reinterpret_cast<const uint4 *>(src)[u];
    if (blockIdx.x == 0 && threadIdx.x < 8) info[threadIdx.x] = info_dev[threadIdx.x];
}

hipError_t nmea_slot_info_enqueue(const uint32_t *totals, const uint32_t *ring_count, uint32_t *info8_dev, hipStream_t s)
{
    hipLaunchKernelGGL(slot_info_kernel, dim3(1), dim3(64), 0, s, totals, ring_count, info8_dev);
    return hipGetLastError();
}

hipError_t nmea_text_copy_enqueue(const char *src, const uint32_t *info8_dev, char *dst_pinned, size_t dst_cap,
                                  uint32_t *info8_pinned, int workgroups, hipStream_t s)
{
    hipLaunchKernelGGL(text_copy_kernel, dim3(workgroups > 0 ? workgroups : 64), dim3(64), 0, s, src, info8_dev,
                       dst_pinned, (unsigned long long) (dst_cap & ~(size_t) 15), info8_pinned);
    return hipGetLastError();
}

// how nmea_format_enqueue() cuts up its scratch: the message-line pass reads the same arrays
struct NmeaLayout {
    uint64_t *keys, *keys2;
    uint32_t *idx, *idx2, *chan;
    Tri *tri, *scan;
    uint32_t *chunk_off, *totals;
    void *tmp;
    size_t tmp_bytes;
};
static NmeaLayout nmea_layout(void *scratch, size_t scratch_bytes, size_t m, int n_chunks)
{
    NmeaLayout l;
    char *p = static_cast<char *>(scratch);
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) / 256 * 256; return (void *) q; };
    l.keys = (uint64_t *) take(8 * m); l.keys2 = (uint64_t *) take(8 * m);
    l.idx = (uint32_t *) take(4 * m); l.idx2 = (uint32_t *) take(4 * m);
    l.chan = (uint32_t *) take(4 * m);
    l.tri = (Tri *) take(sizeof(Tri) * m); l.scan = (Tri *) take(sizeof(Tri) * m);
    l.chunk_off = (uint32_t *) take(4 * ((size_t) n_chunks + 1));
    l.totals = (uint32_t *) take(16);
    l.tmp = p;
    l.tmp_bytes = scratch_bytes - (size_t) (p - static_cast<char *>(scratch));
    return l;
}

// The stdout lines of the same frames, after nmea_format_enqueue(frames, n > 0, ...) on the same scratch and
// stream (its order, acceptance flags and sequence-digit prefix sums are reused; seq_in is the digit array it was
// given).  lines: n * 512 bytes, len / off: n words each; info2 (device): [0] bytes, [1] lines.
size_t messages_line_bytes() { return (size_t) MSG_LINE_MAX; }

hipError_t messages_format_enqueue(const gnuais_frame *frames, int n, int n_channels, const uint8_t *seq_in,
                                   const char *chanid_dev, void *scratch, size_t scratch_bytes, char *lines,
                                   uint32_t *len, uint32_t *off, char *out, size_t out_cap, uint32_t *info2,
                                   hipStream_t s)
{
    if (n <= 0) return hipErrorInvalidValue;
    const NmeaLayout lay = nmea_layout(scratch, scratch_bytes, (size_t) n, 0);
    hipError_t e;
    if ((e = hipMemsetAsync(info2, 0, 8, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(message_lines_kernel, dim3((n + 127) / 128), dim3(128), 0, s, frames, lay.idx2, lay.chan,
                       lay.tri, lay.scan, n, n_channels, seq_in, chanid_dev, lines, len);
    size_t t = lay.tmp_bytes;
    if ((e = rocprim::exclusive_scan(lay.tmp, t, len, off, 0u, (size_t) n, rocprim::plus<uint32_t>(), s)) != hipSuccess)
        return e;
    constexpr size_t MESSAGE_PACK_LDS = PACK_FRAMES * MSG_LINE_MAX + 32;
    hipLaunchKernelGGL(message_pack_kernel, dim3((n + PACK_FRAMES - 1) / PACK_FRAMES), dim3(256), MESSAGE_PACK_LDS, s, lines, len, off, n, out,
                       (unsigned long long) out_cap, info2);
    return hipGetLastError();
}

// The batch's vessel table (struct cache_ent per MMSI, sorted by MMSI) from the frames of the ring, on the
// device; *count_dev = vessels found (may exceed cap: then only the first cap were written).
hipError_t vessels_fold_enqueue(const gnuais_frame *frames, int n, void *scratch, size_t scratch_bytes,
                                gnuais_vessel *out, int cap, uint32_t *count_dev, hipStream_t s)
{
    if (n <= 0) return hipErrorInvalidValue;
    if (scratch_bytes < nmea_scratch_bytes(n, 0)) return hipErrorInvalidValue;
    const size_t m = (size_t) n;
    const NmeaLayout lay = nmea_layout(scratch, scratch_bytes, m, 0);
    const int grid = (n + 255) / 256;
    hipError_t e;
    if ((e = hipMemsetAsync(count_dev, 0, 4, s)) != hipSuccess) return e;
    // print order (channel, then time), as for the sentences
    hipLaunchKernelGGL(nmea_keys_kernel, dim3(grid), dim3(256), 0, s, frames, n, lay.keys, lay.idx);
    size_t t = lay.tmp_bytes;
    if ((e = rocprim::radix_sort_pairs(lay.tmp, t, lay.keys, lay.keys2, lay.idx, lay.idx2, m, 0, 61, s)) != hipSuccess)
        return e;
    // (mmsi, print position) of the frames that touch the cache; ring index as the value
    hipLaunchKernelGGL(vessel_keys_kernel, dim3(grid), dim3(256), 0, s, frames, lay.idx2, n, lay.keys, lay.idx);
    t = lay.tmp_bytes;
    if ((e = rocprim::radix_sort_pairs(lay.tmp, t, lay.keys, lay.keys2, lay.idx, lay.chan, m, 0, 64, s)) != hipSuccess)
        return e;
    hipLaunchKernelGGL(vessel_heads_kernel, dim3(grid), dim3(256), 0, s, lay.keys2, n, lay.idx2);
    t = lay.tmp_bytes;
    if ((e = rocprim::exclusive_scan(lay.tmp, t, lay.idx2, lay.idx, 0u, m, rocprim::plus<uint32_t>(), s)) != hipSuccess)
        return e;
    hipLaunchKernelGGL(vessel_fold_kernel, dim3((n + 127) / 128), dim3(128), 0, s, frames, lay.keys2, lay.chan, lay.idx2,
                       lay.idx, n, out, cap, count_dev);
    return hipGetLastError();
}

static_assert(sizeof(((gnuais_vessel *) 0)->name) == sizeof(((gnuais_vessel *) 0)->destination), "copied together");

size_t vessel_table_bytes(uint32_t slots)
{
    return (size_t) slots * (4 + sizeof(gnuais_vessel) + 5 * 8) + 64;
}

// the table's arrays inside one allocation of vessel_table_bytes(slots): stamps, entries, keys, info
static void vessel_table_layout(void *mem, uint32_t slots, unsigned long long **stamps, gnuais_vessel **ent,
                                uint32_t **keys, uint32_t **info)
{
    char *p = static_cast<char *>(mem);
    *stamps = (unsigned long long *) p; p += (size_t) slots * 5 * 8;
    *ent = (gnuais_vessel *) p; p += (size_t) slots * sizeof(gnuais_vessel);
    *keys = (uint32_t *) p; p += (size_t) slots * 4;
    *info = (uint32_t *) p;
}

// Folds the ring's frames (count[0] of them, at most n_max; the count is read on the device) into the table.
// fslot: n_max words of scratch.  Queued on `s`; nothing is waited for.
hipError_t vessel_table_update_enqueue(const gnuais_frame *frames, const uint32_t *count, int n_max, void *table,
                                       uint32_t slots, uint32_t *fslot, hipStream_t s)
{
    if (n_max <= 0 || !slots || (slots & (slots - 1))) return hipErrorInvalidValue;
    unsigned long long *stamps; gnuais_vessel *ent; uint32_t *keys, *info;
    vessel_table_layout(table, slots, &stamps, &ent, &keys, &info);
    const int grid = (n_max + 255) / 256;
    hipLaunchKernelGGL(vtable_stamp_kernel, dim3(grid), dim3(256), 0, s, frames, count, (uint32_t) n_max, keys, slots - 1u,
                       ent, stamps, fslot, info);
    hipLaunchKernelGGL(vtable_apply_kernel, dim3(grid), dim3(256), 0, s, frames, count, (uint32_t) n_max, ent, stamps, fslot);
    return hipGetLastError();
}

// The table's entries (unsorted, `max` at most) and its info words to the host; the stream is waited for.
hipError_t vessel_table_fetch(const void *table, uint32_t slots, gnuais_vessel *h_out, int max, int *n_out,
                              uint32_t h_info[4], hipStream_t s)
{
    unsigned long long *stamps; gnuais_vessel *ent; uint32_t *keys, *info;
    vessel_table_layout(const_cast<void *>(table), slots, &stamps, &ent, &keys, &info);
    hipError_t e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    if ((e = hipMemcpy(h_info, info, 16, hipMemcpyDeviceToHost)) != hipSuccess) return e;
    *n_out = (int) h_info[0];
    if ((int) h_info[0] > max) return hipSuccess;               // the caller reports the size needed
    std::vector<uint32_t> k(slots);
    if ((e = hipMemcpy(k.data(), keys, (size_t) slots * 4, hipMemcpyDeviceToHost)) != hipSuccess) return e;
    int n = 0;
    for (uint32_t h = 0; h < slots; ) {              // runs of occupied slots, one copy each
        if (!k[h]) { ++h; continue; }
        uint32_t h1 = h;
        while (h1 < slots && k[h1]) ++h1;
        if ((e = hipMemcpy(h_out + n, ent + h, (size_t) (h1 - h) * sizeof(gnuais_vessel), hipMemcpyDeviceToHost)) != hipSuccess)
            return e;
        n += (int) (h1 - h);
        h = h1;
    }
    *n_out = n;
    return hipSuccess;
}

// everything of nmea_format() that runs on the device, queued on `s` without waiting for it.
// n > 0: the host knows the count; the order comes from a radix sort.  n < 0: the ring holds exactly one
// call, the order AND the count (at most n_max) come from K3's chunk table -- no host value needed.
hipError_t nmea_format_enqueue(const gnuais_frame *frames, int n, int n_max, int n_channels, const uint8_t *seq_in,
                               uint8_t *seq_out, char *out, size_t out_cap, void *scratch, size_t scratch_bytes,
                               uint32_t *h_info4, const uint2 *chunks, int n_chunks, int chunk_passes,
                               uint32_t **totals_dev, hipStream_t s)
{
    const bool by_chunks = n < 0;
    if (by_chunks && (chunk_passes <= 0 || n_chunks % chunk_passes)) return hipErrorInvalidValue;
    if (n == 0 || (by_chunks && (!chunks || n_max <= 0))) return hipErrorInvalidValue;
    if (!by_chunks) { chunks = nullptr; n_chunks = 0; n_max = n; }
    if (scratch_bytes < nmea_scratch_bytes(n_max, n_chunks)) return hipErrorInvalidValue;
    const size_t m = (size_t) n_max;
    const NmeaLayout lay = nmea_layout(scratch, scratch_bytes, m, n_chunks);
    uint64_t *keys = lay.keys, *keys2 = lay.keys2;
    uint32_t *idx = lay.idx, *idx2 = lay.idx2, *chan = lay.chan;
    Tri *tri = lay.tri, *scan = lay.scan;
    uint32_t *chunk_off = lay.chunk_off, *totals = lay.totals;
    void *tmp = lay.tmp;
    const size_t tmp_bytes = lay.tmp_bytes;
    const int grid = (n_max + 255) / 256;
    hipError_t e;
    size_t t = tmp_bytes;
    if (by_chunks) {
        hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(256), 0, s, chunks, n_chunks / chunk_passes, chunk_passes, chunk_off,
                           totals);
    } else {
        if ((e = hipMemsetAsync(totals, 0, 16, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(nmea_keys_kernel, dim3(grid), dim3(256), 0, s, frames, n, keys, idx);
        // channel < 2^24 in any realistic batch; the key's top three bits are never set
        if ((e = rocprim::radix_sort_pairs(tmp, t, keys, keys2, idx, idx2, m, 0, 61, s)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(nmea_meta_kernel, dim3(grid), dim3(256), 0, s, frames, idx2, n, chunks, chunk_off, n_chunks,
                       tri, chan, n_max);
    t = tmp_bytes;
    if ((e = rocprim::inclusive_scan(tmp, t, tri, scan, m, TriOp(), s)) != hipSuccess) return e;
    hipLaunchKernelGGL(nmea_write_kernel, dim3(grid), dim3(256), NMEA_WRITE_LDS_BYTES, s, frames, idx2, chan, tri, scan, n, n_channels,
                       seq_in, seq_out, out, (unsigned long long) out_cap, totals,
                       by_chunks ? chunk_off + n_chunks : (const uint32_t *) nullptr);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (totals_dev) *totals_dev = totals;
    // h_info4 (pinned host memory when the caller does not wait here): [0] offset of the last frame's
    // text, [1] its length, [2] sentences, [3] frames that named a channel outside the batch
    return h_info4 ? hipMemcpyAsync(h_info4, totals, 16, hipMemcpyDeviceToHost, s) : hipSuccess;
}

hipError_t nmea_format(const gnuais_frame *frames, int n, int n_channels, const uint8_t *seq_in,
                       uint8_t *seq_out, char *out, size_t out_cap, void *scratch, size_t scratch_bytes,
                       uint32_t *h_info /* [0] bytes, [1] sentences, [2] bad channel */, hipStream_t s)
{
    h_info[0] = h_info[1] = h_info[2] = 0;
    if (n <= 0) return hipSuccess;
    uint32_t raw[4] = {0, 0, 0, 0};
    hipError_t e = nmea_format_enqueue(frames, n, n, n_channels, seq_in, seq_out, out, out_cap, scratch, scratch_bytes, raw,
                                       nullptr, 0, 0, nullptr, s);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    h_info[0] = raw[0] + raw[1];
    h_info[1] = raw[2];
    h_info[2] = raw[3];
    return hipSuccess;
}

} // namespace gnuais
