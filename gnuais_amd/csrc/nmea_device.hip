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
//   3. exclusive scans of bytes and of accepted, max-scan of the head positions: every frame knows
//      where its text starts and how many accepted frames of its channel came before it, hence
//      its sequence digit (seq0[channel] + that) mod 10;
//   4. one thread per frame writes its one or two sentences; the last frame of a channel leaves the
//      channel's new digit.
// HBM-bound byte shuffling: 64 B read + ~50 B written per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

__global__ __launch_bounds__(256) void nmea_meta_kernel(const gnuais_frame *__restrict__ frames,
                                                        const uint64_t *__restrict__ keys_sorted,
                                                        const uint32_t *__restrict__ order, int n,
                                                        uint32_t *__restrict__ bytes, uint32_t *__restrict__ acc,
                                                        uint32_t *__restrict__ head)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const gnuais_frame *f = frames + order[j];
    const unsigned first = reinterpret_cast<const uint32_t *>(f)[2] & 0xffu;       // payload[0]
    const int nbits = (int) (reinterpret_cast<const uint32_t *>(f)[15] >> 16);
    const unsigned type = nbits >= 6 ? first >> 2 : (nbits > 0 ? (first >> 2) & ~((1u << (6 - nbits)) - 1u) : 0u);
    const bool ok = type >= 1 && type <= MAX_TYPE;
    const Geo g = geometry(nbits);
    bytes[j] = ok ? (uint32_t) (g.nchars + 21 * g.parts) : 0u;
    acc[j] = ok ? 1u : 0u;
    head[j] = (j > 0 && (keys_sorted[j] >> KEY_CH_SHIFT) != (keys_sorted[j - 1] >> KEY_CH_SHIFT)) ? (uint32_t) j : 0u;
}

__device__ __forceinline__ char armor(unsigned v) { return (char) (v < 40 ? v + 48 : v + 56); }   // protodec.c:826-830
__device__ __forceinline__ char hexdigit(unsigned v) { return (char) (v < 10 ? '0' + v : 'A' + v - 10); }

__global__ __launch_bounds__(256) void nmea_write_kernel(
    const gnuais_frame *__restrict__ frames, const uint64_t *__restrict__ keys_sorted,
    const uint32_t *__restrict__ order, const uint32_t *__restrict__ off, const uint32_t *__restrict__ accpre,
    const uint32_t *__restrict__ headpos, const uint32_t *__restrict__ bytes, int n, int n_channels,
    const uint8_t *__restrict__ seq_in, uint8_t *__restrict__ seq_out, char *__restrict__ out,
    unsigned long long out_cap, uint32_t *__restrict__ totals /* [0] sentences, [1] bad channel */)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint32_t ch = (uint32_t) (keys_sorted[j] >> KEY_CH_SHIFT);
    if (ch >= (uint32_t) n_channels) {
        atomicOr(&totals[1], 1u);
        return;
    }
    const bool ok = bytes[j] != 0;
    const uint32_t before = accpre[j] - accpre[headpos[j]];          // accepted frames of this channel before j
    const bool last = (j + 1 == n) || (uint32_t) (keys_sorted[j + 1] >> KEY_CH_SHIFT) != ch;
    if (last) seq_out[ch] = (uint8_t) ((seq_in[ch] + before + (ok ? 1u : 0u)) % 10u);
    if (!ok) return;
    if ((unsigned long long) off[j] + bytes[j] > out_cap) return;    // the host reports the overflow
    const FrameView f = load_frame(frames, order[j]);
    const Geo g = geometry(f.nbits());
    const char seq = (char) ('0' + (seq_in[ch] + before) % 10u);
    char *o = out + off[j];
    int done = 0;
    for (int part = 1; part <= g.parts; ++part) {
        char s[84];
        int k = 0;
        s[k++] = '!'; s[k++] = 'A'; s[k++] = 'I'; s[k++] = 'V'; s[k++] = 'D'; s[k++] = 'M'; s[k++] = ',';
        s[k++] = (char) ('0' + g.parts); s[k++] = ',';
        s[k++] = (char) ('0' + part); s[k++] = ',';
        if (g.parts > 1) {                                          // protodec.c:847-849: no channel letter
            s[k++] = seq; s[k++] = ','; s[k++] = ',';
        } else {                                                    // :857-859: always 'A'
            s[k++] = ','; s[k++] = 'A'; s[k++] = ',';
        }
        for (int i = 0; i < CHARS_PER_SENTENCE && done < g.nchars; ++i, ++done) s[k++] = armor(f.six(done));
        s[k++] = ',';
        s[k++] = (char) ((g.parts > 1 && part == g.parts) ? '0' + g.fill : '0');
        unsigned x = 0;
        for (int i = 1; i < k; ++i) x ^= (unsigned char) s[i];      // :864-869
        s[k++] = '*'; s[k++] = hexdigit(x >> 4); s[k++] = hexdigit(x & 15u);
        s[k++] = '\r'; s[k++] = '\n';
        for (int i = 0; i < k; ++i) o[i] = s[i];
        o += k;
    }
    atomicAdd(&totals[0], (uint32_t) g.parts);
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

size_t nmea_scratch_bytes(int n)
{
    const size_t m = (size_t) (n > 0 ? n : 1);
    size_t sort_tmp = 0, scan_tmp = 0, scan_tmp2 = 0;
    (void) rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint64_t *) nullptr, (uint64_t *) nullptr,
                                     (uint32_t *) nullptr, (uint32_t *) nullptr, m, 0, 64, (hipStream_t) 0);
    (void) rocprim::exclusive_scan(nullptr, scan_tmp, (uint32_t *) nullptr, (uint32_t *) nullptr, 0u, m,
                                   rocprim::plus<uint32_t>(), (hipStream_t) 0);
    (void) rocprim::inclusive_scan(nullptr, scan_tmp2, (uint32_t *) nullptr, (uint32_t *) nullptr, m, MaxOp(),
                                   (hipStream_t) 0);
    size_t tmp = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
    tmp = tmp > scan_tmp2 ? tmp : scan_tmp2;
    // keys x2, idx x2, bytes, off, acc, accpre, head, headpos, totals, rocPRIM temp (256-byte slots)
    return 2 * 8 * m + 8 * 4 * m + 256 * 12 + tmp + 64;
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

// everything of nmea_format() that runs on the device, queued on `s` without waiting for it (n > 0)
hipError_t nmea_format_enqueue(const gnuais_frame *frames, int n, int n_channels, const uint8_t *seq_in,
                               uint8_t *seq_out, char *out, size_t out_cap, void *scratch, size_t scratch_bytes,
                               uint32_t *h_info4, hipStream_t s)
{
    if (n <= 0) return hipErrorInvalidValue;
    if (scratch_bytes < nmea_scratch_bytes(n)) return hipErrorInvalidValue;
    const size_t m = (size_t) n;
    char *p = static_cast<char *>(scratch);
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) / 256 * 256; return (void *) q; };
    uint64_t *keys = (uint64_t *) take(8 * m), *keys2 = (uint64_t *) take(8 * m);
    uint32_t *idx = (uint32_t *) take(4 * m), *idx2 = (uint32_t *) take(4 * m);
    uint32_t *bytes = (uint32_t *) take(4 * m), *off = (uint32_t *) take(4 * m);
    uint32_t *acc = (uint32_t *) take(4 * m), *accpre = (uint32_t *) take(4 * m);
    uint32_t *head = (uint32_t *) take(4 * m), *headpos = (uint32_t *) take(4 * m);
    uint32_t *totals = (uint32_t *) take(16);
    void *tmp = p;
    size_t tmp_bytes = scratch_bytes - (size_t) (p - static_cast<char *>(scratch));
    const int grid = (n + 255) / 256;
    hipError_t e;
    if ((e = hipMemsetAsync(totals, 0, 16, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(nmea_keys_kernel, dim3(grid), dim3(256), 0, s, frames, n, keys, idx);
    // channel < 2^24 in any realistic batch; the key's top three bits are never set
    size_t t = tmp_bytes;
    if ((e = rocprim::radix_sort_pairs(tmp, t, keys, keys2, idx, idx2, m, 0, 61, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(nmea_meta_kernel, dim3(grid), dim3(256), 0, s, frames, keys2, idx2, n, bytes, acc, head);
    t = tmp_bytes;
    if ((e = rocprim::exclusive_scan(tmp, t, bytes, off, 0u, m, rocprim::plus<uint32_t>(), s)) != hipSuccess) return e;
    t = tmp_bytes;
    if ((e = rocprim::exclusive_scan(tmp, t, acc, accpre, 0u, m, rocprim::plus<uint32_t>(), s)) != hipSuccess) return e;
    t = tmp_bytes;
    if ((e = rocprim::inclusive_scan(tmp, t, head, headpos, m, MaxOp(), s)) != hipSuccess) return e;
    hipLaunchKernelGGL(nmea_write_kernel, dim3(grid), dim3(256), 0, s, frames, keys2, idx2, off, accpre, headpos,
                       bytes, n, n_channels, seq_in, seq_out, out, (unsigned long long) out_cap, totals);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // h_info4 (pinned host memory when the caller does not wait here): [0] offset of the last frame's
    // text, [1] its length, [2] sentences, [3] frames that named a channel outside the batch
    if ((e = hipMemcpyAsync(&h_info4[0], off + (m - 1), 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(&h_info4[1], bytes + (m - 1), 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    return hipMemcpyAsync(&h_info4[2], totals, 8, hipMemcpyDeviceToHost, s);
}

hipError_t nmea_format(const gnuais_frame *frames, int n, int n_channels, const uint8_t *seq_in,
                       uint8_t *seq_out, char *out, size_t out_cap, void *scratch, size_t scratch_bytes,
                       uint32_t *h_info /* [0] bytes, [1] sentences, [2] bad channel */, hipStream_t s)
{
    h_info[0] = h_info[1] = h_info[2] = 0;
    if (n <= 0) return hipSuccess;
    uint32_t raw[4] = {0, 0, 0, 0};
    hipError_t e = nmea_format_enqueue(frames, n, n_channels, seq_in, seq_out, out, out_cap, scratch, scratch_bytes, raw, s);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    h_info[0] = raw[0] + raw[1];
    h_info[1] = raw[2];
    h_info[2] = raw[3];
    return hipSuccess;
}

} // namespace gnuais
