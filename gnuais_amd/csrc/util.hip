// util.hip -- small helper kernels: benchmark input tiling and a device CRC-16
// known-answer entry point.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gnuais {

// out[l][c] = base[c % n_base][(l + (c * 7919) % len) % len]   (SURVEY.md 8d:
// 256 base streams replayed with a per-channel circular rotation).  One thread
// writes 8 adjacent channels of one sample time (16 bytes).
__global__ void tile_channels_kernel(const int16_t *__restrict__ base, int n_base, int len,
                                     int16_t *__restrict__ out, int n_channels)
{
    const int l = blockIdx.x;
    const int c = (blockIdx.y * blockDim.x + threadIdx.x) * 8;
    if (c >= n_channels) return;
    int16_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = c + i;
        int16_t s = 0;
        if (ch < n_channels) {
            const int rot = (int) (((long long) ch * 7919) % len);
            int src = l + rot;
            if (src >= len) src -= len;
            s = base[(size_t) (ch % n_base) * (size_t) len + src];
        }
        v[i] = s;
    }
    int16_t *dst = out + (size_t) l * (size_t) n_channels + c;
    if (c + 8 <= n_channels && (((size_t) l * (size_t) n_channels + c) & 7) == 0) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(v);
    } else {
        for (int i = 0; i < 8 && c + i < n_channels; ++i) dst[i] = v[i];
    }
}

// protodec_sdlc_crc, gnuais src/protodec.c:106-118: one thread per message
__global__ void crc16_kernel(const uint8_t *__restrict__ data, int stride,
                             const int32_t *__restrict__ len, int n, uint16_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t crc = 0xffffu;
    const uint8_t *p = data + (size_t) i * (size_t) stride;
    for (int k = 0; k < len[i]; ++k) {
        uint32_t v = p[k];
        for (int b = 0; b < 8; ++b) {
            const uint32_t fb = (crc ^ (v >> b)) & 1u;
            crc = (crc >> 1) ^ (fb ? 0x8408u : 0u);
        }
    }
    out[i] = (uint16_t) (~crc & 0xffffu);
}

// protodec_calculate_crc's arithmetic (gnuais src/protodec.c:120-167) for ONE frame, one wave: lane j forms byte j
// from the eight one-bit-per-byte cells bits[8j .. 8j+7], the first cell in the byte's LEAST significant position
// (protodec.c:138-143; a cell is shifted as it is, like the reference's `buffer[..] << i` narrowed to a byte), lane 0
// runs the CRC-16/X-25 register over the n_bytes bytes (protodec.c:106-118), and lane j writes cells 8j .. 8j+7 of
// msb[] with the byte's MOST significant bit first (protodec.c:150-162), as far as n_out reaches.
__global__ __launch_bounds__(64) void crc16_bits_kernel(const uint8_t *__restrict__ bits, int n_bytes,
                                                        uint16_t *__restrict__ crc_out, uint8_t *__restrict__ msb,
                                                        int n_out)
{
    __shared__ uint8_t bytes[64];
    const int j = threadIdx.x;
    uint32_t v = 0;
    if (j < n_bytes)
        for (int i = 0; i < 8; ++i) v |= ((uint32_t) bits[8 * j + i] << i) & 0xffu;
    bytes[j] = (uint8_t) v;
    __syncthreads();
    if (j == 0) {
        uint32_t crc = 0xffffu;
        for (int k = 0; k < n_bytes; ++k) {
            crc ^= bytes[k];
            for (int b = 0; b < 8; ++b) crc = (crc >> 1) ^ (0x8408u & (0u - (crc & 1u)));
        }
        *crc_out = (uint16_t) (~crc & 0xffffu);
    }
    if (msb && j < n_bytes)
        for (int i = 0; i < 8; ++i)
            if (8 * j + i < n_out) msb[8 * j + i] = (uint8_t) ((v >> (7 - i)) & 1u);
}

hipError_t launch_crc16_bits(const uint8_t *bits, int n_bytes, uint16_t *crc, uint8_t *msb, int n_out,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(crc16_bits_kernel, dim3(1), dim3(64), 0, stream, bits, n_bytes, crc, msb, n_out);
    return hipGetLastError();
}

hipError_t launch_tile_channels(const int16_t *base, int n_base, int len, int16_t *out,
                                int n_channels, hipStream_t stream)
{
    const int threads = 128;
    dim3 grid(len, ((n_channels + 7) / 8 + threads - 1) / threads), block(threads);
    hipLaunchKernelGGL(tile_channels_kernel, grid, block, 0, stream, base, n_base, len, out,
                       n_channels);
    return hipGetLastError();
}

hipError_t launch_crc16(const uint8_t *data, int stride, const int32_t *len, int n,
                        uint16_t *crc, hipStream_t stream)
{
    hipLaunchKernelGGL(crc16_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, data, stride,
                       len, n, crc);
    return hipGetLastError();
}

} // namespace gnuais
