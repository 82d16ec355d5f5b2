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
