// hdlc_crc.hip -- K2b: HDLC deframer (bit-unstuffing FSM) + CRC-16 for gfx950.
//
// Stands in for protodec_decode() (gnuais src/protodec.c:988-1122), for
// protodec_calculate_crc()/protodec_sdlc_crc() (src/protodec.c:106-167) and for
// protodec_reset() (src/protodec.c:87-100), for a whole batch of channels.
//
// One lane = one channel walking its own recovered bit stream (K2a's output);
// one wave = 64 adjacent channels.  The five-state machine is reproduced
// literally, quirks included: `last` is rewritten
// after every state, `nstartsign++` runs even after a reset in ST_STARTSIGN,
// frames reset at bufferpos >= 449.
//
// CRC: the reference recomputes the CRC over the whole frame at the closing
// flag (464 shift steps).  Here the CRC register runs along with the data bits,
// delayed by the 6 bits of the closing flag that are stored before it is
// recognised (0 + five 1s), so that at ST_STOPSIGN it already covers
// buffer[0 .. bufferpos-6).  The reference checks buffer[0 .. 8*(n/8+2)) with
// n = bufferpos-22: for n % 8 == 0 that is the same span; otherwise the
// register is un-clocked n % 8 steps (the CRC LFSR is invertible).  A good frame
// leaves the X-25 residue: ~crc == 0x0f47 (protodec.c:166), i.e. crc == 0xf0b8.
//
// Frame bits are kept per channel in buf[w][c] (bit k at word k/32, LSB first),
// which is also their carry between calls; payload byte j of a frame is then
// simply byte j of that little-endian bit image (protodec.c:138-143 packs the
// same way), which is what the 64-byte frame record carries.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace gnuais {

enum { ST_SKURR = 1, ST_PREAMBLE = 2, ST_STARTSIGN = 3, ST_DATA = 4, ST_STOPSIGN = 5 };

// ctl[0]: state[2:0] nstartsign[6:3] antallpreamble[10:7] antallenner[13:11]
//         bitstuff[14] last[15] bufferpos[24:16]
// ctl[1]: running CRC register [15:0]
// ctl[2]: the last 32 stored data bits, newest at bit 0
// ctl[3]: partially filled buffer word
// ctl[4]: bits fed since reset

__global__ void hdlc_reset_kernel(uint32_t *__restrict__ ctl, int N)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    ctl[c] = ST_SKURR;                      // protodec.c:89-99
    ctl[(size_t) N + c] = 0xffffu;
    ctl[(size_t) 2 * N + c] = 0;
    ctl[(size_t) 3 * N + c] = 0;
    ctl[(size_t) 4 * N + c] = 0;
}

__global__ __launch_bounds__(64) void hdlc_crc_kernel(
    const uint32_t *__restrict__ bits, const uint32_t *__restrict__ nbits,
    uint32_t *__restrict__ ctl, uint32_t *__restrict__ buf, int32_t *__restrict__ counters,
    uint32_t *__restrict__ frames, uint32_t *__restrict__ frame_count, uint32_t frame_cap,
    int N, int bits_words)
{
    const int cg = blockIdx.x * 64 + threadIdx.x;
    if (cg >= N) return;
    const size_t c = (size_t) cg, n_ = (size_t) N;

    const uint32_t c0 = ctl[c];
    int state = c0 & 7, nstartsign = (c0 >> 3) & 15, antallpreamble = (c0 >> 7) & 15;
    int antallenner = (c0 >> 11) & 7, bitstuff = (c0 >> 14) & 1, last = (c0 >> 15) & 1;
    int bufferpos = (c0 >> 16) & 511;
    uint32_t crc = ctl[n_ + c] & 0xffffu;
    uint32_t recent = ctl[2 * n_ + c];
    uint32_t cur = ctl[3 * n_ + c];
    uint32_t seen = ctl[4 * n_ + c];
    int received = 0, lost = 0, lost2 = 0;

    int total = (int) nbits[c];
    if (total > bits_words * 32) total = bits_words * 32;

#define HDLC_RESET()                                                                    \
    do { state = ST_SKURR; nstartsign = 0; antallpreamble = 0; antallenner = 0;        \
         last = 0; bitstuff = 0; bufferpos = 0; } while (0)

    uint32_t word = 0;
    for (int k = 0; k < total; ++k) {
        if ((k & 31) == 0) word = bits[(size_t) (k >> 5) * n_ + c];
        const int x = (int) (word & 1u);
        word >>= 1;

        switch (state) {
        case ST_DATA:                                   // protodec.c:995-1028
            if (bitstuff) {
                if (x == 1) state = ST_STOPSIGN;
                bitstuff = 0;
            } else {
                if (x == last && x == 1) {
                    if (++antallenner == 4) { bitstuff = 1; antallenner = 0; }
                } else {
                    antallenner = 0;
                }
                // buffer[bufferpos++] = x
                cur |= (uint32_t) x << (bufferpos & 31);
                if ((bufferpos & 31) == 31) {
                    buf[(size_t) (bufferpos >> 5) * n_ + c] = cur;
                    cur = 0;
                }
                recent = (recent << 1) | (uint32_t) x;
                if (bufferpos >= 6) {                   // CRC runs 6 bits behind
                    const uint32_t fb = (crc ^ (recent >> 6)) & 1u;
                    crc = (crc >> 1) ^ (fb ? 0x8408u : 0u);
                }
                ++bufferpos;
                if (bufferpos >= 449) HDLC_RESET();
            }
            break;

        case ST_SKURR:                                  // protodec.c:1030-1043
            if (x != last) { if (antallpreamble < 15) ++antallpreamble; }
            else antallpreamble = 0;
            last = x;
            if (antallpreamble > 14 && x == 0) { state = ST_PREAMBLE; antallpreamble = 0; }
            break;

        case ST_PREAMBLE:                               // protodec.c:1045-1072
            if (x != last && nstartsign == 0) {
                if (antallpreamble < 15) ++antallpreamble;
            } else if (x == 1) {
                if (nstartsign == 0) { nstartsign = 3; last = x; }
                else if (nstartsign == 5) { nstartsign = 6; antallpreamble = 0; state = ST_STARTSIGN; }
                else ++nstartsign;
            } else {
                if (nstartsign == 0) nstartsign = 1;
                else HDLC_RESET();
            }
            break;

        case ST_STARTSIGN:                              // protodec.c:1074-1093
            if (nstartsign >= 7) {
                if (x == 0) {
                    state = ST_DATA; nstartsign = 0; antallenner = 0;
                    bufferpos = 0; cur = 0; crc = 0xffffu; recent = 0;
                } else {
                    HDLC_RESET();
                }
            } else if (x == 0) {
                HDLC_RESET();
            }
            ++nstartsign;                               // also after a reset
            break;

        case ST_STOPSIGN: {                             // protodec.c:1095-1115
            const int n = bufferpos - 6 - 16;
            if (x == 0 && n > 0) {
                // protodec.c:120-167: CRC over n/8 + 2 bytes
                uint32_t r = crc;
                const int back = n & 7;
                for (int t = 0; t < back; ++t) {        // un-clock the trailing n%8 bits
                    const uint32_t fb = r >> 15;
                    const uint32_t bit = (recent >> (6 + t)) & 1u;
                    r = (((r ^ (fb ? 0x8408u : 0u)) << 1) | (fb ^ bit)) & 0xffffu;
                }
                if (r == 0xf0b8u) {
                    ++received;                         // protodec.c:1103
                    const uint32_t idx = atomicAdd(&frame_count[0], 1u);
                    if (idx < frame_cap) {
                        const int nbytes = n >> 3;
                        const int partial = bufferpos >> 5;   // word still held in `cur`
                        uint32_t *rec = frames + (size_t) idx * 16;
                        rec[0] = (uint32_t) cg;
                        rec[1] = seen;
#pragma unroll
                        for (int wq = 0; wq < 14; ++wq) {
                            uint32_t v = 0;
                            if (wq * 4 < nbytes) {
                                v = (wq == partial) ? cur : buf[(size_t) wq * n_ + c];
                                const int keep = nbytes - wq * 4;     // bytes of this word
                                if (keep < 4) v &= (1u << (8 * keep)) - 1u;
                            }
                            if (wq == 13) v = (v & 0xffu) | (1u << 8) | ((uint32_t) n << 16);
                            rec[2 + wq] = v;
                        }
                    } else {
                        frame_count[1] = 1;             // overflow: frame dropped, counted
                    }
                } else {
                    ++lost;                             // protodec.c:1107
                }
            } else {
                ++lost2;                                // protodec.c:1112
            }
            HDLC_RESET();
            break;
        }
        default:
            HDLC_RESET();
            break;
        }
        last = x;                                       // protodec.c:1119
        ++seen;
    }
#undef HDLC_RESET

    ctl[c] = (uint32_t) state | ((uint32_t) nstartsign << 3) | ((uint32_t) antallpreamble << 7) |
             ((uint32_t) antallenner << 11) | ((uint32_t) bitstuff << 14) |
             ((uint32_t) last << 15) | ((uint32_t) bufferpos << 16);
    ctl[n_ + c] = crc;
    ctl[2 * n_ + c] = recent;
    ctl[3 * n_ + c] = cur;
    ctl[4 * n_ + c] = seen;
    if (received) counters[c] += received;
    if (lost) counters[n_ + c] += lost;
    if (lost2) counters[2 * n_ + c] += lost2;
}

hipError_t launch_hdlc_crc(const HdlcLaunch &a, hipStream_t stream)
{
    dim3 grid((a.N + 63) / 64), block(64);
    hipLaunchKernelGGL(hdlc_crc_kernel, grid, block, 0, stream, a.bits, a.nbits, a.ctl, a.buf,
                       a.counters, (uint32_t *) a.frames, a.frame_count, a.frame_cap, a.N,
                       a.bits_words);
    return hipGetLastError();
}

hipError_t launch_hdlc_reset(uint32_t *ctl, int N, hipStream_t stream)
{
    hipLaunchKernelGGL(hdlc_reset_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, ctl, N);
    return hipGetLastError();
}

} // namespace gnuais
