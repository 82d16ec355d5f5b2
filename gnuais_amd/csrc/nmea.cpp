// nmea.cpp -- row f1 (first part): the NMEA 0183 sentences gnuais emits for a CRC-valid frame.
//
// Host-side restatement of protodec_getdata()'s sentence path, gnuais src/protodec.c:896-926,
// and protodec_generate_nmea(), src/protodec.c:780-894, working from the 64-byte frame records
// the device chain delivers (gnuais_batch_drain_frames) instead of from d->rbuffer.  This is
// the reference's own post-stage (it runs once per valid frame, after the hot path) and, like
// there, it runs on the host.  Output is byte-identical to what the reference hands to
// serial_write() (src/protodec.c:883-885), including its quirks:
//   * frames whose first 6 bits (the AIS type) are 0 or > 24 produce nothing and do not
//     advance the sequence digit (:898-900);
//   * the payload is padded with 0 bits to a multiple of 6 (:909-915), 61 characters per
//     sentence (:793), 6-bit value v -> v + 48 if v < 40 else v + 56 (:810-815);
//   * single-sentence messages carry channel 'A' and fill digit '0' whatever the padding was;
//     multi-sentence messages carry the rolling sequence digit, an EMPTY channel field, and the
//     fill digit only on the last part (:842-860);
//   * the sequence digit advances 0,1,..,9,0 after EVERY accepted frame, single- or
//     multi-sentence (:922-926);
//   * checksum = XOR of everything between '!' and '*', upper-case hex, two digits (:864-881).
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "gnuais_hip.h"

namespace {

// `count` bits of the payload starting at bit `pos`, MSB first (protodec_henten, protodec.c:205-214);
// bits at or beyond `nbits` read as 0
inline unsigned take_bits(const gnuais_frame &f, int pos, int count, int nbits)
{
    unsigned v = 0;
    for (int i = 0; i < count; ++i) {
        const int b = pos + i;
        const unsigned bit = (b < nbits && b < 8 * (int) sizeof f.payload) ? (f.payload[b >> 3] >> (7 - (b & 7))) & 1u : 0u;
        v = (v << 1) | bit;
    }
    return v;
}

constexpr int CHARS_PER_SENTENCE = 61;      // protodec.c:793
constexpr int MAX_TYPE = 24;                // cfg.h:48 MAX_AIS_PACKET_TYPE
const char HEX[] = "0123456789ABCDEF";

} // namespace

extern "C" int gnuais_nmea_from_frames(const gnuais_frame *frames, int n_frames, uint8_t *seqnr,
                                       int n_channels, char *out, size_t out_cap, size_t *out_len,
                                       int *n_sentences)
{
    if (n_frames < 0 || (n_frames > 0 && !frames) || !seqnr || n_channels <= 0 || !out_len)
        return GNUAIS_E_ARG;
    size_t len = 0;
    int total = 0;
    bool fits = true;
    for (int k = 0; k < n_frames; ++k) {
        const gnuais_frame &f = frames[k];
        if (f.channel >= (uint32_t) n_channels) return GNUAIS_E_ARG;
        const int nbits = f.nbits;
        if (nbits > 8 * (int) sizeof f.payload) return GNUAIS_E_ARG;
        const unsigned type = take_bits(f, 0, 6, nbits);
        if (type < 1 || type > MAX_TYPE) continue;
        const int fill = (6 - nbits % 6) % 6;
        const int nchars = (nbits + fill) / 6;
        const int parts = nchars <= CHARS_PER_SENTENCE ? 1 : (nchars + CHARS_PER_SENTENCE - 1) / CHARS_PER_SENTENCE;
        uint8_t &seq = seqnr[f.channel];
        int done = 0;
        for (int part = 1; part <= parts; ++part) {
            char s[96];
            int n = 0;
            s[n++] = '!';
            memcpy(s + n, "AIVDM,", 6); n += 6;
            s[n++] = (char) ('0' + parts);
            s[n++] = ',';
            s[n++] = (char) ('0' + part);
            s[n++] = ',';
            if (parts > 1) {
                s[n++] = (char) ('0' + seq);
                s[n++] = ',';
                s[n++] = ',';
            } else {
                s[n++] = ',';
                s[n++] = 'A';
                s[n++] = ',';
            }
            for (int i = 0; i < CHARS_PER_SENTENCE && done < nchars; ++i, ++done) {
                const unsigned v = take_bits(f, 6 * done, 6, nbits);
                s[n++] = (char) (v < 40 ? v + 48 : v + 56);
            }
            s[n++] = ',';
            s[n++] = (char) ((parts > 1 && part == parts) ? '0' + fill : '0');
            unsigned char x = 0;
            for (int i = 1; i < n; ++i) x ^= (unsigned char) s[i];
            s[n++] = '*';
            s[n++] = HEX[x >> 4];
            s[n++] = HEX[x & 15];
            s[n++] = '\r';
            s[n++] = '\n';
            if (out && len + (size_t) n <= out_cap) memcpy(out + len, s, (size_t) n);
            else fits = false;
            len += (size_t) n;
            ++total;
        }
        seq = (uint8_t) (seq >= 9 ? 0 : seq + 1);
    }
    *out_len = len;
    if (n_sentences) *n_sentences = total;
    return (out && !fits) ? GNUAIS_E_OVERFLOW : GNUAIS_OK;
}
