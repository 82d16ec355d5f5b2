// nmea.cpp -- row f1: what gnuais does with a CRC-valid frame after the hot path.
//
// Host-side restatement of protodec_getdata(), gnuais src/protodec.c:896-986, working from the
// 64-byte frame records the device chain delivers (gnuais_batch_drain_frames) instead of from
// d->rbuffer.  This is the reference's own post-stage (once per valid frame) and, like there,
// it runs on the host.  Two outputs, both byte-identical to the reference's:
//
//  * the NMEA 0183 sentences it hands to serial_write() / ipc_write() / myout_nmea()
//    (protodec_generate_nmea, src/protodec.c:780-894), with its quirks:
//      - frames whose first 6 bits (the AIS type) are 0 or > 24 produce nothing and do not
//        advance the sequence digit (:898-900);
//      - the payload is padded with 0 bits to a multiple of 6 (:909-915), 61 characters per
//        sentence (:793), 6-bit value v -> v + 48 if v < 40 else v + 56 (:810-815);
//      - single-sentence messages carry channel 'A' and fill digit '0' whatever the padding
//        was; multi-sentence messages carry the rolling sequence digit, an EMPTY channel field,
//        and the fill digit only on the last part (:842-860);
//      - the sequence digit advances 0,1,..,9,0 after EVERY accepted frame (:922-926);
//      - checksum = XOR of everything between '!' and '*', upper-case hex, two digits (:864-881);
//
//  * the line it prints on stdout (:934-985): "ch <id> type <t> mmsi <9 digits>:", the fields the
//    per-type decoders print (src/protodec.c:357-776: types 1-3, 4, 5, 6, 7/13, 8, 18, 19, 20, 24
//    and the two binary applications 1/11 and 1/40), and " (!<last sentence>)".  The field
//    positions are the reference's, including the ones that differ from ITU-R M.1371 (a 2-bit
//    navigation status at bit 38, the base-station date starting at bit 40, the weather report's
//    wind speed overlapping its longitude); floats are formed and rounded exactly as there
//    ((float) value / double constant, printed with %f).
#include <stdint.h>
#include <stddef.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "gnuais_hip.h"

namespace {

constexpr int CHARS_PER_SENTENCE = 61;      // protodec.c:793
// bufferlen of the longest frame the deframer delivers: bufferpos 448 - 22 (protodec.c:1023,1096).
// The record holds its 53 whole bytes; the bits past them are 0 here as in d->rbuffer (:150-162).
constexpr int MAX_NBITS = 426;
constexpr int MAX_TYPE = 24;                // cfg.h:48 MAX_AIS_PACKET_TYPE
const char HEX[] = "0123456789ABCDEF";

// the frame's payload as the reference's d->rbuffer: bit k, 0 beyond the frame
struct Bits {
    // the frame's nbits / 8 whole bytes, zero beyond: protodec_calculate_crc() clears d->rbuffer and
    // fills in whole bytes only (protodec.c:133,150-162), so the nbits % 8 bits of a ragged frame are 0
    // whatever the record holds there; padded so that any field can be fetched with one 8-byte load
    uint8_t w[72];
    int nbits;
    Bits(const gnuais_frame &f, int n) : nbits(n)
    {
        memset(w, 0, sizeof w);
        const int whole = std::min(n, 8 * (int) sizeof f.payload) >> 3;
        memcpy(w, f.payload, (size_t) whole);
    }
    // `count` (<= 32) bits from `pos`, MSB first (protodec_henten, protodec.c:205-214)
    unsigned long get(int pos, int count) const
    {
        if (pos >= 8 * 64) return 0;
        uint64_t v;
        memcpy(&v, w + (pos >> 3), 8);
        v = __builtin_bswap64(v);
        return (unsigned long) ((v << (pos & 7)) >> (64 - count));
    }
    // two's complement field of `count` bits (the reference ORs the upper bits in by hand)
    int sget(int pos, int count) const
    {
        unsigned long v = get(pos, count);
        if ((v >> (count - 1)) & 1) v |= ~0ul << count;
        return (int) v;
    }
    // `n` six-bit characters from `pos` (protodec_decode_sixbit_ascii :189-203), trailing
    // blanks dropped (remove_trailing_spaces :173-184)
    std::string text(int pos, int n) const
    {
        std::string s;
        for (int k = 0; k < n; ++k) {
            const int v = (int) get(pos + 6 * k, 6);
            s.push_back(v >= 1 && v <= 31 ? (char) (v + 64) : (v >= 32 ? (char) v : ' '));
        }
        while (!s.empty() && s.back() == ' ') s.pop_back();
        return s;
    }
};

struct Out {
    std::string s;
    __attribute__((format(printf, 2, 3))) void add(const char *fmt, ...)
    {
        char tmp[512];
        va_list ap;
        va_start(ap, fmt);
        const int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
        va_end(ap);
        if (n > 0) s.append(tmp, (size_t) (n < (int) sizeof tmp ? n : (int) sizeof tmp - 1));
    }
};

// International function identifiers of DAC 1 (appid_ifm, protodec.c:220-273)
const char *ifm_name(int fi)
{
    static const struct { int fi; const char *name; } names[] = {
        {0, "text-telegram"}, {1, "application-ack"}, {2, "iai-fi-capab-interrogation"},
        {3, "iai-capabi-interrogation"}, {4, "capability-reply"}, {11, "tide-weather"},
        {16, "vts-targets"}, {17, "ship-waypoints"}, {18, "advice-of-waypoints"},
        {19, "extended-ship-data"}, {20, "berthing-data"}, {21, "weather-obs-report"},
        {22, "area-notice-bc"}, {23, "area-notice-addr"}, {24, "extended-ship-static"},
        {25, "dangerous-cargo-info"}, {26, "environmental"}, {27, "route-info-bc"},
        {28, "route-info-addr"}, {29, "text-description-bc"}, {30, "text-description-addr"},
        {40, "persons-on-board"}};
    for (const auto &e : names)
        if (e.fi == fi) return e.name;
    return "unknown";
}

// binary applications of DAC 1 (protodec_msg_bin :338-350)
void binary_payload(const Bits &b, int fi, int at, Out &o)
{
    if (fi == 40) {                              // protodec_msg_40 :279-285
        o.add(" persons-on-board %d", (int) b.get(at, 13));
    } else if (fi == 11) {                       // protodec_msg_11 :287-336, its offsets as they are
        const int lat = (int) b.get(at, 24), lon = (int) b.get(at + 24, 25);
        const int wind = (int) b.get(at + 40, 7), gust = (int) b.get(at + 47, 7);
        const int wdir = (int) b.get(at + 54, 9), gdir = (int) b.get(at + 63, 9);
        const int temp = (int) b.get(at + 72, 11), hum = (int) b.get(at + 83, 7);
        const int dew = (int) b.get(at + 90, 10), pres = (int) b.get(at + 100, 9) + 800;
        const int tend = (int) b.get(at + 109, 2), vis = (int) b.get(at + 111, 8);
        const int level = (int) b.get(at + 119, 9), wave = (int) b.get(at + 124, 8);
        const int wtemp = (int) b.get(at + 128, 10);
        o.add(" lat %.6f lon %.6f wind_speed %dkt wind_gust %dkt wind_dir %d wind_gust_dir %d air_temp %.1fC"
              " rel_humid %d%% dew_point %.1fC pressure %d pressure_tend %d visib %.1fNM water_level %.1fm"
              " wave_height %.1fm water_temp %.1fC",
              (float) lat / 60000.0, (float) lon / 60000.0, wind, gust, wdir, gdir, (float) temp / 10.0 - 60.0,
              hum, (float) dew / 10.0 - 20.0, pres, tend, (float) vis / 10.0, (float) level / 10.0 - 10.0,
              (float) wave / 10.0, (float) wtemp / 10.0 - 10.0);
    }
}

void position_fields(int lat, int lon, unsigned course, unsigned sog, int rot, int navstat, unsigned heading, Out &o)
{
    o.add(" lat %.6f lon %.6f course %.0f speed %.1f rateofturn %d navstat %d heading %d",
          (float) lat / 600000.0, (float) lon / 600000.0, (float) (unsigned short) course / 10.0,
          (float) (unsigned short) sog / 10.0, rot, navstat, (int) (unsigned short) heading);
}

// the fields the reference prints for one frame (the switch at protodec.c:936-982)
void describe(const Bits &b, unsigned type, int padded_len, Out &o)
{
    switch (type) {
    case 1: case 2: case 3:                      // protodec_pos :357-402
        position_fields(b.sget(89, 27), b.sget(61, 28), (unsigned) b.get(116, 12), (unsigned) b.get(50, 10),
                        (int) (signed char) b.get(40, 8), (int) (signed char) b.get(38, 2),
                        (unsigned) b.get(128, 9), o);
        break;
    case 4: {                                    // protodec_4 :404-441
        const float lon = (float) ((float) b.sget(79, 28) / 10000.0 / 60.0);
        const float lat = (float) ((float) b.sget(107, 27) / 10000.0 / 60.0);
        o.add(" date %ld-%ld-%ld time %02ld:%02ld:%02ld lat %.6f lon %.6f", (long) b.get(40, 12),
              (long) b.get(52, 4), (long) b.get(56, 5), (long) b.get(61, 5), (long) b.get(66, 6),
              (long) b.get(72, 6), lat, lon);
        break;
    }
    case 5: {                                    // protodec_5 :443-519
        const unsigned A = (unsigned) b.get(240, 9), B = (unsigned) b.get(249, 9);
        const unsigned char C = (unsigned char) b.get(258, 6), D = (unsigned char) b.get(264, 6);
        const unsigned char draught = (unsigned char) b.get(294, 8);
        o.add(" name \"%s\" destination \"%s\" type %d length %d width %d draught %.1f", b.text(112, 20).c_str(),
              b.text(302, 20).c_str(), (int) b.get(232, 8), (int) (A + B), C + D, (float) draught / 10.0);
        break;
    }
    case 6: {                                    // protodec_6 :525-542
        const int dac = (int) b.get(72, 10), fi = (int) b.get(82, 6);
        o.add(" dst_mmsi %09ld seq %d retransmitted %d appid %d app_dac %d app_fi %d", (long) b.get(40, 30),
              (int) b.get(38, 2), (int) b.get(70, 1), (int) b.get(72, 16), dac, fi);
        if (dac == 1) {
            o.add("(%s)", ifm_name(fi));
            binary_payload(b, fi, 88, o);
        }
        break;
    }
    case 7: case 13: {                           // protodec_7_13 :549-567
        int pos = 40;
        o.add(" buflen %d pos+32 %d", padded_len, pos + 32);
        for (int i = 0; i < 4 && pos + 32 <= padded_len; pos += 32, ++i)
            o.add(" ack %d (to %09ld seq %d)", i + 1, (long) b.get(pos, 30), (int) b.get(pos + 30, 2));
        break;
    }
    case 8: {                                    // protodec_8 :573-584
        const int dac = (int) b.get(40, 10), fi = (int) b.get(50, 6);
        o.add(" appid %d app_dac %d app_fi %d", (int) b.get(40, 16), dac, fi);
        if (dac == 1) {
            o.add("(%s)", ifm_name(fi));
            binary_payload(b, fi, 56, o);
        }
        break;
    }
    case 18:                                     // protodec_18 :586-635 (no turn rate / status in class B)
        position_fields(b.sget(85, 27), b.sget(57, 28), (unsigned) b.get(112, 12), (unsigned) b.get(46, 10), 0, 15,
                        (unsigned) b.get(124, 9), o);
        break;
    case 19: {                                   // protodec_19 :637-684
        const unsigned A = (unsigned) b.get(271, 9), B = (unsigned) b.get(280, 9);
        const unsigned char C = (unsigned char) b.get(289, 6), D = (unsigned char) b.get(295, 6);
        o.add(" name \"%s\" type %d length %d  width %d", b.text(143, 20).c_str(), (int) b.get(263, 8),
              (int) (A + B), C + D);
        break;
    }
    case 20: {                                   // protodec_20 :686-705
        int pos = 40;
        for (int i = 0; i < 4 && pos + 30 < padded_len; pos += 30, ++i)
            o.add(" reserve %d (ofs %d slots %d timeout %d incr %d)", i + 1, (int) b.get(pos, 12),
                  (int) b.get(pos + 12, 4), (int) b.get(pos + 16, 3), (int) b.get(pos + 19, 11));
        break;
    }
    case 24: {                                   // protodec_24 :707-776
        const int part = (int) b.get(38, 2);
        if (part == 0) o.add(" name \"%s\"", b.text(40, 20).c_str());
        if (part == 1) {
            const unsigned A = (unsigned) b.get(132, 9), B = (unsigned) b.get(141, 9);
            const unsigned char C = (unsigned char) b.get(150, 6), D = (unsigned char) b.get(156, 6);
            o.add(" callsign \"%s\" type %d length %d width %d", b.text(90, 6).c_str(), (int) b.get(40, 8),
                  (int) (A + B), C + D);
        }
        break;
    }
    default:
        break;
    }
}

struct Sink {
    char *p;
    size_t cap, len;
    bool fits;
    void put(const char *s, size_t n)
    {
        if (p && len + n <= cap) memcpy(p + len, s, n);
        else if (p) fits = false;
        len += n;
    }
};

} // namespace

// frames [k0, k1) -> sentences and/or text appended to the two strings; returns an error code
static int format_range(const gnuais_frame *frames, int k0, int k1, uint8_t *seqnr, const char *chanid,
                        int n_channels, bool want_nmea, bool want_text, std::string &nm, std::string &tx,
                        int &sentences, int &lines)
{
    for (int k = k0; k < k1; ++k) {
        const gnuais_frame &f = frames[k];
        if (f.channel >= (uint32_t) n_channels) return GNUAIS_E_ARG;
        const int nbits = f.nbits;
        if (nbits > MAX_NBITS) return GNUAIS_E_ARG;
        const Bits b(f, nbits);
        const unsigned type = (unsigned) b.get(0, 6);
        if (type < 1 || type > MAX_TYPE) continue;
        const int fill = (6 - nbits % 6) % 6;
        const int nchars = (nbits + fill) / 6;
        const int parts = nchars <= CHARS_PER_SENTENCE ? 1 : (nchars + CHARS_PER_SENTENCE - 1) / CHARS_PER_SENTENCE;
        uint8_t &seq = seqnr[f.channel];
        char s[96];
        int n = 0, done = 0;
        for (int part = 1; part <= parts; ++part) {
            n = 0;
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
                const unsigned v = (unsigned) b.get(6 * done, 6);
                s[n++] = (char) (v < 40 ? v + 48 : v + 56);
            }
            s[n++] = ',';
            s[n++] = (char) ((parts > 1 && part == parts) ? '0' + fill : '0');
            unsigned char x = 0;
            for (int i = 1; i < n; ++i) x ^= (unsigned char) s[i];
            s[n++] = '*';
            s[n++] = HEX[x >> 4];
            s[n++] = HEX[x & 15];
            if (want_nmea) {
                nm.append(s, (size_t) n);
                nm.append("\r\n", 2);
            }
            ++sentences;
        }
        seq = (uint8_t) (seq >= 9 ? 0 : seq + 1);
        if (want_text) {                         // protodec.c:934, 984: only the last sentence is shown
            Out o;
            o.add("ch %c type %d mmsi %09ld:", chanid ? chanid[f.channel] : (char) ('A' + f.channel % 26),
                  (int) type, (long) b.get(8, 30));
            describe(b, type, nbits + fill, o);
            tx.append(o.s);
            tx.append(" (");
            tx.append(s, (size_t) n);
            tx.append(")\n");
            ++lines;
        }
    }
    return GNUAIS_OK;
}

extern "C" int gnuais_messages_from_frames(const gnuais_frame *frames, int n_frames, uint8_t *seqnr,
                                           const char *chanid, int n_channels, char *nmea, size_t nmea_cap,
                                           size_t *nmea_len, int *n_sentences, char *text, size_t text_cap,
                                           size_t *text_len, int *n_lines)
{
    if (n_frames < 0 || (n_frames > 0 && !frames) || !seqnr || n_channels <= 0) return GNUAIS_E_ARG;
    const bool want_nmea = nmea_len != nullptr, want_text = text_len != nullptr;
    // The only state is the per-channel sequence digit, so frames that arrive grouped by channel
    // (gnuais_batch_drain_frames delivers them that way) split into independent ranges at channel
    // boundaries: one host thread per range, outputs concatenated in order.  Anything else, and
    // small batches, take the serial path.  (The reference's post-stage is one thread at <= 75
    // msgs/s; the device chain delivers 3e8 msgs/s.)
    int n_thr = 1;
    if (n_frames >= 16384) {
        bool grouped = true;
        for (int k = 1; k < n_frames && grouped; ++k) grouped = frames[k].channel >= frames[k - 1].channel;
        const unsigned hw = std::thread::hardware_concurrency();
        if (grouped) n_thr = (int) std::min<unsigned>({hw ? hw : 1u, 64u, (unsigned) (n_frames / 8192)});
    }
    std::vector<int> cut(n_thr + 1, n_frames);
    cut[0] = 0;
    for (int t = 1; t < n_thr; ++t) {
        int k = (int) ((long long) n_frames * t / n_thr);
        while (k < n_frames && k > 0 && frames[k].channel == frames[k - 1].channel) ++k;
        cut[t] = k < cut[t - 1] ? cut[t - 1] : k;
    }
    std::vector<std::string> nm(n_thr), tx(n_thr);
    std::vector<int> ns(n_thr, 0), nl(n_thr, 0), rc(n_thr, GNUAIS_OK);
    auto work = [&](int t) {
        rc[t] = format_range(frames, cut[t], cut[t + 1], seqnr, chanid, n_channels, want_nmea, want_text, nm[t],
                             tx[t], ns[t], nl[t]);
    };
    if (n_thr == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    Sink sn{nmea, nmea_cap, 0, true}, st{text, text_cap, 0, true};
    int sentences = 0, lines = 0;
    for (int t = 0; t < n_thr; ++t) {
        if (rc[t] != GNUAIS_OK) return rc[t];
        sn.put(nm[t].data(), nm[t].size());
        st.put(tx[t].data(), tx[t].size());
        sentences += ns[t];
        lines += nl[t];
    }
    if (nmea_len) *nmea_len = sn.len;
    if (n_sentences) *n_sentences = sentences;
    if (text_len) *text_len = st.len;
    if (n_lines) *n_lines = lines;
    return (sn.fits && st.fits) ? GNUAIS_OK : GNUAIS_E_OVERFLOW;
}

// ---- range statistics (range.c:18-53) ---------------------------------------------------------------
// The float/double mix of every expression follows the reference: its doubles are the literals 0.5, 1.0,
// 2.0 and M_PI inside otherwise-float arithmetic, each result rounded to float where the reference stores
// or passes a float.  sinf/cosf/atan2f/sqrtf are the host libm's, as there.
static float to_rad(float deg) { return (float) (deg * (M_PI / 180.0)); }             // lat2rad/lon2rad :8-16

static float km_distance(float lat1, float lon1, float lat2, float lon2)              // :18-30
{
    const float sdlat = sinf((float) ((lat1 - lat2) * 0.5));
    const float sdlon = sinf((float) ((lon1 - lon2) * 0.5));
    const float c1 = cosf(lat1), c2 = cosf(lat2);
    const float a = sdlat * sdlat + c1 * c2 * sdlon * sdlon;
    const float c = (float) (2.0 * atan2f(sqrtf(a), sqrtf((float) (1.0 - a))));
    return (float) ((111.2 * 180.0 / M_PI) * c);
}

extern "C" int gnuais_range_from_frames(const gnuais_frame *frames, int n_frames, int n_channels,
                                        float my_lat_deg, float my_lon_deg, float *best_range_km)
{
    if (n_frames < 0 || (n_frames > 0 && !frames) || n_channels <= 0 || !best_range_km) return GNUAIS_E_ARG;
    // cfg.c:364-368: a station position outside these bounds means "no location", and nothing is tracked
    if (!(my_lat_deg > -90 && my_lat_deg < 90 && my_lon_deg > -180 && my_lon_deg < 180)) return GNUAIS_OK;
    const float mylat = to_rad(my_lat_deg), mylng = to_rad(my_lon_deg);
    for (int k = 0; k < n_frames; ++k) {
        const gnuais_frame &f = frames[k];
        if (f.channel >= (uint32_t) n_channels || f.nbits > MAX_NBITS) return GNUAIS_E_ARG;
        const Bits b(f, (int) f.nbits);
        long latitude, longitude;
        switch ((unsigned) b.get(0, 6)) {
        case 1: case 2: case 3: latitude = b.sget(89, 27); longitude = b.sget(61, 28); break;   // protodec.c:399
        case 4: latitude = b.sget(107, 27); longitude = b.sget(79, 28); break;                  // :441
        case 18: latitude = b.sget(85, 27); longitude = b.sget(57, 28); break;                  // :628
        default: continue;
        }
        const float lat = (float) ((float) latitude / 600000.0), lon = (float) ((float) longitude / 600000.0);
        // update_range :32-45: bad fixes and the 0/0 "no position" are ignored
        if (lat > 89.0 || lat < -89.0 || lon > 180.01 || lon < -180.01) continue;
        if (lat < 0.001 && lat > -0.001 && lon < 0.001 && lon > -0.001) continue;
        const float d = km_distance(mylat, mylng, to_rad(lat), to_rad(lon));
        if (d > best_range_km[f.channel]) best_range_km[f.channel] = d;
    }
    return GNUAIS_OK;
}

// ---- vessel table: the reference's position cache, folded per batch (cache.c:163-384) ------------
namespace {

static_assert(sizeof(gnuais_vessel) == 120, "gnuais_vessel layout");

gnuais_vessel fresh_vessel(int mmsi)                 // cache_get's new entry :175-196
{
    gnuais_vessel v;
    memset(&v, 0, sizeof v);
    v.mmsi = mmsi;
    v.hdg = -1; v.course = -1; v.sog = -1; v.shiptype = -1; v.imo = -1; v.navstat = -1;
    v.A = v.B = v.C = v.D = -1;
    v.persons_on_board = -1;
    return v;
}

void put(char *dst, size_t cap, const std::string &s)
{
    memset(dst, 0, cap);
    memcpy(dst, s.data(), std::min(s.size(), cap - 1));
}

void set_position(gnuais_vessel &v, int navstat, long latitude, long longitude, int hdg, unsigned course,
                  unsigned sog)                      // cache_position :204-229
{
    v.set |= GNUAIS_V_POSITION;
    v.lat = (float) ((float) latitude / 600000.0);
    v.lon = (float) ((float) longitude / 600000.0);
    v.hdg = hdg;
    v.course = (float) ((float) (unsigned short) course / 10.0);
    v.sog = (float) ((float) (unsigned short) sog / 10.0);
    v.navstat = navstat;
}

void set_static(gnuais_vessel &v, int imo, int shiptype, int A, int B, int C, int D, float draught)
{
    v.set |= GNUAIS_V_DATA | GNUAIS_V_STATIC;
    v.imo = imo; v.shiptype = shiptype; v.A = A; v.B = B; v.C = C; v.D = D; v.draught = draught;
}

void set_name(gnuais_vessel &v, const std::string &name, const std::string &destination)   // :336-361
{
    v.set |= GNUAIS_V_DATA | GNUAIS_V_NAME;
    put(v.name, sizeof v.name, name);
    put(v.destination, sizeof v.destination, destination);
}

void set_callsign(gnuais_vessel &v, const std::string &cs)
{
    v.set |= GNUAIS_V_CALLSIGN;
    put(v.callsign, sizeof v.callsign, cs);
}

// what one frame does to its vessel's entry: the cache_*() calls of the per-type decoders
bool touches_cache(const Bits &b, unsigned type)
{
    switch (type) {
    case 1: case 2: case 3: case 4: case 5: case 18: case 19: return true;
    case 24: return b.get(38, 2) <= 1;
    case 6: return b.get(72, 10) == 1 && b.get(82, 6) == 40;
    case 8: return b.get(40, 10) == 1 && b.get(50, 6) == 40;
    default: return false;
    }
}

void fold(gnuais_vessel &v, const Bits &b, unsigned type)
{
    switch (type) {
    case 1: case 2: case 3:                          // protodec_pos :390-397
        set_position(v, (int) (signed char) b.get(38, 2), b.sget(89, 27), b.sget(61, 28), (int) b.get(128, 9),
                     (unsigned) b.get(116, 12), (unsigned) b.get(50, 10));
        break;
    case 4:                                          // protodec_4 :435-439: a position, nothing else known
        set_position(v, 0, b.sget(107, 27), b.sget(79, 28), 0, 0, 0);
        break;
    case 18:                                         // protodec_18 :619-626
        set_position(v, 15, b.sget(85, 27), b.sget(57, 28), (int) b.get(124, 9), (unsigned) b.get(112, 12),
                     (unsigned) b.get(46, 10));
        break;
    case 5: {                                        // protodec_5 :516-518 -> cache_vesseldata :235-275
        const unsigned char draught = (unsigned char) b.get(294, 8);
        set_callsign(v, b.text(70, 6));
        set_name(v, b.text(112, 20), b.text(302, 20));
        set_static(v, (int) b.get(40, 30), (int) b.get(232, 8), (int) b.get(240, 9), (int) b.get(249, 9),
                   (unsigned char) b.get(258, 6), (unsigned char) b.get(264, 6), (float) (draught / 10.0));
        break;
    }
    case 19:                                         // protodec_19 :676-678: class B has no destination
        set_name(v, b.text(143, 20), "CLASS B");
        set_static(v, 0, (int) b.get(263, 8), (int) b.get(271, 9), (int) b.get(280, 9),
                   (unsigned char) b.get(289, 6), (unsigned char) b.get(295, 6), 0);
        break;
    case 24:                                         // protodec_24 :740-741, :772-774
        if (b.get(38, 2) == 0) set_name(v, b.text(40, 20), "CLASS B");
        if (b.get(38, 2) == 1) {
            set_callsign(v, b.text(90, 6));
            set_static(v, 0, (int) b.get(40, 8), (int) b.get(132, 9), (int) b.get(141, 9),
                       (unsigned char) b.get(150, 6), (unsigned char) b.get(156, 6), 0);
        }
        break;
    case 6: case 8:                                  // protodec_msg_40 :279-285
        v.set |= GNUAIS_V_PERSONS;
        v.persons_on_board = (int) b.get(type == 6 ? 88 : 56, 13);
        break;
    default: break;
    }
}

} // namespace

extern "C" int gnuais_vessels_from_frames(const gnuais_frame *frames, int n_frames, gnuais_vessel *vessels,
                                          int cap, int *n_vessels)
{
    if (n_frames < 0 || (n_frames > 0 && !frames) || !vessels || cap < 0 || !n_vessels || *n_vessels < 0 ||
        *n_vessels > cap)
        return GNUAIS_E_ARG;
    const int n_old = *n_vessels;
    for (int i = 1; i < n_old; ++i)
        if (vessels[i].mmsi <= vessels[i - 1].mmsi) return GNUAIS_E_ARG;
    // (mmsi, arrival order) of every frame that reaches a cache_*() call; sorted, each vessel's
    // updates are contiguous and still in arrival order
    std::vector<std::pair<int, int>> upd;
    upd.reserve((size_t) n_frames);
    for (int k = 0; k < n_frames; ++k) {
        const gnuais_frame &f = frames[k];
        if (f.nbits > MAX_NBITS) return GNUAIS_E_ARG;
        const Bits b(f, (int) f.nbits);
        const unsigned type = (unsigned) b.get(0, 6);
        if (type < 1 || type > MAX_TYPE || !touches_cache(b, type)) continue;
        upd.emplace_back((int) b.get(8, 30), k);
    }
    std::sort(upd.begin(), upd.end());
    std::vector<gnuais_vessel> out;
    out.reserve((size_t) n_old + upd.size());
    int o = 0;
    for (size_t u = 0; u < upd.size();) {
        const int mmsi = upd[u].first;
        while (o < n_old && vessels[o].mmsi < mmsi) out.push_back(vessels[o++]);
        gnuais_vessel v = (o < n_old && vessels[o].mmsi == mmsi) ? vessels[o++] : fresh_vessel(mmsi);
        for (; u < upd.size() && upd[u].first == mmsi; ++u) {
            const gnuais_frame &f = frames[upd[u].second];
            const Bits b(f, (int) f.nbits);
            fold(v, b, (unsigned) b.get(0, 6));
        }
        out.push_back(v);
    }
    while (o < n_old) out.push_back(vessels[o++]);
    if ((int) out.size() > cap) return GNUAIS_E_OVERFLOW;
    if (!out.empty()) memcpy(vessels, out.data(), out.size() * sizeof(gnuais_vessel));
    *n_vessels = (int) out.size();
    return GNUAIS_OK;
}

// The MySQL sink's calls for a batch (include/gnuais_hip.h).  Every argument is formed as the reference forms it at its
// call site (protodec.c:383-388, 430-433, 510-514, 612-617, 670-674, 737-738, 768-771): float expressions in double,
// narrowed at the call.  keepsmall != 0 (the reference's mysql_keepsmall, out_mysql.c:140: UPDATE ... WHERE mmsi, INSERT
// only when no row was touched): per (vessel, kind) only the last call leaves anything, so only that one is kept.
// keepsmall == 0 (the reference's default, cfg.c:74): every call INSERTs a row of its own, so every call is kept, in
// arrival order.
extern "C" int gnuais_sql_calls_from_frames(const gnuais_frame *frames, int n_frames, int keepsmall, gnuais_sql_call *out,
                                            int cap, int *n_out)
{
    if (n_frames < 0 || (n_frames > 0 && !frames) || !out || cap < 0 || !n_out) return GNUAIS_E_ARG;
    struct Rec { int mmsi, kind, order; gnuais_sql_call c; };
    std::vector<Rec> recs;
    auto add = [&](const gnuais_sql_call &c, int order) { recs.push_back(Rec{c.mmsi, c.kind, order, c}); };
    for (int k = 0; k < n_frames; ++k) {
        const gnuais_frame &f = frames[k];
        if (f.nbits > MAX_NBITS) return GNUAIS_E_ARG;
        const Bits b(f, (int) f.nbits);
        const unsigned type = (unsigned) b.get(0, 6);
        if (type < 1 || type > MAX_TYPE) continue;
        gnuais_sql_call c;
        memset(&c, 0, sizeof c);
        c.mmsi = (int) b.get(8, 30);
        auto pos = [&](long lat, long lon, int hdg, unsigned course, unsigned sog) {
            c.kind = GNUAIS_SQL_POSITION;
            c.lat = (float) ((float) lat / 600000.0);
            c.lon = (float) ((float) lon / 600000.0);
            c.hdg = (float) hdg;
            c.course = (float) ((float) (unsigned short) course / 10.0);
            c.sog = (float) ((float) (unsigned short) sog / 10.0);
            add(c, 2 * k);
        };
        auto datab = [&](int A, int B, int C_, int D, int order) {
            gnuais_sql_call e = c;
            e.kind = GNUAIS_SQL_VESSELDATAB;
            e.A = A; e.B = B; e.C = C_; e.D = D;
            add(e, order);
        };
        auto vname = [&](const std::string &name, const std::string &dest, int order) {
            gnuais_sql_call e = c;
            e.kind = GNUAIS_SQL_VESSELNAME;
            put(e.name, sizeof e.name, name);
            put(e.destination, sizeof e.destination, dest);
            add(e, order);
        };
        switch (type) {
        case 1: case 2: case 3:
            pos(b.sget(89, 27), b.sget(61, 28), (int) b.get(128, 9), (unsigned) b.get(116, 12), (unsigned) b.get(50, 10));
            break;
        case 18:
            pos(b.sget(85, 27), b.sget(57, 28), (int) b.get(124, 9), (unsigned) b.get(112, 12), (unsigned) b.get(46, 10));
            break;
        case 4:
            c.kind = GNUAIS_SQL_BASESTATION;
            c.lat = (float) ((float) b.sget(107, 27) / 600000.0);
            c.lon = (float) ((float) b.sget(79, 28) / 600000.0);
            add(c, 2 * k);
            break;
        case 5: {
            const unsigned char draught = (unsigned char) b.get(294, 8);
            c.kind = GNUAIS_SQL_VESSELDATA;
            put(c.name, sizeof c.name, b.text(112, 20));
            put(c.destination, sizeof c.destination, b.text(302, 20));
            c.draught = (float) ((float) draught / 10.0);
            c.A = (int) b.get(240, 9); c.B = (int) b.get(249, 9);
            c.C = (unsigned char) b.get(258, 6); c.D = (unsigned char) b.get(264, 6);
            add(c, 2 * k);
            break;
        }
        case 19:                                     // name first, then the dimensions (protodec.c:671-673)
            vname(b.text(143, 20), "CLASS B", 2 * k);
            datab((int) b.get(271, 9), (int) b.get(280, 9), (unsigned char) b.get(289, 6), (unsigned char) b.get(295, 6), 2 * k + 1);
            break;
        case 24:
            if (b.get(38, 2) == 0) vname(b.text(40, 20), "CLASS B", 2 * k);
            if (b.get(38, 2) == 1)
                datab((int) b.get(132, 9), (int) b.get(141, 9), (unsigned char) b.get(150, 6), (unsigned char) b.get(156, 6), 2 * k);
            break;
        default: break;
        }
    }
    if (!keepsmall) {                               // every call is a row: nothing to reduce
        *n_out = (int) recs.size();
        if ((int) recs.size() > cap) return GNUAIS_E_OVERFLOW;
        for (size_t i2 = 0; i2 < recs.size(); ++i2) out[i2] = recs[i2].c;
        return GNUAIS_OK;
    }
    // per (vessel, kind) the last call; survivors in arrival order
    std::stable_sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b_) {
        return a.mmsi != b_.mmsi ? a.mmsi < b_.mmsi : (a.kind != b_.kind ? a.kind < b_.kind : a.order < b_.order);
    });
    std::vector<Rec> keep;
    for (size_t i2 = 0; i2 < recs.size(); ++i2)
        if (i2 + 1 == recs.size() || recs[i2 + 1].mmsi != recs[i2].mmsi || recs[i2 + 1].kind != recs[i2].kind) keep.push_back(recs[i2]);
    std::sort(keep.begin(), keep.end(), [](const Rec &a, const Rec &b_) { return a.order < b_.order; });
    *n_out = (int) keep.size();
    if ((int) keep.size() > cap) return GNUAIS_E_OVERFLOW;
    for (size_t i2 = 0; i2 < keep.size(); ++i2) out[i2] = keep[i2].c;
    return GNUAIS_OK;
}

// the reduced plan under its round-3 name: what a database run with mysql_keepsmall on needs
extern "C" int gnuais_sql_plan_from_frames(const gnuais_frame *frames, int n_frames, gnuais_sql_call *out, int cap, int *n_out)
{
    return gnuais_sql_calls_from_frames(frames, n_frames, 1, out, cap, n_out);
}

extern "C" int gnuais_nmea_from_frames(const gnuais_frame *frames, int n_frames, uint8_t *seqnr,
                                       int n_channels, char *out, size_t out_cap, size_t *out_len,
                                       int *n_sentences)
{
    if (!out_len) return GNUAIS_E_ARG;
    return gnuais_messages_from_frames(frames, n_frames, seqnr, nullptr, n_channels, out, out_cap, out_len,
                                       n_sentences, nullptr, 0, nullptr, nullptr);
}
