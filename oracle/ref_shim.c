/*
 * ref_shim.c -- capture harness around the REAL gnuais reference objects.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is linked (by oracle/Makefile) together
 * with the reference's own, unmodified translation units compiled in place
 * from /root/reference/src (filter.c, receiver.c, protodec.c and their
 * support files) into oracle/_ref/libgnuais_ref.so.  Nothing from the
 * reference is copied into this repository; this file only *calls* the
 * reference's public functions and records what they do, so that
 *   - oracle/ais_oracle.c (the CPU restatement) can be pinned against the real
 *     code on arbitrary inputs (tests/test_oracle_vs_ref.py), and
 *   - tests/golden/make_golden.py can emit golden vectors.
 *
 * Tap points:
 *   floats : filter_run_buf()      reference src/filter.c:106-143
 *   bits   : --wrap=protodec_decode  (called from src/receiver.c:130)
 *   frames : observed from the same wrap (receivedframes moves, protodec.c:1103)
 *   NMEA   : --wrap=serial_write   (every sentence protodec_generate_nmea emits,
 *            protodec.c:883-885; the decoder gets a dummy non-NULL d->serial)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the resulting library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "receiver.h"   /* reference header: struct receiver, receiver_run   */
#include "filter.h"     /* reference header: struct filter, filter_run_buf   */
#include "protodec.h"   /* reference header: struct demod_state_t            */
#include "cfg.h"        /* reference header: skip_type[]                     */

/* ------------------------------------------------------------------ */
/* capture buffers                                                     */

typedef struct {
	uint32_t channel;     /* index of the ref receiver that produced it   */
	uint32_t end_bit;     /* number of bits fed to protodec before the
	                         STOPSIGN bit that closed the frame           */
	uint8_t  payload[53]; /* on-air bytes: bit i of byte j = buffer[8j+i]  */
	uint8_t  flags;       /* 1 = CRC ok (always 1 here)                    */
	uint16_t nbits;       /* bufferlen handed to protodec_getdata          */
} ref_frame_t;

#define MAX_RX 65536

static struct receiver *g_rx[MAX_RX];
static uint32_t g_bits_seen[MAX_RX];
static int g_nrx = 0;

static uint8_t *g_bits = NULL;          /* captured bits of g_bit_rx only */
static size_t g_bits_n = 0, g_bits_cap = 0;
static int g_bit_rx = -1;               /* which receiver's bits to keep, -2 = all (in call order) */

static ref_frame_t *g_frames = NULL;
static size_t g_frames_n = 0, g_frames_cap = 0;


static int g_fast = 0;                  /* bench mode: no bookkeeping in the wrap */
static int g_last_idx = 0;

static int rx_index_of_decoder(struct demod_state_t *d)
{
	int i;
	if (g_last_idx < g_nrx && g_rx[g_last_idx] && g_rx[g_last_idx]->decoder == d)
		return g_last_idx;
	for (i = 0; i < g_nrx; i++)
		if (g_rx[i] && g_rx[i]->decoder == d) {
			g_last_idx = i;
			return i;
		}
	return -1;
}

/* ------------------------------------------------------------------ */
/* link-time wraps                                                     */

void __real_protodec_decode(char *in, int count, struct demod_state_t *d);

/* A CRC-valid frame is recognised from the outside: receivedframes moves
 * (protodec.c:1103) during a step that started in ST_STOPSIGN, where the frame
 * length is bufferpos - 22 (protodec.c:1096) and d->rbuffer was just filled by
 * protodec_calculate_crc (protodec.c:150-162).  protodec_getdata() is called
 * from inside the same translation unit, so it cannot be link-wrapped. */
static void record_frame(int idx, int bufferlen, struct demod_state_t *d)
{
	ref_frame_t *f;
	int j, i, nbytes;

	if (g_frames_n == g_frames_cap) {
		g_frames_cap = g_frames_cap ? g_frames_cap * 2 : 1024;
		g_frames = realloc(g_frames, g_frames_cap * sizeof(*g_frames));
	}
	f = &g_frames[g_frames_n++];
	memset(f, 0, sizeof(*f));
	f->channel = (uint32_t) idx;
	f->end_bit = idx >= 0 ? g_bits_seen[idx] : 0;
	f->nbits = (uint16_t) bufferlen;
	f->flags = 1;
	nbytes = bufferlen / 8;
	if (nbytes > 53)
		nbytes = 53;
	/* d->rbuffer holds the payload MSB-first per byte; fold it back to the
	 * on-air bytes (bit i of byte j = buffer[8j+i], protodec.c:138-143) */
	for (j = 0; j < nbytes; j++) {
		unsigned v = 0;
		for (i = 0; i < 8; i++)
			v |= (unsigned) (d->rbuffer[8 * j + i] & 1) << (7 - i);
		f->payload[j] = (uint8_t) v;
	}
}

void __wrap_protodec_decode(char *in, int count, struct demod_state_t *d)
{
	int idx, i;
	if (g_fast) {                   /* timed runs: straight through */
		__real_protodec_decode(in, count, d);
		return;
	}
	idx = rx_index_of_decoder(d);
	if (idx >= 0 && (g_bit_rx == idx || g_bit_rx == -2)) {
		for (i = 0; i < count; i++) {
			if (g_bits_n == g_bits_cap) {
				g_bits_cap = g_bits_cap ? g_bits_cap * 2 : 65536;
				g_bits = realloc(g_bits, g_bits_cap);
			}
			g_bits[g_bits_n++] = (uint8_t) in[i];
		}
	}
	/* the reference always calls with count == 1 (receiver.c:130); feed one
	 * bit at a time so the per-bit bookkeeping is exact for any count */
	for (i = 0; i < count; i++) {
		int pre_len = (d->state == ST_STOPSIGN) ? d->bufferpos - 22 : -1;
		int pre_rx = d->receivedframes;
		__real_protodec_decode(in + i, 1, d);
		if (d->receivedframes != pre_rx)
			record_frame(idx, pre_len, d);
		if (idx >= 0)
			g_bits_seen[idx]++;
	}
}

/* ------------------------------------------------------------------ */
/* exported control surface (ctypes)                                   */

void ref_reset_all(void)
{
	int i;
	for (i = 0; i < g_nrx; i++) {
		if (g_rx[i]) {
			/* the reference leaks the decoder (receiver.c:76-82); tidy up */
			free(g_rx[i]->decoder->buffer);
			free(g_rx[i]->decoder->rbuffer);
			free(g_rx[i]->decoder->serbuffer);
			free(g_rx[i]->decoder->ipcbuffer);
			free(g_rx[i]->decoder->nmea);
			free(g_rx[i]->decoder);
			free_receiver(g_rx[i]);
			g_rx[i] = NULL;
		}
		g_bits_seen[i] = 0;
	}
	g_nrx = 0;
	g_bits_n = 0;
	g_frames_n = 0;
	g_bit_rx = -1;
}

/* stdout text of protodec_getdata on/off; when off every type is skipped */
void ref_set_text(int on)
{
	int i;
	for (i = 0; i <= MAX_AIS_PACKET_TYPE; i++)
		skip_type[i] = on ? 0 : 1;
}

/* create a reference receiver exactly as ais.c:139-147 does */
int ref_receiver_new(char name, int num_ch, int ch_ofs)
{
	if (g_nrx >= MAX_RX)
		return -1;
	g_rx[g_nrx] = init_receiver(name, num_ch, ch_ofs, NULL, NULL);
	g_bits_seen[g_nrx] = 0;
	return g_nrx++;
}

/* drive the unmodified reference functions with
 * another tap table / pllinc through the public structs */
int ref_receiver_set_params(int idx, const float *taps, int n_taps, unsigned pllinc)
{
	if (idx < 0 || idx >= g_nrx)
		return -1;
	if (taps && n_taps > 0) {
		filter_free(g_rx[idx]->filter);
		g_rx[idx]->filter = filter_init(n_taps, (float *) taps);
	}
	if (pllinc)
		g_rx[idx]->pllinc = pllinc;
	return 0;
}

void ref_receiver_run(int idx, short *buf, int len)
{
	receiver_run(g_rx[idx], buf, len);
}

/* run every receiver over an interleaved stream in chunks the way the main
 * loop does (ais.c:214-247): chunk, then receiver 0..n-1 */
void ref_run_stream(short *buf, int total_len, int chunk)
{
	int pos = 0, i;
	while (pos < total_len) {
		int n = total_len - pos;
		int stride = g_nrx ? g_rx[0]->num_ch : 1;
		if (n > chunk)
			n = chunk;
		for (i = 0; i < g_nrx; i++)
			receiver_run(g_rx[i], buf + (size_t) pos * stride, n);
		pos += n;
	}
}

int ref_get_taps(int idx, float *out, int max)
{
	int n = g_rx[idx]->filter->length, i;
	for (i = 0; i < n && i < max; i++)
		out[i] = g_rx[idx]->filter->taps[i];
	return n;
}

void ref_get_pll(int idx, unsigned *pll, int *prev, int *lastbit)
{
	*pll = g_rx[idx]->pll;
	*prev = g_rx[idx]->prev;
	*lastbit = g_rx[idx]->lastbit;
}

void ref_get_counters(int idx, int *received, int *lost, int *lost2)
{
	*received = g_rx[idx]->decoder->receivedframes;
	*lost = g_rx[idx]->decoder->lostframes;
	*lost2 = g_rx[idx]->decoder->lostframes2;
}

/* FSM snapshot used to pin the restatement's carry state */
void ref_get_fsm(int idx, int *out7)
{
	struct demod_state_t *d = g_rx[idx]->decoder;
	out7[0] = d->state;
	out7[1] = d->nstartsign;
	out7[2] = d->antallpreamble;
	out7[3] = d->antallenner;
	out7[4] = d->bitstuff;
	out7[5] = d->last;
	out7[6] = d->bufferpos;
}

void ref_capture_bits_of(int idx) { g_bit_rx = idx; g_bits_n = 0; }
size_t ref_bits_count(void) { return g_bits_n; }
const uint8_t *ref_bits_ptr(void) { return g_bits; }
void ref_bits_clear(void) { g_bits_n = 0; }

size_t ref_frames_count(void) { return g_frames_n; }
const ref_frame_t *ref_frames_ptr(void) { return g_frames; }
void ref_frames_clear(void) { g_frames_n = 0; }
int ref_frame_size(void) { return (int) sizeof(ref_frame_t); }

/* filter_run_buf() on a private filter: floats + maxval per chunk */
int ref_filter_stream(const float *taps, int n_taps, short *in, int step,
		      int total_len, int chunk, float *out, short *maxvals)
{
	struct filter *f = filter_init(n_taps, (float *) taps);
	int pos = 0, c = 0;
	while (pos < total_len) {
		int n = total_len - pos;
		if (n > chunk)
			n = chunk;
		short m = filter_run_buf(f, in + (size_t) pos * step, out + pos, step, n);
		if (maxvals)
			maxvals[c] = m;
		c++;
		pos += n;
	}
	filter_free(f);
	return c;
}

/* feed raw bits to a fresh reference decoder (protodec_decode parity) */
void ref_decode_bits(int idx, const uint8_t *bits, int n)
{
	int i;
	for (i = 0; i < n; i++) {
		char b = (char) bits[i];
		protodec_decode(&b, 1, g_rx[idx]->decoder);
	}
}

/* ------------------------------------------------------------------ */
/* message layer (row f1): NMEA sentences of CRC-valid frames          */

static char *g_nmea = NULL;
static size_t g_nmea_n = 0, g_nmea_cap = 0;
static int g_nmea_sentences = 0;

extern int __real_serial_write(struct serial_state_t *state, char *s, int len);
static int g_serial_live = 0;   /* timing runs: let the reference's serial_write() do its write() */

int __wrap_serial_write(struct serial_state_t *state, char *s, int len)
{
	if (g_serial_live)
		return __real_serial_write(state, s, len);
	if (g_nmea_n + (size_t) len > g_nmea_cap) {
		g_nmea_cap = (g_nmea_n + (size_t) len) * 2 + 4096;
		g_nmea = realloc(g_nmea, g_nmea_cap);
	}
	memcpy(g_nmea + g_nmea_n, s, (size_t) len);
	g_nmea_n += (size_t) len;
	g_nmea_sentences++;
	return len;
}

void ref_nmea_clear(void) { g_nmea_n = 0; g_nmea_sentences = 0; }
size_t ref_nmea_bytes(void) { return g_nmea_n; }
int ref_nmea_sentences(void) { return g_nmea_sentences; }
const char *ref_nmea_ptr(void) { return g_nmea; }
int ref_get_seqnr(int idx) { return g_rx[idx]->decoder->seqnr; }
void ref_set_seqnr(int idx, int v) { g_rx[idx]->decoder->seqnr = (unsigned char) v; }

/* Hand one CRC-valid frame to the reference's message layer the way protodec_decode does
 * (protodec.c:1104 after protodec_calculate_crc filled d->rbuffer, :150-162): payload bits
 * MSB-first per byte, one bit per char, then protodec_getdata(bufferlen, d).  `payload`
 * is the packed form of the frame records (bit 7-i of byte j = rbuffer[8j+i]). */
void ref_getdata(int idx, const uint8_t *payload, int nbits)
{
	static struct serial_state_t *dummy = (struct serial_state_t *) 16;   /* never dereferenced: wrapped */
	struct demod_state_t *d = g_rx[idx]->decoder;
	int i;
	memset(d->rbuffer, 0, DEMOD_BUFFER_LEN);
	for (i = 0; i < nbits && i < DEMOD_BUFFER_LEN - 8; i++)
		d->rbuffer[i] = (i >> 3) < 53 ? (payload[i >> 3] >> (7 - (i & 7))) & 1 : 0;   /* 53 payload bytes */
	d->serial = dummy;
	protodec_getdata(nbits, d);
	d->serial = NULL;
}

/* the same, but also keeping what protodec_getdata() prints on stdout (protodec.c:934-985):
 * fd 1 is pointed at a scratch file for the duration of the call */
static char *g_text = NULL;
static size_t g_text_n = 0, g_text_cap = 0;
void ref_text_clear(void) { g_text_n = 0; }
size_t ref_text_bytes(void) { return g_text_n; }
const char *ref_text_ptr(void) { return g_text; }

#include <unistd.h>
void ref_getdata_text(int idx, const uint8_t *payload, int nbits)
{
	static FILE *scratch = NULL;
	int saved, i;
	long n;
	if (!scratch)
		scratch = tmpfile();
	if (!scratch) {
		ref_getdata(idx, payload, nbits);
		return;
	}
	for (i = 0; i <= MAX_AIS_PACKET_TYPE; i++)
		skip_type[i] = 0;
	fflush(stdout);
	rewind(scratch);
	if (ftruncate(fileno(scratch), 0) != 0) { /* keep going: the read below is bounded by ftell */ }
	saved = dup(1);
	dup2(fileno(scratch), 1);
	ref_getdata(idx, payload, nbits);
	fflush(stdout);
	dup2(saved, 1);
	close(saved);
	n = lseek(fileno(scratch), 0, SEEK_END);
	if (n > 0) {
		if (g_text_n + (size_t) n > g_text_cap) {
			g_text_cap = (g_text_n + (size_t) n) * 2 + 4096;
			g_text = realloc(g_text, g_text_cap);
		}
		lseek(fileno(scratch), 0, SEEK_SET);
		if (read(fileno(scratch), g_text + g_text_n, (size_t) n) == n)
			g_text_n += (size_t) n;
	}
	for (i = 0; i <= MAX_AIS_PACKET_TYPE; i++)
		skip_type[i] = 1;
}

/* range statistics: the station position as cfg.c:364-368 stores it after parsing the config
 * (degrees -> radians through the reference's own lat2rad/lon2rad), then the reference's
 * update_range() runs inside protodec_getdata() for position reports (protodec.c:399,441,628) */
extern float mylat, mylng;
extern int have_my_loc;
extern float lat2rad(float), lon2rad(float);
void ref_set_location(float lat_deg, float lon_deg)
{
	mylat = lat_deg;
	mylng = lon_deg;
	have_my_loc = (mylat > -90 && mylat < 90 && mylng > -180 && mylng < 180);
	if (have_my_loc) {
		mylat = lat2rad(mylat);
		mylng = lon2rad(mylng);
	}
}
float ref_best_range(int idx) { return g_rx[idx]->decoder->best_range; }

/* position cache: switch the reference's own cache on (cache.c:59-77), let protodec_getdata()
 * fill it, then take it out with cache_rotate() (cache.c:141-155) and flatten it the way
 * out_json.c:241-262 walks it (sp_fhead/sp_fnext, key order).  Record = gnuais_vessel of
 * include/gnuais_hip.h; timestamps are reported as "set or not" only. */
#include "cache.h"
struct flat_vessel {
	int32_t mmsi; uint32_t set; float lat, lon; int32_t hdg; float course, sog; int32_t navstat;
	int32_t imo, shiptype, A, B, C, D; float draught; int32_t persons_on_board;
	char callsign[8], name[24], destination[24];
};
void ref_cache_enable(void)
{
	if (!cache_positions)
		cache_init();
}
int ref_cache_take(struct flat_vessel *out, int cap)
{
	struct sptree *sp = cache_rotate();
	struct spblk *x;
	int n = 0;
	for (x = sp_fhead(sp); x != NULL; x = sp_fnext(x)) {
		struct cache_ent *e = (struct cache_ent *) x->data;
		if (n < cap) {
			struct flat_vessel *v = &out[n];
			memset(v, 0, sizeof(*v));
			v->mmsi = e->mmsi;
			v->set = (e->received_pos ? 1u : 0) | (e->received_data ? 2u : 0) |
				 (e->received_persons_on_board ? 4u : 0) | ((e->name && e->destination) ? 8u : 0) |
				 (e->callsign ? 16u : 0) | (e->shiptype != -1 || e->imo != -1 ? 32u : 0);
			v->lat = e->lat; v->lon = e->lon; v->hdg = e->hdg; v->course = e->course; v->sog = e->sog;
			v->navstat = e->navstat; v->imo = e->imo; v->shiptype = e->shiptype;
			v->A = e->A; v->B = e->B; v->C = e->C; v->D = e->D; v->draught = e->draught;
			v->persons_on_board = e->persons_on_board;
			if (e->callsign) strncpy(v->callsign, e->callsign, sizeof(v->callsign) - 1);
			if (e->name) strncpy(v->name, e->name, sizeof(v->name) - 1);
			if (e->destination) strncpy(v->destination, e->destination, sizeof(v->destination) - 1);
		}
		n++;
	}
	cache_free(sp);
	hfree(sp);
	return n;
}

/* timing (scripts/time_sinks.py): `n` frame records (gnuais_frame, 64 bytes: channel at 0, payload
 * at 8, nbits at 62) through the reference's protodec_getdata() one by one, with its sinks live:
 * serial_write() on `serial_fd` (>= 0), printf/fflush on whatever fd 1 is (with_text), the
 * position cache when it was enabled.  Returns seconds. */
#include <time.h>
double ref_getdata_many(const uint8_t *records, int n, int serial_fd, int with_text)
{
	struct serial_state_t port;
	struct timespec t0, t1;
	int i, k;
	port.fd = serial_fd;
	for (i = 0; i <= MAX_AIS_PACKET_TYPE; i++)
		skip_type[i] = with_text ? 0 : 1;
	g_serial_live = 1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (k = 0; k < n; k++) {
		const uint8_t *r = records + (size_t) k * 64;
		struct demod_state_t *d = g_rx[*(const uint32_t *) r]->decoder;
		const int nbits = *(const uint16_t *) (r + 62);
		memset(d->rbuffer, 0, DEMOD_BUFFER_LEN);
		for (i = 0; i < nbits && i < DEMOD_BUFFER_LEN - 8; i++)
			d->rbuffer[i] = (i >> 3) < 53 ? (r[8 + (i >> 3)] >> (7 - (i & 7))) & 1 : 0;
		d->serial = serial_fd >= 0 ? &port : NULL;
		protodec_getdata(nbits, d);
		d->serial = NULL;
	}
	fflush(stdout);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	g_serial_live = 0;
	for (i = 0; i <= MAX_AIS_PACKET_TYPE; i++)
		skip_type[i] = 1;
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

extern unsigned short protodec_sdlc_crc(const unsigned char *data, unsigned len);
unsigned ref_sdlc_crc(const unsigned char *data, unsigned len)
{
	return protodec_sdlc_crc(data, len);
}

/* timed loop for bench.py's cpu_baseline (kind "reference"): every receiver
 * over the whole interleaved stream, text off.  Returns frames received. */
long ref_bench_run(short *buf, int total_len, int chunk)
{
	int i;
	long tot = 0;
	g_fast = 1;
	ref_run_stream(buf, total_len, chunk);
	g_fast = 0;
	for (i = 0; i < g_nrx; i++)
		tot += g_rx[i]->decoder->receivedframes;
	return tot;
}


/* ---- the MySQL sink's call log.  libmysqlclient is absent here, so the reference's out_mysql.c compiles to stubs that
 * return -1; what the message layer ASKS of the sink is observable all the same: the myout_*() calls are wrapped
 * (-Wl,--wrap) and written down, one line per call, floats with nine significant digits (exact round trip). */
#include <stdarg.h>
#include "out_mysql.h"
static char *g_sql = NULL;
static size_t g_sql_n = 0, g_sql_cap = 0;
static void sql_log(const char *fmt, ...)
{
	va_list ap;
	int n;
	if (g_sql_n + 512 > g_sql_cap) {
		g_sql_cap = g_sql_cap * 2 + 65536;
		g_sql = realloc(g_sql, g_sql_cap);
	}
	va_start(ap, fmt);
	n = vsnprintf(g_sql + g_sql_n, 512, fmt, ap);
	va_end(ap);
	if (n > 0)
		g_sql_n += (size_t) (n < 512 ? n : 511);
}
int __wrap_myout_ais_position(struct mysql_state_t *m, time_t t, int mmsi, float lat, float lon, float hdg, float course, float sog)
{ (void) m; (void) t; sql_log("position %d %.9g %.9g %.9g %.9g %.9g\n", mmsi, lat, lon, hdg, course, sog); return 0; }
int __wrap_myout_ais_basestation(struct mysql_state_t *m, time_t t, int mmsi, float lat, float lon)
{ (void) m; (void) t; sql_log("basestation %d %.9g %.9g\n", mmsi, lat, lon); return 0; }
int __wrap_myout_ais_vesseldata(struct mysql_state_t *m, time_t t, int mmsi, char *name, char *destination, float draught, int A, int B, int C, int D)
{ (void) m; (void) t; sql_log("vesseldata %d %.9g %d %d %d %d |%s|%s|\n", mmsi, draught, A, B, C, D, name, destination); return 0; }
int __wrap_myout_ais_vesseldatab(struct mysql_state_t *m, time_t t, int mmsi, int A, int B, int C, int D)
{ (void) m; (void) t; sql_log("vesseldatab %d %d %d %d %d\n", mmsi, A, B, C, D); return 0; }
int __wrap_myout_ais_vesselname(struct mysql_state_t *m, time_t t, int mmsi, const char *name, const char *destination)
{ (void) m; (void) t; sql_log("vesselname %d |%s|%s|\n", mmsi, name, destination); return 0; }
int __wrap_myout_nmea(struct mysql_state_t *m, time_t t, char *nmea)
{ (void) m; (void) t; sql_log("nmea %s\n", nmea); return 0; }
void ref_mysql_enable(int on)
{
	static struct mysql_state_t dummy;
	my = on ? &dummy : NULL;
}
void ref_sql_clear(void) { g_sql_n = 0; }
size_t ref_sql_bytes(void) { return g_sql_n; }
const char *ref_sql_ptr(void) { return g_sql; }
