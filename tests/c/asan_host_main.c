/*
 * asan_host_main.c -- TEST INFRASTRUCTURE.  Runs the host-side C/C++ of the drop-in under
 * -fsanitize=address,undefined on the CPU (`make -C gnuais_amd/csrc asan` -> tests/c/asan_host.bin,
 * tests/test_sanitizers.py): wavio.c (raw + RIFF readers), nmea.cpp (message layer, vessel fold, range, SQL plan),
 * sinks_batch.c (the batched front, against recording sinks) and -- over tests/c/fake_gnuais_hip.c, a double of the C
 * ABI on the CPU oracle -- receiver_hip.c (single- and multi-threaded) and protodec_hip.c (per-bit protodec_decode at
 * the default batching, the CRC names, filter_run_buf).  It writes what it saw into files the Python test compares with
 * the golden vectors; every finding of the sanitizers ends the program (-fno-sanitize-recover).
 *
 * usage: asan_host.bin <dir>     reads <dir>/stereo.raw stereo.wav frames.bin bits_a.bin, writes <dir>/out_*.
 */
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gnuais_receiver_abi.h"
#include "gnuais_hip.h"
#include "gnuais_sinks.h"

/* ------------------------------------------------------------ the message layer's two entry points, recording */

static pthread_mutex_t out_lock = PTHREAD_MUTEX_INITIALIZER;
static FILE *frames_out;

void protodec_initialize(struct demod_state_t *d, struct serial_state_t *serial, struct ipc_state_t *ipc, char chanid)
{
	memset(d, 0, sizeof(*d));
	d->chanid = chanid;
	d->serial = serial;
	d->ipc = ipc;
	d->state = 1;
	d->buffer = calloc(1, DEMOD_BUFFER_LEN);
	d->rbuffer = calloc(1, DEMOD_BUFFER_LEN);
}

void protodec_getdata(int bufferlen, struct demod_state_t *d)
{
	int i;
	pthread_mutex_lock(&out_lock);
	fprintf(frames_out, "ch %c bits %d payload ", d->chanid, bufferlen);
	for (i = 0; i + 8 <= bufferlen; i += 8) {
		int v = 0, k;
		for (k = 0; k < 8; k++)
			v = (v << 1) | d->rbuffer[i + k];
		fprintf(frames_out, "%02x", v);
	}
	fprintf(frames_out, "\n");
	pthread_mutex_unlock(&out_lock);
}

/* ------------------------------------------------------------ recording sinks (the names sinks_batch.c links to) */

static FILE *sink_log;
int mysql_keepsmall;
struct mysql_state_t { int unused; };

int serial_write(struct serial_state_t *s, char *p, int len) { (void) s; fprintf(sink_log, "serial %d %.20s\n", len, p); return len; }
int ipc_write(struct ipc_state_t *i, char *p, int len) { (void) i; (void) p; fprintf(sink_log, "ipc %d\n", len); return len; }
int cache_position(int t, int mmsi, int navstat, float lat, float lon, int hdg, float course, int rateofturn, float sog)
{ (void) t; fprintf(sink_log, "cpos %d %d %.9g %.9g %d %.9g %d %.9g\n", mmsi, navstat, lat, lon, hdg, course, rateofturn, sog); return 0; }
int cache_vesseldata(int t, int mmsi, int imo, char *callsign, char *name, char *destination, int shiptype, int A, int B,
		     int C, int D, float draught)
{ (void) t; fprintf(sink_log, "cvd %d %d |%s|%s|%s| %d %d %d %d %d %.9g\n", mmsi, imo, callsign, name, destination, shiptype, A, B, C, D, draught); return 0; }
int cache_vesseldatab(int t, int mmsi, char *callsign, int shiptype, int A, int B, int C, int D)
{ (void) t; fprintf(sink_log, "cvdb %d |%s| %d %d %d %d %d\n", mmsi, callsign, shiptype, A, B, C, D); return 0; }
int cache_vesseldatabb(int t, int mmsi, int shiptype, int A, int B, int C, int D)
{ (void) t; fprintf(sink_log, "cvdbb %d %d %d %d %d %d\n", mmsi, shiptype, A, B, C, D); return 0; }
int cache_vesselname(int t, int mmsi, char *name, const char *destination)
{ (void) t; fprintf(sink_log, "cname %d |%s|%s|\n", mmsi, name, destination); return 0; }
int cache_vessel_persons(int t, int mmsi, int persons) { (void) t; fprintf(sink_log, "cpers %d %d\n", mmsi, persons); return 0; }
int myout_ais_position(struct mysql_state_t *m, time_t t, int mmsi, float lat, float lon, float hdg, float course, float sog)
{ (void) m; (void) t; fprintf(sink_log, "position %d %.9g %.9g %.9g %.9g %.9g\n", mmsi, lat, lon, hdg, course, sog); return 0; }
int myout_ais_basestation(struct mysql_state_t *m, time_t t, int mmsi, float lat, float lon)
{ (void) m; (void) t; fprintf(sink_log, "basestation %d %.9g %.9g\n", mmsi, lat, lon); return 0; }
int myout_ais_vesseldata(struct mysql_state_t *m, time_t t, int mmsi, char *name, char *destination, float draught, int A,
			 int B, int C, int D)
{ (void) m; (void) t; fprintf(sink_log, "vesseldata %d %.9g %d %d %d %d |%s|%s|\n", mmsi, draught, A, B, C, D, name, destination); return 0; }
int myout_ais_vesseldatab(struct mysql_state_t *m, time_t t, int mmsi, int A, int B, int C, int D)
{ (void) m; (void) t; fprintf(sink_log, "vesseldatab %d %d %d %d %d\n", mmsi, A, B, C, D); return 0; }
int myout_ais_vesselname(struct mysql_state_t *m, time_t t, int mmsi, const char *name, const char *destination)
{ (void) m; (void) t; fprintf(sink_log, "vesselname %d |%s|%s|\n", mmsi, name, destination); return 0; }
int myout_nmea(struct mysql_state_t *m, time_t t, char *nmea) { (void) m; (void) t; fprintf(sink_log, "nmea %s\n", nmea); return 0; }

/* the reference's names served by protodec_hip.c */
struct filter;
struct filter *filter_init(int len, float *taps);
void filter_free(struct filter *f);
short filter_run_buf(struct filter *f, short *in, float *out, int step, int len);
void filter_run(struct filter *f, float in, float *out);
void protodec_decode(char *in, int count, struct demod_state_t *d);
unsigned short protodec_sdlc_crc(const unsigned char *data, unsigned len);
int protodec_calculate_crc(int length_bits, struct demod_state_t *d);
void protodec_reset(struct demod_state_t *d);
void protodec_deinit(struct demod_state_t *d);
void gnuais_protodec_flush(struct demod_state_t *d);
void gnuais_protodec_release(struct demod_state_t *d);
void gnuais_protodec_set_batching(int bits);

static char path[1024];
static const char *in_dir(const char *dir, const char *name)
{
	snprintf(path, sizeof path, "%s/%s", dir, name);
	return path;
}

static void *slurp(const char *p, size_t *n)
{
	FILE *f = fopen(p, "rb");
	void *buf;
	long len;
	if (!f) {
		perror(p);
		exit(2);
	}
	fseek(f, 0, SEEK_END);
	len = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf = malloc((size_t) len + 1);
	if (fread(buf, 1, (size_t) len, f) != (size_t) len)
		exit(2);
	fclose(f);
	*n = (size_t) len;
	return buf;
}

/* ------------------------------------------------------------ 1. file readers */

static void run_wav(const char *dir)
{
	FILE *out = fopen(in_dir(dir, "out_wav.txt"), "w");
	const char *names[2] = { "stereo.raw", "stereo.wav" };
	int k;
	for (k = 0; k < 2; k++) {
		gnuais_wav *w = NULL;
		int16_t buf[1020 * 2];
		long got, total = 0;
		unsigned long long sum = 0;
		if (gnuais_wav_open(&w, in_dir(dir, names[k]), k == 0 ? 2 : 0) != GNUAIS_OK) {
			fprintf(out, "%s: open failed\n", names[k]);
			continue;
		}
		while ((got = gnuais_wav_read(w, buf, 1020)) > 0) {
			long i;
			for (i = 0; i < got * gnuais_wav_channels(w); i++)
				sum = sum * 1000003ull + (uint16_t) buf[i];
			total += got;
		}
		fprintf(out, "%s channels %d rate %d frames %ld sum %llu\n", names[k], gnuais_wav_channels(w), gnuais_wav_rate(w),
			total, sum);
		gnuais_wav_close(w);
	}
	{       /* things that are not sound files */
		gnuais_wav *w = NULL;
		fprintf(out, "missing %d\n", gnuais_wav_open(&w, in_dir(dir, "no_such_file"), 0) != GNUAIS_OK);
		fprintf(out, "truncated %d\n", gnuais_wav_open(&w, in_dir(dir, "truncated.wav"), 0) != GNUAIS_OK);
		if (w)
			gnuais_wav_close(w);
	}
	fclose(out);
}

/* ------------------------------------------------------------ 2. message layer + 3. sinks */

static void run_messages_and_sinks(const char *dir)
{
	size_t bytes, nmea_len = 0, text_len = 0;
	gnuais_frame *fr = slurp(in_dir(dir, "frames.bin"), &bytes);
	const int n = (int) (bytes / sizeof(gnuais_frame)), n_ch = 64;
	uint8_t seq[64];
	int n_sent = 0, n_lines = 0, n_v = 0, n_sql = 0, i, mode;
	char *nmea, *text;
	gnuais_vessel *ves = calloc((size_t) n + 8, sizeof *ves);
	gnuais_sql_call *sql = calloc((size_t) 2 * n + 1, sizeof *sql);
	float range[64];
	FILE *out;

	memset(seq, 0, sizeof seq);
	if (gnuais_messages_from_frames(fr, n, seq, NULL, n_ch, NULL, 0, &nmea_len, &n_sent, NULL, 0, &text_len, &n_lines) != GNUAIS_OK)
		exit(3);
	nmea = malloc(nmea_len + 1);
	text = malloc(text_len + 1);
	memset(seq, 0, sizeof seq);
	if (gnuais_messages_from_frames(fr, n, seq, NULL, n_ch, nmea, nmea_len, &nmea_len, &n_sent, text, text_len, &text_len, &n_lines) != GNUAIS_OK)
		exit(3);
	out = fopen(in_dir(dir, "out_nmea.bin"), "wb");
	fwrite(nmea, 1, nmea_len, out);
	fclose(out);
	out = fopen(in_dir(dir, "out_text.bin"), "wb");
	fwrite(text, 1, text_len, out);
	fclose(out);
	/* too small a buffer is refused, not overrun */
	memset(seq, 0, sizeof seq);
	if (nmea_len > 8 && gnuais_nmea_from_frames(fr, n, seq, n_ch, nmea, nmea_len - 8, &nmea_len, &n_sent) != GNUAIS_E_OVERFLOW)
		exit(4);
	if (gnuais_vessels_from_frames(fr, n, ves, n + 8, &n_v) != GNUAIS_OK)
		exit(3);
	for (i = 0; i < n_ch; i++)
		range[i] = 0.0f;
	if (gnuais_range_from_frames(fr, n, n_ch, 60.0f, 10.0f, range) != GNUAIS_OK)
		exit(3);
	out = fopen(in_dir(dir, "out_misc.txt"), "w");
	fprintf(out, "sentences %d lines %d vessels %d\n", n_sent, n_lines, n_v);
	for (mode = 0; mode < 2; mode++) {
		if (gnuais_sql_calls_from_frames(fr, n, mode, sql, 2 * n + 1, &n_sql) != GNUAIS_OK)
			exit(3);
		fprintf(out, "sql keepsmall %d calls %d\n", mode, n_sql);
	}
	fclose(out);

	sink_log = fopen(in_dir(dir, "out_sinks.txt"), "w");
	{
		gnuais_sinks s;
		long counts[2] = { 0, 0 };
		struct mysql_state_t my = { 0 };
		const int cuts[5] = { 0, 1, n / 3, n - 1, n };
		memset(&s, 0, sizeof s);
		memset(seq, 0, sizeof seq);
		s.serial = (struct serial_state_t *) &my;   /* any non-NULL handle: the recording sinks never look */
		s.ipc = (struct ipc_state_t *) &my;
		s.text_out = sink_log;
		s.use_cache = 1;
		s.seqnr = seq;
		s.n_channels = n_ch;
		for (i = 0; i < 4; i++)
			if (gnuais_sinks_deliver(&s, fr + cuts[i], cuts[i + 1] - cuts[i]) != GNUAIS_OK)
				exit(5);
		for (mysql_keepsmall = 0; mysql_keepsmall < 2; mysql_keepsmall++)
			if (gnuais_sinks_deliver_mysql(&s, &my, 1234, fr, n, nmea, nmea_len, counts) != GNUAIS_OK)
				exit(5);
		fprintf(sink_log, "summary frames %ld sentences %ld serial_calls %ld ipc_calls %ld sql %ld\n", s.frames,
			s.sentences, s.serial_calls, s.ipc_calls, counts[0]);
		gnuais_sinks_free(&s);
	}
	fclose(sink_log);
	free(nmea);
	free(text);
	free(ves);
	free(sql);
	free(fr);
}

/* ------------------------------------------------------------ 4. the drop-in over the test double */

struct drive {
	const short *x;
	long frames;
	int channels;           /* receivers of the group = columns of x */
	char first;             /* name of the first receiver */
};

static void *drive_group(void *arg)
{
	struct drive *dv = arg;
	struct receiver *rx[2] = { NULL, NULL };
	short *buffer = malloc(sizeof(short) * 1020 * (size_t) dv->channels);
	long at = 0;
	int i;
	for (i = 0; i < dv->channels; i++)
		rx[i] = init_receiver((char) (dv->first + i), dv->channels, i, NULL, NULL);
	while (at < dv->frames) {       /* ais.c:214-247: one buffer, every receiver of the group */
		const long n = dv->frames - at < 1020 ? dv->frames - at : 1020;
		memcpy(buffer, dv->x + at * dv->channels, sizeof(short) * (size_t) n * (size_t) dv->channels);
		for (i = 0; i < dv->channels; i++)
			receiver_run(rx[i], buffer, (int) n);
		at += n;
	}
	pthread_mutex_lock(&out_lock);
	for (i = 0; i < dv->channels; i++)
		fprintf(frames_out, "%c: received %d lost %d lost2 %d pll %u\n", dv->first + i, rx[i]->decoder->receivedframes,
			rx[i]->decoder->lostframes, rx[i]->decoder->lostframes2, rx[i]->pll);
	pthread_mutex_unlock(&out_lock);
	for (i = 0; i < dv->channels; i++) {
		free(rx[i]->decoder->buffer);
		free(rx[i]->decoder->rbuffer);
		free(rx[i]->decoder);           /* the drop-in, like the reference, leaves the decoder to the caller */
		free_receiver(rx[i]);
	}
	free(buffer);
	return NULL;
}

static void run_dropin(const char *dir)
{
	size_t bytes;
	short *x = slurp(in_dir(dir, "stereo.raw"), &bytes);
	const long frames = (long) (bytes / 4);
	short *mono = malloc(sizeof(short) * (size_t) frames);
	struct drive two = { x, frames, 2, 'A' }, one = { mono, frames, 1, 'M' };
	pthread_t t1, t2;
	long i;
	for (i = 0; i < frames; i++)
		mono[i] = x[2 * i + 1];         /* the 1-channel group hears what B hears */
	frames_out = fopen(in_dir(dir, "out_dropin.txt"), "w");
	drive_group(&two);
	fclose(frames_out);
	/* two groups from two threads at once: the 2-channel group again and a 1-channel group */
	frames_out = fopen(in_dir(dir, "out_dropin_mt.txt"), "w");
	pthread_create(&t1, NULL, drive_group, &two);
	pthread_create(&t2, NULL, drive_group, &one);
	pthread_join(t1, NULL);
	pthread_join(t2, NULL);
	fclose(frames_out);
	free(mono);
	free(x);
}

/* ------------------------------------------------------------ 5. the reference-named shims over the test double */

static void run_shims(const char *dir)
{
	size_t bytes, nbits;
	short *x = slurp(in_dir(dir, "stereo.raw"), &bytes);
	unsigned char *bits = slurp(in_dir(dir, "bits_a.bin"), &nbits);
	const long frames = (long) (bytes / 4);
	float taps[36], *y = malloc(sizeof(float) * 1020);
	struct filter *f;
	struct demod_state_t d, e;
	unsigned long long sum = 0;
	short peak = 0;
	long at;
	size_t i;
	FILE *out = fopen(in_dir(dir, "out_shims.txt"), "w");

	frames_out = out;
	gnuais_default_taps(taps);
	f = filter_init(36, taps);
	for (at = 0; at < frames; at += 1020) {         /* receiver.c:107: channel A of the interleaved buffer */
		const int n = (int) (frames - at < 1020 ? frames - at : 1020);
		const short m = filter_run_buf(f, x + 2 * at, y, 2, n);
		int k;
		for (k = 0; k < n; k++) {
			uint32_t u;
			memcpy(&u, &y[k], 4);
			sum = sum * 1000003ull + u;
		}
		if (m > peak)
			peak = m;
	}
	fprintf(out, "filter sum %llu peak %d\n", sum, peak);
	{
		float one = 0.0f;
		filter_run(f, 123.0f, &one);
		fprintf(out, "filter_run %08x\n", *(uint32_t *) &one);
	}
	filter_free(f);
	filter_free(NULL);

	protodec_initialize(&d, NULL, NULL, 'A');
	protodec_initialize(&e, NULL, NULL, 'E');
	for (i = 0; i < nbits; i++) {                   /* receiver.c:130: one bit a call, two decoders alternating */
		char b = (char) bits[i];
		protodec_decode(&b, 1, &d);
		if (i % 3 == 0)
			protodec_decode(&b, 1, &e);         /* a second decoder that sees every third bit: noise */
		if (i % 204 == 203) {                       /* a buffer's worth of bits: the next filter_run_buf() flushes */
			struct filter *g = filter_init(36, taps);
			short s = 0;
			float o;
			filter_run_buf(g, &s, &o, 1, 1);
			filter_free(g);
		}
	}
	gnuais_protodec_flush(NULL);
	fprintf(out, "A: received %d lost %d lost2 %d state %d\n", d.receivedframes, d.lostframes, d.lostframes2, d.state);
	gnuais_protodec_release(&d);
	gnuais_protodec_release(&e);
	gnuais_protodec_release(&e);                    /* twice: a no-op */
	{       /* protodec_reset() / protodec_deinit() of the shim (protodec.c:78-100): a decoder reset every 997 bits -- inside
		 * frames too --, its d->buffer cells looked at every 500, its buffers freed by protodec_deinit */
		struct demod_state_t r;
		unsigned long long h = 1469598103934665603ull;
		int k, inside = 0;
		protodec_initialize(&r, NULL, NULL, 'R');
		r.serbuffer = malloc(8);                /* the three buffers the test's protodec_initialize leaves out: deinit frees them */
		r.ipcbuffer = malloc(8);
		r.nmea = malloc(8);
		protodec_reset(&r);                     /* before the first bit: no device object yet */
		for (i = 0; i < nbits; i++) {
			char b = (char) bits[i];
			if (i % 997 == 500)
				protodec_reset(&r);
			protodec_decode(&b, 1, &r);
			if (i % 500 == 499) {
				gnuais_protodec_flush(&r);
				if (r.state == 4 || r.state == 5) {
					inside++;
					for (k = 0; k < r.bufferpos; k++)
						h = (h ^ r.buffer[k]) * 1099511628211ull;
				}
			}
		}
		gnuais_protodec_flush(&r);
		fprintf(out, "R: received %d lost %d lost2 %d state %d inside %d cells %016llx\n", r.receivedframes, r.lostframes,
			r.lostframes2, r.state, inside, h);
		protodec_deinit(&r);
	}
	{       /* the CRC names: "123456789" and a frame with its FCS appended */
		unsigned char msg[9] = "123456789", body[8] = { 0x04, 0x43, 0x12, 0x34, 0x56, 0x78, 0x9a, 0xbc };
		unsigned short fcs;
		int k, j;
		fprintf(out, "crc %04x empty %04x\n", protodec_sdlc_crc(msg, 9), protodec_sdlc_crc(msg, 0));
		fcs = protodec_sdlc_crc(body, 6);
		body[6] = (unsigned char) (fcs & 0xff);
		body[7] = (unsigned char) (fcs >> 8);
		memset(d.buffer, 0, DEMOD_BUFFER_LEN);
		for (j = 0; j < 8; j++)
			for (k = 0; k < 8; k++)
				d.buffer[8 * j + k] = (body[j] >> k) & 1;       /* least significant bit first on the air */
		fprintf(out, "calculate_crc good %d", protodec_calculate_crc(48, &d));
		for (j = 0; j < 48; j++)
			fprintf(out, "%s%d", j % 8 ? "" : " ", d.rbuffer[j]);
		d.buffer[5] ^= 1;
		fprintf(out, "\ncalculate_crc bad %d nonpositive %d %d huge %d\n", protodec_calculate_crc(48, &d),
			protodec_calculate_crc(0, &d), protodec_calculate_crc(-8, &d), protodec_calculate_crc(100000, &d));
		fprintf(out, "calculate_crc long %d\n", protodec_calculate_crc(448, &d));
	}
	free(d.buffer);
	free(d.rbuffer);
	free(e.buffer);
	free(e.rbuffer);
	fclose(out);
	free(y);
	free(bits);
	free(x);
}


/* ------------------------------------------------------------ 5b. the shims from two threads at once */

/* Each thread owns a filter and decoders of its own (distinct objects, as the reference's threading rules allow) and
 * drives them the way receiver.c does: protodec_decode() bit by bit, a filter_run_buf() per buffer -- which flushes
 * EVERY decoder of the process, the other thread's included -- and, every so often, gnuais_protodec_release() of a
 * scratch decoder while the other thread's flush may be walking the table.  A's frames must be the golden ones. */
struct shim_job {
	const unsigned char *bits;
	size_t nbits;
	char id;
	int received, lost, lost2;
};

static void *drive_shims(void *arg)
{
	struct shim_job *j = arg;
	struct demod_state_t d, *scratch = NULL;
	float taps[36], o;
	short s = 0;
	size_t i;
	struct filter *f;

	gnuais_default_taps(taps);
	f = filter_init(36, taps);
	protodec_initialize(&d, NULL, NULL, j->id);
	for (i = 0; i < j->nbits; i++) {
		char b = (char) j->bits[i];
		protodec_decode(&b, 1, &d);
		if (i % 7 == 0) {                               /* a short-lived decoder: created, fed, released */
			if (!scratch) {
				scratch = malloc(sizeof *scratch);
				protodec_initialize(scratch, NULL, NULL, (char) (j->id + 1));
			}
			protodec_decode(&b, 1, scratch);
		}
		if (i % 204 == 203)
			filter_run_buf(f, &s, &o, 1, 1);        /* flush_all(): every decoder, both threads' */
		if (i % 1021 == 1020 && scratch) {
			gnuais_protodec_release(scratch);       /* while the other thread may be inside flush_all() */
			free(scratch->buffer);
			free(scratch->rbuffer);
			free(scratch);
			scratch = NULL;
		}
	}
	gnuais_protodec_flush(&d);
	j->received = d.receivedframes;
	j->lost = d.lostframes;
	j->lost2 = d.lostframes2;
	gnuais_protodec_release(&d);
	if (scratch) {
		gnuais_protodec_release(scratch);
		free(scratch->buffer);
		free(scratch->rbuffer);
		free(scratch);
	}
	free(d.buffer);
	free(d.rbuffer);
	filter_free(f);
	return NULL;
}

static void run_shims_mt(const char *dir)
{
	size_t nbits;
	unsigned char *bits = slurp(in_dir(dir, "bits_a.bin"), &nbits);
	struct shim_job a = { bits, nbits, 'A', 0, 0, 0 }, p = { bits, nbits, 'P', 0, 0, 0 };
	pthread_t t1, t2;
	FILE *out = fopen(in_dir(dir, "out_shims_mt.txt"), "w");

	frames_out = out;
	pthread_create(&t1, NULL, drive_shims, &a);
	pthread_create(&t2, NULL, drive_shims, &p);
	pthread_join(t1, NULL);
	pthread_join(t2, NULL);
	fprintf(out, "A: received %d lost %d lost2 %d\n", a.received, a.lost, a.lost2);
	fprintf(out, "P: received %d lost %d lost2 %d\n", p.received, p.lost, p.lost2);
	fclose(out);
	free(bits);
}

/* ------------------------------------------------------------ 6. a buffer mismatch inside a round is refused */

static struct receiver *mm_a, *mm_b;

static void mismatch_handler(const char *message)
{
	/* a host closes things down from its handler: free_receiver() takes the drop-in's lock again, which die()
	 * still holds -- it must not deadlock */
	fprintf(stderr, "handler: %s\n", message);
	free_receiver(mm_a);
	free_receiver(mm_b);
	fprintf(stderr, "handler: receivers freed\n");
	exit(7);
}

static int run_mismatch(void)
{
	static short one[2 * 100], other[2 * 100];
	gnuais_receiver_on_fatal(mismatch_handler);
	frames_out = stderr;
	mm_a = init_receiver('A', 2, 0, NULL, NULL);
	mm_b = init_receiver('B', 2, 1, NULL, NULL);
	receiver_run(mm_a, one, 100);
	receiver_run(mm_b, one, 100);           /* a complete round */
	receiver_run(mm_b, one, 100);           /* B starts the next one (same address, as ais.c reuses its buffer) */
	receiver_run(mm_a, other, 100);         /* A, not served from it yet, brings another buffer: refused */
	return 0;                               /* not reached */
}

int main(int argc, char **argv)
{
	if (argc > 2 && !strcmp(argv[2], "mismatch"))
		return run_mismatch();
	if (argc < 2) {
		fprintf(stderr, "usage: %s dir\n", argv[0]);
		return 2;
	}
	run_wav(argv[1]);
	run_messages_and_sinks(argv[1]);
	run_dropin(argv[1]);
	run_shims(argv[1]);
	run_shims_mt(argv[1]);
	printf("asan_host: done\n");
	return 0;
}
