/*
 * protodec_hip.c -- the reference's OTHER public names of the hot path, on top of libgnuais_hip.so:
 *
 *   src/filter.h:64-68     filter_init  filter_free  filter_run  filter_run_buf
 *   src/protodec.h:76      protodec_decode            (src/protodec.c:988-1122, the HDLC deframer)
 *   src/protodec.c:106,120 protodec_sdlc_crc  protodec_calculate_crc
 *
 * This is the inverse of receiver_hip.c: there the reference's receiver.c / filter.c are replaced and its
 * protodec.c kept; here the reference's receiver.c stays UNMODIFIED -- its slicer / PLL / NRZI loop runs on the
 * host as it always did -- and what it calls is served by the device: filter_run_buf() is the exact FIR kernel
 * (gnuais_batch_filter: bit-identical floats), protodec_decode() feeds the device deframer + CRC kernels
 * (gnuais_batch_decode_bits) and hands every CRC-valid frame to the reference's own protodec_getdata().  Meant for
 * code that holds on to these names; the fast path is receiver_hip.c / gnuais_batch_run(), which never leaves
 * the device between the stages.
 *
 * Linking: the reference's protodec.c defines protodec_decode / protodec_calculate_crc / protodec_sdlc_crc next to
 * the message layer this file needs from it (protodec_initialize, protodec_getdata, ...).  Compile that one file
 * with the three names renamed (-Dprotodec_decode=ref_protodec_decode ...: nothing of its source changes) and leave
 * filter.c out; INTEGRATION.md has the link line, and the test suite builds and runs exactly that.
 *
 * protodec_decode() semantics.  The reference consumes `count` bits and returns with d current.  By default this
 * file does the same (every call is a device round trip: correct, slow -- receiver.c calls it once per bit).
 * gnuais_protodec_set_batching(n) lets up to n bits queue per decoder before they go to the device;
 * d's public fields (state, nstartsign, antallpreamble, antallenner, bitstuff, last, bufferpos, receivedframes,
 * lostframes, lostframes2) and the protodec_getdata() calls then happen at the flush -- the same calls in the same
 * order -- and gnuais_protodec_flush(d) forces one (NULL: every decoder).  d->buffer (the raw bits of a frame in
 * progress) is not mirrored.  Single caller thread, like the reference's main loop.  No CPU fallback: a HIP
 * failure aborts, as the reference does on its own fatal errors.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef GNUAIS_TREE
#include "filter.h"
#include "protodec.h"
#include "hmalloc.h"
#else
#include "gnuais_receiver_abi.h"
#define hmalloc malloc
#define hfree free
#define BufferLen 1024
struct filter {
	int length;
	float *taps;
	float buffer[BufferLen];
	int pointer;
};
void protodec_getdata(int bufferlengde, struct demod_state_t *d);
#endif
#include "gnuais_hip.h"

static void die(const char *what)
{
	fprintf(stderr, "gnuais-hip: %s: %s\n", what, gnuais_last_error());
	abort();
}

/* ---------------------------------------------------------------- side tables (pointer -> device object) */

struct f_ent {
	struct filter *f;
	gnuais_batch *b;        /* one channel, the filter's taps */
	int16_t *in;
	int cap;
};
struct d_ent {
	struct demod_state_t *d;
	gnuais_batch *b;        /* one channel: only its deframer + CRC stages are used */
	unsigned char *bits;
	int n_bits, cap_bits;
	gnuais_frame *frames;
	int cap_frames;
};
static struct f_ent *f_tab;
static struct d_ent *d_tab;
static int n_f, n_d, batching = 1;

static struct f_ent *f_of(struct filter *f)
{
	int i;
	for (i = 0; i < n_f; i++)
		if (f_tab[i].f == f)
			return &f_tab[i];
	return NULL;
}

/* ---------------------------------------------------------------- src/filter.h:64-68 */

/* src/filter.c:57-71 */
struct filter *filter_init(int len, float *taps)
{
	struct filter *f;
	struct f_ent *e;

	if (len <= 0 || len >= BufferLen) {
		fprintf(stderr, "gnuais-hip: filter_init: %d taps (1..%d)\n", len, BufferLen - 1);
		abort();
	}
	f = (struct filter *) hmalloc(sizeof(struct filter));
	memset(f, 0, sizeof(struct filter));
	f->taps = (float *) hmalloc(len * sizeof(float));
	memcpy(f->taps, taps, len * sizeof(float));
	f->length = len;
	f->pointer = f->length;         /* kept for whoever looks; the window itself lives on the device */
	f_tab = realloc(f_tab, sizeof(*f_tab) * (size_t) (n_f + 1));
	e = &f_tab[n_f++];
	memset(e, 0, sizeof(*e));
	e->f = f;
	if (gnuais_batch_create(&e->b, 0, 1, taps, len, 0, 4096, 0) != GNUAIS_OK)
		die("filter_init: gnuais_batch_create");
	return f;
}

/* src/filter.c:73-79 */
void filter_free(struct filter *f)
{
	struct f_ent *e;
	if (!f)
		return;
	e = f_of(f);
	if (e) {
		gnuais_batch_destroy(e->b);
		free(e->in);
		*e = f_tab[--n_f];
	}
	hfree(f->taps);
	hfree(f);
}

/* src/filter.c:106-143: in[0], in[step], ... -> out[0 .. len), returns the peak positive sample */
short filter_run_buf(struct filter *f, short *in, float *out, int step, int len)
{
	struct f_ent *e = f_of(f);
	int16_t peak = 0;
	int i, done = 0;

	if (!e) {
		fprintf(stderr, "gnuais-hip: filter_run_buf: not a filter_init() object\n");
		abort();
	}
	while (done < len) {                    /* the batch takes 4096 samples a call; the peak is a running maximum */
		const int n = len - done < 4096 ? len - done : 4096;
		int16_t m = 0;
		if (e->cap < n) {
			e->in = realloc(e->in, sizeof(int16_t) * (size_t) n);
			e->cap = n;
		}
		for (i = 0; i < n; i++)
			e->in[i] = in[(size_t) (done + i) * (size_t) step];
		if (gnuais_batch_filter_host(e->b, e->in, n, out + done) != GNUAIS_OK ||
		    gnuais_batch_maxval(e->b, &m) != GNUAIS_OK)
			die("filter_run_buf");
		if (m > peak)
			peak = m;
		done += n;
	}
	f->pointer = f->length + (int) (((long) (f->pointer - f->length) + len) % (BufferLen - f->length));
	return peak;
}

/* src/filter.c:83-104: one sample in, one out.  The window is the same as filter_run_buf()'s (the `length` samples
 * before this one), so an int16-valued input goes the same way; the reference never calls it with anything else --
 * it never calls it at all -- and a value the device's int16 path cannot represent is refused loudly */
void filter_run(struct filter *f, float in, float *out)
{
	short s = (short) in;
	if ((float) s != in) {
		fprintf(stderr, "gnuais-hip: filter_run: %g is not an int16 sample value\n", (double) in);
		abort();
	}
	(void) filter_run_buf(f, &s, out, 1, 1);
}

/* ---------------------------------------------------------------- src/protodec.c:106-167 */

unsigned short protodec_sdlc_crc(const unsigned char *data, unsigned len)
{
	uint16_t crc = 0;
	int32_t n = (int32_t) len;
	if (len == 0)
		return (unsigned short) ~0xffff;        /* no byte touches crc = 0xffff: ~crc */
	if (gnuais_crc16_batch(0, data, (int) len, &n, 1, &crc) != GNUAIS_OK)
		die("protodec_sdlc_crc");
	return crc;
}

int protodec_calculate_crc(int length_bits, struct demod_state_t *d)
{
	int length_bytes, buflen, i, j, x;
	unsigned char *buf;
	unsigned short crc;

	if (length_bits <= 0)                           /* protodec.c:128-131 (the reference logs and returns 0) */
		return 0;
	length_bytes = length_bits / 8;
	buflen = length_bytes + 2;
	buf = (unsigned char *) hmalloc(sizeof(*buf) * buflen);
	for (j = 0; j < buflen; j++) {                  /* protodec.c:138-143: bits LSB first */
		unsigned char tmp = 0;
		for (i = 0; i < 8; i++)
			tmp |= (unsigned char) (d->buffer[i + 8 * j] << i);
		buf[j] = tmp;
	}
	crc = protodec_sdlc_crc(buf, (unsigned) buflen);        /* on the device */
	memset(d->rbuffer, 0, DEMOD_BUFFER_LEN);        /* protodec.c:150-162: payload bits MSB first */
	for (j = 0; j < length_bytes; j++)
		for (i = 0; i < 8; i++) {
			x = j * 8 + i;
			if (x >= DEMOD_BUFFER_LEN) {
				hfree(buf);
				return 0;
			}
			d->rbuffer[x] = (buf[j] >> (7 - i)) & 1;
		}
	hfree(buf);
	return crc == 0x0f47;
}

/* ---------------------------------------------------------------- src/protodec.c:988-1122 */

static struct d_ent *d_of(struct demod_state_t *d)
{
	struct d_ent *e;
	int i;
	for (i = 0; i < n_d; i++)
		if (d_tab[i].d == d)
			return &d_tab[i];
	d_tab = realloc(d_tab, sizeof(*d_tab) * (size_t) (n_d + 1));
	e = &d_tab[n_d++];
	memset(e, 0, sizeof(*e));
	e->d = d;
	if (gnuais_batch_create(&e->b, 0, 1, NULL, 0, 0, 4096, 4096) != GNUAIS_OK)
		die("protodec_decode: gnuais_batch_create");
	e->cap_bits = 8192;
	e->bits = malloc((size_t) e->cap_bits);
	e->cap_frames = 256;
	e->frames = malloc(sizeof(gnuais_frame) * (size_t) e->cap_frames);
	return e;
}

static void flush_one(struct d_ent *e)
{
	struct demod_state_t *d = e->d;
	gnuais_counters c;
	gnuais_fsm_state st;
	int32_t n = e->n_bits;
	int got = 0, pending = 0, i, j, k;

	if (n == 0)
		return;
	e->n_bits = 0;
	if (gnuais_batch_decode_bits(e->b, e->bits, n, &n) != GNUAIS_OK)
		die("protodec_decode: gnuais_batch_decode_bits");
	if (gnuais_batch_pending_frames(e->b, &pending) != GNUAIS_OK)
		die("protodec_decode: gnuais_batch_pending_frames");
	if (pending > e->cap_frames) {
		e->cap_frames = pending * 2;
		e->frames = realloc(e->frames, sizeof(gnuais_frame) * (size_t) e->cap_frames);
	}
	if (gnuais_batch_drain_frames(e->b, e->frames, e->cap_frames, &got) != GNUAIS_OK)
		die("protodec_decode: gnuais_batch_drain_frames");
	for (i = 0; i < got; i++) {                     /* time order; protodec.c:1100-1104 */
		const gnuais_frame *f = &e->frames[i];
		const int nbytes = f->nbits / 8;
		memset(d->rbuffer, 0, DEMOD_BUFFER_LEN);
		for (j = 0; j < nbytes; j++)
			for (k = 0; k < 8; k++)
				d->rbuffer[8 * j + k] = (f->payload[j] >> (7 - k)) & 1;
		d->receivedframes++;
		protodec_getdata(f->nbits, d);
	}
	if (gnuais_batch_counters(e->b, &c) != GNUAIS_OK || gnuais_batch_fsm_state(e->b, &st) != GNUAIS_OK)
		die("protodec_decode: state readback");
	d->receivedframes = c.receivedframes;
	d->lostframes = c.lostframes;
	d->lostframes2 = c.lostframes2;
	d->state = st.state;
	d->nstartsign = st.nstartsign;
	d->antallpreamble = st.antallpreamble;
	d->antallenner = st.antallenner;
	d->bitstuff = st.bitstuff;
	d->last = (char) st.last;
	d->bufferpos = st.bufferpos;
}

void protodec_decode(char *in, int count, struct demod_state_t *d)
{
	struct d_ent *e = d_of(d);
	int i;
	for (i = 0; i < count; i++) {
		if (e->n_bits == e->cap_bits)
			flush_one(e);
		e->bits[e->n_bits++] = in[i] ? 1 : 0;
	}
	if (e->n_bits >= batching)
		flush_one(e);
}

/* additive, not reference names: how many bits may wait per decoder (1: every call returns with d current) */
void gnuais_protodec_set_batching(int bits)
{
	batching = bits < 1 ? 1 : (bits > 8192 ? 8192 : bits);
}

void gnuais_protodec_flush(struct demod_state_t *d)
{
	int i;
	for (i = 0; i < n_d; i++)
		if (!d || d_tab[i].d == d)
			flush_one(&d_tab[i]);
}

/* releases the device objects of a decoder the caller is done with (the reference itself never frees one) */
void gnuais_protodec_release(struct demod_state_t *d)
{
	int i;
	for (i = 0; i < n_d; i++)
		if (d_tab[i].d == d) {
			flush_one(&d_tab[i]);
			gnuais_batch_destroy(d_tab[i].b);
			free(d_tab[i].bits);
			free(d_tab[i].frames);
			d_tab[i] = d_tab[--n_d];
			return;
		}
}
