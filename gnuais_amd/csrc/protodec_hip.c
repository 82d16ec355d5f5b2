/*
 * protodec_hip.c -- the reference's OTHER public names of the hot path, on top of libgnuais_hip.so:
 *
 *   src/filter.h:64-68     filter_init  filter_free  filter_run  filter_run_buf
 *   src/protodec.h:76      protodec_decode            (src/protodec.c:988-1122, the HDLC deframer)
 *   src/protodec.h:74      protodec_reset             (src/protodec.c:87-100) -- reaches the machine on the device
 *   src/protodec.c:78      protodec_deinit            -- flushes, releases the device objects, frees d's buffers
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
 * with the five names renamed (-Dprotodec_decode=gnuais_ref_protodec_decode ... -Dprotodec_reset=gnuais_ref_protodec_reset
 * -Dprotodec_deinit=gnuais_ref_protodec_deinit: nothing of its source changes; its protodec_initialize() then calls
 * the renamed reset, which is right -- no device object exists before the first bit) and leave filter.c out;
 * INTEGRATION.md has the link line, and the test suite builds and runs exactly that.
 *
 * protodec_decode() semantics.  The reference consumes `count` bits and returns with d current; its receiver.c
 * calls it once per BIT (receiver.c:130), and a device round trip per bit is unusable at any rate.  So bits queue
 * per decoder and go to the device in chunks: when GNUAIS_PROTODEC_CHUNK (2048) bits wait, and -- what keeps the
 * observable order the reference's -- at the start of every filter_run_buf(), i.e. whenever receiver_run() begins
 * its next buffer: every decoder with queued bits is flushed in creation order, so the protodec_getdata() calls of
 * one buffer come out receiver by receiver, in time order inside a receiver, exactly as the per-bit path makes them
 * (ais.c:237-247: A's frames of the buffer, then B's).  d's public fields (state, nstartsign, antallpreamble,
 * antallenner, bitstuff, last, bufferpos, receivedframes, lostframes, lostframes2) are current as of the last flush:
 * at most one buffer behind; gnuais_protodec_flush(d) (NULL: every decoder) brings them up to the last bit -- call
 * it before reading the counters at shutdown (ais.c:296-310).  gnuais_protodec_set_batching(1) restores the strict
 * per-call behaviour (every call returns with d current; slow).  d->buffer (protodec.h:52: the stored bits of the
 * frame in progress, or of the last frame that reached its stop bit) is refreshed at every flush as well
 * (gnuais_batch_frame_bits); the one case it is left as it was: a frame given up at 449 bits.
 *
 * Threads: the two tables are hashed by object address and guarded by one mutex, which is NOT held while the
 * reference's protodec_getdata() runs (it writes to the serial port, takes the cache's lock and flushes stdout);
 * one decoder / filter object must not be driven from two threads at once, as in the reference.  No CPU fallback:
 * a HIP failure aborts, as the reference does on its own fatal errors.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

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
#ifndef ST_SKURR
#define ST_SKURR 1                      /* src/protodec.h:30 */
#endif
#include "gnuais_hip.h"

#define GNUAIS_PROTODEC_CHUNK 2048      /* bits that may wait per decoder before they go to the device */
#define QUEUE_BITS 8192                 /* what one gnuais_batch_decode_bits() call of these batches takes */

static void die(const char *what)
{
	fprintf(stderr, "gnuais-hip: %s: %s\n", what, gnuais_last_error());
	abort();
}

/* ------------------------------------------------- object address -> device object (hashed, one lock) */

struct f_ent {
	struct filter *f;
	gnuais_batch *b;        /* one channel, the filter's taps */
	int16_t *in;
	int cap;
};
struct d_ent {
	struct demod_state_t *d;
	gnuais_batch *b;        /* one channel: only its deframer + CRC stages are used */
	pthread_mutex_t lock;   /* the queue and d's mirror; held while this decoder's frames go to protodec_getdata() */
	unsigned char *bits;
	int n_bits;
	gnuais_frame *frames;
	int cap_frames;
	int refs;               /* under tab_lock: 1 for the table + 1 per thread that holds the entry outside tab_lock;
	                         * whoever drops the last one frees the entry (d_unref) */
	int released;           /* under e->lock: gnuais_protodec_release() has flushed it; d may be gone, nothing touches it */
};

/* open addressing, linear probing, backward-shift deletion; the entries themselves never move */
struct ptab {
	const void **key;
	void **ent;
	unsigned cap, used;
};
static struct ptab f_tab, d_tab;
static struct d_ent **d_order;          /* decoders in creation order: the order flush-all serves them in */
static int n_order, cap_order;
static int batching = GNUAIS_PROTODEC_CHUNK;
static pthread_mutex_t tab_lock = PTHREAD_MUTEX_INITIALIZER;

static unsigned slot_of(const struct ptab *t, const void *key)
{
	unsigned long long h = (unsigned long long) (size_t) key;
	h ^= h >> 33;
	h *= 0xff51afd7ed558ccdull;
	h ^= h >> 29;
	return (unsigned) h & (t->cap - 1);
}

static void *ptab_get(const struct ptab *t, const void *key)
{
	unsigned i;
	if (!t->cap)
		return NULL;
	for (i = slot_of(t, key); t->key[i]; i = (i + 1) & (t->cap - 1))
		if (t->key[i] == key)
			return t->ent[i];
	return NULL;
}

static void ptab_put(struct ptab *t, const void *key, void *ent)
{
	unsigned i;
	if (4 * (t->used + 1) > 3 * t->cap) {           /* grow: rehash into twice the slots */
		struct ptab n;
		n.cap = t->cap ? 2 * t->cap : 16;
		n.used = 0;
		n.key = calloc(n.cap, sizeof(*n.key));
		n.ent = calloc(n.cap, sizeof(*n.ent));
		if (!n.key || !n.ent)
			abort();
		for (i = 0; i < t->cap; i++)
			if (t->key[i])
				ptab_put(&n, t->key[i], t->ent[i]);
		free(t->key);
		free(t->ent);
		*t = n;
	}
	for (i = slot_of(t, key); t->key[i]; i = (i + 1) & (t->cap - 1))
		;
	t->key[i] = key;
	t->ent[i] = ent;
	t->used++;
}

static void ptab_del(struct ptab *t, const void *key)
{
	unsigned i, j, k;
	if (!t->cap)
		return;
	for (i = slot_of(t, key); t->key[i] != key; i = (i + 1) & (t->cap - 1))
		if (!t->key[i])
			return;
	for (j = i;;) {                                 /* close the gap so that every probe sequence stays unbroken */
		j = (j + 1) & (t->cap - 1);
		if (!t->key[j])
			break;
		k = slot_of(t, t->key[j]);
		if ((i <= j) ? (i < k && k <= j) : (i < k || k <= j))
			continue;
		t->key[i] = t->key[j];
		t->ent[i] = t->ent[j];
		i = j;
	}
	t->key[i] = NULL;
	t->ent[i] = NULL;
	t->used--;
}

static struct f_ent *f_of(struct filter *f)
{
	struct f_ent *e;
	pthread_mutex_lock(&tab_lock);
	e = ptab_get(&f_tab, f);
	pthread_mutex_unlock(&tab_lock);
	return e;
}

static void flush_all(void);

/* ---------------------------------------------------------------- src/filter.h:64-68 */

/* src/filter.c:57-71: the object the caller sees, with its `length` taps; the sample window is on the device */
struct filter *filter_init(int len, float *taps)
{
	struct filter *f;
	struct f_ent *e;

	if (len <= 0 || len >= BufferLen) {
		fprintf(stderr, "gnuais-hip: filter_init: %d taps (1..%d)\n", len, BufferLen - 1);
		abort();
	}
	f = hmalloc(sizeof *f);
	memset(f, 0, sizeof *f);
	f->length = len;
	f->pointer = len;                       /* for whoever looks: the reference starts its ring here */
	f->taps = hmalloc((size_t) len * sizeof(float));
	memcpy(f->taps, taps, (size_t) len * sizeof(float));
	e = calloc(1, sizeof *e);
	if (!e)
		abort();
	e->f = f;
	if (gnuais_batch_create(&e->b, 0, 1, taps, len, 0, 4096, 0) != GNUAIS_OK)
		die("filter_init: gnuais_batch_create");
	pthread_mutex_lock(&tab_lock);
	ptab_put(&f_tab, f, e);
	pthread_mutex_unlock(&tab_lock);
	return f;
}

/* src/filter.c:73-79 */
void filter_free(struct filter *f)
{
	struct f_ent *e;
	if (!f)
		return;
	pthread_mutex_lock(&tab_lock);
	e = ptab_get(&f_tab, f);
	if (e)
		ptab_del(&f_tab, f);
	pthread_mutex_unlock(&tab_lock);
	if (e) {
		gnuais_batch_destroy(e->b);
		free(e->in);
		free(e);
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
	/* receiver_run() starts every buffer here (receiver.c:107): what the decoders still hold of the buffer
	 * before goes to the device now, receiver by receiver */
	flush_all();
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

/* src/protodec.c:120-167, as ONE device call (gnuais_crc16_bits): the frame's cells d->buffer[0 .. 8 * (bytes + 2))
 * are packed least-significant-bit first and run through the CRC there, and the payload comes back most-significant
 * -bit first, which is what the message layer reads from d->rbuffer.  Verdict: the register equals 0x0f47.
 * Two corners: a length that is not positive is refused (protodec.c:128-131); a payload that does not fit
 * d->rbuffer fills it and is refused, as the reference's loop does when it reaches the end of the array.  The
 * reference would read d->buffer past its DEMOD_BUFFER_LEN cells for frames that long (its deframer never produces
 * them, protodec.c:1024); here those cells count as 0. */
int protodec_calculate_crc(int length_bits, struct demod_state_t *d)
{
	unsigned char cells[8 * 64];
	uint16_t reg = 0;
	const int payload = length_bits / 8, framed = payload + 2;
	int have, shown;

	if (length_bits <= 0)
		return 0;
	memset(d->rbuffer, 0, DEMOD_BUFFER_LEN);
	if (framed > 64)                                /* 496 payload bits: far beyond d->rbuffer, nothing to show */
		return 0;
	have = 8 * framed < DEMOD_BUFFER_LEN ? 8 * framed : DEMOD_BUFFER_LEN;
	memset(cells, 0, sizeof cells);
	memcpy(cells, d->buffer, (size_t) have);
	shown = 8 * payload < DEMOD_BUFFER_LEN ? 8 * payload : DEMOD_BUFFER_LEN;
	if (gnuais_crc16_bits(0, cells, framed, &reg, (uint8_t *) d->rbuffer, shown) != GNUAIS_OK)
		die("protodec_calculate_crc");
	if (8 * payload > DEMOD_BUFFER_LEN)
		return 0;
	return reg == 0x0f47;
}

/* ---------------------------------------------------------------- src/protodec.c:988-1122 */

static struct d_ent *d_of(struct demod_state_t *d)
{
	struct d_ent *e;
	pthread_mutex_lock(&tab_lock);
	e = ptab_get(&d_tab, d);
	if (!e) {
		e = calloc(1, sizeof *e);
		if (!e)
			abort();
		e->d = d;
		pthread_mutex_init(&e->lock, NULL);
		if (gnuais_batch_create(&e->b, 0, 1, NULL, 0, 0, 4096, 4096) != GNUAIS_OK)
			die("protodec_decode: gnuais_batch_create");
		e->bits = malloc(QUEUE_BITS);
		e->cap_frames = 256;
		e->frames = malloc(sizeof(gnuais_frame) * (size_t) e->cap_frames);
		if (!e->bits || !e->frames)
			abort();
		ptab_put(&d_tab, d, e);
		if (n_order == cap_order) {
			cap_order = cap_order ? 2 * cap_order : 8;
			d_order = realloc(d_order, sizeof(*d_order) * (size_t) cap_order);
			if (!d_order)
				abort();
		}
		d_order[n_order++] = e;
		e->refs = 1;                            /* the table's */
	}
	e->refs++;                                      /* the caller's: d_unref() when it is done with e */
	pthread_mutex_unlock(&tab_lock);
	return e;
}

/* drops a reference taken under tab_lock (d_of, flush_all's snapshot, gnuais_protodec_flush); the last one -- only
 * after gnuais_protodec_release() has unlinked the entry and dropped the table's -- frees it */
static void d_unref(struct d_ent *e)
{
	int last;
	pthread_mutex_lock(&tab_lock);
	last = --e->refs == 0;
	pthread_mutex_unlock(&tab_lock);
	if (!last)
		return;
	pthread_mutex_destroy(&e->lock);
	gnuais_batch_destroy(e->b);
	free(e->bits);
	free(e->frames);
	free(e);
}

/* e->lock held: the queued bits through the device deframer, the frames it closed to protodec_getdata() in time
 * order (protodec.c:1100-1104), then d's public fields from the device */
static void flush_locked(struct d_ent *e)
{
	struct demod_state_t *d = e->d;
	gnuais_counters c;
	gnuais_fsm_state st;
	int32_t n = e->n_bits;
	int got = 0, pending = 0, i, j, k;

	if (n == 0 || e->released)
		return;
	e->n_bits = 0;
	if (gnuais_batch_decode_bits(e->b, e->bits, n, &n) != GNUAIS_OK)
		die("protodec_decode: gnuais_batch_decode_bits");
	if (gnuais_batch_pending_frames(e->b, &pending) != GNUAIS_OK)
		die("protodec_decode: gnuais_batch_pending_frames");
	if (pending > e->cap_frames) {
		e->cap_frames = pending * 2;
		e->frames = realloc(e->frames, sizeof(gnuais_frame) * (size_t) e->cap_frames);
		if (!e->frames)
			abort();
	}
	if (gnuais_batch_drain_frames(e->b, e->frames, e->cap_frames, &got) != GNUAIS_OK)
		die("protodec_decode: gnuais_batch_drain_frames");
	for (i = 0; i < got; i++) {
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
	if (d->buffer) {                                /* protodec.c:1019: the cells a host may look at */
		unsigned char cells[DEMOD_BUFFER_LEN];
		int nb = -1;
		if (gnuais_batch_frame_bits(e->b, 0, cells, DEMOD_BUFFER_LEN, &nb) != GNUAIS_OK)
			die("protodec_decode: gnuais_batch_frame_bits");
		if (nb >= 0) {
			if (nb > DEMOD_BUFFER_LEN)
				nb = DEMOD_BUFFER_LEN;
			memset(d->buffer, 0, DEMOD_BUFFER_LEN);         /* protodec.c:1080 at the frame's start */
			memcpy(d->buffer, cells, (size_t) nb);
		}
	}
}

/* every decoder with queued bits, in creation order.  The order is SNAPSHOT under the table lock with a reference
 * on every entry, so that a gnuais_protodec_release() on another thread can neither free an entry between the look-up
 * and its lock nor shift the indices under the loop; the table lock is not held while a decoder is served. */
static void flush_all(void)
{
	struct d_ent *few[16], **snap = few;
	int i, n;
	pthread_mutex_lock(&tab_lock);
	n = n_order;
	if (n > 16) {
		snap = malloc(sizeof(*snap) * (size_t) n);
		if (!snap)
			abort();
	}
	for (i = 0; i < n; i++) {
		snap[i] = d_order[i];
		snap[i]->refs++;
	}
	pthread_mutex_unlock(&tab_lock);
	for (i = 0; i < n; i++) {
		pthread_mutex_lock(&snap[i]->lock);
		flush_locked(snap[i]);
		pthread_mutex_unlock(&snap[i]->lock);
		d_unref(snap[i]);
	}
	if (snap != few)
		free(snap);
}

void protodec_decode(char *in, int count, struct demod_state_t *d)
{
	struct d_ent *e = d_of(d);
	int i;
	pthread_mutex_lock(&e->lock);
	for (i = 0; i < count; i++) {
		if (e->n_bits == QUEUE_BITS)
			flush_locked(e);
		e->bits[e->n_bits++] = in[i] ? 1 : 0;
	}
	if (e->n_bits >= batching)
		flush_locked(e);
	pthread_mutex_unlock(&e->lock);
	d_unref(e);
}

/* src/protodec.c:87-100.  The bits d was given before this call are decoded first (with the state they met), then the
 * machine on the device and d's own fields go back to ST_SKURR; counters stay.  A d that has not seen a bit yet has no
 * device object: only its fields are set, which is all the reference does. */
void protodec_reset(struct demod_state_t *d)
{
	struct d_ent *e;
	pthread_mutex_lock(&tab_lock);
	e = ptab_get(&d_tab, d);
	if (e)
		e->refs++;
	pthread_mutex_unlock(&tab_lock);
	if (e) {
		pthread_mutex_lock(&e->lock);
		flush_locked(e);
		if (!e->released && gnuais_batch_protodec_reset(e->b) != GNUAIS_OK)
			die("protodec_reset");
		pthread_mutex_unlock(&e->lock);
		d_unref(e);
	}
	d->state = ST_SKURR;
	d->nskurr = 0;
	d->ndata = 0;
	d->npreamble = 0;
	d->nstartsign = 0;
	d->nstopsign = 0;
	d->antallpreamble = 0;
	d->antallenner = 0;
	d->last = 0;
	d->bitstuff = 0;
	d->bufferpos = 0;
}

void gnuais_protodec_release(struct demod_state_t *d);

/* src/protodec.c:78-85, after whatever d still holds has been decoded and delivered and its device objects are gone */
void protodec_deinit(struct demod_state_t *d)
{
	gnuais_protodec_release(d);
	hfree(d->buffer);
	hfree(d->rbuffer);
	hfree(d->serbuffer);
	hfree(d->ipcbuffer);
	hfree(d->nmea);
}

/* additive, not reference names: how many bits may wait per decoder (1: every call returns with d current;
 * default GNUAIS_PROTODEC_CHUNK) */
void gnuais_protodec_set_batching(int bits)
{
	batching = bits < 1 ? 1 : (bits > QUEUE_BITS ? QUEUE_BITS : bits);
}

void gnuais_protodec_flush(struct demod_state_t *d)
{
	struct d_ent *e;
	if (!d) {
		flush_all();
		return;
	}
	pthread_mutex_lock(&tab_lock);
	e = ptab_get(&d_tab, d);
	if (e)
		e->refs++;
	pthread_mutex_unlock(&tab_lock);
	if (e) {
		pthread_mutex_lock(&e->lock);
		flush_locked(e);
		pthread_mutex_unlock(&e->lock);
		d_unref(e);
	}
}

/* releases the device objects of a decoder the caller is done with (the reference itself never frees one) */
void gnuais_protodec_release(struct demod_state_t *d)
{
	struct d_ent *e;
	int i;
	pthread_mutex_lock(&tab_lock);
	e = ptab_get(&d_tab, d);
	if (e) {
		ptab_del(&d_tab, d);
		for (i = 0; i < n_order && d_order[i] != e; i++)
			;
		if (i < n_order)
			memmove(&d_order[i], &d_order[i + 1], sizeof(*d_order) * (size_t) (n_order - 1 - i));
		n_order--;
	}
	pthread_mutex_unlock(&tab_lock);
	if (!e)
		return;
	/* unlinked: no new reference can be taken.  Flush while d is still the caller's, mark the entry so that a
	 * flush_all() that took its reference before the unlink leaves d alone, and drop the table's reference; the
	 * entry goes when the last holder lets go of it. */
	pthread_mutex_lock(&e->lock);
	flush_locked(e);
	e->released = 1;
	pthread_mutex_unlock(&e->lock);
	d_unref(e);
}
