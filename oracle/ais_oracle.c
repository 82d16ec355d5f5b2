/*
 * ais_oracle.c -- CPU restatement of the gnuais receive chain (the ORACLE).
 *
 * TEST INFRASTRUCTURE ONLY (see ais_oracle.h).  Parity status: PINNED against
 * oracle/_ref/libgnuais_ref.so (the real reference compiled in place) by
 * tests/test_oracle_vs_ref.py and against tests/golden/.
 *
 * Each function cites the reference lines it restates.  Build with
 * -ffp-contract=off: the reference object code is scalar mulss/addss, one
 * rounding per multiply and one per add, strictly in tap order.
 */
#include "ais_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

struct chan {
	uint32_t pll;          /* receiver.h:40 */
	int32_t prev;          /* receiver.h:44 */
	int32_t lastbit;       /* receiver.h:38 */
};

struct ais_oracle {
	int n_ch, n_taps;
	uint32_t pllinc;
	float *taps;
	int16_t *hist;         /* [n_ch][n_taps] the last n_taps input samples, oldest
	                          first: the only live part of filter.h:60 buffer[]  */
	struct chan *ch;
	ais_hdlc *hd;
	ais_frame *frames;
	size_t n_frames, cap_frames;
	pthread_mutex_t frame_lock;
};

/* ------------------------------------------------------------------ */

/* receiver.c:39-49: the coefficient table is symmetric; the double literals
 * round to fp32 on assignment (k = 0,1,34,35 -> 0.0f; k = 2,33 -> the
 * subnormal 0x00000069). */
static const double tap_half[18] = {
	2.5959e-55, 2.9479e-49, 1.4741e-43, 3.2462e-38, 3.1480e-33, 1.3443e-28,
	2.5280e-24, 2.0934e-20, 7.6339e-17, 1.2259e-13, 8.6690e-11, 2.6996e-08,
	3.7020e-06, 2.2355e-04, 5.9448e-03, 6.9616e-02, 3.5899e-01, 8.1522e-01
};

int ais_oracle_default_taps(float *out36)
{
	int k;
	for (k = 0; k < 18; k++) {
		out36[k] = (float) tap_half[k];
		out36[35 - k] = (float) tap_half[k];
	}
	return 36;
}

/* ------------------------------------------------------------------ */
/* protodec.c:87-100 protodec_reset: counters and the bit buffer survive */
static void hdlc_reset(ais_hdlc *h)
{
	h->state = AIS_ST_SKURR;
	h->nstartsign = 0;
	h->antallpreamble = 0;
	h->antallenner = 0;
	h->last = 0;
	h->bitstuff = 0;
	h->bufferpos = 0;
}

/* protodec.c:54-76 protodec_initialize */
static void hdlc_init(ais_hdlc *h)
{
	memset(h, 0, sizeof(*h));
	hdlc_reset(h);
}

/* protodec.c:106-118 protodec_sdlc_crc: reflected 0x8408, init 0xffff, bits
 * of each byte LSB first, final complement */
uint16_t ais_crc16_x25(const uint8_t *data, unsigned len)
{
	uint16_t crc = 0xffff;
	unsigned i, b;
	for (i = 0; i < len; i++) {
		unsigned v = data[i];
		for (b = 0; b < 8; b++) {
			unsigned fb = (crc ^ (v >> b)) & 1u;
			crc >>= 1;
			if (fb)
				crc ^= 0x8408;
		}
	}
	return (uint16_t) ~crc;
}

/* protodec.c:120-167 protodec_calculate_crc: nbits/8 (truncating) payload
 * bytes + 2 FCS bytes, buffer bits packed LSB first; good frame <=> 0x0f47.
 * On success the on-air bytes are left in bytes_out (nbits/8 of them). */
static int hdlc_check_crc(const ais_hdlc *h, int nbits, uint8_t *bytes_out)
{
	uint8_t buf[AIS_DEMOD_BUFFER_LEN / 8 + 4];
	int nbytes, buflen, j, i;

	if (nbits <= 0)
		return 0;
	nbytes = nbits / 8;
	buflen = nbytes + 2;
	for (j = 0; j < buflen; j++) {
		unsigned v = 0;
		for (i = 0; i < 8; i++)
			v |= (unsigned) h->buffer[8 * j + i] << i;
		buf[j] = (uint8_t) v;
	}
	if (ais_crc16_x25(buf, (unsigned) buflen) != 0x0f47)
		return 0;
	/* protodec.c:150-162 writes the same bytes MSB first into rbuffer */
	memcpy(bytes_out, buf, (size_t) (nbytes > 53 ? 53 : nbytes));
	return 1;
}

static void push_frame(ais_oracle *o, int ch, const ais_hdlc *h, int nbits, const uint8_t *bytes)
{
	ais_frame *f;
	pthread_mutex_lock(&o->frame_lock);
	if (o->n_frames == o->cap_frames) {
		o->cap_frames = o->cap_frames ? o->cap_frames * 2 : 1024;
		o->frames = realloc(o->frames, o->cap_frames * sizeof(*o->frames));
	}
	f = &o->frames[o->n_frames++];
	memset(f, 0, sizeof(*f));
	f->channel = (uint32_t) ch;
	f->end_bit = h->bits_seen;
	f->nbits = (uint16_t) nbits;
	f->flags = 1;
	memcpy(f->payload, bytes, (size_t) (nbits / 8 > 53 ? 53 : nbits / 8));
	pthread_mutex_unlock(&o->frame_lock);
}

/* protodec.c:988-1122 protodec_decode, one bit */
static void hdlc_step(ais_oracle *o, int ch, ais_hdlc *h, int x)
{
	switch (h->state) {
	case AIS_ST_DATA:                       /* protodec.c:995-1028 */
		if (h->bitstuff) {
			if (x == 1)
				h->state = AIS_ST_STOPSIGN;  /* sixth 1: closing flag */
			/* else: stuffed 0 dropped */
			h->bitstuff = 0;
		} else {
			if (x == h->last && x == 1) {
				if (++h->antallenner == 4) {
					h->bitstuff = 1;
					h->antallenner = 0;
				}
			} else {
				h->antallenner = 0;
			}
			h->buffer[h->bufferpos++] = (uint8_t) x;
			if (h->bufferpos >= 449)
				hdlc_reset(h);
		}
		break;

	case AIS_ST_SKURR:                      /* protodec.c:1030-1043 */
		if (x != h->last)
			h->antallpreamble++;
		else
			h->antallpreamble = 0;
		h->last = x;
		if (h->antallpreamble > 14 && x == 0) {
			h->state = AIS_ST_PREAMBLE;
			h->antallpreamble = 0;
		}
		break;

	case AIS_ST_PREAMBLE:                   /* protodec.c:1045-1072 */
		if (x != h->last && h->nstartsign == 0) {
			h->antallpreamble++;
		} else if (x == 1) {
			if (h->nstartsign == 0) {
				h->nstartsign = 3;
				h->last = x;
			} else if (h->nstartsign == 5) {
				h->nstartsign = 6;
				h->antallpreamble = 0;
				h->state = AIS_ST_STARTSIGN;
			} else {
				h->nstartsign++;
			}
		} else {
			if (h->nstartsign == 0)
				h->nstartsign = 1;
			else
				hdlc_reset(h);
		}
		break;

	case AIS_ST_STARTSIGN:                  /* protodec.c:1074-1093 */
		if (h->nstartsign >= 7) {
			if (x == 0) {
				h->state = AIS_ST_DATA;
				h->nstartsign = 0;
				h->antallenner = 0;
				memset(h->buffer, 0, AIS_DEMOD_BUFFER_LEN);
				h->bufferpos = 0;
			} else {
				hdlc_reset(h);
			}
		} else if (x == 0) {
			hdlc_reset(h);
		}
		h->nstartsign++;                /* also after a reset */
		break;

	case AIS_ST_STOPSIGN: {                 /* protodec.c:1095-1115 */
		int nbits = h->bufferpos - 6 - 16;
		if (x == 0 && nbits > 0) {
			uint8_t bytes[64];
			if (hdlc_check_crc(h, nbits, bytes)) {
				h->receivedframes++;
				push_frame(o, ch, h, nbits, bytes);
			} else {
				h->lostframes++;
			}
		} else {
			h->lostframes2++;
		}
		hdlc_reset(h);
		break;
	}
	}
	h->last = x;                            /* protodec.c:1119, every state */
	h->bits_seen++;
}

/* ------------------------------------------------------------------ */

ais_oracle *ais_oracle_create(int n_ch, const float *taps, int n_taps, unsigned pllinc)
{
	ais_oracle *o;
	if (n_ch <= 0 || n_taps <= 0 || n_taps > AIS_MAX_TAPS || !taps)
		return NULL;
	o = calloc(1, sizeof(*o));
	o->n_ch = n_ch;
	o->n_taps = n_taps;
	o->pllinc = pllinc ? pllinc : AIS_DEFAULT_PLLINC;
	o->taps = malloc(sizeof(float) * (size_t) n_taps);
	memcpy(o->taps, taps, sizeof(float) * (size_t) n_taps);
	o->hist = calloc((size_t) n_ch * (size_t) n_taps, sizeof(int16_t));
	o->ch = calloc((size_t) n_ch, sizeof(*o->ch));
	o->hd = calloc((size_t) n_ch, sizeof(*o->hd));
	pthread_mutex_init(&o->frame_lock, NULL);
	ais_oracle_reset(o);
	return o;
}

void ais_oracle_destroy(ais_oracle *o)
{
	if (!o)
		return;
	free(o->taps);
	free(o->hist);
	free(o->ch);
	free(o->hd);
	free(o->frames);
	pthread_mutex_destroy(&o->frame_lock);
	free(o);
}

/* receiver.c:52-74 init_receiver + filter.c:57-71 filter_init: zero history
 * (the first window is n_taps zeros), pll = 0, prev = 0, lastbit = 0 */
void ais_oracle_reset(ais_oracle *o)
{
	int c;
	memset(o->hist, 0, sizeof(int16_t) * (size_t) o->n_ch * (size_t) o->n_taps);
	memset(o->ch, 0, sizeof(*o->ch) * (size_t) o->n_ch);
	for (c = 0; c < o->n_ch; c++)
		hdlc_init(&o->hd[c]);
	o->n_frames = 0;
}

/* filter.c:106-143 filter_run_buf with filter.h:40-49 filter_mac: the window
 * of output n is the n_taps samples BEFORE sample n (the sample stored at
 * filter.c:115 is not part of its own window, filter.c:122), oldest first;
 * sum starts at +0.0f; maxval tracks positive samples only. `w` is scratch of
 * n_taps + len floats. */
static void fir_channel(const ais_oracle *o, int16_t *hist, const int16_t *in, int step,
			int len, float *out, int16_t *maxval_out, float *w)
{
	const int nt = o->n_taps;
	const float *taps = o->taps;
	int16_t maxval = 0;
	int i, k;

	for (k = 0; k < nt; k++)
		w[k] = (float) hist[k];
	for (i = 0; i < len; i++) {
		int16_t s = in[(size_t) i * (size_t) step];
		w[nt + i] = (float) s;
		if (s > maxval)
			maxval = s;
	}
	for (i = 0; i < len; i++) {
		const float *a = w + i;
		float sum = 0;
		for (k = 0; k < nt; k++)
			sum += a[k] * taps[k];
		out[i] = sum;
	}
	/* carry: the last n_taps samples seen (filter.c:129-134 ring wrap) */
	if (len >= nt) {
		for (k = 0; k < nt; k++)
			hist[k] = in[(size_t) (len - nt + k) * (size_t) step];
	} else {
		memmove(hist, hist + len, sizeof(int16_t) * (size_t) (nt - len));
		for (k = 0; k < len; k++)
			hist[nt - len + k] = in[(size_t) k * (size_t) step];
	}
	if (maxval_out)
		*maxval_out = maxval;
}

void ais_oracle_filter_channel(ais_oracle *o, int ch, const int16_t *in, int step, int len,
			       float *out, int16_t *maxval)
{
	float *w = malloc(sizeof(float) * ((size_t) o->n_taps + (size_t) len));
	fir_channel(o, o->hist + (size_t) ch * (size_t) o->n_taps, in, step, len, out, maxval, w);
	free(w);
}

/* receiver.c:109-135: slicer, PLL nudge on a zero crossing, phase advance,
 * on overflow slice + NRZI decode + one protodec step */
static uint32_t pll_channel(ais_oracle *o, int c, const float *filtered, int len,
			    uint8_t *bits, uint32_t bits_cap)
{
	struct chan *s = &o->ch[c];
	ais_hdlc *h = &o->hd[c];
	const uint32_t inc = o->pllinc, nudge = o->pllinc / 16;   /* receiver.c:84 INC */
	uint32_t nb = 0;
	int i;

	for (i = 0; i < len; i++) {
		int curr = filtered[i] > 0;
		if ((curr ^ s->prev) == 1) {
			if (s->pll < 0x10000 / 2)
				s->pll += nudge;
			else
				s->pll -= nudge;
		}
		s->prev = curr;
		s->pll += inc;
		if (s->pll > 0xffff) {
			int b = !(curr ^ s->lastbit);
			if (bits && nb < bits_cap)
				bits[nb] = (uint8_t) b;
			nb++;
			hdlc_step(o, c, h, b);
			s->lastbit = curr;
			s->pll &= 0xffff;
		}
	}
	return nb;
}

/* `planar` != 0: channel c's samples are contiguous at in + c*len (a de-interleaved copy of the
 * reference's layout; only the benchmark's "CPU best case" leg uses it), else interleaved with
 * stride n_ch as receiver_run() reads them (receiver.c:102,107). */
static int run_range(ais_oracle *o, const int16_t *in, int len, int ch0, int ch1, ais_run_out *out,
		     int planar)
{
	float *w, *filt;
	int c, i;

	if (len <= 0)
		return 0;
	w = malloc(sizeof(float) * ((size_t) o->n_taps + (size_t) len));
	filt = malloc(sizeof(float) * (size_t) len);
	for (c = ch0; c < ch1; c++) {
		int16_t mv;
		uint32_t nb;
		fir_channel(o, o->hist + (size_t) c * (size_t) o->n_taps,
			    planar ? in + (size_t) c * (size_t) len : in + c, planar ? 1 : o->n_ch, len,
			    filt, &mv, w);
		if (out && out->filtered)
			for (i = 0; i < len; i++)
				out->filtered[(size_t) i * (size_t) o->n_ch + (size_t) c] = filt[i];
		if (out && out->maxval)
			out->maxval[c] = mv;
		nb = pll_channel(o, c, filt, len,
				 (out && out->bits) ? out->bits + (size_t) c * out->bits_cap : NULL,
				 out ? out->bits_cap : 0);
		if (out && out->nbits)
			out->nbits[c] = nb;
	}
	free(w);
	free(filt);
	return 0;
}

int ais_oracle_run_range(ais_oracle *o, const int16_t *in, int len, int ch0, int ch1,
			 ais_run_out *out)
{
	return run_range(o, in, len, ch0, ch1, out, 0);
}

int ais_oracle_run(ais_oracle *o, const int16_t *in, int len, ais_run_out *out)
{
	return ais_oracle_run_range(o, in, len, 0, o->n_ch, out);
}

struct mt_job {
	ais_oracle *o;
	const int16_t *in;
	int len, ch0, ch1, planar;
};

static void *mt_main(void *p)
{
	struct mt_job *j = p;
	run_range(j->o, j->in, j->len, j->ch0, j->ch1, NULL, j->planar);
	return NULL;
}

static int run_mt(ais_oracle *o, const int16_t *in, int len, int n_threads, int planar);

int ais_oracle_run_mt(ais_oracle *o, const int16_t *in, int len, int n_threads)
{
	return run_mt(o, in, len, n_threads, 0);
}

int ais_oracle_run_planar_mt(ais_oracle *o, const int16_t *in_planar, int len, int n_threads)
{
	return run_mt(o, in_planar, len, n_threads, 1);
}

static int run_mt(ais_oracle *o, const int16_t *in, int len, int n_threads, int planar)
{
	pthread_t *th;
	struct mt_job *jobs;
	int t;

	if (n_threads < 1)
		n_threads = 1;
	if (n_threads > o->n_ch)
		n_threads = o->n_ch;
	th = malloc(sizeof(*th) * (size_t) n_threads);
	jobs = malloc(sizeof(*jobs) * (size_t) n_threads);
	for (t = 0; t < n_threads; t++) {
		jobs[t].o = o;
		jobs[t].in = in;
		jobs[t].len = len;
		jobs[t].planar = planar;
		jobs[t].ch0 = (int) ((long) o->n_ch * t / n_threads);
		jobs[t].ch1 = (int) ((long) o->n_ch * (t + 1) / n_threads);
		pthread_create(&th[t], NULL, mt_main, &jobs[t]);
	}
	for (t = 0; t < n_threads; t++)
		pthread_join(th[t], NULL);
	free(th);
	free(jobs);
	return 0;
}

void ais_oracle_decode_bits(ais_oracle *o, int ch, const uint8_t *bits, int n)
{
	int i;
	for (i = 0; i < n; i++)
		hdlc_step(o, ch, &o->hd[ch], bits[i] & 1);
}

/* ------------------------------------------------------------------ */

size_t ais_oracle_frame_count(const ais_oracle *o) { return o->n_frames; }
const ais_frame *ais_oracle_frames(const ais_oracle *o) { return o->frames; }
void ais_oracle_clear_frames(ais_oracle *o) { o->n_frames = 0; }
const ais_hdlc *ais_oracle_hdlc(const ais_oracle *o, int ch) { return &o->hd[ch]; }
/* protodec.c:87-100 called from outside (protodec_reset() is a public name, protodec.h:74) */
void ais_oracle_protodec_reset(ais_oracle *o, int ch) { hdlc_reset(&o->hd[ch]); }

static int frame_cmp(const void *a, const void *b)
{
	const ais_frame *x = a, *y = b;
	if (x->channel != y->channel)
		return x->channel < y->channel ? -1 : 1;
	if (x->end_bit != y->end_bit)
		return x->end_bit < y->end_bit ? -1 : 1;
	return 0;
}

/* reference print order within one chunk: channel 0..N-1, then time
 * (ais.c:237-247 calls receiver_run per channel) */
void ais_oracle_sort_frames(ais_oracle *o)
{
	if (o->n_frames > 1)
		qsort(o->frames, o->n_frames, sizeof(ais_frame), frame_cmp);
}

void ais_oracle_get_pll(const ais_oracle *o, int ch, uint32_t *pll, int *prev, int *lastbit)
{
	*pll = o->ch[ch].pll;
	*prev = o->ch[ch].prev;
	*lastbit = o->ch[ch].lastbit;
}

void ais_oracle_get_history(const ais_oracle *o, int ch, int16_t *out_n_taps)
{
	memcpy(out_n_taps, o->hist + (size_t) ch * (size_t) o->n_taps,
	       sizeof(int16_t) * (size_t) o->n_taps);
}
