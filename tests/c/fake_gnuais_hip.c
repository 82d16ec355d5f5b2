/*
 * fake_gnuais_hip.c -- TEST INFRASTRUCTURE.  A test double of the part of the C ABI (include/gnuais_hip.h) that the
 * host-side C of the drop-in calls -- gnuais_amd/csrc/receiver_hip.c and protodec_hip.c -- implemented on the CPU
 * oracle (oracle/ais_oracle.h).  It exists so that those two files, which end up inside someone else's daemon, can be
 * RUN under -fsanitize=address,undefined in the CPU suite (`make -C gnuais_amd/csrc asan`,
 * tests/test_sanitizers.py): their tables, locks, rounds, queues and buffer arithmetic do not care who computes the
 * frames.  Never linked into the product; the product library has no CPU path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnuais_hip.h"
#include "ais_oracle.h"

struct gnuais_batch {
	ais_oracle *o;
	int n_ch, max_len;
	int16_t *maxval;
};

static const char *last_error = "";

const char *gnuais_last_error(void) { return last_error; }

int gnuais_batch_create(gnuais_batch **out, int device, int n_channels, const float *taps, int n_taps,
			unsigned pllinc, int max_len, int frame_capacity)
{
	gnuais_batch *b;
	float def[36];
	(void) device;
	(void) frame_capacity;
	if (!out || n_channels < 1 || max_len < 1) {
		last_error = "fake create: argument";
		return GNUAIS_E_ARG;
	}
	if (!taps) {
		n_taps = ais_oracle_default_taps(def);
		taps = def;
	}
	b = calloc(1, sizeof *b);
	b->o = ais_oracle_create(n_channels, taps, n_taps, pllinc ? pllinc : AIS_DEFAULT_PLLINC);
	b->n_ch = n_channels;
	b->max_len = max_len;
	b->maxval = calloc((size_t) n_channels, sizeof *b->maxval);
	*out = b;
	return GNUAIS_OK;
}

void gnuais_batch_destroy(gnuais_batch *b)
{
	if (!b)
		return;
	ais_oracle_destroy(b->o);
	free(b->maxval);
	free(b);
}

int gnuais_batch_run_host(gnuais_batch *b, const int16_t *h_samples, int len)
{
	ais_run_out out;
	if (!b || !h_samples || len < 1 || len > b->max_len) {
		last_error = "fake run_host: argument";
		return GNUAIS_E_ARG;
	}
	memset(&out, 0, sizeof out);
	out.maxval = b->maxval;
	ais_oracle_run(b->o, h_samples, len, &out);
	return GNUAIS_OK;
}

int gnuais_batch_pending_frames(gnuais_batch *b, int *n_out)
{
	*n_out = (int) ais_oracle_frame_count(b->o);
	return GNUAIS_OK;
}

int gnuais_batch_drain_frames(gnuais_batch *b, gnuais_frame *h_out, int max, int *n_out)
{
	size_t n;
	ais_oracle_sort_frames(b->o);
	n = ais_oracle_frame_count(b->o);
	if ((int) n > max) {
		last_error = "fake drain_frames: overflow";
		return GNUAIS_E_OVERFLOW;
	}
	_Static_assert(sizeof(gnuais_frame) == sizeof(ais_frame), "one 64-byte record");
	if (n)
		memcpy(h_out, ais_oracle_frames(b->o), n * sizeof(gnuais_frame));
	ais_oracle_clear_frames(b->o);
	*n_out = (int) n;
	return GNUAIS_OK;
}

int gnuais_batch_counters(gnuais_batch *b, gnuais_counters *h_out)
{
	int c;
	for (c = 0; c < b->n_ch; c++) {
		const ais_hdlc *h = ais_oracle_hdlc(b->o, c);
		h_out[c].receivedframes = h->receivedframes;
		h_out[c].lostframes = h->lostframes;
		h_out[c].lostframes2 = h->lostframes2;
	}
	return GNUAIS_OK;
}

int gnuais_batch_fsm_state(gnuais_batch *b, gnuais_fsm_state *h_out)
{
	int c;
	for (c = 0; c < b->n_ch; c++) {
		const ais_hdlc *h = ais_oracle_hdlc(b->o, c);
		h_out[c].state = h->state;
		h_out[c].nstartsign = h->nstartsign;
		h_out[c].antallpreamble = h->antallpreamble > 15 ? 15 : h->antallpreamble;
		h_out[c].antallenner = h->antallenner;
		h_out[c].bitstuff = h->bitstuff;
		h_out[c].last = h->last;
		h_out[c].bufferpos = h->bufferpos;
	}
	return GNUAIS_OK;
}

int gnuais_batch_protodec_reset(gnuais_batch *b)
{
	int c;
	for (c = 0; c < b->n_ch; c++)
		ais_oracle_protodec_reset(b->o, c);
	return GNUAIS_OK;
}

/* the reference's d->buffer as the oracle keeps it: bufferpos cells inside a frame, the last frame's cells between */
int gnuais_batch_frame_bits(gnuais_batch *b, int channel, uint8_t *h_bits, int cap, int *n_bits)
{
	const ais_hdlc *h = ais_oracle_hdlc(b->o, channel);
	int n = h->bufferpos, i;
	if (h->state != 4 && h->state != 5) {
		*n_bits = -1;                           /* the double keeps no record of closed frames */
		return GNUAIS_OK;
	}
	for (i = 0; i < n && i < cap; i++)
		h_bits[i] = h->buffer[i];
	*n_bits = n;
	return GNUAIS_OK;
}

int gnuais_batch_pll_state(gnuais_batch *b, gnuais_pll_state *h_out)
{
	int c;
	for (c = 0; c < b->n_ch; c++) {
		int prev, lastbit;
		ais_oracle_get_pll(b->o, c, &h_out[c].pll, &prev, &lastbit);
		h_out[c].prev = prev;
		h_out[c].lastbit = lastbit;
	}
	return GNUAIS_OK;
}

int gnuais_batch_maxval(gnuais_batch *b, int16_t *h_out)
{
	memcpy(h_out, b->maxval, sizeof(int16_t) * (size_t) b->n_ch);
	return GNUAIS_OK;
}

int gnuais_batch_filter_host(gnuais_batch *b, const int16_t *h_samples, int len, float *h_out)
{
	int c;
	if (len < 1 || len > b->max_len) {
		last_error = "fake filter_host: argument";
		return GNUAIS_E_ARG;
	}
	for (c = 0; c < b->n_ch; c++) {         /* [len][n_ch] in and out; one channel in the shims' use */
		float *tmp = malloc(sizeof(float) * (size_t) len);
		int i;
		ais_oracle_filter_channel(b->o, c, h_samples + c, b->n_ch, len, tmp, &b->maxval[c]);
		for (i = 0; i < len; i++)
			h_out[(size_t) i * (size_t) b->n_ch + (size_t) c] = tmp[i];
		free(tmp);
	}
	return GNUAIS_OK;
}

int gnuais_batch_decode_bits(gnuais_batch *b, const uint8_t *h_bits, int stride, const int32_t *h_count)
{
	int c;
	for (c = 0; c < b->n_ch; c++)
		ais_oracle_decode_bits(b->o, c, h_bits + (size_t) c * (size_t) stride, h_count[c]);
	return GNUAIS_OK;
}

int gnuais_crc16_batch(int device, const uint8_t *h_data, int stride, const int32_t *h_len, int n_msgs,
		       uint16_t *h_crc)
{
	int i;
	(void) device;
	for (i = 0; i < n_msgs; i++)
		h_crc[i] = ais_crc16_x25(h_data + (size_t) i * (size_t) stride, (unsigned) h_len[i]);
	return GNUAIS_OK;
}

int gnuais_crc16_bits(int device, const uint8_t *h_bits, int n_bytes, uint16_t *h_crc, uint8_t *h_msb, int n_out)
{
	uint8_t bytes[64];
	int i, j;
	(void) device;
	if (n_bytes < 1 || n_bytes > 64 || n_out < 0 || n_out > 8 * n_bytes) {
		last_error = "fake crc16_bits: argument";
		return GNUAIS_E_ARG;
	}
	for (j = 0; j < n_bytes; j++) {
		unsigned v = 0;
		for (i = 0; i < 8; i++)
			v |= ((unsigned) h_bits[8 * j + i] << i) & 0xffu;
		bytes[j] = (uint8_t) v;
	}
	*h_crc = ais_crc16_x25(bytes, (unsigned) n_bytes);
	for (j = 0; j < n_out; j++)
		h_msb[j] = (uint8_t) ((bytes[j / 8] >> (7 - j % 8)) & 1);
	return GNUAIS_OK;
}

int gnuais_default_taps(float *out36)
{
	return ais_oracle_default_taps(out36) == 36 ? GNUAIS_OK : GNUAIS_E_ARG;
}
