/*
 * protodec_reset_main.c -- TEST INFRASTRUCTURE.  Drives the reference's public decoder names the way a host that holds
 * on to them may (src/protodec.h:73-76, src/protodec.c:54-100): protodec_initialize(), protodec_decode() bit by bit
 * (receiver.c:130), protodec_reset() in the middle of the stream -- also in the middle of a frame --, the fields and
 * the d->buffer cells a host can look at (protodec.h:44-71), protodec_deinit().  Linked twice by oracle/Makefile:
 *   oracle/_ref/reset_ref.bin   over the reference's OWN protodec.c (pure CPU): what must come out;
 *   oracle/_ref/reset_shim.bin  over gnuais_amd/csrc/protodec_hip.c (the decoder on the device) + the reference's
 *                               message layer with the replaced names renamed (-DGNUAIS_SHIM).
 * stdout: the reference's own message lines; stderr: one line per checkpoint.
 *
 * usage: reset_*.bin bits.bin every reset_at[,reset_at...]      (bits.bin: one byte per bit)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "protodec.h"
#include "cfg.h"

void protodec_deinit(struct demod_state_t *d);          /* src/protodec.c:78 (not in protodec.h) */
#ifdef GNUAIS_SHIM
void gnuais_protodec_flush(struct demod_state_t *d);
#endif

static void show(const char *what, long at, struct demod_state_t *d)
{
	unsigned long long h = 1469598103934665603ull;
	int i;
#ifdef GNUAIS_SHIM
	gnuais_protodec_flush(d);                       /* the shim's fields are current as of its last flush */
#endif
	for (i = 0; i < DEMOD_BUFFER_LEN; i++)
		h = (h ^ d->buffer[i]) * 1099511628211ull;
	fprintf(stderr, "%s %ld: state %d nstartsign %d antallpreamble %d antallenner %d bitstuff %d last %d bufferpos %d "
		"received %d lost %d lost2 %d seqnr %d buffer %016llx\n", what, at, d->state, d->nstartsign,
		d->antallpreamble > 15 ? 15 : d->antallpreamble, d->antallenner, d->bitstuff, d->last, d->bufferpos,
		d->receivedframes, d->lostframes, d->lostframes2, d->seqnr, h);
}

int main(int argc, char **argv)
{
	struct demod_state_t *d;
	long n, i, every, resets[64];
	int n_resets = 0, r = 0;
	char *bits, *p;
	FILE *in;

	if (argc < 4) {
		fprintf(stderr, "usage: %s bits.bin every reset_at[,reset_at...]\n", argv[0]);
		return 2;
	}
	in = fopen(argv[1], "rb");
	if (!in) {
		perror(argv[1]);
		return 2;
	}
	fseek(in, 0, SEEK_END);
	n = ftell(in);
	fseek(in, 0, SEEK_SET);
	bits = malloc((size_t) n + 1);
	if (fread(bits, 1, (size_t) n, in) != (size_t) n)
		return 2;
	fclose(in);
	every = atol(argv[2]);
	for (p = argv[3]; *p && n_resets < 64; p += strcspn(p, ","), p += *p == ',')
		resets[n_resets++] = atol(p);
	for (i = 0; i <= MAX_AIS_PACKET_TYPE; i++)
		skip_type[i] = 0;                       /* print every message type */
	d = malloc(sizeof *d);
	protodec_initialize(d, NULL, NULL, 'A');        /* protodec.c:54-76 */
	memset(d->buffer, 0, DEMOD_BUFFER_LEN);         /* hmalloc'ed and never cleared before the first frame (protodec.c:70):
	                                                 * the two programs must start from the same cells */
	protodec_reset(d);                              /* before the first bit: nothing to reach yet */
	for (i = 0; i < n; i++) {
		if (r < n_resets && resets[r] == i) {
			show("before-reset", i, d);
			protodec_reset(d);              /* protodec.c:87-100 */
			show("after-reset", i, d);
			r++;
		}
		protodec_decode(&bits[i], 1, d);        /* receiver.c:130 */
		if (every > 0 && (i + 1) % every == 0)
			show("at", i + 1, d);
	}
	show("end", n, d);
	fflush(stdout);
	protodec_deinit(d);                             /* protodec.c:78-85 */
	free(d);
	free(bits);
	return 0;
}
