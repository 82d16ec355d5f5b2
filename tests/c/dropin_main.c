/*
 * dropin_main.c -- test driver for gnuais_amd/csrc/receiver_hip.c: does what
 * gnuais's main loop does for a stereo raw file (src/ais.c:139-147 init,
 * ais.c:173-182 file input with 1020-frame chunks, ais.c:214-247 loop,
 * ais.c:296-310 counters) and prints one line per delivered frame.  The two
 * message-layer entry points below stand in for the reference's protodec.c,
 * which is not part of this repository.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnuais_receiver_abi.h"

void protodec_initialize(struct demod_state_t *d, struct serial_state_t *serial,
			 struct ipc_state_t *ipc, char chanid)
{
	memset(d, 0, sizeof(*d));
	d->chanid = chanid;
	d->serial = serial;
	d->ipc = ipc;
	d->state = 1;
	d->buffer = malloc(DEMOD_BUFFER_LEN);
	d->rbuffer = malloc(DEMOD_BUFFER_LEN);
}

void protodec_getdata(int bufferlen, struct demod_state_t *d)
{
	int i;
	printf("ch %c bits %d payload ", d->chanid, bufferlen);
	for (i = 0; i + 8 <= bufferlen; i += 8) {
		int v = 0, k;
		for (k = 0; k < 8; k++)
			v = (v << 1) | d->rbuffer[i + k];
		printf("%02x", v);
	}
	printf("\n");
}

int main(int argc, char **argv)
{
	FILE *f;
	short *buffer;
	int channels = 2, buffer_l = 1024, n;
	struct receiver *rx_a, *rx_b;

	if (argc < 2) {
		fprintf(stderr, "usage: %s stereo.raw\n", argv[0]);
		return 2;
	}
	f = fopen(argv[1], "rb");
	if (!f) {
		perror(argv[1]);
		return 2;
	}
	rx_a = init_receiver('A', 2, 0, NULL, NULL);
	rx_b = init_receiver('B', 2, 1, NULL, NULL);
	buffer_l -= buffer_l % 5;                               /* ais.c:179-181 */
	buffer = malloc(sizeof(short) * (size_t) buffer_l * channels);
	while ((n = (int) fread(buffer, channels * sizeof(short), (size_t) buffer_l, f)) > 0) {
		receiver_run(rx_a, buffer, n);
		receiver_run(rx_b, buffer, n);
	}
	printf("A: received %d lost %d lost2 %d\n", rx_a->decoder->receivedframes,
	       rx_a->decoder->lostframes, rx_a->decoder->lostframes2);
	printf("B: received %d lost %d lost2 %d\n", rx_b->decoder->receivedframes,
	       rx_b->decoder->lostframes, rx_b->decoder->lostframes2);
	free_receiver(rx_a);
	free_receiver(rx_b);
	fclose(f);
	return 0;
}
