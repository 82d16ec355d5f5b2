/*
 * dropin_ais_main.c -- TEST INFRASTRUCTURE.  Drives gnuais_amd/csrc/receiver_hip.c the way
 * gnuais's own main() drives src/receiver.c for a sound file (src/ais.c:139-147 receivers,
 * ais.c:173-182 file input in 1020-frame chunks, ais.c:214-247 the loop, ais.c:296-313
 * counters and teardown), linked against the reference's UNMODIFIED message layer: its
 * protodec.c (protodec_initialize / protodec_getdata and everything below) and the support
 * files that pulls in.  Only filter.c and receiver.c are replaced, by the HIP drop-in.
 * Built by `make -C oracle dropin` into oracle/_ref/dropin_ais.bin (the reference's sources are
 * compiled where they lie; nothing of them enters this repository).
 *
 * usage: dropin_ais.bin stereo.raw [channels]
 */
#include <stdio.h>
#include <stdlib.h>

#include "receiver.h"
#include "protodec.h"
#include "cfg.h"

#ifdef GNUAIS_SHIM      /* `make shim`: receiver.c is the reference's, the names below it are the device shims' */
void gnuais_protodec_set_batching(int bits);
void gnuais_protodec_flush(struct demod_state_t *d);
#endif

int main(int argc, char **argv)
{
	struct receiver *rx[2] = { NULL, NULL };
	int channels = argc > 2 ? atoi(argv[2]) : 2;
	int frames = 1024, got, i;
	short *buffer;
	FILE *in;

	if (argc < 2 || channels < 1 || channels > 2) {
		fprintf(stderr, "usage: %s interleaved.raw [1|2]\n", argv[0]);
		return 2;
	}
	in = fopen(argv[1], "rb");
	if (!in) {
		perror(argv[1]);
		return 2;
	}
#ifdef GNUAIS_SHIM
	if (getenv("GNUAIS_PROTODEC_BATCH"))            /* default: bits queue per decoder until the next buffer starts */
		gnuais_protodec_set_batching(atoi(getenv("GNUAIS_PROTODEC_BATCH")));
#endif
	if (getenv("GNUAIS_LEVELLOG"))                  /* exercises receiver_run()'s level log (receiver.c:137-147) */
		sound_levellog = atoi(getenv("GNUAIS_LEVELLOG"));
	for (i = 0; i <= MAX_AIS_PACKET_TYPE; i++)
		skip_type[i] = 0;                       /* print every message type */
	for (i = 0; i < channels; i++)                  /* ais.c:139-147 */
		rx[i] = init_receiver((char) ('A' + i), channels, i, NULL, NULL);
	frames -= frames % 5;                           /* ais.c:179-181 */
	buffer = malloc(sizeof(short) * (size_t) frames * (size_t) channels);
	while ((got = (int) fread(buffer, sizeof(short) * (size_t) channels, (size_t) frames, in)) > 0)
		for (i = 0; i < channels; i++)          /* ais.c:232-247 */
			receiver_run(rx[i], buffer, got);
#ifdef GNUAIS_SHIM
	gnuais_protodec_flush(NULL);
#endif
	for (i = 0; i < channels; i++)                  /* ais.c:296-310 */
		fprintf(stderr, "%c: received %d lost %d lost2 %d\n", 'A' + i, rx[i]->decoder->receivedframes,
			rx[i]->decoder->lostframes, rx[i]->decoder->lostframes2);
	for (i = 0; i < channels; i++)                  /* ais.c:312-313 */
		free_receiver(rx[i]);
	free(buffer);
	fclose(in);
	return 0;
}
