/* node_decode.c -- the product path of BASELINE's C4 in 40 lines: N channels (131072 by default) of interleaved
 * int16 samples from a raw file (or silence + a ramp if no file is given), decoded on whatever GPUs exist.
 *   gcc -O2 -Iinclude examples/node_decode.c -Lgnuais_amd -lgnuais_hip -Wl,-rpath,$PWD/gnuais_amd -o node_decode
 *   ./node_decode [n_channels [chunk [file.raw]]]                                  (src/ais.c:141-147, 214-263) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gnuais_hip.h"

int main(int argc, char **argv)
{
	const int n_ch = argc > 1 ? atoi(argv[1]) : 131072, chunk = argc > 2 ? atoi(argv[2]) : 4096;
	FILE *f = argc > 3 ? fopen(argv[3], "rb") : NULL;
	gnuais_node *nd = NULL;
	if (gnuais_node_create(&nd, NULL, 0, n_ch, NULL, 0, 0, chunk, 0) != GNUAIS_OK) {   /* every visible device */
		fprintf(stderr, "create: %s\n", gnuais_node_last_error());
		return 1;
	}
	fprintf(stderr, "%d channels over %d device shard(s)\n", n_ch, gnuais_node_n_devices(nd));
	int16_t *buf = calloc((size_t) chunk * n_ch, sizeof *buf);
	gnuais_frame *fr = malloc(sizeof *fr * 1000000);
	long long frames = 0;
	for (int it = 0; f ? 1 : it < 4; ++it) {
		int len = chunk;
		if (f && (len = (int) (fread(buf, sizeof *buf * n_ch, chunk, f))) <= 0) break;
		if (gnuais_node_run_host(nd, buf, len) != GNUAIS_OK) { fprintf(stderr, "run: %s\n", gnuais_node_last_error()); return 1; }
		int n = 0;                                  /* merged: global channel numbers, channel then time */
		if (gnuais_node_drain_frames(nd, fr, 1000000, &n) != GNUAIS_OK) { fprintf(stderr, "drain: %s\n", gnuais_node_last_error()); return 1; }
		for (int k = 0; k < n; ++k)
			printf("ch %u bit %u: %u bits, type %u\n", fr[k].channel, fr[k].end_bit, fr[k].nbits, fr[k].payload[0] >> 2);
		frames += n;
	}
	long long rx = 0;
	gnuais_node_total_received(nd, &rx);
	fprintf(stderr, "%lld frames drained, %lld received in all\n", frames, rx);
	gnuais_node_destroy(nd);
	free(buf); free(fr);
	return 0;
}
