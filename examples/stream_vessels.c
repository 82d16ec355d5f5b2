/* stream_vessels.c -- the streamed delivery in 45 lines: interleaved int16 samples from a raw file, NMEA sentences
 * on stdout as they leave the device (formatted there, gnuais_batch_stream_nmea), and the position cache kept on the
 * device for the whole run (gnuais_batch_vessel_table_*), printed once at the end the way out_json.c walks the
 * reference's cache.
 *   gcc -O2 -Iinclude examples/stream_vessels.c -Lgnuais_amd -lgnuais_hip -Wl,-rpath,$PWD/gnuais_amd -o stream_vessels
 *   ./stream_vessels n_channels chunk file.raw                    (src/ais.c:214-263, src/cache.c, src/out_json.c) */
#include <stdio.h>
#include <stdlib.h>
#include "gnuais_hip.h"

#define TRY(x) do { if ((x) != GNUAIS_OK) { fprintf(stderr, "%s: %s\n", #x, gnuais_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s n_channels chunk file.raw\n", argv[0]); return 2; }
	const int n_ch = atoi(argv[1]), chunk = atoi(argv[2]);
	FILE *f = fopen(argv[3], "rb");
	if (!f) { perror(argv[3]); return 2; }
	gnuais_batch *b = NULL;
	TRY(gnuais_batch_create(&b, 0, n_ch, NULL, 0, 0, chunk, 0));
	TRY(gnuais_batch_vessel_table_enable(b, 65536));
	int16_t *buf = malloc(sizeof *buf * (size_t) chunk * n_ch);
	const char *text;
	size_t len;
	int n_sent, n_frames, rd, more = 1;
	double depth = 0;
	long long frames = 0, sentences = 0;
	while (more || depth-- > 0) {                  /* after the file: `stream_depth` more calls flush what is in flight */
		if (more && (rd = (int) fread(buf, sizeof *buf * n_ch, chunk, f)) > 0) {
			TRY(gnuais_batch_run_host_async(b, buf, rd));
		} else if (more) {
			more = 0;
			TRY(gnuais_batch_info(b, "stream_depth", &depth));
		}
		TRY(gnuais_batch_stream_nmea(b, &text, &len, &n_sent, &n_frames));
		if (n_frames > 0) {
			fwrite(text, 1, len, stdout);
			frames += n_frames;
			sentences += n_sent;
		}
	}
	gnuais_vessel *v = malloc(sizeof *v * 65536);
	int n_v = 0;
	TRY(gnuais_batch_vessel_table(b, v, 65536, &n_v));
	for (int k = 0; k < n_v; ++k)
		fprintf(stderr, "vessel %d: set %#x lat %.6f lon %.6f name \"%s\"\n", v[k].mmsi, v[k].set, v[k].lat, v[k].lon, v[k].name);
	fprintf(stderr, "%lld frames, %lld sentences, %d vessels\n", frames, sentences, n_v);
	gnuais_batch_destroy(b);
	return 0;
}
