/*
 * wavio.c -- the input side of the receive chain on the C boundary: sample files ->
 * the interleaved int16 [frames][channels] stream that receiver_run() / gnuais_batch_run_host*()
 * consume (gnuais src/receiver.c:102,107).
 *
 *   raw   what the reference itself does with a sound file (src/ais.c:173-182, 214-217): the file is
 *         a bare stream of little-endian int16 frames; a WAV header, if there is one, is demodulated
 *         as if it were 11 stereo frames of audio;
 *   RIFF  a proper WAVE reader: 16-bit PCM (plain or WAVE_FORMAT_EXTENSIBLE), any number of
 *         channels, unknown chunks skipped, odd chunk sizes padded, a truncated data chunk tolerated.
 *
 * Streaming: gnuais_wav_read() returns whole frames, as many as asked for while the data lasts, so a
 * caller can read in the reference's 1020-frame chunks or in batches of seconds.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/gnuais_hip.h"

struct gnuais_wav {
	FILE *f;
	int channels, rate;
	long long left;         /* bytes of sample data still to read; -1 = until the end of the file */
};

static unsigned le16(const unsigned char *p) { return p[0] | (p[1] << 8); }
static unsigned long le32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned long) p[3] << 24); }

int gnuais_wav_open(gnuais_wav **out, const char *path, int raw_channels)
{
	gnuais_wav *w;
	unsigned char h[12], ck[8], fmt[40];
	int have_fmt = 0;

	if (!out || !path || raw_channels < 0)
		return GNUAIS_E_ARG;
	*out = NULL;
	w = calloc(1, sizeof(*w));
	if (!w)
		return GNUAIS_E_STATE;
	w->f = fopen(path, "rb");
	if (!w->f) {
		free(w);
		return GNUAIS_E_ARG;
	}
	if (raw_channels > 0) {                 /* ais.c:216: the bytes as they are */
		w->channels = raw_channels;
		w->rate = 48000;
		w->left = -1;
		*out = w;
		return GNUAIS_OK;
	}
	if (fread(h, 1, 12, w->f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4))
		goto bad;
	for (;;) {
		unsigned long size;
		if (fread(ck, 1, 8, w->f) != 8)
			goto bad;               /* no data chunk */
		size = le32(ck + 4);
		if (!memcmp(ck, "fmt ", 4)) {
			unsigned long n = size < sizeof(fmt) ? size : sizeof(fmt);
			unsigned code, bits, align;
			if (size < 16 || fread(fmt, 1, n, w->f) != n)
				goto bad;
			code = le16(fmt);
			if (code == 0xFFFE && n >= 26)  /* WAVE_FORMAT_EXTENSIBLE: the sub-format GUID's first word */
				code = le16(fmt + 24);
			w->channels = (int) le16(fmt + 2);
			w->rate = (int) le32(fmt + 4);
			align = le16(fmt + 12);
			bits = le16(fmt + 14);
			if (code != 1 || bits != 16 || w->channels < 1 || align != 2u * (unsigned) w->channels)
				goto bad;       /* only 16-bit PCM */
			have_fmt = 1;
			if (fseek(w->f, (long) (size - n + (size & 1)), SEEK_CUR))
				goto bad;
		} else if (!memcmp(ck, "data", 4)) {
			if (!have_fmt)
				goto bad;
			w->left = (long long) size;     /* a file shorter than this just ends early */
			*out = w;
			return GNUAIS_OK;
		} else if (fseek(w->f, (long) (size + (size & 1)), SEEK_CUR)) {
			goto bad;
		}
	}
bad:
	fclose(w->f);
	free(w);
	return GNUAIS_E_ARG;
}

int gnuais_wav_channels(const gnuais_wav *w) { return w ? w->channels : 0; }
int gnuais_wav_rate(const gnuais_wav *w) { return w ? w->rate : 0; }

long gnuais_wav_read(gnuais_wav *w, int16_t *frames, long max_frames)
{
	size_t fb, want, got;
	if (!w || !frames || max_frames < 0)
		return GNUAIS_E_ARG;
	fb = sizeof(int16_t) * (size_t) w->channels;
	want = (size_t) max_frames;
	if (w->left >= 0 && (long long) (want * fb) > w->left)
		want = (size_t) (w->left / (long long) fb);
	got = fread(frames, fb, want, w->f);     /* fread() drops a partial trailing frame, like ais.c:216 */
	if (w->left >= 0)
		w->left -= (long long) (got * fb);
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_BIG_ENDIAN__
	{
		size_t i, n = got * (size_t) w->channels;
		for (i = 0; i < n; i++)
			frames[i] = (int16_t) (((uint16_t) frames[i] >> 8) | ((uint16_t) frames[i] << 8));
	}
#endif
	return (long) got;
}

void gnuais_wav_close(gnuais_wav *w)
{
	if (!w)
		return;
	fclose(w->f);
	free(w);
}
