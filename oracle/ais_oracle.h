/*
 * ais_oracle.h -- CPU restatement of the gnuais receive chain (the ORACLE).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may include, link or call this.  The product
 * path (gnuais_amd/, include/gnuais_hip.h) never does.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * the real reference compiled in place (oracle/_ref/libgnuais_ref.so, recipe in
 * oracle/Makefile) by tests/test_oracle_vs_ref.py, and against the golden
 * vectors under tests/golden/ that were generated from that same library.
 *
 * The restatement works on the *batched* layout the GPU path uses: N
 * independent channels, interleaved int16 input [len][n_ch] (the reference's
 * own layout, receiver.c:102,107 with step = num_ch), per-channel carry state.
 */
#ifndef AIS_ORACLE_H
#define AIS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIS_DEMOD_BUFFER_LEN 450   /* protodec.h:42 DEMOD_BUFFER_LEN */
#define AIS_MAX_TAPS         1023  /* filter.h:28 BufferLen 1024: len < BufferLen */

/* protodec.h:30-34 */
enum { AIS_ST_SKURR = 1, AIS_ST_PREAMBLE = 2, AIS_ST_STARTSIGN = 3,
       AIS_ST_DATA = 4, AIS_ST_STOPSIGN = 5 };

/* One CRC-valid HDLC frame.  Same 64-byte record the HIP path emits
 * (include/gnuais_hip.h gnuais_frame). */
typedef struct ais_frame {
	uint32_t channel;
	uint32_t end_bit;     /* bits fed to the deframer (since reset) before the
	                         STOPSIGN bit that closed this frame */
	uint8_t  payload[53]; /* nbits/8 on-air bytes; AIS bit x (MSB first, the
	                         reference's rbuffer[x]) = payload[x/8] >> (7 - x%8) & 1 */
	uint8_t  flags;       /* bit0: CRC ok */
	uint16_t nbits;       /* bufferpos - 22 (protodec.c:1096) */
} ais_frame;

/* HDLC deframer carry: the live fields of struct demod_state_t
 * (protodec.h:44-71; nskurr/npreamble/ndata/nstopsign/offset are write-only) */
typedef struct ais_hdlc {
	int32_t state, nstartsign, antallpreamble, antallenner, bitstuff, last, bufferpos;
	int32_t receivedframes, lostframes, lostframes2;
	uint32_t bits_seen;
	uint8_t buffer[AIS_DEMOD_BUFFER_LEN];   /* one byte per bit, as the reference */
} ais_hdlc;

typedef struct ais_oracle ais_oracle;

/* optional per-call taps of the intermediate stages */
typedef struct ais_run_out {
	float    *filtered;   /* [len][n_ch] like the input, or NULL          */
	uint8_t  *bits;       /* [n_ch][bits_cap] one byte per recovered bit   */
	uint32_t *nbits;      /* [n_ch] bits recovered in THIS call            */
	uint32_t  bits_cap;
	int16_t  *maxval;     /* [n_ch] filter_run_buf() return value, or NULL */
} ais_run_out;

/* receiver.c:39-49 coefficient table rounded to fp32; returns 36 */
int ais_oracle_default_taps(float *out36);
/* receiver.c:69 */
#define AIS_DEFAULT_PLLINC (0x10000 / 5)

ais_oracle *ais_oracle_create(int n_ch, const float *taps, int n_taps, unsigned pllinc);
void ais_oracle_destroy(ais_oracle *o);
void ais_oracle_reset(ais_oracle *o);

/* full chain, all channels: filter.c:106-143 -> receiver.c:109-135 ->
 * protodec.c:988-1122.  `in` is interleaved [len][n_ch].  No limit on len
 * (the reference's 4096 cap, receiver.c:104, is a stack-buffer artefact and
 * the result is chunk-size independent). */
int ais_oracle_run(ais_oracle *o, const int16_t *in, int len, ais_run_out *out);
/* same, channels [ch0, ch1) only; thread-safe for disjoint ranges when
 * out == NULL or out's buffers are per-channel */
int ais_oracle_run_range(ais_oracle *o, const int16_t *in, int len, int ch0, int ch1,
			 ais_run_out *out);
/* channels statically partitioned over n_threads pthreads */
int ais_oracle_run_mt(ais_oracle *o, const int16_t *in, int len, int n_threads);
/* same work on a de-interleaved copy: channel c at in_planar + c*len (benchmark context only) */
int ais_oracle_run_planar_mt(ais_oracle *o, const int16_t *in_planar, int len, int n_threads);

/* stage entry points */
void ais_oracle_filter_channel(ais_oracle *o, int ch, const int16_t *in, int step, int len,
			       float *out, int16_t *maxval);
void ais_oracle_decode_bits(ais_oracle *o, int ch, const uint8_t *bits, int n);

/* results / state */
size_t ais_oracle_frame_count(const ais_oracle *o);
const ais_frame *ais_oracle_frames(const ais_oracle *o);
void ais_oracle_sort_frames(ais_oracle *o);   /* (channel, end_bit) = reference order */
void ais_oracle_clear_frames(ais_oracle *o);
const ais_hdlc *ais_oracle_hdlc(const ais_oracle *o, int ch);
void ais_oracle_protodec_reset(ais_oracle *o, int ch);   /* protodec.c:87-100 */
void ais_oracle_get_pll(const ais_oracle *o, int ch, uint32_t *pll, int *prev, int *lastbit);
void ais_oracle_get_history(const ais_oracle *o, int ch, int16_t *out_n_taps);

/* protodec.c:106-118 (CRC-16/X-25; check("123456789") = 0x906E) */
uint16_t ais_crc16_x25(const uint8_t *data, unsigned len);

#ifdef __cplusplus
}
#endif
#endif
