/*
 * gnuais_hip.h -- C ABI of the MI355X (gfx950) batch AIS receive chain.
 *
 * This is the drop-in boundary for gnuais's per-sample hot path.  Every entry
 * point cites the reference interface it stands in for (paths relative to the
 * gnuais tree).  Plain C: opaque handle, plain pointers and sizes, int status.
 * libgnuais_hip.so is built by gnuais_amd/csrc/Makefile with hipcc for gfx950;
 * there is no CPU fallback -- every call fails with GNUAIS_E_HIP when no HIP
 * device is usable.
 *
 * One batch = N independent receivers (one per interleaved channel) that the
 * reference would create with N calls of init_receiver() (src/receiver.c:52-74)
 * and drive with N calls of receiver_run() per buffer (src/ais.c:237-247).
 */
#ifndef GNUAIS_HIP_H
#define GNUAIS_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNUAIS_OK          0
#define GNUAIS_E_ARG      -1   /* bad argument                                   */
#define GNUAIS_E_HIP      -2   /* HIP runtime / device error (see last_error)    */
#define GNUAIS_E_OVERFLOW -3   /* frame ring overflowed; oldest results kept     */
#define GNUAIS_E_STATE    -4   /* call sequence error                            */

#define GNUAIS_MAX_TAPS 1023   /* src/filter.h:28 BufferLen 1024 (len < BufferLen) */

/* One CRC-valid HDLC frame = what src/protodec.c:1100-1104 hands to
 * protodec_getdata(): `nbits` = bufferlen, payload = the bits of d->rbuffer
 * packed MSB first (AIS bit x = payload[x/8] >> (7 - x%8) & 1).  64 bytes. */
typedef struct gnuais_frame {
	uint32_t channel;      /* interleaved channel index (receiver ch_ofs)      */
	uint32_t end_bit;      /* bits fed to the deframer since reset before the
	                          bit that closed the frame, low 32 bits (orders
	                          frames in time together with flags[5:1]) */
	uint8_t  payload[53];  /* nbits/8 bytes used, rest zero                     */
	uint8_t  flags;        /* bit0: CRC ok (always set for delivered frames);
	                          bits 5:1 = bits 36:32 of end_bit: the 37-bit stamp
	                          wraps after 165 days of 9600 bit/s                */
	uint16_t nbits;        /* bufferpos - 22, src/protodec.c:1096               */
} gnuais_frame;

/* One entry of the reference's position cache, struct cache_ent (src/cache.h:27-56), strings
 * inline: what the cache holds for one MMSI after the per-type decoders called cache_position /
 * cache_vesseldata / cache_vesseldatab / cache_vesseldatabb / cache_vesselname /
 * cache_vessel_persons (src/cache.c:204-384).  Fields nobody has set keep cache_get()'s
 * defaults (src/cache.c:181-196): 0 for lat/lon/draught, -1 for the other numbers, and the
 * GNUAIS_V_* bit of an unset string is clear (the reference holds NULL there).  120 bytes. */
#define GNUAIS_V_POSITION 1u    /* received_pos is set: a type 1-3 / 4 / 18 report            */
#define GNUAIS_V_DATA     2u    /* received_data is set: type 5 / 19 / 24                     */
#define GNUAIS_V_PERSONS  4u    /* received_persons_on_board is set: DAC 1 FI 40              */
#define GNUAIS_V_NAME     8u    /* name and destination hold strings                          */
#define GNUAIS_V_CALLSIGN 16u   /* callsign holds a string                                    */
#define GNUAIS_V_STATIC   32u   /* imo, shiptype, A-D, draught were written                   */
typedef struct gnuais_vessel {
	int32_t  mmsi;
	uint32_t set;              /* GNUAIS_V_* */
	float    lat, lon;
	int32_t  hdg;
	float    course, sog;
	int32_t  navstat;
	int32_t  imo;
	int32_t  shiptype, A, B, C, D;
	float    draught;
	int32_t  persons_on_board;
	char     callsign[8];      /* <= 6 characters + NUL */
	char     name[24];         /* <= 20 characters + NUL */
	char     destination[24];
} gnuais_vessel;

/* per-channel counters: src/protodec.h:58-60, bumped at src/protodec.c:1103,1107,1112,
 * read by src/ais.c:296-310 */
typedef struct gnuais_counters {
	int32_t receivedframes, lostframes, lostframes2;
} gnuais_counters;

/* per-channel carry of receiver.c's loop: src/receiver.h:38-44 */
typedef struct gnuais_pll_state {
	uint32_t pll;
	int32_t prev, lastbit;
} gnuais_pll_state;

/* per-channel deframer control state: live fields of struct demod_state_t,
 * src/protodec.h:44-57.  antallpreamble is reported saturated at 15 (only
 * "> 14" is ever tested, src/protodec.c:1036). */
typedef struct gnuais_fsm_state {
	int32_t state, nstartsign, antallpreamble, antallenner, bitstuff, last, bufferpos;
} gnuais_fsm_state;

typedef struct gnuais_batch gnuais_batch;

/* ---- lifetime: init_receiver()/free_receiver(), src/receiver.c:52-82 --------
 * taps/n_taps: filter_init(len, taps) arguments (src/filter.c:57); NULL/0 =
 *   the reference table (src/receiver.c:39-50).
 * pllinc: rx->pllinc (src/receiver.c:69); 0 = 0x10000/5.  Values above 14 426 (fewer than
 * ~4.6 samples per bit; AIS at 48 kHz has 5) are refused with GNUAIS_E_ARG: the deframer's
 * per-segment bit pack is sized for at most 16 words.
 * max_len: largest `len` a run call will be given.
 * frame_capacity: frames that can be queued between two drains; 0 = default. */
int  gnuais_batch_create(gnuais_batch **out, int device, int n_channels,
			 const float *taps, int n_taps, unsigned pllinc,
			 int max_len, int frame_capacity);
void gnuais_batch_destroy(gnuais_batch *b);
/* back to the state init_receiver()+protodec_initialize() leave (src/protodec.c:54-76) */
int  gnuais_batch_reset(gnuais_batch *b);

/* ---- hot path: receiver_run() for every channel, src/receiver.c:87-135 ------
 * d_samples: DEVICE pointer, interleaved int16 [len][n_channels] (the layout
 *   receiver_run reads with step = num_ch, src/receiver.c:102,107).
 * stream: hipStream_t (NULL = default stream).  Asynchronous; results are
 *   read after gnuais_batch_sync().  len may exceed the reference's 4096.
 *   The calls of one batch are ordered (each continues the receivers' state): use one stream for a
 *   batch; when the stream changes between two calls the previous one is synchronised first.
 *   One batch must not be driven from two host threads at once; different batches may. */
int  gnuais_batch_run(gnuais_batch *b, const int16_t *d_samples, int len, void *stream);
/* same from a HOST buffer (what src/ais.c:216-247 holds): copies H2D, runs, syncs */
int  gnuais_batch_run_host(gnuais_batch *b, const int16_t *h_samples, int len);
/* the same without waiting for the device: the samples are copied into one of two pinned staging
 * buffers (the call returns when that copy is done and `h_samples` may be reused), the transfer and
 * the chain run asynchronously on an internal stream; the next call's copy overlaps them.  Results
 * after gnuais_batch_sync().  For callers that read a file or a socket in large pieces. */
int  gnuais_batch_run_host_async(gnuais_batch *b, const int16_t *h_samples, int len);
int  gnuais_batch_sync(gnuais_batch *b);

/* ---- input side, row f2: sample files -> interleaved int16 frames (src/ais.c:173-182,214-217) ---
 * raw_channels > 0: the file is a bare stream of little-endian int16 frames of that many channels,
 *   header and all, exactly as the reference reads a sound file; 0: parse RIFF/WAVE (16-bit PCM,
 *   plain or extensible, any channel count).  gnuais_wav_read(): whole frames read, 0 at the end. */
typedef struct gnuais_wav gnuais_wav;
int  gnuais_wav_open(gnuais_wav **out, const char *path, int raw_channels);
int  gnuais_wav_channels(const gnuais_wav *w);
int  gnuais_wav_rate(const gnuais_wav *w);
long gnuais_wav_read(gnuais_wav *w, int16_t *frames, long max_frames);
void gnuais_wav_close(gnuais_wav *w);

/* ---- stage entry points (parity taps; not needed by a drop-in user) ---------
 * filter_run_buf(), src/filter.c:106-143: d_out = DEVICE float [len][n_channels];
 * advances the FIR history exactly like a run call, nothing else. */
int  gnuais_batch_filter(gnuais_batch *b, const int16_t *d_samples, int len,
			 float *d_out, void *stream);
/* the same from and to HOST memory (plain-C callers: gnuais_amd/csrc/protodec_hip.c): h_samples int16
 * [len][n_channels], h_out float [len][n_channels]; synchronous */
int  gnuais_batch_filter_host(gnuais_batch *b, const int16_t *h_samples, int len, float *h_out);
/* protodec_decode(in, count, d), src/protodec.c:988-1122, for every channel:
 * h_bits = HOST uint8 [n_channels][stride], one byte per bit; h_count[n_channels] */
int  gnuais_batch_decode_bits(gnuais_batch *b, const uint8_t *h_bits, int stride,
			      const int32_t *h_count);
/* recovered (NRZI-decoded) bits of the LAST run call, one byte per bit:
 * h_bits HOST uint8 [n_channels][stride]; h_count[n_channels] = bits per channel */
int  gnuais_batch_last_bits(gnuais_batch *b, uint8_t *h_bits, int stride, int32_t *h_count);

/* the slicer's decisions of the LAST run call (`out > 0`, src/receiver.c:111), one byte per sample:
 * h_out HOST uint8 [n_channels][stride], stride >= len of that call */
int  gnuais_batch_last_signs(gnuais_batch *b, uint8_t *h_out, int stride);
/* facts about a batch, by name: "sign_exact" (1 if the receive path runs the sign-exact slicer),
 * "sign_eps" (its certification threshold), "sign_central_taps", "first_effective_tap",
 * "n_effective_taps", "compute_units", "device", "segments", "stream_depth" (calls between a
 * gnuais_batch_stream_nmea() call and the one that hands its text out) */
int  gnuais_batch_info(const gnuais_batch *b, const char *name, double *value);

/* ---- results ------------------------------------------------------------------
 * drain queued frames to the host, in the reference's print order within the
 * drained span (channel 0..N-1, then time).  *n_out = frames written. */
int  gnuais_batch_drain_frames(gnuais_batch *b, gnuais_frame *h_out, int max, int *n_out);
int  gnuais_batch_pending_frames(gnuais_batch *b, int *n_out);
/* stream-ordered: drop everything queued so far without copying it out (the
 * counters keep counting); for consumers that only read counters */
int  gnuais_batch_discard_frames(gnuais_batch *b, void *stream);
int  gnuais_batch_counters(gnuais_batch *b, gnuais_counters *h_out /* [n_channels] */);
int  gnuais_batch_total_received(gnuais_batch *b, long long *total);
/* filter_run_buf()'s return value for the last run: peak positive sample per channel
 * (src/filter.c:118-119; feeds the level log at src/receiver.c:137-147) */
int  gnuais_batch_maxval(gnuais_batch *b, int16_t *h_out /* [n_channels] */);
int  gnuais_batch_pll_state(gnuais_batch *b, gnuais_pll_state *h_out /* [n_channels] */);
int  gnuais_batch_fsm_state(gnuais_batch *b, gnuais_fsm_state *h_out /* [n_channels] */);
/* protodec_reset() (src/protodec.c:87-100) for every decoder of the batch: back to ST_SKURR, the frame in progress
 * dropped, counters untouched.  Synchronises. */
int  gnuais_batch_protodec_reset(gnuais_batch *b);
/* d->buffer of one channel (src/protodec.h:52, written at protodec.c:1019): the stored bits, one per byte, of the frame
 * in progress or -- between frames -- of the last frame that reached its stop bit; *n_bits = their number (may exceed
 * cap: the first cap are written), -1 when neither is on record (no frame yet; the last one given up at 449 bits). */
int  gnuais_batch_frame_bits(gnuais_batch *b, int channel, uint8_t *h_bits, int cap, int *n_bits);
/* the last n_taps input samples per channel, oldest first (live part of
 * struct filter.buffer, src/filter.h:60): h_out int16 [n_channels][n_taps] */
int  gnuais_batch_history(gnuais_batch *b, int16_t *h_out);

int  gnuais_batch_n_channels(const gnuais_batch *b);
int  gnuais_batch_n_taps(const gnuais_batch *b);

/* ---- misc -------------------------------------------------------------------- */
/* the reference coefficient table, src/receiver.c:39-50, rounded to fp32 */
int  gnuais_default_taps(float *out36);
/* CRC-16/X-25 on the device (protodec_sdlc_crc, src/protodec.c:106-118):
 * h_data = n_msgs rows of `stride` bytes, h_len[n_msgs]; h_crc[n_msgs] */
int  gnuais_crc16_batch(int device, const uint8_t *h_data, int stride, const int32_t *h_len,
			int n_msgs, uint16_t *h_crc);
/* protodec_calculate_crc's arithmetic (src/protodec.c:120-167) for one frame in one device call:
 * h_bits = 8 * n_bytes cells of one bit each, as the deframer leaves them in d->buffer (the first
 * cell of a byte is its least significant bit, src/protodec.c:138-143); *h_crc = CRC-16/X-25 over the
 * n_bytes bytes (a good frame + FCS gives 0x0f47, src/protodec.c:166); h_msb receives the first n_out
 * cells byte by byte with the most significant bit first (d->rbuffer, src/protodec.c:150-162;
 * n_out = 0: not wanted).  1 <= n_bytes <= 64, n_out <= 8 * n_bytes. */
int  gnuais_crc16_bits(int device, const uint8_t *h_bits, int n_bytes, uint16_t *h_crc,
		       uint8_t *h_msb, int n_out);
/* Row f1, first part -- the NMEA 0183 sentences of CRC-valid frames, byte-identical to what the
 * reference passes to serial_write() (protodec_getdata src/protodec.c:896-926 +
 * protodec_generate_nmea src/protodec.c:780-894).  Host-side, like the reference's own
 * post-stage.  frames[n_frames] in delivery order (gnuais_batch_drain_frames);
 * seqnr[n_channels] is the per-receiver rolling sequence digit d->seqnr (in/out, start at 0).
 * Sentences are written back to back, each "!AIVDM,...*hh\r\n".  *out_len = bytes needed;
 * out == NULL only sizes (seqnr still advances); too small a buffer -> GNUAIS_E_OVERFLOW. */
int  gnuais_nmea_from_frames(const gnuais_frame *frames, int n_frames, uint8_t *seqnr,
			     int n_channels, char *out, size_t out_cap, size_t *out_len,
			     int *n_sentences);
/* Row f1, complete -- everything protodec_getdata() (src/protodec.c:896-986) produces for CRC-valid
 * frames: the NMEA sentences as above, and the line it prints on stdout for each accepted frame:
 * "ch <id> type <t> mmsi <9 digits>:" + the fields of the per-type decoders (src/protodec.c:357-776)
 * + " (!<last sentence>)\n".  chanid[n_channels] are the receivers' names (receiver.h name /
 * d->chanid); NULL = 'A' + channel % 26.  Either output may be left out (nmea_len == NULL /
 * text_len == NULL); a NULL buffer with a non-NULL length only sizes. */
int  gnuais_messages_from_frames(const gnuais_frame *frames, int n_frames, uint8_t *seqnr,
				 const char *chanid, int n_channels, char *nmea, size_t nmea_cap,
				 size_t *nmea_len, int *n_sentences, char *text, size_t text_cap,
				 size_t *text_len, int *n_lines);
/* The same sentences formatted ON THE DEVICE, straight from the HBM frame ring (sort into the
 * reference's order, prefix sums for the text offsets and the per-channel sequence digit, one
 * thread per frame): consumes the queued frames like gnuais_batch_drain_frames() and copies only
 * the text to the host.  Byte-identical to gnuais_nmea_from_frames() over the drained records.
 * seqnr[n_channels] in/out as there.  GNUAIS_E_ARG if `out_cap` is too small (nothing consumed;
 * 164 bytes per pending frame always suffice). */
int  gnuais_batch_drain_nmea(gnuais_batch *b, uint8_t *seqnr, char *out, size_t out_cap,
			     size_t *out_len, int *n_sentences, int *n_frames);
/* Row f1 complete ON THE DEVICE: the sentences as above AND the stdout lines of
 * gnuais_messages_from_frames() -- "ch <id> type <t> mmsi <9 digits>:" + the fields of the per-type
 * decoders (src/protodec.c:357-776) + " (!<last sentence>)\n" -- one thread per frame, printf's %.6f /
 * %.1f / %.0f done exactly in integer arithmetic; only the two texts cross PCIe.  Consumes the queued
 * frames like gnuais_batch_drain_frames().  Byte-identical to the host formatter over the drained
 * records.  chanid[n_channels] or NULL as there.  GNUAIS_E_ARG (nothing consumed) unless nmea_cap >=
 * 164 and text_cap >= 512 bytes per pending frame (gnuais_batch_pending_frames). */
int  gnuais_batch_drain_messages(gnuais_batch *b, uint8_t *seqnr, const char *chanid, char *nmea,
				 size_t nmea_cap, size_t *nmea_len, int *n_sentences, char *text,
				 size_t text_cap, size_t *text_len, int *n_lines, int *n_frames);
/* Row f3 ON THE DEVICE: gnuais_vessels_from_frames() over the queued frames without draining them -- the
 * batch's position-cache entries (one gnuais_vessel per MMSI seen, sorted by MMSI, exactly what the
 * per-type decoders' cache_*() calls leave for a fresh cache), folded by a sort on (MMSI, arrival order)
 * and one thread per vessel.  Call it before the drain that consumes the frames.  *n_vessels = entries;
 * GNUAIS_E_OVERFLOW (with *n_vessels set) if cap is too small. */
int  gnuais_batch_fold_vessels(gnuais_batch *b, gnuais_vessel *vessels, int cap, int *n_vessels);
/* Row f3, CARRIED: the reference keeps one position cache for the whole run (src/cache.c:163-384) and every
 * decoder call updates it; this is that cache kept in device memory from batch to batch.
 * gnuais_batch_vessel_table_enable(capacity) makes an empty table for `capacity` vessels (a hash table keyed by MMSI,
 * twice as many slots).  From then on every gnuais_batch_stream_nmea() call also folds the frames it takes off
 * into the table, queued behind their formatter (two small kernels, no sort, nothing waited for).  A drain-type
 * caller calls gnuais_batch_vessel_table_update() before the drain that consumes the frames: the queued
 * frames are folded (and stay queued); once per span of frames, or a frame is applied twice (harmless for the
 * result: the fold is idempotent per span).  gnuais_batch_vessel_table() waits for what has been queued and
 * returns the entries sorted by MMSI: byte for byte what gnuais_vessels_from_frames() leaves when it is fed the
 * same spans one after another.  GNUAIS_E_OVERFLOW (with *n_vessels set) when `cap` is too small or more vessels
 * were seen than the table was enabled for.  _clear() empties the table. */
int  gnuais_batch_vessel_table_enable(gnuais_batch *b, int capacity);
int  gnuais_batch_vessel_table_update(gnuais_batch *b);
int  gnuais_batch_vessel_table(gnuais_batch *b, gnuais_vessel *vessels, int cap, int *n_vessels);
int  gnuais_batch_vessel_table_clear(gnuais_batch *b);
/* Streaming delivery of the same sentences.  Call once after every gnuais_batch_run(): the frames of
 * the runs since the previous call are taken off at once (the chain moves on to another frame ring);
 * their formatter and the copy of the text into pinned host memory are queued behind the chain with every
 * size taken on the device, so the call itself waits for nothing it has queued.  *text / *len: the
 * sentences of the call D calls ago, D = gnuais_batch_info(b, "stream_depth", &D) (7: a call takes about
 * five call periods from its first kernel to its text on the host); valid until the next call;
 * *n_frames = -1 while the pipeline fills.  The per-channel sequence digit is carried on the device.
 * Calls without runs in between flush what is in flight.  Errors of a call (ring overflow, watchdog)
 * are reported when its text is handed out.  Not to be mixed with the gnuais_batch_drain_*() calls on
 * one batch. */
int  gnuais_batch_stream_nmea(gnuais_batch *b, const char **text, size_t *len, int *n_sentences,
			      int *n_frames);
/* both at once: the records (for the host-side consumers: stdout text, vessel table, range) and the
 * device-formatted sentences (for serial / IPC) of the same drained span */
int  gnuais_batch_drain_frames_nmea(gnuais_batch *b, gnuais_frame *h_frames, int max, int *n_frames,
				    uint8_t *seqnr, char *out, size_t out_cap, size_t *out_len,
				    int *n_sentences);
/* Range statistics (range.c:32-45, called from the position decoders protodec.c:399,441,628):
 * best_range_km[channel] = max(itself, great-circle km from the station to every plausible
 * position in frames of type 1-3, 4 and 18), the reference's float arithmetic step for step.
 * A station position outside (-90,90) x (-180,180) degrees means "no location" as in
 * cfg.c:364 and leaves the array untouched.  log_range() (range.c:47-53) is the caller's:
 * print entries > 0.1 and zero them. */
int  gnuais_range_from_frames(const gnuais_frame *frames, int n_frames, int n_channels,
			      float my_lat_deg, float my_lon_deg, float *best_range_km);
/* The batched front of the reference's position cache (row f3).  Folds the frames, in order,
 * into `vessels` exactly as the reference's decoders fold them into its cache one
 * cache_*() call at a time (protodec.c:282,390,435,516,619,676-678,740,772): on entry
 * `*n_vessels` entries sorted by MMSI (0 = empty cache), on return the updated table, sorted.
 * A consumer then makes one sink call per vessel and batch instead of one per message.
 * GNUAIS_E_OVERFLOW when more than `cap` vessels are needed (table left as on entry). */
int  gnuais_vessels_from_frames(const gnuais_frame *frames, int n_frames, gnuais_vessel *vessels,
				int cap, int *n_vessels);
/* benchmark input builder: d_out[l][c] = d_base[c % n_base][(l + (c * 7919) % len) % len] */
int  gnuais_tile_channels(const int16_t *d_base, int n_base, int len, int16_t *d_out,
			  int n_channels, void *stream);
/* per-kernel timing of the last run (HIP events recorded on the stream each
 * kernel is launched on), ms[5]: [0] K1 fir_slice  [1] K2 pll  [2] K2b hdlc_deframe
 * [3] K3 hdlc_crc  [4] whole call.  Needs gnuais_batch_set_timing(b, 1) before the run. */
int  gnuais_batch_set_timing(gnuais_batch *b, int on);
int  gnuais_batch_last_timing(gnuais_batch *b, float *ms5);
/* mean over the (up to 64) most recent timed runs since set_timing(b, 1); with the
 * stage pipeline on, these are the durations WHILE the stages of neighbouring
 * calls overlap */
int  gnuais_batch_mean_timing(gnuais_batch *b, float *ms5, int *n_calls);
/* Options (every setting is bit-exact):
 *   "pipeline"      1 (default): every stage on its own stream, the stages of consecutive calls overlap; 0: one stream
 *   "nbuf"          hand-off sets in use = calls that may be in flight, 2..8 (default 3; more are allocated on demand)
 *   "fir_T"         outputs per wave in K1 (multiple of 32; default 512)
 *   "fir_variant"   3 = the sign-exact slicer (default where the table allows); 0 = the exact ordered sum for every sample
 *   "fir_mfma"      long tables (192 kHz), whole groups of 64 channels, calls longer than a segment: 1 = every segment but a call's
 *                   first runs its 48 central taps as an exact integer Toeplitz product on the matrix pipe (default), 0 = packed kernel only
 *   "fir_pk_taps"   long tables (192 kHz): 0 = 40 central taps where the table's bound allows (default), 48 = 48 of them
 *   "fir_flag2"     1 = the slicer reads sign and threshold of an output off one scaled sum (default); 0 = subtract + two gathers
 *   "pll_variant"   0 = by channel count (default: the time-parallel form, pll_tp.hip, up to 1536 channels; pll_h3.hip above);
 *                   7 / 8 force one
 *   "hdlc_variant"  1 = the event-driven deframer (default); 0 = the bit-serial one
 *   "hdlc_lpw"      channels per deframer wave, 1..64 (default by channel count: as few as keep the launch at <= 512 waves --
 *                   1, 2, 4, 8 or 16 -- up to 1536 channels, 16 up to 8192, 64 where the batch fills the chip)
 *   "streaming"     0 leaves the streamed delivery (gnuais_batch_stream_nmea switches it on)
 *   "timing_stride" with set_timing on, time every n-th call only (the event records of a timed call cost stream time)
 *   "stage_mask"    measurement only: bit 0 FIR/slicer, 1 PLL/NRZI, 3 deframer, 4 unstuff/CRC; results are wrong unless 0x1f */
int  gnuais_batch_set_option(gnuais_batch *b, const char *name, int value);
/* Optional, once, before real work, for processes that created HIP streams of their own before the batch: time the
 * stage -> stream assignments on `d_samples` (about 1.3 s of pipelined calls: two greedy searches and a longer
 * head-to-head with the default) and keep a searched assignment only if it beats the default by 3 % or more; RESETS the
 * batch.  Which hardware queue a stream gets depends on what else the process created before, is not queryable, and
 * matters by up to 1.7x; in a process without streams of its own the default assignment (the creation order inside
 * gnuais_batch_create) is the best one found on every box measured (bench.py reports both: uncalibrated_ms_per_step). */
int  gnuais_batch_autotune(gnuais_batch *b, const int16_t *d_samples, int len, void *stream,
			   float *ms_per_call);
/* The same for the delivery loop (gnuais_batch_run + gnuais_batch_stream_nmea): places the stream of the
 * kernel that copies the text to pinned memory (one more tenant for the few hardware queues: 0.62 or 1.5 ms
 * per C3 call, by luck of creation order, unless measured).  Call it after gnuais_batch_autotune(), once,
 * before real work; about 0.2 s; RESETS the batch and leaves it in streaming mode. */
int  gnuais_batch_autotune_delivery(gnuais_batch *b, const int16_t *d_samples, int len, void *stream,
				    float *ms_per_call);
const char *gnuais_last_error(void);
const char *gnuais_version(void);

/* ---- the MySQL sink behind a batch (src/out_mysql.c:174-297, called from src/protodec.c:383,430,510,612,670,737,768,891) ----
 * What a myout_ais_*() call does depends on the reference's mysql_keepsmall (src/out_mysql.c:133-166, cfg.h:80):
 *   keepsmall on : "UPDATE <table> SET <this call's columns> WHERE mmsi", INSERT only if no row was touched -- per
 *                  table and vessel only the LAST call of each kind leaves anything in the database;
 *   keepsmall off: (the reference's default, cfg.c:74) every call INSERTs a row of its own -- nothing may be dropped.
 * gnuais_sql_calls_from_frames() walks a batch of frame records in arrival order and returns the calls to make -- kind,
 * mmsi and the argument values the reference's decoders would pass, bit for bit: with keepsmall != 0 the surviving
 * calls in their original relative order (the same final rows with <= 5 statements per vessel and batch instead of one
 * or two per message), with keepsmall == 0 every call.  gnuais_sql_plan_from_frames() is the keepsmall != 0 form.
 * (The per-sentence myout_nmea() log rows are never reduced: one per sentence of gnuais_nmea_from_frames().)  Host code. */
#define GNUAIS_SQL_POSITION    1   /* myout_ais_position(my, t, mmsi, lat, lon, hdg, course, sog)            types 1-3, 18 */
#define GNUAIS_SQL_BASESTATION 2   /* myout_ais_basestation(my, t, mmsi, lat, lon)                            type 4        */
#define GNUAIS_SQL_VESSELDATA  3   /* myout_ais_vesseldata(my, t, mmsi, name, destination, draught, A, B, C, D) type 5      */
#define GNUAIS_SQL_VESSELDATAB 4   /* myout_ais_vesseldatab(my, t, mmsi, A, B, C, D)                          types 19, 24B */
#define GNUAIS_SQL_VESSELNAME  5   /* myout_ais_vesselname(my, t, mmsi, name, destination)                    types 19, 24A */
typedef struct gnuais_sql_call {
	int32_t kind, mmsi;
	float   lat, lon, hdg, course, sog, draught;
	int32_t A, B, C, D;
	char    name[24], destination[24];
} gnuais_sql_call;
int  gnuais_sql_calls_from_frames(const gnuais_frame *frames, int n_frames, int keepsmall, gnuais_sql_call *out,
				  int cap, int *n_out);
int  gnuais_sql_plan_from_frames(const gnuais_frame *frames, int n_frames, gnuais_sql_call *out, int cap, int *n_out);

/* ---- the receivers of one NODE: N channels over several GPUs (src/ais.c:141-147, 237-247) ----------
 * gnuais creates its receivers one by one and they share nothing; the main loop hands every receiver the same
 * interleaved buffer.  A node is that over the node's devices: device g owns the contiguous channel block
 * [g*N/G, (g+1)*N/G) as a gnuais_batch of its own, one host thread per device issues that device's copies and
 * launches, nothing is exchanged between devices (no collective).  Results are merged on the host: frame records
 * carry GLOBAL channel numbers and come in the reference's order (channel, then time).
 * devices / n_devices: HIP device indices, one shard each (an index may repeat: several shards on one device);
 *   NULL / 0 = every visible device.  The other arguments as gnuais_batch_create(), per shard. */
typedef struct gnuais_node gnuais_node;
int  gnuais_node_create(gnuais_node **out, const int *devices, int n_devices, int n_channels, const float *taps,
			int n_taps, unsigned pllinc, int max_len, int frame_capacity_per_device);
void gnuais_node_destroy(gnuais_node *nd);
int  gnuais_node_reset(gnuais_node *nd);
int  gnuais_node_n_devices(const gnuais_node *nd);      /* shards */
int  gnuais_node_n_channels(const gnuais_node *nd);
/* shard i: its device, first global channel, channel count and batch (for the per-batch calls above) */
int  gnuais_node_shard(const gnuais_node *nd, int i, int *device, int *first_channel, int *n_channels,
		       gnuais_batch **batch);
/* receiver_run() for all N channels from ONE host buffer, interleaved int16 [len][N] as src/ais.c:216 holds it:
 * every device's thread copies its columns (a strided 2-D copy) and queues its chain; the call returns when
 * the buffer may be reused.  Results after gnuais_node_sync(). */
int  gnuais_node_run_host(gnuais_node *nd, const int16_t *h_samples, int len);
/* the same with the samples already on the devices: d_samples[i] = shard i's DEVICE slab, interleaved
 * int16 [len][n_channels of shard i]; streams[i] (or NULL) as in gnuais_batch_run().  Asynchronous. */
int  gnuais_node_run(gnuais_node *nd, const int16_t *const *d_samples, int len, void *const *streams);
int  gnuais_node_sync(gnuais_node *nd);
/* merged results: records of every device, channel = global index, reference order (channel, then time) */
int  gnuais_node_pending_frames(gnuais_node *nd, int *n_out);
int  gnuais_node_drain_frames(gnuais_node *nd, gnuais_frame *h_out, int max, int *n_out);
/* gnuais_batch_stream_nmea() on every shard (each from its own thread): texts[g] / lens[g] = shard g's sentences of
 * the call `stream_depth` calls ago (n_devices entries; valid until the next call).  Written out in shard order they
 * are the node's sentences in the reference's order for that call: shard g's channels all lie before shard g+1's and
 * a sentence does not name its channel.  *n_frames = -1 while the pipelines fill, else the total. */
int  gnuais_node_stream_nmea(gnuais_node *nd, const char **texts, size_t *lens, int *n_sentences, int *n_frames);
int  gnuais_node_discard_frames(gnuais_node *nd);
int  gnuais_node_counters(gnuais_node *nd, gnuais_counters *h_out /* [n_channels] */);
int  gnuais_node_total_received(gnuais_node *nd, long long *total);
int  gnuais_node_maxval(gnuais_node *nd, int16_t *h_out /* [n_channels] */);
int  gnuais_node_pll_state(gnuais_node *nd, gnuais_pll_state *h_out /* [n_channels] */);
int  gnuais_node_set_option(gnuais_node *nd, const char *name, int value);      /* on every shard */
/* gnuais_batch_autotune() on every shard, one after the other; *best_ms_max = the slowest shard's best */
int  gnuais_node_autotune(gnuais_node *nd, const int16_t *const *d_samples, int len, void *const *streams,
			  float *best_ms_max);
const char *gnuais_node_last_error(void);   /* of the calling thread: which device failed and why */
/* what gnuais_node_create() could not do without failing, one line per shard ("" when nothing): a shard's host thread that
 * could not be pinned to its device's NUMA node.  gnuais_node_create itself fails -- with the device and both figures in
 * gnuais_node_last_error() -- for a device index that is not visible and for a shard that does not fit its device's free memory. */
const char *gnuais_node_warnings(const gnuais_node *nd);
/* Where a shard runs and how it fared, for whoever times a node (a slow device must show by itself): every shard's
 * host thread is pinned, when it starts, to the CPUs of the NUMA node its device hangs off (the device's PCI address
 * -> /sys/bus/pci/devices/<addr>/numa_node -> /sys/devices/system/node/node<n>/cpulist, intersected with what the
 * process may use; GNUAIS_NODE_PIN=0 leaves the threads alone), so that what it allocates from then on -- the batch's
 * pinned staging buffers among it -- is local to that device.  gnuais_node_mark() starts a measurement;
 * after a gnuais_node_sync(), shard i reports: calls since the mark, submit_ms = host time inside its run calls,
 * busy_ms = from its first submission to the end of its own sync. */
typedef struct gnuais_node_shard_stat {
	int32_t device, first_channel, n_channels;
	int32_t numa_node;      /* -1: unknown */
	int32_t pinned_cpus;    /* 0: not pinned */
	long long calls;
	double  submit_ms, busy_ms;
	char    pci[32];
} gnuais_node_shard_stat;
int  gnuais_node_mark(gnuais_node *nd);
int  gnuais_node_shard_stats(const gnuais_node *nd, int i, gnuais_node_shard_stat *out);

#ifdef __cplusplus
}
#endif
#endif
