/*
 * gnuais_sinks.h -- batched front of gnuais's output sinks.
 *
 * The reference's protodec_getdata() (src/protodec.c:896-986) feeds its sinks one message at a
 * time: serial_write() and ipc_write() per sentence (src/protodec.c:883-888), one cache_*()
 * call under a global mutex per decoded message (src/cache.c:204-384), printf + fflush(stdout)
 * per message (src/protodec.c:934,984-985).  That is sized for the <= 75 msgs/s of two radio
 * channels.  gnuais_sinks_deliver() hands a whole drained batch of frame records to the SAME,
 * unchanged sink functions with the same bytes and the same final cache contents, in
 *   1 serial_write()  (src/serial.c:110-122)      for all sentences of the batch,
 *   1 ipc_write()     (src/ipc.c:121-134)         for all sentences of the batch,
 *   1 fwrite + 1 fflush                           for all stdout lines of the batch,
 *   <= 3 cache_*() calls per VESSEL seen in the batch (position, static data, persons)
 *                                                 instead of one per message.
 * The MySQL sink (src/out_mysql.c) gets a front of its own, gnuais_sinks_deliver_mysql().  With the reference's
 * mysql_keepsmall set (src/out_mysql.c:140) its statements are "UPDATE ... WHERE mmsi, else INSERT" per vessel and
 * table, so only the last call of each kind per vessel and batch is issued; with it clear (the default, cfg.c:74)
 * every message INSERTs its own row and every call is issued, in arrival order (gnuais_sql_calls_from_frames); in
 * both cases plus the per-sentence myout_nmea() log rows.  The JSON uplink
 * (src/out_json.c) reads the position cache (cache_rotate) and is thereby served by the cache front above.
 *
 * Host C above libgnuais_hip.so (gnuais_amd/csrc/sinks_batch.c); links against the gnuais tree's
 * serial.o / ipc.o / cache.o.
 */
#ifndef GNUAIS_SINKS_H
#define GNUAIS_SINKS_H

#include <stdio.h>
#include "gnuais_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

struct serial_state_t;          /* src/serial.h:24-26 */
struct ipc_state_t;             /* src/ipc.h:29-32    */

typedef struct gnuais_sinks {
	struct serial_state_t *serial;  /* d->serial of the receivers, or NULL (protodec.c:884) */
	struct ipc_state_t *ipc;        /* d->ipc, or NULL (protodec.c:887)                     */
	FILE *text_out;                 /* where protodec_getdata() prints: stdout; NULL = off  */
	int use_cache;                  /* cache_positions (src/cache.c:49)                     */
	uint8_t *seqnr;                 /* [n_channels] d->seqnr of every receiver, carried     */
	const char *chanid;             /* [n_channels] d->chanid, or NULL for 'A','B',...      */
	int n_channels;
	/* what the adapter did so far */
	long frames, sentences, vessels, serial_calls, ipc_calls, cache_calls, flushes;
	/* scratch, grown on demand; zero-initialise the struct, release with gnuais_sinks_free() */
	char *nmea, *text, *ipcbuf;
	size_t nmea_cap, text_cap, ipc_cap;
	gnuais_vessel *table;
	int table_cap;
	gnuais_sql_call *sql;
	int sql_cap;
} gnuais_sinks;

/* frames: one drained batch (gnuais_batch_drain_frames order).  GNUAIS_OK or GNUAIS_E_ARG. */
int  gnuais_sinks_deliver(gnuais_sinks *s, const gnuais_frame *frames, int n_frames);
/* The same sink calls for a batch whose message layer ran on the DEVICE: nmea / text from
 * gnuais_batch_drain_messages(b, s->seqnr, s->chanid, ...), vessels from gnuais_batch_fold_vessels()
 * (called before that drain).  Nothing is formatted or folded on the host. */
int  gnuais_sinks_deliver_formatted(gnuais_sinks *s, int n_frames, int n_sentences, const char *nmea,
				    size_t nmea_len, const char *text, size_t text_len,
				    const gnuais_vessel *vessels, int n_vessels);
/* The reference's own myout_ais_*() / myout_nmea() (src/out_mysql.h:37-45, unchanged) for one batch: the calls of
 * gnuais_sql_calls_from_frames(..., mysql_keepsmall, ...) -- the reference's own switch (cfg.h:80), read at every
 * call -- in arrival order, then one myout_nmea() per sentence of `nmea` (the batch's
 * "!AIVDM...\r\n" text; NULL: none).  t = received_t (src/protodec.c:905).  counts[0] += vessel-table calls,
 * counts[1] += myout_nmea calls.  Scratch comes from `s` (may be a zeroed struct). */
struct mysql_state_t;           /* src/out_mysql.h:29-35 */
int  gnuais_sinks_deliver_mysql(gnuais_sinks *s, struct mysql_state_t *my, long t, const gnuais_frame *frames,
				int n_frames, const char *nmea, size_t nmea_len, long counts[2]);
void gnuais_sinks_free(gnuais_sinks *s);

#ifdef __cplusplus
}
#endif
#endif
