/*
 * sinks_batch.c -- see include/gnuais_sinks.h.  Plain C on top of libgnuais_hip.so's message
 * layer (gnuais_messages_from_frames, gnuais_vessels_from_frames) and the gnuais tree's own,
 * unchanged sink functions.  Build inside the tree with -DGNUAIS_TREE (its headers), or
 * standalone with the prototypes below.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gnuais_sinks.h"

#ifdef GNUAIS_TREE
#include "serial.h"
#include "ipc.h"
#include "cache.h"
#else
extern int serial_write(struct serial_state_t *state, char *s, int len);           /* serial.h:29 */
extern int ipc_write(struct ipc_state_t *ipc, char *buffer, int buflength);        /* ipc.h:36    */
extern int cache_position(int received_t, int mmsi, int navstat, float lat, float lon, int hdg,
			  float course, int rateofturn, float sog);                 /* cache.h:66  */
extern int cache_vesseldata(int received_t, int mmsi, int imo, char *callsign, char *name,
			    char *destination, int shiptype, int A, int B, int C, int D, float draught);
extern int cache_vesseldatab(int received_t, int mmsi, char *callsign, int shiptype, int A, int B,
			     int C, int D);
extern int cache_vesseldatabb(int received_t, int mmsi, int shiptype, int A, int B, int C, int D);
extern int cache_vesselname(int received_t, int mmsi, char *name, const char *destination);
extern int cache_vessel_persons(int received_t, int mmsi, int persons_on_board);
#endif

static int grow(char **p, size_t *cap, size_t need)
{
	char *q;
	if (*cap >= need)
		return 1;
	q = realloc(*p, need);
	if (!q)
		return 0;
	*p = q;
	*cap = need;
	return 1;
}

/* One vessel's folded state -> the fewest cache_*() calls that leave the reference's cache entry
 * as the per-message calls would (each call overwrites whole field groups, cache.c:204-384):
 *   name + callsign + static   cache_vesseldata     (everything; a type 5, or 24A/19 + 24B)
 *   callsign + static          cache_vesseldatab    (type 24 part B alone: imo = draught = 0)
 *   name + static              cache_vesselname + cache_vesseldatabb   (type 19 alone)
 *   name                       cache_vesselname     (type 24 part A alone)
 * The table of a batch starts empty, so its bits say what THIS batch wrote. */
static int deliver_vessel(const gnuais_vessel *v, int t)
{
	int calls = 0;
	char cs[8], name[24], dest[24];
	memcpy(cs, v->callsign, sizeof cs);
	memcpy(name, v->name, sizeof name);
	memcpy(dest, v->destination, sizeof dest);
	if (v->set & GNUAIS_V_POSITION) {
		cache_position(t, v->mmsi, v->navstat, v->lat, v->lon, v->hdg, v->course, 0, v->sog);
		calls++;
	}
	switch (v->set & (GNUAIS_V_NAME | GNUAIS_V_CALLSIGN | GNUAIS_V_STATIC)) {
	case GNUAIS_V_NAME | GNUAIS_V_CALLSIGN | GNUAIS_V_STATIC:
		cache_vesseldata(t, v->mmsi, v->imo, cs, name, dest, v->shiptype, v->A, v->B, v->C, v->D,
				 v->draught);
		calls++;
		break;
	case GNUAIS_V_CALLSIGN | GNUAIS_V_STATIC:
		cache_vesseldatab(t, v->mmsi, cs, v->shiptype, v->A, v->B, v->C, v->D);
		calls++;
		break;
	case GNUAIS_V_NAME | GNUAIS_V_STATIC:
		cache_vesselname(t, v->mmsi, name, dest);
		cache_vesseldatabb(t, v->mmsi, v->shiptype, v->A, v->B, v->C, v->D);
		calls += 2;
		break;
	case GNUAIS_V_NAME:
		cache_vesselname(t, v->mmsi, name, dest);
		calls++;
		break;
	default:
		break;
	}
	if (v->set & GNUAIS_V_PERSONS) {
		cache_vessel_persons(t, v->mmsi, v->persons_on_board);
		calls++;
	}
	return calls;
}

/* the sink calls of one batch: sentences, stdout lines and the batch's folded vessel entries are there */
static int deliver_outputs(gnuais_sinks *s, const char *nmea, size_t nmea_len, const char *text, size_t text_len,
			   const gnuais_vessel *table, int n_v)
{
	const int want_nmea = s->serial || s->ipc, want_text = s->text_out != NULL;
	int i;

	if (want_nmea && nmea_len) {
		if (s->serial) {                        /* "!AIVDM,...*hh\r\n" x n  (protodec.c:883-885) */
			size_t off = 0;
			while (off < nmea_len) {        /* serial_write() takes an int length */
				size_t n = nmea_len - off > (1u << 30) ? (1u << 30) : nmea_len - off;
				serial_write(s->serial, (char *) nmea + off, (int) n);
				s->serial_calls++;
				off += n;
			}
		}
		if (s->ipc) {                           /* the same without CR LF  (protodec.c:886-888) */
			size_t n = 0, k;
			if (!grow(&s->ipcbuf, &s->ipc_cap, nmea_len + 1))
				return GNUAIS_E_ARG;
			for (k = 0; k < nmea_len; k++)
				if (nmea[k] != '\r' && nmea[k] != '\n')
					s->ipcbuf[n++] = nmea[k];
			s->ipcbuf[n] = 0;               /* ipc_write() logs the buffer with %s */
			ipc_write(s->ipc, s->ipcbuf, (int) n);
			s->ipc_calls++;
		}
	}
	if (want_text && text_len) {
		fwrite(text, 1, text_len, s->text_out);
		fflush(s->text_out);                    /* protodec.c:985, once per batch */
		s->flushes++;
	}
	if (s->use_cache && n_v) {
		const int t = (int) time(NULL);         /* received_t, protodec.c:905 */
		for (i = 0; i < n_v; i++)
			s->cache_calls += deliver_vessel(&table[i], t);
		s->vessels += n_v;
	}
	return GNUAIS_OK;
}

int gnuais_sinks_deliver(gnuais_sinks *s, const gnuais_frame *frames, int n_frames)
{
	size_t nmea_len = 0, text_len = 0;
	int n_sent = 0, n_lines = 0, n_v = 0, rc;
	const int want_text = s && s->text_out;

	if (!s || n_frames < 0 || (n_frames && !frames) || !s->seqnr || s->n_channels <= 0)
		return GNUAIS_E_ARG;
	if (n_frames == 0)
		return GNUAIS_OK;
	/* <= 2 sentences of <= 82 bytes and one line of < 1 KB per frame */
	if (!grow(&s->nmea, &s->nmea_cap, (size_t) n_frames * 168 + 64) ||
	    (want_text && !grow(&s->text, &s->text_cap, (size_t) n_frames * 1024 + 64)))
		return GNUAIS_E_ARG;
	/* the sequence digits advance for every accepted frame whether or not anybody listens
	 * (protodec.c:922-926), so the message layer always runs */
	rc = gnuais_messages_from_frames(frames, n_frames, s->seqnr, s->chanid, s->n_channels, s->nmea,
					 s->nmea_cap, &nmea_len, &n_sent, want_text ? s->text : NULL,
					 want_text ? s->text_cap : 0, want_text ? &text_len : NULL, &n_lines);
	if (rc != GNUAIS_OK)
		return rc;
	s->frames += n_frames;
	s->sentences += n_sent;
	if (s->use_cache) {
		if (s->table_cap < n_frames) {
			gnuais_vessel *q = realloc(s->table, sizeof(gnuais_vessel) * (size_t) n_frames);
			if (!q)
				return GNUAIS_E_ARG;
			s->table = q;
			s->table_cap = n_frames;
		}
		rc = gnuais_vessels_from_frames(frames, n_frames, s->table, s->table_cap, &n_v);
		if (rc != GNUAIS_OK)
			return rc;
	}
	return deliver_outputs(s, s->nmea, nmea_len, s->text, text_len, s->table, n_v);
}

/* The same delivery for a batch whose message layer ran on the device: sentences and stdout lines from
 * gnuais_batch_drain_messages() (called with s->seqnr and s->chanid), the vessel entries from
 * gnuais_batch_fold_vessels() -- nothing is formatted or folded on the host. */
int gnuais_sinks_deliver_formatted(gnuais_sinks *s, int n_frames, int n_sentences, const char *nmea, size_t nmea_len,
				   const char *text, size_t text_len, const gnuais_vessel *vessels, int n_vessels)
{
	if (!s || n_frames < 0 || n_sentences < 0 || (nmea_len && !nmea) || (text_len && !text) ||
	    n_vessels < 0 || (n_vessels && !vessels))
		return GNUAIS_E_ARG;
	s->frames += n_frames;
	s->sentences += n_sentences;
	return deliver_outputs(s, nmea, nmea_len, text, text_len, vessels, n_vessels);
}

/* src/out_mysql.h:37-45 */
extern int myout_ais_position(struct mysql_state_t *my, time_t tid, int mmsi, float lat, float lon, float hdg,
			      float course, float sog);
extern int myout_ais_basestation(struct mysql_state_t *my, time_t tid, int mmsi, float lat, float lon);
extern int myout_ais_vesseldata(struct mysql_state_t *my, time_t tid, int mmsi, char *name, char *destination,
				float draught, int A, int B, int C, int D);
extern int myout_ais_vesseldatab(struct mysql_state_t *my, time_t tid, int mmsi, int A, int B, int C, int D);
extern int myout_ais_vesselname(struct mysql_state_t *my, time_t tid, int mmsi, const char *name,
				const char *destination);
extern int myout_nmea(struct mysql_state_t *my, time_t tid, char *nmea);
extern int mysql_keepsmall;     /* src/cfg.h:80, set by the "mysql_keepsmall" directive (cfg.c:74); out_mysql.c:140 */

int gnuais_sinks_deliver_mysql(gnuais_sinks *s, struct mysql_state_t *my, long t, const gnuais_frame *frames,
			       int n_frames, const char *nmea, size_t nmea_len, long counts[2])
{
	int n = 0, i, rc;
	size_t off = 0;

	if (!s || !my || n_frames < 0 || (n_frames && !frames) || (nmea_len && !nmea))
		return GNUAIS_E_ARG;
	if (s->sql_cap < 2 * n_frames + 1) {            /* a type 19 message is two calls */
		gnuais_sql_call *q = realloc(s->sql, sizeof(gnuais_sql_call) * (size_t) (2 * n_frames + 1));
		if (!q)
			return GNUAIS_E_ARG;
		s->sql = q;
		s->sql_cap = 2 * n_frames + 1;
	}
	/* keepsmall off (the default): every call INSERTs its own row and none may be dropped; on: UPDATE-else-INSERT,
	 * only the last call of a kind per vessel leaves anything */
	rc = gnuais_sql_calls_from_frames(frames, n_frames, mysql_keepsmall, s->sql, s->sql_cap, &n);
	if (rc != GNUAIS_OK)
		return rc;
	for (i = 0; i < n; i++) {
		gnuais_sql_call *c = &s->sql[i];
		switch (c->kind) {
		case GNUAIS_SQL_POSITION:
			myout_ais_position(my, (time_t) t, c->mmsi, c->lat, c->lon, c->hdg, c->course, c->sog);
			break;
		case GNUAIS_SQL_BASESTATION:
			myout_ais_basestation(my, (time_t) t, c->mmsi, c->lat, c->lon);
			break;
		case GNUAIS_SQL_VESSELDATA:
			myout_ais_vesseldata(my, (time_t) t, c->mmsi, c->name, c->destination, c->draught, c->A, c->B,
					     c->C, c->D);
			break;
		case GNUAIS_SQL_VESSELDATAB:
			myout_ais_vesseldatab(my, (time_t) t, c->mmsi, c->A, c->B, c->C, c->D);
			break;
		case GNUAIS_SQL_VESSELNAME:
			myout_ais_vesselname(my, (time_t) t, c->mmsi, c->name, c->destination);
			break;
		default:
			break;
		}
	}
	if (counts)
		counts[0] += n;
	/* the sentence log: myout_nmea() gets d->nmea, the sentence without '!' and CR LF (protodec.c:891-892) */
	while (off < nmea_len) {
		char line[128];
		size_t e = off, len;
		while (e < nmea_len && nmea[e] != '\r' && nmea[e] != '\n')
			e++;
		len = e - off;
		if (len > 1 && nmea[off] == '!' && len - 1 < sizeof line) {
			memcpy(line, nmea + off + 1, len - 1);
			line[len - 1] = 0;
			myout_nmea(my, (time_t) t, line);
			if (counts)
				counts[1]++;
		}
		while (e < nmea_len && (nmea[e] == '\r' || nmea[e] == '\n'))
			e++;
		off = e;
	}
	return GNUAIS_OK;
}

void gnuais_sinks_free(gnuais_sinks *s)
{
	if (!s)
		return;
	free(s->sql);
	s->sql = NULL;
	s->sql_cap = 0;
	free(s->nmea);
	free(s->text);
	free(s->ipcbuf);
	free(s->table);
	s->nmea = s->text = s->ipcbuf = NULL;
	s->table = NULL;
	s->nmea_cap = s->text_cap = s->ipc_cap = 0;
	s->table_cap = 0;
}
