/*
 * receiver_hip.c -- drop-in for gnuais's src/receiver.c + src/filter.c (+ the
 * deframer half of src/protodec.c) on top of libgnuais_hip.so.
 *
 * Plain C, same exported names and semantics as src/receiver.h:48-51:
 *   init_receiver()  src/receiver.c:52-74    free_receiver()  src/receiver.c:76-82
 *   receiver_run()   src/receiver.c:87-148
 * so that src/ais.c (init at ais.c:139-147, run at ais.c:237-247, counters at
 * ais.c:296-310), src/input.c and every sink stay untouched.  The message layer
 * (protodec_getdata() and below, src/protodec.c:169-986) is NOT replaced: for
 * every CRC-valid frame this file fills d->rbuffer exactly as
 * protodec_calculate_crc() would (src/protodec.c:150-162) and calls the
 * reference's own protodec_getdata(bufferlen, d) (src/protodec.c:1104).
 *
 * ais.c calls receiver_run() once per receiver on the SAME interleaved buffer.
 * Receivers created with the same num_ch therefore share one GPU batch of
 * num_ch channels: the first call of a round copies the buffer to the device
 * and runs the whole chain for all channels in one launch; every call then
 * hands its own channel's frames to protodec_getdata(), in time order, and
 * refreshes the public counters -- the reference's order (per buffer: receiver
 * A's frames, then receiver B's).
 *
 * The reference's main loop is one thread (ais.c:214-263).  This file takes one process-wide lock around its table
 * of groups and around the device work of a call, so receivers may be driven from several threads -- under the one
 * rule the shared batch imposes: the receivers of a group (same num_ch) are given the SAME buffer, round by round,
 * as ais.c:237-247 does.  A round is started by the first call that brings a new buffer; a call that brings a
 * different buffer while other members have not been served from the current one yet would advance every channel
 * of the group with the wrong samples, so it is refused loudly (fatal) instead of decoding garbage.  The lock is NOT
 * held while the reference's protodec_getdata() runs (it writes to the serial port, takes the position cache's lock
 * and flushes stdout): a call copies its channel's frames out and delivers them unlocked.  It is recursive, so a
 * fatal handler may call free_receiver().  The table grows as needed.
 *
 * Build inside the gnuais tree with -DGNUAIS_TREE (uses the tree's headers and
 * hlog); outside it, include/gnuais_receiver_abi.h carries the two public
 * structs.  There is no CPU fallback: if the HIP library cannot run, this
 * aborts like the reference does on its own fatal errors (receiver.c:104-105).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE                     /* PTHREAD_RECURSIVE_MUTEX_INITIALIZER_NP */
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>

#ifdef GNUAIS_TREE
#include "receiver.h"
#include "hlog.h"
#include "cfg.h"
#include "hmalloc.h"
#else
#include "gnuais_receiver_abi.h"
#define hmalloc malloc
#define hfree free
#endif
#include "gnuais_hip.h"

#define MAX_LEN 4096                 /* receiver.c:85 FILTERED_LEN */

struct rx_group {
	int num_ch;
	gnuais_batch *batch;
	struct receiver **members;   /* [num_ch], NULL where no receiver was created */
	unsigned char *ran;          /* [num_ch] member already served in this round */
	gnuais_frame *frames;        /* frames of the current round, reference order */
	int n_frames, cap_frames;
	gnuais_counters *counters;
	gnuais_pll_state *pll;
	int16_t *maxval;
	const short *round_buf;
	int round_len;
};

static struct rx_group *groups;
static int n_groups, cap_groups;
static pthread_mutex_t big_lock = PTHREAD_RECURSIVE_MUTEX_INITIALIZER_NP;  /* die() may run a handler that frees receivers */

/* A fatal device error ends the program, as the reference's own fatal errors do (there is no CPU path to fall
 * back to).  A host that wants to close its sinks first installs a handler: it is called with the message and,
 * if it returns, abort() follows. */
static void (*fatal_handler)(const char *message);

void gnuais_receiver_on_fatal(void (*handler)(const char *message))
{
	fatal_handler = handler;
}

static void die(const char *what)
{
	char msg[600];
	snprintf(msg, sizeof msg, "gnuais-hip: %s: %s", what, gnuais_last_error());
	fprintf(stderr, "%s\n", msg);
	if (fatal_handler)
		fatal_handler(msg);
	abort();
}

static struct rx_group *group_for(int num_ch)
{
	int i;
	struct rx_group *g;
	for (i = 0; i < n_groups; i++)
		if (groups[i].num_ch == num_ch)
			return &groups[i];
	if (n_groups == cap_groups) {
		cap_groups = cap_groups ? 2 * cap_groups : 4;
		groups = realloc(groups, sizeof(*groups) * (size_t) cap_groups);
		if (!groups)
			abort();
	}
	g = &groups[n_groups++];
	memset(g, 0, sizeof(*g));
	g->num_ch = num_ch;
	if (gnuais_batch_create(&g->batch, 0, num_ch, NULL, 0, 0, MAX_LEN, 0) != GNUAIS_OK)
		die("gnuais_batch_create");
	g->members = calloc((size_t) num_ch, sizeof(*g->members));
	g->ran = calloc((size_t) num_ch, 1);
	g->counters = calloc((size_t) num_ch, sizeof(*g->counters));
	g->pll = calloc((size_t) num_ch, sizeof(*g->pll));
	g->maxval = calloc((size_t) num_ch, sizeof(*g->maxval));
	g->cap_frames = 1024;
	g->frames = malloc(sizeof(gnuais_frame) * (size_t) g->cap_frames);
	return g;
}

/* src/receiver.c:52-74 */
struct receiver *init_receiver(char name, int num_ch, int ch_ofs, struct serial_state_t *serial,
			       struct ipc_state_t *ipc)
{
	struct receiver *rx;
	struct rx_group *g;

	if (num_ch < 1 || ch_ofs < 0 || ch_ofs >= num_ch) {
		fprintf(stderr, "gnuais-hip: init_receiver: bad channel layout\n");
		abort();
	}
	rx = (struct receiver *) hmalloc(sizeof(struct receiver));
	memset(rx, 0, sizeof(struct receiver));
	rx->filter = NULL;                      /* the FIR state lives on the device */
	rx->decoder = hmalloc(sizeof(struct demod_state_t));
	protodec_initialize(rx->decoder, serial, ipc, name);
	rx->name = name;
	rx->lastbit = 0;
	rx->num_ch = num_ch;
	rx->ch_ofs = ch_ofs;
	rx->pll = 0;
	rx->pllinc = 0x10000 / 5;
	rx->prev = 0;
	rx->last_levellog = 0;

	pthread_mutex_lock(&big_lock);
	g = group_for(num_ch);
	g->members[ch_ofs] = rx;
	pthread_mutex_unlock(&big_lock);
	return rx;
}

/* src/receiver.c:76-82.  The device batch of a group goes away with its last member. */
void free_receiver(struct receiver *rx)
{
	int i, k, left;
	if (!rx)
		return;
	pthread_mutex_lock(&big_lock);
	for (i = 0; i < n_groups; i++) {
		struct rx_group *g = &groups[i];
		if (g->num_ch != rx->num_ch || !g->members || g->members[rx->ch_ofs] != rx)
			continue;
		g->members[rx->ch_ofs] = NULL;
		for (left = 0, k = 0; k < g->num_ch; k++)
			left += g->members[k] != NULL;
		if (left == 0) {
			gnuais_batch_destroy(g->batch);
			free(g->members);
			free(g->ran);
			free(g->frames);
			free(g->counters);
			free(g->pll);
			free(g->maxval);
			*g = groups[--n_groups];        /* keep the table dense */
			memset(&groups[n_groups], 0, sizeof(groups[n_groups]));
		}
		break;
	}
	pthread_mutex_unlock(&big_lock);
	hfree(rx);      /* like the reference, the decoder itself is not freed (receiver.c:76-82) */
}

static void start_round(struct rx_group *g, const short *buf, int len)
{
	int n = 0, rc;

	if (gnuais_batch_run_host(g->batch, (const int16_t *) buf, len) != GNUAIS_OK)
		die("gnuais_batch_run_host");
	rc = gnuais_batch_pending_frames(g->batch, &n);
	if (rc != GNUAIS_OK)
		die("gnuais_batch_pending_frames");
	if (n > g->cap_frames) {
		g->cap_frames = n * 2;
		g->frames = realloc(g->frames, sizeof(gnuais_frame) * (size_t) g->cap_frames);
	}
	if (gnuais_batch_drain_frames(g->batch, g->frames, g->cap_frames, &g->n_frames) != GNUAIS_OK)
		die("gnuais_batch_drain_frames");
	if (gnuais_batch_counters(g->batch, g->counters) != GNUAIS_OK ||
	    gnuais_batch_pll_state(g->batch, g->pll) != GNUAIS_OK ||
	    gnuais_batch_maxval(g->batch, g->maxval) != GNUAIS_OK)
		die("gnuais_batch state readback");
	memset(g->ran, 0, (size_t) g->num_ch);
	g->round_buf = buf;
	g->round_len = len;
}

/* src/receiver.c:87-148 */
void receiver_run(struct receiver *rx, short *buf, int len)
{
	struct rx_group *g;
	struct demod_state_t *d = rx->decoder;
	const int ch = rx->ch_ofs;
	gnuais_frame local[16], *mine = local;  /* this call's frames of this channel, delivered after the lock is gone */
	int n_mine = 0, cap_mine = 16, i, j, k;
	gnuais_counters cnt;
	int16_t peak;

	if (len > MAX_LEN)                      /* receiver.c:104-105 */
		abort();
	if (len <= 0)
		return;
	pthread_mutex_lock(&big_lock);
	g = group_for(rx->num_ch);
	/* a new round starts with a new buffer, or when this receiver has already
	 * been served from the current one (ais.c reuses the same buffer address) */
	if (g->round_buf != buf || g->round_len != len || g->ran[ch]) {
		if (g->round_buf && !g->ran[ch]) {
			/* this member has not been served from the current round and brings something else:
			 * a round started now would hand the members still waiting the wrong samples */
			int served = 0;
			for (k = 0; k < g->num_ch; k++)
				served += g->members[k] && g->ran[k];
			if (served) {
				fprintf(stderr, "gnuais-hip: receiver_run: ch_ofs %d of a %d-channel group was given "
					"another buffer (%p, %d) than the round in progress (%p, %d); the receivers of "
					"one group share one buffer per round (ais.c:237-247)\n", ch, g->num_ch,
					(void *) buf, len, (const void *) g->round_buf, g->round_len);
				die("receiver_run: buffer mismatch inside a round");
			}
		}
		start_round(g, buf, len);
	}
	g->ran[ch] = 1;

	/* this channel's frames, already in time order (protodec.c:1100-1104) */
	for (i = 0; i < g->n_frames; i++) {
		if ((int) g->frames[i].channel != ch)
			continue;
		if (n_mine == cap_mine) {
			gnuais_frame *grown = malloc(sizeof(*grown) * (size_t) cap_mine * 2);
			if (!grown)
				abort();
			memcpy(grown, mine, sizeof(*grown) * (size_t) n_mine);
			if (mine != local)
				free(mine);
			mine = grown;
			cap_mine *= 2;
		}
		mine[n_mine++] = g->frames[i];
	}
	cnt = g->counters[ch];
	/* public state the caller may read (receiver.h:38-44) */
	rx->pll = g->pll[ch].pll;
	rx->prev = g->pll[ch].prev;
	rx->lastbit = g->pll[ch].lastbit;
	peak = g->maxval[ch];
	pthread_mutex_unlock(&big_lock);

	for (i = 0; i < n_mine; i++) {
		const gnuais_frame *f = &mine[i];
		const int nbytes = f->nbits / 8;
		memset(d->rbuffer, 0, DEMOD_BUFFER_LEN);        /* protodec.c:150 */
		for (j = 0; j < nbytes; j++)
			for (k = 0; k < 8; k++)                 /* protodec.c:151-162 */
				d->rbuffer[8 * j + k] = (f->payload[j] >> (7 - k)) & 1;
		d->receivedframes++;                            /* protodec.c:1103 */
		protodec_getdata(f->nbits, d);                  /* protodec.c:1104 */
	}
	if (mine != local)
		free(mine);
	/* the counters the caller may read (ais.c:296-310) */
	d->receivedframes = cnt.receivedframes;
	d->lostframes = cnt.lostframes;
	d->lostframes2 = cnt.lostframes2;
	(void) peak;
#ifdef GNUAIS_TREE
	{       /* receiver.c:137-147 level log, from filter_run_buf()'s return value */
		float level = (float) peak / (float) 32768 * (float) 100;
		int level_distance = time(NULL) - rx->last_levellog;
		if (level > 95.0 && (level_distance >= 30 || level_distance >= sound_levellog)) {
			hlog(LOG_NOTICE, "Level on ch %c too high: %.0f %%", d->chanid, level);
			time(&rx->last_levellog);
		} else if (sound_levellog != 0 && level_distance >= sound_levellog) {
			hlog(LOG_INFO, "Level on ch %c: %.0f %%", d->chanid, level);
			time(&rx->last_levellog);
		}
	}
#endif
}
