/*
 * gnuais_receiver_abi.h -- the caller-visible structures of gnuais's receiver
 * interface, for building the drop-in (gnuais_amd/csrc/receiver_hip.c) OUTSIDE
 * the gnuais tree.  Inside the tree the reference's own headers are used
 * instead (-DGNUAIS_TREE): src/receiver.h:35-51 and src/protodec.h:44-76.
 *
 * These are interface facts, not code: ais.c dereferences both structs
 * (ais.c:258-260 rx->decoder, ais.c:296-310 decoder->receivedframes/...;
 * range.c:47-53 best_range/chanid), so a replacement must keep the field
 * layout.  sizeof(struct receiver) == 56, sizeof(struct demod_state_t) == 144 on
 * x86-64.
 */
#ifndef GNUAIS_RECEIVER_ABI_H
#define GNUAIS_RECEIVER_ABI_H

#include <time.h>

struct serial_state_t;
struct ipc_state_t;
struct filter;

#define DEMOD_BUFFER_LEN 450

struct demod_state_t {            /* src/protodec.h:44-71 */
	char chanid;
	int state;
	unsigned int offset;
	int nskurr, npreamble, nstartsign, ndata, nstopsign;
	int antallenner;
	unsigned char *buffer;
	unsigned char *rbuffer;
	char *tbuffer;
	int bufferpos;
	char last;
	int antallpreamble;
	int bitstuff;
	int receivedframes;
	int lostframes;
	int lostframes2;
	unsigned char seqnr;
	float best_range;
	struct serial_state_t *serial;
	struct ipc_state_t *ipc;
	char *serbuffer;
	char *ipcbuffer;
	char *nmea;
};

struct receiver {                 /* src/receiver.h:35-46 */
	struct filter *filter;
	char name;
	int lastbit;
	int num_ch;
	int ch_ofs;
	unsigned int pll;
	unsigned int pllinc;
	struct demod_state_t *decoder;
	int prev;
	time_t last_levellog;
};

/* src/receiver.h:48-51 */
extern struct receiver *init_receiver(char name, int num_ch, int ch_ofs,
				      struct serial_state_t *serial, struct ipc_state_t *ipc);
extern void free_receiver(struct receiver *rx);
extern void receiver_run(struct receiver *rx, short *buf, int len);

/* not in the reference: receiver_hip.c ends the program on a device error (no CPU path exists to fall back to, and
 * receiver_run() returns nothing); a host may install a handler that is called with the message first, e.g. to close
 * its sinks.  abort() follows if the handler returns. */
extern void gnuais_receiver_on_fatal(void (*handler)(const char *message));

/* src/protodec.h:73-76: stay in the reference's protodec.c (message layer) */
void protodec_initialize(struct demod_state_t *d, struct serial_state_t *serial,
			 struct ipc_state_t *ipc, char chanid);
void protodec_getdata(int bufferlengde, struct demod_state_t *d);

#endif
