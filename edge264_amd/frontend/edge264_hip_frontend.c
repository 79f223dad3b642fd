/*
 * edge264_hip_frontend.c -- edge264.h surface in front of the MI355X back end.
 *
 * One translation unit = the reference's front end (parsers, mvpred, DPB bookkeeping; compiled
 * from /root/reference through the include farm of edge264_amd/frontend/Makefile, with the four sample-kernel
 * files replaced by the emitters of this directory) + thin wrappers around its 7 public
 * functions (edge264.h:64-70).  The wrappers
 *   - run the reference's synchronous mode (n_threads = 0, src/edge264_headers.c:1285-1286: n_threads is a
 *     performance hint, the slices are parsed inside edge264_decode_NAL whatever the caller asked for) and own the
 *     DEVICE frame memory through the reference's own Edge264AllocCb hook (edge264.h:42-43); the HOST side of a frame
 *     (what Edge264Frame.samples points into) comes from the caller's allocator when one is given,
 *   - number the NAL units so that the emitters can tell slices apart,
 *   - after every NAL, close the frames whose last macroblock has been parsed
 *     (next_deblock_addr[pic]==INT_MAX, src/edge264_headers.c:567) and hand their command packet
 *     to the sink,
 *   - in get_frame, make the samples the reference points at (src/edge264.c:385-387) valid by
 *     waiting for the device and copying the frame into the host mirror.
 * Sinks: 2 = frames on the device like 0, packets queued like 1 (a driver batches the packets of many decoders);
 * 0 = libedge264_hip.so (include/edge264_hip.h, resolved with dlopen so that this file has
 * no link-time dependency on ROCm), 1 = capture (packets are queued for the caller; used by the
 * tests to replay them through the CPU oracle, and by tools/ to write capture files).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>

#define edge264_find_start_code e264ref_find_start_code
#define edge264_alloc e264ref_alloc
#define edge264_flush e264ref_flush
#define edge264_free e264ref_free
#define edge264_decode_NAL e264ref_decode_NAL
#define edge264_get_frame e264ref_get_frame
#define edge264_return_frame e264ref_return_frame
#include E264_FARM_EDGE264_C
#undef edge264_find_start_code
#undef edge264_alloc
#undef edge264_flush
#undef edge264_free
#undef edge264_decode_NAL
#undef edge264_get_frame
#undef edge264_return_frame

#include "e264_emit.h"
#include "edge264_hip.h"

__thread E264Emitter *e264_tls_emitter __attribute__((visibility("hidden"))); /* (declared in e264_emit.h) */

#define PUBLIC __attribute__((visibility("default")))
#define ON_DEVICE(e) ((e)->sink_kind != 1) /* sinks 0 and 2 keep the frames in HBM */
#define E264_FRONT_MAX_DEVICES 16

/* ---- back end binding (dlopen) ------------------------------------------------------------ */
static struct {
	void *lib;
	int (*device_open)(int, E264Device **);
	int (*stream_open)(E264Device *, E264Stream **);
	void (*stream_close)(E264Stream *);
	int (*frame_alloc)(E264Stream *, int, size_t, void **);
	void (*frame_free)(E264Stream *, int);
	int (*frame_fill)(E264Stream *, int, int);
	int (*frame_submit)(E264Stream *, const void *, size_t);
	void *(*packet_buffer)(E264Stream *, size_t);
	int (*frame_wait)(E264Stream *, int);
	int (*frame_download)(E264Stream *, int, void *, size_t);
	int (*stream_flush)(E264Stream *);
	void *(*host_alloc)(E264Device *, size_t);
	void (*host_free)(E264Device *, void *);
	void *(*frame_device_ptr)(E264Stream *, int);
	int (*packet_check)(const void *, size_t);
	const char *(*build_flags)(void);
	E264Device *devs[E264_FRONT_MAX_DEVICES]; /* one device object per GPU ordinal, shared by every decoder bound to that GPU */
} hip;
static pthread_mutex_t g_dev_lock = PTHREAD_MUTEX_INITIALIZER;

static int g_sink_kind = 0;

/* ---- packet buffers of the capture / batch sinks: recycled, never returned to the allocator while decoders run.
 * A 1080p packet is 0.6 - 2 MB; malloc / free of that size per picture means mmap / munmap (or heap growth and trimming)
 * per picture, i.e. the process-wide mmap lock and a TLB shootdown across every parser thread: with one packet
 * allocation per picture 16 host threads parsed 5.7 k pictures/s and 128 threads 4.5 k (tools/gpu_multi.sh).  Buffers
 * carry their capacity in a 64-byte prefix; e264front_free_packet puts them back on one process-wide list. */
struct E264PktBuf { size_t cap; struct E264PktBuf *next; E264Device *pinned_by; char pad[64 - sizeof(size_t) - 2 * sizeof(void *)]; };
static pthread_mutex_t g_pkt_lock = PTHREAD_MUTEX_INITIALIZER;
static struct E264PktBuf *g_pkt_free = NULL;
static size_t g_pkt_free_bytes = 0;
static int g_pkt_pinned = 0; /* e264front_set_pinned: the pool's buffers are page-locked (e264hip_host_alloc) -- the "pinned-arena ring" of
                              * the emitters assemble a picture's packet where the H2D transfer reads it, and the batch
                              * front end submits it in place (e264hip_submit_batch_pinned) instead of having it validated again and copied */
#define E264_PKT_POOL_MAX ((size_t)2 << 30) /* bytes kept for reuse; beyond that buffers go back to the allocator */
static size_t g_pkt_pinned_bytes = 0; /* page-locked buffers in the pool: counted apart (they are never handed back: hipHostFree drains every queue of the
                                       * device), so that they do not push ordinary buffers out of the pool's budget */
static uint8_t *e264_pkt_alloc_on(size_t bytes, E264Device *dev)
{
	struct E264PktBuf *b = NULL, **pp;
	const int pinned = g_pkt_pinned && dev != NULL && hip.host_alloc != NULL;
	pthread_mutex_lock(&g_pkt_lock);
	for (pp = &g_pkt_free; *pp; pp = &(*pp)->next)
		if ((*pp)->cap >= bytes && (*pp)->pinned_by == (pinned ? dev : NULL)) { /* page-locked through THIS device (its H2D engine reads it in place) */
			b = *pp; *pp = b->next;
			if (b->pinned_by) g_pkt_pinned_bytes -= b->cap; else g_pkt_free_bytes -= b->cap;
			break;
		}
	pthread_mutex_unlock(&g_pkt_lock);
	if (!b) {
		size_t cap = (bytes + bytes / 4 + 65535) & ~(size_t)65535; /* pictures of one stream differ in size: a quarter of headroom */
		b = pinned ? hip.host_alloc(dev, sizeof(*b) + cap) : malloc(sizeof(*b) + cap);
		if (!b) return NULL;
		b->cap = cap;
		b->pinned_by = pinned ? dev : NULL;
	}
	return (uint8_t *)(b + 1);
}
static void e264_pkt_free(void *data)
{
	if (!data) return;
	struct E264PktBuf *b = (struct E264PktBuf *)data - 1;
	pthread_mutex_lock(&g_pkt_lock);
	if (b->pinned_by) { b->next = g_pkt_free; g_pkt_free = b; g_pkt_pinned_bytes += b->cap; b = NULL; } /* always pooled */
	else if (g_pkt_free_bytes + b->cap <= E264_PKT_POOL_MAX) { b->next = g_pkt_free; g_pkt_free = b; g_pkt_free_bytes += b->cap; b = NULL; }
	pthread_mutex_unlock(&g_pkt_lock);
	free(b);
}
static int g_device_ordinal = 0;
static int g_download = 1; /* 0: edge264_get_frame leaves the samples in HBM (decode-to-device use, throughput runs) */

PUBLIC void e264front_set_sink(int kind) { g_sink_kind = kind; }
PUBLIC void e264front_set_device(int ordinal) { g_device_ordinal = ordinal; }
PUBLIC void e264front_set_download(int on) { g_download = on; }
PUBLIC void e264front_set_pinned(int on) { g_pkt_pinned = on; }
/* 1: pictures with inter macroblocks leave in the WIRE form (version 5, include/edge264_compact.h: P_Skip / plain 16x16 macroblocks without residual in 12
 * bytes instead of 40), which the back end unfolds on the device; decoders allocated afterwards.  Also E264_FRONT_COMPACT=1 in the environment. */
static int g_compact = -1;
PUBLIC void e264front_set_compact(int on) { g_compact = on != 0; }

/* the device object of GPU `ordinal` (opened on first use); a decoder is bound to the GPU selected by e264front_set_device at
 * the time of its edge264_alloc -- several GPUs in one process: one decoder population per GPU, SURVEY.md 8(e) */
static int hip_load(void);
static int hip_bind_ordinal(int ordinal, E264Device **dev)
{
	if (ordinal < 0 || ordinal >= E264_FRONT_MAX_DEVICES)
		return EINVAL;
	pthread_mutex_lock(&g_dev_lock);
	int r = hip_load();
	if (!r && !hip.devs[ordinal])
		r = hip.device_open(ordinal, &hip.devs[ordinal]);
	*dev = r ? NULL : hip.devs[ordinal];
	pthread_mutex_unlock(&g_dev_lock);
	return r;
}
static int hip_load(void)
{
	if (hip.lib)
		return hip.device_open ? 0 : ENODEV;
	const char *path = getenv("E264_HIP_LIB");
	char buf[4096];
	if (!path) { /* the back end is the sibling of this library: edge264_amd/libedge264_hipfront.so -> edge264_amd/libedge264_hip.so */
		Dl_info info;
		if (dladdr((void *)hip_load, &info) && info.dli_fname) {
			snprintf(buf, sizeof(buf), "%s", info.dli_fname);
			char *s = strrchr(buf, '/');
			if (s) {
				snprintf(s, sizeof(buf) - (size_t)(s - buf), "/libedge264_hip.so");
				path = buf;
			}
		}
	}
	hip.lib = dlopen(path ? path : "libedge264_hip.so", RTLD_NOW | RTLD_LOCAL);
	if (!hip.lib)
		return ENODEV;
#define BIND(n) if (!(*(void **)&hip.n = dlsym(hip.lib, "e264hip_" #n))) return ENODEV
	BIND(device_open); BIND(stream_open); BIND(stream_close); BIND(frame_alloc); BIND(frame_free); BIND(frame_fill);
	BIND(frame_submit); BIND(packet_buffer); BIND(frame_wait); BIND(frame_download); BIND(stream_flush);
	BIND(host_alloc); BIND(host_free); BIND(packet_check); BIND(build_flags); BIND(frame_device_ptr);
#undef BIND
	{ /* a timing-ablation build of the back end (wrong samples by design) is not a decoder: refused unless asked for by name */
		const char *f = hip.build_flags(), *a = getenv("E264_ALLOW_ABLATION");
		if ((strstr(f, "E264_ABL_") || strstr(f, "E264_PHASE_")) && !(a && a[0] == '1')) { hip.device_open = NULL; return ENODEV; }
	}
	return 0;
}

/* ---- frame memory: the reference's alloc/free hook (src/edge264_headers.c:113-133) --------
 * HOST side of a slot = what the reference dereferences: `samples` (Edge264Frame.samples points into it, I_PCM macroblocks
 * are written there by the parser) and `mbs` (its per-macroblock records).  DEVICE side = the HBM slot the kernels write.
 * With caller allocators (edge264.h:42-43) the host samples are the CALLER's memory -- that is where edge264_get_frame
 * delivers the picture, the point of handing an allocator to a decoder -- and the caller's free_cb gets back exactly the
 * two pointers its alloc_cb returned.  mbs is always ours (guard bands, see below); the caller's mbs block is held and
 * returned untouched. */
static void e264_alloc_cb(void **samples, unsigned samples_size, void **mbs, unsigned mbs_size, int errno_on_fail, void *arg)
{
	E264Emitter *e = arg;
	int slot = (int)((uint8_t **)samples - e->dec->samples_buffers);
	*samples = NULL;
	*mbs = NULL;
	if (slot < 0 || slot >= E264_MAX_SLOTS)
		return;
	void *mirror = NULL, *user_samples = NULL, *user_mbs = NULL;
	if (e->user_alloc) {
		e->user_alloc(&user_samples, samples_size, &user_mbs, mbs_size, errno_on_fail, e->user_arg);
		if (!user_samples || !user_mbs) {
			if ((user_samples || user_mbs) && e->user_free) e->user_free(user_samples, user_mbs, e->user_arg);
			return;
		}
	}
	/* Decode-to-device (e264front_set_download(0)): no picture is ever read back, the mirror only serves what the parser itself writes into sample
	 * memory (I_PCM, the concealment's blend): ordinary memory, not 3 MB of page-locked memory per slot -- 768 hipHostMalloc calls at the start of a
	 * 128-decoder run, each of which stalls every parsing thread of the process (profiles/r05_host.txt item 4).  A later set_download(1) still works:
	 * the download then lands in pageable memory. */
	const int plain_mirror = ON_DEVICE(e) && !g_download && !e->user_alloc;
	if (ON_DEVICE(e)) {
		if (hip.frame_alloc(e->hip_stream, slot, samples_size, (e->user_alloc || plain_mirror) ? NULL : &mirror)) { /* no pinned mirror beside caller memory */
			if (e->user_alloc && e->user_free) e->user_free(user_samples, user_mbs, e->user_arg);
			return;
		}
		/* "non-existing" frames are never written (headers.c:1122-1144): the reference leaves their memory as its allocator
		 * returned it; the device slot is cleared so that what a damaged stream predicts from them is at least deterministic */
		hip.frame_fill(e->hip_stream, slot, 0);
		if (plain_mirror) mirror = aligned_alloc(64, ((size_t)samples_size + 63) & ~(size_t)63);
	} else if (!e->user_alloc) {
		mirror = aligned_alloc(64, ((size_t)samples_size + 63) & ~(size_t)63);
	}
	if (e->user_alloc)
		mirror = user_samples;
	/* The reference's default allocator returns samples and mbs as ONE block, mbs right after samples
	 * (src/edge264.c:123-130), and its parsers lean on that: the neighbour records of the first macroblock row
	 * (mb - pic_width_in_mbs - 2 ...) are read -- then masked by the availability flags -- at addresses BEFORE
	 * mb_buffers[slot].  With a separate allocation those reads fall off the mapping (seen as sporadic SIGSEGVs
	 * on 1080p CABAC streams), so mbs sits between zeroed guard bands large enough for one macroblock row. */
	size_t guard = (size_t)mbs_size < 65536 ? 65536 : (size_t)mbs_size > (1u << 20) ? (1u << 20) : (((size_t)mbs_size + 4095) & ~(size_t)4095);
	size_t body = ((size_t)mbs_size + 63) & ~(size_t)63;
	uint8_t *base = aligned_alloc(4096, (2 * guard + body + 4095) & ~(size_t)4095);
	void *m = base ? base + guard : NULL;
	if (!mirror || !m) {
		free(base);
		if (ON_DEVICE(e)) hip.frame_free(e->hip_stream, slot);
		if (e->user_alloc) { if (e->user_free) e->user_free(user_samples, user_mbs, e->user_arg); }
		else if (!ON_DEVICE(e) || plain_mirror) free(mirror);
		return;
	}
	memset(base, 0, guard);
	memset(base + guard + body, 0, guard);
	e->slot[slot].mbs_base = base;
	e->slot[slot].samples = mirror;
	e->slot[slot].samples_size = samples_size;
	e->slot[slot].mbs = m;
	e->slot[slot].user_mbs = user_mbs;
	e->slot[slot].plain_mirror = plain_mirror;
	*samples = mirror;
	*mbs = m;
}

static void e264_free_cb(void *samples, void *mbs, void *arg)
{
	E264Emitter *e = arg;
	for (int s = 0; s < E264_MAX_SLOTS; s++) {
		if (samples && e->slot[s].samples == samples) {
			if (e->cur.valid && e->cur.slot == s)
				e->cur.valid = 0;
			e->fb[s].active = 0;
			if (ON_DEVICE(e)) hip.frame_free(e->hip_stream, s);
			if (e->user_alloc) { if (e->user_free) e->user_free(samples, e->slot[s].user_mbs, e->user_arg); }
			else if (!ON_DEVICE(e) || e->slot[s].plain_mirror) free(samples);
			free(e->slot[s].mbs_base);
			memset(&e->slot[s], 0, sizeof(e->slot[s]));
			return;
		}
	}
	(void)mbs;
}

/* I_PCM macroblocks issue no leaf call: the parser wrote their samples into the host mirror (src/edge264_slice.c:914-935);
 * decoded-but-unseen macroblocks are PCM.  Called when a picture is closed, and from the unref callback the moment a slice
 * fails: recover_slice is about to blend or overwrite those samples (src/edge264_headers.c:527-529 comes after :495-497). */
static void e264_lift_pcm(E264Emitter *e, int slot)
{
	E264FrameBuilder *b = &e->fb[slot];
	Edge264Decoder *dec = e->dec;
	if (!b->active)
		return;
	if (!b->multi && b->n_flushed >= b->n_mbs)
		return; /* every macroblock of the picture issued leaf calls exactly once: none is I_PCM (the usual case; the scan
		         * below reads 304 bytes of parser state per macroblock) */
	int flip = dec->frame_flip_bits >> slot & 1;
	const Edge264Macroblock *mbs = e->slot[slot].mbs;
	for (int a = 0; a < b->n_mbs; a++) {
		E264Mb *m = &b->mbs[a];
		if (m->kind != E264_MB_ABSENT && !(b->side[a].state & E264_ST_ERR))
			continue;
		const Edge264Macroblock *M = mbs + a % b->width_mbs + (a / b->width_mbs) * (b->width_mbs + 1);
		if ((b->side[a].state & E264_ST_ERR) && M->recovery_bits == flip) {
			/* marked erroneous by recover_slice, decoded again since, and no leaf call has started a new record (e264_touch
			 * clears the mark): it came back as I_PCM.  Reconstructed again from the new samples; not deblocked again unless
			 * deblock_mb saw it (emit_deblock.c then has already reset the record). */
			memset(m, 0, sizeof(*m));
			/* a new version of the macroblock: what an earlier packet did to the old one no longer counts.  It is deblocked again when -- and only when --
			 * the reference calls deblock_mb on it from here on (round 5, tools/damage_sweep.py --wide: the state kept "deblocked by an earlier packet" and
			 * the slice that had done it, so the deblocking the reference does to the new samples at the end of the picture was left out) */
			b->side[a].state = 0;
			b->side[a].dbk_slice = 0xffff;
		}
		if (m->kind == E264_MB_ABSENT && !(b->side[a].state & E264_ST_RECON) && M->recovery_bits == flip && !M->mbIsInterFlag) {
			m->kind = E264_MB_PCM;
			b->n_lifted++;
			m->qp[0] = M->QP[0]; m->qp[1] = M->QP[1]; m->qp[2] = M->QP[2];
			const int fe = b->side[a].dbk_slice != 0xffff ? b->side[a].fedges : M->filter_edges; /* deblock_mb has cleared what it filtered */
			m->flags = (uint8_t)((fe & 1 ? E264_MBF_EDGE_LEFT : 0) | (fe & 2 ? E264_MBF_EDGE_TOP : 0) | (fe ? E264_MBF_DEBLOCK : 0));
			m->nz_mask = 0xffff;
			m->slice = 0;
			/* whose it is: the slice that starts closest below it; of two entries with the same start (a slice that failed and its copy) the later one.
			 * (Slices may arrive in any order, 7.4.1.2.5 arbitrary slice order: the entries are in arrival order, not in address order.) */
			uint32_t best = 0;
			for (int i = 0; i < b->n_slices; i++)
				if (b->slice_filled[i] && b->slices[i].first_mb <= (uint32_t)a && b->slices[i].first_mb >= best) { best = b->slices[i].first_mb; m->slice = (uint16_t)i; }
			m->dbk_slice = b->side[a].dbk_slice != 0xffff ? b->side[a].dbk_slice : m->slice;
			while (b->payload_len & 7)
				e264_payload_append(b, "\0", 1);
			m->payload_off = (uint32_t)b->payload_len;
			int mbx = a % b->width_mbs, mby = a / b->width_mbs;
			const uint8_t *Y = e->slot[slot].samples + (size_t)(mby * 16) * dec->out.stride_Y + mbx * 16;
			const uint8_t *C = e->slot[slot].samples + dec->plane_size_Y + (size_t)(mby * 8) * dec->out.stride_C + mbx * 8;
			for (int y = 0; y < 16; y++) e264_payload_append(b, Y + (size_t)y * dec->out.stride_Y, 16);
			for (int y = 0; y < 8; y++) e264_payload_append(b, C + (size_t)y * dec->out.stride_C, 8);
			for (int y = 0; y < 8; y++) e264_payload_append(b, C + (dec->out.stride_C >> 1) + (size_t)y * dec->out.stride_C, 8);
		}
	}
}

/* ---- closing a frame: assemble the packet, give it to the sink ---------------------------
 * partial = 0: the picture is complete (next_deblock_addr[pic] == INT_MAX, src/edge264_headers.c:567).
 * partial = 1: a slice of the picture has just FAILED (src/edge264_headers.c:527-529).  By then the reference has
 *   reconstructed the slice's macroblocks up to the error and deblocked everything it had decoded (:499-525); what that
 *   deblocking did to macroblocks that are NOT decoded again stays in the picture for good, so it has to happen on the
 *   device too, on the samples of this moment: the picture-so-far goes out as a packet of its own.  A later packet of the
 *   same picture (the slice arrives again and the picture completes) carries the macroblocks decoded since; the ones an
 *   earlier packet has reconstructed are marked E264_MBF_DONE (records kept for their neighbours' bS, no payload, not
 *   reconstructed again), and a macroblock is deblocked by exactly the packet during which the reference called deblock_mb
 *   on it (emit_deblock.c) -- a macroblock that was deblocked, concealed and decoded again is NOT deblocked a second time
 *   unless the reference does so (its next_deblock_addr has moved past it, src/edge264_headers.c:922-925, 531-535). */
static int e264_finish_frame_(E264Emitter *e, int slot, int partial);
static int e264_finish_frame(E264Emitter *e, int slot, int partial)
{
	E264_PF_BEGIN;
	const int r = e264_finish_frame_(e, slot, partial);
	E264_PF_END(E264_PF_FINISH);
	return r;
}
static int e264_finish_frame_(E264Emitter *e, int slot, int partial)
{
	E264FrameBuilder *b = &e->fb[slot];
	Edge264Decoder *dec = e->dec;
	if (e->cur.valid && e->cur.slot == slot)
		e264_flush_mb(e);
	e264_lift_pcm(e, slot);
	if (b->oom) { /* a record or payload buffer could not grow: the picture's packet would be short of macroblocks */
		b->active = 0; b->oom = 0;
		return ENOMEM;
	}
	/* what the header says about the records (e264hip_packet_check holds it against them).  Every macroblock written once:
	 * counted as they were closed.  A picture with a failed slice (records superseded, motion records orphaned): counted here. */
	int n_coded = b->n_flushed + b->n_lifted, n_inter = b->n_inter;
	uint32_t ref_slots = b->ref_slots;
	if (b->multi) {
		n_coded = n_inter = 0;
		ref_slots = 0;
		for (int a = 0; a < b->n_mbs; a++) {
			const E264Mb *m = &b->mbs[a];
			n_coded += m->kind != E264_MB_ABSENT;
			if (m->kind != E264_MB_INTER)
				continue;
			n_inter++;
			uint32_t d[2];
			E264Motion mx;
			memcpy(d, m->modes, 8);
			e264_motion_expand(d[1], b->mot + d[0], &mx);
			for (int i = 0; i < 8; i++)
				if (mx.refPic[i] >= 0) ref_slots |= 1u << mx.refPic[i];
		}
	}
	if (b->n_slices == 0) { /* cannot happen for a decoded frame; keep the packet well formed */
		int serial = e->serial;
		e->serial = -1;
		e264_slice_index(e, b);
		e->serial = serial;
	}
	if (partial) { /* nothing new since the last packet of the picture? */
		int dirty = 0;
		for (int a = 0; a < b->n_mbs && !dirty; a++)
			dirty = (b->mbs[a].kind != E264_MB_ABSENT && !(b->side[a].state & E264_ST_RECON)) || (b->side[a].dbk_slice != 0xffff && !(b->side[a].state & E264_ST_DBK));
		if (!dirty)
			return 0;
	}
	const uint32_t motion_bytes = n_inter ? (uint32_t)b->mot_len : 0;
	/* layout: hdr | slices | mbs | motion records (if any inter MB) | payload */
	uint32_t slices_off = E264_ALIGN16((uint32_t)sizeof(E264FrameHdr));
	uint32_t mbs_off = E264_ALIGN16(slices_off + (uint32_t)sizeof(E264SliceParams) * (uint32_t)b->n_slices);
	uint32_t motion_off = E264_ALIGN16(mbs_off + (uint32_t)sizeof(E264Mb) * (uint32_t)b->n_mbs);
	uint32_t payload_off = E264_ALIGN16(motion_off + motion_bytes);
	uint32_t payload_bytes = E264_ALIGN16((uint32_t)b->payload_len);
	size_t total = (size_t)payload_off + payload_bytes;
	/* wire form: the picture's sections are folded into the outgoing buffer straight from the builder's arrays (instead of being copied there: 30 % fewer
	 * bytes to write, and over PCIe, for an encoder's pictures); a picture that goes out in several packets edits its records in a version-4 packet first
	 * (a buffer of the emitter's), which is then folded; pictures without inter macroblocks have nothing to fold and leave as version 4 */
	const int fold = e->compact && n_inter > 0;
	const size_t out_cap = fold ? total + e264_compact_table_bytes((uint32_t)b->width_mbs, (uint32_t)b->height_mbs) + 64 : total;
	uint8_t *out;
	if (e->sink_kind == 0) out = hip.packet_buffer(e->hip_stream, out_cap);
	else out = e264_pkt_alloc_on(out_cap, ON_DEVICE(e) ? (E264Device *)e->hip_dev : NULL);
	if (!out)
		return ENOMEM;
	uint8_t *pkt = out;
	const int fold_direct = fold && !partial && !b->multi; /* straight out of the builder's arrays; a picture in several packets edits its records in the packet first */
	if (fold && !fold_direct) {
		if (e->fold_cap < total) {
			free(e->fold_buf);
			e->fold_cap = total + total / 4;
			if (!(e->fold_buf = malloc(e->fold_cap))) {
				e->fold_cap = 0;
				if (e->sink_kind != 0) e264_pkt_free(out);
				return ENOMEM;
			}
		}
		pkt = e->fold_buf;
	}
	E264FrameHdr h = {0};
	h.magic = E264_MAGIC; h.version = E264_VERSION; h.total_bytes = (uint32_t)total;
	h.width_mbs = (uint16_t)b->width_mbs; h.height_mbs = (uint16_t)b->height_mbs;
	h.stride_Y = (uint32_t)dec->out.stride_Y; h.stride_C = (uint32_t)dec->out.stride_C;
	h.plane_size_Y = (uint32_t)dec->plane_size_Y; h.plane_size_C = (uint32_t)dec->plane_size_C;
	h.n_slices = (uint32_t)b->n_slices; h.slices_off = slices_off; h.mbs_off = mbs_off;
	h.motion_off = n_inter ? motion_off : 0;
	h.payload_off = payload_off; h.payload_bytes = payload_bytes;
	h.dst_slot = slot; h.frame_id = b->frame_id;
	h.n_coded_mbs = (uint32_t)n_coded; h.n_inter_mbs = (uint32_t)n_inter;
	h.ref_slots = ref_slots;
	/* the sections, and zeros in the (at most 15-byte) gaps between them: the packet's bytes are a function of the picture alone.
	 * (Until round 4 the whole 0.4 MB in front of the payload was cleared first and then overwritten.) */
	if (fold_direct) {
		h.total_bytes = (uint32_t)total; /* (of the version-4 form: the bound the fold checks its room against) */
		total = e264_compact_sections(&h, b->slices, b->mbs, b->mot, b->payload, b->payload_len, out, out_cap); /* (never 0: out_cap is its bound) */
		b->active = 0;
		goto send;
	}
	uint32_t at = 0;
#define E264_SECTION(off, src, n) do { memset(pkt + at, 0, (off) - at); memcpy(pkt + (off), (src), (n)); at = (uint32_t)((off) + (n)); } while (0)
	E264_SECTION(0, &h, sizeof(h));
	E264_SECTION(slices_off, b->slices, sizeof(E264SliceParams) * (size_t)b->n_slices);
	E264_SECTION(mbs_off, b->mbs, sizeof(E264Mb) * (size_t)b->n_mbs);
	if (motion_bytes)
		E264_SECTION(motion_off, b->mot, motion_bytes);
	memset(pkt + at, 0, payload_off - at);
#undef E264_SECTION
	if (b->payload_len) /* (a picture without a coded block has no payload buffer at all) */
		memcpy(pkt + payload_off, b->payload, b->payload_len);
	memset(pkt + payload_off + b->payload_len, 0, payload_bytes - b->payload_len);
	if (partial || b->multi) {
		/* a picture in several packets: what THIS packet reconstructs and deblocks, record by record (in the packet's copy) */
		E264Mb *pm = (E264Mb *)(pkt + mbs_off);
		for (int a = 0; a < b->n_mbs; a++) {
			if (pm[a].kind == E264_MB_ABSENT)
				continue;
			const int dbk_now = b->side[a].dbk_slice != 0xffff && !(b->side[a].state & E264_ST_DBK);
			if (!dbk_now)
				pm[a].flags &= (uint8_t)~(E264_MBF_DEBLOCK | E264_MBF_EDGE_LEFT | E264_MBF_EDGE_TOP);
			if (b->side[a].state & E264_ST_RECON) {
				pm[a].flags |= E264_MBF_DONE;
				pm[a].coded = 0;
				pm[a].payload_off = 0;
			}
			b->side[a].state |= (uint8_t)(E264_ST_RECON | (b->side[a].dbk_slice != 0xffff ? E264_ST_DBK : 0));
		}
		b->multi = 1;
		b->payload_len = 0; /* the records kept for a later packet carry no payload (DONE) */
	}
	if (!partial)
		b->active = 0;
	if (fold)
		total = e264_compact_packet(pkt, total, out, out_cap);
send:
	pkt = out;
	if (e->sink_kind == 0)
		return hip.frame_submit(e->hip_stream, pkt, total);
	/* sink 2: the batch driver submits these bytes with E264_SUBMIT_TRUSTED, which means "they have passed e264hip_packet_check"
	 * (include/edge264_hip.h): they pass it HERE, on the parser thread that produced them (host only, parallel per decoder; 0.1 ms
	 * per 1080p packet).  An emitter bug on a damaged stream is then EBADMSG from edge264_decode_NAL, not a GPU fault. */
	if (e->sink_kind == 2 && hip.packet_check && hip.packet_check(pkt, total)) {
		e264_pkt_free(pkt);
		return EBADMSG;
	}
	struct E264Captured *c = malloc(sizeof(*c));
	c->data = pkt; c->bytes = total; c->next = NULL;
	if (e->cap_tail) e->cap_tail->next = c; else e->cap_head = c;
	e->cap_tail = c;
	return 0;
}

static int e264_flush_partial(E264Emitter *e, int slot) { return e264_finish_frame(e, slot, 1); }

static int e264_close_ready_frames(E264Emitter *e)
{
	int ret = 0;
	for (int s = 0; s < E264_MAX_SLOTS; s++)
		if (e->fb[s].active && e->dec->next_deblock_addr[s] == INT_MAX) {
			int r = e264_finish_frame(e, s, 0);
			if (r) ret = r;
		}
	return ret;
}

/* ---- public API (edge264.h:64-70) ---------------------------------------------------------- */
PUBLIC const uint8_t *edge264_find_start_code(const uint8_t *buf, const uint8_t *end, int four_byte)
{
	return e264ref_find_start_code(buf, end, four_byte);
}

PUBLIC Edge264Decoder *edge264_alloc(int n_threads, Edge264LogCb log_cb, void *log_arg, int log_mbs,
	Edge264AllocCb alloc_cb, Edge264FreeCb free_cb, void *alloc_arg)
{
	/* n_threads: a performance hint in the reference (src/edge264.c:222-233: how many pthreads run worker_loop).  Here every
	 * slice is parsed inside edge264_decode_NAL (the n_threads == 0 path, src/edge264_headers.c:1285-1286) whatever the
	 * caller asks for: the API contract is the same (return codes, unref_cb once per slice NAL -- from the calling thread
	 * instead of a worker --, frame order); host parallelism comes from running one decoder per stream, which is also what
	 * fills the GPU.  Said once through the log callback, not hidden.
	 * alloc_cb / free_cb (edge264.h:42-43): honoured for the host side of every frame, see e264_alloc_cb; one without the
	 * other is EINVAL-class misuse and refused. */
	if ((alloc_cb != NULL) != (free_cb != NULL)) {
		if (log_cb) log_cb("edge264_alloc: alloc_cb and free_cb must be given together\n", log_arg);
		errno = EINVAL;
		return NULL;
	}
	if (n_threads > 0 && log_cb)
		log_cb("edge264_alloc: n_threads is a hint; the MI355X back end parses slices inside edge264_decode_NAL (run one decoder per stream for host parallelism)\n", log_arg);
	(void)log_mbs;
	E264Emitter *e = calloc(1, sizeof(*e));
	if (!e)
		return NULL;
	e->sink_kind = g_sink_kind;
	e->compact = g_compact >= 0 ? g_compact : (getenv("E264_FRONT_COMPACT") && atoi(getenv("E264_FRONT_COMPACT")) != 0);
	e->user_alloc = alloc_cb; e->user_free = free_cb; e->user_arg = alloc_arg;
	e->flush_partial = e264_flush_partial;
	if (ON_DEVICE(e)) {
		/* why a decoder cannot be had is said through the caller's log callback (and on stderr with E264_FRONT_VERBOSE set): "NULL" alone
		 * does not tell an integrator whether the back-end library, the GPU or memory is missing */
		int r = hip_bind_ordinal(g_device_ordinal, (E264Device **)&e->hip_dev);
		const char *what = "the HIP back end (libedge264_hip.so beside this library, or $E264_HIP_LIB) could not be loaded or no gfx950 device opened";
		if (!r) { r = hip.stream_open(e->hip_dev, (E264Stream **)&e->hip_stream); what = "e264hip_stream_open failed"; }
		if (r) {
			char msg[256];
			snprintf(msg, sizeof(msg), "edge264_alloc: %s (error %d: %s)\n", what, r, strerror(r));
			if (log_cb) log_cb(msg, log_arg);
			if (getenv("E264_FRONT_VERBOSE")) fputs(msg, stderr);
			free(e);
			errno = r;
			return NULL;
		}
	}
	Edge264Decoder *dec = e264ref_alloc(0, log_cb, log_arg, 0, e264_alloc_cb, e264_free_cb, e);
	if (!dec) {
		if (ON_DEVICE(e)) hip.stream_close(e->hip_stream);
		free(e);
		return NULL;
	}
	e->dec = dec;
	return dec;
}

static E264Emitter *emitter_of(Edge264Decoder *dec) { return dec ? dec->alloc_arg : NULL; }

/* The slice's result reaches the application through unref_cb (src/edge264_headers.c:495-497); edge264_decode_NAL itself
 * returns the header parser's code.  The wrapper learns from the same callback that a slice failed. */
struct E264Unref { E264Emitter *e; Edge264UnrefCb user_cb; void *user_arg; };
static void e264_unref_cb(int ret, void *arg)
{
	struct E264Unref *u = arg;
	E264Emitter *e = u->e;
	const int slot = e->trk_serial == e->serial ? e->trk_slot : e->dec->currPic; /* a slice may fail before its first leaf call */
	/* A slice that made no leaf call, for a picture that was complete before it came and has gone out (the reference marks completion after this
	 * callback, src/edge264_headers.c:538-567: INT_MAX here is an EARLIER slice's doing): a stray copy of a slice the picture already has.  It changes
	 * nothing in the picture and must not bring the picture's builder back -- round 5, tools/damage_sweep.py: a slice NAL cut behind its last
	 * macroblock decodes completely; its intact copy then fails before its first macroblock, the failure path below opened an empty builder for the
	 * finished picture, and the intra macroblocks of that picture went out once more as "I_PCM" lifted from a host mirror that never held them. */
	const int stray = slot >= 0 && slot < E264_MAX_SLOTS && e->trk_serial != e->serial && !e->fb[slot].active && e->dec->next_deblock_addr[slot] == INT_MAX;
	/* (the reference also calls unref_cb for every NAL that is not a slice, at once: src/edge264.c:356-357; a slice's call comes from its task,
	 * src/edge264_headers.c:497, which is still marked busy then: :596) */
	if (!stray && (0x100022u >> e->dec->nal_unit_type & 1) && e->dec->busy_tasks && slot >= 0 && slot < E264_MAX_SLOTS && e->slot[slot].samples) {
		/* A slice may also END without a single leaf call that sees its context: all of its macroblocks I_PCM (no leaf call at all), or intra without
		 * residual, with its deblocking switched off (deblock_mb then returns before it looks at the slice).  The picture still has to go out -- a
		 * picture made of such slices only had no builder and never reached the device (round 5, tools/stream_sweep.py --wide) -- and the slice
		 * needs its entry (first_mb_in_slice tells e264_lift_pcm whose the I_PCM macroblocks are): taken from the task, which is still marked busy
		 * while this callback runs (src/edge264_headers.c:495-497 comes before :596; one task at a time in the synchronous mode). */
		E264FrameBuilder *b = e264_builder(e, slot);
		const int idx = e264_slice_index(e, b);
		if (!b->slice_filled[idx])
			e264_fill_slice_task(e, b, idx, &e->dec->tasks[__builtin_ctz(e->dec->busy_tasks)]);
	}
	if (ret && !stray && slot >= 0 && slot < E264_MAX_SLOTS && e->slot[slot].samples) {
		e->failed_serial = e->serial;
		e->failed_slot = slot;
		if (e->cur.valid && e->cur.slot == slot)
			e264_flush_mb(e);
		/* from here on the picture's deblocking follows the calls the reference actually makes (a failed slice moves its
		 * next_deblock_addr bookkeeping, src/edge264_headers.c:531-535), not the macroblocks' filter_edges alone */
		e264_builder(e, slot)->multi = 1;
		e264_lift_pcm(e, slot); /* before recover_slice touches the samples */
	}
	if (u->user_cb)
		u->user_cb(ret, u->user_arg);
}

PUBLIC int edge264_decode_NAL(Edge264Decoder *dec, const uint8_t *buf, const uint8_t *end, Edge264UnrefCb unref_cb, void *unref_arg)
{
	E264Emitter *e = emitter_of(dec);
	if (!e)
		return EINVAL;
	e264_tls_emitter = e;
	e->serial++;
	struct E264Unref u = {e, unref_cb, unref_arg}; /* synchronous mode: every callback fires before decode_NAL returns */
	int ret = e264ref_decode_NAL(dec, buf, end, e264_unref_cb, &u);
	int r2 = 0;
	if (e->failed_serial == e->serial && e->fb[e->failed_slot].active) { /* what recover_slice has marked (src/edge264_headers.c:411) */
		E264FrameBuilder *b = &e->fb[e->failed_slot];
		const int flip = dec->frame_flip_bits >> e->failed_slot & 1;
		const Edge264Macroblock *Mb = e->slot[e->failed_slot].mbs;
		for (int a = 0; a < b->n_mbs; a++)
			if (Mb[a % b->width_mbs + (a / b->width_mbs) * (b->width_mbs + 1)].recovery_bits == flip + 2)
				b->side[a].state |= E264_ST_ERR;
	}
	/* a picture with a failed slice goes out NAL by NAL from then on: each packet is one state of the picture as the
	 * reference builds it (what the NAL decoded, what it deblocked), in the reference's order */
	for (int s = 0; s < E264_MAX_SLOTS; s++)
		if (e->fb[s].active && e->fb[s].multi && dec->next_deblock_addr[s] != INT_MAX) {
			int r = e264_finish_frame(e, s, 1);
			if (r) r2 = r;
		}
	int r3 = e264_close_ready_frames(e);
	if (!r2) r2 = r3;
	e264_tls_emitter = NULL;
	return ret ? ret : (r2 == ENOMEM || r2 == EBADMSG ? r2 : 0);
}

PUBLIC int edge264_get_frame(Edge264Decoder *dec, Edge264Frame *out, int borrow)
{
	E264Emitter *e = emitter_of(dec);
	if (!e)
		return EINVAL;
	int ret = e264ref_get_frame(dec, out, borrow);
	if (ret == 0 && ON_DEVICE(e) && g_download) {
		uintptr_t mask = (uintptr_t)out->return_arg;
		for (int s = 0; s < E264_MAX_SLOTS; s++)
			if (mask >> s & 1)
				hip.frame_download(e->hip_stream, s, e->slot[s].samples, e->slot[s].samples_size);
	}
	return ret;
}

PUBLIC void edge264_return_frame(Edge264Decoder *dec, void *return_arg)
{
	e264ref_return_frame(dec, return_arg);
}

PUBLIC void edge264_flush(Edge264Decoder *dec)
{
	E264Emitter *e = emitter_of(dec);
	if (!e)
		return;
	e264_tls_emitter = e;
	e264ref_flush(dec);
	e264_tls_emitter = NULL;
	e->cur.valid = 0;
	for (int s = 0; s < E264_MAX_SLOTS; s++)
		e->fb[s].active = 0;
	if (ON_DEVICE(e))
		hip.stream_flush(e->hip_stream);
}

PUBLIC void edge264_free(Edge264Decoder **pdec)
{
	if (!pdec || !*pdec)
		return;
	E264Emitter *e = emitter_of(*pdec);
#ifdef E264_EMIT_PROFILE
	{
		static const char *nm[E264_PF_N] = {"touch(incl. flush)", "levels", "flush_mb", "finish_frame", "deblock_mb", "intra leaves"};
		for (int k = 0; k < E264_PF_N; k++)
			if (e && e->pf_calls[k]) fprintf(stderr, "emit-profile %-20s calls %10llu  cycles %14llu  (%.0f per call)\n", nm[k], e->pf_calls[k], e->pf[k], (double)e->pf[k] / (double)e->pf_calls[k]);
	}
#endif
	e264_tls_emitter = e;
	e264ref_free(pdec); /* releases every slot through e264_free_cb */
	e264_tls_emitter = NULL;
	if (!e)
		return;
	if (ON_DEVICE(e)) hip.stream_close(e->hip_stream);
	for (int s = 0; s < E264_MAX_SLOTS; s++) {
		free(e->fb[s].mbs); free(e->fb[s].side); free(e->fb[s].mot); free(e->fb[s].payload);
		if (e->fb[s].cap_slices) { free(e->fb[s].slices); free(e->fb[s].slice_serial); free(e->fb[s].slice_filled); }
	}
	while (e->cap_head) {
		struct E264Captured *c = e->cap_head;
		e->cap_head = c->next;
		e264_pkt_free(c->data);
		free(c);
	}
	free(e->fold_buf);
	free(e);
}

/* ---- capture sink access (tests, tools) ---------------------------------------------------- */
/* Pops the oldest captured packet; the caller owns *data (free with e264front_free_packet). */
PUBLIC int e264front_take_packet(Edge264Decoder *dec, void **data, size_t *bytes)
{
	E264Emitter *e = emitter_of(dec);
	if (!e || !e->cap_head)
		return ENOMSG;
	struct E264Captured *c = e->cap_head;
	e->cap_head = c->next;
	if (!e->cap_head) e->cap_tail = NULL;
	*data = c->data;
	*bytes = c->bytes;
	free(c);
	return 0;
}

PUBLIC void e264front_free_packet(void *data) { e264_pkt_free(data); }

/* sink 2 (external batcher, edge264_amd/driver/e264_multi.cpp): the device stream that holds this decoder's
 * frames and the process-wide device object, so that the driver can hand the queued packets of MANY decoders
 * to e264hip_submit_batch in one launch. */
PUBLIC void *e264front_stream(Edge264Decoder *dec)
{
	E264Emitter *e = emitter_of(dec);
	return e && ON_DEVICE(e) ? e->hip_stream : NULL;
}
PUBLIC void *e264front_device(void) { E264Device *d; return hip_bind_ordinal(g_device_ordinal, &d) ? NULL : d; } /* of the GPU selected now */
PUBLIC void *e264front_device_of(Edge264Decoder *dec)
{
	E264Emitter *e = emitter_of(dec);
	return e && ON_DEVICE(e) ? e->hip_dev : NULL;
}

/* Decode-to-device (e264front_set_download(0)): the HBM address that corresponds to a plane pointer of an Edge264Frame
 * (samples[i] / samples_mvc[i] as edge264_get_frame filled them in: same offsets, same strides, device memory), after waiting for
 * the kernels that write the picture.  NULL for a pointer that belongs to no frame of this decoder or a decoder without device. */
PUBLIC void *e264front_device_samples(Edge264Decoder *dec, const void *samples)
{
	E264Emitter *e = emitter_of(dec);
	size_t off;
	if (!e || !samples || !ON_DEVICE(e)) return NULL;
	const int slot = e264_locate(e, (const uint8_t *)samples, &off);
	if (slot < 0 || hip.frame_wait(e->hip_stream, slot)) return NULL;
	uint8_t *base = hip.frame_device_ptr(e->hip_stream, slot);
	return base ? base + off : NULL;
}

/* DPB slot a sample pointer of Edge264Frame belongs to (samples[] = base view, samples_mvc[] = second view):
 * return_arg only carries the union of both slots (edge264.c:389,398). */
PUBLIC int e264front_slot_of(Edge264Decoder *dec, const void *samples)
{
	E264Emitter *e = emitter_of(dec);
	size_t off;
	return e && samples ? e264_locate(e, (const uint8_t *)samples, &off) : -1;
}
