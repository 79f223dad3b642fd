/*
 * e264_emit.h -- the reference-side half of the drop-in boundary.
 *
 * This header is compiled INSIDE the reference's translation unit (the farm built by
 * edge264_amd/frontend/Makefile makes `#include "edge264_inter.c"` etc. of src/edge264_headers.c:3-8
 * resolve to emit_*.c of this directory), exactly like the reference's own ISA variants
 * are built by recompiling edge264_headers.c (Makefile:322-329).  The four sample-kernel
 * files are replaced by EMITTERS with the same static signatures
 * (src/edge264_internal.h:1349-1374): instead of touching samples they append to the
 * command packet (include/edge264_cmd.h) of the frame being decoded.  Everything that
 * touches the bitstream (CAVLC/CABAC, mvpred, DPB) is the unmodified reference code.
 *
 * Nothing here reconstructs a sample.  See INTEGRATION.md.
 */
#ifndef E264_EMIT_H
#define E264_EMIT_H

#include <stdlib.h>
#include <string.h>
#include "edge264_cmd.h"
#include "edge264_compact.h"

typedef struct {
	uint8_t *samples;  /* host mirror handed to the reference as samples_buffers[slot] */
	size_t samples_size;
	void *mbs;         /* what the reference sees as mb_buffers[slot] */
	void *mbs_base;    /* start of the allocation (guard bands on both sides of mbs) */
	void *user_mbs;    /* caller allocators: the mbs block the caller's alloc_cb returned (held, handed back to its free_cb) */
	int plain_mirror;  /* the mirror is ordinary memory of this library (decode-to-device: nothing is read back), not the back end's page-locked one */
} E264Slot;

typedef struct {
	int active;
	int width_mbs, height_mbs, n_mbs;
	int frame_id;
	E264Mb *mbs;
	struct E264MbSide {   /* per macroblock, beside its record (one array: deblock_mb touches all three for a macroblock of the row above) */
		uint16_t dbk_slice;  /* slice entry whose task called deblock_mb on it (0xffff: none yet) */
		uint8_t state;       /* pictures sent in several packets (a slice failed): E264_ST_* */
		uint8_t fedges;      /* mb->filter_edges as deblock_mb found it (the emitter clears it like the reference) */
	} *side;
	int multi;            /* a packet of this picture has already been sent, or a macroblock was decoded twice */
	int n_flushed;        /* records written by e264_flush_mb since the builder was reset (every macroblock once: no I_PCM to look for) */
	int n_lifted;         /* I_PCM records lifted from the host mirror */
	uint8_t *mot;         /* the compact motion records of the picture, appended macroblock by macroblock (edge264_cmd.h) */
	size_t mot_len, mot_cap;
	uint32_t ref_slots;   /* DPB slots the records refer to (exact while every macroblock is written once: !multi) */
	uint64_t recip_w1;    /* 2^40 / (width_mbs + 1), rounded up: row of a macroblock from its index in the parser's array */
	uint32_t recip_sY, recip_sC; /* 2^32 / stride, rounded up: row of a sample inside a macroblock */
	E264SliceParams *slices;
	int *slice_serial;  /* decode_NAL serial of each slice entry */
	uint8_t *slice_filled;
	int n_slices, cap_slices;
	uint8_t *payload;
	size_t payload_len, payload_cap;
	int n_inter;
	int oom;              /* a realloc failed while the picture was assembled: e264_finish_frame returns ENOMEM instead of a packet */
} E264FrameBuilder;

/* -DE264_EMIT_PROFILE (tools/hostprof): cycle counters around the emitters' own work, printed by edge264_free.  Off in the product. */
#ifdef E264_EMIT_PROFILE
#include <x86intrin.h>
enum { E264_PF_TOUCH, E264_PF_LEVELS, E264_PF_FLUSH, E264_PF_FINISH, E264_PF_DEBLOCK, E264_PF_INTRA, E264_PF_N };
#define E264_PF_BEGIN unsigned long long pf_t0_ = __rdtsc()
#define E264_PF_END(k) do { if (e264_tls_emitter) { e264_tls_emitter->pf[k] += __rdtsc() - pf_t0_; e264_tls_emitter->pf_calls[k]++; } } while (0)
#else
#define E264_PF_BEGIN do {} while (0)
#define E264_PF_END(k) do {} while (0)
#endif

#define E264_ST_RECON 1   /* reconstructed by an earlier packet of the picture */
#define E264_ST_DBK   2   /* deblocked by an earlier packet */
#define E264_ST_ERR   4   /* marked erroneous by recover_slice (recovery_bits = flip + 2): may be decoded again */

typedef struct { /* macroblock being assembled: leaf calls arrive in decoding order */
	int valid, slot, addr, slice;
	int kind, chroma_mode, i16_mode, t8;
	uint8_t modes[16];
	uint32_t coded;
	int serial; /* NAL serial of the slice that staged this macroblock */
	unsigned wide; /* OR of (level + 128) & 0xff00 over the AC levels staged so far: 0 <=> every one fits a signed byte */
	const uint8_t *ybase, *cbase; /* first luma / Cb sample of the macroblock in the host mirror (intra leaves only get a pointer) */
	int16_t luma_dc[16], chroma_dc[8];
	int16_t luma[16][16];   /* 4x4 blocks, or 4 x 64 coefficients of 8x8 blocks (flat view) */
	int16_t chroma[8][16];
	Edge264Macroblock *mbptr;
} E264MbStage;

typedef struct E264Emitter {
	Edge264Decoder *dec;
	E264Slot slot[E264_MAX_SLOTS];
	E264FrameBuilder fb[E264_MAX_SLOTS];
	E264MbStage cur;
	int serial;         /* incremented by the API wrapper before every NAL: one slice per serial */
	int cabac_of_serial;
	int last_slot;      /* e264_locate: the slot of the previous hit */
	/* a slice that fails (src/edge264_headers.c:527-529) */
	int trk_serial, trk_slot, trk_addr; /* last macroblock staged by the current NAL: addresses only grow inside a slice ... */
	int recover_serial;                 /* ... until recover_slice walks it again from first_mb_in_slice (P_Skip / B_Skip
	                                       concealment, src/edge264_headers.c:399-407): a new generation of those macroblocks */
	int failed_serial, failed_slot;     /* set by the unref wrapper: the NAL's slice ended with an error */
	int (*flush_partial)(struct E264Emitter *, int slot); /* sends the picture-so-far as a packet (edge264_hip_frontend.c) */
	/* sink */
	int sink_kind;      /* 0 HIP back end, 1 capture, 2 HIP frames + queued packets (external batcher) */
	int compact;        /* pictures with inter macroblocks leave in the wire form (include/edge264_compact.h) */
	uint8_t *fold_buf;  /* where such a picture's version-4 packet is assembled before it is folded */
	size_t fold_cap;
	void *hip_dev, *hip_stream;
	/* the caller's allocators (edge264.h:42-43), NULL: ours */
	Edge264AllocCb user_alloc;
	Edge264FreeCb user_free;
	void *user_arg;
#ifdef E264_EMIT_PROFILE
	unsigned long long pf[8], pf_calls[8]; /* per decoder (one thread at a time is inside a decoder) */
#endif
	/* capture queue */
	struct E264Captured { uint8_t *data; size_t bytes; struct E264Captured *next; } *cap_head, *cap_tail;
} E264Emitter;

/* set by the API wrappers around the reference's decode_NAL.  ONE variable for the library: the logging variant of the reference's parser is a
 * second translation unit (front_logs.c, as in the reference's own build) whose copies of the leaf emitters must find the same decoder */
extern __thread E264Emitter *e264_tls_emitter __attribute__((visibility("hidden")));

static inline int16_t e264_sat16(int32_t v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : (int16_t)v; }

/* which DPB slot / plane position does a sample pointer belong to */
static int e264_locate(E264Emitter *e, const uint8_t *p, size_t *off)
{
	int s = e->last_slot; /* nearly every call is for the picture being decoded */
	if ((size_t)(p - e->slot[s].samples) < e->slot[s].samples_size && e->slot[s].samples) {
		*off = (size_t)(p - e->slot[s].samples);
		return s;
	}
	for (s = 0; s < E264_MAX_SLOTS; s++) {
		if (e->slot[s].samples && p >= e->slot[s].samples && p < e->slot[s].samples + e->slot[s].samples_size) {
			*off = (size_t)(p - e->slot[s].samples);
			e->last_slot = s;
			return s;
		}
	}
	return -1;
}

static E264FrameBuilder *e264_builder(E264Emitter *e, int slot)
{
	E264FrameBuilder *b = &e->fb[slot];
	Edge264Decoder *dec = e->dec;
	int w = dec->sps.pic_width_in_mbs, h = dec->sps.pic_height_in_mbs;
	/* a new picture in this slot starts from empty records: also when the previous one never completed (a lost slice
	 * leaves next_deblock_addr short of INT_MAX) and the reference reuses the slot */
	if (b->active && (b->width_mbs != w || b->height_mbs != h || b->frame_id != dec->FrameIds[slot]))
		b->active = 0;
	if (!b->active) {
		if (b->n_mbs != w * h) {
			free(b->mbs); free(b->side);
			b->side = malloc(sizeof(*b->side) * (size_t)(w * h));
			b->mbs = malloc(sizeof(E264Mb) * (size_t)(w * h));
		}
		b->width_mbs = w; b->height_mbs = h; b->n_mbs = w * h;
		memset(b->mbs, 0, sizeof(E264Mb) * (size_t)b->n_mbs);
		for (int i = 0; i < b->n_mbs; i++)
			b->side[i] = (struct E264MbSide){0xffff, 0, 0};
		b->multi = 0;
		b->n_flushed = b->n_lifted = 0;
		b->mot_len = 0;
		b->ref_slots = 0;
		b->recip_w1 = ((((uint64_t)1 << 40) + (uint64_t)w) / (uint64_t)(w + 1));
		b->recip_sY = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)dec->out.stride_Y - 1) / (uint64_t)dec->out.stride_Y);
		b->recip_sC = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)dec->out.stride_C - 1) / (uint64_t)dec->out.stride_C);
		b->n_slices = 0;
		b->payload_len = 0;
		b->n_inter = 0;
		b->oom = 0;
		b->frame_id = dec->FrameIds[slot];
		b->active = 1;
	}
	return b;
}

static int e264_slice_index(E264Emitter *e, E264FrameBuilder *b)
{
	for (int i = b->n_slices - 1; i >= 0; i--)
		if (b->slice_serial[i] == e->serial)
			return i;
	if (b->oom) /* the tables stopped growing (below): the picture is dropped anyway, every later slice of it shares the last entry */
		return b->n_slices > 0 ? b->n_slices - 1 : 0;
	if (b->n_slices == b->cap_slices) {
		const int cap = b->cap_slices ? b->cap_slices * 2 : 8;
		E264SliceParams *ns = malloc(sizeof(E264SliceParams) * (size_t)cap);
		int *nser = malloc(sizeof(int) * (size_t)cap);
		uint8_t *nf = malloc((size_t)cap);
		if (!ns || !nser || !nf) { /* out of memory: the picture is dropped (ENOMEM from e264_finish_frame); keep the tables usable */
			free(ns); free(nser); free(nf);
			b->oom = 1;
			if (b->n_slices > 0) return b->n_slices - 1;
			static E264SliceParams spare_slice; static int spare_serial; static uint8_t spare_filled; /* one element each: n_slices stays 1 (b->oom above).  Shared by
			   all decoders, and never written after this point: slice_filled[0] = 1 keeps e264_fill_slice out, nothing else stores through them (they cannot be
			   thread-local: 2 KB more of initial-exec TLS and the library no longer loads with dlopen) */
			b->slices = &spare_slice; b->slice_serial = &spare_serial; b->slice_filled = &spare_filled; /* never freed through these (cap_slices stays 0: see edge264_free) */
			b->slice_filled[0] = 1; b->n_slices = 1;
			return 0;
		}
		if (b->n_slices) {
			memcpy(ns, b->slices, sizeof(E264SliceParams) * (size_t)b->n_slices);
			memcpy(nser, b->slice_serial, sizeof(int) * (size_t)b->n_slices);
			memcpy(nf, b->slice_filled, (size_t)b->n_slices);
		}
		if (b->cap_slices) { free(b->slices); free(b->slice_serial); free(b->slice_filled); }
		b->slices = ns; b->slice_serial = nser; b->slice_filled = nf; b->cap_slices = cap;
	}
	E264SliceParams *s = &b->slices[b->n_slices];
	memset(s, 0, sizeof(*s));
	memset(s->weightScale4x4, 16, sizeof(s->weightScale4x4));
	memset(s->weightScale8x8, 16, sizeof(s->weightScale8x8));
	s->slice_type = 2;
	s->disable_deblocking_filter_idc = 1;
	b->slice_serial[b->n_slices] = e->serial;
	b->slice_filled[b->n_slices] = 0;
	return b->n_slices++;
}

/* what the leaf functions read from the slice: captured at the first
 * context-bearing leaf call of the slice */
/* ... from the slice's task (src/edge264_internal.h Edge264Task: everything but the implicit weights, which initialize_context derives per worker) */
static void e264_fill_slice_task(E264Emitter *e, E264FrameBuilder *b, int idx, const Edge264Task *t)
{
	E264SliceParams *s = &b->slices[idx];
	s->slice_type = t->slice_type;
	s->weighted_bipred_idc = t->pps.weighted_bipred_idc;
	s->luma_log2_weight_denom = t->luma_log2_weight_denom;
	s->chroma_log2_weight_denom = t->chroma_log2_weight_denom;
	s->FilterOffsetA = t->FilterOffsetA;
	s->FilterOffsetB = t->FilterOffsetB;
	s->disable_deblocking_filter_idc = t->disable_deblocking_filter_idc;
	s->cabac = t->pps.entropy_coding_mode_flag;
	s->first_mb = t->first_mb_in_slice;
	memcpy(s->weightScale4x4, t->pps.weightScale4x4, sizeof(s->weightScale4x4));
	memcpy(s->weightScale8x8, t->pps.weightScale8x8, sizeof(s->weightScale8x8));
	memcpy(s->explicit_weights, t->explicit_weights, sizeof(s->explicit_weights));
	memcpy(s->explicit_offsets, t->explicit_offsets, sizeof(s->explicit_offsets));
	b->slice_filled[idx] = 1;
	e->cabac_of_serial = s->cabac;
}
static void e264_fill_slice(E264Emitter *e, E264FrameBuilder *b, int idx, const Edge264Context *ctx)
{
	if (b->slice_filled[idx])
		return;
	e264_fill_slice_task(e, b, idx, &ctx->t);
	/* the entries initialize_context has written (src/edge264_headers.c:231-252: B slices, active references) and decode_inter reads (implicit
	 * weighting only, src/edge264_inter.c:1149); the rest of the worker's table is whatever its stack held and stays zero here */
	if (ctx->t.slice_type == 1 && ctx->t.pps.weighted_bipred_idc == 2)
		for (int i = 0; i < ctx->t.pps.num_ref_idx_active[0] && i < 32; i++)
			memcpy(b->slices[idx].implicit_weights[i], ctx->implicit_weights[i], ctx->t.pps.num_ref_idx_active[1] < 32 ? (size_t)ctx->t.pps.num_ref_idx_active[1] : 32);
}

static void e264_payload_append(E264FrameBuilder *b, const void *src, size_t n)
{
	if (b->payload_len + n > b->payload_cap) {
		const size_t cap = (b->payload_len + n) * 2 + 4096;
		uint8_t *p = realloc(b->payload, cap);
		if (!p) { b->oom = 1; return; }
		b->payload = p; b->payload_cap = cap;
	}
	memcpy(b->payload + b->payload_len, src, n);
	b->payload_len += n;
}
/* room for the largest macroblock payload (816 bytes of levels / 384 of I_PCM + alignment) */
static uint8_t *e264_payload_reserve(E264FrameBuilder *b)
{
	if (b->payload_len + 1024 > b->payload_cap) {
		const size_t cap = (b->payload_len + 1024) * 2 + 65536;
		uint8_t *p = realloc(b->payload, cap);
		if (!p) { b->oom = 1; return NULL; }
		b->payload = p; b->payload_cap = cap;
	}
	size_t pad = (size_t)-(ptrdiff_t)b->payload_len & 7;
	memset(b->payload + b->payload_len, 0, pad);
	b->payload_len += pad;
	return b->payload + b->payload_len;
}

/* mb->refPic / refIdx / mvs (src/edge264_internal.h:139-142) -> the compact record of edge264_cmd.h, same bytes as
 * e264_motion_compact produces from the expanded form (tests replay the packets through the oracle and the kernels, which
 * parse them with the header's own definitions), with the common shapes decided by block compares. */
static uint32_t e264_motion_emit(const int8_t *refPic, const int8_t *refIdx, const int32_t *mvs, uint8_t *rec, uint32_t *mot_hdr, uint32_t *ref_slots)
{
	uint32_t h = 0, n = 0;
	for (int l = 0; l < 2; l++) {
		const int32_t *mv = mvs + l * 16;
		uint32_t rp, ri;
		memcpy(&rp, refPic + l * 4, 4);
		memcpy(&ri, refIdx + l * 4, 4);
		if (rp == 0xffffffffu)
			continue; /* list unused */
		if (!(rp & 0x80u) && rp == (rp & 255u) * 0x01010101u && ri == (ri & 255u) * 0x01010101u && !memcmp(mv, mv + 1, 60)) {
			h |= 15u << (l * 4) | 1u << (8 + l);
			rec[n] = (uint8_t)rp; rec[n + 1] = (uint8_t)ri; rec[n + 2] = rec[n + 3] = 0;
			memcpy(rec + n + 4, mv, 4);
			n += 8;
			*ref_slots |= 1u << (rp & 31u);
			continue;
		}
		for (int q = 0; q < 4; q++) {
			if (refPic[l * 4 + q] < 0)
				continue;
			const int32_t *v = mv + q * 4;
			const uint32_t sub = (v[0] == v[1] && v[2] == v[3]) ? (v[0] == v[2] ? 0 : 1) : (v[0] == v[2] && v[1] == v[3]) ? 2 : 3;
			h |= 1u << (l * 4 + q) | sub << (10 + 2 * (l * 4 + q));
			rec[n] = (uint8_t)refPic[l * 4 + q]; rec[n + 1] = (uint8_t)refIdx[l * 4 + q]; rec[n + 2] = rec[n + 3] = 0;
			*ref_slots |= 1u << (refPic[l * 4 + q] & 31);
			n += 4;
			memcpy(rec + n, v, 4); n += 4;
			if (sub == 1) { memcpy(rec + n, v + 2, 4); n += 4; }
			else if (sub == 2) { memcpy(rec + n, v + 1, 4); n += 4; }
			else if (sub == 3) { memcpy(rec + n, v + 1, 12); n += 12; }
		}
	}
	*mot_hdr = h;
	return n;
}

/* close the macroblock under assembly: header from the reference's own Edge264Macroblock
 * (src/edge264_internal.h:128-143), motion record and payload in the order of include/edge264_cmd.h */
static void e264_flush_mb_(E264Emitter *e);
static void e264_flush_mb(E264Emitter *e)
{
	if (!e->cur.valid)
		return;
	E264_PF_BEGIN;
	e264_flush_mb_(e);
	E264_PF_END(E264_PF_FLUSH);
}
static void e264_flush_mb_(E264Emitter *e)
{
	E264MbStage *c = &e->cur;
	c->valid = 0;
	E264FrameBuilder *b = &e->fb[c->slot];
	if (!b->active || c->addr >= b->n_mbs)
		return;
	const Edge264Macroblock *M = c->mbptr;
	E264Mb *m = &b->mbs[c->addr];
	b->n_flushed += c->kind != E264_MB_ABSENT;
	memset(m, 0, sizeof(*m));
	m->kind = (uint8_t)c->kind;
	m->qp[0] = M->QP[0]; m->qp[1] = M->QP[1]; m->qp[2] = M->QP[2];
	m->chroma_mode = (uint8_t)c->chroma_mode;
	m->i16_mode = (uint8_t)c->i16_mode;
	m->slice = m->dbk_slice = (uint16_t)c->slice; /* dbk_slice: until deblock_mb says otherwise (emit_deblock.c) */
	const int t8 = M->f.transform_size_8x8_flag, fe = M->filter_edges;
	const int narrow = (c->coded & 0xffffff) != 0 && c->wide == 0; /* E264_MBF_LEV8: every AC level fits a signed byte */
	m->flags = (uint8_t)((t8 ? E264_MBF_T8x8 : 0) | (fe & 1 ? E264_MBF_EDGE_LEFT : 0) | (fe & 2 ? E264_MBF_EDGE_TOP : 0) |
		(fe ? E264_MBF_DEBLOCK : 0) | (narrow ? E264_MBF_LEV8 : 0));
	/* bS=2 test reads mb->nC (deblock.c:1093-1108) after the CAVLC 8x8 broadcast (deblock.c:1094-1096),
	 * which the reference only applies to macroblocks it deblocks */
#ifdef __SSE2__
	unsigned nz = (unsigned)__builtin_ia32_pmovmskb128((i8x16)(M->nC_v[0] != (i8x16){})) & 0xffffu;
#else
	unsigned nz = 0;
	for (int k = 0; k < 16; k++)
		nz |= (unsigned)(M->nC[k] != 0) << k;
#endif
	/* (the broadcast itself happens where the reference does it: in deblock_mb, emit_deblock.c -- a macroblock that is decoded again after a failed slice
	 * and NOT deblocked again keeps the raw flags, and its neighbours' bS is computed from those) */
	m->nz_mask = (uint16_t)nz;
	if (c->kind == E264_MB_I4x4)
		for (int k = 0; k < 16; k++)
			m->modes[k >> 1] |= (uint8_t)((c->modes[k] & 15) << (4 * (k & 1)));
	else if (c->kind == E264_MB_I8x8)
		for (int q = 0; q < 4; q++)
			m->modes[q] = c->modes[q];
	if (c->kind == E264_MB_INTER) {
		if (b->mot_len + 160 > b->mot_cap) {
			const size_t cap = b->mot_cap * 2 + (size_t)b->n_mbs * 24 + 4096;
			uint8_t *p = realloc(b->mot, cap);
			if (!p) { b->oom = 1; m->kind = E264_MB_ABSENT; return; }
			b->mot = p; b->mot_cap = cap;
		}
		uint32_t d[2] = {(uint32_t)b->mot_len, 0};
		b->mot_len += e264_motion_emit(M->refPic, M->refIdx, (const int32_t *)(const void *)M->mvs, b->mot + b->mot_len, &d[1], &b->ref_slots);
		memcpy(m->modes, d, 8);
		b->n_inter++;
	}
	m->coded = c->coded;
	if (!c->coded) {
		m->payload_off = (uint32_t)(b->payload_len & ~(size_t)7); /* nothing stored: any aligned offset inside the payload will do */
		return;
	}
	uint8_t *w = e264_payload_reserve(b), *w0 = w;
	if (!w) { m->coded = 0; m->payload_off = 0; return; } /* (b->oom is set: the picture will not be sent) */
	m->payload_off = (uint32_t)b->payload_len;
	if (c->coded & E264_CODED_LUMA_DC) { memcpy(w, c->luma_dc, 32); w += 32; }
	if (c->coded & E264_CODED_CHROMA_DC) { memcpy(w, c->chroma_dc, 16); w += 16; }
#define E264_PUT(src, n) do { if (narrow) { for (int i_ = 0; i_ < (n); i_++) w[i_] = (uint8_t)(int8_t)(src)[i_]; w += (n); } \
		else { memcpy(w, (src), (size_t)(n) * 2); w += (n) * 2; } } while (0)
	if (t8) {
		for (int q = 0; q < 4; q++)
			if (c->coded >> (q * 4) & 1)
				E264_PUT(&c->luma[q * 4][0], 64);
	} else {
		for (unsigned bits = c->coded & 0xffff; bits; bits &= bits - 1)
			E264_PUT(c->luma[__builtin_ctz(bits)], 16);
	}
	for (unsigned bits = c->coded >> 16 & 0xff; bits; bits &= bits - 1)
		E264_PUT(c->chroma[__builtin_ctz(bits)], 16);
#undef E264_PUT
	b->payload_len += (size_t)(w - w0);
}

/* make (slot, addr) the macroblock under assembly */
static E264MbStage *e264_touch_(E264Emitter *e, int slot, int mbx, int mby, int addr);
static E264MbStage *e264_touch(E264Emitter *e, int slot, int mbx, int mby, int width_mbs)
{
	E264_PF_BEGIN;
	E264MbStage *c = e264_touch_(e, slot, mbx, mby, mbx + mby * width_mbs);
	E264_PF_END(E264_PF_TOUCH);
	return c;
}
static E264MbStage *e264_touch_(E264Emitter *e, int slot, int mbx, int mby, int addr)
{
	E264MbStage *c = &e->cur;
	if (c->valid && c->slot == slot && c->addr == addr && c->serial == e->serial)
		return c;
	if (e->trk_serial == e->serial && e->trk_slot == slot && addr <= e->trk_addr && e->recover_serial != e->serial) {
		/* The slice went BACK to a macroblock it has already closed: only recover_slice does that (src/edge264_headers.c:295-305), after
		 * the slice's own deblocking (:499-525).  What was decoded and deblocked up to here is one state of the picture --
		 * it goes out as a packet now --, the concealment that starts here writes the next one. */
		e264_flush_mb(e);
		e->recover_serial = e->serial;
		if (e->flush_partial)
			e->flush_partial(e, slot);
	}
	e->trk_serial = e->serial; e->trk_slot = slot; e->trk_addr = addr;
	if (c->valid && c->slot == slot && c->addr == addr)
		c->valid = 0; /* the same macroblock decoded again by a later NAL (a slice that failed and is resent): start over */
	e264_flush_mb(e);
	E264FrameBuilder *b = e264_builder(e, slot);
	if (addr < b->n_mbs && (b->mbs[addr].kind != E264_MB_ABSENT || b->side[addr].state)) {
		/* decoded before (a slice failed: recover_slice conceals with skips, and recovery_bits let a later copy of the
		 * slice decode the macroblock again, src/edge264_slice.c:1686): a new generation, reconstructed by the next
		 * packet and deblocked only if the reference calls deblock_mb on it again with filter_edges set */
		b->side[addr].state = 0;
		b->side[addr].dbk_slice = 0xffff;
		b->mbs[addr].kind = E264_MB_ABSENT;
		b->multi = 1;
	}
	memset(c, 0, offsetof(E264MbStage, luma_dc));
	c->valid = 1;
	c->slot = slot;
	c->addr = addr;
	c->serial = e->serial;
	c->slice = e264_slice_index(e, b);
	c->kind = E264_MB_ABSENT;
	c->mbptr = (Edge264Macroblock *)e->slot[slot].mbs + mbx + mby * (b->width_mbs + 1); /* (no division: the callers know the column and the row) */
	return c;
}

static E264MbStage *e264_touch_ctx(Edge264Context *ctx)
{
	E264Emitter *e = e264_tls_emitter;
	E264MbStage *c = &e->cur;
	if (c->valid && c->mbptr == ctx->_mb && c->serial == e->serial) { /* the macroblock under assembly (most calls: one per coded block) */
		/* ... which an intra leaf may have opened from a bare sample pointer: this is then the first call of the slice that sees its constants.
		 * (Round 5, found by tools/stream_sweep.py: an I slice with disable_deblocking_filter_idc 1 -- deblock_mb returns before it looks at the
		 * slice -- kept the default flat scaling lists whatever the parameter sets said.) */
		E264FrameBuilder *b = &e->fb[c->slot];
		if (__builtin_expect(!b->slice_filled[c->slice], 0))
			e264_fill_slice(e, b, c->slice, ctx);
		return c;
	}
	size_t off;
	int slot = e264_locate(e, ctx->samples_mb[0], &off);
	if (slot < 0)
		return NULL;
	/* the position, not ctx->CurrMbAddr: recover_slice walks mbx / mby / samples_mb back over the failed slice while
	 * CurrMbAddr stays where the error was found (src/edge264_headers.c:297-305, 414-428) */
	c = e264_touch(e, slot, ctx->mbx, ctx->mby, ctx->t.pic_width_in_mbs);
	if (c)
		e264_fill_slice(e, &e->fb[slot], c->slice, ctx);
	return c;
}

/* for the intra leaves, which only receive a sample pointer (src/edge264_internal.h:1358-1361) */
static E264MbStage *e264_touch_ptr(const uint8_t *p, int *x_in_mb, int *y_in_mb, int *plane)
{
	E264Emitter *e = e264_tls_emitter;
	Edge264Decoder *dec = e->dec;
	E264MbStage *c = &e->cur;
	if (c->valid && c->serial == e->serial && c->ybase) { /* inside the macroblock under assembly?  (no division: rows by reciprocal) */
		const E264FrameBuilder *b = &e->fb[c->slot];
		size_t d = (size_t)(p - c->ybase);
		if (d < (size_t)dec->out.stride_Y * 16) {
			unsigned y = (unsigned)(((uint64_t)d * b->recip_sY) >> 32), x = (unsigned)(d - (size_t)y * dec->out.stride_Y);
			if (x < 16) { *plane = 0; *x_in_mb = (int)x; *y_in_mb = (int)y; return c; }
		}
		d = (size_t)(p - c->cbase);
		if (d < (size_t)dec->out.stride_C * 8) {
			unsigned y = (unsigned)(((uint64_t)d * b->recip_sC) >> 32), x = (unsigned)(d - (size_t)y * dec->out.stride_C);
			const unsigned half = (unsigned)dec->out.stride_C >> 1;
			if (x < 8) { *plane = 1; *x_in_mb = (int)x; *y_in_mb = (int)y; return c; }
			if (x - half < 8) { *plane = 2; *x_in_mb = (int)(x - half); *y_in_mb = (int)y; return c; }
		}
	}
	size_t off;
	int slot = e264_locate(e, p, &off);
	if (slot < 0)
		return NULL;
	int x, y, mbx, mby;
	if (off < (size_t)dec->plane_size_Y) {
		*plane = 0;
		y = (int)(off / (size_t)dec->out.stride_Y); x = (int)(off % (size_t)dec->out.stride_Y);
		*x_in_mb = x & 15; *y_in_mb = y & 15;
		mbx = x >> 4; mby = y >> 4;
	} else {
		off -= (size_t)dec->plane_size_Y;
		y = (int)(off / (size_t)dec->out.stride_C); x = (int)(off % (size_t)dec->out.stride_C);
		*plane = 1 + (x >= (dec->out.stride_C >> 1));
		x %= dec->out.stride_C >> 1;
		*x_in_mb = x & 7; *y_in_mb = y & 7;
		mbx = x >> 3; mby = y >> 3;
	}
	c = e264_touch(e, slot, mbx, mby, dec->sps.pic_width_in_mbs);
	if (c && !c->ybase) {
		c->ybase = e->slot[slot].samples + (size_t)(mby * 16) * dec->out.stride_Y + mbx * 16;
		c->cbase = e->slot[slot].samples + dec->plane_size_Y + (size_t)(mby * 8) * dec->out.stride_C + mbx * 8;
	}
	return c;
}

/* tools/hostprof builds this translation unit with -DE264_NULL_LEAVES: every leaf returns after the clearing the parser relies
 * on, nothing is located, staged or emitted -- the parse floor the emitters' cost is measured against.  Never set in the product. */
#ifdef E264_NULL_LEAVES
#define E264_NULL_LEAF 1
#define E264_TOUCH_CTX(ctx) ((void)(ctx), (E264MbStage *)NULL)
#define E264_TOUCH_PTR(p, x, y, pl) ((void)(p), *(x) = *(y) = *(pl) = 0, (E264MbStage *)NULL)
#else
#define E264_NULL_LEAF 0
#define E264_TOUCH_CTX(ctx) e264_touch_ctx(ctx)
#define E264_TOUCH_PTR(p, x, y, pl) e264_touch_ptr(p, x, y, pl)
#endif

/* ---- helpers referenced by the reference's error concealment (recover_slice, src/edge264_headers.c:295-430),
 * which lived in the kernel files we replace.  What recover_slice writes (a blend with the neighbours' DC for I slices, on
 * the host mirror; P_Skip / B_Skip for P / B slices, through decode_inter) never reaches a picture the API hands out: the
 * picture of a failed slice stays incomplete unless the slice arrives again, and then every concealed macroblock is
 * decoded again on top -- except macroblocks of OTHER slices that the failed slice ran over, which keep their concealed
 * (skip) version.  The emitters reproduce every state of the picture that can survive: the picture
 * as decoded and deblocked up to the failure goes out as a packet of its own, the P_Skip / B_Skip concealment is the next
 * generation of its macroblocks (e264_touch), and from then on the picture is sent NAL by NAL
 * (edge264_hip_frontend.c, e264_finish_frame with partial = 1).  The I-slice blend happens on the host mirror only: every
 * blended macroblock is decoded again before the picture can complete. */
static inline i8x16 ldleftC(const uint8_t *p, size_t stride, size_t mstride)
{ /* left neighbours of two 8-row planes interleaved row by row: even rows first, then odd rows */
	i8x16 v;
	const uint8_t *q = p - 1;
	for (int j = 0; j < 8; j++) {
		size_t extra = j >= 4 ? mstride : 0;
		v[j] = (int8_t)q[(size_t)(2 * j) * stride + extra];
		v[8 + j] = (int8_t)q[(size_t)(2 * j + 1) * stride + extra];
	}
	return v;
}
static inline i8x16 maddshrL(i8x16 q, i8x16 p, i8x16 w0, i8x16 w1, i16x8 o, i64x2 wd)
{ /* clip255((sat16(sat16(q*w[0] + p*w[1]) + o)) >> wd) per byte */
	i8x16 r;
	(void)w1;
	for (int i = 0; i < 16; i++) {
		int x = (uint8_t)q[i] * w0[0] + (uint8_t)p[i] * w0[1];
		x = x < -32768 ? -32768 : x > 32767 ? 32767 : x;
		x += o[i & 7];
		x = x < -32768 ? -32768 : x > 32767 ? 32767 : x;
		x >>= wd[0];
		r[i] = (int8_t)(x < 0 ? 0 : x > 255 ? 255 : x);
	}
	return r;
}

static inline int e264_blk(int x, int y) { return (y >> 3) * 8 + (x >> 3) * 4 + ((y >> 2) & 1) * 2 + ((x >> 2) & 1); }

#endif
