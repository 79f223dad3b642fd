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

typedef struct {
	uint8_t *samples;  /* host mirror handed to the reference as samples_buffers[slot] */
	size_t samples_size;
	void *mbs;         /* what the reference sees as mb_buffers[slot] */
	void *mbs_base;    /* start of the allocation (guard bands on both sides of mbs) */
	void *user_mbs;    /* caller allocators: the mbs block the caller's alloc_cb returned (held, handed back to its free_cb) */
} E264Slot;

typedef struct {
	int active;
	int width_mbs, height_mbs, n_mbs;
	int frame_id;
	E264Mb *mbs;
	E264Motion *motion;
	uint16_t *dbk_slice;  /* per macroblock: slice entry whose task called deblock_mb on it (0xffff: none yet) */
	uint8_t *state;       /* per macroblock, pictures sent in several packets (a slice failed): E264_ST_* */
	uint8_t *fedges;      /* per macroblock: mb->filter_edges as deblock_mb found it (the emitter clears it like the reference) */
	int multi;            /* a packet of this picture has already been sent, or a macroblock was decoded twice */
	uint8_t *mot;         /* scratch of e264_finish_frame: the compact motion records */
	size_t mot_cap;
	E264SliceParams *slices;
	int *slice_serial;  /* decode_NAL serial of each slice entry */
	uint8_t *slice_filled;
	int n_slices, cap_slices;
	uint8_t *payload;
	size_t payload_len, payload_cap;
	int n_inter;
} E264FrameBuilder;

#define E264_ST_RECON 1   /* reconstructed by an earlier packet of the picture */
#define E264_ST_DBK   2   /* deblocked by an earlier packet */
#define E264_ST_ERR   4   /* marked erroneous by recover_slice (recovery_bits = flip + 2): may be decoded again */

typedef struct { /* macroblock being assembled: leaf calls arrive in decoding order */
	int valid, slot, addr, slice;
	int kind, chroma_mode, i16_mode, t8;
	uint8_t modes[16];
	uint32_t coded;
	int serial; /* NAL serial of the slice that staged this macroblock */
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
	/* a slice that fails (src/edge264_headers.c:527-529) */
	int trk_serial, trk_slot, trk_addr; /* last macroblock staged by the current NAL: addresses only grow inside a slice ... */
	int recover_serial;                 /* ... until recover_slice walks it again from first_mb_in_slice (P_Skip / B_Skip
	                                       concealment, src/edge264_headers.c:399-407): a new generation of those macroblocks */
	int failed_serial, failed_slot;     /* set by the unref wrapper: the NAL's slice ended with an error */
	int (*flush_partial)(struct E264Emitter *, int slot); /* sends the picture-so-far as a packet (edge264_hip_frontend.c) */
	/* sink */
	int sink_kind;      /* 0 HIP back end, 1 capture, 2 HIP frames + queued packets (external batcher) */
	void *hip_dev, *hip_stream;
	/* the caller's allocators (edge264.h:42-43), NULL: ours */
	Edge264AllocCb user_alloc;
	Edge264FreeCb user_free;
	void *user_arg;
	/* capture queue */
	struct E264Captured { uint8_t *data; size_t bytes; struct E264Captured *next; } *cap_head, *cap_tail;
} E264Emitter;

static __thread E264Emitter *e264_tls_emitter; /* set by the API wrappers around the reference's decode_NAL */

static inline int16_t e264_sat16(int32_t v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : (int16_t)v; }

/* which DPB slot / plane position does a sample pointer belong to */
static int e264_locate(E264Emitter *e, const uint8_t *p, size_t *off)
{
	for (int s = 0; s < E264_MAX_SLOTS; s++) {
		if (e->slot[s].samples && p >= e->slot[s].samples && p < e->slot[s].samples + e->slot[s].samples_size) {
			*off = (size_t)(p - e->slot[s].samples);
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
			free(b->mbs); free(b->motion); free(b->dbk_slice); free(b->state); free(b->fedges);
			b->dbk_slice = malloc(sizeof(uint16_t) * (size_t)(w * h));
			b->state = malloc((size_t)(w * h));
			b->fedges = malloc((size_t)(w * h));
			b->mbs = malloc(sizeof(E264Mb) * (size_t)(w * h));
			b->motion = malloc(sizeof(E264Motion) * (size_t)(w * h));
		}
		b->width_mbs = w; b->height_mbs = h; b->n_mbs = w * h;
		memset(b->mbs, 0, sizeof(E264Mb) * (size_t)b->n_mbs);
		memset(b->motion, 0, sizeof(E264Motion) * (size_t)b->n_mbs);
		memset(b->dbk_slice, 0xff, sizeof(uint16_t) * (size_t)b->n_mbs);
		memset(b->state, 0, (size_t)b->n_mbs);
		memset(b->fedges, 0, (size_t)b->n_mbs);
		b->multi = 0;
		for (int i = 0; i < b->n_mbs; i++) {
			memset(b->motion[i].refPic, -1, 8);
			memset(b->motion[i].refIdx, -1, 8);
		}
		b->n_slices = 0;
		b->payload_len = 0;
		b->n_inter = 0;
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
	if (b->n_slices == b->cap_slices) {
		b->cap_slices = b->cap_slices ? b->cap_slices * 2 : 8;
		b->slices = realloc(b->slices, sizeof(E264SliceParams) * (size_t)b->cap_slices);
		b->slice_serial = realloc(b->slice_serial, sizeof(int) * (size_t)b->cap_slices);
		b->slice_filled = realloc(b->slice_filled, (size_t)b->cap_slices);
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
static void e264_fill_slice(E264Emitter *e, E264FrameBuilder *b, int idx, const Edge264Context *ctx)
{
	if (b->slice_filled[idx])
		return;
	E264SliceParams *s = &b->slices[idx];
	s->slice_type = ctx->t.slice_type;
	s->weighted_bipred_idc = ctx->t.pps.weighted_bipred_idc;
	s->luma_log2_weight_denom = ctx->t.luma_log2_weight_denom;
	s->chroma_log2_weight_denom = ctx->t.chroma_log2_weight_denom;
	s->FilterOffsetA = ctx->t.FilterOffsetA;
	s->FilterOffsetB = ctx->t.FilterOffsetB;
	s->disable_deblocking_filter_idc = ctx->t.disable_deblocking_filter_idc;
	s->cabac = ctx->t.pps.entropy_coding_mode_flag;
	s->first_mb = ctx->t.first_mb_in_slice;
	memcpy(s->weightScale4x4, ctx->t.pps.weightScale4x4, sizeof(s->weightScale4x4));
	memcpy(s->weightScale8x8, ctx->t.pps.weightScale8x8, sizeof(s->weightScale8x8));
	memcpy(s->explicit_weights, ctx->t.explicit_weights, sizeof(s->explicit_weights));
	memcpy(s->explicit_offsets, ctx->t.explicit_offsets, sizeof(s->explicit_offsets));
	memcpy(s->implicit_weights, ctx->implicit_weights, sizeof(s->implicit_weights));
	b->slice_filled[idx] = 1;
	e->cabac_of_serial = s->cabac;
}

static void e264_payload_append(E264FrameBuilder *b, const void *src, size_t n)
{
	if (b->payload_len + n > b->payload_cap) {
		b->payload_cap = (b->payload_len + n) * 2 + 4096;
		b->payload = realloc(b->payload, b->payload_cap);
	}
	memcpy(b->payload + b->payload_len, src, n);
	b->payload_len += n;
}

static void e264_levels_append(E264FrameBuilder *b, const int16_t *lev, int n, int narrow)
{
	if (!narrow) { e264_payload_append(b, lev, (size_t)n * 2); return; }
	int8_t t[64];
	for (int i = 0; i < n; i++) t[i] = (int8_t)lev[i];
	e264_payload_append(b, t, (size_t)n);
}

/* close the macroblock under assembly: header from the reference's own Edge264Macroblock
 * (src/edge264_internal.h:128-143), payload in the order of include/edge264_cmd.h */
static void e264_flush_mb(E264Emitter *e)
{
	E264MbStage *c = &e->cur;
	if (!c->valid)
		return;
	c->valid = 0;
	E264FrameBuilder *b = &e->fb[c->slot];
	if (!b->active || c->addr >= b->n_mbs)
		return;
	const Edge264Macroblock *M = c->mbptr;
	E264Mb *m = &b->mbs[c->addr];
	memset(m, 0, sizeof(*m));
	m->kind = (uint8_t)c->kind;
	m->qp[0] = M->QP[0]; m->qp[1] = M->QP[1]; m->qp[2] = M->QP[2];
	m->chroma_mode = (uint8_t)c->chroma_mode;
	m->i16_mode = (uint8_t)c->i16_mode;
	m->slice = (uint16_t)c->slice;
	int t8 = M->f.transform_size_8x8_flag;
	m->flags = (uint8_t)((t8 ? E264_MBF_T8x8 : 0) | (M->filter_edges & 1 ? E264_MBF_EDGE_LEFT : 0) |
		(M->filter_edges & 2 ? E264_MBF_EDGE_TOP : 0) | (M->filter_edges ? E264_MBF_DEBLOCK : 0));
	/* bS=2 test reads mb->nC (deblock.c:1093-1108) after the CAVLC 8x8 broadcast (deblock.c:1094-1096),
	 * which the reference only applies to macroblocks it deblocks */
	unsigned nz = 0;
	for (int k = 0; k < 16; k++)
		nz |= (unsigned)(M->nC[k] != 0) << k;
	if (t8 && M->filter_edges && !b->slices[c->slice].cabac)
		for (int q = 0; q < 4; q++)
			if (nz >> (q * 4) & 15)
				nz |= 15u << (q * 4);
	m->nz_mask = (uint16_t)nz;
	if (c->kind == E264_MB_I4x4)
		for (int k = 0; k < 16; k++)
			m->modes[k >> 1] |= (uint8_t)((c->modes[k] & 15) << (4 * (k & 1)));
	else if (c->kind == E264_MB_I8x8)
		for (int q = 0; q < 4; q++)
			m->modes[q] = c->modes[q];
	if (c->kind == E264_MB_INTER) {
		E264Motion *mo = &b->motion[c->addr];
		memcpy(mo->refPic, M->refPic, 8);
		memcpy(mo->refIdx, M->refIdx, 8);
		memcpy(mo->mvs, M->mvs, 128);
		b->n_inter++;
	}
	while (b->payload_len & 7)
		e264_payload_append(b, "\0", 1);
	m->payload_off = (uint32_t)b->payload_len;
	m->coded = c->coded;
	if (c->coded & E264_CODED_LUMA_DC) e264_payload_append(b, c->luma_dc, 32);
	if (c->coded & E264_CODED_CHROMA_DC) e264_payload_append(b, c->chroma_dc, 16);
	/* AC levels as bytes when every one of them fits (E264_MBF_LEV8): nearly always, and half the coefficient payload */
	int narrow = (c->coded & 0xffffff) != 0;
	for (int k = 0; k < 16 && narrow; k++)
		if (c->coded >> (t8 ? k & ~3 : k) & 1)
			for (int i = 0; i < 16; i++) narrow &= c->luma[k][i] >= -128 && c->luma[k][i] <= 127;
	for (int k = 0; k < 8 && narrow; k++)
		if (c->coded >> (16 + k) & 1)
			for (int i = 0; i < 16; i++) narrow &= c->chroma[k][i] >= -128 && c->chroma[k][i] <= 127;
	if (narrow)
		m->flags |= E264_MBF_LEV8;
	if (t8) {
		for (int q = 0; q < 4; q++)
			if (c->coded >> (q * 4) & 1)
				e264_levels_append(b, &c->luma[q * 4][0], 64, narrow);
	} else {
		for (int k = 0; k < 16; k++)
			if (c->coded >> k & 1)
				e264_levels_append(b, c->luma[k], 16, narrow);
	}
	for (int k = 0; k < 8; k++)
		if (c->coded >> (16 + k) & 1)
			e264_levels_append(b, c->chroma[k], 16, narrow);
}

/* make (slot, addr) the macroblock under assembly */
static E264MbStage *e264_touch(E264Emitter *e, int slot, int addr)
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
	if (addr < b->n_mbs && (b->mbs[addr].kind != E264_MB_ABSENT || b->state[addr])) {
		/* decoded before (a slice failed: recover_slice conceals with skips, and recovery_bits let a later copy of the
		 * slice decode the macroblock again, src/edge264_slice.c:1686): a new generation, reconstructed by the next
		 * packet and deblocked only if the reference calls deblock_mb on it again with filter_edges set */
		b->state[addr] = 0;
		b->dbk_slice[addr] = 0xffff;
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
	int mbx = addr % b->width_mbs, mby = addr / b->width_mbs;
	c->mbptr = (Edge264Macroblock *)e->slot[slot].mbs + mbx + mby * (b->width_mbs + 1);
	return c;
}

static E264MbStage *e264_touch_ctx(Edge264Context *ctx)
{
	E264Emitter *e = e264_tls_emitter;
	size_t off;
	int slot = e264_locate(e, ctx->samples_mb[0], &off);
	if (slot < 0)
		return NULL;
	/* the position, not ctx->CurrMbAddr: recover_slice walks mbx / mby / samples_mb back over the failed slice while
	 * CurrMbAddr stays where the error was found (src/edge264_headers.c:297-305, 414-428) */
	E264MbStage *c = e264_touch(e, slot, ctx->mbx + ctx->mby * ctx->t.pic_width_in_mbs);
	if (c)
		e264_fill_slice(e, &e->fb[slot], c->slice, ctx);
	return c;
}

/* for the intra leaves, which only receive a sample pointer (src/edge264_internal.h:1358-1361) */
static E264MbStage *e264_touch_ptr(const uint8_t *p, int *x_in_mb, int *y_in_mb, int *plane)
{
	E264Emitter *e = e264_tls_emitter;
	size_t off;
	int slot = e264_locate(e, p, &off);
	if (slot < 0)
		return NULL;
	Edge264Decoder *dec = e->dec;
	int x, y;
	if (off < (size_t)dec->plane_size_Y) {
		*plane = 0;
		y = (int)(off / (size_t)dec->out.stride_Y); x = (int)(off % (size_t)dec->out.stride_Y);
		*x_in_mb = x & 15; *y_in_mb = y & 15;
		return e264_touch(e, slot, (y >> 4) * dec->sps.pic_width_in_mbs + (x >> 4));
	}
	off -= (size_t)dec->plane_size_Y;
	y = (int)(off / (size_t)dec->out.stride_C); x = (int)(off % (size_t)dec->out.stride_C);
	*plane = 1 + (x >= (dec->out.stride_C >> 1));
	x %= dec->out.stride_C >> 1;
	*x_in_mb = x & 7; *y_in_mb = y & 7;
	return e264_touch(e, slot, (y >> 3) * dec->sps.pic_width_in_mbs + (x >> 3));
}

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
