/*
 * ref_kernels_harness.c -- TEST INFRASTRUCTURE, built only where /root/reference
 * exists (oracle/Makefile -> oracle/_ref/libe264_refkernels.so).
 *
 * It compiles the REFERENCE's own static sample kernels straight from
 * /root/reference/src (the same way the reference's unit test does,
 * src/edge264_check.c:20-22) and drives them with the command packet of
 * include/edge264_cmd.h: every leaf call the reference front end would issue
 * for a macroblock (SURVEY.md 8b "the actual seam") is re-issued here on a
 * hand-built Edge264Context.  The result is what the reference computes for
 * that packet, and is the yardstick for oracle/e264_oracle.c and for the HIP
 * kernels.  No reference source text lives in this repository.
 */
#include "edge264_internal.h"
#include "edge264_intra.c"
#include "edge264_inter.c"
#include "edge264_residual.c"
#include "edge264_deblock.c"

#include "../include/edge264_cmd.h"

#define EXPORT __attribute__((visibility("default")))

typedef struct {
	Edge264Context c;
	Edge264Macroblock *mbs; /* (W+1)*H entries, sentinel column as in headers.c:114-125 */
	int W, H;
} RefHarness;

static int g_force4x4;
EXPORT void ref_force_4x4_calls(int on) { g_force4x4 = on; }
EXPORT int ref_sizeof_macroblock(void) { return (int)sizeof(Edge264Macroblock); }

EXPORT RefHarness *ref_new(int width_mbs, int height_mbs)
{
	RefHarness *h = aligned_alloc(64, (sizeof(*h) + 63) & ~(size_t)63);
	memset(h, 0, sizeof(*h));
	h->W = width_mbs;
	h->H = height_mbs;
	size_t n = (size_t)(width_mbs + 1) * (height_mbs + 1) + 2;
	h->mbs = aligned_alloc(64, ((n * sizeof(Edge264Macroblock)) + 63) & ~(size_t)63);
	memset(h->mbs, 0, n * sizeof(Edge264Macroblock));
	return h;
}

EXPORT void ref_free(RefHarness *h)
{
	if (h) {
		free(h->mbs);
		free(h);
	}
}

static Edge264Macroblock *mb_at(RefHarness *h, int mbx, int mby)
{ /* one spare row above so that mb-1-W of row 0 stays inside the allocation */
	return h->mbs + (h->W + 1) + 1 + mbx + mby * (h->W + 1);
}

static void set_pos(RefHarness *h, const E264FrameHdr *fh, uint8_t *cur, int mbx, int mby)
{
	Edge264Context *ctx = &h->c;
	ctx->mbx = mbx;
	ctx->mby = mby;
	ctx->CurrMbAddr = mby * h->W + mbx;
	ctx->samples_mb[0] = cur + (mbx + mby * fh->stride_Y) * 16;
	ctx->samples_mb[1] = cur + (mbx + mby * fh->stride_C) * 8 + fh->plane_size_Y;
	ctx->samples_mb[2] = ctx->samples_mb[1] + (fh->stride_C >> 1);
	ctx->_mb = mb_at(h, mbx, mby);
}

static void load_slice(Edge264Context *ctx, const E264SliceParams *s)
{
	ctx->t.slice_type = s->slice_type;
	ctx->t.pps.weighted_bipred_idc = s->weighted_bipred_idc;
	ctx->t.luma_log2_weight_denom = s->luma_log2_weight_denom;
	ctx->t.chroma_log2_weight_denom = s->chroma_log2_weight_denom;
	ctx->t.FilterOffsetA = s->FilterOffsetA;
	ctx->t.FilterOffsetB = s->FilterOffsetB;
	ctx->t.disable_deblocking_filter_idc = s->disable_deblocking_filter_idc;
	ctx->t.pps.entropy_coding_mode_flag = 1; /* nz_mask already carries the CAVLC 8x8 fix-up (deblock.c:1094-1096) */
	memcpy(ctx->t.pps.weightScale4x4, s->weightScale4x4, sizeof(s->weightScale4x4));
	memcpy(ctx->t.pps.weightScale8x8, s->weightScale8x8, sizeof(s->weightScale8x8));
	memcpy(ctx->t.explicit_weights, s->explicit_weights, sizeof(s->explicit_weights));
	memcpy(ctx->t.explicit_offsets, s->explicit_offsets, sizeof(s->explicit_offsets));
	memcpy(ctx->implicit_weights, s->implicit_weights, sizeof(s->implicit_weights));
}

static void load_coeffs(Edge264Context *ctx, const int16_t *src, int n)
{
	for (int i = 0; i < n; i++)
		ctx->c[i] = src[i];
}

/* Replays one packet with the reference's kernels.  passes: bit0 recon, bit1 deblock. */
EXPORT int ref_replay_packet(RefHarness *h, const uint8_t *pkt, size_t bytes, uint8_t *const *dpb, int passes)
{
	const E264FrameHdr *fh = (const E264FrameHdr *)pkt;
	if (bytes < sizeof(*fh) || fh->magic != E264_MAGIC || fh->width_mbs != h->W || fh->height_mbs != h->H)
		return -1;
	const E264SliceParams *slices = (const E264SliceParams *)(pkt + fh->slices_off);
	const E264Mb *mbs = (const E264Mb *)(pkt + fh->mbs_off);
	const uint8_t *payload = pkt + fh->payload_off;
	const uint8_t *motion_sec = fh->motion_off ? pkt + fh->motion_off : NULL; /* compact records, expanded per macroblock below */
	Edge264Context *ctx = &h->c;
	uint8_t *cur = dpb[fh->dst_slot];
	ctx->t.pic_width_in_mbs = h->W;
	ctx->t.pic_height_in_mbs = h->H;
	ctx->t.stride[0] = fh->stride_Y;
	ctx->t.stride[1] = ctx->t.stride[2] = fh->stride_C;
	ctx->t.plane_size_Y = fh->plane_size_Y;
	ctx->t.plane_size_C = fh->plane_size_C;
	ctx->t.ChromaArrayType = 1;
	for (int i = 0; i < 32; i++)
		ctx->t.samples_buffers[i] = dpb[i];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 8; j++)
			ctx->t.samples_clip[i][j] = 255;
	ctx->t.mb_buffer = mb_at(h, 0, 0);
	memset(ctx->c, 0, sizeof(ctx->c));

	/* metadata of every macroblock first (deblocking and bS look at neighbours) */
	for (int mby = 0; mby < h->H; mby++) {
		for (int mbx = 0; mbx < h->W; mbx++) {
			const E264Mb *m = mbs + mby * h->W + mbx;
			Edge264Macroblock *M = mb_at(h, mbx, mby);
			memset(M, 0, sizeof(*M));
			M->mbIsInterFlag = m->kind == E264_MB_INTER;
			M->filter_edges = !(m->flags & E264_MBF_DEBLOCK) ? 0 :
				4 | (m->flags & E264_MBF_EDGE_LEFT ? 1 : 0) | (m->flags & E264_MBF_EDGE_TOP ? 2 : 0);
			M->QP[0] = m->qp[0]; M->QP[1] = m->qp[1]; M->QP[2] = m->qp[2];
			M->f.transform_size_8x8_flag = (m->flags & E264_MBF_T8x8) != 0;
			M->f.CodedBlockPatternChromaDC = (m->coded & E264_CODED_CHROMA_DC) != 0;
			M->f.CodedBlockPatternChromaAC = (m->coded >> 16 & 0xff) != 0;
			M->bits[0] = (m->kind == E264_MB_I16x16 && (m->coded & 0xffff)) ? 0xac : 0;
			for (int k = 0; k < 16; k++)
				M->nC[k] = m->nz_mask >> k & 1;
			M->refIdx_l = -1;
			M->refPic_l = -1;
			if (m->kind == E264_MB_INTER) {
				E264Motion mo_x, *mo = &mo_x;
				uint32_t mdir[2];
				memcpy(mdir, m->modes, 8);
				e264_motion_expand(mdir[1], motion_sec + mdir[0], mo);
				memcpy(M->refPic, mo->refPic, 8);
				memcpy(M->refIdx, mo->refIdx, 8);
				memcpy(M->mvs, mo->mvs, 128);
			}
		}
	}

	if (passes & 1) {
		for (int mby = 0; mby < h->H; mby++) {
			for (int mbx = 0; mbx < h->W; mbx++) {
				const E264Mb *m = mbs + mby * h->W + mbx;
				if (m->kind == E264_MB_ABSENT)
					continue;
				const E264SliceParams *s = slices + m->slice;
				load_slice(ctx, s);
				set_pos(h, fh, cur, mbx, mby);
				ctx->t.QP[0] = m->qp[0]; ctx->t.QP[1] = m->qp[1]; ctx->t.QP[2] = m->qp[2];
				const uint8_t *pl = payload + m->payload_off;
				if (m->kind == E264_MB_PCM) {
					for (int y = 0; y < 16; y++) memcpy(ctx->samples_mb[0] + y * fh->stride_Y, pl + y * 16, 16);
					for (int y = 0; y < 8; y++) {
						memcpy(ctx->samples_mb[1] + y * fh->stride_C, pl + 256 + y * 8, 8);
						memcpy(ctx->samples_mb[2] + y * fh->stride_C, pl + 320 + y * 8, 8);
					}
					continue;
				}
				const int16_t *ldc = NULL, *cdc = NULL;
				if (m->coded & E264_CODED_LUMA_DC) { ldc = (const int16_t *)pl; pl += 32; }
				if (m->coded & E264_CODED_CHROMA_DC) { cdc = (const int16_t *)pl; pl += 16; }
				const int16_t *co = (const int16_t *)pl;
				int16_t wide[384]; /* E264_MBF_LEV8: the AC blocks arrive as bytes (edge264_cmd.h) */
				if (m->flags & E264_MBF_LEV8) {
					E264Mb dense = *m;
					dense.flags &= (uint8_t)~E264_MBF_LEV8;
					dense.coded &= ~(E264_CODED_LUMA_DC | E264_CODED_CHROMA_DC);
					int n = (int)e264_mb_payload_bytes(&dense) / 2;
					for (int i = 0; i < n; i++) wide[i] = (int8_t)pl[i];
					co = wide;
				}
				if (m->flags & E264_MBF_DONE)
					continue;
				size_t sY = fh->stride_Y;
				i16x8 clip = ctx->t.samples_clip_v[0];

				if (m->kind == E264_MB_INTER) {
					/* Largest partitions with uniform motion, list 0 then list 1, as the front end
					 * would call them (slice.c:1226-1264, 1503-1537; mvpred.c:73-513).  The result does
					 * not depend on the partitioning, only the call count (CPU baseline fairness). */
					for (int list = 0; list < 2; list++) {
						const int32_t *mv = mb->mvs_s + list * 16;
						const int8_t *rp = mb->refPic + list * 4, *ri = mb->refIdx + list * 4;
						int uni8[4], all = 1;
						for (int b = 0; b < 4; b++) {
							uni8[b] = mv[b * 4] == mv[b * 4 + 1] && mv[b * 4] == mv[b * 4 + 2] && mv[b * 4] == mv[b * 4 + 3];
							all &= uni8[b] && mv[b * 4] == mv[0] && rp[b] == rp[0] && ri[b] == ri[0]
								&& mb->refIdx[(list ^ 1) * 4 + b] == mb->refIdx[(list ^ 1) * 4];
						}
						if (g_force4x4) all = uni8[0] = uni8[1] = uni8[2] = uni8[3] = 0;
						if (all) {
							if (rp[0] >= 0)
								decode_inter(ctx, list * 16, 16, 16);
							continue;
						}
						for (int b = 0; b < 4; b++) {
							if (rp[b] < 0)
								continue;
							if (uni8[b]) {
								decode_inter(ctx, list * 16 + b * 4, 8, 8);
							} else {
								for (int k = b * 4; k < b * 4 + 4; k++)
									decode_inter(ctx, list * 16 + k, 4, 4);
							}
						}
					}
				}
				if (m->kind == E264_MB_I16x16) {
					decode_intra16x16(ctx->samples_mb[0], sY, m->i16_mode, clip);
					if (ldc) {
						load_coeffs(ctx, ldc, 16);
						transform_dc4x4(ctx, 0);
					}
					if (m->coded & 0xffff) {
						for (int k = 0; k < 16; k++) {
							uint8_t *p = ctx->samples_mb[0] + y444[k] * sY + x444[k];
							if (m->coded & E264_CODED_LUMA(k)) {
								load_coeffs(ctx, co, 16); co += 16;
								add_idct4x4(ctx, 0, k, p);
							} else {
								add_dc4x4(ctx, 0, k, p);
							}
						}
					}
					memset(ctx->c + 16, 0, 16 * sizeof(int32_t));
				} else if (!(m->flags & E264_MBF_T8x8)) {
					for (int k = 0; k < 16; k++) {
						uint8_t *p = ctx->samples_mb[0] + y444[k] * sY + x444[k];
						if (m->kind == E264_MB_I4x4)
							decode_intra4x4(p, sY, m->modes[k >> 1] >> (4 * (k & 1)) & 15, clip);
						if (m->coded & E264_CODED_LUMA(k)) {
							load_coeffs(ctx, co, 16); co += 16;
							add_idct4x4(ctx, 0, -1, p);
						}
					}
				} else {
					for (int b = 0; b < 4; b++) {
						uint8_t *p = ctx->samples_mb[0] + y444[b * 4] * sY + x444[b * 4];
						if (m->kind == E264_MB_I8x8)
							decode_intra8x8(p, sY, m->modes[b], clip);
						if (m->coded & E264_CODED_LUMA(b * 4)) {
							load_coeffs(ctx, co, 64); co += 64;
							add_idct8x8(ctx, 0, p);
						}
					}
				}
				if (m->kind != E264_MB_INTER)
					decode_intraChroma(ctx->samples_mb[1], fh->stride_C >> 1, m->chroma_mode, clip);
				if (cdc) {
					load_coeffs(ctx, cdc, 8);
					transform_dc2x2(ctx);
					if (mb->f.CodedBlockPatternChromaAC) {
						for (int k = 0; k < 8; k++) {
							int iYCbCr = 1 + (k >> 2);
							uint8_t *p = ctx->samples_mb[iYCbCr] + y420[k] * fh->stride_C + x420[k];
							if (m->coded & E264_CODED_CHROMA(k)) {
								load_coeffs(ctx, co, 16); co += 16;
								add_idct4x4(ctx, iYCbCr, k, p);
							} else {
								add_dc4x4(ctx, iYCbCr, k, p);
							}
						}
					}
					memset(ctx->c + 16, 0, 8 * sizeof(int32_t));
				}
			}
		}
	}

	if (passes & 2) {
		for (int mby = 0; mby < h->H; mby++) {
			for (int mbx = 0; mbx < h->W; mbx++) {
				const E264Mb *m = mbs + mby * h->W + mbx;
				if (m->kind == E264_MB_ABSENT)
					continue;
				load_slice(ctx, slices + m->slice);
				set_pos(h, fh, cur, mbx, mby);
				deblock_mb(ctx);
			}
		}
	}
	return 0;
}

/* thin single-function entry points (golden vectors of src/edge264_check.c use these signatures) */
EXPORT void ref_intra4x4(uint8_t *p, size_t stride, int mode) { decode_intra4x4(p, stride, mode, (i16x8){255, 255, 255, 255, 255, 255, 255, 255}); }
EXPORT void ref_intra8x8(uint8_t *p, size_t stride, int mode) { decode_intra8x8(p, stride, mode, (i16x8){255, 255, 255, 255, 255, 255, 255, 255}); }
EXPORT void ref_intra16x16(uint8_t *p, size_t stride, int mode) { decode_intra16x16(p, stride, mode, (i16x8){255, 255, 255, 255, 255, 255, 255, 255}); }
EXPORT void ref_intra_chroma(uint8_t *p, size_t half_stride, int mode) { decode_intraChroma(p, half_stride, mode, (i16x8){255, 255, 255, 255, 255, 255, 255, 255}); }
EXPORT void ref_inter_luma(int mode, int h, size_t sstride, const uint8_t *src, size_t dstride, uint8_t *dst)
{
	decode_inter_luma(mode, h, sstride, src, dstride, dst, (i16x8)(i8x16){0, 1});
}
EXPORT void ref_inter_chroma(int w, int h, size_t sstride, const uint8_t *src, size_t dstride, uint8_t *dst, int A, int B, int C, int D)
{
	i8x16 ABCD = {A, B, C, D};
	i8x16 wod = {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0};
	decode_inter_chroma(w, h, sstride, src, dstride, dst, ABCD, (i16x8)wod);
}
