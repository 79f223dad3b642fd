/* emit_residual.c -- stands in for src/edge264_residual.c inside the reference's front end.
 * Same static signatures (src/edge264_internal.h:1368-1374); instead of dequantising and
 * transforming, each call snapshots ctx->c[] (levels, transposed order) into the command packet
 * and clears it exactly as the reference does (residual.c:122, 248, 359, 461), because the parser
 * relies on a zeroed scratch. */
#include "edge264_internal.h"
#include "e264_emit.h"

/* luma 4x4 block (zig order) a sample pointer inside the current macroblock starts: row by reciprocal (two divisions per coded
 * block were a tenth of the emitters' time; exact for offsets below 16 rows, like e264_touch_ptr) */
static inline int e264_blk_of(const E264MbStage *c, const uint8_t *p, const Edge264Context *ctx)
{
	const size_t d = (size_t)(p - ctx->samples_mb[0]);
	const unsigned y = (unsigned)(((uint64_t)d * e264_tls_emitter->fb[c->slot].recip_sY) >> 32), x = (unsigned)(d - (size_t)y * ctx->t.stride[0]);
	return e264_blk((int)x, (int)y);
}

static noinline void add_idct4x4(Edge264Context *ctx, int iYCbCr, int DCidx, uint8_t *p)
{
	E264_PF_BEGIN;
	E264MbStage *c = E264_TOUCH_CTX(ctx);
	if (c) {
		int16_t *dst;
		if (iYCbCr == 0) {
			int k = DCidx;
			if (k < 0) { /* I_NxN / inter: block position from the sample pointer (slice.c:617-626) */
				k = e264_blk_of(c, p, ctx);
			}
			c->coded |= E264_CODED_LUMA(k);
			dst = c->luma[k];
		} else {
			c->coded |= E264_CODED_CHROMA(DCidx & 7);
			dst = c->chroma[DCidx & 7];
		}
		unsigned wide = 0;
		for (int i = 0; i < 16; i++) {
			dst[i] = e264_sat16(ctx->c[i]);
			wide |= (unsigned)(dst[i] + 128) & 0xff00u; /* outside a signed byte? (E264_MBF_LEV8) */
		}
		c->wide |= wide;
	}
	ctx->c_v[0] = ctx->c_v[1] = ctx->c_v[2] = ctx->c_v[3] = (i8x16){};
	E264_PF_END(E264_PF_LEVELS);
}

static void add_dc4x4(Edge264Context *ctx, int iYCbCr, int DCidx, uint8_t *p)
{
	/* DC-only block: the back end derives it from the DC block of the macroblock (residual.c:174-187) */
	(void)ctx; (void)iYCbCr; (void)DCidx; (void)p;
}

static void add_idct8x8(Edge264Context *ctx, int iYCbCr, uint8_t *p)
{
	E264_PF_BEGIN;
	E264MbStage *c = E264_TOUCH_CTX(ctx);
	if (c && iYCbCr == 0) {
		int k = e264_blk_of(c, p, ctx) & ~3;
		c->coded |= E264_CODED_LUMA(k);
		int16_t *dst = &c->luma[k][0]; /* 64 coefficients span luma[k..k+3] */
		unsigned wide = 0;
		for (int i = 0; i < 64; i++) {
			dst[i] = e264_sat16(ctx->c[i]);
			wide |= (unsigned)(dst[i] + 128) & 0xff00u;
		}
		c->wide |= wide;
	}
	for (int i = 0; i < 16; i++)
		ctx->c_v[i] = (i8x16){};
	E264_PF_END(E264_PF_LEVELS);
}

static void transform_dc4x4(Edge264Context *ctx, int iYCbCr)
{
	E264_PF_BEGIN;
	E264MbStage *c = E264_TOUCH_CTX(ctx);
	if (c && iYCbCr == 0) {
		c->coded |= E264_CODED_LUMA_DC;
		for (int i = 0; i < 16; i++)
			c->luma_dc[i] = e264_sat16(ctx->c[i]);
	}
	ctx->c_v[0] = ctx->c_v[1] = ctx->c_v[2] = ctx->c_v[3] = (i8x16){};
	E264_PF_END(E264_PF_LEVELS);
}

static void transform_dc2x2(Edge264Context *ctx)
{
	E264_PF_BEGIN;
	E264MbStage *c = E264_TOUCH_CTX(ctx);
	if (c) {
		c->coded |= E264_CODED_CHROMA_DC;
		for (int i = 0; i < 8; i++)
			c->chroma_dc[i] = e264_sat16(ctx->c[i]);
	}
	ctx->c_v[0] = ctx->c_v[1] = (i8x16){};
	E264_PF_END(E264_PF_LEVELS);
}
