/* emit_intra.c -- stands in for src/edge264_intra.c: records the (already availability-remapped,
 * src/edge264_slice.c:573-594) internal prediction mode of each block.  The leaves only get a sample
 * pointer (src/edge264_internal.h:1358-1361), so the position is recovered from it. */
#include "edge264_internal.h"
#include "e264_emit.h"

static cold noinline void decode_intra4x4(uint8_t * restrict p, size_t stride, int mode, i16x8 clip)
{
	int x, y, plane;
	E264_PF_BEGIN;
	E264MbStage *c = E264_TOUCH_PTR(p, &x, &y, &plane);
	if (c && plane == 0) {
		c->kind = E264_MB_I4x4;
		c->modes[e264_blk(x, y)] = (uint8_t)mode;
	}
	(void)stride; (void)clip;
	E264_PF_END(E264_PF_INTRA);
}

static cold noinline void decode_intra8x8(uint8_t * restrict p, size_t stride, int mode, i16x8 clip)
{
	int x, y, plane;
	E264_PF_BEGIN;
	E264MbStage *c = E264_TOUCH_PTR(p, &x, &y, &plane);
	if (c && plane == 0) {
		c->kind = E264_MB_I8x8;
		c->modes[e264_blk(x, y) >> 2] = (uint8_t)mode;
	}
	(void)stride; (void)clip;
	E264_PF_END(E264_PF_INTRA);
}

static cold noinline void decode_intra16x16(uint8_t * restrict p, size_t stride, int mode, i16x8 clip)
{
	int x, y, plane;
	E264_PF_BEGIN;
	E264MbStage *c = E264_TOUCH_PTR(p, &x, &y, &plane);
	if (c && plane == 0) {
		c->kind = E264_MB_I16x16;
		c->i16_mode = mode;
	}
	(void)stride; (void)clip;
	E264_PF_END(E264_PF_INTRA);
}

static cold noinline void decode_intraChroma(uint8_t * restrict p, size_t stride, int mode, i16x8 clip)
{
	int x, y, plane;
	E264_PF_BEGIN;
	E264MbStage *c = E264_TOUCH_PTR(p, &x, &y, &plane);
	if (c)
		c->chroma_mode = mode;
	(void)stride; (void)clip;
	E264_PF_END(E264_PF_INTRA);
}
