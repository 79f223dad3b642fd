/* emit_inter.c -- stands in for src/edge264_inter.c.  decode_inter(ctx,i,w,h) is called once per
 * partition and list (src/edge264_slice.c:1263,1536; src/edge264_mvpred.c:73-513) after the host has
 * written the final vectors and references into mb->mvs / mb->refPic / mb->refIdx; those arrays are
 * copied into the packet when the macroblock is closed, so the call itself only marks the
 * macroblock as inter. */
#include "edge264_internal.h"
#include "e264_emit.h"

static void noinline decode_inter(Edge264Context *ctx, int i, int w, int h)
{
	E264MbStage *c = E264_TOUCH_CTX(ctx);
	if (c)
		c->kind = E264_MB_INTER;
	(void)i; (void)w; (void)h;
}
