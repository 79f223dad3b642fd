/* emit_deblock.c -- stands in for src/edge264_deblock.c.  bS, alpha/beta/tC0 and the filters all run
 * on the device from the per-macroblock metadata of the packet.  The call captures (a) the slice
 * constants of slices that issue no other context-bearing leaf call and (b) WHICH slice's task
 * deblocks the macroblock: the reference filters with ctx->t.FilterOffsetA/B of the calling task
 * (src/edge264_deblock.c:945-955), and for macroblocks whose deblocking was deferred (arbitrary slice
 * order, src/edge264_headers.c:538-567) that is the task completing the picture, not the
 * macroblock's own slice -> E264Mb.dbk_slice. */
#include "edge264_internal.h"
#include "e264_emit.h"

static noinline void deblock_mb(Edge264Context *ctx)
{
	E264Emitter *e = e264_tls_emitter;
	size_t off;
	int slot = e264_locate(e, ctx->samples_mb[0], &off);
	if (slot >= 0 && e->fb[slot].active) {
		E264FrameBuilder *b = &e->fb[slot];
		int idx = e264_slice_index(e, b);
		e264_fill_slice(e, b, idx, ctx);
		size_t px = off >> 4, stride = (size_t)ctx->t.stride[0]; /* samples_mb[0] = base + (mbx + mby * stride) * 16 */
		size_t mby = px / stride, mbx = px % stride;
		/* the first call is the one that filters: the reference clears mb->filter_edges afterwards (deblock.c:500) and the
		 * picture-completing pass runs over macroblocks that were deblocked earlier without touching them */
		if (mbx < (size_t)b->width_mbs && mby < (size_t)b->height_mbs && b->dbk_slice[mby * (size_t)b->width_mbs + mbx] == 0xffff)
			b->dbk_slice[mby * (size_t)b->width_mbs + mbx] = (uint16_t)idx;
	}
}
