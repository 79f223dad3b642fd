/* emit_deblock.c -- stands in for src/edge264_deblock.c.  bS, alpha/beta/tC0 and the filters all run
 * on the device from the per-macroblock metadata of the packet; the call is only used to capture the
 * slice constants of slices that issue no other context-bearing leaf call. */
#include "edge264_internal.h"
#include "e264_emit.h"

static noinline void deblock_mb(Edge264Context *ctx)
{
	E264Emitter *e = e264_tls_emitter;
	size_t off;
	int slot = e264_locate(e, ctx->samples_mb[0], &off);
	if (slot >= 0 && e->fb[slot].active) {
		E264FrameBuilder *b = &e->fb[slot];
		e264_fill_slice(e, b, e264_slice_index(e, b), ctx);
	}
}
