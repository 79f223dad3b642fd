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
	E264_PF_BEGIN;
	if (!mb->filter_edges) /* src/edge264_deblock.c:938: already filtered (or never to be) */
		return;
	size_t off;
	int slot = E264_NULL_LEAF ? -1 : e264_locate(e, ctx->samples_mb[0], &off);
	if (slot >= 0 && e->fb[slot].active) {
		E264FrameBuilder *b = &e->fb[slot];
		if (e->cur.valid && e->cur.slot == slot)
			e264_flush_mb(e); /* the record takes its flags from filter_edges, which is cleared below */
		int idx = e264_slice_index(e, b);
		e264_fill_slice(e, b, idx, ctx);
		/* which macroblock: from its record's place in the parser's array ((width + 1) entries per row, src/edge264_headers.c:
		 * 114-125); the row by reciprocal, exact for any picture the reference accepts */
		const size_t i = (size_t)(mb - (const Edge264Macroblock *)e->slot[slot].mbs);
		const size_t mby = (size_t)(((uint64_t)i * b->recip_w1) >> 40), mbx = i - mby * (size_t)(b->width_mbs + 1);
		if (mbx < (size_t)b->width_mbs && mby < (size_t)b->height_mbs) {
			size_t a = mby * (size_t)b->width_mbs + mbx;
			/* how far the NAL has got counts macroblocks it DEBLOCKS too: a slice's last macroblocks may be I_PCM (no leaf call), and the reference deblocks
			 * the rest of a slice AFTER the unref callback (src/edge264_headers.c:497-525).  When recover_slice then walks back over them, e264_touch_ must
			 * see that it goes back, or the failed attempt's own deblocking shares a packet with the concealment that overwrites it and is lost
			 * (round 5, tools/damage_sweep.py --wide, one sample). */
			if (e->trk_serial != e->serial) { e->trk_serial = e->serial; e->trk_slot = slot; e->trk_addr = (int)a; }
			else if (e->trk_slot == slot && (int)a > e->trk_addr) e->trk_addr = (int)a;
			if ((b->side[a].state & E264_ST_ERR) && mb->recovery_bits == ctx->t.frame_flip_bit) {
				/* marked erroneous, then decoded again without a single leaf call: as I_PCM (src/edge264_slice.c:914-935).
				 * Its record starts over; e264_lift_pcm picks the samples up when the packet is closed. */
				memset(&b->mbs[a], 0, sizeof(E264Mb));
				b->side[a].state = 0;
				b->side[a].dbk_slice = 0xffff;
			}
			b->side[a].fedges = mb->filter_edges;
			/* CAVLC + 8x8 transform: the reference replaces the sixteen per-4x4 flags of an inter macroblock by one flag per 8x8 block the moment it
			 * deblocks it (src/edge264_deblock.c:1092-1096, in place in mb->nC) and the bS = 2 tests of this macroblock and of its later neighbours read
			 * that.  Mirrored here, on the record, at the same moment (round 5, tools/damage_sweep.py: until then e264_flush_mb did it for every
			 * macroblock with filter_edges set, which is the same thing unless the macroblock is decoded again and never deblocked again). */
			E264Mb *m = &b->mbs[a];
			if (m->kind == E264_MB_INTER && (m->flags & E264_MBF_T8x8) && !ctx->t.pps.entropy_coding_mode_flag && m->nz_mask) {
				unsigned nz = m->nz_mask;
				for (int q = 0; q < 4; q++)
					if (nz >> (q * 4) & 15)
						nz |= 15u << (q * 4);
				m->nz_mask = (uint16_t)nz;
			}
			/* the first call is the one that filters (the picture-completing pass runs over macroblocks that were deblocked
			 * earlier without touching them) */
			if (b->side[a].dbk_slice == 0xffff) {
				b->side[a].dbk_slice = (uint16_t)idx;
				b->mbs[a].dbk_slice = (uint16_t)idx; /* (a record that does not exist yet -- I_PCM -- takes it from the array when it is lifted) */
			}
		}
	}
	mb->filter_edges = 0; /* src/edge264_deblock.c:500: a macroblock is filtered once per parse */
	E264_PF_END(E264_PF_DEBLOCK);
}
