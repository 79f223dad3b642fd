// e264_expand.h -- e264_expand_kernel: a WIRE packet (version 5, include/edge264_compact.h) back into the record array and motion section the four
// kernels read, in the stream's expansion buffer (E264Job.expand):  [E264Mb x n_mbs][the wire's motion records][one 8-byte record per compact macroblock and list].
// One thread per macroblock restates e264_expand_mb (THE definition; tests/test_compact.py holds this source, compiled for the host, against it byte for byte),
// then the threads of the grid copy the wire's motion section dword by dword.  HBM-bound and small: 0.3 MB written per 1080p picture.
#ifndef E264_EXPAND_H
#define E264_EXPAND_H
#include "e264_dev.h"
#include "../../include/edge264_compact.h"

namespace {

#define XP_NT 256

// thread `t` of `nt` (the whole grid row of this job)
E264_DEV void expand_thread(const E264Job &job, const uint32_t t, const uint32_t nt)
{
	const uint8_t *pkt = job.packet;
	chdr_t h = (chdr_t)pkt;
	if (h->magic != E264_MAGIC || h->version != E264_VERSION_COMPACT || !job.expand)
		return;
	const uint32_t wm = h->width_mbs, hm = h->height_mbs, n = wm * hm;
	const gu32 *tab = (const gu32 *)(pkt + h->mbs_off);
	const uint32_t wpr = tab[3];
	const gu32 *row_off = tab + 4, *row_cbase = row_off + hm, *row_bbase = row_cbase + hm, *cbits = row_bbase + hm, *bbits = cbits + hm * wpr;
	const uint8_t *ent = pkt + h->mbs_off + ((16u + 12u * hm + 8u * hm * wpr + 7u) & ~7u); // e264_compact_table_bytes
	const uint32_t mot = h->motion_off ? h->payload_off - h->motion_off : 0;
	gu32 *out = (gu32 *)job.expand;
	gu32 *omot = out + 8 * n;
	for (uint32_t a = t; a < n; a += nt) {
		const uint32_t y = a / wm, x = a - y * wm;
		uint32_t c = 0, b = 0;
		for (uint32_t w = 0; w < (x >> 5); w++) { c += (uint32_t)__builtin_popcount(cbits[y * wpr + w]); b += (uint32_t)__builtin_popcount(bbits[y * wpr + w]); }
		const uint32_t cw = cbits[y * wpr + (x >> 5)], bw = bbits[y * wpr + (x >> 5)], below = (1u << (x & 31)) - 1u;
		c += (uint32_t)__builtin_popcount(cw & below); b += (uint32_t)__builtin_popcount(bw & below);
		const gu32 *e = (const gu32 *)(ent + row_off[y] + 32u * x - 20u * c + 8u * b); // entries are 12, 20 or 32 bytes: dword-aligned
		gv4u *o = (gv4u *)(out + 8 * a); // two 16-byte stores per record
		if (!(cw >> (x & 31) & 1u)) {
			const v4u lo = ((const gv4u *)e)[0], hi = ((const gv4u *)e)[1];
			o[0] = lo; o[1] = hi;
			continue;
		}
		// E264MbCompact {flags, ref_slot, ref_idx, slice | qp[3], dbk_slice | mv[2]} [E264MbCompactL1 {ref_slot, ref_idx, 0, 0 | mv[2]}] ->
		// E264Mb {kind, flags, qp0, qp1 | qp2, chroma_mode, i16_mode, - | nz_mask, slice | coded | payload_off | mot_off | mot_hdr | dbk_slice, -}
		const bool both = bw >> (x & 31) & 1u;
		const uint32_t e0 = e[0], e1 = e[1], e2 = e[2];
		const uint32_t k = 2 * (row_cbase[y] + c) + 2 * (row_bbase[y] + b); // dwords of compact motion records before this one
		o[0] = (v4u){E264_MB_INTER | (e0 & 0x7fu) << 8 | (e1 & 0xffffu) << 16, e1 >> 16 & 255u, (e0 >> 24) << 16, 0u};
		o[1] = (v4u){0u, mot + 4u * k, both ? E264_MOT_HDR_UNI01 : (e0 & E264_MBCF_LIST1) ? E264_MOT_HDR_UNI1 : E264_MOT_HDR_UNI0, e1 >> 24};
		gv2u *r = (gv2u *)(omot + (mot >> 2) + k);
		r[0] = (v2u){e0 >> 8 & 0xffffu, e2}; // refPic, refIdx, 0, 0 | mv
		if (both) r[1] = (v2u){e[3], e[4]};
	}
	const gv2u *imot = (const gv2u *)(pkt + h->motion_off); // (section offsets and sizes are multiples of 8)
	for (uint32_t i = t; i < (mot >> 3); i += nt) ((gv2u *)omot)[i] = imot[i];
}

} // namespace
#endif
