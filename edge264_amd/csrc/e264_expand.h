// e264_expand.h -- e264_expand_kernel: a WIRE packet (version 5, include/edge264_compact.h) back into the record array and motion section the four
// kernels read, in the stream's expansion buffer (E264Job.expand):  [E264Mb x n_mbs][the wire's motion records][one 8-byte record per compact macroblock].
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
	const uint32_t entries_off = tab[1], wpr = tab[3];
	const gu32 *row_off = tab + 4, *row_cbase = row_off + hm, *bits = row_cbase + hm;
	const uint32_t mot = h->motion_off ? h->payload_off - h->motion_off : 0;
	gu32 *out = (gu32 *)job.expand;
	gu32 *omot = out + 8 * n;
	for (uint32_t a = t; a < n; a += nt) {
		const uint32_t y = a / wm, x = a - y * wm;
		uint32_t c = 0;
		for (uint32_t w = 0; w < (x >> 5); w++) c += (uint32_t)__builtin_popcount(bits[y * wpr + w]);
		const uint32_t word = bits[y * wpr + (x >> 5)];
		c += (uint32_t)__builtin_popcount(word & ((1u << (x & 31)) - 1u));
		const gu32 *e = (const gu32 *)(pkt + h->mbs_off + entries_off + row_off[y] + 32u * x - 20u * c); // entries are 12 or 32 bytes: dword-aligned
		gu32 *o = out + 8 * a;
		if (!(word >> (x & 31) & 1u)) {
#pragma unroll
			for (int i = 0; i < 8; i++) o[i] = e[i];
			continue;
		}
		// E264MbCompact {flags, ref_slot, ref_idx, slice | qp[3], dbk_slice | mv[2]} ->
		// E264Mb {kind, flags, qp0, qp1 | qp2, chroma_mode, i16_mode, - | nz_mask, slice | coded | payload_off | mot_off | mot_hdr | dbk_slice, -}
		const uint32_t e0 = e[0], e1 = e[1], e2 = e[2];
		const uint32_t k = row_cbase[y] + c;
		o[0] = E264_MB_INTER | (e0 & 255u) << 8 | (e1 & 0xffffu) << 16;
		o[1] = e1 >> 16 & 255u;
		o[2] = (e0 >> 24) << 16;
		o[3] = 0;
		o[4] = 0;
		o[5] = mot + 8u * k;
		o[6] = E264_MOT_HDR_UNI0;
		o[7] = e1 >> 24;
		omot[(mot >> 2) + 2 * k] = (e0 >> 8 & 0xffffu); // refPic, refIdx, 0, 0
		omot[(mot >> 2) + 2 * k + 1] = e2;
	}
	const gu32 *imot = (const gu32 *)(pkt + h->motion_off);
	for (uint32_t i = t; i < (mot >> 2); i += nt) omot[i] = imot[i];
}

} // namespace
#endif
