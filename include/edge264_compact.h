/* edge264_compact.h -- the WIRE form of a command packet (version 5): the same information as version 4 (edge264_cmd.h) in fewer bytes for the link.
 *
 * Why.  A packet crosses PCIe once and is then read by four kernels; with the copy inside the clock the back end runs at 0.9 of the link
 * (bench.py pcie_inclusive.link_frac), and most macroblocks of an encoder-made stream say "one vector per list, no residual" -- P_Skip, B_Skip / Direct with one
 * motion for the whole macroblock, plain 16x16 -- in 40 or 48 bytes: a 32-byte record of which 7 bytes carry information and an 8-byte motion record per list
 * (5 200 of the 8 160 macroblocks of an average picture of tests/golden/streams/nat1080_ipp30.264, 5 700 of cabac_hd1080_ibbp30.264).  The stream itself
 * spends a run length on them (mb_skip_run, /root/reference/src/edge264_slice.c:1651-1849).
 *
 * What.  Version 5 keeps header, slice table, motion records and payload of version 4 and replaces the array of E264Mb by a table in which such a macroblock
 * is a 12-byte entry (one list) or a 20-byte entry (both lists); every other macroblock keeps its 32-byte record.  Two bitmaps (one bit per macroblock: compact
 * at all / both lists) and three numbers per macroblock row (byte offset of the row's first entry, compact and two-list macroblocks before the row) let ANY
 * macroblock's entry be found with two popcounts, so the table expands in parallel: on the DEVICE, by e264_expand_kernel (edge264_amd/csrc/e264_expand.h),
 * into the version-4 record array and motion section the four kernels read -- in a buffer of the stream's; the kernels change by where two pointers point -- and
 * on the HOST, by e264_expand_packet below, into a canonical version-4 packet (validation, oracle, tools).  Packets that stay in HBM (capture replay, benchmarks'
 * resident legs) are uploaded expanded.
 *
 *   [E264FrameHdr, version 5][E264SliceParams x n_slices]
 *   mbs_off ->  E264CompactHdr
 *               uint32_t row_off[height_mbs]                 byte offset of the row's first entry from the start of the entry area
 *               uint32_t row_cbase[height_mbs]               compact macroblocks in the rows before
 *               uint32_t row_bbase[height_mbs]               two-list compact macroblocks in the rows before
 *               uint32_t cbits[height_mbs][words_per_row]    bit x of row y: macroblock (x, y) is compact
 *               uint32_t bbits[height_mbs][words_per_row]    ... and predicts from both lists (a subset of cbits)
 *               (padding to 8)  entries in raster order: E264MbCompact (12 bytes), E264MbCompact + E264MbCompactL1 (20 bytes) or E264Mb (32 bytes; mot_off
 *                               counts in THIS packet's motion section)
 *   motion_off -> the motion records of the FULL inter macroblocks only      payload_off -> as in version 4
 *
 * Plain C99, no dependencies: the front end (C), the back end (C++) and, through the back end's exports, the Python tools share this one definition. */
#ifndef EDGE264_COMPACT_H
#define EDGE264_COMPACT_H
#include <stddef.h>
#include <string.h>
#include "edge264_cmd.h"
#ifdef __cplusplus
extern "C" {
#endif

#define E264_VERSION_COMPACT 5u
/* mot_hdr of "one partition per list": the list's four quadrant bits + its uniform bit (edge264_cmd.h E264_MOT_*) */
#define E264_MOT_HDR_UNI0 0x10fu
#define E264_MOT_HDR_UNI1 0x2f0u
#define E264_MOT_HDR_UNI01 0x3ffu
#define E264_MBCF_LIST1 0x80u /* E264MbCompact.flags: the (single) list is list 1 */

typedef struct E264CompactHdr { /* 16 bytes at mbs_off of a version-5 packet */
	uint32_t n_compact;      /* compact macroblocks, the two-list ones included */
	uint32_t n_both;         /* ... of which predict from both lists */
	uint32_t entries_bytes;  /* 32 * n_mbs - 20 * n_compact + 8 * n_both */
	uint32_t words_per_row;  /* (width_mbs + 31) / 32 */
} E264CompactHdr;

typedef struct E264MbCompact { /* 12 bytes: an inter macroblock of ONE partition without residual */
	uint8_t flags;       /* E264Mb.flags: EDGE_LEFT, EDGE_TOP, DEBLOCK (a macroblock with any other flag keeps its full record) | E264_MBCF_LIST1 */
	uint8_t ref_slot;    /* refPic: DPB slot, 0..31 (of list 0 in a two-list entry) */
	uint8_t ref_idx;     /* refIdx, 0..31 */
	uint8_t slice;       /* E264Mb.slice (a macroblock of slice 256 or beyond keeps its full record) */
	uint8_t qp[3];
	uint8_t dbk_slice;   /* E264Mb.dbk_slice (< 256) */
	int16_t mv[2];
} E264MbCompact;
typedef struct E264MbCompactL1 { /* 8 bytes behind the E264MbCompact of a two-list macroblock: list 1, as its motion record has it */
	uint8_t ref_slot, ref_idx, zero[2];
	int16_t mv[2];
} E264MbCompactL1;
E264_SIZE_CHECK(sizeof(E264CompactHdr) == 16, "E264CompactHdr");
E264_SIZE_CHECK(sizeof(E264MbCompact) == 12, "E264MbCompact");
E264_SIZE_CHECK(sizeof(E264MbCompactL1) == 8, "E264MbCompactL1");

static inline uint32_t e264_compact_table_bytes(uint32_t width_mbs, uint32_t height_mbs)
{ /* header + directory + bitmaps, padded to 8: where the entry area starts (from mbs_off) */
	const uint32_t wpr = (width_mbs + 31u) >> 5;
	return (16u + 12u * height_mbs + 8u * height_mbs * wpr + 7u) & ~7u;
}

/* how can this version-4 record go?  0: as it is, 1: one-list entry, 2: two-list entry.  rec: its motion record (NULL when the packet has no motion section) */
static inline int e264_mb_compact_class(const E264Mb *m, const uint8_t *rec)
{
	uint32_t d[2], r0, r1;
	if (m->kind != E264_MB_INTER || (m->flags & ~(E264_MBF_EDGE_LEFT | E264_MBF_EDGE_TOP | E264_MBF_DEBLOCK)) || m->coded || m->nz_mask || m->slice > 255 || m->dbk_slice > 255 || !rec)
		return 0;
	memcpy(d, m->modes, 8);
	memcpy(&r0, rec, 4);
	if (r0 & 0xffffe0e0u) /* slot and index below 32, the two spare bytes zero */
		return 0;
	if (d[1] == E264_MOT_HDR_UNI0 || d[1] == E264_MOT_HDR_UNI1)
		return 1;
	if (d[1] != E264_MOT_HDR_UNI01)
		return 0;
	memcpy(&r1, rec + 8, 4);
	return (r1 & 0xffffe0e0u) ? 0 : 2;
}

/* upper bound of what e264_compact_packet / e264_compact_sections write for a picture whose version-4 packet has this header */
static inline size_t e264_compact_bound(const void *v4)
{
	const E264FrameHdr *h = (const E264FrameHdr *)v4;
	return (size_t)h->total_bytes + e264_compact_table_bytes(h->width_mbs, h->height_mbs) + 64;
}

/* The fold, from the SECTIONS of a version-4 packet wherever they lie (a front end folds straight out of its builder's arrays; e264_compact_packet below
 * folds a packet).  h: the version-4 header (magic, sizes, slices_off, mbs_off, n_slices, payload_bytes and the summary fields are taken as they are;
 * motion_off, payload_off, total_bytes, version are written anew).  mot: NULL when no macroblock is inter.  payload_len <= h->payload_bytes bytes are
 * copied, the rest is zeros.  The sections must be what e264hip_packet_check would accept.  Returns the wire size, 0 when `cap` is too small. */
static inline size_t e264_compact_sections(const E264FrameHdr *h, const void *slices, const E264Mb *mbs, const uint8_t *mot, const uint8_t *payload, size_t payload_len,
	void *out, size_t cap)
{
	const uint32_t wm = h->width_mbs, hm = h->height_mbs, wpr = (wm + 31u) >> 5;
	const uint32_t tb = e264_compact_table_bytes(wm, hm);
	if (cap < (size_t)h->total_bytes + tb + 64)
		return 0;
	uint8_t *o = (uint8_t *)out;
	memcpy(o, h, sizeof(*h));
	memset(o + sizeof(*h), 0, h->slices_off - sizeof(*h));
	memcpy(o + h->slices_off, slices, sizeof(E264SliceParams) * (size_t)h->n_slices);
	const uint32_t send = h->slices_off + (uint32_t)sizeof(E264SliceParams) * h->n_slices;
	memset(o + send, 0, h->mbs_off - send + tb);
	E264FrameHdr *oh = (E264FrameHdr *)o;
	E264CompactHdr *ch = (E264CompactHdr *)(o + h->mbs_off);
	uint32_t *row_off = (uint32_t *)(o + h->mbs_off + 16), *row_cbase = row_off + hm, *row_bbase = row_cbase + hm, *cbits = row_bbase + hm, *bbits = cbits + hm * wpr;
	uint8_t *ent = o + h->mbs_off + tb;
	uint32_t eoff = 0, nc = 0, nb = 0, moff = 0;
	const E264Mb *m = mbs;
	for (uint32_t y = 0; y < hm; y++) {
		row_off[y] = eoff; row_cbase[y] = nc; row_bbase[y] = nb;
		for (uint32_t x = 0; x < wm; x++, m++) {
			if (m->kind != E264_MB_INTER) { memcpy(ent + eoff, m, 32); eoff += 32; continue; }
			uint32_t d[2];
			memcpy(d, m->modes, 8);
			const uint8_t *rec = mot ? mot + d[0] : NULL;
			const int cls = e264_mb_compact_class(m, rec);
			if (!cls) {
				memcpy(ent + eoff, m, 32);
				memcpy(ent + eoff + offsetof(E264Mb, modes), &moff, 4); /* mot_off in this packet's motion section */
				moff += e264_mot_record_bytes(d[1]);
				eoff += 32;
				continue;
			}
			uint8_t *c = ent + eoff; /* E264MbCompact */
			c[0] = (uint8_t)(m->flags | (d[1] == E264_MOT_HDR_UNI1 ? E264_MBCF_LIST1 : 0)); c[1] = rec[0]; c[2] = rec[1]; c[3] = (uint8_t)m->slice;
			c[4] = m->qp[0]; c[5] = m->qp[1]; c[6] = m->qp[2]; c[7] = (uint8_t)m->dbk_slice;
			memcpy(c + 8, rec + 4, 4);
			eoff += 12; nc++;
			cbits[y * wpr + (x >> 5)] |= 1u << (x & 31);
			if (cls == 2) {
				memcpy(c + 12, rec + 8, 8);
				eoff += 8; nb++;
				bbits[y * wpr + (x >> 5)] |= 1u << (x & 31);
			}
		}
	}
	ch->n_compact = nc; ch->n_both = nb; ch->entries_bytes = eoff; ch->words_per_row = wpr;
	/* motion section: the records of the full inter macroblocks, in macroblock order (the bitmap says which are not) */
	uint32_t pos = (h->mbs_off + tb + eoff + 7u) & ~7u;
	memset(ent + eoff, 0, pos - (h->mbs_off + tb + eoff));
	oh->motion_off = moff ? pos : 0;
	if (moff) {
		m = mbs;
		for (uint32_t y = 0; y < hm; y++)
			for (uint32_t x = 0; x < wm; x++, m++) {
				if (m->kind != E264_MB_INTER || (cbits[y * wpr + (x >> 5)] >> (x & 31) & 1u)) continue;
				uint32_t d[2];
				memcpy(d, m->modes, 8);
				const uint32_t rb = e264_mot_record_bytes(d[1]);
				memcpy(o + pos, mot + d[0], rb);
				pos += rb;
			}
	}
	const uint32_t pad = ((pos + 7u) & ~7u) - pos;
	memset(o + pos, 0, pad);
	pos += pad;
	if (payload_len) memcpy(o + pos, payload, payload_len);
	memset(o + pos + payload_len, 0, h->payload_bytes - payload_len);
	oh->version = E264_VERSION_COMPACT;
	oh->payload_off = pos;
	oh->total_bytes = pos + h->payload_bytes;
	return oh->total_bytes;
}

/* version 4 (already vetted: e264hip_packet_check, or the front end's own product) -> version 5.  Returns the wire size, 0 when `cap` is too small or the
 * input is not a version-4 packet. */
static inline size_t e264_compact_packet(const void *v4, size_t bytes, void *out, size_t cap)
{
	const uint8_t *p = (const uint8_t *)v4;
	const E264FrameHdr *h = (const E264FrameHdr *)v4;
	if (bytes < sizeof(*h) || h->magic != E264_MAGIC || h->version != E264_VERSION)
		return 0;
	return e264_compact_sections(h, p + h->slices_off, (const E264Mb *)(p + h->mbs_off), h->motion_off ? p + h->motion_off : NULL, p + h->payload_off, h->payload_bytes, out, cap);
}

/* Structure of a version-5 packet: everything an expansion (host or device) reads lies inside the packet and says what the bitmaps say.  0 = sound.
 * (What the records then MEAN is vetted on the expanded packet by the back end's per-macroblock walk, as for version 4.) */
static inline int e264_check_compact(const void *wire, size_t bytes)
{
	const uint8_t *p = (const uint8_t *)wire;
	const E264FrameHdr *h = (const E264FrameHdr *)wire;
	if (bytes < sizeof(*h) || h->magic != E264_MAGIC || h->version != E264_VERSION_COMPACT || h->total_bytes != bytes) return -1;
	const uint32_t wm = h->width_mbs, hm = h->height_mbs;
	if (wm == 0 || hm == 0 || hm > 1056) return -1;
	const uint64_t n = (uint64_t)wm * hm;
	const uint32_t wpr = (wm + 31u) >> 5, tb = e264_compact_table_bytes(wm, hm);
	if ((h->mbs_off & 7) || (uint64_t)h->slices_off + (uint64_t)h->n_slices * sizeof(E264SliceParams) > h->mbs_off || (uint64_t)h->mbs_off + tb > bytes) return -1;
	const E264CompactHdr *ch = (const E264CompactHdr *)(p + h->mbs_off);
	if (ch->words_per_row != wpr || ch->n_compact > n || ch->n_both > ch->n_compact || ch->entries_bytes != 32ull * n - 20ull * ch->n_compact + 8ull * ch->n_both) return -1;
	const uint64_t end = (uint64_t)h->mbs_off + tb + ch->entries_bytes;
	if (end > bytes || (h->motion_off && (h->motion_off < end || h->motion_off > bytes)) || h->payload_off < end || (uint64_t)h->payload_off + h->payload_bytes > bytes) return -1;
	if ((h->motion_off && h->motion_off > h->payload_off) || ((h->slices_off | h->motion_off | h->payload_off) & 7)) return -1;
	const uint32_t *row_off = (const uint32_t *)(p + h->mbs_off + 16), *row_cbase = row_off + hm, *row_bbase = row_cbase + hm, *cbits = row_bbase + hm, *bbits = cbits + (size_t)hm * wpr;
	uint32_t eoff = 0, nc = 0, nb = 0;
	for (uint32_t y = 0; y < hm; y++) {
		if (row_off[y] != eoff || row_cbase[y] != nc || row_bbase[y] != nb) return -1;
		uint32_t c = 0, b = 0;
		for (uint32_t w = 0; w < wpr; w++) {
			const uint32_t cw = cbits[y * wpr + w], bw = bbits[y * wpr + w];
			if (w == wpr - 1 && (wm & 31) && (cw >> (wm & 31))) return -1; /* bits beyond the row */
			if (bw & ~cw) return -1;
			c += (uint32_t)__builtin_popcount(cw); b += (uint32_t)__builtin_popcount(bw);
		}
		nc += c; nb += b; eoff += 32 * wm - 20 * c + 8 * b;
	}
	if (nc != ch->n_compact || nb != ch->n_both || eoff != ch->entries_bytes) return -1;
	/* the expansion is a version-4 packet: 32-bit offsets (and what the kernels add to them) */
	return ((uint64_t)h->mbs_off + 32 * n + (h->motion_off ? h->payload_off - h->motion_off : 0) + 8ull * nc + 8ull * nb + 8 + h->payload_bytes) >> 31 ? -1 : 0;
}

/* bytes of the motion section of the expansion: the wire's records, then 8 bytes per compact macroblock and list */
static inline size_t e264_expanded_motion_bytes(const void *wire)
{
	const uint8_t *p = (const uint8_t *)wire;
	const E264FrameHdr *h = (const E264FrameHdr *)wire;
	const E264CompactHdr *ch = (const E264CompactHdr *)(p + h->mbs_off);
	return (size_t)(h->motion_off ? h->payload_off - h->motion_off : 0) + (size_t)8 * ch->n_compact + (size_t)8 * ch->n_both; /* (all three multiples of 8) */
}
/* size of the canonical version-4 packet e264_expand_packet makes of this (structurally sound) version-5 packet */
static inline size_t e264_expanded_bytes(const void *wire)
{
	const E264FrameHdr *h = (const E264FrameHdr *)wire;
	return (size_t)h->mbs_off + (size_t)32 * h->width_mbs * h->height_mbs + e264_expanded_motion_bytes(wire) + h->payload_bytes;
}
/* bytes of the DEVICE expansion (the stream's expansion buffer): the canonical packet's record array and motion section -- the same bytes e264_expand_packet
 * puts at mbs_off .. payload_off; the payload is read where the wire packet lies */
static inline size_t e264_expand_area_bytes(const void *wire)
{
	const E264FrameHdr *h = (const E264FrameHdr *)wire;
	return (size_t)32 * h->width_mbs * h->height_mbs + e264_expanded_motion_bytes(wire);
}

/* the version-4 record of macroblock (x, y) of a version-5 packet and, for a compact one, its motion record (8 or 16 bytes, *rec_bytes; 0 for a full record).
 * synth_off: mot_off of the FIRST compact macroblock's motion record (they follow one another in macroblock order).  THE definition of the expansion: the
 * device kernel restates it. */
static inline void e264_expand_mb(const uint8_t *wire, uint32_t x, uint32_t y, uint32_t synth_off, E264Mb *out, uint8_t rec[16], uint32_t *rec_bytes)
{
	const E264FrameHdr *h = (const E264FrameHdr *)wire;
	const E264CompactHdr *ch = (const E264CompactHdr *)(wire + h->mbs_off);
	const uint32_t hm = h->height_mbs, wpr = ch->words_per_row;
	const uint32_t *row_off = (const uint32_t *)(wire + h->mbs_off + 16), *row_cbase = row_off + hm, *row_bbase = row_cbase + hm, *cbits = row_bbase + hm, *bbits = cbits + hm * wpr;
	uint32_t c = 0, b = 0;
	for (uint32_t w = 0; w < (x >> 5); w++) { c += (uint32_t)__builtin_popcount(cbits[y * wpr + w]); b += (uint32_t)__builtin_popcount(bbits[y * wpr + w]); }
	const uint32_t cw = cbits[y * wpr + (x >> 5)], bw = bbits[y * wpr + (x >> 5)], below = (1u << (x & 31)) - 1u;
	c += (uint32_t)__builtin_popcount(cw & below); b += (uint32_t)__builtin_popcount(bw & below);
	const uint8_t *e = wire + h->mbs_off + e264_compact_table_bytes(h->width_mbs, hm) + row_off[y] + 32u * x - 20u * c + 8u * b;
	*rec_bytes = 0;
	if (!(cw >> (x & 31) & 1u)) { memcpy(out, e, 32); return; }
	const int both = (int)(bw >> (x & 31) & 1u);
	E264MbCompact k;
	memcpy(&k, e, 12);
	memset(out, 0, 32);
	out->kind = E264_MB_INTER; out->flags = (uint8_t)(k.flags & ~E264_MBCF_LIST1);
	out->qp[0] = k.qp[0]; out->qp[1] = k.qp[1]; out->qp[2] = k.qp[2];
	out->slice = k.slice; out->dbk_slice = k.dbk_slice;
	const uint32_t d[2] = {synth_off + 8u * (row_cbase[y] + c) + 8u * (row_bbase[y] + b), both ? E264_MOT_HDR_UNI01 : (k.flags & E264_MBCF_LIST1) ? E264_MOT_HDR_UNI1 : E264_MOT_HDR_UNI0};
	memcpy(out->modes, d, 8);
	rec[0] = k.ref_slot; rec[1] = k.ref_idx; rec[2] = rec[3] = 0;
	memcpy(rec + 4, k.mv, 4);
	*rec_bytes = 8;
	if (both) { memcpy(rec + 8, e + 12, 8); *rec_bytes = 16; }
}

/* version 5 (e264_check_compact has said 0) -> canonical version 4: [header][slices][E264Mb x n][motion: the wire's records, then the compact macroblocks'][payload].
 * Returns the size, 0 when `cap` is too small. */
static inline size_t e264_expand_packet(const void *wire, size_t bytes, void *out, size_t cap)
{
	const uint8_t *p = (const uint8_t *)wire;
	const E264FrameHdr *h = (const E264FrameHdr *)wire;
	const size_t need = e264_expanded_bytes(wire);
	(void)bytes;
	if (cap < need) return 0;
	const E264CompactHdr *ch = (const E264CompactHdr *)(p + h->mbs_off);
	const uint32_t wm = h->width_mbs, hm = h->height_mbs, n = wm * hm;
	const uint32_t mot = h->motion_off ? h->payload_off - h->motion_off : 0; /* (the padding in front of the payload included: never addressed) */
	const uint32_t mtot = (uint32_t)e264_expanded_motion_bytes(wire);
	uint8_t *o = (uint8_t *)out;
	memcpy(o, p, h->mbs_off);
	E264FrameHdr *oh = (E264FrameHdr *)o;
	E264Mb *mbs = (E264Mb *)(o + h->mbs_off);
	const uint32_t mo = h->mbs_off + 32u * n;
	if (mot) memcpy(o + mo, p + h->motion_off, mot);
	for (uint32_t y = 0; y < hm; y++)
		for (uint32_t x = 0; x < wm; x++) {
			uint8_t rec[16];
			uint32_t rb;
			E264Mb *m = &mbs[y * wm + x];
			e264_expand_mb(p, x, y, mot, m, rec, &rb);
			if (rb) { uint32_t d[2]; memcpy(d, m->modes, 8); memcpy(o + mo + d[0], rec, rb); }
		}
	memcpy(o + mo + mtot, p + h->payload_off, h->payload_bytes);
	oh->version = E264_VERSION;
	oh->motion_off = (mot || ch->n_compact) ? mo : 0;
	oh->payload_off = mo + mtot;
	oh->total_bytes = mo + mtot + h->payload_bytes;
	return oh->total_bytes;
}

#ifdef __cplusplus
}
#endif
#endif
