// e264_dbkp.h -- e264_dbkparam_kernel, second generation: deblocking parameters (bS, alpha, beta, indexA) of every
// macroblock from the command packet alone.
//
// Device restatement of deblock_mb's parameter part, /root/reference/src/edge264_deblock.c:927-1123 (bS: :958-1118 incl. the
// L0/L1 cross match of :913-925; alpha / beta / indexA: :945-955).
//
// Why a rewrite: round 1 let every lane fetch the few bytes it needed straight from the packet -- 625 lane-level loads per
// macroblock.  The vector memory path of gfx950 spends 1.5 - 2.5 cycles on every lane of a non-contiguous load however
// small it is (tools/calib/load_rate.hip), so that kernel was bound by address processing: 0.49 ms per 256 frames for a
// few hundred integer operations per macroblock.  Here a workgroup copies the records of 64 consecutive macroblocks, of
// their 64 top neighbours and of the left neighbour of the first one into LDS with CONTIGUOUS 16-byte loads (11 per
// macroblock), computes from LDS, and writes the 64-byte parameter records back as contiguous 16-byte stores.
//
// Output record (E264_DBK_BYTES = 64 per macroblock), unchanged:
//   [0..31]  bS[dir][edge][segment]      [32..40] alpha[plane*3 + t], t = 0 internal edges, 1 left MB edge, 2 top MB edge
//   [41..49] beta                        [50..58] indexA (tC0 lookup)                       [59..63] zero
// The phases are plain functions of (LDS, frame, first macroblock, thread id): tests/emu runs them on the host.
#ifndef E264_DBKP_H
#define E264_DBKP_H
#include "e264_dev.h"

namespace {

#define DP_MBS 64
#define DP_NT 256
struct __attribute__((aligned(16))) DbkpLds {
	uint32_t hdr[2 * DP_MBS + 1][8];   // E264Mb: [0] left neighbour of the first macroblock, [1..64] own, [65..128] top neighbours
	uint32_t mo[2 * DP_MBS + 1][36];   // motion in expanded (E264Motion) form, same order; filled from the compact records
	uint32_t out[DP_MBS][16];
	int8_t fo[DP_MBS][2];              // FilterOffsetA / B of each macroblock's slice
	uint8_t alpha[52], beta[52];
};

// macroblock address of record j
E264_DEV int dbkp_addr(const FrameCtx &f, int a0, int j)
{
	const int n = f.wm * f.hm;
	const int a = j <= DP_MBS ? a0 - 1 + j : a0 + (j - DP_MBS - 1) - f.wm;
	return min(max(a, 0), n - 1); // clamped: a record that is not a real neighbour is never used (the edge flags gate it)
}

E264_DEV void dbkp_phase_load(DbkpLds &L, const FrameCtx &f, int a0, int tid)
{
	const gu8 *mbs_g = f.payload - f.h->payload_off + f.h->mbs_off;
	for (int i = tid; i < (2 * DP_MBS + 1) * 2; i += DP_NT) { // 32-byte records: 2 pieces of 16 bytes
		const int j = i >> 1, part = i & 1;
		*(v4u *)&L.hdr[j][part * 4] = *(const gv4u *)(mbs_g + (size_t)dbkp_addr(f, a0, j) * 32 + part * 16);
	}
	if (tid < 52) { L.alpha[tid] = c_alpha[tid]; L.beta[tid] = c_beta[tid]; }
}

// after the records have landed: the motion of the inter macroblocks among them, expanded from the packet's compact
// records into the per-4x4 form the comparisons index (one task per record, list and quadrant); slice offsets
E264_DEV void dbkp_phase_slices(DbkpLds &L, const FrameCtx &f, int tid)
{
	if (f.motion)
		for (int i = tid; i < (2 * DP_MBS + 1) * 8; i += DP_NT) {
			const int j = i >> 3, l = i >> 2 & 1, q = i & 3;
			if ((L.hdr[j][0] & 255) != E264_MB_INTER)
				continue;
			uint32_t refword = 0xffffu, mv[4] = {0, 0, 0, 0};
			mot_quadrant(f.motion, L.hdr[j][5], L.hdr[j][6], l, q, refword, mv); // unused: no reference (-1), zero vectors
			((uint8_t *)&L.mo[j][l])[q] = (uint8_t)refword;
#pragma unroll
			for (int k = 0; k < 4; k++) L.mo[j][4 + l * 16 + q * 4 + k] = mv[k];
		}
	if (tid < DP_MBS) {
		cslice_t s = f.slices + (L.hdr[1 + tid][7] & 0xffff); // E264Mb.dbk_slice
		L.fo[tid][0] = s->FilterOffsetA; L.fo[tid][1] = s->FilterOffsetB;
	}
}

struct DbkpMo { int ref0, ref1; uint32_t mv0, mv1; };
// motion of 4x4 block k of a record (hdr: its E264Mb dwords, mo: its E264Motion dwords); intra / absent macroblocks count
// as "no reference, zero vector"
E264_DEV DbkpMo dbkp_motion(const uint32_t *hdr, const uint32_t *mo, bool has_motion, int k)
{
	DbkpMo o = {-1, -1, 0, 0};
	if (has_motion && (hdr[0] & 255) == E264_MB_INTER) {
		o.ref0 = (int)(int8_t)(mo[0] >> (8 * (k >> 2)));
		o.ref1 = (int)(int8_t)(mo[1] >> (8 * (k >> 2)));
		o.mv0 = mo[4 + k]; o.mv1 = mo[20 + k];
	}
	return o;
}
E264_DEV int dbkp_far(uint32_t a, uint32_t b)
{ // either component differs by 4 quarter samples or more (deblock.c:981-991)
	const int ax = (int16_t)(a & 0xffff), ay = (int)a >> 16, bx = (int16_t)(b & 0xffff), by = (int)b >> 16;
	const int dx = ax - bx, dy = ay - by;
	return ((dx < 0 ? -dx : dx) >= 4) | ((dy < 0 ? -dy : dy) >= 4);
}
// bS of role hl = (dir, edge, segment) of macroblock m (edge264_deblock.c:958-1118); L / T: records of the left / top neighbour
// (anything when the edge flag is off); `on`: the macroblock is deblocked at all
E264_DEV int dbkp_bs_value(const uint32_t *hm, const uint32_t *mm, const uint32_t *hL, const uint32_t *mL, const uint32_t *hT, const uint32_t *mT,
	bool has_motion, bool on, int hl)
{
	const int dir = hl >> 4 & 1, e = hl >> 2 & 3, sg = hl & 3;
	const uint32_t h0 = hm[0];
	const int kind = h0 & 255, flags = h0 >> 8 & 255;
	const bool intra = kind != E264_MB_INTER;
	const bool has_edge = e != 0 || (flags & (dir ? E264_MBF_EDGE_TOP : E264_MBF_EDGE_LEFT));
	const uint32_t *hn = e == 0 ? (dir ? hT : hL) : hm, *mn = e == 0 ? (dir ? mT : mL) : mm; // record holding the p side
	const int nkind = hn[0] & 255;
	const int kq = dir ? blk_of(sg, e) : blk_of(e, sg);
	const int kp = dir ? blk_of(sg, (e + 3) & 3) : blk_of((e + 3) & 3, sg);
	const int coded = ((hn[2] & 0xffff) >> kp & 1) | ((hm[2] & 0xffff) >> kq & 1);
	const DbkpMo p = dbkp_motion(hn, mn, has_motion, kp), q = dbkp_motion(hm, mm, has_motion, kq);
	const int refs_p = (p.ref0 != q.ref0) | (p.ref1 != q.ref1), refs_c = (p.ref0 != q.ref1) | (p.ref1 != q.ref0);
	const int mvs_p = dbkp_far(p.mv0, q.mv0) | dbkp_far(p.mv1, q.mv1), mvs_c = dbkp_far(p.mv0, q.mv1) | dbkp_far(p.mv1, q.mv0);
	const int bmo = (refs_p | mvs_p) & (refs_c | mvs_c);
	const bool skip8 = e != 0 && (flags & E264_MBF_T8x8) && (e & 1);
	int bs = coded ? 2 : bmo;
	bs = intra ? 3 : bs;
	bs = (e == 0 && (intra || nkind != E264_MB_INTER)) ? 4 : bs;
	return (!on || !has_edge || skip8) ? 0 : bs;
}
// alpha / beta / indexA value hl (0..26; 27..31 -> 0) of macroblock m (edge264_deblock.c:945-955)
E264_DEV int dbkp_ab_value(const uint32_t *hm, const uint32_t *hL, const uint32_t *hT, int foA, int foB, const uint8_t *alpha, const uint8_t *beta, bool on, int hl)
{
	if (hl >= 27 || !on)
		return 0;
	const uint32_t h0 = hm[0], h1 = hm[1];
	const int flags = h0 >> 8 & 255;
	const int what = hl / 9, pt = hl - what * 9, pl = pt / 3, t = pt - pl * 3;
	const uint32_t *hn = t == 0 ? hm : t == 2 ? hT : hL;
	const uint32_t n0 = hn[0], n1 = hn[1];
	const int qm = pl == 0 ? (int)(h0 >> 16 & 255) : pl == 1 ? (int)(h0 >> 24) : (int)(h1 & 255);
	const int qn = pl == 0 ? (int)(n0 >> 16 & 255) : pl == 1 ? (int)(n0 >> 24) : (int)(n1 & 255);
	const bool use_nb = (t == 1 && (flags & E264_MBF_EDGE_LEFT)) || (t == 2 && (flags & E264_MBF_EDGE_TOP));
	const int qPav = (qm + (use_nb ? qn : qm) + 1) >> 1;
	const int iA = min(max(qPav + foA, 0), 51), iB = min(max(qPav + foB, 0), 51);
	return what == 0 ? alpha[iA] : what == 1 ? beta[iB] : iA;
}

E264_DEV void dbkp_phase_compute(DbkpLds &L, const FrameCtx &f, int a0, int tid)
{
	const bool has_motion = f.motion != nullptr;
	const int n_mbs = f.wm * f.hm;
	uint8_t *out8 = (uint8_t *)&L.out[0][0];
	for (int it = 0; it < DP_MBS * 32 / DP_NT; it++) {
		const int id = it * DP_NT + tid, i = id >> 5, hl = id & 31;
		const int rm = 1 + i, rl = i, rt = 1 + DP_MBS + i; // own record, left and top neighbours
		const uint32_t h0 = L.hdr[rm][0];
		const bool on = a0 + i < n_mbs && (h0 >> 8 & E264_MBF_DEBLOCK) && (h0 & 255) != E264_MB_ABSENT;
		out8[i * 64 + hl] = (uint8_t)dbkp_bs_value(L.hdr[rm], L.mo[rm], L.hdr[rl], L.mo[rl], L.hdr[rt], L.mo[rt], has_motion, on, hl);
		out8[i * 64 + 32 + hl] = (uint8_t)dbkp_ab_value(L.hdr[rm], L.hdr[rl], L.hdr[rt], L.fo[i][0], L.fo[i][1], L.alpha, L.beta, on, hl);
	}
}

E264_DEV void dbkp_phase_store(const DbkpLds &L, const FrameCtx &f, int a0, int tid)
{
	const int n_mbs = f.wm * f.hm;
	const int i = tid >> 2, part = tid & 3; // 64 records x 4 pieces of 16 bytes
	if (a0 + i < n_mbs)
		*(gv4u *)(f.dbk + (size_t)(a0 + i) * E264_DBK_BYTES + part * 16) = *(const v4u *)&L.out[i][part * 4];
}

} // namespace
#endif
