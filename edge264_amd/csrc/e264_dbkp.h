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
// Raw record (DP_RAW = 64 bytes per macroblock, in LDS only since round 6):
//   [0..31]  bS[dir][edge][segment]      [32..40] alpha[plane*3 + t], t = 0 internal edges, 1 left MB edge, 2 top MB edge
//   [41..49] beta                        [50..58] indexA (tC0 lookup)                       [59..63] zero
// What leaves for memory (round 6, E264_DBK_BYTES = 144 per macroblock) is the same information IN THE LAYOUT OF THE DEBLOCKING KERNEL'S
// LANES: sixteen 8-byte pieces, one per (plane kind, direction, line-pair segment) -- exactly what one lane of e264_dbk.h needs for its
// four edge slots of one direction, ready to be spread into packed 16-bit pairs with one byte permute per value -- and 16 bytes all lanes share:
//   piece (luma: byte offset seg * 16 + dir * 8; chroma: 64 + line pair * 16 + dir * 8 -- a lane's two directions are ONE 16-byte load):
//     dword 0  alphaE[slot 0..3]   alpha of the slot's edge, or 0 where its bS is 0 (nothing is below 0: the edge is left alone)
//     dword 1  tC[slot 0..3]       tC0 of (bS, indexA); chroma: + 1 (its tC); 0 for bS 0 and 4 (at most 26: bit 7 is free);
//                                  bit 7 of byte 0: slot 0 has bS 4, bit 7 of byte 2: slot 2 has bS 4
//   (luma slot e = edge e of the macroblock; chroma slot 0 = Cb macroblock edge, 1 = Cb inner edge, 2 / 3 = the same of Cr)
//   bytes 128..143: beta of the luma plane {inner, left, top} at 128..130, of Cb at 136..138, of Cr at 139..141 (a lane's own are within 8 bytes)
//   ((alpha >> 2) + 2 of the bS 4 luma test is not stored: where bS is 4, alphaE IS alpha.)
// (The first version of the round had 16-byte pieces with beta and the threshold in each: 256 bytes per macroblock, whose stores alone cost this kernel
// 0.044 ms per launch, profiles/r06_ablations.txt item 8.)
// Until round 5 every lane of the deblocking kernel derived these from the raw record in every step: ~26 LDS reads, a dependent tC0
// table look-up and ~160 VALU instructions per step and wave, i.e. once per line-pair segment of each of the 8 (15) macroblocks of a step
// AND per lane pair that shares it; here it is done once per macroblock by four threads.
// The phases are plain functions of (LDS, frame, first macroblock, thread id): tests/emu runs them on the host.
#ifndef E264_DBKP_H
#define E264_DBKP_H
#include "e264_dev.h"

namespace {

#define DP_MBS 64
#define DP_NT 256
#define DP_RAW 64
// HAS_L1 = false (round 6): the launcher knows that no packet of the batch predicts from list 1 (every P and I picture: found by the validation that vets a
// packet anyway) -- the expanded motion has no room for list 1 (80 instead of 144 bytes per record): 18.9 KB of LDS instead of 27.2, EIGHT workgroups per CU
// instead of six.  This kernel is two dependent trips to memory and three barriers: 0.244 -> 0.192 ms per launch (profiles/r06_ablations.txt item 13).
template <bool HAS_L1_>
struct __attribute__((aligned(16))) DbkpLdsT {
	static constexpr bool HAS_L1 = HAS_L1_;
	static constexpr int MO = HAS_L1_ ? 36 : 20; // dwords of expanded motion per record: references of both lists (2, 2 unused), 16 vectors per list
	// TOPC (the full form, round 6): a TOP neighbour is only ever compared along its bottom row of 4x4 blocks -- its motion is kept as 9 dwords (the references of its
	// two lower quadrants of both lists as bytes, four vectors per list) instead of 36: 20.3 KB instead of 27.2, eight workgroups per CU for B pictures too
	static constexpr bool TOPC = HAS_L1_;
	static constexpr int NFULL = TOPC ? DP_MBS + 1 : 2 * DP_MBS + 1; // records whose motion is kept in full: [0] the left neighbour, [1..64] own (, [65..128] top)
	uint32_t hdr[2 * DP_MBS + 1][8];   // E264Mb: [0] left neighbour of the first macroblock, [1..64] own, [65..128] top neighbours
	union {
		uint32_t mo[NFULL][MO];                        // motion in expanded (E264Motion) form, same order; filled from the compact records
		uint32_t pieces[DP_MBS][E264_DBK_BYTES / 4];   // (once the comparisons are done) the records in the lanes' layout, on their way out
	};
	uint32_t mot[TOPC ? DP_MBS : 1][9]; // TOPC: top neighbour i: [0] = {l0 ref of quadrant 2, of 3, l1 ref of 2, of 3}, [1..4] l0 vectors of blocks 10 11 14 15, [5..8] l1
	uint32_t out[DP_MBS][DP_RAW / 4];  // raw records
	int8_t fo[DP_MBS][2];              // FilterOffsetA / B of each macroblock's slice
	uint8_t alpha[52], beta[52];
	uint32_t tc3[52];                  // per indexA: bytes {0, tC0 of bS 1, of bS 2, of bS 3} (bS 0 and 4 have no tC0: entry bS & 3 = 0)
	uint32_t any_l1;                   // some record of the workgroup (own, left, top) predicts from list 1
#ifdef E264_DBKP_LDS_PAD // measuring aid: the kernel at a lower occupancy (bytes of LDS nobody uses)
	uint8_t pad[E264_DBKP_LDS_PAD];
#endif
};
typedef DbkpLdsT<true> DbkpLds;
static_assert(sizeof(((DbkpLdsT<true> *)0)->pieces) <= sizeof(((DbkpLdsT<true> *)0)->mo) && sizeof(((DbkpLdsT<false> *)0)->pieces) <= sizeof(((DbkpLdsT<false> *)0)->mo), "the pieces reuse the motion area");

#ifndef E264_HOST_INTRINSICS
E264_DEV void dbkp_note_l1(uint32_t *p) { atomicOr(p, 1u); }
#else
E264_DEV void dbkp_note_l1(uint32_t *p) { *p |= 1u; }
#endif
// macroblock address of record j
E264_DEV int dbkp_addr(const FrameCtx &f, int a0, int j)
{
	const int n = f.wm * f.hm;
	const int a = j <= DP_MBS ? a0 - 1 + j : a0 + (j - DP_MBS - 1) - f.wm;
	return min(max(a, 0), n - 1); // clamped: a record that is not a real neighbour is never used (the edge flags gate it)
}

template <class LDS> E264_DEV void dbkp_phase_load(LDS &L, const FrameCtx &f, int a0, int tid)
{
	const gu8 *mbs_g = f.mbs_g;
	for (int i = tid; i < (2 * DP_MBS + 1) * 2; i += DP_NT) { // 32-byte records: 2 pieces of 16 bytes
		const int j = i >> 1, part = i & 1;
		*(v4u *)&L.hdr[j][part * 4] = *(const gv4u *)(mbs_g + (size_t)dbkp_addr(f, a0, j) * 32 + part * 16);
	}
	if (tid < 52) { L.alpha[tid] = c_alpha[tid]; L.beta[tid] = c_beta[tid]; }
	if (tid < 52) L.tc3[tid] = (uint32_t)c_tc0[0][tid] << 8 | (uint32_t)c_tc0[1][tid] << 16 | (uint32_t)c_tc0[2][tid] << 24;
	if (tid == 0) L.any_l1 = 0;
}

// One list of a macroblock's compact motion record (edge264_cmd.h E264_MOT_*) -> the per-4x4 form the comparisons index:
// mo[l] = the four quadrants' DPB slots as bytes (0xff: unused), mo[4 + l * 16 + k] = vector of 4x4 block k (unused: 0).
// Every offset follows from the shape word alone, so the loads do not wait for one another.
E264_DEV void dbkp_expand_list(const gu8 *motion, uint32_t mot_off, uint32_t h, int l, uint32_t *mo, const bool bottom_row_only = false)
{
	uint32_t off = mot_off;
	if (l) { // skip the list-0 part
		uint32_t n0 = 0;
#pragma unroll
		for (int k = 0; k < 4; k++)
			n0 += E264_MOT_USED(h, k) ? 4 + 4 * mot_nmv(E264_MOT_SUB(h, k)) : 0;
		off += E264_MOT_UNI(h, 0) ? 8 : n0;
	}
	const gu32 *rec = (const gu32 *)(motion + off);
	uint32_t refs = 0xffffffffu;
	v4u mv[4];
#ifdef E264_ABL_DBKP_NOMOT // timing ablation: the motion records are not read (wrong bS): 0.225 -> 0.157 ms, profiles/r03_ablations.txt item 17
	if (off != 0xfffffffcu && !bottom_row_only) { mo[l] = off; for (int q = 0; q < 4; q++) *(v4u *)&mo[4 + l * 16 + q * 4] = (v4u){off, h, off, h}; return; }
#endif
	if (E264_MOT_UNI(h, l)) {
		const uint32_t r = rec[0], v = rec[1];
		refs = (r & 255u) * 0x01010101u;
#pragma unroll
		for (int q = 0; q < 4; q++) mv[q] = (v4u){v, v, v, v};
	} else {
#pragma unroll
		for (int q = 0; q < 4; q++) {
			mv[q] = (v4u){0, 0, 0, 0};
			if (!E264_MOT_USED(h, l * 4 + q))
				continue;
			const uint32_t sub = E264_MOT_SUB(h, l * 4 + q);
			const uint32_t r = rec[0], v0 = rec[1], v1 = sub ? rec[2] : v0;
			refs = (refs & ~(255u << (8 * q))) | (r & 255u) << (8 * q);
			if (sub == 3) mv[q] = (v4u){v0, v1, rec[3], rec[4]};
			else if (sub == 2) mv[q] = (v4u){v0, v1, v0, v1};
			else mv[q] = (v4u){v0, v0, v1, v1};
			rec += 1 + mot_nmv(sub);
		}
	}
	if (bottom_row_only) { // a top neighbour (DbkpLdsT::mot): the two lower quadrants' references, the four vectors of its bottom row
		((uint16_t *)mo)[l] = (uint16_t)(refs >> 16);
		mo[1 + l * 4] = mv[2].z; mo[2 + l * 4] = mv[2].w; mo[3 + l * 4] = mv[3].z; mo[4 + l * 4] = mv[3].w;
		return;
	}
	mo[l] = refs;
#pragma unroll
	for (int q = 0; q < 4; q++) *(v4u *)&mo[4 + l * 16 + q * 4] = mv[q];
}

// after the records have landed: the motion of the inter macroblocks among them (one task per record and list); slice offsets
template <class LDS> E264_DEV void dbkp_phase_slices(LDS &L, const FrameCtx &f, int tid)
{
	if (f.motion)
		for (int i = tid; i < (2 * DP_MBS + 1) * 2; i += DP_NT) {
			const int j = i >> 1, l = i & 1;
			if ((L.hdr[j][0] & 255) == E264_MB_INTER) {
				if (!LDS::HAS_L1 && l) // (no room for it and, by the launcher's word, nothing to put there)
					continue;
				if (LDS::TOPC && j > DP_MBS) dbkp_expand_list(f.motion, L.hdr[j][5], L.hdr[j][6], l, L.mot[j - DP_MBS - 1], true);
				else dbkp_expand_list(f.motion, L.hdr[j][5], L.hdr[j][6], l, L.mo[LDS::TOPC && j > DP_MBS ? 0 : j]);
				const uint32_t h = L.hdr[j][6];
				if (l && (E264_MOT_UNI(h, 1) || (h >> 4 & 15u))) dbkp_note_l1(&L.any_l1); // (E264_MOT_USED(h, 4..7): the quadrants of list 1)
			}
		}
	if (tid < DP_MBS) {
		cslice_t s = f.slices + (L.hdr[1 + tid][7] & 0xffff); // E264Mb.dbk_slice
		L.fo[tid][0] = s->FilterOffsetA; L.fo[tid][1] = s->FilterOffsetB;
	}
}

// |a - b| of both 16-bit halves at once; A, B: int16 pairs biased by 0x8000 (unsigned order), so the saturating unsigned
// subtractions are exact for any pair of vectors
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
E264_DEV uint32_t dbkp_absdiff2(uint32_t A, uint32_t B)
{
	const u16x2 a = __builtin_bit_cast(u16x2, A), b = __builtin_bit_cast(u16x2, B);
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(a, b)) | __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(b, a));
}
#define DBKP_FAR 0xfffcfffcu // bits of dbkp_absdiff2 that say "4 quarter samples or more" (edge264_deblock.c:981-991)
#define DBKP_BIAS 0x80008000u

// The four bS of edge e in direction dir of a macroblock, one per byte (edge264_deblock.c:958-1118).  hm / mm: the
// macroblock's E264Mb dwords and expanded motion; L / T: the left / top neighbour's (anything when the edge flag is off);
// `on`: the macroblock is deblocked at all.
// Round 2 computed one bS per task (32 tasks per macroblock): neighbour selection, flags and the block numbering were
// worked out 32 times, every vector difference in 32-bit arithmetic -- the kernel ran at the VALU issue limit (170 M
// wave-instructions per 256 pictures, profiles/r03_pmc_sq_instruction_mix.txt).
// L1 = false (round 5): no record of the workgroup predicts from list 1 (every workgroup of a P picture): the list-1 halves of the comparison are
// constants -- references 0xff on both sides, vectors 0 -- so "same lists" reduces to the list-0 reference and vector, and "crossed lists" is always a
// difference (a used list-0 slot against 0xff).  A quarter of the vector arithmetic of the general form.
// TOPC: the top neighbour's motion (mT) is in the 9-dword form of DbkpLdsT::mot
template <bool L1, bool TOPC>
E264_DEV uint32_t dbkp_bs4(const uint32_t *hm, const uint32_t *mm, const uint32_t *hL, const uint32_t *mL, const uint32_t *hT, const uint32_t *mT,
	bool has_motion, bool on, int dir, int e)
{
	const uint32_t h0 = hm[0];
	const int kind = h0 & 255, flags = h0 >> 8 & 255;
	const bool intra = kind != E264_MB_INTER;
	const bool has_edge = e != 0 || (flags & (dir ? E264_MBF_EDGE_TOP : E264_MBF_EDGE_LEFT));
	const bool skip8 = e != 0 && (flags & E264_MBF_T8x8) && (e & 1);
	if (!on || !has_edge || skip8)
		return 0;
	const uint32_t *hn = e == 0 ? (dir ? hT : hL) : hm, *mn = e == 0 ? (dir ? mT : mL) : mm; // record holding the p side
	if (e == 0 && (intra || (hn[0] & 255) != E264_MB_INTER))
		return 0x04040404u;
	if (intra)
		return 0x03030303u;
	// both sides are inter macroblocks.  Blocks in zig order: q side x = e (dir 0) or y = e (dir 1), the p side one column / row before
	const int pe = (e + 3) & 3;
	const int qb = dir ? (e >> 1) * 8 + (e & 1) * 2 : (e >> 1) * 4 + (e & 1);
	const int pb = dir ? (pe >> 1) * 8 + (pe & 1) * 2 : (pe >> 1) * 4 + (pe & 1);
	const uint32_t offs = dir ? 0x5410u : 0xa820u; // + what the segment adds: x = 0..3 / y = 0..3
	const uint32_t nzq = hm[2] & 0xffffu, nzp = hn[2] & 0xffffu;
	uint32_t out = 0;
#pragma unroll
	for (int sg = 0; sg < 4; sg++) {
		const int o = (int)(offs >> (4 * sg) & 15u), kq = qb + o, kp = pb + o;
		uint32_t bs = ((nzp >> kp | nzq >> kq) & 1u) ? 2u : 0u;
		if (has_motion) { // (uniform) references as bytes (unused list: 0xff), vectors biased for dbkp_absdiff2 (unused list: 0)
			const int sq = (kq >> 2) * 8;
			// where the p side's references and vectors are: the full form, or (top edge, TOPC) the top neighbour's bottom-row form: kp is 10, 11, 14 or 15 there
			const bool top = TOPC && e == 0 && dir == 1;
			const int tq = (kp >> 2) & 1, ci = tq * 2 + (kp & 1);
			const int sp = top ? 8 * tq : (kp >> 2) * 8, sp1 = top ? 16 + 8 * tq : sp;
			const int ip0 = top ? 1 + ci : 4 + kp, ip1 = top ? 5 + ci : 20 + kp, ir1 = top ? 0 : 1;
			if (L1) {
				const uint32_t q0 = mm[0] >> sq & 255u, q1 = mm[1] >> sq & 255u, p0 = mn[0] >> sp & 255u, p1 = mn[ir1] >> sp1 & 255u;
				const uint32_t vq0 = mm[4 + kq] ^ DBKP_BIAS, vq1 = mm[20 + kq] ^ DBKP_BIAS, vp0 = mn[ip0] ^ DBKP_BIAS, vp1 = mn[ip1] ^ DBKP_BIAS;
				// same lists, or crossed (deblock.c:913-925): a difference either way
				const uint32_t par = (p0 ^ q0) | (p1 ^ q1) | ((dbkp_absdiff2(vp0, vq0) | dbkp_absdiff2(vp1, vq1)) & DBKP_FAR);
				const uint32_t crs = (p0 ^ q1) | (p1 ^ q0) | ((dbkp_absdiff2(vp0, vq1) | dbkp_absdiff2(vp1, vq0)) & DBKP_FAR);
				if (bs == 0) bs = (par != 0 && crs != 0) ? 1u : 0u;
			} else { // list 0 only on both sides (the bias cancels in the difference)
				const uint32_t q0 = mm[0] >> sq & 255u, p0 = mn[0] >> sp & 255u;
				const uint32_t par = (p0 ^ q0) | (dbkp_absdiff2(mn[ip0] ^ DBKP_BIAS, mm[4 + kq] ^ DBKP_BIAS) & DBKP_FAR);
				if (bs == 0) bs = par != 0 ? 1u : 0u;
			}
		}
		out |= bs << (8 * sg);
	}
	return out;
}
// alpha, beta, indexA (edge264_deblock.c:945-955) of plane pl against neighbour class t = 0 internal edges, 1 left, 2 top
E264_DEV void dbkp_ab3(const uint32_t *hm, const uint32_t *hL, const uint32_t *hT, int foA, int foB, const uint8_t *alpha, const uint8_t *beta, bool on, int pl, int t,
	int &a, int &b, int &ia)
{
	const uint32_t h0 = hm[0], h1 = hm[1];
	const int flags = h0 >> 8 & 255;
	const uint32_t *hn = t == 0 ? hm : t == 2 ? hT : hL;
	const uint32_t n0 = hn[0], n1 = hn[1];
	const int qm = pl == 0 ? (int)(h0 >> 16 & 255) : pl == 1 ? (int)(h0 >> 24) : (int)(h1 & 255);
	const int qn = pl == 0 ? (int)(n0 >> 16 & 255) : pl == 1 ? (int)(n0 >> 24) : (int)(n1 & 255);
	const bool use_nb = (t == 1 && (flags & E264_MBF_EDGE_LEFT)) || (t == 2 && (flags & E264_MBF_EDGE_TOP));
	const int qPav = (qm + (use_nb ? qn : qm) + 1) >> 1;
	const int iA = min(max(qPav + foA, 0), 51), iB = min(max(qPav + foB, 0), 51);
	a = on ? alpha[iA] : 0; b = on ? beta[iB] : 0; ia = on ? iA : 0;
}

// four threads per macroblock: thread r computes the bS of two (direction, edge) pairs -- 8 bytes of the record -- and, r < 3,
// alpha / beta / indexA of plane r; r == 3 clears the tail of the record
template <class LDS> E264_DEV void dbkp_phase_compute(LDS &L, const FrameCtx &f, int a0, int tid)
{
	const bool has_motion = f.motion != nullptr;
	const int n_mbs = f.wm * f.hm;
	const int i = tid >> 2, r = tid & 3;
	const int rm = 1 + i, rl = i, rt = 1 + DP_MBS + i; // own record, left and top neighbours
	const uint32_t h0 = L.hdr[rm][0];
	const bool on = a0 + i < n_mbs && (h0 >> 8 & E264_MBF_DEBLOCK) && (h0 & 255) != E264_MB_ABSENT;
	const int dir = r >> 1, e0 = (r & 1) * 2;
	v2u bs;
	const uint32_t *mtop = LDS::TOPC ? L.mot[LDS::TOPC ? i : 0] : L.mo[LDS::TOPC ? 0 : rt]; // the top neighbour's motion, in the form this LDS keeps it
	if (LDS::HAS_L1 && L.any_l1) { // (uniform over the workgroup)
		bs.x = dbkp_bs4<true, LDS::TOPC>(L.hdr[rm], L.mo[rm], L.hdr[rl], L.mo[rl], L.hdr[rt], mtop, has_motion, on, dir, e0);
		bs.y = dbkp_bs4<true, LDS::TOPC>(L.hdr[rm], L.mo[rm], L.hdr[rl], L.mo[rl], L.hdr[rt], mtop, has_motion, on, dir, e0 + 1);
	} else {
		bs.x = dbkp_bs4<false, LDS::TOPC>(L.hdr[rm], L.mo[rm], L.hdr[rl], L.mo[rl], L.hdr[rt], mtop, has_motion, on, dir, e0);
		bs.y = dbkp_bs4<false, LDS::TOPC>(L.hdr[rm], L.mo[rm], L.hdr[rl], L.mo[rl], L.hdr[rt], mtop, has_motion, on, dir, e0 + 1);
	}
	*(v2u *)&L.out[i][2 * r] = bs;
	uint8_t *o8 = (uint8_t *)&L.out[i][0];
	if (r < 3) {
#pragma unroll
		for (int t = 0; t < 3; t++) {
			int a, b, ia;
			dbkp_ab3(L.hdr[rm], L.hdr[rl], L.hdr[rt], L.fo[i][0], L.fo[i][1], L.alpha, L.beta, on, r, t, a, b, ia);
			o8[32 + r * 3 + t] = (uint8_t)a; o8[41 + r * 3 + t] = (uint8_t)b; o8[50 + r * 3 + t] = (uint8_t)ia;
		}
	} else {
#pragma unroll
		for (int k = 59; k < 64; k++) o8[k] = 0;
	}
}

// One piece of the lanes' layout out of a raw record (see the head of this file; the selection of bS / alpha / indexA per slot is
// what e264_dbk.h's dk_params did per lane and step until round 5).  s: luma segment / chroma line pair, 0..3.
E264_DEV uint32_t dk_bfi_u(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); } // v_bfi_b32
E264_DEV v2u dbkp_piece(const uint8_t *rec, const uint8_t *tc0tab, bool chroma, int dir, int s)
{
	uint32_t al = 0, tc = 0;
#pragma unroll
	for (int e = 0; e < 4; e++) {
		const bool mbe = chroma ? !(e & 1) : e == 0;                              // a macroblock edge: its own alpha / beta / indexA
		const int bso = (chroma ? (e & 1) * 8 : e * 4) + s;                       // chroma inner edge = luma edge 2
		const int abi = (chroma ? (1 + (e >> 1)) * 3 : 0) + (mbe ? 1 + dir : 0);  // [plane * 3 + {inner, left, top}]
		const uint32_t bS = rec[dir * 16 + bso], alpha = rec[32 + abi], ia = rec[50 + abi];
		al |= (bS ? alpha : 0u) << (8 * e);
		tc |= ((uint32_t)tc0tab[(bS & 3) * 52 + ia] + (chroma ? 1u : 0u)) << (8 * e);
		if ((e == 0 || e == 2) && bS == 4) tc |= 0x80u << (8 * e);
	}
	return (v2u){al, tc};
}
// the 16 bytes of a record that belong to the whole macroblock: beta[plane * 3 + {inner, left, top}] (raw bytes 41..49)
E264_DEV v4u dbkp_mbwide(const uint8_t *rec)
{
	const uint32_t y = (uint32_t)rec[41] | (uint32_t)rec[42] << 8 | (uint32_t)rec[43] << 16;
	const uint32_t c0 = (uint32_t)rec[44] | (uint32_t)rec[45] << 8 | (uint32_t)rec[46] << 16 | (uint32_t)rec[47] << 24, c1 = (uint32_t)rec[48] | (uint32_t)rec[49] << 8;
	return (v4u){y, 0, c0, c1};
}
// after the raw records: thread r of a macroblock builds the four pieces of direction r >> 1, segments / line pairs 2 (r & 1), 2 (r & 1) + 1.
// dbkp_piece above is the definition (and what the host tests compare with); this is the same on four slots at a time: the bS of a piece's
// four slots gathered into one dword, "bS != 0" as a byte mask, and the tC0 of all four out of ONE byte permute whose selector is bS itself
// (L.tc3[indexA] = {0, tC0 of bS 1, 2, 3}: the macroblock-edge entry in the low source, the inner-edge entry in the high one).
E264_DEV uint32_t dbkp_byte(const uint32_t *w, int k) { return w[k >> 2] >> (8 * (k & 3)) & 255u; } // byte k of the raw record's dwords 8..15 (k - 32)
template <class LDS> E264_DEV void dbkp_phase_pieces(LDS &L, int tid)
{
	const int i = tid >> 2, r = tid & 3, dir = r >> 1, s0 = (r & 1) * 2;
	const v4u B = *(const v4u *)&L.out[i][dir * 4]; // bS of edges 0..3, one segment per byte
	const v4u w0 = *(const v4u *)&L.out[i][8], w1 = *(const v4u *)&L.out[i][12];
	const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
	uint32_t alv[3], t_mb[3], t_in[3];
#pragma unroll
	for (int pl = 0; pl < 3; pl++) { // [plane * 3 + {inner, left, top}] at bytes 32 (alpha), 41 (beta), 50 (indexA) of the record
		const uint32_t a_in = dbkp_byte(w, 3 * pl), a_mb = dir ? dbkp_byte(w, 3 * pl + 2) : dbkp_byte(w, 3 * pl + 1);
		const uint32_t i_in = dbkp_byte(w, 18 + 3 * pl), i_mb = dir ? dbkp_byte(w, 18 + 3 * pl + 2) : dbkp_byte(w, 18 + 3 * pl + 1);
		alv[pl] = a_mb | a_in << 8; // (macroblock edge, inner edges)
		t_mb[pl] = L.tc3[i_mb]; t_in[pl] = L.tc3[i_in];
	}
	const uint32_t al_l = v_perm(0, alv[0], 0x01010100u);        // luma slots: mb, inner, inner, inner
	const uint32_t al_c = v_perm(alv[2], alv[1], 0x05040100u);   // chroma slots: Cb mb, Cb inner, Cr mb, Cr inner
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const uint32_t sg = (uint32_t)(s0 + k);
		{ // luma: slot e = edge e
			const uint32_t sel = sg * 0x0101u + 0x0400u; // bytes (sg of the low source, sg of the high source)
			const uint32_t bsv = v_perm(v_perm(B.w, B.z, sel), v_perm(B.y, B.x, sel), 0x05040100u);
			const uint32_t nz7 = (bsv + 0x7f7f7f7fu) & 0x80808080u, full = (nz7 << 1) - (nz7 >> 7); // 0xff where bS != 0 (bS <= 4: no carry between bytes)
			const uint32_t tc = v_perm(t_in[0], t_mb[0], (bsv & 0x03030303u) + 0x04040400u);
			*(v2u *)&L.pieces[i][(s0 + k) * 4 + dir * 2] = (v2u){al_l & full, tc | (bsv & 0x00040004u) << 5}; // bS == 4 <=> bit 2 (bS is 0..4) -> bit 7
		}
		{ // chroma: slots (Cb edge 0, Cb edge 2, Cr edge 0, Cr edge 2)
			const uint32_t sel = sg * 0x01010101u + 0x04000400u;
			const uint32_t bsv = v_perm(B.z, B.x, sel);
			const uint32_t nz7 = (bsv + 0x7f7f7f7fu) & 0x80808080u, full = (nz7 << 1) - (nz7 >> 7);
			const uint32_t tsel = (bsv & 0x03030303u) + 0x04000400u;
			const uint32_t tc = dk_bfi_u(0x0000ffffu, v_perm(t_in[1], t_mb[1], tsel), v_perm(t_in[2], t_mb[2], tsel)) + 0x01010101u;
			*(v2u *)&L.pieces[i][16 + (s0 + k) * 4 + dir * 2] = (v2u){al_c & full, tc | (bsv & 0x00040004u) << 5};
		}
	}
	if (r == 0) // beta: raw bytes 41..49 = dwords 10..12 of the record
		*(v4u *)&L.pieces[i][32] = (v4u){dbkp_byte(w, 9) | dbkp_byte(w, 10) << 8 | dbkp_byte(w, 11) << 16, 0u,
		                                 dbkp_byte(w, 12) | dbkp_byte(w, 13) << 8 | dbkp_byte(w, 14) << 16 | dbkp_byte(w, 15) << 24, dbkp_byte(w, 16) | dbkp_byte(w, 17) << 8};
}

template <class LDS> E264_DEV void dbkp_phase_store(const LDS &L, const FrameCtx &f, int a0, int tid)
{
	const int n_mbs = f.wm * f.hm;
	constexpr int P16 = E264_DBK_BYTES / 16; // 16-byte pieces per record: consecutive threads write consecutive pieces of consecutive records
#ifdef E264_ABL_DBKP_STORE1 // timing ablation: part of every record is written (wrong parameters): what the records cost in stores
	for (int idx = tid; idx < DP_MBS * 4; idx += DP_NT) {
#else
	for (int idx = tid; idx < DP_MBS * P16; idx += DP_NT) {
#endif
		const int i = idx / P16, part = idx - i * P16;
		if (a0 + i < n_mbs)
			*(gv4u *)(f.dbk + (size_t)(a0 + i) * E264_DBK_BYTES + part * 16) = *(const v4u *)&L.pieces[i][part * 4];
	}
}

} // namespace
#endif
