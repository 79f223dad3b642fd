// e264_pred.h -- e264_pred_kernel: inter prediction + residual of every inter / PCM macroblock (second generation of the
// macroblock-parallel kernel; replaces the strip-per-wave e264_mbpar_kernel of round 1).
//
// Device restatement of (file:line in /root/reference/src):
//   decode_inter            edge264_inter.c:1108-1251   partition -> reference window, weights, edge emulation
//   decode_inter_luma       edge264_inter.c:416-968     6-tap luma, 16 quarter-sample positions
//   decode_inter_chroma     edge264_inter.c:977-1091    bilinear chroma
//   add_idct4x4 / add_dc4x4 / add_idct8x8 / transform_dc2x2   edge264_residual.c:108-343, 456-538
//   I_PCM                   edge264_slice.c:914-935
//
// Why a rewrite: round 1 walked one macroblock per wave (4 samples per lane per step); its counters showed 430 VALU +
// 240 SALU wave-instructions per macroblock of which the filter taps were a tenth -- the kernel was bound by issuing
// per-macroblock bookkeeping, window addressing and LDS round trips (profiles/r01k_ablation_breakdown.txt).  Here:
//
//   * ONE LANE = ONE 8x8 LUMA BLOCK (+ its 4x4 Cb and Cr): 96 samples per lane per item, 16 macroblocks per wave-pass.
//     The lane fetches its OWN 13x13 reference window (13 rows x one 16-byte load, served by L1/L2) and keeps it in
//     registers: rows, columns, horizontal and vertical taps are all lane-private, so there is no LDS round trip, no
//     cross-lane traffic and no wave-level pipeline state at all.  Overlapping windows of neighbouring lanes are cache
//     hits by construction ("6-tap halos are cache hits by definition").
//   * A WORKGROUP = A TILE of 16 x 4 macroblocks, 256 threads (round 2: 16 x 8, 512 threads; four 37-KB workgroups per CU
//     instead of two 76-KB ones interleave their fetch-bound and arithmetic-bound phases better: 1.43 -> 1.35 ms, 16 x 2 the
//     same -- profiles/r03_ablations.txt).  The tile's samples are assembled in LDS (24 KB) and leave as whole 256-byte luma /
//     128-byte chroma rows (round 1 measured 3.6x write amplification for 16-byte row pieces).
//   * CLASS-SORTED WORK LISTS.  The 16 quarter-sample positions need three different data flows (reference
//     edge264_inter.c: "horizontal then vertical" for xFrac == 2, "vertical then horizontal" for yFrac == 2, one-dimensional
//     otherwise).  The items of a tile are binned by that class in LDS (one atomic per item) and lanes take items in class
//     order, so a wave runs one flow (two at a class boundary) instead of the union of all three.
//   * The residual is a second, COMPACTED pass: one lane = one coded 4x4 (or 8x8) block, transformed entirely in registers
//     and added to the tile in LDS.  Macroblocks without coefficients cost nothing (round 1: the whole wave walked the
//     transform of every coded macroblock with a quarter of its lanes busy).
//
// The kernel is a sequence of PHASES separated by workgroup barriers; threads only communicate through LDS between
// phases.  Each phase is a plain function of (LDS, frame, tile, thread id): tests/emu/ compiles this very file for the
// host, runs the phases thread by thread and compares the tile with the CPU oracle (tests/test_pred_emu.py).
#ifndef E264_PRED_H
#define E264_PRED_H
#include "e264_dev.h"

namespace {

#ifndef E264_HOST_INTRINSICS
E264_DEV int lds_add(int *p, int v) { return atomicAdd(p, v); }
E264_DEV void lds_or(uint32_t *p, uint32_t v) { atomicOr(p, v); }
#endif

#define PT_W 16                 // tile width in macroblocks (256-byte luma rows, 128-byte chroma rows)
#ifndef E264_PT_H
#define E264_PT_H 4             // tile height in macroblocks (build-time: `make variant DEFS=-DE264_PT_H=8` = 512-thread workgroups, 2 per CU)
#endif
#define PT_H E264_PT_H
#define PT_MBS (PT_W * PT_H)
#define PT_NT (PT_MBS * 4)      // one thread per 8x8 quadrant
#define PT_LIST (PT_MBS * 16)   // items per class list: every quadrant split into four 4x4 partitions

struct __attribute__((aligned(16))) PredLds {
	uint32_t y[PT_H * 16][PT_W * 4];    // luma samples of the tile
	uint32_t c[2][PT_H * 8][PT_W * 2];  // Cb, Cr
	uint32_t hdr[PT_MBS][8];            // E264Mb records (kind 0 = ABSENT for macroblocks outside the frame)
	uint16_t lst[3][PT_LIST];           // prediction items by class; then the residual lists (4x4: lst[0..1], 8x8: lst[2])
	uint32_t mvq[4][PT_NT];             // motion of every quadrant for the list being predicted: its four 4x4 vectors ...
	uint32_t refq[PT_NT];               // ... and DPB slot | refIdx << 8 | refIdx of the other list << 16 | weighted_bipred_idc << 24
	int cnt[6];                         // items per class
	int rcnt[2];                        // residual items: 4x4 blocks, 8x8 blocks
	uint32_t staged[PT_MBS / 32];       // macroblocks this kernel writes (inter, PCM)
	int any_l1;                         // some quadrant of the tile uses list 1
	int n_inter, n_uni;                 // inter macroblocks of the tile; of which one partition per list (16x16, P_Skip, B_Skip / direct with one vector)
	generic_u8p dpb[E264_MAX_SLOTS];
#ifdef E264_PRED_LDS_PAD // measuring aid: what the kernel does at a lower occupancy (bytes of LDS nobody uses)
	uint8_t pad[E264_PRED_LDS_PAD];
#endif
};

struct PredTile { int tx0, ty0; };     // first macroblock of the tile

// prediction item: bits 0..8 quadrant slot (macroblock of the tile * 4 + 8x8 index), 9..10 4x4 block of the quadrant the
// partition starts at, 11..12 shape (0 8x8, 1 8x4, 2 4x8, 3 4x4).  The class is implied by the list.
#define PRED_NO_REF 0xffffffffu // L.refq of a quadrant the list does not predict (a DPB slot is 0..31: no valid word looks like it)
#define PI_SLOT(d) ((d) & 511)
#define PI_SUB(d) ((d) >> 9 & 3)
#define PI_SHAPE(d) ((d) >> 11 & 3)

// ---------------------------------------------------------------------------------------------------------------------
// phase 0: headers of the tile, counters
// ---------------------------------------------------------------------------------------------------------------------
E264_DEV void pred_phase_setup(PredLds &L, const FrameCtx &f, const PredTile &t, int tid)
{
	if (tid < 6) L.cnt[tid] = 0;
	if (tid < 2) L.rcnt[tid] = 0;
	if (tid < PT_MBS / 32) L.staged[tid] = 0;
	if (tid == 0) { L.any_l1 = 0; L.n_inter = 0; L.n_uni = 0; }
	if (tid < E264_MAX_SLOTS) L.dpb[tid] = f.dpb[tid];
	const gu32 *mbs_g = (const gu32 *)f.mbs_g;
	for (int i = tid; i < PT_MBS * 8; i += PT_NT) {
		const int mb = i >> 3, mbx = t.tx0 + (mb & (PT_W - 1)), mby = t.ty0 + mb / PT_W;
		uint32_t v = 0;
		if (mbx < f.wm && mby < f.hm) v = mbs_g[(size_t)(mby * f.wm + mbx) * 8 + (i & 7)];
		L.hdr[mb][i & 7] = v;
	}
}

// The intra bitmap of the tile's rows (e264_kernels.h E264_BITMAP_OFF): one uint16_t per macroblock row of the tile, bit i = macroblock tx0 + i is
// Intra4x4 / 8x8 / 16x16 and this packet's to reconstruct -- the very test e264_intra_kernel's scan makes on the records (e264_intra.h), made here
// where the records already are in LDS, so that the intra kernel of this submission can leave rows and chunks without intra macroblocks alone.
// Every tile of the picture writes its entries, the tiles of an I picture before they leave.
static_assert(PT_W == 16, "the bitmap has one 16-bit entry per tile row");
E264_DEV void pred_phase_bitmap(const PredLds &L, const FrameCtx &f, const PredTile &t, int tid)
{
	if (tid >= PT_H || !f.dbk || t.ty0 + tid >= f.hm)
		return;
	uint32_t bits = 0;
#pragma unroll
	for (int i = 0; i < PT_W; i++) {
		const uint32_t d0 = L.hdr[tid * PT_W + i][0];
		const int kind = (d0 >> 8 & E264_MBF_DONE) ? E264_MB_ABSENT : (int)(d0 & 255);
		if (kind == E264_MB_I4x4 || kind == E264_MB_I8x8 || kind == E264_MB_I16x16) bits |= 1u << i;
	}
	gu16 *bitmap = (gu16 *)(f.dbk + E264_BITMAP_OFF(f.wm * f.hm));
	bitmap[(size_t)(t.ty0 + tid) * ((f.wm + 15) >> 4) + (t.tx0 >> 4)] = (uint16_t)bits;
}

// ---------------------------------------------------------------------------------------------------------------------
// phase 1: one thread per quadrant: partition shape of list `list`, items into the class lists; PCM samples
// ---------------------------------------------------------------------------------------------------------------------
// Class of a partition = the data flow its quarter-sample position needs (edge264_inter.c:416-968):
//   0 G   integer position                      3 D   both fractions odd: b of one row averaged with h of one column
//   1 B   yFrac == 0: b (+ G)                   4 JH  xFrac == 2, yFrac != 0: horizontal taps, then sixtapHV down the columns
//   2 H   xFrac == 0: h (+ G)                   5 JV  yFrac == 2, xFrac odd: vertical taps, then sixtapHV along the rows
// Lists: class c and class 5 - c share one array (c from the front, 5 - c from the back): the six lists together never hold
// more than PT_LIST items.
E264_DEV int pred_class(uint32_t mv)
{
	const int xF = mv & 3, yF = mv >> 16 & 3;
	if (yF == 0) return xF == 0 ? 0 : 1;
	if (xF == 0) return 2;
	if (xF == 2) return 4;
	return yF == 2 ? 5 : 3;
}
E264_DEV uint16_t *pred_list_slot(PredLds &L, int cls, int idx)
{
	return cls < 3 ? &L.lst[cls][idx] : &L.lst[5 - cls][PT_LIST - 1 - idx];
}
E264_DEV void pred_push(PredLds &L, uint32_t mv, int desc)
{
	const int cls = pred_class(mv);
	*pred_list_slot(L, cls, lds_add(&L.cnt[cls], 1)) = (uint16_t)desc;
}
E264_DEV void pred_phase_classify(PredLds &L, const FrameCtx &f, const PredTile &t, int list, int tid)
{
	const int mb = tid >> 2, q = tid & 3;
	const uint32_t d0 = L.hdr[mb][0];
	const int kind = (d0 >> 8 & E264_MBF_DONE) ? E264_MB_ABSENT : (int)(d0 & 255); // DONE: written by an earlier packet of the picture, not ours to touch
	const int mbx = t.tx0 + (mb & (PT_W - 1)), mby = t.ty0 + mb / PT_W;
	L.refq[tid] = PRED_NO_REF; // until proven otherwise this list does not predict the quadrant (pred_item looks at the neighbour's word)
	if (list == 0 && (kind == E264_MB_INTER || kind == E264_MB_PCM) && q == 0)
		lds_or(&L.staged[mb >> 5], 1u << (mb & 31));
	if (list == 0 && kind == E264_MB_PCM) { // edge264_slice.c:914-935: this quadrant's 8x8 luma + 4x4 Cb + 4x4 Cr
		const gu8 *pl = f.payload + L.hdr[mb][4];
		const int qx = q & 1, qy = q >> 1;
		for (int r = 0; r < 8; r++) {
			const gu32 *s = (const gu32 *)(pl + (qy * 8 + r) * 16 + qx * 8);
			uint32_t *d = &L.y[(mb / PT_W) * 16 + qy * 8 + r][(mb & (PT_W - 1)) * 4 + qx * 2];
			d[0] = s[0]; d[1] = s[1];
		}
		for (int p = 0; p < 2; p++)
			for (int r = 0; r < 4; r++)
				L.c[p][(mb / PT_W) * 8 + qy * 4 + r][(mb & (PT_W - 1)) * 2 + qx] = *(const gu32 *)(pl + 256 + p * 64 + (qy * 4 + r) * 8 + qx * 4);
	}
	if (kind != E264_MB_INTER || !f.motion)
		return;
	const uint32_t mot_off = L.hdr[mb][5], mot_hdr = L.hdr[mb][6]; // E264Mb.modes of an inter macroblock: its motion directory
	if (list == 0 && (E264_MOT_UNI(mot_hdr, 1) || E264_MOT_USED(mot_hdr, 4 + q)))
		L.any_l1 = 1;
	if (list == 0 && q == 0) { // what the tile is made of: decides how its chroma is fetched (pred_pair_chroma)
		lds_add(&L.n_inter, 1);
		const bool u0 = E264_MOT_UNI(mot_hdr, 0), u1 = E264_MOT_UNI(mot_hdr, 1), any0 = (mot_hdr & 15u) != 0, any1 = (mot_hdr >> 4 & 15u) != 0;
		if ((u0 || !any0) && (u1 || !any1) && (u0 || u1)) lds_add(&L.n_uni, 1);
	}
	uint32_t refword, mvq[4];
	if (!mot_quadrant(f.motion, mot_off, mot_hdr, list, q, refword, mvq))
		return;
	const int pic = (int)(int8_t)refword;
	if (pic < 0)
		return;
	uint32_t otherref = 0xffffffffu, dummy[4]; // refIdx of the other list decides how the two predictions combine
	if (E264_MOT_UNI(mot_hdr, list ^ 1) || E264_MOT_USED(mot_hdr, (list ^ 1) * 4 + q))
		mot_quadrant(f.motion, mot_off, mot_hdr, list ^ 1, q, otherref, dummy);
	const v4u m = {mvq[0], mvq[1], mvq[2], mvq[3]}; // the quadrant's four 4x4 vectors: (0,0) (4,0) (0,4) (4,4)
	const int base = tid;
	// everything the prediction of this quadrant needs goes to LDS here, so that the item phase starts with ONE round
	// trip to memory (its reference windows) instead of a chain motion -> slot table -> window.
	// (Tried: entering the quadrants of one 16x16 / 16x8 partition as neighbours of the class list so that their row
	// fetches share cache lines within an instruction -- no gain, tools/calib/load_rate.hip: the vector memory path takes
	// 1.5 - 2.5 cycles per lane-row whether the lanes are neighbours or scattered.)
	cslice_t s = f.slices + (L.hdr[mb][2] >> 16);
	L.mvq[0][tid] = m.x; L.mvq[1][tid] = m.y; L.mvq[2][tid] = m.z; L.mvq[3][tid] = m.w;
	L.refq[tid] = (refword & 0xffffu) | (otherref >> 8 & 255u) << 16 | (uint32_t)(uint8_t)s->weighted_bipred_idc << 24;
	if (m.x == m.y && m.x == m.z && m.x == m.w) {
		pred_push(L, m.x, base);
	} else if (m.x == m.y && m.z == m.w) { // two 8x4
		pred_push(L, m.x, base | 1 << 11);
		pred_push(L, m.z, base | 2 << 9 | 1 << 11);
	} else if (m.x == m.z && m.y == m.w) { // two 4x8
		pred_push(L, m.x, base | 2 << 11);
		pred_push(L, m.y, base | 1 << 9 | 2 << 11);
	} else {
		pred_push(L, m.x, base | 3 << 11);
		pred_push(L, m.y, base | 1 << 9 | 3 << 11);
		pred_push(L, m.z, base | 2 << 9 | 3 << 11);
		pred_push(L, m.w, base | 3 << 9 | 3 << 11);
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// phase 2: prediction items
// ---------------------------------------------------------------------------------------------------------------------
// Reference rows: 16 bytes per row starting at the dword that holds column X (X may be anywhere: the reference's edge
// emulation == per-sample clamp of the coordinates, edge264_inter.c:1199-1235).  Frame widths are multiples of 16 (8 for
// chroma), so an aligned dword is either entirely inside the frame or entirely outside; outside dwords replicate the
// edge sample.
E264_DEV uint32_t ref_dword(const gu8 *row, int x, int Wb)
{ // Wb: width of the plane in bytes; x: dword-aligned column, possibly outside
	const int xc = min(max(x, 0), Wb - 4);
	uint32_t v = *(const gu32 *)(row + xc);
	if (x < 0) v = v_perm(v, v, 0x00000000u);       // byte 0 four times (v_perm_b32; `(v & 255) * 0x01010101` is a quarter-rate multiply)
	if (x > Wb - 4) v = v_perm(v, v, 0x03030303u);  // byte 3 four times
	return v;
}
struct Row4 { uint32_t a0, a1, a2, a3; };
// A register that holds SOMETHING, at no cost: rows a partition does not need are not fetched (the fetch is what this kernel's time
// is, profiles/r03_ablations.txt item 19) but still run through the arithmetic of the wave, whose results for them are dropped.
#ifndef E264_HOST_INTRINSICS
E264_DEV uint32_t any_u32() { uint32_t v; asm("" : "=v"(v)); return v; }
#else
E264_DEV uint32_t any_u32() { return 0xa55a5aa5u; } // (host build: something that would show in a result that depended on it)
#endif
// NR x 16 bytes around (X, Y): the loads (nothing here uses a loaded value) ...  all: the partition is 8 rows high; else (4 rows)
// the last four rows of the window are not needed and not fetched
template <int NR>
E264_DEV void load_window(const gu8 *plane, int sY, int W, int H, int X, int Y, bool all, Row4 *A)
{
	const int XA = X & ~3;
#pragma unroll
	for (int r = NR - 4; r < NR; r++) { A[r].a0 = any_u32(); A[r].a1 = any_u32(); A[r].a2 = any_u32(); A[r].a3 = any_u32(); }
#ifdef E264_ABL_NOEDGE // timing ablation: no edge emulation (windows outside the frame read the wrong samples)
	if (true) {
		const gu8 *p = plane + (size_t)min(max(Y, 0), H - NR) * sY + min(max(XA, 0), W - 16);
#pragma unroll
		for (int r = 0; r < NR; r++) {
			const v4u v = *(const gv4u *)(p + (size_t)r * sY);
			A[r].a0 = v.x; A[r].a1 = v.y; A[r].a2 = v.z; A[r].a3 = v.w;
		}
	} else
#endif
	if (XA >= 0 && XA + 12 <= W - 4 && Y >= 0 && Y + NR - 1 <= H - 1) { // the common case: inside the frame, one 16-byte load per row
		const gu8 *p = plane + (size_t)Y * sY + XA;
#pragma unroll
		for (int r = 0; r < NR; r++) {
			if (r >= NR - 4 && !all)
				continue;
			const v4u v = *(const gv4u *)(p + (size_t)r * sY);
			A[r].a0 = v.x; A[r].a1 = v.y; A[r].a2 = v.z; A[r].a3 = v.w;
		}
	} else {
#pragma unroll
		for (int r = 0; r < NR; r++) {
			if (r >= NR - 4 && !all)
				continue;
			const gu8 *row = plane + (uint32_t)mul24(min(max(Y + r, 0), H - 1), sY);
			A[r].a0 = ref_dword(row, XA, W); A[r].a1 = ref_dword(row, XA + 4, W); A[r].a2 = ref_dword(row, XA + 8, W); A[r].a3 = ref_dword(row, XA + 12, W);
		}
	}
}
// Rows lo .. hi-1 of the 13-row window at (X, Y), the others left alone (per lane: the one-dimensional classes run in ONE flow, its
// integer / horizontal-only lanes need rows 2..9, the others 0..12; a partition 4 rows high needs 4 rows fewer).  LOADS ONLY.
E264_DEV void load_window_rows(const gu8 *plane, int sY, int W, int H, int X, int Y, int lo, int hi, Row4 *A)
{
	const int XA = X & ~3;
#pragma unroll
	for (int r = 0; r < 13; r++) { A[r].a0 = any_u32(); A[r].a1 = any_u32(); A[r].a2 = any_u32(); A[r].a3 = any_u32(); }
	if (XA >= 0 && XA + 12 <= W - 4 && Y + lo >= 0 && Y + hi - 1 <= H - 1) { // inside the frame: one 16-byte load per row
		const gu8 *p = plane + (ptrdiff_t)Y * sY + XA;
#pragma unroll
		for (int r = 0; r < 13; r++) {
			if (r < lo || r >= hi)
				continue;
			const v4u v = *(const gv4u *)(p + (ptrdiff_t)r * sY);
			A[r].a0 = v.x; A[r].a1 = v.y; A[r].a2 = v.z; A[r].a3 = v.w;
		}
	} else {
#pragma unroll
		for (int r = 0; r < 13; r++) {
			if (r < lo || r >= hi)
				continue;
			const gu8 *row = plane + (uint32_t)mul24(min(max(Y + r, 0), H - 1), sY);
			A[r].a0 = ref_dword(row, XA, W); A[r].a1 = ref_dword(row, XA + 4, W); A[r].a2 = ref_dword(row, XA + 8, W); A[r].a3 = ref_dword(row, XA + 12, W);
		}
	}
}
// ... and the byte alignment: afterwards byte c of row r is sample (X + c, Y + r)
template <int NR>
E264_DEV void align_window(Row4 *A, int X)
{
	const uint32_t o = (uint32_t)X & 3u;
#pragma unroll
	for (int r = 0; r < NR; r++) {
		const uint32_t w0 = A[r].a0, w1 = A[r].a1, w2 = A[r].a2, w3 = A[r].a3;
		A[r].a0 = v_alignbyte(w1, w0, o); A[r].a1 = v_alignbyte(w2, w1, o); A[r].a2 = v_alignbyte(w3, w2, o); A[r].a3 = w3 >> (8 * o);
	}
}
// the 12 sliding pairs Q_c = (B_c, B_c+1), c = 0..11, of a row
E264_DEV void pairs12(const Row4 &a, s16x2 Q[12])
{
	Q[0] = pair_at<0>(a.a1, a.a0); Q[1] = pair_at<1>(a.a1, a.a0); Q[2] = pair_at<2>(a.a1, a.a0); Q[3] = pair_at<3>(a.a1, a.a0);
	Q[4] = pair_at<0>(a.a2, a.a1); Q[5] = pair_at<1>(a.a2, a.a1); Q[6] = pair_at<2>(a.a2, a.a1); Q[7] = pair_at<3>(a.a2, a.a1);
	Q[8] = pair_at<0>(a.a3, a.a2); Q[9] = pair_at<1>(a.a3, a.a2); Q[10] = pair_at<2>(a.a3, a.a2); Q[11] = pair_at<3>(a.a3, a.a2);
}
// horizontal 6-tap sums of the 8 outputs of a row (unclipped, 16 bits): T[p] = outputs (2p, 2p+1)
E264_DEV void htaps8(const Row4 &a, s16x2 T[4])
{
	s16x2 Q[12];
	pairs12(a, Q);
#pragma unroll
	for (int p = 0; p < 4; p++)
		T[p] = tap6u(Q[2 * p], Q[2 * p + 1], Q[2 * p + 2], Q[2 * p + 3], Q[2 * p + 4], Q[2 * p + 5]);
}
// two pairs of values 0..255 -> 4 bytes
E264_DEV uint32_t pack4(s16x2 lo, s16x2 hi) { return v_perm(as_u(hi), as_u(lo), 0x06040200u); }
#define ONES8 0x01010101u

// Where the rows of a partition go: the tile in LDS, combined with what list 0 left there (edge264_inter.c:1099-1106:
// bi-prediction is two calls, the second one blends):
//   mode 0 plain     (anything unweighted; list 1 alone)                  tile = p
//   mode 1 average   (no weighting, list 1 on top of list 0)              tile = (q + p + 1) >> 1
//   mode 2 weighted                                                       tile = wpred(q, p)
struct Wod { int w0, w1, o, wd; };
// maddshrL, edge264_inter.c:17-21, on two samples at a time exactly as the reference's vector code runs it: pmaddubsw
// (int8 weights, the two products added with signed saturation), adds16 (offset, saturating), sra, packus.  Packed
// 16-bit lanes with the saturating adds spelled out (v_pk_add_i16 clamp).
E264_DEV s16x2 wpred2(s16x2 q, s16x2 p, const Wod &w)
{
	const short w0 = (short)(int8_t)w.w0, w1 = (short)(int8_t)w.w1, o = (short)w.o, wd = (short)w.wd;
	const s16x2 vw0 = {w0, w0}, vw1 = {w1, w1}, vo = {o, o}, vwd = {wd, wd};
	s16x2 x = __builtin_elementwise_add_sat(q * vw0, p * vw1); // |255 * 128| fits 16 bits: the products are exact
	x = __builtin_elementwise_add_sat(x, vo);
	return x >> vwd; // the caller packs with unsigned saturation (packus)
}
E264_DEV uint32_t wpred4(uint32_t q, uint32_t p, const Wod &w)
{
	// (a scalar per-byte version of this -- sat16 / clip255 on ints, four times per dword -- came out of hipcc 7.2 with 255 in
	// the two upper bytes on the device although the host build of the same source was right; the packed form is also half
	// the instructions)
	return packus4(wpred2(pair_at<0>(0, q), pair_at<0>(0, p), w), wpred2(pair_at<2>(0, q), pair_at<2>(0, p), w));
}
struct LumaSink {
	uint32_t *ty;  // first dword of the partition in the tile
	int mode, nr;  // rows of the partition (4 or 8)
	bool w8;       // 8 samples wide (else 4)
	Wod w;
};
E264_DEV void sink_row(const LumaSink &k, int j, uint32_t v0, uint32_t v1)
{
	if (j >= 4 && j >= k.nr) // (a partition has 4 or 8 rows: no test at all for the first four)
		return;
	uint32_t *d = k.ty + j * PT_W * 4;
	if (k.mode) { // one test where no lane of the wave combines with list 0 (every P picture without explicit weights)
		if (k.mode == 1) { v0 = v_lerp_u8(d[0], v0, ONES8); v1 = v_lerp_u8(d[1], v1, ONES8); }
		else { v0 = wpred4(d[0], v0, k.w); v1 = wpred4(d[1], v1, k.w); }
	}
	d[0] = v0;
	if (k.w8) d[1] = v1;
}

// Rounding tails: (v + 16) >> 5 and the centre's (x3 + 32) >> 6, still 16 bits wide; packus4 clips and packs them.
E264_DEV s16x2 rs5(s16x2 v) { const s16x2 r = {16, 16}, s = {5, 5}; return (v + r) >> s; }
E264_DEV s16x2 centre6r(s16x2 t0, s16x2 t1, s16x2 t2, s16x2 t3, s16x2 t4, s16x2 t5)
{ // sixtapHV + (x + 32) >> 6: 16-bit lanes wrap exactly like the reference's int16 vectors (edge264_inter.c:4-9,14)
	const s16x2 s2 = {2, 2}, s6 = {6, 6}, r32 = {32, 32};
	const s16x2 af = t0 + t5, be = t1 + t4, cd = t2 + t3;
	const s16x2 x1 = af - be;
	const s16x2 x2 = (x1 >> s2) + (cd - be);
	const s16x2 x3 = (x2 >> s2) + cd;
	return (x3 + r32) >> s6;
}
// columns 2 + dx .. 9 + dx of an aligned row, as bytes
E264_DEV void gcols(const Row4 &a, uint32_t dx, uint32_t g[2])
{
	const uint32_t s0 = v_alignbyte(a.a1, a.a0, dx), s1 = v_alignbyte(a.a2, a.a1, dx), s2 = v_alignbyte(a.a3, a.a2, dx);
	g[0] = v_alignbyte(s1, s0, 2); g[1] = v_alignbyte(s2, s1, 2);
}
// b of one aligned row: 8 horizontal half samples, clipped, as bytes
E264_DEV void brow(const Row4 &a, uint32_t b[2])
{
	const s16x2 s5 = {5, 5};
	s16x2 Q[12], T[4];
	pairs12(a, Q);
#pragma unroll
	for (int p = 0; p < 4; p++)
		T[p] = tap6u16(Q[2 * p], Q[2 * p + 1], Q[2 * p + 2], Q[2 * p + 3], Q[2 * p + 4], Q[2 * p + 5]) >> s5;
	b[0] = packus4(T[0], T[1]); b[1] = packus4(T[2], T[3]);
}
// h of one output row from six rows of byte columns
E264_DEV void hrow(const uint32_t g[][2], int j, uint32_t h[2])
{
	const s16x2 s5 = {5, 5};
	s16x2 V[4];
#pragma unroll
	for (int p = 0; p < 4; p++) {
		const int w = p >> 1;
		s16x2 c[6];
#pragma unroll
		for (int k = 0; k < 6; k++)
			c[k] = (p & 1) ? pair_at<2>(0, g[j + k][w]) : pair_at<0>(0, g[j + k][w]);
		V[p] = tap6u16(c[0], c[1], c[2], c[3], c[4], c[5]) >> s5;
	}
	h[0] = packus4(V[0], V[1]); h[1] = packus4(V[2], V[3]);
}

// Round 6 (-DE264_PRED_HROW_ONCE): the byte columns of a row spread into 16-bit pairs ONCE per input row (4 permutes) instead of once per output row that
// taps it (hrow: 6 rows x 4 permutes per output row -- 192 against 52 per 8 x 8 block)
#ifndef E264_PRED_HROW_ONCE
#define E264_PRED_HROW_ONCE 0 // measured (profiles/r06_ablations.txt item 9): 140 permutes fewer per item of classes 2 / 3, and no change in time
#endif
E264_DEV void urow(const uint32_t g[2], s16x2 U[4])
{
	U[0] = pair_at<0>(0, g[0]); U[1] = pair_at<2>(0, g[0]); U[2] = pair_at<0>(0, g[1]); U[3] = pair_at<2>(0, g[1]);
}
E264_DEV void hrow_u(const s16x2 U[][4], int j, uint32_t h[2])
{
	const s16x2 s5 = {5, 5};
	s16x2 V[4];
#pragma unroll
	for (int p = 0; p < 4; p++)
		V[p] = tap6u16(U[j][p], U[j + 1][p], U[j + 2][p], U[j + 3][p], U[j + 4][p], U[j + 5][p]) >> s5;
	h[0] = packus4(V[0], V[1]); h[1] = packus4(V[2], V[3]);
}

// class 0: integer position.  A holds window rows 2..9 only (8 rows), aligned.
E264_DEV void luma_g(const Row4 A[8], const LumaSink &sink)
{
#pragma unroll
	for (int j = 0; j < 8; j++)
		sink_row(sink, j, v_alignbyte(A[j].a1, A[j].a0, 2), v_alignbyte(A[j].a2, A[j].a1, 2));
}
// class 1: yFrac == 0 (edge264_inter.c:439-470): b of row 2; xFrac odd: averaged with G of column 2 + (xF == 3).  A: rows 2..9.
E264_DEV void luma_b(const Row4 A[8], int xF, const LumaSink &sink)
{
	const uint32_t dx = xF == 3;
	const bool q = xF != 2;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		uint32_t b[2], g[2];
		brow(A[j], b);
		gcols(A[j], dx, g);
		sink_row(sink, j, v_lerp_u8(b[0], q ? g[0] : b[0], ONES8), v_lerp_u8(b[1], q ? g[1] : b[1], ONES8));
	}
}
// class 2: xFrac == 0 (edge264_inter.c:472-508): h of column 2; yFrac odd: averaged with G of row 2 + (yF == 3)
E264_DEV void luma_h(const Row4 A[13], int yF, const LumaSink &sink)
{
	const bool dy = yF == 3, q = yF != 2;
	uint32_t g[13][2];
#if E264_PRED_HROW_ONCE
	s16x2 U[13][4];
#endif
#pragma unroll
	for (int r = 0; r < 13; r++) {
		g[r][0] = v_alignbyte(A[r].a1, A[r].a0, 2); g[r][1] = v_alignbyte(A[r].a2, A[r].a1, 2);
#if E264_PRED_HROW_ONCE
		urow(g[r], U[r]);
#endif
		if (r < 5)
			continue;
		const int j = r - 5;
		uint32_t h[2];
#if E264_PRED_HROW_ONCE
		hrow_u(U, j, h);
#else
		hrow(g, j, h);
#endif
		const uint32_t G0 = dy ? g[j + 3][0] : g[j + 2][0], G1 = dy ? g[j + 3][1] : g[j + 2][1];
		sink_row(sink, j, v_lerp_u8(h[0], q ? G0 : h[0], ONES8), v_lerp_u8(h[1], q ? G1 : h[1], ONES8));
	}
}
// class 3: both fractions odd (edge264_inter.c:510-557): b of row 2 + (yF == 3) averaged with h of column 2 + (xF == 3)
E264_DEV void luma_d(const Row4 A[13], int xF, int yF, const LumaSink &sink)
{
	const uint32_t dx = xF == 3;
	const bool dy = yF == 3;
	uint32_t g[13][2];
#if E264_PRED_HROW_ONCE
	s16x2 U[13][4];
#endif
#pragma unroll
	for (int r = 0; r < 13; r++) {
		gcols(A[r], dx, g[r]);
#if E264_PRED_HROW_ONCE
		urow(g[r], U[r]);
#endif
		if (r < 5)
			continue;
		const int j = r - 5;
		Row4 s;
		s.a0 = dy ? A[j + 3].a0 : A[j + 2].a0; s.a1 = dy ? A[j + 3].a1 : A[j + 2].a1;
		s.a2 = dy ? A[j + 3].a2 : A[j + 2].a2; s.a3 = dy ? A[j + 3].a3 : A[j + 2].a3;
		uint32_t b[2], h[2];
		brow(s, b);
#if E264_PRED_HROW_ONCE
		hrow_u(U, j, h);
#else
		hrow(g, j, h);
#endif
		sink_row(sink, j, v_lerp_u8(b[0], h[0], ONES8), v_lerp_u8(b[1], h[1], ONES8));
	}
}

// Classes 0 .. 3 as ONE flow (round 4).  A tile's items are few per class (a 16 x 4 tile of a P picture: ~20 / 65 / 65 / 90 of the
// four), every class list padded to whole waves left most waves half empty: 7 class-waves where the items fill 4.  The four flows
// nest -- G is a byte shuffle, class 1 is b (+ G), class 2 is h (+ G), class 3 is b + h -- so here they are one list, sorted by
// class, and a wave computes the PARTS its lanes need (wave-uniform branches) and every lane selects: a wave that straddles two
// classes costs what its dearer class costs, not the sum.  A: the 13-row window, aligned (rows the lane did not fetch are
// never selected).
//   out = lerp(P, Q):  class 0: G, G   class 1: b, (xF == 2 ? b : G(dx = xF == 3))   class 2: h, (yF == 2 ? h : G(dy = yF == 3))
//                      class 3: b of row 2 + (yF == 3), h of column 2 + (xF == 3)
#ifdef E264_HOST_INTRINSICS
#define PRED_ANY(x) true // (host build: every part computed; the selects decide)
#else
#define PRED_ANY(x) __any(x)
#endif
E264_DEV void luma_1d(const Row4 A[13], int xF, int yF, const LumaSink &sink)
{
	const bool use_b = xF != 0, use_h = yF != 0;
	const bool any_b = PRED_ANY(use_b), any_h = PRED_ANY(use_h), any_g = PRED_ANY(!(use_b && use_h));
	const uint32_t dxh = use_b && xF == 3;          // class 3: h of column 2 + dx (class 2: column 2)
	const bool dyb = use_h && yF == 3;              // class 3: b of row 2 + dy (class 1: row 2)
	const uint32_t dxg = !use_h && xF == 3;         // class 1: G of column 2 + dx
	const bool dyg = !use_b && yF == 3;             // class 2: G of row 2 + dy
	const bool q_is_p = xF == 2 || yF == 2;         // no quarter-sample average: the half sample itself
	uint32_t g[13][2];
#pragma unroll
	for (int r = 0; r < 13; r++) {
		if (any_h) gcols(A[r], dxh, g[r]);
		else { g[r][0] = any_u32(); g[r][1] = any_u32(); }
		if (r < 5)
			continue;
		const int j = r - 5;
		uint32_t b[2] = {any_u32(), any_u32()}, h[2] = {any_u32(), any_u32()}, G[2] = {any_u32(), any_u32()};
		if (any_b) {
			Row4 s;
			s.a0 = dyb ? A[j + 3].a0 : A[j + 2].a0; s.a1 = dyb ? A[j + 3].a1 : A[j + 2].a1;
			s.a2 = dyb ? A[j + 3].a2 : A[j + 2].a2; s.a3 = dyb ? A[j + 3].a3 : A[j + 2].a3;
			brow(s, b);
		}
		if (any_h) hrow(g, j, h);
		if (any_g) {
			Row4 s;
			s.a0 = dyg ? A[j + 3].a0 : A[j + 2].a0; s.a1 = dyg ? A[j + 3].a1 : A[j + 2].a1;
			s.a2 = dyg ? A[j + 3].a2 : A[j + 2].a2; s.a3 = dyg ? A[j + 3].a3 : A[j + 2].a3;
			gcols(s, dxg, G);
		}
		const uint32_t P0 = use_b ? b[0] : use_h ? h[0] : G[0], P1 = use_b ? b[1] : use_h ? h[1] : G[1];
		const uint32_t Q0 = (use_b && use_h) ? h[0] : q_is_p ? P0 : G[0], Q1 = (use_b && use_h) ? h[1] : q_is_p ? P1 : G[1];
		sink_row(sink, j, v_lerp_u8(P0, Q0, ONES8), v_lerp_u8(P1, Q1, ONES8));
	}
}

// class 4: xFrac == 2, yFrac != 0 (edge264_inter.c:611-646, 779-802, 929-966): horizontal taps of 13 rows, then sixtapHV down
// the columns; yFrac odd: averaged with b of row 2 + (yF == 3)
E264_DEV void luma_2dh(const Row4 A[13], int yF, const LumaSink &sink)
{
	const bool dy = yF == 3, q = yF != 2;
	s16x2 T[13][4];
#pragma unroll
	for (int r = 0; r < 13; r++) {
		htaps8(A[r], T[r]);
		if (r < 5)
			continue;
		const int j = r - 5;
		s16x2 J[4], B[4];
#pragma unroll
		for (int p = 0; p < 4; p++) {
			J[p] = centre6r(T[j][p], T[j + 1][p], T[j + 2][p], T[j + 3][p], T[j + 4][p], T[j + 5][p]);
			B[p] = rs5(dy ? T[j + 3][p] : T[j + 2][p]);
		}
		const uint32_t j0 = packus4(J[0], J[1]), j1 = packus4(J[2], J[3]);
		const uint32_t b0 = packus4(B[0], B[1]), b1 = packus4(B[2], B[3]);
		sink_row(sink, j, v_lerp_u8(j0, q ? b0 : j0, ONES8), v_lerp_u8(j1, q ? b1 : j1, ONES8));
	}
}

// class 5: yFrac == 2, xFrac odd (edge264_inter.c:559-609, 741-777, 887-927): vertical taps of 13 columns, then sixtapHV along
// the rows; averaged with h of column 2 + (xF == 3)
E264_DEV void luma_2dv(const Row4 A[13], int xF, const LumaSink &sink)
{
	const bool dx = xF == 3;
	// column pairs E_k = (B_2k, B_2k+1), k = 0..6 (the second half of E_6 is not a window sample and is never used)
	s16x2 E[13][7];
#pragma unroll
	for (int r = 0; r < 13; r++) {
		E[r][0] = pair_at<0>(A[r].a1, A[r].a0); E[r][1] = pair_at<2>(A[r].a1, A[r].a0);
		E[r][2] = pair_at<0>(A[r].a2, A[r].a1); E[r][3] = pair_at<2>(A[r].a2, A[r].a1);
		E[r][4] = pair_at<0>(A[r].a3, A[r].a2); E[r][5] = pair_at<2>(A[r].a3, A[r].a2);
		E[r][6] = pair_at<0>(0, A[r].a3 & 255u);
		if (r < 5)
			continue;
		const int j = r - 5;
		s16x2 V[7], O[6];
#pragma unroll
		for (int k = 0; k < 7; k++)
			V[k] = tap6u(E[j][k], E[j + 1][k], E[j + 2][k], E[j + 3][k], E[j + 4][k], E[j + 5][k]);
#pragma unroll
		for (int k = 0; k < 6; k++) // (V_2k+1, V_2k+2)
			O[k] = as_s2(v_alignbit(as_u(V[k + 1]), as_u(V[k]), 16));
		s16x2 J[4], Hh[4];
#pragma unroll
		for (int p = 0; p < 4; p++) {
			J[p] = centre6r(V[p], O[p], V[p + 1], O[p + 1], V[p + 2], O[p + 2]);
			Hh[p] = rs5(dx ? O[p + 1] : V[p + 1]);
		}
		sink_row(sink, j, v_lerp_u8(packus4(J[0], J[1]), packus4(Hh[0], Hh[1]), ONES8), v_lerp_u8(packus4(J[2], J[3]), packus4(Hh[2], Hh[3]), ONES8));
	}
}

// bilinear chroma of a 4x4 block of one plane (edge264_inter.c:977-1091; ABCD of :1242 factored into a horizontal and a
// vertical blend, identical integers): window rows YC..YC+4, columns XC..XC+4.  Loads and arithmetic are separate so that
// the caller can have every load of an item in flight before the first use.
E264_DEV void chroma_load(const gu8 *plane, int sC, int Wc, int Hc, int XC, int YC, bool all, bool frac, uint32_t w[5][2])
{ // all: 4 output rows (else 2); frac: the vertical fraction is not zero, one more row below is blended in.  Rows nobody needs are not fetched
	const int XA = XC & ~3;
	const bool need[5] = {true, true, all || frac, all, all && frac};
#pragma unroll
	for (int r = 2; r < 5; r++) { w[r][0] = any_u32(); w[r][1] = any_u32(); }
	if (XA >= 0 && XA + 4 <= Wc - 4) {
#pragma unroll
		for (int r = 0; r < 5; r++) {
			if (!need[r])
				continue;
			const v2u v = *(const gv2u *)(plane + (uint32_t)mul24(min(max(YC + r, 0), Hc - 1), sC) + XA);
			w[r][0] = v.x; w[r][1] = v.y;
		}
	} else {
#pragma unroll
		for (int r = 0; r < 5; r++) {
			if (!need[r])
				continue;
			const gu8 *row = plane + (uint32_t)mul24(min(max(YC + r, 0), Hc - 1), sC);
			w[r][0] = ref_dword(row, XA, Wc); w[r][1] = ref_dword(row, XA + 4, Wc);
		}
	}
}
// The same rows 12 bytes wide: the 8 x 4 chroma block of ONE plane that two horizontally adjacent quadrants with the same motion
// share (16x16 and 16x8 partitions).  Each of the two lanes takes one plane: 5 lane-rows instead of 10 (the fetch is what this
// kernel waits for: one lane-row costs the vector memory path 1.5 - 2.5 cycles whatever its width).
E264_DEV void chroma_load12(const gu8 *plane, int sC, int Wc, int Hc, int XC, int YC, bool frac, uint32_t w[5][3])
{
	const int XA = XC & ~3;
	if (XA >= 0 && XA + 8 <= Wc - 4) {
#pragma unroll
		for (int r = 0; r < 5; r++) {
			if (r == 4 && !frac) { w[r][0] = any_u32(); w[r][1] = any_u32(); w[r][2] = any_u32(); continue; }
			const v3u v = *(const gv3u *)(plane + (uint32_t)mul24(min(max(YC + r, 0), Hc - 1), sC) + XA);
			w[r][0] = v.x; w[r][1] = v.y; w[r][2] = v.z;
		}
	} else {
#pragma unroll
		for (int r = 0; r < 5; r++) {
			if (r == 4 && !frac) { w[r][0] = any_u32(); w[r][1] = any_u32(); w[r][2] = any_u32(); continue; }
			const gu8 *row = plane + (uint32_t)mul24(min(max(YC + r, 0), Hc - 1), sC);
			w[r][0] = ref_dword(row, XA, Wc); w[r][1] = ref_dword(row, XA + 4, Wc); w[r][2] = ref_dword(row, XA + 8, Wc);
		}
	}
}
E264_DEV void chroma4x4(const uint32_t w[5][2], int XC, int xF, int yF, uint32_t out[4])
{
	const uint32_t o = (uint32_t)XC & 3u;
	const s16x2 cx0 = {(short)(8 - xF), (short)(8 - xF)}, cx1 = {(short)xF, (short)xF};
	const s16x2 cy0 = {(short)(8 - yF), (short)(8 - yF)}, cy1 = {(short)yF, (short)yF};
	s16x2 hr[5][2];
#pragma unroll
	for (int r = 0; r < 5; r++) {
		const uint32_t a0 = v_alignbyte(w[r][1], w[r][0], o), a1 = w[r][1] >> (8 * o);
		hr[r][0] = pair_at<0>(a1, a0) * cx0 + pair_at<1>(a1, a0) * cx1;
		hr[r][1] = pair_at<2>(a1, a0) * cx0 + pair_at<3>(a1, a0) * cx1;
	}
	const s16x2 r32 = {32, 32}, s6 = {6, 6};
#pragma unroll
	for (int j = 0; j < 4; j++)
		out[j] = pack4((hr[j][0] * cy0 + hr[j + 1][0] * cy1 + r32) >> s6, (hr[j][1] * cy0 + hr[j + 1][1] * cy1 + r32) >> s6);
}

// decode_inter weight selection, edge264_inter.c:1137-1197 (all schemes except "no weights" and the default average,
// which the caller handles on packed bytes)
E264_DEV void pred_weights(cslice_t s, int list, int refIdx, int refIdxX, Wod &wY, Wod &wCb, Wod &wCr)
{
	Wod nw = {0, 1, 0, 0};
	wY = wCb = wCr = nw;
	const int idc = s->weighted_bipred_idc;
	if (idc != 1) {
		if (list == 1 && refIdxX >= 0) {
			if (idc == 0) {
				Wod d = {1, 1, 1, 1};
				wY = wCb = wCr = d;
			} else {
				int w1 = (int)s->implicit_weights[refIdxX][refIdx] - 64;
				Wod d = {64 - w1, w1, 32, 6};
				if ((unsigned)(w1 + 63) >= 191u) { d.w0 = 2 - (w1 >> 5); d.w1 = w1 >> 5; d.o = 1; d.wd = 1; }
				wY = wCb = wCr = d;
			}
		}
	} else if (refIdxX < 0) {
		const int i = refIdx + list * 32;
		const int lwd = s->luma_log2_weight_denom, cwd = s->chroma_log2_weight_denom;
		if (s->explicit_weights[0][i] < 128) {
			wY.w1 = s->explicit_weights[0][i];
			wY.o = w16(((s->explicit_offsets[0][i] * 2 + 1) << lwd) >> 1);
			wY.wd = lwd;
		}
		if (s->explicit_weights[1][i] < 128) {
			wCb.w1 = s->explicit_weights[1][i];
			wCr.w1 = s->explicit_weights[2][i];
			wCb.o = w16(((s->explicit_offsets[1][i] * 2 + 1) << cwd) >> 1);
			wCr.o = w16(((s->explicit_offsets[2][i] * 2 + 1) << cwd) >> 1);
			wCb.wd = wCr.wd = cwd;
		}
	} else if (list == 1) {
		const int i = refIdx + 32, x = refIdxX;
		const int lwd = s->luma_log2_weight_denom, cwd = s->chroma_log2_weight_denom;
		const int a = s->explicit_weights[0][x], b = s->explicit_weights[0][i];
		const int oo = ((s->explicit_offsets[0][x] + s->explicit_offsets[0][i] + 1) | 1) << lwd;
		if ((a & b) != 128) { wY.w0 = a; wY.w1 = b; wY.o = w16(oo); wY.wd = lwd + 1; }
		else { wY.w0 = a >> 1; wY.w1 = b >> 1; wY.o = w16(oo >> 1); wY.wd = lwd; }
		const int a1 = s->explicit_weights[1][x], b1 = s->explicit_weights[1][i];
		const int a2 = s->explicit_weights[2][x], b2 = s->explicit_weights[2][i];
		const int o1 = ((s->explicit_offsets[1][x] + s->explicit_offsets[1][i] + 1) | 1) << cwd;
		const int o2 = ((s->explicit_offsets[2][x] + s->explicit_offsets[2][i] + 1) | 1) << cwd;
		if ((a1 & b1) != 128) {
			wCb.w0 = a1; wCb.w1 = b1; wCr.w0 = a2; wCr.w1 = b2;
			wCb.o = w16(o1); wCr.o = w16(o2); wCb.wd = wCr.wd = cwd + 1;
		} else {
			wCb.w0 = a1 >> 1; wCb.w1 = b1 >> 1; wCr.w0 = a2 >> 1; wCr.w1 = b2 >> 1;
			wCb.o = w16(o1 >> 1); wCr.o = w16(o2 >> 1); wCb.wd = wCr.wd = cwd;
		}
	}
}

// One prediction item of list `list`, class `cls`: luma + chroma of the partition -> the tile in LDS
#ifndef E264_PRED_PAIR_TILES
#define E264_PRED_PAIR_TILES 0 // 0: never pair (the default: measured, profiles/r06_ablations.txt item 6); 1: per tile, where three quarters of its inter macroblocks are one partition per list; 2: always (the -DE264_PRED_HPAIR build of round 4)
#endif
E264_DEV bool pred_pair_chroma(const PredLds &L)
{
	return E264_PRED_PAIR_TILES == 2 || (E264_PRED_PAIR_TILES == 1 && L.n_uni * 4 >= L.n_inter * 3);
}
E264_DEV void pred_item(PredLds &L, const FrameCtx &f, const PredTile &t, int list, int cls, int desc, const bool pair_tile)
{
	const int slot = PI_SLOT(desc), sub = PI_SUB(desc), shape = PI_SHAPE(desc);
	const int mb = slot >> 2, q = slot & 3;
	const bool w8 = shape == 0 || shape == 1, h8 = shape == 0 || shape == 2; // width / height 8 (else 4)
	const int px0 = (mb & (PT_W - 1)) * 16 + (q & 1) * 8 + (sub & 1) * 4, py0 = (mb / PT_W) * 16 + (q >> 1) * 8 + (sub >> 1) * 4; // in the tile
	const uint32_t mv = L.mvq[sub][slot], rq = L.refq[slot];
	const int mx = (int)(int16_t)(mv & 0xffff), my = (int)mv >> 16;
	const int refIdx = (int)(int8_t)(rq >> 8), refIdxX = (int)(int8_t)(rq >> 16), idc = (int)(int8_t)(rq >> 24);
	const gu8 *ref = (const gu8 *)L.dpb[rq & 255];
	const int gx = t.tx0 * 16 + px0, gy = t.ty0 * 16 + py0;
	// every load of the item goes out first: 13 luma rows, 5 + 5 chroma rows
	const int X = gx + (mx >> 2) - 2, Y = gy + (my >> 2) - 2;
	const int XC = (gx >> 1) + (mx >> 3), YC = (gy >> 1) + (my >> 3);
	Row4 A[13];
	uint32_t cw[2][5][2];
	// -DE264_PRED_HPAIR (measured, NOT the default: profiles/r04_ablations.txt item 1): chroma by PLANE where the quadrant beside this
	// one (q ^ 1: same macroblock, same rows) has the same motion -- 16x16 and 16x8 partitions, /root/reference/src/edge264_inter.c:977-1091
	// predicts the chroma of a partition in one call --: this lane predicts the 8 x 4 block of plane q & 1 for both, its neighbour the
	// other plane's (5 lane-rows of 12 bytes instead of 10 of 8).  Both lanes decide alike (the test is symmetric).  Bit-exact, 18 % fewer
	// lane-rows on the bench GOP -- and 11 % SLOWER (1.266 -> 1.407 ms): a wave of one class holds paired and unpaired items side by side
	// and runs both chroma paths; the kernel's time follows its VALU count, not its lane-rows.
	// Round 6 (-DE264_PRED_PAIR_TILES=1, measured, NOT the default: profiles/r06_ablations.txt item 6): a choice PER TILE (pair_tile, uniform over the
	// workgroup): on, where at least three quarters of the tile's inter macroblocks are one partition per list -- 94 % of the inter macroblocks of
	// encoder-made content, every wave then runs the paired path alone; off on tiles of mixed partitions (the synthetic bench GOP: 55 %).  22 % fewer
	// lane-rows on nat1080_ipp30 -- and 5 % SLOWER there (0.899 -> 0.948 ms), 2 % slower on the bench GOP where no tile pairs: the 12-byte rows cost more
	// VALU instructions than the 8-byte ones, and this kernel's time is its instruction count.
	const bool hp = pair_tile && shape == 0 && L.refq[slot ^ 1] == rq && L.mvq[0][slot ^ 1] == mv && L.mvq[1][slot ^ 1] == mv && L.mvq[2][slot ^ 1] == mv && L.mvq[3][slot ^ 1] == mv;
	const int XCp = ((gx & ~15) >> 1) + (mx >> 3); // the partition's first chroma column
	uint32_t (*cw3)[3] = (uint32_t (*)[3])&cw[0][0][0]; // the same registers as rows of three dwords
	const gu8 *cplane = ref + f.psY + ((q & 1) ? (f.sC >> 1) : 0);
#ifdef E264_ABL_NOLOAD // timing ablation: no reference fetch at all (results are wrong on purpose)
	for (int r = 0; r < 13; r++) { A[r].a0 = X + r; A[r].a1 = Y; A[r].a2 = X * r; A[r].a3 = Y - r; }
	for (int r = 0; r < 5; r++) { cw[0][r][0] = XC + r; cw[0][r][1] = YC; cw[1][r][0] = XC; cw[1][r][1] = YC * r; }
#else
#ifndef E264_PRED_CHROMA_LAST // the chroma rows are consumed first: requested first, their arithmetic runs while the luma rows arrive (1.266 -> 1.249 ms, profiles/r04_ablations.txt item 1)
	if (hp) chroma_load12(cplane, f.sC, f.W >> 1, f.H >> 1, XCp, YC, (my & 7) != 0, cw3);
	else {
		chroma_load(ref + f.psY, f.sC, f.W >> 1, f.H >> 1, XC, YC, h8, (my & 7) != 0, cw[0]);
		chroma_load(ref + f.psY + (f.sC >> 1), f.sC, f.W >> 1, f.H >> 1, XC, YC, h8, (my & 7) != 0, cw[1]);
	}
#endif
#ifdef E264_PRED_MERGE1D
	if (cls <= 3) load_window_rows(ref, f.sY, f.W, f.H, X, Y, cls <= 1 ? 2 : 0, (cls <= 1 ? 10 : 13) - (h8 ? 0 : 4), A); // (per lane: G and b only look at rows 2..9)
	else load_window<13>(ref, f.sY, f.W, f.H, X, Y, h8, A);
#else
	if (cls <= 1) load_window<8>(ref, f.sY, f.W, f.H, X, Y + 2, h8, A); // G and b only look at rows 2..9
	else load_window<13>(ref, f.sY, f.W, f.H, X, Y, h8, A);
#endif
#ifdef E264_PRED_CHROMA_LAST
	if (hp) chroma_load12(cplane, f.sC, f.W >> 1, f.H >> 1, XCp, YC, (my & 7) != 0, cw3);
	else {
		chroma_load(ref + f.psY, f.sC, f.W >> 1, f.H >> 1, XC, YC, h8, (my & 7) != 0, cw[0]);
		chroma_load(ref + f.psY + (f.sC >> 1), f.sC, f.W >> 1, f.H >> 1, XC, YC, h8, (my & 7) != 0, cw[1]);
	}
#endif
#endif
	// combination with list 0 / weights
	const bool second = list == 1 && refIdxX >= 0;
	int mode = second ? 1 : 0;
	Wod wY = {0, 1, 0, 0}, wCb = wY, wCr = wY;
	if (idc != 0) {
		mode = (idc == 1 ? (refIdxX < 0 || list == 1) : second) ? 2 : 0;
		if (mode == 2)
			pred_weights(f.slices + (L.hdr[mb][2] >> 16), list, refIdx, refIdxX, wY, wCb, wCr);
	}
	if (hp) { // the 8 x 4 block of plane q & 1: two 4 x 4 blocks out of the same rows
		uint32_t oc[2][4], wl[5][2], wr[5][2];
#pragma unroll
		for (int r = 0; r < 5; r++) { wl[r][0] = cw3[r][0]; wl[r][1] = cw3[r][1]; wr[r][0] = cw3[r][1]; wr[r][1] = cw3[r][2]; }
		chroma4x4(wl, XCp, mx & 7, my & 7, oc[0]);
		chroma4x4(wr, XCp, mx & 7, my & 7, oc[1]);
		const Wod w = (q & 1) ? wCr : wCb;
		uint32_t *tc = &L.c[q & 1][py0 >> 1][(px0 >> 4) * 2];
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int hx = 0; hx < 2; hx++) {
				uint32_t v = oc[hx][j];
				if (mode != 0) {
					const uint32_t qv = tc[j * PT_W * 2 + hx];
					v = mode == 1 ? v_lerp_u8(qv, v, ONES8) : wpred4(qv, v, w);
				}
				tc[j * PT_W * 2 + hx] = v;
			}
	} else { // chroma: small, and its rows leave at once
		uint32_t oc[2][4];
		chroma4x4(cw[0], XC, mx & 7, my & 7, oc[0]);
		chroma4x4(cw[1], XC, mx & 7, my & 7, oc[1]);
		// Every test once per item, not once per row and plane (round 4: each `if` of a lane-dependent condition is three scalar
		// instructions of execution-mask bookkeeping, and there were twenty-four of them here): the combination with list 0, then the
		// stores by width, the lower two rows of a partition 8 high under one test.
		uint32_t *tc0 = &L.c[0][py0 >> 1][px0 >> 3];
		uint16_t *hc0 = (uint16_t *)&L.c[0][py0 >> 1][0] + (px0 >> 2);
		constexpr int PL = (int)(sizeof(L.c[0]) / 4); // dwords per chroma plane of the tile
		if (mode != 0) {
#pragma unroll
			for (int pc = 0; pc < 2; pc++) {
				const Wod &w = pc ? wCr : wCb;
#pragma unroll
				for (int j = 0; j < 4; j++) { // (rows 2, 3 of a partition 4 high: read and combined for nothing, never stored)
					const uint32_t qv = w8 ? tc0[pc * PL + j * PT_W * 2] : hc0[pc * PL * 2 + j * PT_W * 4];
					oc[pc][j] = mode == 1 ? v_lerp_u8(qv, oc[pc][j], ONES8) : wpred4(qv, oc[pc][j], w);
				}
			}
		}
		if (w8) {
#pragma unroll
			for (int pc = 0; pc < 2; pc++) { tc0[pc * PL] = oc[pc][0]; tc0[pc * PL + PT_W * 2] = oc[pc][1]; }
			if (h8) {
#pragma unroll
				for (int pc = 0; pc < 2; pc++) { tc0[pc * PL + 2 * PT_W * 2] = oc[pc][2]; tc0[pc * PL + 3 * PT_W * 2] = oc[pc][3]; }
			}
		} else {
#pragma unroll
			for (int pc = 0; pc < 2; pc++) { hc0[pc * PL * 2] = (uint16_t)oc[pc][0]; hc0[pc * PL * 2 + PT_W * 4] = (uint16_t)oc[pc][1]; }
			if (h8) {
#pragma unroll
				for (int pc = 0; pc < 2; pc++) { hc0[pc * PL * 2 + 2 * PT_W * 4] = (uint16_t)oc[pc][2]; hc0[pc * PL * 2 + 3 * PT_W * 4] = (uint16_t)oc[pc][3]; }
			}
		}
	}
	LumaSink sink;
	sink.ty = &L.y[py0][px0 >> 2]; sink.mode = mode; sink.nr = h8 ? 8 : 4; sink.w8 = w8; sink.w = wY;
#ifdef E264_ABL_NOLUMA // timing ablation: the windows are fetched but only folded into one row
	{
		uint32_t x0 = 0, x1 = 0;
		for (int r = 0; r < 13; r++) { x0 ^= A[r].a0 ^ A[r].a2; x1 ^= A[r].a1 ^ A[r].a3; }
		sink_row(sink, 0, x0, x1);
		return;
	}
#endif
#ifdef E264_PRED_MERGE1D
	if (cls <= 3) { // (wave-uniform: the one-dimensional classes share their waves with nobody else)
		align_window<13>(A, X);
		luma_1d(A, mx & 3, my & 3, sink);
	} else if (cls == 4) { align_window<13>(A, X); luma_2dh(A, my & 3, sink); }
	else { align_window<13>(A, X); luma_2dv(A, mx & 3, sink); }
#else
	if (cls <= 1) {
		align_window<8>(A, X);
		if (cls == 0) luma_g(A, sink);
		else luma_b(A, mx & 3, sink);
	} else {
		align_window<13>(A, X);
		if (cls == 2) luma_h(A, my & 3, sink);
		else if (cls == 3) luma_d(A, mx & 3, my & 3, sink);
		else if (cls == 4) luma_2dh(A, my & 3, sink);
		else luma_2dv(A, mx & 3, sink);
	}
#endif
}

E264_DEV void pred_phase_items(PredLds &L, const FrameCtx &f, const PredTile &t, int list, int tid)
{
	const bool pair_tile = pred_pair_chroma(L);
	// The sorted list is walked from its END (the two-dimensional classes first): when the tile has more items than
	// threads, the extra pass that only the first waves make -- while the others wait at the barrier -- then holds the
	// cheapest items (integer and one-dimensional positions) instead of the most expensive ones.
#ifdef E264_PRED_MERGE1D
	// the two-dimensional classes (5, 4) each from a wave boundary, then classes 2, 3, 1, 0 back to back (luma_1d): dear items first
	const int c5 = L.cnt[5], c4 = L.cnt[4], c2 = L.cnt[2], c3 = L.cnt[3], c1 = L.cnt[1], c0 = L.cnt[0];
	const int k5 = (c5 + 63) & ~63, k4 = (c4 + 63) & ~63, n1d = c2 + c3 + c1 + c0;
	for (int p = tid; p < k5 + k4 + n1d; p += PT_NT) {
		int cls, idx;
		if (p < k5) { cls = 5; idx = p; if (idx >= c5) continue; }
		else if (p < k5 + k4) { cls = 4; idx = p - k5; if (idx >= c4) continue; }
		else {
			idx = p - k5 - k4; cls = 2;
			if (idx >= c2) { idx -= c2; cls = 3; if (idx >= c3) { idx -= c3; cls = 1; if (idx >= c1) { idx -= c1; cls = 0; } } }
		}
		pred_item(L, f, t, list, cls, *pred_list_slot(L, cls, idx), pair_tile);
	}
#elif !defined(E264_PRED_CLASS_PACKED)
	// Every class starts at a wave boundary: a wave runs ONE of the six flows (packed back to back, a wave of a 256-thread tile
	// straddles two classes more often than not and executes both: 418 M instead of 348 M VALU wave-instructions per launch,
	// profiles/r03_pmc_sq_instruction_mix.txt).  More partly filled waves, fewer instructions: 1.380 -> 1.312 ms.
	int np = 0;
#pragma unroll
	for (int c = 0; c < 6; c++) np += (L.cnt[c] + 63) & ~63;
	for (int p = tid; p < np; p += PT_NT) {
		int cls = 5, idx = p;
#pragma unroll
		for (int c = 5; c > 0; c--) { // expensive classes first
			const int k = (L.cnt[c] + 63) & ~63;
			if (cls == c && idx >= k) { cls = c - 1; idx -= k; }
		}
		if (idx < L.cnt[cls])
			pred_item(L, f, t, list, cls, *pred_list_slot(L, cls, idx), pair_tile);
	}
#else // -DE264_PRED_CLASS_PACKED: the lists back to back (round 2)
	int n = 0;
#pragma unroll
	for (int c = 0; c < 6; c++) n += L.cnt[c];
	for (int p = tid; p < n; p += PT_NT) {
		int cls = 0, idx = n - 1 - p;
#pragma unroll
		for (int c = 0; c < 5; c++) {
			const int k = L.cnt[c];
			if (cls == c && idx >= k) { cls = c + 1; idx -= k; }
		}
		pred_item(L, f, t, list, cls, *pred_list_slot(L, cls, idx), pair_tile);
	}
#endif
}
E264_DEV void pred_phase_reset(PredLds &L, int tid)
{
	if (tid < 6) L.cnt[tid] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// phases 5, 6: residual of the inter macroblocks, one lane per coded block
// ---------------------------------------------------------------------------------------------------------------------
// residual item: bits 0..4 block (0..15 luma 4x4 in zig order, 16..23 chroma 4x4 Cb 0..3 / Cr 4..7, 24..27 luma 8x8), 5..11 macroblock
E264_DEV void pred_phase_reslist(PredLds &L, int tid)
{ // thread = (macroblock, byte q of its 32-bit candidate mask: luma 4x4 blocks 0..7, 8..15, chroma blocks, 8x8 blocks)
	uint16_t *l4 = &L.lst[0][0], *l8 = &L.lst[2][0];
	const int mb = tid >> 2, q = tid & 3;
	const uint32_t d0 = L.hdr[mb][0], coded = L.hdr[mb][3];
	if ((d0 & 255) != E264_MB_INTER || coded == 0 || (d0 >> 8 & E264_MBF_DONE))
		return;
	const bool t8 = (d0 >> 8) & E264_MBF_T8x8;
	uint32_t bits;
	if (q < 2) bits = t8 ? 0 : coded >> (q * 8) & 255;
	else if (q == 2) bits = (coded & E264_CODED_CHROMA_DC) ? 255 : coded >> 16 & 255;
	else bits = t8 ? ((coded & 1) | (coded >> 3 & 2) | (coded >> 6 & 4) | (coded >> 9 & 8)) : 0;
	while (bits) {
		const int b = __builtin_ctz(bits);
		bits &= bits - 1;
		const uint16_t item = (uint16_t)(mb << 5 | q << 3 | b);
		if (q < 3) l4[lds_add(&L.rcnt[0], 1)] = item;
		else l8[lds_add(&L.rcnt[1], 1)] = item;
	}
}

// add 4 residuals to 4 samples: int16 wrap add then packus (edge264_residual.c:160-171), two samples per instruction
E264_DEV uint32_t add_res4p(uint32_t px, s16x2 r01, s16x2 r23)
{
	return packus4(pair_at<0>(0, px) + r01, pair_at<2>(0, px) + r23);
}
E264_DEV uint32_t add_res4(uint32_t px, int r0, int r1, int r2, int r3)
{
	const s16x2 r01 = {(short)r0, (short)r1}, r23 = {(short)r2, (short)r3};
	return add_res4p(px, r01, r23);
}

// Levels of a block as packed int16 pairs, whatever the packet stores (int16, or int8 with E264_MBF_LEV8): 8 levels ...
E264_DEV uint32_t sext2(uint32_t w, int hi) // bytes (2 hi, 2 hi + 1) of w, sign-extended to two int16
{
	const uint32_t b0 = hi ? w >> 16 : w;
	return ((uint32_t)(int)(int8_t)b0 & 0xffffu) | (uint32_t)(int)(int8_t)(b0 >> 8) << 16;
}
E264_DEV void levels8(const gu8 *co, int l8, uint32_t cw[4])
{
	if (l8) {
		const v2u v = *(const gv2u *)co;
		cw[0] = sext2(v.x, 0); cw[1] = sext2(v.x, 1); cw[2] = sext2(v.y, 0); cw[3] = sext2(v.y, 1);
	} else {
		const v4u v = *(const gv4u *)co;
		cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
	}
}
// ... and 16
E264_DEV void levels16(const gu8 *co, int l8, uint32_t cw[8])
{
	if (l8) {
		const v4u v = *(const gv4u *)co;
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int i = 0; i < 4; i++) { cw[2 * i] = sext2(w[i], 0); cw[2 * i + 1] = sext2(w[i], 1); }
	} else {
		const v4u v0 = *(const gv4u *)co, v1 = *(const gv4u *)(co + 16);
		cw[0] = v0.x; cw[1] = v0.y; cw[2] = v0.z; cw[3] = v0.w; cw[4] = v1.x; cw[5] = v1.y; cw[6] = v1.z; cw[7] = v1.w;
	}
}

// one 4x4 block (edge264_residual.c:108-187): dequantisation, both butterflies and the add in this lane's registers
E264_DEV void res_item4(PredLds &L, const FrameCtx &f, int item)
{
	const int mb = item >> 5, b = item & 31;
	const uint32_t d0 = L.hdr[mb][0], d1 = L.hdr[mb][1], coded = L.hdr[mb][3];
	const bool t8 = (d0 >> 8) & E264_MBF_T8x8;
	cslice_t s = f.slices + (L.hdr[mb][2] >> 16);
	const gu8 *pl = f.payload + L.hdr[mb][4];
	const gi16 *cdc = (const gi16 *)pl;
	if (coded & E264_CODED_CHROMA_DC) pl += 16;
	const bool chroma = b >= 16;
	const int k = b & 15;        // luma block 0..15 / chroma block 0..7
	const int pc = (k >> 2) & 1; // chroma plane
	const int l8 = (d0 >> 8 & E264_MBF_LEV8) ? 1 : 0; // one byte per level (edge264_cmd.h): block sizes halve
	const int lumab = (t8 ? __builtin_popcount(coded & 0x1111) * 128 : __builtin_popcount(coded & 0xffff) * 32) >> l8;
	const gu8 *co = chroma ? pl + lumab + ((__builtin_popcount((coded >> 16) & ((1u << k) - 1)) * 32) >> l8)
	                       : pl + ((__builtin_popcount(coded & ((1u << k) - 1)) * 32) >> l8);
	const int qP = chroma ? (pc ? (int)(d1 & 255) : (int)(d0 >> 24)) : (int)(d0 >> 16 & 255);
	const int sh = qP / 6, m = qP - sh * 6;
	const int wl = chroma ? 4 + pc : 3; // weightScale4x4 list of an inter block: Y 3, Cb 4, Cr 5
	const v4u wsv = *(const gv4u *)((const gu8 *)s + offsetof(E264SliceParams, weightScale4x4) + wl * 16);
	const uint32_t ws[4] = {wsv.x, wsv.y, wsv.z, wsv.w};
	const int na = na_byte(NA4_0, m), nb = na_byte(NA4_1, m), nc = na_byte(0x171412100e0dull, m);
	// chroma DC of this block (transform_dc2x2, edge264_residual.c:456-480)
	int dc = 0;
	const bool has_dc = chroma && (coded & E264_CODED_CHROMA_DC);
	if (has_dc) {
		const int c0 = cdc[pc], c4 = cdc[4 + pc], c2 = cdc[2 + pc], c6 = cdc[6 + pc], n = k & 3;
		const int v = n == 0 ? c0 + c4 + c2 + c6 : n == 1 ? c0 - c4 + c2 - c6 : n == 2 ? c0 + c4 - c2 - c6 : c0 - c4 - c2 + c6;
		const int LS = (int)((ws[0] & 255u) * (uint32_t)na) << sh;
		dc = (int)((uint32_t)v * (uint32_t)LS) >> 5;
	}
	const bool ac = chroma ? (coded >> (16 + k) & 1) : true;
	int r[4][4]; // [row][col]
	if (ac) {
		uint32_t cw[8];
		levels16(co, l8, cw);
		int tt[4][4]; // [xx][y]
#pragma unroll
		for (int y = 0; y < 4; y++) {
			int d[4];
#pragma unroll
			for (int x = 0; x < 4; x++) {
				const int pos = x * 4 + y;
				const int lev = (int)(int16_t)(cw[pos >> 1] >> ((pos & 1) * 16));
				const int wsb = (int)(ws[pos >> 2] >> ((pos & 3) * 8) & 255u);
				const int nrm = (x & y & 1) ? nb : ((x | y) & 1) ? nc : na;
				d[x] = (int)(((uint32_t)(lev * (wsb * nrm)) << sh) + 8u) >> 4;
			}
			if (chroma && y == 0) d[0] = dc; // chroma blocks take their DC from transform_dc2x2 (0 when there is none), residual.c:123-124
			const int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = (d[3] >> 1) + d[1];
			const int add = y == 0 ? 32 : 0;
			tt[0][y] = e0 + e3 + add; tt[1][y] = e1 + e2 + add; tt[2][y] = e1 - e2 + add; tt[3][y] = e0 - e3 + add;
		}
#pragma unroll
		for (int x = 0; x < 4; x++) {
			const int f0 = tt[x][0], f1 = tt[x][1], f2 = tt[x][2], f3 = tt[x][3];
			const int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = (f3 >> 1) + f1;
			r[0][x] = sat16((g0 + g3) >> 6); r[1][x] = sat16((g1 + g2) >> 6); r[2][x] = sat16((g1 - g2) >> 6); r[3][x] = sat16((g0 - g3) >> 6);
		}
	} else { // add_dc4x4
		const int v = (int)(int16_t)((dc + 32) >> 6);
#pragma unroll
		for (int y = 0; y < 4; y++)
#pragma unroll
			for (int x = 0; x < 4; x++) r[y][x] = v;
	}
	uint32_t *px;
	int stride;
	if (chroma) { px = &L.c[pc][(mb / PT_W) * 8 + ((k >> 1) & 1) * 4][(mb & (PT_W - 1)) * 2 + (k & 1)]; stride = PT_W * 2; }
	else { px = &L.y[(mb / PT_W) * 16 + BYf(k)][(mb & (PT_W - 1)) * 4 + (BXf(k) >> 2)]; stride = PT_W * 4; }
#pragma unroll
	for (int y = 0; y < 4; y++)
		px[y * stride] = add_res4(px[y * stride], r[y][0], r[y][1], r[y][2], r[y][3]);
}

// one 8x8 block (edge264_residual.c:194-343): int16 arithmetic with wraparound after a saturating dequantisation, as the
// reference's vectors; packed 16-bit lanes reproduce it exactly
E264_DEV void idct8_1dp(s16x2 d[8])
{
	const s16x2 s1 = {1, 1}, s2 = {2, 2};
	const s16x2 d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5], d6 = d[6], d7 = d[7];
	const s16x2 e0 = d0 + d4;
	const s16x2 e1 = d5 - d3 - ((d7 >> s1) + d7);
	const s16x2 e2 = d0 - d4;
	const s16x2 e3 = d1 + d7 - ((d3 >> s1) + d3);
	const s16x2 e4 = (d2 >> s1) - d6;
	const s16x2 e5 = d7 - d1 + ((d5 >> s1) + d5);
	const s16x2 e6 = (d6 >> s1) + d2;
	const s16x2 e7 = d3 + d5 + ((d1 >> s1) + d1);
	const s16x2 f0 = e0 + e6, f1 = (e7 >> s2) + e1, f2 = e2 + e4, f3 = (e5 >> s2) + e3;
	const s16x2 f4 = e2 - e4, f5 = (e3 >> s2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> s2);
	d[0] = f0 + f7; d[1] = f2 + f5; d[2] = f4 + f3; d[3] = f6 + f1;
	d[4] = f6 - f1; d[5] = f4 - f3; d[6] = f2 - f5; d[7] = f0 - f7;
}
E264_DEV void res_item8(PredLds &L, const FrameCtx &f, int item)
{
	const int mb = item >> 5, bq = (item & 31) - 24;
	const uint32_t d0 = L.hdr[mb][0], coded = L.hdr[mb][3];
	cslice_t s = f.slices + (L.hdr[mb][2] >> 16);
	const gu8 *pl = f.payload + L.hdr[mb][4];
	if (coded & E264_CODED_CHROMA_DC) pl += 16;
	const int l8 = (d0 >> 8 & E264_MBF_LEV8) ? 1 : 0;
	const gu8 *co = pl + ((__builtin_popcount(coded & 0x1111 & ((1u << (bq * 4)) - 1)) * 128) >> l8);
	const gu8 *wsp = (const gu8 *)s + offsetof(E264SliceParams, weightScale8x8) + 64; // list 1: inter Y
	const int qP = (int)(d0 >> 16 & 255), div = qP / 6, m = qP - div * 6;
	// residual.c:214-247 has two forms (qP < 36: saturate((level * LS + 2^(5 - div)) >> (6 - div)); else level * int16(LS << (div - 6)), wrapped): one
	// flow for both -- qP is per lane here, a two-way form is a divergent branch around every one of the 64 coefficients
	const int shl = max(div - 6, 0), shr = max(6 - div, 0), rnd = div < 6 ? 1 << (5 - div) : 0;
	const int lo = div < 6 ? -32768 : (int)0x80000000, hi = div < 6 ? 32767 : 0x7fffffff;
	// pass 1 (residual.c:250-296): for every j, the 1-D transform over i of d[i][j] = level c[i*8+j] dequantised; packed pairs (j, j+1)
	s16x2 t[8][4]; // [i][pair of j]
#pragma unroll
	for (int i = 0; i < 8; i++) {
		uint32_t cw[4];
		levels8(co + ((i * 16) >> l8), l8, cw);
		const v2u wv = *(const gv2u *)(wsp + i * 8);
		const uint32_t ww[2] = {wv.x, wv.y};
#pragma unroll
		for (int jp = 0; jp < 4; jp++) {
			int dq[2];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int j = jp * 2 + h, pos = i * 8 + j;
				const int lev = (int)(int16_t)(cw[jp] >> (h * 16));
				const int LS = (int)(ww[j >> 2] >> ((j & 3) * 8) & 255u) * norm8(m, pos);
				dq[h] = min(max((lev * (int)(int16_t)(LS << shl) + rnd) >> shr, lo), hi); // (stored as int16: wraps where the second form wraps)
			}
			const s16x2 v = {(short)dq[0], (short)dq[1]};
			t[i][jp] = v;
		}
	}
#pragma unroll
	for (int jp = 0; jp < 4; jp++) {
		s16x2 d[8];
#pragma unroll
		for (int i = 0; i < 8; i++) d[i] = t[i][jp];
		idct8_1dp(d);
#pragma unroll
		for (int i = 0; i < 8; i++) t[i][jp] = d[i];
	}
	// +32 on the elements with j == 0 (residual.c:298), then pass 2 over j for every i; output sample (row j, column i).
	// Repack: u[j][pair of i] = (t[i][j], t[i+1][j])
	s16x2 u[8][4];
#pragma unroll
	for (int ip = 0; ip < 4; ip++)
#pragma unroll
		for (int jp = 0; jp < 4; jp++) {
			const uint32_t a = as_u(t[2 * ip][jp]), b = as_u(t[2 * ip + 1][jp]);
			u[2 * jp][ip] = as_s2(v_perm(b, a, 0x05040100u));     // (a.lo, b.lo)
			u[2 * jp + 1][ip] = as_s2(v_perm(b, a, 0x07060302u)); // (a.hi, b.hi)
		}
	const s16x2 r32 = {32, 32}, s6 = {6, 6};
#pragma unroll
	for (int ip = 0; ip < 4; ip++) {
		s16x2 d[8];
#pragma unroll
		for (int j = 0; j < 8; j++) d[j] = u[j][ip];
		d[0] = d[0] + r32;
		idct8_1dp(d);
#pragma unroll
		for (int j = 0; j < 8; j++) u[j][ip] = d[j] >> s6;
	}
	uint32_t *px = &L.y[(mb / PT_W) * 16 + (bq >> 1) * 8][(mb & (PT_W - 1)) * 4 + (bq & 1) * 2];
#pragma unroll
	for (int j = 0; j < 8; j++) {
#pragma unroll
		for (int w = 0; w < 2; w++) {
			px[j * PT_W * 4 + w] = add_res4p(px[j * PT_W * 4 + w], u[j][2 * w], u[j][2 * w + 1]);
		}
	}
}
E264_DEV void pred_phase_residual(PredLds &L, const FrameCtx &f, int tid)
{
	const int n4 = L.rcnt[0], n8 = L.rcnt[1];
	const uint16_t *l4 = &L.lst[0][0], *l8 = &L.lst[2][0];
	for (int p = tid; p < n4; p += PT_NT)
		res_item4(L, f, l4[p]);
#ifdef E264_PRED_RES8_FIRST_WAVES
	for (int p = tid; p < n8; p += PT_NT)
		res_item8(L, f, l8[p]);
#else
	// the 8x8 blocks from the workgroup's LAST threads: the 4x4 list rarely fills all four waves, so the long 8x8 transforms run beside
	// the 4x4 ones instead of behind them on the first wave (the others wait at the barrier for the slowest: 20 % of the kernel's wave time)
	for (int p = PT_NT - 1 - tid; p < n8; p += PT_NT)
		res_item8(L, f, l8[p]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// phase 7: the tile leaves as whole rows, 16 bytes per lane; pieces of macroblocks this kernel does not own are skipped
// ---------------------------------------------------------------------------------------------------------------------
E264_DEV void pred_phase_flush(const PredLds &L, const FrameCtx &f, const PredTile &t, int tid)
{
	gu8 *Yb = f.cur + (size_t)(t.ty0 * 16) * f.sY + t.tx0 * 16;
	for (int i = tid; i < PT_H * 16 * PT_W; i += PT_NT) { // luma: 128 rows x 16 pieces
		const int row = i / PT_W, c = i & (PT_W - 1), mb = (row >> 4) * PT_W + c;
		if (L.staged[mb >> 5] >> (mb & 31) & 1)
			*(gv4u *)(Yb + (size_t)row * f.sY + c * 16) = *(const v4u *)&L.y[row][c * 4];
	}
	for (int i = tid; i < 2 * PT_H * 8 * (PT_W / 2); i += PT_NT) { // chroma: 2 planes x 64 rows x 8 pieces (2 macroblocks each)
		const int pc = i / (PT_H * 8 * (PT_W / 2)), rem = i - pc * (PT_H * 8 * (PT_W / 2));
		const int row = rem / (PT_W / 2), c = rem & (PT_W / 2 - 1), mb = (row >> 3) * PT_W + c * 2;
		const uint32_t st = L.staged[mb >> 5] >> (mb & 31) & 3; // mb is even: both bits in the same word
		gu8 *dst = plane_base(f, f.cur, 1 + pc) + (size_t)(t.ty0 * 8 + row) * f.sC + t.tx0 * 8 + c * 16;
		const v4u v = *(const v4u *)&L.c[pc][row][c * 4];
		if (st == 3) *(gv4u *)dst = v;
		else if (st == 1) { const v2u h = {v.x, v.y}; *(gv2u *)dst = h; }
		else if (st == 2) { const v2u h = {v.z, v.w}; *(gv2u *)(dst + 8) = h; }
	}
}

} // namespace
#endif
