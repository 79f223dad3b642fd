// e264_dbk.h -- e264_deblock_kernel, second generation: the in-loop deblocking filter of one picture.
//
// Device restatement of deblock_mb's filtering part, /root/reference/src/edge264_deblock.c:284-895 (the per-edge arithmetic:
// :95-152 bS < 4, :213-276 bS == 4; luma edge order :300-620, chroma :640-890), driven by the parameter records of
// e264_dbkparam2_kernel (e264_dbkp.h).
//
// Why a rewrite.  Round 1 walked a macroblock row with one half-wave, one LINE per lane, in scalar 32-bit arithmetic and
// moved every sample through LDS byte by byte: ~1000 VALU instructions per pair of macroblocks, and the kernel was bound
// by VALU issue (34 waves x 122 steps x 1000 instructions x 4 cycles on 4 SIMDs = the 1.8 ms it took per 256 pictures).
// Here a lane filters TWO adjacent lines at once in packed 16-bit arithmetic (v_pk_*_i16; the two lines of a pair share
// their 4-line segment, hence bS and tC0), so a macroblock needs 12 lanes -- 8 for luma, 4 for chroma, each chroma lane
// carrying a line pair of Cb AND of Cr in the four edge slots luma uses for its four edges -- and a wave walks FIVE
// macroblock rows, each one macroblock behind the row above:
//
//   step t, row g of the wave (lanes 12g .. 12g+11), macroblock x = t - g:
//     V phase  vertical edges: lane = row pair.  Own samples come straight from registers: every fourth step a lane fetches
//              the next FOUR macroblocks of its two rows (dk_fetch4, two or more steps ahead of their use); the left
//              neighbour's last 4 columns come from the row's strip in LDS; result -> strip.
//     H phase  horizontal edges: lane = column pair, read from the strip; the 4 rows above are rows 12..15 of the strip
//              of row g-1 (which filtered macroblock x+1's left edge in this step's V phase: exactly the decoding order
//              (x+1, y-1) before (x, y) of edge264_deblock.c), or, for the wave's first row, a small top strip that is
//              fetched from and written back to memory.
//   every 4th step each row writes the group of 4 macroblocks that has become final as whole 64-byte row pieces.
//
// The strip of a row holds 8 macroblocks (two groups of 4): the group being filtered and the one waiting for the row
// below to finish with its last rows.  Rows of different waves meet through memory: a wave publishes how many
// macroblocks of its last row have left for memory (progress[]), the wave below fetches the top rows of a group of 4 once
// that group is there.  Nothing else synchronises: the five rows of a wave run in lockstep by construction.
//
// Every function here is a plain function of (LDS, frame, lane role, step): tests/emu runs them on the host.
#ifndef E264_DBK_H
#define E264_DBK_H
#include "e264_dev.h"

namespace {

// Geometry of a wave's walk, a build-time kind K (round 4):
//   K = 2  "mixed" (rounds 2 and 3): 5 macroblock rows x 12 lanes (8 luma line pairs + 4 chroma line pairs) = 60 lanes
//   K = 0  luma only:   8 rows x 8 lanes = 64 lanes          K = 1  chroma only: 15 rows x 4 lanes = 60 lanes
// Luma and chroma deblocking never read each other's samples (edge264_deblock.c filters the three planes in turn): with waves of
// their own the luma instruction stream -- which every lane of a mixed wave executes, chroma lanes included -- serves 8 macroblocks per
// step instead of 5, and the chroma waves run a third of it (no p1 / q1 updates, no bS 4 luma filter) for 16.
// Macroblocks per fetch / flush group, a build-time choice (round 5):
//   4  strips of 8 macroblocks, 64-byte row pieces in and out, 20 KB of LDS per wave: 8 waves per picture (2 per SIMD)
//   2  strips of 4 macroblocks, 32-byte row pieces, 12.6 KB per wave: 12 waves per picture (3 per SIMD) -- VERDICT r4 item 1: at two
//      waves per SIMD the VALU pipe is ~57 % busy and a step's time is the latency of its dependent chain
#ifndef E264_DBK_GS
#define E264_DBK_GS 4
#endif
#define DK_GS E264_DBK_GS
#define DK_LG (DK_GS == 4 ? 2 : 1)
#define DK_SLOTS (2 * DK_GS)    // macroblocks a strip row holds: the group being filtered and the one waiting for the row below
static_assert(DK_GS == 4 || DK_GS == 2, "E264_DBK_GS: 4 or 2");
template <int K> struct DkGeom {
	static constexpr int ROWS = K == 0 ? 8 : K == 1 ? (DK_GS == 4 ? 15 : 16) : 5; // macroblock rows per wave (chroma with groups of 4: 15, not 16 -- 60 lanes -- so that EIGHT waves' strips fit the CU's 160 KB; a 1080p picture is 5 chroma groups either way)
	static constexpr int LANES = K == 0 ? 8 : K == 1 ? 4 : 12;    // lanes per row
	static constexpr int NLUMA = K == 1 ? 0 : 8;                  // of which luma line pairs (the others: chroma line pairs, Cb and Cr)
	static constexpr bool LUMA = K != 1, CHROMA = K != 0;
	static constexpr int YROWS = LUMA ? ROWS : 1, CROWS = CHROMA ? ROWS : 1; // strips that exist
};
#define DK_ROWS_OF(K) (DkGeom<K>::ROWS)
#define DK_STRIDE (DK_SLOTS * 16 + 16) // bytes per strip row: 8 (4) macroblocks x 16 + 16 (keeps the row pairs of a V phase on different banks)
#define DK_CR (DK_SLOTS * 8)           // chroma strip row: Cb at +0, Cr at +64 (+32)

template <int K> struct __attribute__((aligned(16))) DkWaveT { // (the members a kind does not use shrink to 16 bytes)
	uint8_t y[DkGeom<K>::YROWS][DkGeom<K>::LUMA ? 16 : 1][DkGeom<K>::LUMA ? DK_STRIDE : 16];   // luma strips: macroblock x at columns (x & 7) * 16
	uint8_t c[DkGeom<K>::CROWS][DkGeom<K>::CHROMA ? 8 : 1][DkGeom<K>::CHROMA ? DK_STRIDE : 16]; // chroma strips: macroblock x at columns (x & 7) * 8 (+ DK_CR for Cr)
	uint8_t ty[DkGeom<K>::LUMA ? 4 : 1][DkGeom<K>::LUMA ? DK_STRIDE : 16];          // rows -4..-1 above the wave's first row (luma)
	uint8_t tcp[DkGeom<K>::CHROMA ? 2 : 1][DkGeom<K>::CHROMA ? DK_STRIDE : 16];     // never holds samples: the unused taps of chroma lanes land here
	uint8_t tc[DkGeom<K>::CHROMA ? 2 : 1][DkGeom<K>::CHROMA ? DK_STRIDE : 16];      // chroma rows -2, -1 above the wave's first row (right behind tcp)
};
// (until round 5 the strips were followed by the raw parameter records of two macroblocks per row; the parameters now arrive in the
// lanes' own layout and never touch LDS: DkRaw below)

// Final samples leave with a streaming hint: nothing in this kernel reads them again, and every line they would occupy in
// the L2 pushes out a line of samples still waiting for its neighbours (a line is visited over 8 steps, the L2 of an XCD
// turns over in about 2: PMC read requests 33.4 M -> 26.1 M x 128 B per 256 pictures, same run time).
#if !defined(E264_DBK_PLAIN_STORE) && !defined(E264_HOST_INTRINSICS)
#define DK_STORE4(p, v) __builtin_nontemporal_store((v), (p))
#else
#define DK_STORE4(p, v) (*(p) = (v))
#endif
#ifdef E264_HOST_INTRINSICS
#define DK_ANY(x) true  // skipping an edge no lane filters is an optimisation only: the masked arithmetic leaves the samples alone
#else
#define DK_ANY(x) __any((x) != 0)
#endif

// Where p0 + delta / q0 - delta are clipped to 0..255 (round 5): 1 = in the pack back to bytes (v_sat_pk_u8_i16 on the eight p0 / q0 registers of a
// line pair; no later edge reads them before: the next edge's p3 is only looked at by the bS 4 luma filter, which exists on edge 0 alone), 0 = in
// dk_edge with a packed max + min per value (rounds 2 - 4).
#ifndef E264_DBK_SATPACK
#define E264_DBK_SATPACK 1
#endif
#ifndef E264_DBK_ALSKIP
#define E264_DBK_ALSKIP 0
#endif
E264_DEV bool dk_is_p0q0(int k) { return k >= 3 && k <= 16 && ((k & 3) == 3 || (k & 3) == 0); } // v[4e + 3], v[4e + 4]
E264_DEV uint32_t dk_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); } // v_bfi_b32
E264_DEV s16x2 dk_dup(uint32_t x) { return as_s2(x | x << 16); }
E264_DEV s16x2 dk_sel(s16x2 mask, s16x2 a, s16x2 b) { return as_s2(dk_bfi(as_u(mask), as_u(a), as_u(b))); }
E264_DEV s16x2 dk_abs(s16x2 v) { return __builtin_elementwise_max(v, -v); }
E264_DEV s16x2 dk_clip(s16x2 v, s16x2 lo, s16x2 hi) { return __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi); }

// What a lane is: constant over the kernel.
struct DkRole {
	int g, r;          // row of the wave, lane within the row
	bool idle, chroma;
	int pi, seg;       // line pair (luma 0..7: lines 2pi, 2pi+1; chroma 0..3), its 4-line segment in luma units
	int rowa;          // V phase: byte offset inside DkWave of the first line of the pair at slot 0 (chroma: Cb)
	int slot_mul;      // bytes per macroblock in a strip row: 16 / 8
	int cr_off;        // chroma: DK_CR; luma: 0
	uint32_t a1_mask;  // V phase: bits of the third dword that come from the macroblock itself (chroma: Cb columns 4, 5; the rest is Cr's left neighbour)
	int hT, hO, hT2, hO2; // H phase: bases of taps 0..3 / 4..9 / 10..11 / 12..19 at slot 0 (see dk_haddr)
	uint32_t luma_mask;   // all ones for luma lanes
	uint32_t tc_add;      // chroma: tC = tC0 + 1
};

template <int K> E264_DEV bool dk_chroma(const DkRole &R) { return K == 0 ? false : K == 1 ? true : R.chroma; } // (a constant for the one-plane kinds)
template <int K> E264_DEV DkRole dk_role(int lane)
{
	typedef DkGeom<K> G;
	typedef DkWaveT<K> DkWave;
	DkRole R;
	R.g = lane / G::LANES; R.r = lane - R.g * G::LANES;
	R.idle = R.g >= G::ROWS;
	if (R.idle) R.g = G::ROWS - 1; // addresses stay valid; the lane never acts
	R.chroma = K == 2 ? R.r >= G::NLUMA : K == 1; // (a constant for the one-plane kinds: everything derived from it below folds)
	R.pi = R.chroma ? R.r - G::NLUMA : R.r;
	R.seg = R.chroma ? R.pi : R.pi >> 1;
	const int g = R.g, col = 2 * R.pi;
	const int oy = (int)offsetof(DkWave, y) + (G::LUMA ? g : 0) * 16 * DK_STRIDE, oc = (int)offsetof(DkWave, c) + (G::CHROMA ? g : 0) * 8 * DK_STRIDE;
	const int uy = g ? oy - 4 * DK_STRIDE : (int)offsetof(DkWave, ty);                    // rows -4..-1: rows 12..15 of the strip above
	const int uc = g ? oc - 2 * DK_STRIDE : (int)offsetof(DkWave, tc);                    // chroma rows -2, -1
	R.rowa = (R.chroma ? oc : oy) + 2 * R.pi * DK_STRIDE;
	R.slot_mul = R.chroma ? 8 : 16;
	R.cr_off = R.chroma ? DK_CR : 0;
	R.a1_mask = R.chroma ? 0x0000ffffu : 0xffffffffu;
	R.luma_mask = R.chroma ? 0u : 0xffffffffu;
	R.tc_add = R.chroma ? 1u : 0u;
	if (!R.chroma) {
		R.hT = uy + col; R.hO = oy + col; R.hT2 = R.hO; R.hO2 = R.hO;
	} else { // taps 2..5: Cb rows -2..1, 6..9: Cb rows 2..5, 10..13: Cr rows -2..1, 14..17: Cr rows 2..5
		R.hT = uc - 2 * DK_STRIDE + col;
		R.hO = oc + col;
		R.hT2 = uc + DK_CR + col - 6 * DK_STRIDE;
		R.hO2 = oc + DK_CR + col - 8 * DK_STRIDE;
	}
	return R;
}

// ---------------------------------------------------------------------------------------------------------------------
// One edge on 8 sample pairs p3 p2 p1 p0 | q0 q1 q2 q3 = v[0..7] (edge264_deblock.c:95-152, 213-276).
//   alphaE  alpha, or 0 where bS == 0 (nothing is below 0: the edge is left alone)
//   betal   beta for luma lanes, 0 for chroma lanes (ap = aq = false: chroma never touches p1 / q1)
//   tc0     tC0 (chroma: + 1, its tC); 0 where bS == 4
//   STRONG  0: no lane of this slot can have bS 4; 1: luma + chroma macroblock edge; 2: chroma macroblock edge only
// ---------------------------------------------------------------------------------------------------------------------
//   LUMA    false: every lane of the wave is a chroma lane (K = 1): ap = aq = 0 at compile time, the p1 / q1 updates and the luma bS 4 filter fall away
template <int STRONG, bool LUMA>
E264_DEV void dk_edge(s16x2 *v, s16x2 alphaE, s16x2 beta, s16x2 betal, s16x2 tc0, s16x2 small_thr, s16x2 strong)
{
	s16x2 &p3 = v[0], &p2 = v[1], &p1 = v[2], &p0 = v[3], &q0 = v[4], &q1 = v[5], &q2 = v[6], &q3 = v[7];
#if E264_DBK_ALSKIP // (wave-uniform) no lane has a boundary strength here: not even the three differences are needed (encoder-made content: 89 % of all bS are 0)
	if (!DK_ANY(as_u(alphaE)))
		return;
#endif
	const s16x2 d = p0 - q0, dpq = dk_abs(d);
	// filterSamplesFlag: all three differences below their thresholds <=> all three (difference - threshold) negative
	const s16x2 go = ((dpq - alphaE) & (dk_abs(p1 - p0) - beta) & (dk_abs(q1 - q0) - beta)) >> 15;
#ifdef E264_ABL_DBK_NOFILTER // timing ablation
	if (as_u(go) != 0x12345) return;
#endif
#ifndef E264_DBK_NOEARLY // (measuring aid: what the wave-wide "no lane filters this edge" test is worth)
	if (!DK_ANY(as_u(go)))
		return;
#endif
	const s16x2 zero2 = {0, 0};
	const s16x2 ap = LUMA ? ((dk_abs(p2 - p0) - betal) >> 15) & go : zero2, aq = LUMA ? ((dk_abs(q2 - q0) - betal) >> 15) & go : zero2;
	// ---- bS < 4
	const s16x2 tc = tc0 - ap - aq;
	const s16x2 delta = dk_clip((d * (short)-4 + p1 - q1 + (short)4) >> 3, -tc, tc) & go;
	const s16x2 avg = as_s2(v_lerp_u8(as_u(p0), as_u(q0), 0x00010001u)); // (p0 + q0 + 1) >> 1
	const s16x2 z = {0, 0}, m255 = {255, 255};
	const s16x2 dp1 = LUMA ? dk_clip((p2 + avg - p1 * (short)2) >> 1, -tc0, tc0) & ap : zero2; // tc0 is 0 on bS 4 lines: p1 stays
	const s16x2 dq1 = LUMA ? dk_clip((q2 + avg - q1 * (short)2) >> 1, -tc0, tc0) & aq : zero2;
#if E264_DBK_SATPACK // p0 / q0 leave the edge unclipped (-27 .. 282): the pack back to bytes saturates them (dk_vpass / dk_hpass), 4 instructions per edge fewer
	s16x2 np0 = p0 + delta, nq0 = q0 - delta;
#else
	s16x2 np0 = dk_clip(p0 + delta, z, m255), nq0 = dk_clip(q0 - delta, z, m255);
#endif
	s16x2 np1 = p1 + dp1, nq1 = q1 + dq1;
	if (STRONG) {
		const s16x2 S = strong & go;
		if (DK_ANY(as_u(S))) { // ---- bS == 4
			const s16x2 wp0 = (p1 * (short)2 + p0 + q1 + (short)2) >> 2, wq0 = (q1 * (short)2 + q0 + p1 + (short)2) >> 2;
			np0 = dk_sel(S, wp0, np0); nq0 = dk_sel(S, wq0, nq0);
			if (STRONG == 1) {
				const s16x2 sm = (dpq - small_thr) >> 15; // |p0 - q0| < (alpha >> 2) + 2
				const s16x2 sp = ap & sm & S, sq = aq & sm & S;
				const s16x2 pq = p0 + q0, tp = p2 + p1 + pq, tq = q2 + q1 + pq;
				np0 = dk_sel(sp, (tp + p1 + pq + q1 + (short)4) >> 3, np0);
				nq0 = dk_sel(sq, (tq + q1 + pq + p1 + (short)4) >> 3, nq0);
				np1 = dk_sel(sp, (tp + (short)2) >> 2, np1);
				nq1 = dk_sel(sq, (tq + (short)2) >> 2, nq1);
				p2 = dk_sel(sp, ((p3 + p2) * (short)2 + tp + (short)4) >> 3, p2);
				q2 = dk_sel(sq, ((q3 + q2) * (short)2 + tq + (short)4) >> 3, q2);
			}
		}
	}
	p0 = np0; q0 = nq0; p1 = np1; q1 = nq1;
}

// The parameters of the four edge slots of a lane for one direction, ready for dk_edge.  They are fetched for BOTH
// directions in one batch at the top of a step (two LDS round trips in all: the record's bytes, then the tC0 table) --
// fetched where they are used, slot by slot behind the branches that skip idle edges, they cost a dozen exposed LDS
// latencies per step with only two waves per SIMD to hide them.
#ifndef E264_DBK_PIN_PARAMS
#define E264_DBK_PIN_PARAMS 1
#endif
struct DkPrm { s16x2 al[4], be[4], tc[4]; s16x2 thr, strong0, strong2; }; // al: alpha, or 0 where bS is 0
// Luma lanes: slot e = edge e of the macroblock.  Chroma lanes: slot 0 = Cb macroblock edge, 1 = Cb inner edge, 2 = Cr
// macroblock edge, 3 = Cr inner edge.
// Round 6: the parameter kernel delivers them per lane (e264_dbkp.h: one 16-byte piece per plane kind, direction and segment): a lane
// fetches the two pieces of its segment (V and H direction) and its plane's 8 bytes of beta straight into registers one step ahead and spreads each
// byte into a packed pair with one v_perm -- ~24 byte permutes and 4 bit-field extracts where rounds 2 - 5 spent ~26 LDS reads, a dependent table
// look-up and ~150 VALU instructions per step.
struct DkRaw { v2u v, h; v2u w; }; // the lane's pieces of one macroblock: vertical edges (direction 0), horizontal edges (direction 1); its plane kind's 8 bytes of beta
E264_DEV s16x2 dk_dupb(uint32_t w, int k) { return as_s2(v_perm(0, w, 0x0c000c00u | (uint32_t)k * 0x00010001u)); } // byte k of w in both halves
E264_DEV s16x2 dk_dupb2(uint32_t hi, uint32_t lo, int k) { return as_s2(v_perm(hi, lo, 0x0c000c00u | (uint32_t)k * 0x00010001u)); } // byte k of {hi, lo}
E264_DEV uint32_t dk_bit_mask(uint32_t w, int bit) { return (uint32_t)((int32_t)(w << (31 - bit)) >> 31); } // all ones where the bit is set (v_bfe_i32)
// beta of the four slots: luma {macroblock edge of this direction, inner, inner, inner} out of bytes {inner, left, top}; chroma {Cb mb, Cb inner, Cr mb, Cr inner}
// out of bytes {Cb inner, left, top, Cr inner, left, top}
template <int K> E264_DEV void dk_params1(const v2u &piece, const v2u &w, int dir, const DkRole &R, DkPrm &P)
{
	const uint32_t tcw = piece.y & 0x7f7f7f7fu;
#pragma unroll
	for (int e = 0; e < 4; e++) {
		P.al[e] = dk_dupb(piece.x, e);
		P.tc[e] = dk_dupb(tcw, e);
	}
	if (K == 0) { // (every lane a luma lane: constants)
		P.be[0] = dk_dupb(w.x, 1 + dir); P.be[1] = P.be[2] = P.be[3] = dk_dupb(w.x, 0);
	} else if (K == 1) {
		P.be[0] = dk_dupb2(w.y, w.x, 1 + dir); P.be[1] = dk_dupb2(w.y, w.x, 0); P.be[2] = dk_dupb2(w.y, w.x, 4 + dir); P.be[3] = dk_dupb2(w.y, w.x, 3);
	} else { // mixed waves: by the lane's kind
		const bool c = R.chroma;
		const uint32_t s0 = 0x0c000c00u | (uint32_t)(1 + dir) * 0x00010001u, s1 = 0x0c000c00u, s2 = 0x0c000c00u | (uint32_t)(c ? 4 + dir : 0) * 0x00010001u, s3 = 0x0c000c00u | (c ? 3u : 0u) * 0x00010001u;
		P.be[0] = as_s2(v_perm(w.y, w.x, s0)); P.be[1] = as_s2(v_perm(w.y, w.x, s1)); P.be[2] = as_s2(v_perm(w.y, w.x, s2)); P.be[3] = as_s2(v_perm(w.y, w.x, s3));
	}
	const s16x2 two = {2, 2}, sh2 = {2, 2};
	P.thr = (P.al[0] >> sh2) + two; // only looked at where slot 0 has bS 4: alphaE is alpha there
	P.strong0 = as_s2(dk_bit_mask(piece.y, 7));
	P.strong2 = as_s2(dk_bit_mask(piece.y, 23));
}
template <int K> E264_DEV void dk_params(const DkRaw &raw, const DkRole &R, DkPrm P[2])
{
	dk_params1<K>(raw.v, raw.w, 0, R, P[0]);
	dk_params1<K>(raw.h, raw.w, 1, R, P[1]);
}
// The four edge slots of a line pair held in v[0..19] (positions -4..15): slot e works on v[4e .. 4e+7]; in chroma lanes
// only p1 p0 q0 q1 = v[4e+2 .. 4e+5] matter.
template <int K>
E264_DEV void dk_filter(s16x2 *v, const DkPrm &P, const DkRole &R)
{
#pragma unroll
	for (int e = 0; e < 4; e++) {
		const s16x2 bl = as_s2(as_u(P.be[e]) & R.luma_mask);
		if (K == 1) { // chroma lanes only: slots 0 and 2 are the macroblock edges of Cb and Cr
			if (e == 0) dk_edge<2, false>(v, P.al[0], P.be[0], bl, P.tc[0], P.thr, P.strong0);
			else if (e == 2) dk_edge<2, false>(v + 8, P.al[2], P.be[2], bl, P.tc[2], P.thr, P.strong2);
			else dk_edge<0, false>(v + 4 * e, P.al[e], P.be[e], bl, P.tc[e], P.thr, P.thr);
		} else if (e == 0) dk_edge<1, true>(v, P.al[0], P.be[0], bl, P.tc[0], P.thr, P.strong0);
		else if (e == 2 && K == 2) dk_edge<2, true>(v + 8, P.al[2], P.be[2], bl, P.tc[2], P.thr, P.strong2); // (mixed waves: the Cr macroblock edge of the chroma lanes)
		else dk_edge<0, true>(v + 4 * e, P.al[e], P.be[e], bl, P.tc[e], P.thr, P.thr);
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// input: the unfiltered samples of macroblock (x, y) -> registers.  Luma lane: its two rows (16 bytes each); chroma lane:
// a = {Cb row 2pi, Cr row 2pi}, b = the same of row 2pi+1 (8 bytes each).  LOADS ONLY.
// ---------------------------------------------------------------------------------------------------------------------
// FOUR macroblocks per fetch (round 3; round 2 fetched one macroblock per step with four 8-byte loads per lane).  A lane's row of
// samples is 16 bytes per macroblock (8 + 8 for a chroma lane: Cb and Cr), so the 128-byte line it sits in was asked for at
// eight different steps ~36 us apart -- longer than a line survives in the XCD's L2 under the 4.6 MB of lines the 32 CUs
// have open at any time: the L2 re-fetched it (26 M read requests of 128 B per 256 pictures = 4.2 x the samples,
// profiles/r03_pmc_hbm_requests.txt).  Now every fourth step a lane fetches the next four macroblocks of its two rows with
// EIGHT 16-byte loads (64 contiguous bytes per row; a chroma lane: 32 of Cb + 32 of Cr per row), the same instruction
// sequence for every lane at per-lane addresses (one sequence per role would make the second role's loads wait for the
// first's: they share destination registers).  Pieces outside the picture are fetched from clamped addresses and never used.
//   load k = 0..7:  address = row base + ((k & 1) * 16 + (k >> 1 & 1) * e1) + (k >> 2) * d2
//     luma lane:    e1 = 32: pieces k & 3 = macroblock k & 3 of row (k >> 2);          chroma lane: e1 = distance Cb -> Cr:
//                   piece k & 1 = macroblocks (2 (k & 1), 2 (k & 1) + 1) of plane (k >> 1 & 1), row (k >> 2)
struct DkSrc { const gu8 *base; int e1, d2; }; // row y of the lane's wave: address of macroblock 0, see above
template <int K> E264_DEV DkSrc dk_src(const FrameCtx &f, const DkRole &R, int y)
{
	DkSrc s;
	if (!dk_chroma<K>(R)) { s.base = f.cur + (size_t)(y * 16 + 2 * R.pi) * f.sY; s.e1 = 32; s.d2 = f.sY; }
	else { s.base = plane_base(f, f.cur, 1) + (size_t)(y * 8 + 2 * R.pi) * f.sC; s.e1 = f.sC >> 1; s.d2 = f.sC; }
	return s;
}
// macroblocks x0 .. x0 + 3 of the lane's row -> N[0..7].  LOADS ONLY.  A chroma piece may start one macroblock before the row
// (x0 = -1: its second half is macroblock 0) or end one after it: 8 bytes before / after the row, which are the previous /
// next row, the end of the luma plane, or the slack every frame allocation ends with (e264hip_frame_alloc).
// (groups of 2, DK_GS == 2: FOUR loads, pieces k & 1 of row k >> 1; a chroma lane: piece = plane k & 1, both macroblocks)
template <int K> E264_DEV void dk_fetch4(const DkSrc &S, const DkRole &R0, int x0, int wm, v4u N[2 * DK_GS])
{
	DkRole R = R0;
	R.chroma = dk_chroma<K>(R0);
#ifdef E264_ABL_DBK_NOLOAD // timing ablation
	if (R.slot_mul) { for (int k = 0; k < 2 * DK_GS; k++) N[k] = (v4u){(uint32_t)x0, 1, 2, 3}; return; }
#endif
#pragma unroll
	for (int k = 0; k < 2 * DK_GS; k++) {
		// first macroblock of the piece: luma piece i = k % GS; chroma piece i % (GS / 2) = two macroblocks of plane i / (GS / 2)
		const int i = k & (DK_GS - 1);
		const int mb = R.chroma ? min(max(x0 + 2 * (i & (DK_GS / 2 - 1)), -1), wm - 1) : min(max(x0 + i, 0), wm - 1);
		const int plane_off = R.chroma ? (i / (DK_GS / 2)) * S.e1 : 0;
		N[k] = *(const gv4u *)(S.base + mb * R.slot_mul + plane_off + (k >> DK_LG) * S.d2);
	}
}
// the registers of macroblock x0 + k out of a fetched group: a = the lane's first row (luma: 16 bytes; chroma: Cb 8 bytes, Cr 8
// bytes), b = its second row
template <int K> E264_DEV void dk_pick(const v4u N[2 * DK_GS], const DkRole &R0, int k, v4u &a, v4u &b)
{
	DkRole R = R0;
	R.chroma = dk_chroma<K>(R0);
#pragma unroll
	for (int row = 0; row < 2; row++) {
		const v4u *L = N + DK_GS * row;
		const v4u lu = L[k];                        // luma: piece k
		const v4u cb = L[k >> 1], cr = L[DK_GS / 2 + (k >> 1)]; // chroma: half (k & 1) of the plane's piece k >> 1
		v4u ch;
		if (k & 1) { ch.x = cb.z; ch.y = cb.w; ch.z = cr.z; ch.w = cr.w; }
		else { ch.x = cb.x; ch.y = cb.y; ch.z = cr.x; ch.w = cr.y; }
		v4u &o = row ? b : a;
		o.x = R.chroma ? ch.x : lu.x; o.y = R.chroma ? ch.y : lu.y; o.z = R.chroma ? ch.z : lu.z; o.w = R.chroma ? ch.w : lu.w;
	}
}
// the lane's two parameter pieces of macroblock (x, y) and its 8 bytes of beta.  LOADS ONLY.
template <int K> E264_DEV void dk_fetch_prm(const FrameCtx &f, const DkRole &R, int x, int y, DkRaw &p)
{
	const gu8 *rec = f.dbk + (size_t)(y * f.wm + x) * E264_DBK_BYTES;
	const v4u both = *(const gv4u *)(rec + (dk_chroma<K>(R) ? 64 : 0) + R.seg * 16); // {V piece, H piece}
	p.v = (v2u){both.x, both.y};
	p.h = (v2u){both.z, both.w};
	p.w = *(const gv2u *)(rec + 128 + (dk_chroma<K>(R) ? 8 : 0));
}

// ---------------------------------------------------------------------------------------------------------------------
// V phase: the vertical edges of macroblock x of the lane's row (a, b: dk_fetch's registers)
// ---------------------------------------------------------------------------------------------------------------------
template <int K> E264_DEV void dk_vpass(DkWaveT<K> &W, const DkPrm &P, const DkRole &R, const v4u &ra, const v4u &rb, int x)
{
	uint8_t *W8 = (uint8_t *)&W;
	const bool chroma = dk_chroma<K>(R);
	const int own = R.rowa + (x & (DK_SLOTS - 1)) * R.slot_mul;
	const int prev = R.rowa + ((x - 1) & (DK_SLOTS - 1)) * R.slot_mul + R.slot_mul - 4; // the left neighbour's last 4 bytes (garbage at x = 0: its bS is 0)
	const uint32_t La = *(const uint32_t *)(W8 + prev), Lb = *(const uint32_t *)(W8 + prev + DK_STRIDE);
	const uint32_t Ca = *(const uint32_t *)(W8 + prev + R.cr_off), Cb = *(const uint32_t *)(W8 + prev + R.cr_off + DK_STRIDE);
	// the line as 5 dwords.  Chroma: {Cb left, Cb 0..3, Cb 4,5 | Cr left 6,7, Cr 0..3, Cr 4..7}
	uint32_t A[5] = {La, ra.x, dk_bfi(R.a1_mask, ra.y, Ca), ra.z, ra.w};
	uint32_t B[5] = {Lb, rb.x, dk_bfi(R.a1_mask, rb.y, Cb), rb.z, rb.w};
	s16x2 v[20];
#pragma unroll
	for (int k = 0; k < 20; k++)
		v[k] = as_s2(v_perm(B[k >> 2], A[k >> 2], 0x0c040c00u + (uint32_t)(k & 3) * 0x00010001u)); // (a_k, b_k) as two 16-bit lanes
	dk_filter<K>(v, P, R);
#pragma unroll
	for (int d = 0; d < 5; d++) {
#if E264_DBK_SATPACK // v[4d] (d >= 1) and v[4d + 3] (d <= 3) are a q0 / p0: saturated to two bytes first, picked up from bytes 0, 1
		const uint32_t t01 = d >= 1 ? v_perm(as_u(v[4 * d + 1]), v_sat_pk_u8_i16(as_u(v[4 * d])), 0x06010400u) : v_perm(as_u(v[4 * d + 1]), as_u(v[4 * d]), 0x06020400u);
		const uint32_t t23 = d <= 3 ? v_perm(v_sat_pk_u8_i16(as_u(v[4 * d + 3])), as_u(v[4 * d + 2]), 0x05020400u) : v_perm(as_u(v[4 * d + 3]), as_u(v[4 * d + 2]), 0x06020400u);
#else
		const uint32_t t01 = v_perm(as_u(v[4 * d + 1]), as_u(v[4 * d]), 0x06020400u), t23 = v_perm(as_u(v[4 * d + 3]), as_u(v[4 * d + 2]), 0x06020400u);
#endif
		A[d] = v_perm(t23, t01, 0x05040100u);
		B[d] = v_perm(t23, t01, 0x07060302u);
	}
	*(uint32_t *)(W8 + prev) = A[0];
	*(uint32_t *)(W8 + prev + DK_STRIDE) = B[0];
	if (!chroma) {
		*(v4u *)(W8 + own) = (v4u){A[1], A[2], A[3], A[4]};
		*(v4u *)(W8 + own + DK_STRIDE) = (v4u){B[1], B[2], B[3], B[4]};
	} else {
		*(uint32_t *)(W8 + prev + DK_CR) = dk_bfi(0x0000ffffu, Ca, A[2]);
		*(uint32_t *)(W8 + prev + DK_CR + DK_STRIDE) = dk_bfi(0x0000ffffu, Cb, B[2]);
		*(v2u *)(W8 + own) = (v2u){A[1], dk_bfi(0x0000ffffu, A[2], ra.y)};
		*(v2u *)(W8 + own + DK_STRIDE) = (v2u){B[1], dk_bfi(0x0000ffffu, B[2], rb.y)};
		*(v2u *)(W8 + own + DK_CR) = (v2u){A[3], A[4]};
		*(v2u *)(W8 + own + DK_CR + DK_STRIDE) = (v2u){B[3], B[4]};
	}
}

// The V phase of a step in which NO macroblock of the wave has an edge to filter (every bS of every row's record is 0): the samples
// only move from the fetch registers into the strip, exactly where dk_vpass would have left them (its stores with nothing changed: the left
// neighbour's columns stay as they are).  Round 5: on encoder-made streams 53 % of the macroblocks have every bS 0 and they come in
// runs (skipped background), tools/stream_stats.py; the synthetic bench GOP never takes this path.
template <int K> E264_DEV void dk_vcopy(DkWaveT<K> &W, const DkRole &R, const v4u &ra, const v4u &rb, int x)
{
	uint8_t *W8 = (uint8_t *)&W;
	const int own = R.rowa + (x & (DK_SLOTS - 1)) * R.slot_mul;
	if (!dk_chroma<K>(R)) {
		*(v4u *)(W8 + own) = ra;
		*(v4u *)(W8 + own + DK_STRIDE) = rb;
	} else {
		*(v2u *)(W8 + own) = (v2u){ra.x, ra.y};
		*(v2u *)(W8 + own + DK_STRIDE) = (v2u){rb.x, rb.y};
		*(v2u *)(W8 + own + DK_CR) = (v2u){ra.z, ra.w};
		*(v2u *)(W8 + own + DK_CR + DK_STRIDE) = (v2u){rb.z, rb.w};
	}
}
// can any of the lane's eight edge slots filter at all?  (alpha is 0 where bS is 0, and for the low indexA no edge passes)
E264_DEV uint32_t dk_any_bs(const DkRaw &raw) { return raw.v.x | raw.h.x; }
#ifndef E264_DBK_ZEROSKIP
#define E264_DBK_ZEROSKIP 0 // measured (profiles/r05_ablations.txt item 5): the eight macroblocks of a step lie on a diagonal through eight rows; on the
                            // encoder-made fixtures 58 % of the macroblocks have no edge but only 6.6 % of the steps: no gain there, -1.4 % on the bench GOP
#endif

// ---------------------------------------------------------------------------------------------------------------------
// H phase: the horizontal edges; the lane owns columns 2pi, 2pi+1 (chroma: of Cb and of Cr)
// ---------------------------------------------------------------------------------------------------------------------
E264_DEV int dk_haddr(int bT, int bO, int bT2, int bO2, int k)
{
	return (k < 4 ? bT + k * DK_STRIDE : k < 10 ? bO + (k - 4) * DK_STRIDE : k < 12 ? bT2 + (k - 4) * DK_STRIDE : bO2 + (k - 4) * DK_STRIDE);
}
template <int K> E264_DEV void dk_hpass(DkWaveT<K> &W, const DkPrm &P, const DkRole &R, int x)
{
	uint8_t *W8 = (uint8_t *)&W;
	const int sl = (x & (DK_SLOTS - 1)) * R.slot_mul;
	const int bT = R.hT + sl, bO = R.hO + sl, bT2 = R.hT2 + sl, bO2 = R.hO2 + sl;
	s16x2 v[20];
#pragma unroll
	for (int k = 0; k < 20; k++)
		v[k] = as_s2(v_perm(0, *(const uint16_t *)(W8 + dk_haddr(bT, bO, bT2, bO2, k)), 0x0c010c00u));
	dk_filter<K>(v, P, R);
#pragma unroll
	for (int k = 1; k < 19; k++)
		*(uint16_t *)(W8 + dk_haddr(bT, bO, bT2, bO2, k)) = (uint16_t)((E264_DBK_SATPACK && dk_is_p0q0(k)) ? v_sat_pk_u8_i16(as_u(v[k])) : v_perm(0, as_u(v[k]), 0x0c0c0200u));
}

// ---------------------------------------------------------------------------------------------------------------------
// output: group q (macroblocks 4q .. 4q+3) of row y, all 16 rows, as 64-byte row pieces (32 for a chroma plane)
// ---------------------------------------------------------------------------------------------------------------------
template <int K> E264_DEV void dk_flush(const DkWaveT<K> &W, const FrameCtx &f, const DkRole &R, int q, int y)
{
	typedef DkGeom<K> G;
	const int s0 = (q * DK_GS) & (DK_SLOTS - 1), x0 = q * DK_GS;
#ifdef E264_ABL_DBK_NOSTORE // timing ablation
	if (f.wm > 0) return;
#endif
	if (G::LUMA) {
#pragma unroll
		for (int it = 0; it < (16 * DK_GS + G::LANES - 1) / G::LANES; it++) { // luma: 16 rows x 4 (2) pieces of 16 bytes, dealt to the row's lanes
			const int idx = it * G::LANES + R.r, row = min(idx >> DK_LG, 15), m = idx & (DK_GS - 1);
			const v4u val = *(const v4u *)&W.y[G::LUMA ? R.g : 0][row][(s0 + m) * 16];
			if (idx < 16 * DK_GS && x0 + m < f.wm)
				DK_STORE4((gv4u *)(f.cur + (size_t)(y * 16 + row) * f.sY + (x0 + m) * 16), val);
		}
	}
	if (G::CHROMA) {
#pragma unroll
		for (int it = 0; it < (8 * DK_GS + G::LANES - 1) / G::LANES; it++) { // chroma: 2 planes x 8 rows x 2 (1) pieces of 16 bytes (two macroblocks each)
			const int idx = it * G::LANES + R.r, pl = min(idx / (4 * DK_GS), 1), row = idx / (DK_GS / 2) & 7, h = idx & (DK_GS / 2 - 1);
			const v4u val = *(const v4u *)&W.c[G::CHROMA ? R.g : 0][row][pl * DK_CR + (s0 + 2 * h) * 8];
			const int n = f.wm - (x0 + 2 * h);
			gu8 *dst = plane_base(f, f.cur, 1 + pl) + (size_t)(y * 8 + row) * f.sC + (x0 + 2 * h) * 8;
			if (idx < 8 * DK_GS && n >= 2) DK_STORE4((gv4u *)dst, val);
			else if (idx < 8 * DK_GS && n == 1) *(gv2u *)dst = (v2u){val.x, val.y};
		}
	}
}

// The top strip of the wave's first row y0 (> 0): group q, lanes 0..23 of the wave: 16 luma pieces (rows -4..-1) and
// 8 chroma pieces (2 planes x rows -2, -1 x 2).  fetch: memory -> register; commit: register -> LDS; flush: LDS -> memory.
struct DkTopAddr { int lds; gu8 *mem; int n; }; // n: 16-byte piece (2), 8-byte piece (1), nothing (0)
// lanes of the wave that carry the top strip: mixed 0..15 luma + 16..23 chroma, luma-only 0..15, chroma-only 0..7
template <int K> E264_DEV DkTopAddr dk_top_addr(const FrameCtx &f, int lane, int q, int y0)
{
	typedef DkWaveT<K> DkWave;
	DkTopAddr t;
	const int s0 = (q * DK_GS) & (DK_SLOTS - 1), x0 = q * DK_GS;
	const int nl = K == 1 ? 0 : 4 * DK_GS; // luma pieces come first: 4 rows x the group's macroblocks
	if (lane < nl) {
		const int row = lane >> DK_LG, m = lane & (DK_GS - 1);
		t.lds = (int)offsetof(DkWave, ty) + row * DK_STRIDE + (s0 + m) * 16;
		t.mem = f.cur + (size_t)(y0 * 16 - 4 + row) * f.sY + (x0 + m) * 16;
		t.n = x0 + m < f.wm ? 2 : 0;
	} else {
		const int i = lane - nl, pl = i / DK_GS & 1, row = i / (DK_GS / 2) & 1, h = i & (DK_GS / 2 - 1); // 2 planes x 2 rows x pieces of two macroblocks
		t.lds = (int)offsetof(DkWave, tc) + row * DK_STRIDE + pl * DK_CR + (s0 + 2 * h) * 8;
		t.mem = plane_base(f, f.cur, 1 + pl) + (size_t)(y0 * 8 - 2 + row) * f.sC + (x0 + 2 * h) * 8;
		t.n = (K != 0 && i < 2 * DK_GS) ? min(max(f.wm - (x0 + 2 * h), 0), 2) : 0;
	}
	return t;
}
template <int K> E264_DEV void dk_top_fetch(const FrameCtx &f, int lane, int q, int y0, v4u &tt)
{
	const DkTopAddr t = dk_top_addr<K>(f, lane, q, y0);
	if (t.n == 2) tt = *(const gv4u *)t.mem;
	else if (t.n == 1) { const v2u h = *(const gv2u *)t.mem; tt.x = h.x; tt.y = h.y; }
}
template <int K> E264_DEV void dk_top_commit(DkWaveT<K> &W, const FrameCtx &f, int lane, int q, int y0, const v4u &tt)
{
	const DkTopAddr t = dk_top_addr<K>(f, lane, q, y0);
	if (t.n) *(v4u *)((uint8_t *)&W + t.lds) = tt;
}
template <int K> E264_DEV void dk_top_flush(const DkWaveT<K> &W, const FrameCtx &f, int lane, int q, int y0)
{
	const DkTopAddr t = dk_top_addr<K>(f, lane, q, y0);
	const v4u val = *(const v4u *)((const uint8_t *)&W + t.lds);
	if (t.n == 2) *(gv4u *)t.mem = val;
	else if (t.n == 1) *(gv2u *)t.mem = (v2u){val.x, val.y};
}

// ---------------------------------------------------------------------------------------------------------------------
// What a lane does at step t of its wave's walk over rows y0 .. y0+ROWS-1 (t runs from DK_FIRST_STEP: the pipeline fills first).
#define DK_FIRST_STEP (-4)
// ---------------------------------------------------------------------------------------------------------------------
struct DkPlan {
	int x;             // the row's macroblock at this step
	bool act;          // V and H phases
	bool prm_fetch;                       // parameters of x+1 -> the register set of the other parity (used by the next step)
	bool grp_fetch;                       // (t % 4 == 0) samples of x+2 .. x+5 -> registers: consumed at steps t+2 .. t+5
	int top_fetch, top_commit;            // group of the top strip to fetch / commit, -1: none (wave lanes 0..23, when the wave has rows above)
	int flush, top_flush;                 // group to write out before the V phase, -1: none
	bool publish;                         // (wave-uniform) the groups written at the top of this step are announced at its end
};
E264_DEV int dk_groups(int wm) { return (wm + DK_GS - 1) >> DK_LG; }
// the last row of the wave (g = ROWS - 1) writes its last group at the top of step t0 + 1, t0 the first multiple of 4 with t0 - g >= 4 * groups
template <int K> E264_DEV int dk_last_step(int wm) { return DK_GS * dk_groups(wm) + ((DkGeom<K>::ROWS - 1 + DK_GS - 1) & ~(DK_GS - 1)) + 1; }
E264_DEV DkPlan dk_plan(int t, const DkRole &R, bool row_ok, bool top, int wm)
{
	DkPlan p;
	const int nq = dk_groups(wm);
	p.x = t - R.g;
	p.act = row_ok && p.x >= 0 && p.x < wm;
	p.prm_fetch = row_ok && p.x + 1 >= 0 && p.x + 1 < wm;
	p.grp_fetch = (t & (DK_GS - 1)) == 0 && row_ok && p.x + 1 + DK_GS >= 0 && p.x + 2 < wm;
	// the wave's first row filters group Q's first macroblock at t = 4Q: fetched at t = 4Q - 2, committed at t = 4Q - 1 (groups of 2: 2Q ...)
	p.top_fetch = (top && ((t + 2) & (DK_GS - 1)) == 0 && (t + 2) >> DK_LG < nq) ? (t + 2) >> DK_LG : -1;
	// (groups of 2: committed at the top of step 2Q itself -- the top strip has four slots, group Q - 2 leaves them at the top of step 2Q - 1)
	const int tc = t + (DK_GS == 4 ? 1 : 0);
	p.top_commit = (top && (tc & (DK_GS - 1)) == 0 && tc >> DK_LG < nq) ? tc >> DK_LG : -1;
	// after step t0 (a multiple of 4) macroblocks 0 .. t0-g-1 of row g are final, rows 13..15 included (the row below has
	// passed them); whole groups: up to ((t0 - g) >> 2) - 1 (t0/4 - 1 for the first row, t0/4 - 2 for rows 1..4, ...).  They are
	// written at the TOP of step t0 + 1 (their strip slots are reused in that step's V phase at the earliest) and announced at its
	// END: the stores then had a whole step to drain and the release fence does not wait for them.
	const int t0 = t - 1;
	p.publish = (t0 & (DK_GS - 1)) == 0 && t0 >= DK_GS;
	const int qf = ((t0 - R.g) >> DK_LG) - 1;
	p.flush = (p.publish && row_ok && qf >= 0 && qf < nq) ? qf : -1;
	p.top_flush = (p.publish && top && (t0 >> DK_LG) - 1 < nq) ? (t0 >> DK_LG) - 1 : -1;
	return p;
}
// macroblocks of row g that have left for memory with the flush of step t (a publishing step)
E264_DEV int dk_progress(int t, int g, int wm) { return min(max(((t - 1 - g) >> DK_LG) * DK_GS, 0), wm); }

} // namespace
#endif
