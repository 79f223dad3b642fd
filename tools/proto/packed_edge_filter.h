/* PROTOTYPE for the next round: the H.264 edge filter (8.7.2.3 / 8.7.2.4; reference
 * edge264_deblock.c:95-152, 213-276) on TWO lines at once in packed 16-bit arithmetic -- what v_pk_{add,sub,max,min,
 * ashr,lshl}_i16 + 32-bit bit operations do on gfx950.  Not part of the product: tests/test_packed_edge_filter.py checks
 * it against the scalar formulation that the shipped kernel uses (edge_filter<> in edge264_amd/csrc/e264_kernels.hip).
 *
 * P2 is a pair of int16 lanes in one 32-bit word (lo = line A, hi = line B).  In this host build the operations are
 * emulated; on the device each pk_* is one instruction on `short __attribute__((ext_vector_type(2)))`.
 * Conditions become all-ones / all-zeros half-word masks: lt(a,b) = (a - b) >> 15 (operands are samples, thresholds
 * and small sums: no overflow), selections are (m & x) | (~m & y) = v_bfi_b32. */
#ifndef PACKED_EDGE_FILTER_H
#define PACKED_EDGE_FILTER_H
#include <stdint.h>

typedef uint32_t P2;
static inline P2 pk(int a, int b) { return (uint32_t)(uint16_t)(int16_t)a | (uint32_t)(uint16_t)(int16_t)b << 16; }
static inline int pk_lo(P2 v) { return (int16_t)(v & 0xffff); }
static inline int pk_hi(P2 v) { return (int16_t)(v >> 16); }
#define PK_OP2(name, expr) static inline P2 name(P2 x, P2 y) { int a = pk_lo(x), b = pk_lo(y), r0, r1; r0 = (expr); a = pk_hi(x); b = pk_hi(y); r1 = (expr); return pk(r0, r1); }
PK_OP2(pk_add, a + b)
PK_OP2(pk_sub, a - b)
PK_OP2(pk_max, a > b ? a : b)
PK_OP2(pk_min, a < b ? a : b)
static inline P2 pk_ashr(P2 x, int n) { return pk(pk_lo(x) >> n, pk_hi(x) >> n); }
static inline P2 pk_shl(P2 x, int n) { return pk(pk_lo(x) << n, pk_hi(x) << n); }
static inline P2 pk_splat(int v) { return pk(v, v); }
static inline P2 pk_abs(P2 x) { return pk_max(x, pk_sub(0, x)); }
static inline P2 pk_lt(P2 a, P2 b) { return pk_ashr(pk_sub(a, b), 15); }       /* a < b ? 0xffff : 0 per half */
static inline P2 pk_sel(P2 m, P2 x, P2 y) { return (m & x) | (~m & y); }       /* v_bfi_b32 */
static inline P2 pk_clip3(P2 lo, P2 hi, P2 v) { return pk_min(pk_max(v, lo), hi); }

/* One edge, two lines.  bS, tc0: per half (0..4 / table value); alpha, beta: the same for both lines of the lane (same
 * plane, same edge class); chroma: the lane filters chroma lines (only p0/q0 change, tc = tc0 + 1, no strong luma rule:
 * bS 4 uses the 2-tap chroma form).  Values p3..q3 are updated in place. */
static inline void edge_filter_p2(P2 *p3, P2 *p2, P2 *p1, P2 *p0, P2 *q0, P2 *q1, P2 *q2, P2 *q3,
	P2 bS, int alpha, int beta, P2 tc0, int chroma)
{
	const P2 A = pk_splat(alpha), B = pk_splat(beta), one = pk_splat(1), zero = 0, ones = 0xffffffffu, c255 = pk_splat(255);
	const P2 dpq = pk_abs(pk_sub(*p0, *q0));
	const P2 bs_nz = pk_lt(zero, bS);
	const P2 go = bs_nz & pk_lt(dpq, A) & pk_lt(pk_abs(pk_sub(*p1, *p0)), B) & pk_lt(pk_abs(pk_sub(*q1, *q0)), B);
	if (!go)
		return;                                   /* device: skipped when no line of the WAVE passes */
	const P2 lum = chroma ? zero : ones;
	const P2 ap = lum & pk_lt(pk_abs(pk_sub(*p2, *p0)), B), aq = lum & pk_lt(pk_abs(pk_sub(*q2, *q0)), B);
	/* bS < 4 */
	const P2 tc = chroma ? pk_add(tc0, one) : pk_add(tc0, pk_add(ap & one, aq & one));
	const P2 d0 = pk_ashr(pk_add(pk_add(pk_shl(pk_sub(*q0, *p0), 2), pk_sub(*p1, *q1)), pk_splat(4)), 3);
	const P2 delta = pk_clip3(pk_sub(zero, tc), tc, d0);
	const P2 avg = pk_ashr(pk_add(pk_add(*p0, *q0), one), 1);
	const P2 w_p0 = pk_clip3(zero, c255, pk_add(*p0, delta)), w_q0 = pk_clip3(zero, c255, pk_sub(*q0, delta));
	const P2 ntc0 = pk_sub(zero, tc0);
	const P2 w_p1 = pk_add(*p1, pk_clip3(ntc0, tc0, pk_ashr(pk_sub(pk_add(*p2, avg), pk_shl(*p1, 1)), 1)));
	const P2 w_q1 = pk_add(*q1, pk_clip3(ntc0, tc0, pk_ashr(pk_sub(pk_add(*q2, avg), pk_shl(*q1, 1)), 1)));
	/* bS == 4 */
	const P2 strong = pk_lt(pk_splat(3), bS);
	const P2 small = pk_lt(dpq, pk_add(pk_ashr(A, 2), pk_splat(2)));
	const P2 sp = ap & small, sq = aq & small;
	const P2 p1x2 = pk_shl(*p1, 1), q1x2 = pk_shl(*q1, 1), pq0 = pk_add(*p0, *q0);
	const P2 s_p0w = pk_ashr(pk_add(pk_add(pk_add(*p2, p1x2), pk_add(pk_shl(pq0, 1), *q1)), pk_splat(4)), 3);
	const P2 s_p0n = pk_ashr(pk_add(pk_add(p1x2, *p0), pk_add(*q1, pk_splat(2))), 2);
	const P2 s_p1 = pk_ashr(pk_add(pk_add(*p2, *p1), pk_add(pq0, pk_splat(2))), 2);
	const P2 s_p2 = pk_ashr(pk_add(pk_add(pk_shl(*p3, 1), pk_add(pk_shl(*p2, 1), *p2)), pk_add(pk_add(*p1, pq0), pk_splat(4))), 3);
	const P2 s_q0w = pk_ashr(pk_add(pk_add(pk_add(*q2, q1x2), pk_add(pk_shl(pq0, 1), *p1)), pk_splat(4)), 3);
	const P2 s_q0n = pk_ashr(pk_add(pk_add(q1x2, *q0), pk_add(*p1, pk_splat(2))), 2);
	const P2 s_q1 = pk_ashr(pk_add(pk_add(*q2, *q1), pk_add(pq0, pk_splat(2))), 2);
	const P2 s_q2 = pk_ashr(pk_add(pk_add(pk_shl(*q3, 1), pk_add(pk_shl(*q2, 1), *q2)), pk_add(pk_add(*q1, pq0), pk_splat(4))), 3);
	const P2 s_p0 = pk_sel(sp, s_p0w, s_p0n), s_q0 = pk_sel(sq, s_q0w, s_q0n);
	/* select per half: go ? (strong ? s : w) : old */
	const P2 n_p0 = pk_sel(strong, s_p0, w_p0), n_q0 = pk_sel(strong, s_q0, w_q0);
	const P2 m_p1 = pk_sel(strong, sp, ap), m_q1 = pk_sel(strong, sq, aq);
	const P2 n_p1 = pk_sel(strong, s_p1, w_p1), n_q1 = pk_sel(strong, s_q1, w_q1);
	*p0 = pk_sel(go, n_p0, *p0);
	*q0 = pk_sel(go, n_q0, *q0);
	*p1 = pk_sel(go & m_p1, n_p1, *p1);
	*q1 = pk_sel(go & m_q1, n_q1, *q1);
	*p2 = pk_sel(go & strong & sp, s_p2, *p2);
	*q2 = pk_sel(go & strong & sq, s_q2, *q2);
}
#endif
