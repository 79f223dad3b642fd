/* Host check of tools/proto/packed_edge_filter.h against the scalar edge filter (the formulation of edge_filter<> in
 * edge264_amd/csrc/e264_kernels.hip, itself bit-exact with the reference on the GPU tests).  Prints the mismatch count. */
#include <stdio.h>
#include <stdlib.h>
#include "packed_edge_filter.h"

static int clip3i(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static void edge_scalar(int *p3, int *p2, int *p1, int *p0, int *q0, int *q1, int *q2, int *q3, int bS, int alpha, int beta, int tc0, int chroma)
{
	const int dpq = abs(*p0 - *q0);
	if (!(bS != 0 && dpq < alpha && abs(*p1 - *p0) < beta && abs(*q1 - *q0) < beta))
		return;
	const int ap = !chroma && abs(*p2 - *p0) < beta, aq = !chroma && abs(*q2 - *q0) < beta;
	if (bS < 4) {
		const int tc = tc0 + (chroma ? 1 : ap + aq);
		const int delta = clip3i(-tc, tc, (((*q0 - *p0) * 4) + (*p1 - *q1) + 4) >> 3);
		const int avg = (*p0 + *q0 + 1) >> 1;
		const int w_p1 = *p1 + clip3i(-tc0, tc0, (*p2 + avg - 2 * *p1) >> 1), w_q1 = *q1 + clip3i(-tc0, tc0, (*q2 + avg - 2 * *q1) >> 1);
		const int w_p0 = clip3i(0, 255, *p0 + delta), w_q0 = clip3i(0, 255, *q0 - delta);
		*p0 = w_p0; *q0 = w_q0;
		if (ap) *p1 = w_p1;
		if (aq) *q1 = w_q1;
	} else {
		const int small = dpq < (alpha >> 2) + 2, sp = ap && small, sq = aq && small;
		const int P3 = *p3, P2_ = *p2, P1 = *p1, P0 = *p0, Q0 = *q0, Q1 = *q1, Q2 = *q2, Q3 = *q3;
		*p0 = sp ? (P2_ + 2 * P1 + 2 * P0 + 2 * Q0 + Q1 + 4) >> 3 : (2 * P1 + P0 + Q1 + 2) >> 2;
		*q0 = sq ? (P1 + 2 * P0 + 2 * Q0 + 2 * Q1 + Q2 + 4) >> 3 : (2 * Q1 + Q0 + P1 + 2) >> 2;
		if (sp) { *p1 = (P2_ + P1 + P0 + Q0 + 2) >> 2; *p2 = (2 * P3 + 3 * P2_ + P1 + P0 + Q0 + 4) >> 3; }
		if (sq) { *q1 = (P0 + Q0 + Q1 + Q2 + 2) >> 2; *q2 = (2 * Q3 + 3 * Q2 + Q1 + Q0 + P0 + 4) >> 3; }
	}
}

static uint32_t rs = 12345;
static int rnd(int n) { rs = rs * 1664525u + 1013904223u; return (int)((rs >> 8) % (uint32_t)n); }
/* samples: mostly near a common level so that the thresholds are actually passed */
static int sample(int base, int spread) { int v = base + rnd(2 * spread + 1) - spread; return v < 0 ? 0 : v > 255 ? 255 : v; }

int main(int argc, char **argv)
{
	long n = argc > 1 ? atol(argv[1]) : 2000000, bad = 0, changed = 0;
	for (long it = 0; it < n; it++) {
		int v[2][8], r[2][8], bS[2], tc0[2];
		const int alpha = rnd(4) ? rnd(256) : rnd(8), beta = rnd(19), chroma = rnd(3) == 0;
		for (int h = 0; h < 2; h++) {
			const int base = rnd(256), spread = rnd(4) == 0 ? 255 : 1 + rnd(12);
			for (int i = 0; i < 8; i++) r[h][i] = v[h][i] = sample(base, spread);
			bS[h] = rnd(5);
			tc0[h] = bS[h] == 4 ? 0 : rnd(26);   /* the kernel masks tC0 to 0 for bS 4 (no table row) */
			edge_scalar(&r[h][0], &r[h][1], &r[h][2], &r[h][3], &r[h][4], &r[h][5], &r[h][6], &r[h][7], bS[h], alpha, beta, tc0[h], chroma);
		}
		P2 x[8];
		for (int i = 0; i < 8; i++) x[i] = pk(v[0][i], v[1][i]);
		edge_filter_p2(&x[0], &x[1], &x[2], &x[3], &x[4], &x[5], &x[6], &x[7], pk(bS[0], bS[1]), alpha, beta, pk(tc0[0], tc0[1]), chroma);
		for (int i = 0; i < 8; i++) {
			if (pk_lo(x[i]) != r[0][i] || pk_hi(x[i]) != r[1][i]) { bad++; break; }
		}
		for (int i = 0; i < 8; i++) changed += r[0][i] != v[0][i];
	}
	printf("%ld %ld %ld\n", n, bad, changed);
	return bad != 0;
}
