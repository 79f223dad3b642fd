/*
 * e264_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Scalar CPU restatement of edge264's sample-reconstruction path (residual,
 * intra, inter, deblock) driven by the command packet of include/edge264_cmd.h.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call this file; the product (edge264_amd/csrc) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py, tests/test_oracle_vs_refkernels.py and
 * tests/test_frontend_capture.py check every function below against (a) the golden vectors of the reference's own unit
 * tests (src/edge264_check.c:169-359, lifted by tests/golden/extract_check_vectors.py),
 * (b) the reference's static kernels compiled from /root/reference by
 * oracle/Makefile into oracle/_ref/libe264_refkernels.so, on randomised inputs,
 * and (c) whole .264 streams decoded by the unmodified reference library.
 *
 * All file:line citations are relative to /root/reference/src/.
 * Where the reference's x86 SIMD arithmetic deviates from the H.264 text on
 * non-conformant input (int16 wrap in the 8x8 IDCT and in the 2-D six-tap,
 * int8 truncation of weights, int16 saturation of weighted sums) this file
 * follows the REFERENCE (its SSE build), because "identical output to the
 * reference" is the bar.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/edge264_cmd.h"

#define EXPORT __attribute__((visibility("default")))

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static inline int clip255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static inline int sat16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
static inline int16_t w16(int v) { return (int16_t)v; } /* int16 wraparound */

/* zig-zag 4x4 block index -> pixel offsets (edge264_internal.h:550-553 x444/y444) */
static const uint8_t BX[16] = {0, 4, 0, 4, 8, 12, 8, 12, 0, 4, 0, 4, 8, 12, 8, 12};
static const uint8_t BY[16] = {0, 0, 4, 4, 0, 0, 4, 4, 8, 8, 12, 12, 8, 8, 12, 12};

/* ------------------------------------------------------------------------ */
/* Residual (edge264_residual.c)                                             */
/* ------------------------------------------------------------------------ */

/* Table 8-?? normAdjust4x4 (edge264_residual.c:77-84): v[qP%6][class of (i,j)] */
static int norm_adjust4x4(int m, int pos)
{
	static const uint8_t v[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
	int i = pos >> 2, j = pos & 3;
	return v[m][(i & 1) && (j & 1) ? 1 : !(i & 1) && !(j & 1) ? 0 : 2];
}

/* normAdjust8x8 (edge264_residual.c:85-98) */
static int norm_adjust8x8(int m, int pos)
{
	static const uint8_t v[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
		{28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
	int i = pos >> 3, j = pos & 7;
	int k;
	if ((i & 3) == 0 && (j & 3) == 0) k = 0;
	else if ((i & 1) && (j & 1)) k = 1;
	else if ((i & 3) == 2 && (j & 3) == 2) k = 2;
	else if (((i & 3) == 0 && (j & 1)) || ((i & 1) && (j & 3) == 0)) k = 3;
	else if (((i & 3) == 0 && (j & 3) == 2) || ((i & 3) == 2 && (j & 3) == 0)) k = 4;
	else k = 5;
	return v[m][k];
}

/* add one 4x4 residual row-major r[16] to dst with the reference's clipping:
 * residual saturated to int16 (shrps32, residual.c:157), int16 wrap add, packus (residual.c:166) */
static void add_res4x4(uint8_t *dst, int stride, const int *r)
{
	for (int y = 0; y < 4; y++)
		for (int x = 0; x < 4; x++)
			dst[y * stride + x] = (uint8_t)clip255(w16(dst[y * stride + x] + sat16(r[y * 4 + x])));
}

/* add_idct4x4, edge264_residual.c:108-172.  c[] in transposed order c[x*4+y].
 * has_dc: c[0] is replaced AFTER scaling by the pre-scaled dc (residual.c:123-124). */
EXPORT void e264o_add_idct4x4(const int16_t *c, int qP, const uint8_t *wS, int has_dc, int dc, uint8_t *dst, int stride)
{
	int d[16], f[16], r[16];
	int sh = qP / 6, m = qP % 6;
	for (int i = 0; i < 16; i++) {
		int32_t LS = wS[i] * norm_adjust4x4(m, i);
		d[i] = (int32_t)(((uint32_t)((int32_t)c[i] * LS) << sh) + 8) >> 4; /* residual.c:118-121 */
	}
	if (has_dc)
		d[0] = dc;
	/* first pass over index i = x (vectors d0..d3 = c[0..3], c[4..7], ...), residual.c:127-134 */
	for (int j = 0; j < 4; j++) {
		int d0 = d[j], d1 = d[4 + j], d2 = d[8 + j], d3 = d[12 + j];
		int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = (d3 >> 1) + d1;
		/* f[x'][y=j] */
		f[0 * 4 + j] = e0 + e3;
		f[1 * 4 + j] = e1 + e2;
		f[2 * 4 + j] = e1 - e2;
		f[3 * 4 + j] = e0 - e3;
	}
	/* after transposition vectors are indexed by y with lanes x'; +32 on y=0 (residual.c:141) */
	for (int x = 0; x < 4; x++) {
		int f0 = f[x * 4 + 0] + 32, f1 = f[x * 4 + 1], f2 = f[x * 4 + 2], f3 = f[x * 4 + 3];
		int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = (f3 >> 1) + f1;
		r[0 * 4 + x] = (g0 + g3) >> 6;
		r[1 * 4 + x] = (g1 + g2) >> 6;
		r[2 * 4 + x] = (g1 - g2) >> 6;
		r[3 * 4 + x] = (g0 - g3) >> 6;
	}
	add_res4x4(dst, stride, r);
}

/* add_dc4x4, edge264_residual.c:174-187 */
EXPORT void e264o_add_dc4x4(int dc, uint8_t *dst, int stride)
{
	int r[16];
	for (int i = 0; i < 16; i++)
		r[i] = w16((dc + 32) >> 6); /* set16() truncates to int16 */
	for (int y = 0; y < 4; y++)
		for (int x = 0; x < 4; x++)
			dst[y * stride + x] = (uint8_t)clip255(w16(dst[y * stride + x] + r[0]));
}

/* add_idct8x8, edge264_residual.c:194-343 (int16 arithmetic with wraparound) */
EXPORT void e264o_add_idct8x8(const int16_t *c, int qP, const uint8_t *wS, uint8_t *dst, int stride)
{
	int16_t d[8][8]; /* d[i][j]: vector i, lane j  (i = x, j = y before the first pass) */
	int div = qP / 6, m = qP % 6;
	for (int i = 0; i < 8; i++) {
		for (int j = 0; j < 8; j++) {
			int pos = i * 8 + j;
			int LS = wS[pos] * norm_adjust8x8(m, pos); /* u16, residual.c:204-211 */
			if (div < 6) { /* scale32, residual.c:13-17, 214-235 */
				int32_t v = (int32_t)c[pos] * LS;
				d[i][j] = (int16_t)sat16((v + (1 << (5 - div))) >> (6 - div));
			} else { /* residual.c:236-247: packs32(c) * (LS << sh) in int16 */
				d[i][j] = w16((int32_t)c[pos] * (int32_t)w16(LS << (div - 6)));
			}
		}
	}
	for (int pass = 0; pass < 2; pass++) {
		int16_t o[8][8];
		for (int j = 0; j < 8; j++) {
			int16_t d0 = d[0][j], d1 = d[1][j], d2 = d[2][j], d3 = d[3][j], d4 = d[4][j], d5 = d[5][j], d6 = d[6][j], d7 = d[7][j];
			int16_t e0 = w16(d0 + d4);
			int16_t e1 = w16(d5 - d3 - w16((d7 >> 1) + d7));
			int16_t e2 = w16(d0 - d4);
			int16_t e3 = w16(d1 + d7 - w16((d3 >> 1) + d3));
			int16_t e4 = w16((d2 >> 1) - d6);
			int16_t e5 = w16(d7 - d1 + w16((d5 >> 1) + d5));
			int16_t e6 = w16((d6 >> 1) + d2);
			int16_t e7 = w16(d3 + d5 + w16((d1 >> 1) + d1));
			int16_t f0 = w16(e0 + e6);
			int16_t f1 = w16((e7 >> 2) + e1);
			int16_t f2 = w16(e2 + e4);
			int16_t f3 = w16((e5 >> 2) + e3);
			int16_t f4 = w16(e2 - e4);
			int16_t f5 = w16((e3 >> 2) - e5);
			int16_t f6 = w16(e0 - e6);
			int16_t f7 = w16(e7 - (e1 >> 2));
			o[0][j] = w16(f0 + f7);
			o[1][j] = w16(f2 + f5);
			o[2][j] = w16(f4 + f3);
			o[3][j] = w16(f6 + f1);
			o[4][j] = w16(f6 - f1);
			o[5][j] = w16(f4 - f3);
			o[6][j] = w16(f2 - f5);
			o[7][j] = w16(f0 - f7);
		}
		if (pass == 0) {
			/* transpose, +32 on the new vector 0 (residual.c:298) */
			for (int i = 0; i < 8; i++)
				for (int j = 0; j < 8; j++)
					d[j][i] = o[i][j];
			for (int j = 0; j < 8; j++)
				d[0][j] = w16(d[0][j] + 32);
		} else {
			/* vector i is now row i, lane j is column j (residual.c:309-342) */
			for (int i = 0; i < 8; i++)
				for (int j = 0; j < 8; j++)
					dst[i * stride + j] = (uint8_t)clip255(w16(dst[i * stride + j] + (o[i][j] >> 6)));
		}
	}
}

/* transform_dc4x4, edge264_residual.c:352-454. in: c[16] levels (c_v[0..3]).
 * out: dc[k] for 4x4 block k in zig order (what the reference leaves in c[16+k]). */
EXPORT void e264o_transform_dc4x4(const int16_t *c, int qP, int wS0, int32_t *dc)
{
	int32_t x4[4], x5[4], x6[4], x7[4], f[4][4];
	for (int l = 0; l < 4; l++) { /* lane l; vectors c_v[k] = c[4k..4k+3] */
		int x0 = c[0 + l] + c[4 + l], x1 = c[8 + l] + c[12 + l], x2 = c[0 + l] - c[4 + l], x3 = c[8 + l] - c[12 + l];
		x4[l] = x0 + x1; x5[l] = x0 - x1; x6[l] = x2 - x3; x7[l] = x2 + x3;
	}
	/* transpose: xC = {x4[0],x5[0],x6[0],x7[0]}, xD = lanes 1, xE = lanes 2, xF = lanes 3 (residual.c:365-373) */
	const int32_t *t[4] = {x4, x5, x6, x7};
	for (int l = 0; l < 4; l++) {
		int xC = t[l][0], xD = t[l][1], xE = t[l][2], xF = t[l][3];
		int xG = xC + xD, xH = xE + xF, xI = xC - xD, xJ = xE - xF;
		f[0][l] = xG + xH; f[1][l] = xG - xH; f[2][l] = xI - xJ; f[3][l] = xI + xJ;
	}
	int32_t LS = (wS0 * norm_adjust4x4(qP % 6, 0)) << (qP / 6); /* residual.c:388 */
	/* dc_r[l] belongs to block (bx = l, by = r); stored zig-zag (residual.c:395-399) */
	for (int r = 0; r < 4; r++)
		for (int l = 0; l < 4; l++) {
			int k = (r >> 1) * 8 + (l >> 1) * 4 + (r & 1) * 2 + (l & 1);
			dc[k] = (int32_t)((uint32_t)f[r][l] * (uint32_t)LS + 32) >> 6;
		}
}

/* transform_dc2x2, edge264_residual.c:456-538. in c[8] (Cb at 0,2,4,6 / Cr at 1,3,5,7 in
 * the order written by scan {0,4,2,6}/{1,5,3,7}, slice.c:448-451). out dc[0..3]=Cb, dc[4..7]=Cr */
EXPORT void e264o_transform_dc2x2(const int16_t *c, int qPb, int qPr, int wSb, int wSr, int32_t *dc)
{
	int32_t d0[4], d1[4], e0[4], e1[4], f0[4], f1[4];
	for (int l = 0; l < 4; l++) { d0[l] = c[l] + c[4 + l]; d1[l] = c[l] - c[4 + l]; }
	e0[0] = d0[0]; e0[1] = d0[1]; e0[2] = d1[0]; e0[3] = d1[1];
	e1[0] = d0[2]; e1[1] = d0[3]; e1[2] = d1[2]; e1[3] = d1[3];
	for (int l = 0; l < 4; l++) { f0[l] = e0[l] + e1[l]; f1[l] = e0[l] - e1[l]; }
	int32_t LSb = (wSb * norm_adjust4x4(qPb % 6, 0)) << (qPb / 6);
	int32_t LSr = (wSr * norm_adjust4x4(qPr % 6, 0)) << (qPr / 6);
	int32_t cb[4] = {f0[0], f0[2], f1[0], f1[2]}, cr[4] = {f0[1], f0[3], f1[1], f1[3]};
	for (int l = 0; l < 4; l++) {
		dc[l] = (int32_t)((uint32_t)cb[l] * (uint32_t)LSb) >> 5;
		dc[4 + l] = (int32_t)((uint32_t)cr[l] * (uint32_t)LSr) >> 5;
	}
}

/* ------------------------------------------------------------------------ */
/* Intra prediction (edge264_intra.c); internal mode numbers of               */
/* edge264_internal.h:564-634.  Suffix letters name UNAVAILABLE neighbours:   */
/* A left, B top, C top-right, D top-left.                                    */
/* ------------------------------------------------------------------------ */
#define LP(l, m, r) (((l) + 2 * (m) + (r) + 2) >> 2) /* intra.c:31,54 lowpass */

EXPORT void e264o_intra4x4(uint8_t *p, int stride, int mode)
{
	/* neighbours: t[-1..7] top row (t[-1] = corner), l[0..3] left column */
	int T[9], *t = T + 1, l[4];
	const uint8_t *pT = p - stride;
	int need_top = 0, need_tr = 0, need_left = 0, need_corner = 0;
	switch (mode) {
	case 0: case 3: need_top = 1; break;                          /* V, DC_A */
	case 1: case 4: case 13: need_left = 1; break;                /* H, DC_B, HU */
	case 2: need_top = need_left = 1; break;                      /* DC */
	case 5: break;                                                /* DC_AB */
	case 6: case 11: need_top = need_tr = 1; break;               /* DDL, VL */
	case 7: case 12: need_top = 1; break;                         /* DDL_C, VL_C */
	case 8: case 9: case 10: need_top = need_left = need_corner = 1; break; /* DDR, VR, HD */
	}
	for (int i = 0; i < 9; i++) T[i] = 0;
	for (int i = 0; i < 4; i++) l[i] = 0;
	if (need_top) for (int i = 0; i < 4; i++) t[i] = pT[i];
	if (need_tr) for (int i = 4; i < 8; i++) t[i] = pT[i];
	else if (need_top) for (int i = 4; i < 8; i++) t[i] = pT[3]; /* _C variants replicate (intra.c:341-343,360-362) */
	if (need_left) for (int i = 0; i < 4; i++) l[i] = p[i * stride - 1];
	if (need_corner) t[-1] = pT[-1];
	int out[4][4];
	switch (mode) {
	default:
	case 0: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) out[y][x] = t[x]; break;
	case 1: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) out[y][x] = l[y]; break;
	case 2: { int s = (t[0] + t[1] + t[2] + t[3] + l[0] + l[1] + l[2] + l[3] + 4) >> 3;
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) out[y][x] = s; } break;
	case 3: { int s = (t[0] + t[1] + t[2] + t[3] + 2) >> 2;
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) out[y][x] = s; } break;
	case 4: { int s = (l[0] + l[1] + l[2] + l[3] + 2) >> 2;
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) out[y][x] = s; } break;
	case 5: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) out[y][x] = 128; break;
	case 6: case 7: /* diagonal down-left, 8.3.1.2.4 */
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++)
			out[y][x] = (x == 3 && y == 3) ? (t[6] + 3 * t[7] + 2) >> 2 : LP(t[x + y], t[x + y + 1], t[x + y + 2]);
		break;
	case 8: /* diagonal down-right, 8.3.1.2.5 */
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
			if (x > y) out[y][x] = LP(t[x - y - 2], t[x - y - 1], t[x - y]);
			else if (x < y) out[y][x] = LP(y - x - 2 < 0 ? t[-1] : l[y - x - 2], l[y - x - 1], l[y - x]);
			else out[y][x] = LP(t[0], t[-1], l[0]);
		}
		break;
	case 9: /* vertical right, 8.3.1.2.6 */
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
			int z = 2 * x - y;
			if (z >= 0 && !(z & 1)) out[y][x] = (t[x - (y >> 1) - 1] + t[x - (y >> 1)] + 1) >> 1;
			else if (z >= 0) out[y][x] = LP(t[x - (y >> 1) - 2], t[x - (y >> 1) - 1], t[x - (y >> 1)]);
			else if (z == -1) out[y][x] = LP(l[0], t[-1], t[0]);
			else out[y][x] = LP(l[y - 1], l[y - 2], y - 3 < 0 ? t[-1] : l[y - 3]);
		}
		break;
	case 10: /* horizontal down, 8.3.1.2.7 */
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
			int z = 2 * y - x;
			#define L_(i) ((i) < 0 ? t[-1] : l[i])
			if (z >= 0 && !(z & 1)) out[y][x] = (L_(y - (x >> 1) - 1) + L_(y - (x >> 1)) + 1) >> 1;
			else if (z >= 0) out[y][x] = LP(L_(y - (x >> 1) - 2), L_(y - (x >> 1) - 1), L_(y - (x >> 1)));
			else if (z == -1) out[y][x] = LP(l[0], t[-1], t[0]);
			else out[y][x] = LP(t[x - 1], t[x - 2], t[x - 3]);
			#undef L_
		}
		break;
	case 11: case 12: /* vertical left, 8.3.1.2.8 */
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
			int i = x + (y >> 1);
			out[y][x] = (y & 1) ? LP(t[i], t[i + 1], t[i + 2]) : (t[i] + t[i + 1] + 1) >> 1;
		}
		break;
	case 13: /* horizontal up, 8.3.1.2.9 */
		for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
			int z = x + 2 * y;
			if (z > 5) out[y][x] = l[3];
			else if (z == 5) out[y][x] = (l[2] + 3 * l[3] + 2) >> 2;
			else if (z & 1) out[y][x] = LP(l[y + (x >> 1)], l[y + (x >> 1) + 1], l[y + (x >> 1) + 2]);
			else out[y][x] = (l[y + (x >> 1)] + l[y + (x >> 1) + 1] + 1) >> 1;
		}
		break;
	}
	for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) p[y * stride + x] = (uint8_t)out[y][x];
}

/* Intra 8x8 (intra.c:383-616).  Decomposes the 32 internal modes into the spec
 * mode (8.3.2.2.2-10) plus the availability of A (left), B (top), C (top-right),
 * D (top-left), applies the reference-sample filter of 8.3.2.2.1, then predicts. */
EXPORT void e264o_intra8x8(uint8_t *p, int stride, int mode)
{
	/* spec mode: 0 V,1 H,2 DC,3 DDL,4 DDR,5 VR,6 HD,7 VL,8 HU */
	static const int8_t spec[32] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 5, 6, 7, 7, 7, 7, 8, 8};
	/* bit0 A unavailable, bit1 B, bit2 C, bit3 D */
	static const int8_t unav[32] = {0, 4, 8, 12, 0, 8, 0, 1, 5, 9, 13, 2, 10, 4, 8, 12, 3, 0, 4, 8, 12, 0, 4, 0, 4, 0, 0, 4, 8, 12, 0, 8};
	int sm = spec[mode], un = unav[mode];
	/* Which neighbours each spec mode reads; unread ones are treated as unavailable so
	 * that no out-of-frame sample is touched (the reference does the same through its
	 * per-mode loads). */
	int useA = !(un & 1) && (sm == 1 || sm == 2 || sm == 4 || sm == 5 || sm == 6 || sm == 8);
	int useB = !(un & 2) && (sm == 0 || sm == 2 || sm == 3 || sm == 4 || sm == 5 || sm == 6 || sm == 7);
	int useC = useB && !(un & 4);
	int useD = !(un & 8);
	if (mode == 16) { /* DC_AB */
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) p[y * stride + x] = 128;
		return;
	}
	const uint8_t *pT = p - stride;
	int rawT[17], *t = rawT + 1, rawL[8]; /* t[-1..15], l[0..7] */
	int fT[17], *ft = fT + 1, fl[8], fcorner = 0;
	for (int i = 0; i < 17; i++) rawT[i] = fT[i] = 0;
	for (int i = 0; i < 8; i++) rawL[i] = fl[i] = 0;
	if (useB) {
		for (int i = 0; i < 8; i++) t[i] = pT[i];
		if (useC && sm != 6) /* horizontal-down never looks past t[7] (no _C variant exists, internal.h:606) */
			for (int i = 8; i < 16; i++) t[i] = pT[i];
		else
			for (int i = 8; i < 16; i++) t[i] = pT[7];
	}
	if (useA) for (int i = 0; i < 8; i++) rawL[i] = p[i * stride - 1];
	/* the corner is only dereferenced when D is available AND something needs it */
	int cornerAvail = useD && (useA || useB);
	if (cornerAvail) t[-1] = pT[-1];
	/* 8.3.2.2.1 filtering */
	if (useB) {
		ft[0] = cornerAvail ? LP(t[-1], t[0], t[1]) : (3 * t[0] + t[1] + 2) >> 2;
		for (int i = 1; i < 15; i++) ft[i] = LP(t[i - 1], t[i], t[i + 1]);
		ft[15] = (t[14] + 3 * t[15] + 2) >> 2;
	}
	if (cornerAvail) {
		if (!useB) fcorner = (3 * t[-1] + rawL[0] + 2) >> 2;
		else if (!useA) fcorner = (3 * t[-1] + t[0] + 2) >> 2;
		else fcorner = LP(t[0], t[-1], rawL[0]);
	}
	if (useA) {
		fl[0] = cornerAvail ? LP(t[-1], rawL[0], rawL[1]) : (3 * rawL[0] + rawL[1] + 2) >> 2;
		for (int i = 1; i < 7; i++) fl[i] = LP(rawL[i - 1], rawL[i], rawL[i + 1]);
		fl[7] = (rawL[6] + 3 * rawL[7] + 2) >> 2;
	}
	ft[-1] = fcorner;
	int out[8][8];
	#define FL(i) ((i) < 0 ? fcorner : fl[i])
	switch (sm) {
	default:
	case 0: for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y][x] = ft[x]; break;
	case 1: for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y][x] = fl[y]; break;
	case 2: {
		int s = 0;
		if (useA && useB) { for (int i = 0; i < 8; i++) s += ft[i] + fl[i]; s = (s + 8) >> 4; }
		else if (useB) { for (int i = 0; i < 8; i++) s += ft[i]; s = (s + 4) >> 3; }
		else if (useA) { for (int i = 0; i < 8; i++) s += fl[i]; s = (s + 4) >> 3; }
		else s = 128;
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y][x] = s;
		} break;
	case 3:
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++)
			out[y][x] = (x == 7 && y == 7) ? (ft[14] + 3 * ft[15] + 2) >> 2 : LP(ft[x + y], ft[x + y + 1], ft[x + y + 2]);
		break;
	case 4:
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
			if (x > y) out[y][x] = LP(ft[x - y - 2], ft[x - y - 1], ft[x - y]);
			else if (x < y) out[y][x] = LP(FL(y - x - 2), FL(y - x - 1), FL(y - x));
			else out[y][x] = LP(ft[0], fcorner, fl[0]);
		}
		break;
	case 5:
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
			int z = 2 * x - y, i = x - (y >> 1);
			if (z >= 0 && !(z & 1)) out[y][x] = (ft[i - 1] + ft[i] + 1) >> 1;
			else if (z >= 0) out[y][x] = LP(ft[i - 2], ft[i - 1], ft[i]);
			else if (z == -1) out[y][x] = LP(fl[0], fcorner, ft[0]);
			else out[y][x] = LP(FL(y - 2 * x - 1), FL(y - 2 * x - 2), FL(y - 2 * x - 3));
		}
		break;
	case 6:
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
			int z = 2 * y - x, i = y - (x >> 1);
			if (z >= 0 && !(z & 1)) out[y][x] = (FL(i - 1) + FL(i) + 1) >> 1;
			else if (z >= 0) out[y][x] = LP(FL(i - 2), FL(i - 1), FL(i));
			else if (z == -1) out[y][x] = LP(fl[0], fcorner, ft[0]);
			else out[y][x] = LP(ft[x - 2 * y - 1], ft[x - 2 * y - 2], ft[x - 2 * y - 3]);
		}
		break;
	case 7:
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
			int i = x + (y >> 1);
			out[y][x] = (y & 1) ? LP(ft[i], ft[i + 1], ft[i + 2]) : (ft[i] + ft[i + 1] + 1) >> 1;
		}
		break;
	case 8:
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
			int z = x + 2 * y, i = y + (x >> 1);
			if (z > 13) out[y][x] = fl[7];
			else if (z == 13) out[y][x] = (fl[6] + 3 * fl[7] + 2) >> 2;
			else if (z & 1) out[y][x] = LP(fl[i], fl[i + 1], fl[i + 2]);
			else out[y][x] = (fl[i] + fl[i + 1] + 1) >> 1;
		}
		break;
	}
	#undef FL
	for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) p[y * stride + x] = (uint8_t)out[y][x];
}

/* Intra 16x16, intra.c:623-682 */
EXPORT void e264o_intra16x16(uint8_t *p, int stride, int mode)
{
	const uint8_t *pT = p - stride;
	int out = 128;
	switch (mode) {
	case 0: for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) p[y * stride + x] = pT[x]; return;
	case 1: for (int y = 0; y < 16; y++) { int v = p[y * stride - 1]; for (int x = 0; x < 16; x++) p[y * stride + x] = (uint8_t)v; } return;
	case 2: { int s = 0; for (int i = 0; i < 16; i++) s += pT[i] + p[i * stride - 1]; out = (s + 16) >> 5; } break;
	case 3: { int s = 0; for (int i = 0; i < 16; i++) s += pT[i]; out = (s + 8) >> 4; } break;
	case 4: { int s = 0; for (int i = 0; i < 16; i++) s += p[i * stride - 1]; out = (s + 8) >> 4; } break;
	case 5: out = 128; break;
	case 6: { /* plane, 8.3.3.4; intra.c:659-678 */
		int H = 0, V = 0;
		for (int i = 0; i < 8; i++) {
			H += (i + 1) * (pT[8 + i] - (i == 7 ? pT[-1] : pT[6 - i]));
			V += (i + 1) * (p[(8 + i) * stride - 1] - (i == 7 ? pT[-1] : p[(6 - i) * stride - 1]));
		}
		int a = 16 * (p[15 * stride - 1] + pT[15]), b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
		int tmp[16][16];
		for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++)
			tmp[y][x] = clip255((a + b * (x - 7) + c * (y - 7) + 16) >> 5);
		for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) p[y * stride + x] = (uint8_t)tmp[y][x];
		} return;
	}
	for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) p[y * stride + x] = (uint8_t)out;
}

/* Intra chroma for ONE plane (the reference does Cb+Cr in one call with a halved
 * stride, intra.c:689-765); p = top-left of the 8x8 block, stride = true row stride. */
EXPORT void e264o_intra_chroma(uint8_t *p, int stride, int mode)
{
	const uint8_t *pT = p - stride;
	int out[8][8];
	switch (mode) {
	default:
	case 0: case 1: case 2: case 3: {
		int dc[2][2]; /* [by][bx] */
		if (mode == 3) dc[0][0] = dc[0][1] = dc[1][0] = dc[1][1] = 128;
		else {
			int t0 = 0, t1 = 0, l0 = 0, l1 = 0;
			if (mode != 2) for (int i = 0; i < 4; i++) { t0 += pT[i]; t1 += pT[4 + i]; }
			if (mode != 1) for (int i = 0; i < 4; i++) { l0 += p[i * stride - 1]; l1 += p[(4 + i) * stride - 1]; }
			if (mode == 0) {
				dc[0][0] = (t0 + l0 + 4) >> 3; dc[0][1] = (t1 + 2) >> 2;
				dc[1][0] = (l1 + 2) >> 2; dc[1][1] = (t1 + l1 + 4) >> 3;
			} else if (mode == 1) { /* left unavailable (intra.c:716-719) */
				dc[0][0] = dc[1][0] = (t0 + 2) >> 2; dc[0][1] = dc[1][1] = (t1 + 2) >> 2;
			} else { /* top unavailable */
				dc[0][0] = dc[0][1] = (l0 + 2) >> 2; dc[1][0] = dc[1][1] = (l1 + 2) >> 2;
			}
		}
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y][x] = dc[y >> 2][x >> 2];
		} break;
	case 4: for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y][x] = p[y * stride - 1]; break;
	case 5: for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y][x] = pT[x]; break;
	case 6: { /* plane 8.3.4.4, intra.c:733-757 */
		int H = 0, V = 0;
		for (int i = 0; i < 4; i++) {
			H += (i + 1) * (pT[4 + i] - (i == 3 ? pT[-1] : pT[2 - i]));
			V += (i + 1) * (p[(4 + i) * stride - 1] - (i == 3 ? pT[-1] : p[(2 - i) * stride - 1]));
		}
		int a = 16 * (p[7 * stride - 1] + pT[7]), b = (34 * H + 32) >> 6, c = (34 * V + 32) >> 6;
		for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++)
			out[y][x] = clip255((a + b * (x - 3) + c * (y - 3) + 16) >> 5);
		} break;
	}
	for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) p[y * stride + x] = (uint8_t)out[y][x];
}

/* ------------------------------------------------------------------------ */
/* Inter prediction (edge264_inter.c)                                         */
/* ------------------------------------------------------------------------ */
typedef struct { const uint8_t *p; int stride, w, h; } Plane;
static inline int PX(const Plane *r, int x, int y)
{ /* edge emulation == per-sample clamp (inter.c:1199-1235) */
	return r->p[clip3(0, r->h - 1, y) * r->stride + clip3(0, r->w - 1, x)];
}
static inline int tapH(const Plane *r, int x, int y)
{ return PX(r, x - 2, y) - 5 * PX(r, x - 1, y) + 20 * PX(r, x, y) + 20 * PX(r, x + 1, y) - 5 * PX(r, x + 2, y) + PX(r, x + 3, y); }
static inline int tapV(const Plane *r, int x, int y)
{ return PX(r, x, y - 2) - 5 * PX(r, x, y - 1) + 20 * PX(r, x, y) + 20 * PX(r, x, y + 1) - 5 * PX(r, x, y + 2) + PX(r, x, y + 3); }
/* sixtapHV on int16 intermediates with wraparound, inter.c:4-9; then (+32)>>6 packus (shrrpus16, inter.c:14) */
static inline int centre(const int *t)
{
	int16_t af = w16(t[0] + t[5]), be = w16(t[1] + t[4]), cd = w16(t[2] + t[3]);
	int16_t x1 = w16(af - be);
	int16_t x2 = w16((x1 >> 2) + w16(cd - be));
	int16_t x3 = w16((x2 >> 2) + cd);
	return clip255(w16(x3 + 32) >> 6);
}
static inline int avg(int a, int b) { return (a + b + 1) >> 1; }

/* one luma sample at integer position (x,y) + quarter offsets (xF,yF): 8.4.2.2.1 as
 * organised by decode_inter_luma (inter.c:416-968) */
static int luma_sample(const Plane *r, int x, int y, int xF, int yF)
{
	#define B_(xx, yy) clip255((tapH(r, xx, yy) + 16) >> 5)
	#define H_(xx, yy) clip255((tapV(r, xx, yy) + 16) >> 5)
	if (yF == 0) {
		if (xF == 0) return PX(r, x, y);
		int b = B_(x, y);
		return xF == 2 ? b : avg(PX(r, x + (xF == 3), y), b);
	}
	if (xF == 0) {
		int h = H_(x, y);
		return yF == 2 ? h : avg(PX(r, x, y + (yF == 3)), h);
	}
	if ((xF & 1) && (yF & 1)) /* e,g,p,r: inter.c:510-557 */
		return avg(B_(x, y + (yF == 3)), H_(x + (xF == 3), y));
	int t[6];
	if (xF == 2) { /* horizontal first then vertical (inter.c:611-646, 779-802, 929-966) */
		for (int k = 0; k < 6; k++) t[k] = tapH(r, x, y - 2 + k);
		int j = centre(t);
		return yF == 2 ? j : avg(j, B_(x, y + (yF == 3)));
	}
	/* xF odd, yF == 2: vertical first then horizontal (inter.c:559-609, 741-777, 887-927) */
	for (int k = 0; k < 6; k++) t[k] = tapV(r, x - 2 + k, y);
	return avg(centre(t), H_(x + (xF == 3), y));
	#undef B_
	#undef H_
}

static int chroma_sample(const Plane *r, int x, int y, int xF, int yF)
{ /* 8.4.2.2.2; inter.c:977-1091, ABCD at inter.c:1242 */
	int A = (8 - xF) * (8 - yF), B = xF * (8 - yF), C = (8 - xF) * yF, D = xF * yF;
	return (A * PX(r, x, y) + B * PX(r, x + 1, y) + C * PX(r, x, y + 1) + D * PX(r, x + 1, y + 1) + 32) >> 6;
}

/* weights of one prediction call {w0,w1,o,wd}; maddshrL inter.c:17-21 */
typedef struct { int w0, w1, o, wd; } Wod;
static inline int wpred(int q, int p, const Wod *w)
{
	int x = sat16(q * (int8_t)w->w0 + p * (int8_t)w->w1); /* pmaddubsw */
	x = sat16(x + (int16_t)w->o);                         /* adds16 */
	return clip255(x >> w->wd);
}

/* decode_inter weight selection, inter.c:1137-1197.  list = 0/1 of THIS call,
 * refIdx = this list's index, refIdxX = other list's index (-1 if unused). */
static void select_weights(const E264SliceParams *s, int list, int refIdx, int refIdxX, Wod *wY, Wod *wCb, Wod *wCr)
{
	Wod nw = {0, 1, 0, 0};
	*wY = *wCb = *wCr = nw;
	if (s->weighted_bipred_idc != 1) {
		if (list == 1 && refIdxX >= 0) {
			if (s->weighted_bipred_idc == 0) {
				Wod d = {1, 1, 1, 1};
				*wY = *wCb = *wCr = d;
			} else {
				int w1 = s->implicit_weights[refIdxX][refIdx] - 64;
				Wod d = {64 - w1, w1, 32, 6};
				if ((unsigned)(w1 + 63) >= 191) { d.w0 = 2 - (w1 >> 5); d.w1 = w1 >> 5; d.o = 1; d.wd = 1; }
				*wY = *wCb = *wCr = d;
			}
		}
	} else if (refIdxX < 0) { /* explicit1 */
		int i = refIdx + list * 32;
		if (s->explicit_weights[0][i] < 128) {
			wY->w1 = s->explicit_weights[0][i];
			wY->o = w16(((s->explicit_offsets[0][i] * 2 + 1) << s->luma_log2_weight_denom) >> 1);
			wY->wd = s->luma_log2_weight_denom;
		}
		if (s->explicit_weights[1][i] < 128) {
			wCb->w1 = s->explicit_weights[1][i];
			wCr->w1 = s->explicit_weights[2][i];
			wCb->o = w16(((s->explicit_offsets[1][i] * 2 + 1) << s->chroma_log2_weight_denom) >> 1);
			wCr->o = w16(((s->explicit_offsets[2][i] * 2 + 1) << s->chroma_log2_weight_denom) >> 1);
			wCb->wd = wCr->wd = s->chroma_log2_weight_denom;
		}
	} else if (list == 1) { /* explicit2 */
		int i = refIdx + 32, x = refIdxX;
		const int16_t (*ew)[64] = s->explicit_weights;
		const int8_t (*eo)[64] = s->explicit_offsets;
		if ((ew[0][x] & ew[0][i]) != 128) {
			wY->w0 = ew[0][x]; wY->w1 = ew[0][i];
			wY->o = w16(((eo[0][x] + eo[0][i] + 1) | 1) << s->luma_log2_weight_denom);
			wY->wd = s->luma_log2_weight_denom + 1;
		} else {
			wY->w0 = ew[0][x] >> 1; wY->w1 = ew[0][i] >> 1;
			wY->o = w16((((eo[0][x] + eo[0][i] + 1) | 1) << s->luma_log2_weight_denom) >> 1);
			wY->wd = s->luma_log2_weight_denom;
		}
		if ((ew[1][x] & ew[1][i]) != 128) {
			wCb->w0 = ew[1][x]; wCb->w1 = ew[1][i];
			wCr->w0 = ew[2][x]; wCr->w1 = ew[2][i];
			wCb->o = w16(((eo[1][x] + eo[1][i] + 1) | 1) << s->chroma_log2_weight_denom);
			wCr->o = w16(((eo[2][x] + eo[2][i] + 1) | 1) << s->chroma_log2_weight_denom);
			wCb->wd = wCr->wd = s->chroma_log2_weight_denom + 1;
		} else {
			wCb->w0 = ew[1][x] >> 1; wCb->w1 = ew[1][i] >> 1;
			wCr->w0 = ew[2][x] >> 1; wCr->w1 = ew[2][i] >> 1;
			wCb->o = w16((((eo[1][x] + eo[1][i] + 1) | 1) << s->chroma_log2_weight_denom) >> 1);
			wCr->o = w16((((eo[2][x] + eo[2][i] + 1) | 1) << s->chroma_log2_weight_denom) >> 1);
			wCb->wd = wCr->wd = s->chroma_log2_weight_denom;
		}
	}
}

typedef struct {
	const E264FrameHdr *h;
	const E264SliceParams *slices;
	const E264Mb *mbs;
	const E264Motion *motion; /* NULL if the frame has no inter macroblock */
	const uint8_t *payload;
	uint8_t *const *dpb;
	uint8_t *cur;
	int W, H; /* luma size in samples */
} Frame;

static inline uint8_t *plane_ptr(const Frame *f, uint8_t *base, int pl)
{ return pl == 0 ? base : base + f->h->plane_size_Y + (pl == 2 ? f->h->stride_C / 2 : 0); }

/* Inter prediction of one macroblock: for every 4x4 block, list 0 then list 1
 * (slice.c:1390-1436, 1226-1264; mvpred.c:505-514 call order). */
static void inter_mb(const Frame *f, const E264Mb *m, const E264Motion *mo, int mbx, int mby)
{
	const E264SliceParams *s = f->slices + m->slice;
	for (int list = 0; list < 2; list++) {
		for (int k = 0; k < 16; k++) {
			int i8 = k >> 2;
			int pic = mo->refPic[list * 4 + i8];
			if (pic < 0)
				continue;
			int refIdx = mo->refIdx[list * 4 + i8], refIdxX = mo->refIdx[(list ^ 1) * 4 + i8];
			Wod wY, wC[2];
			select_weights(s, list, refIdx, refIdxX, &wY, &wC[0], &wC[1]);
			int mx = mo->mvs[list * 32 + k * 2], my = mo->mvs[list * 32 + k * 2 + 1];
			uint8_t *ref = f->dpb[pic];
			Plane rY = {ref, (int)f->h->stride_Y, f->W, f->H};
			int x0 = mbx * 16 + BX[k], y0 = mby * 16 + BY[k];
			for (int y = 0; y < 4; y++)
				for (int x = 0; x < 4; x++) {
					uint8_t *d = f->cur + (y0 + y) * f->h->stride_Y + x0 + x;
					int pr = luma_sample(&rY, x0 + x + (mx >> 2), y0 + y + (my >> 2), mx & 3, my & 3);
					*d = (uint8_t)wpred(*d, pr, &wY);
				}
			for (int pl = 1; pl < 3; pl++) {
				Plane rC = {plane_ptr(f, ref, pl), (int)f->h->stride_C, f->W / 2, f->H / 2};
				uint8_t *dC = plane_ptr(f, f->cur, pl);
				int cx0 = mbx * 8 + BX[k] / 2, cy0 = mby * 8 + BY[k] / 2;
				for (int y = 0; y < 2; y++)
					for (int x = 0; x < 2; x++) {
						uint8_t *d = dC + (cy0 + y) * f->h->stride_C + cx0 + x;
						int pr = chroma_sample(&rC, cx0 + x + (mx >> 3), cy0 + y + (my >> 3), mx & 7, my & 7);
						*d = (uint8_t)wpred(*d, pr, &wC[pl - 1]);
					}
			}
		}
	}
}

/* ------------------------------------------------------------------------ */
/* Reconstruction of one macroblock (order of sample writes: SURVEY App. B)   */
/* ------------------------------------------------------------------------ */
static void chroma_residual(const Frame *f, const E264Mb *m, const E264SliceParams *s, const int16_t **pp, const int16_t *cdc, int mbx, int mby)
{
	int inter = m->kind == E264_MB_INTER;
	int32_t dc[8] = {0};
	if (m->coded & E264_CODED_CHROMA_DC)
		e264o_transform_dc2x2(cdc, m->qp[1], m->qp[2], s->weightScale4x4[1 + inter * 3][0], s->weightScale4x4[2 + inter * 3][0], dc);
	for (int k = 0; k < 8; k++) {
		int pl = 1 + (k >> 2);
		uint8_t *d = plane_ptr(f, f->cur, pl) + (mby * 8 + (k & 2) * 2) * f->h->stride_C + mbx * 8 + (k & 1) * 4;
		if (m->coded & E264_CODED_CHROMA(k)) {
			e264o_add_idct4x4(*pp, m->qp[pl], s->weightScale4x4[pl + inter * 3], 1, dc[k], d, (int)f->h->stride_C);
			*pp += 16;
		} else if (m->coded & E264_CODED_CHROMA_DC) {
			e264o_add_dc4x4(dc[k], d, (int)f->h->stride_C);
		}
	}
}

static void recon_mb(const Frame *f, int mbx, int mby)
{
	const E264Mb *m = f->mbs + mby * f->h->width_mbs + mbx;
	if (m->kind == E264_MB_ABSENT || (m->flags & E264_MBF_DONE)) /* DONE: an earlier packet of the picture wrote it (edge264_cmd.h) */
		return;
	const E264SliceParams *s = f->slices + m->slice;
	const uint8_t *pl = f->payload + m->payload_off;
	int sY = (int)f->h->stride_Y, sC = (int)f->h->stride_C;
	uint8_t *Y = f->cur + mby * 16 * sY + mbx * 16;
	uint8_t *Cb = plane_ptr(f, f->cur, 1) + mby * 8 * sC + mbx * 8;
	uint8_t *Cr = plane_ptr(f, f->cur, 2) + mby * 8 * sC + mbx * 8;
	if (m->kind == E264_MB_PCM) { /* slice.c:914-935 */
		for (int y = 0; y < 16; y++) memcpy(Y + y * sY, pl + y * 16, 16);
		for (int y = 0; y < 8; y++) { memcpy(Cb + y * sC, pl + 256 + y * 8, 8); memcpy(Cr + y * sC, pl + 320 + y * 8, 8); }
		return;
	}
	const E264Motion *mo = NULL;
	if (m->kind == E264_MB_INTER) mo = f->motion + (m - f->mbs);
	const int16_t *ldc = NULL, *cdc = NULL;
	if (m->coded & E264_CODED_LUMA_DC) { ldc = (const int16_t *)pl; pl += 32; }
	if (m->coded & E264_CODED_CHROMA_DC) { cdc = (const int16_t *)pl; pl += 16; }
	const int16_t *co = (const int16_t *)pl;
	int16_t wide[384]; /* E264_MBF_LEV8: the AC blocks arrive as bytes (edge264_cmd.h) */
	if (m->flags & E264_MBF_LEV8) {
		E264Mb dense = *m;
		dense.flags &= (uint8_t)~E264_MBF_LEV8;
		dense.coded &= ~(E264_CODED_LUMA_DC | E264_CODED_CHROMA_DC);
		int n = (int)e264_mb_payload_bytes(&dense) / 2;
		for (int i = 0; i < n; i++) wide[i] = (int8_t)pl[i];
		co = wide;
	}
	int inter = m->kind == E264_MB_INTER;
	int t8 = m->flags & E264_MBF_T8x8;

	if (inter)
		inter_mb(f, m, mo, mbx, mby);
	if (m->kind == E264_MB_I16x16) { /* slice.c:881, 513-556 */
		e264o_intra16x16(Y, sY, m->i16_mode);
		int32_t dc[16] = {0};
		if (ldc)
			e264o_transform_dc4x4(ldc, m->qp[0], s->weightScale4x4[0][0], dc);
		for (int k = 0; k < 16; k++) {
			uint8_t *d = Y + BY[k] * sY + BX[k];
			if (m->coded & E264_CODED_LUMA(k)) { e264o_add_idct4x4(co, m->qp[0], s->weightScale4x4[0], 1, dc[k], d, sY); co += 16; }
			else if (ldc) e264o_add_dc4x4(dc[k], d, sY);
		}
	} else if (!t8) { /* slice.c:615-635 */
		for (int k = 0; k < 16; k++) {
			uint8_t *d = Y + BY[k] * sY + BX[k];
			if (m->kind == E264_MB_I4x4)
				e264o_intra4x4(d, sY, m->modes[k >> 1] >> (4 * (k & 1)) & 15);
			if (m->coded & E264_CODED_LUMA(k)) { e264o_add_idct4x4(co, m->qp[0], s->weightScale4x4[inter * 3], 0, 0, d, sY); co += 16; }
		}
	} else { /* slice.c:645-668 */
		for (int b = 0; b < 4; b++) {
			uint8_t *d = Y + BY[b * 4] * sY + BX[b * 4];
			if (m->kind == E264_MB_I8x8)
				e264o_intra8x8(d, sY, m->modes[b]);
			if (m->coded & E264_CODED_LUMA(b * 4)) { e264o_add_idct8x8(co, m->qp[0], s->weightScale8x8[inter], d, sY); co += 64; }
		}
	}
	if (!inter) { /* slice.c:740: chroma prediction precedes all residual parsing, but only touches chroma */
		e264o_intra_chroma(Cb, sC, m->chroma_mode);
		e264o_intra_chroma(Cr, sC, m->chroma_mode);
	}
	chroma_residual(f, m, s, &co, cdc, mbx, mby);
}

/* ------------------------------------------------------------------------ */
/* Deblocking (edge264_deblock.c)                                             */
/* ------------------------------------------------------------------------ */
static const uint8_t ALPHA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28,
	32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
static const uint8_t BETA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8,
	9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
static const uint8_t TC0[3][52] = {
	{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13},
	{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17},
	{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25}};

/* motion of one 4x4 block for bS: refs as DPB slots, mvs raw (deblock.c:970-1086) */
typedef struct { int ref[2]; int mv[2][2]; } BlkMo;
static void blk_motion(const Frame *f, const E264Mb *m, int k, BlkMo *o)
{
	if (m->kind != E264_MB_INTER) { /* parse_I_mb: refPic=-1, mvs=0 (slice.c:805-807) */
		o->ref[0] = o->ref[1] = -1;
		o->mv[0][0] = o->mv[0][1] = o->mv[1][0] = o->mv[1][1] = 0;
		return;
	}
	const E264Motion *mo = f->motion + (m - f->mbs);
	for (int l = 0; l < 2; l++) {
		o->ref[l] = mo->refPic[l * 4 + (k >> 2)];
		o->mv[l][0] = mo->mvs[l * 32 + k * 2];
		o->mv[l][1] = mo->mvs[l * 32 + k * 2 + 1];
	}
}
static inline int mvfar(const int *a, const int *b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }

/* bS of one 4-sample edge segment between 4x4 blocks (mp,kp) and (mq,kq) which are both inter */
static int bs_inter(const Frame *f, const E264Mb *mp, int kp, const E264Mb *mq, int kq)
{
	if ((mp->nz_mask >> kp & 1) | (mq->nz_mask >> kq & 1))
		return 2; /* deblock.c:1093-1108 */
	BlkMo p, q;
	blk_motion(f, mp, kp, &p);
	blk_motion(f, mq, kq, &q);
	int refs_p = (p.ref[0] != q.ref[0]) | (p.ref[1] != q.ref[1]);
	int refs_c = (p.ref[0] != q.ref[1]) | (p.ref[1] != q.ref[0]);
	int mvs_p = mvfar(p.mv[0], q.mv[0]) | mvfar(p.mv[1], q.mv[1]);
	int mvs_c = mvfar(p.mv[0], q.mv[1]) | mvfar(p.mv[1], q.mv[0]);
	/* Karnaugh map deblock.c:913-925 */
	return (refs_p | mvs_p) & (refs_c | mvs_c);
}

static void filter_luma_line(uint8_t *q0p, int step, int bS, int alpha, int beta, int tc0)
{
	int p0 = q0p[-step], p1 = q0p[-2 * step], p2 = q0p[-3 * step], p3 = q0p[-4 * step];
	int q0 = q0p[0], q1 = q0p[step], q2 = q0p[2 * step], q3 = q0p[3 * step];
	if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta))
		return;
	int ap = abs(p2 - p0), aq = abs(q2 - q0);
	if (bS < 4) { /* DEBLOCK_LUMA_SOFT deblock.c:95-129 */
		int tc = tc0 + (ap < beta) + (aq < beta);
		int delta = clip3(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3);
		q0p[-step] = (uint8_t)clip255(p0 + delta);
		q0p[0] = (uint8_t)clip255(q0 - delta);
		if (ap < beta) q0p[-2 * step] = (uint8_t)(p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - 2 * p1) >> 1));
		if (aq < beta) q0p[step] = (uint8_t)(q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - 2 * q1) >> 1));
	} else { /* DEBLOCK_LUMA_HARD deblock.c:213-260 */
		int small = abs(p0 - q0) < (alpha >> 2) + 2;
		if (ap < beta && small) {
			q0p[-step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
			q0p[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
			q0p[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
		} else q0p[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
		if (aq < beta && small) {
			q0p[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
			q0p[step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
			q0p[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
		} else q0p[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
	}
}

static void filter_chroma_line(uint8_t *q0p, int step, int bS, int alpha, int beta, int tc0)
{
	int p0 = q0p[-step], p1 = q0p[-2 * step], q0 = q0p[0], q1 = q0p[step];
	if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta))
		return;
	if (bS < 4) { /* DEBLOCK_CHROMA_SOFT deblock.c:130-152 */
		int tc = tc0 + 1;
		int delta = clip3(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3);
		q0p[-step] = (uint8_t)clip255(p0 + delta);
		q0p[0] = (uint8_t)clip255(q0 - delta);
	} else { /* DEBLOCK_CHROMA_HARD deblock.c:261-276 */
		q0p[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
		q0p[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
	}
}

/* 4x4 block index (zig) from block coordinates */
static inline int blk(int bx, int by) { return (by >> 1) * 8 + (bx >> 1) * 4 + (by & 1) * 2 + (bx & 1); }

/* compute bS[dir][edge][seg] for one MB; dir 0 = vertical edges (left..), 1 = horizontal */
EXPORT void e264o_mb_bs(const Frame *f, int mbx, int mby, uint8_t bS[2][4][4])
{
	const E264Mb *m = f->mbs + mby * f->h->width_mbs + mbx;
	memset(bS, 0, 32);
	if (!(m->flags & E264_MBF_DEBLOCK) || m->kind == E264_MB_ABSENT)
		return;
	int intra = m->kind != E264_MB_INTER;
	for (int dir = 0; dir < 2; dir++) {
		for (int e = 0; e < 4; e++) {
			const E264Mb *n = m; /* macroblock holding the p side */
			if (e == 0) {
				if (!(m->flags & (dir ? E264_MBF_EDGE_TOP : E264_MBF_EDGE_LEFT)))
					continue;
				n = dir ? m - f->h->width_mbs : m - 1;
			} else if ((m->flags & E264_MBF_T8x8) && (e & 1)) {
				continue; /* deblock.c:704,766,827,881 */
			}
			for (int sgm = 0; sgm < 4; sgm++) {
				int b;
				if (e == 0 && (intra || n->kind != E264_MB_INTER)) b = 4; /* deblock.c:612-618, 811-819 */
				else if (intra) b = 3;                                   /* deblock.c:958-961 */
				else {
					int kq = dir ? blk(sgm, e) : blk(e, sgm);
					int kp = dir ? blk(sgm, (e + 3) & 3) : blk((e + 3) & 3, sgm);
					b = bs_inter(f, n, kp, m, kq);
				}
				bS[dir][e][sgm] = (uint8_t)b;
			}
		}
	}
}

static void deblock_mb(const Frame *f, int mbx, int mby)
{
	const E264Mb *m = f->mbs + mby * f->h->width_mbs + mbx;
	if (!(m->flags & E264_MBF_DEBLOCK) || m->kind == E264_MB_ABSENT)
		return;
	const E264SliceParams *s = f->slices + m->dbk_slice; /* FilterOffsetA/B: the slice whose task deblocks this macroblock (edge264_cmd.h) */
	uint8_t bS[2][4][4];
	e264o_mb_bs(f, mbx, mby, bS);
	int sY = (int)f->h->stride_Y, sC = (int)f->h->stride_C;
	for (int pl = 0; pl < 3; pl++) {
		uint8_t *base = plane_ptr(f, f->cur, pl) + (pl ? mby * 8 * sC + mbx * 8 : mby * 16 * sY + mbx * 16);
		int stride = pl ? sC : sY;
		for (int dir = 0; dir < 2; dir++) {
			for (int e = 0; e < 4; e++) {
				if (pl && (e & 1))
					continue; /* chroma 4:2:0 has edges at 0 and 4 (luma edges a,c / e,g), deblock.c:1117-1118 */
				const E264Mb *n = m;
				if (e == 0) {
					if (!(m->flags & (dir ? E264_MBF_EDGE_TOP : E264_MBF_EDGE_LEFT)))
						continue;
					n = dir ? m - f->h->width_mbs : m - 1;
				} else if (!pl && (m->flags & E264_MBF_T8x8) && (e & 1)) {
					continue;
				}
				int qPav = (m->qp[pl] + n->qp[pl] + 1) >> 1; /* deblock.c:945-951 */
				int iA = clip3(0, 51, qPav + s->FilterOffsetA), iB = clip3(0, 51, qPav + s->FilterOffsetB);
				int alpha = ALPHA[iA], beta = BETA[iB];
				int len = pl ? 8 : 16, pos = pl ? e * 2 : e * 4;
				for (int i = 0; i < len; i++) {
					int b = bS[dir][e][pl ? i >> 1 : i >> 2];
					if (b == 0)
						continue;
					uint8_t *q0 = dir ? base + pos * stride + i : base + i * stride + pos;
					int tc0 = b < 4 ? TC0[b - 1][iA] : 0;
					if (pl) filter_chroma_line(q0, dir ? stride : 1, b, alpha, beta, tc0);
					else filter_luma_line(q0, dir ? stride : 1, b, alpha, beta, tc0);
				}
			}
		}
	}
}

/* ------------------------------------------------------------------------ */
/* Packet driver                                                              */
/* ------------------------------------------------------------------------ */
static int open_frame(Frame *f, const uint8_t *pkt, size_t bytes, uint8_t *const *dpb)
{
	const E264FrameHdr *h = (const E264FrameHdr *)pkt;
	if (bytes < sizeof(*h) || h->magic != E264_MAGIC || h->version != E264_VERSION || h->total_bytes > bytes)
		return -1;
	f->h = h;
	f->slices = (const E264SliceParams *)(pkt + h->slices_off);
	f->mbs = (const E264Mb *)(pkt + h->mbs_off);
	f->motion = NULL;
	if (h->motion_off) { /* packets carry compact motion records (edge264_cmd.h): expanded once per frame */
		int n = h->width_mbs * h->height_mbs;
		E264Motion *mo = malloc(sizeof(E264Motion) * (size_t)n);
		if (!mo) return -3;
		for (int a = 0; a < n; a++) {
			memset(&mo[a], 0, sizeof(mo[a]));
			memset(mo[a].refPic, -1, 8);
			memset(mo[a].refIdx, -1, 8);
			if (f->mbs[a].kind == E264_MB_INTER) {
				uint32_t d[2];
				memcpy(d, f->mbs[a].modes, 8);
				e264_motion_expand(d[1], pkt + h->motion_off + d[0], &mo[a]);
			}
		}
		f->motion = mo;
	}
	f->payload = pkt + h->payload_off;
	f->dpb = dpb;
	if (h->dst_slot < 0 || h->dst_slot >= E264_MAX_SLOTS || !dpb[h->dst_slot]) {
		free((void *)f->motion);
		return -2;
	}
	f->cur = dpb[h->dst_slot];
	f->W = h->width_mbs * 16;
	f->H = h->height_mbs * 16;
	return 0;
}

/* passes: bit0 reconstruction, bit1 deblocking */
EXPORT int e264_oracle_decode_frame(const uint8_t *pkt, size_t bytes, uint8_t *const *dpb, int passes)
{
	Frame f;
	int r = open_frame(&f, pkt, bytes, dpb);
	if (r)
		return r;
	if (passes & 1)
		for (int y = 0; y < f.h->height_mbs; y++)
			for (int x = 0; x < f.h->width_mbs; x++)
				recon_mb(&f, x, y);
	if (passes & 2)
		for (int y = 0; y < f.h->height_mbs; y++)
			for (int x = 0; x < f.h->width_mbs; x++)
				deblock_mb(&f, x, y);
	free((void *)f.motion);
	return 0;
}

/* bS of the whole frame, 32 bytes per MB ([dir][edge][seg]) -- for kernel-level tests */
EXPORT int e264_oracle_frame_bs(const uint8_t *pkt, size_t bytes, uint8_t *out)
{
	Frame f;
	uint8_t *nodpb[E264_MAX_SLOTS];
	uint8_t dummy = 0;
	for (int i = 0; i < E264_MAX_SLOTS; i++) nodpb[i] = &dummy;
	int r = open_frame(&f, pkt, bytes, nodpb);
	if (r)
		return r;
	for (int y = 0; y < f.h->height_mbs; y++)
		for (int x = 0; x < f.h->width_mbs; x++)
			e264o_mb_bs(&f, x, y, (uint8_t (*)[4][4])(out + (y * f.h->width_mbs + x) * 32));
	free((void *)f.motion);
	return 0;
}

/* direct entry points for function-level differential tests */
EXPORT void e264o_luma_mc(const uint8_t *ref, int stride, int w, int h, int x, int y, int mvx, int mvy, int bw, int bh, uint8_t *dst, int dstride)
{
	Plane r = {ref, stride, w, h};
	for (int j = 0; j < bh; j++)
		for (int i = 0; i < bw; i++)
			dst[j * dstride + i] = (uint8_t)luma_sample(&r, x + i + (mvx >> 2), y + j + (mvy >> 2), mvx & 3, mvy & 3);
}
EXPORT void e264o_chroma_mc(const uint8_t *ref, int stride, int w, int h, int x, int y, int mvx, int mvy, int bw, int bh, uint8_t *dst, int dstride)
{
	Plane r = {ref, stride, w, h};
	for (int j = 0; j < bh; j++)
		for (int i = 0; i < bw; i++)
			dst[j * dstride + i] = (uint8_t)chroma_sample(&r, x + i + (mvx >> 3), y + j + (mvy >> 3), mvx & 7, mvy & 7);
}
EXPORT int e264o_wpred(int q, int p, int w0, int w1, int o, int wd)
{
	Wod w = {w0, w1, o, wd};
	return wpred(q, p, &w);
}
EXPORT void e264o_select_weights(const E264SliceParams *s, int list, int refIdx, int refIdxX, int *out12)
{
	Wod w[3];
	select_weights(s, list, refIdx, refIdxX, &w[0], &w[1], &w[2]);
	for (int i = 0; i < 3; i++) { out12[i * 4] = w[i].w0; out12[i * 4 + 1] = w[i].w1; out12[i * 4 + 2] = w[i].o; out12[i * 4 + 3] = w[i].wd; }
}
