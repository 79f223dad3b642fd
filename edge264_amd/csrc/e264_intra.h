// e264_intra.h -- e264_intra_kernel, the part one wave runs per macroblock: residual + intra prediction of one intra
// macroblock, written as PHASES of plain per-lane functions with a wave-level barrier between them.
//
// Device restatement of (file:line in /root/reference/src):
//   add_idct4x4 / add_dc4x4 / add_idct8x8 / transform_dc4x4 / transform_dc2x2   edge264_residual.c:108-538
//   decode_intra4x4 / 8x8 / 16x16 / Chroma                                        edge264_intra.c:291-765
//   the order of the blocks of a macroblock                                       edge264_slice.c:573-668
//
// Round 3 rewrite of what used to sit in e264_kernels.hip:
//   * ONE residual pass for the 24 4x4 blocks of a macroblock (16 luma + 4 Cb + 4 Cr), two lanes per block, instead of a
//     luma pass with 64 lanes and a chroma pass with 32 (280 -> ~160 VALU wave-instructions for a fully coded macroblock);
//     the pass writes every sample of the residual tile, so the tile is not cleared first.
//   * The DC modes of Intra4x4 without branches inside (one dword + v_sad_u8 for the row above); sums of the neighbours
//     for the Intra16x16 / chroma DC modes are gathered once, when the neighbours are committed to the tiles (LDS adds), instead of
//     16 + 16 byte reads per lane.
//   * Like e264_pred.h and e264_dbk.h the phases are plain functions of (LDS, lane): tests/emu runs them on the host, lane
//     by lane, against the CPU oracle (tests/test_intra_emu.py) before any GPU time is spent.
//
// IR_* hooks: the device build runs a phase on the wave's 64 lanes at once and separates phases with wave_sync(); the host
// build runs `for (lane = 0 .. 63)` around every phase, per-lane state in an array.
#ifndef E264_INTRA_H
#define E264_INTRA_H
#include "e264_dev.h"
#include "e264_pred.h" // lds_add

namespace {

#ifdef E264_HOST_INTRINSICS
#define IR_LANES for (int lane = 0; lane < 64; lane++)
#define IR_SYNC do { } while (0)
#define IR_S(st) (st)[lane]
#define IR_STATE_PARAM IrLane *st
#define IR_LANE_PARAM
#define IR_ANY(x) true
#define IR_UNIFORM(x) (x)
static inline uint32_t v_sad_u8(uint32_t a, uint32_t b, uint32_t c)
{
	for (int i = 0; i < 4; i++) { const int d = (int)(a >> (8 * i) & 255) - (int)(b >> (8 * i) & 255); c += (uint32_t)(d < 0 ? -d : d); }
	return c;
}
#else
#define IR_LANES
#define IR_SYNC wave_sync()
#define IR_S(st) (st)
#define IR_STATE_PARAM IrLane &st
#define IR_LANE_PARAM , const int lane
#define IR_ANY(x) (__ballot(x) != 0)
#define IR_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
E264_DEV uint32_t v_sad_u8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_u8(a, b, c); }
#endif

// luma tile: rows -1..15, columns -8..23 (top-right of an 8x8 block reaches x=23), 32-byte rows
#define YT_STRIDE 32
#define YT(y, x) ytile[((y) + 1) * YT_STRIDE + (x) + 8]
// chroma tiles: rows -1..7, columns -4..11
#define CT_STRIDE 16
#define CT(p, y, x) ctile[p][((y) + 1) * CT_STRIDE + (x) + 4]

struct __attribute__((aligned(16))) IntraWave { // reconstruction scratch of one wave
	int16_t res[384];          // residual: luma [y*16+x], Cb at 256 [y*8+x], Cr at 320
	int32_t tmp[384];          // transform intermediate: 24 4x4 blocks x 16 int32 (8x8: int16 view of the first 512 bytes)
	int32_t dc[24];            // 16 luma DC (zig order), 4 Cb, 4 Cr
	int sum[2];                // neighbours above (x = 0..15) / to the left (y = 0..15) of the macroblock, summed
	int csum[2][4];            // chroma, per plane: above x = 0..3, 4..7; left y = 0..3, 4..7
	uint8_t ytile[17 * YT_STRIDE];
	uint8_t ctile[2][9 * CT_STRIDE];
	uint8_t ftop[32];          // intra 8x8 filtered top  ft[-1..15] at [i+1]
	uint8_t fleft[8];          // intra 8x8 filtered left
	// scaling lists of the cached slice: weightScale4x4[6][16], weightScale8x8[0..1][64]
	__attribute__((aligned(4))) uint8_t ws[224];
	int ws_slice;              // slice index the cache holds (-1: none)
	int ws_idc;                // its weighted_bipred_idc
};

// uniform copy of one macroblock header
struct MbInfo {
	int kind, flags, chroma_mode, i16_mode, slice;
	uint8_t qp[3];
	uint32_t coded, payload_off, modes_lo, modes_hi;
};
// out of an LDS (host: any) copy of the record (8 dwords), all lanes reading the same words
E264_DEV MbInfo mb_from_rec(const uint32_t *rec)
{
	MbInfo m;
	const uint32_t d0 = IR_UNIFORM(rec[0]), d1 = IR_UNIFORM(rec[1]);
	const uint32_t d2 = IR_UNIFORM(rec[2]);
	m.kind = d0 & 255; m.flags = d0 >> 8 & 255; m.qp[0] = d0 >> 16 & 255; m.qp[1] = d0 >> 24;
	m.qp[2] = d1 & 255; m.chroma_mode = d1 >> 8 & 255; m.i16_mode = d1 >> 16 & 255;
	m.slice = d2 >> 16;
	m.coded = IR_UNIFORM(rec[3]); m.payload_off = IR_UNIFORM(rec[4]);
	m.modes_lo = IR_UNIFORM(rec[5]); m.modes_hi = IR_UNIFORM(rec[6]);
	return m;
}
// dwords of the macroblock's payload that the residual reads (edge264_cmd.h: [luma DC 16][chroma DC 8][coded blocks])
E264_DEV int coef_dwords(const MbInfo &m)
{
	if (m.kind == E264_MB_ABSENT || m.kind == E264_MB_PCM || m.coded == 0)
		return 0;
	const uint32_t c = m.coded;
	int ac = __builtin_popcount(c >> 16 & 0xff) * 32;
	ac += (m.kind != E264_MB_I16x16 && (m.flags & E264_MBF_T8x8)) ? __builtin_popcount(c & 0x1111) * 128 : __builtin_popcount(c & 0xffff) * 32;
	if (m.flags & E264_MBF_LEV8) ac >>= 1; // one byte per AC level
	return (((c & E264_CODED_LUMA_DC) ? 32 : 0) + ((c & E264_CODED_CHROMA_DC) ? 16 : 0) + ac + 3) >> 2;
}

// neighbours of the macroblock from the frame: issued early (loads only), committed to the tiles after the residual
struct IntraNb { uint32_t y, c; bool oky, okc; };
struct IrLane { // what a lane keeps from one phase to the next
	IntraNb nb;
	int v;                 // Intra4x4 / 8x8: the sample of the running step
	int bb_n, mode_n;      // Intra4x4: block and mode of the next step ...
	uint32_t e_n;          // ... and its table word
	int pY[4], pC[2];
};

// ---------------------------------------------------------------------------------------------------------------------
// residual
// ---------------------------------------------------------------------------------------------------------------------
struct ResPlan { // uniform
	uint32_t coded;
	bool luma4;            // the luma blocks are 4x4 (Intra16x16, or no 8x8 transform)
	bool luma8;
	bool i16, inter, l8, has_ldc, has_cdc;
	int nluma;             // AC levels of the luma blocks (the chroma blocks follow them in the payload)
	const int16_t *ldc, *cdc;
	const uint8_t *co;     // the AC blocks: int16 levels, or int8 (E264_MBF_LEV8)
	int qp[3];
};
E264_DEV ResPlan res_plan(const MbInfo &m, const int16_t *coefs)
{
	ResPlan p;
	p.coded = m.coded;
	p.i16 = m.kind == E264_MB_I16x16;
	p.inter = m.kind == E264_MB_INTER;
	p.luma8 = !p.i16 && (m.flags & E264_MBF_T8x8);
	p.luma4 = !p.luma8;
	p.l8 = m.flags & E264_MBF_LEV8;
	const int16_t *pl = coefs;
	p.ldc = p.cdc = nullptr;
	p.has_ldc = m.coded & E264_CODED_LUMA_DC;
	p.has_cdc = m.coded & E264_CODED_CHROMA_DC;
	if (p.has_ldc) { p.ldc = pl; pl += 16; }
	if (p.has_cdc) { p.cdc = pl; pl += 8; }
	p.co = (const uint8_t *)pl;
	p.nluma = p.luma8 ? __builtin_popcount(m.coded & 0x1111) * 64 : __builtin_popcount(m.coded & 0xffff) * 16;
	p.qp[0] = m.qp[0]; p.qp[1] = m.qp[1]; p.qp[2] = m.qp[2];
	return p;
}

// phase R0: counters; clears what the transforms will not write
E264_DEV void res_begin(IntraWave &L, const ResPlan &p, int lane)
{
	if (lane < 24) L.dc[lane] = 0;
	else if (lane < 26) L.sum[lane - 24] = 0;
	else if (lane < 34) L.csum[(lane - 26) >> 2][(lane - 26) & 3] = 0;
	uint32_t *rz = (uint32_t *)L.res;
	if (!p.coded) { rz[lane] = 0; rz[lane + 64] = 0; rz[lane + 128] = 0; } // 384 int16
	else if (p.luma8) { rz[lane] = 0; rz[lane + 64] = 0; }                 // the 8x8 transform only writes its coded blocks
}

// phase R1, DC transforms (edge264_residual.c:352-399 and :456-480): one output per lane
E264_DEV void res_dc(IntraWave &L, const ResPlan &p, int lane)
{
	const uint8_t *ws4 = L.ws;
	if (p.ldc && lane < 16) {
		int r = lane >> 2, l = lane & 3, acc = 0;
		// f[r][l] = sum_m sum_i A[r][m] A[l][i] c[4i+m], A = rows {++++, ++--, +--+, +-+-}
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int mm = 0; mm < 4; mm++) {
				int v = p.ldc[4 * i + mm];
				bool neg = (((0xA6C0 >> (4 * r)) >> mm) ^ ((0xA6C0 >> (4 * l)) >> i)) & 1; // rows of A: bit set => negative
				acc += neg ? -v : v;
			}
		int qP = p.qp[0];
		int LS = (ws4[0] * norm4(qP % 6, 0)) << (qP / 6);
		int k = (r >> 1) * 8 + (l >> 1) * 4 + (r & 1) * 2 + (l & 1);
		L.dc[k] = (int)((uint32_t)acc * (uint32_t)LS + 32u) >> 6;
	}
	if (p.cdc && lane >= 16 && lane < 24) {
		int n = lane & 3, pc = (lane >> 2) & 1; // pc: 0 Cb, 1 Cr
		int c0 = p.cdc[pc], c4 = p.cdc[4 + pc], c2 = p.cdc[2 + pc], c6 = p.cdc[6 + pc];
		int v = n == 0 ? c0 + c4 + c2 + c6 : n == 1 ? c0 - c4 + c2 - c6 : n == 2 ? c0 + c4 - c2 - c6 : c0 - c4 - c2 + c6;
		int qP = pc ? p.qp[2] : p.qp[1];
		int LS = (ws4[(1 + pc + (p.inter ? 3 : 0)) * 16] * norm4(qP % 6, 0)) << (qP / 6);
		L.dc[16 + pc * 4 + n] = (int)((uint32_t)v * (uint32_t)LS) >> 5;
	}
}

// The 4x4 blocks of the macroblock, two lanes per block (lanes 0..31: luma block lane >> 1 in zig order, when the luma
// transform is 4x4; lanes 32..47: chroma block (lane - 32) >> 1, Cb 0..3 then Cr 4..7).
//   pass 1  lane h of a block: rows 2h, 2h+1: dequantisation + horizontal butterfly (edge264_residual.c:118-134) -> L.tmp
//   pass 2  lane h of a block: columns 2h, 2h+1: vertical butterfly, >> 6, saturation (:141-158), or the add_dc4x4 value of a
//           block with nothing but a DC (:174-187), or zeros: every sample of the block is written.
struct Res4Lane { int blk, h, k, pc; bool chroma, act, on; };
E264_DEV Res4Lane res4_lane(const ResPlan &p, int lane)
{
	Res4Lane r;
	r.blk = lane >> 1; r.h = lane & 1;
	r.chroma = r.blk >= 16;
	r.k = r.blk & 15;
	r.pc = r.chroma ? r.k >> 2 : 0;
	r.act = lane < 48 && p.coded != 0 && (r.chroma || p.luma4);
	const uint32_t mask = r.chroma ? p.coded >> 16 & 0xffu : p.coded & 0xffffu;
	r.on = r.act && (mask >> r.k & 1);
	return r;
}
E264_DEV void butterfly4(const int d[4], int add, int t[4])
{
	const int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = (d[3] >> 1) + d[1];
	t[0] = e0 + e3 + add; t[1] = e1 + e2 + add; t[2] = e1 - e2 + add; t[3] = e0 - e3 + add;
}
E264_DEV void res4_pass1(IntraWave &L, const ResPlan &p, int lane)
{
	const Res4Lane r = res4_lane(p, lane);
	if (!r.on)
		return;
	const uint32_t mask = r.chroma ? p.coded >> 16 & 0xffu : p.coded & 0xffffu;
	const int cb = (r.chroma ? p.nluma : 0) + __builtin_popcount(mask & ((1u << r.k) - 1)) * 16; // levels in front of this block
	const int qP = r.chroma ? p.qp[1 + r.pc] : p.qp[0];
	const int sh = (qP * 43) >> 8, m = qP - sh * 6; // qP / 6, qP % 6 for qP < 64
	const uint8_t *wS = L.ws + ((p.inter ? 3 : 0) + (r.chroma ? 1 + r.pc : 0)) * 16;
	const int na = na_byte(NA4_0, m), nb = na_byte(NA4_1, m), nc = na_byte(0x171412100e0dull, m);
	const bool use_dc = r.chroma || p.i16;
	int dA[4], dB[4]; // rows 2h (even) and 2h + 1 (odd)
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const int pos = x * 4 + 2 * r.h;
		int l0, l1;
		if (p.l8) {
			const uint32_t w = *(const uint16_t *)(p.co + cb + pos);
			l0 = (int)(int8_t)w; l1 = (int)(int8_t)(w >> 8);
		} else {
			const uint32_t w = *(const uint32_t *)(p.co + 2 * (cb + pos));
			l0 = (int)(int16_t)w; l1 = (int)w >> 16;
		}
		const uint32_t ww = *(const uint16_t *)(wS + pos);
		const int LS0 = (int)(ww & 255u) * ((x & 1) ? nc : na), LS1 = (int)(ww >> 8) * ((x & 1) ? nb : nc);
		dA[x] = (int)(((uint32_t)(l0 * LS0) << sh) + 8u) >> 4;
		dB[x] = (int)(((uint32_t)(l1 * LS1) << sh) + 8u) >> 4;
	}
	if (use_dc && r.h == 0)
		dA[0] = L.dc[r.chroma ? 16 + r.k : r.k];
	int tA[4], tB[4];
	butterfly4(dA, r.h == 0 ? 32 : 0, tA);
	butterfly4(dB, 0, tB);
	int32_t *t = L.tmp + r.blk * 16 + 2 * r.h;
#pragma unroll
	for (int x = 0; x < 4; x++) { t[x * 4] = tA[x]; t[x * 4 + 1] = tB[x]; }
}
E264_DEV void res4_pass2(IntraWave &L, const ResPlan &p, int lane)
{
	const Res4Lane r = res4_lane(p, lane);
	if (!r.act)
		return;
	int out[4][2];
	if (r.on) {
#pragma unroll
		for (int c = 0; c < 2; c++) {
			const int32_t *t = L.tmp + r.blk * 16 + (2 * r.h + c) * 4;
			const int f0 = t[0], f1 = t[1], f2 = t[2], f3 = t[3];
			const int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = (f3 >> 1) + f1;
			out[0][c] = sat16((g0 + g3) >> 6); out[1][c] = sat16((g1 + g2) >> 6);
			out[2][c] = sat16((g1 - g2) >> 6); out[3][c] = sat16((g0 - g3) >> 6);
		}
	} else {
		const bool dc_valid = r.chroma ? p.has_cdc : (p.i16 && p.has_ldc);
		const int v = dc_valid ? (int)(int16_t)((L.dc[r.chroma ? 16 + r.k : r.k] + 32) >> 6) : 0;
#pragma unroll
		for (int j = 0; j < 4; j++) out[j][0] = out[j][1] = v;
	}
	int16_t *dst;
	int stride;
	if (!r.chroma) { dst = L.res + BYf(r.k) * 16 + BXf(r.k) + 2 * r.h; stride = 16; }
	else { dst = L.res + 256 + r.pc * 64 + ((r.k >> 1) & 1) * 32 + (r.k & 1) * 4 + 2 * r.h; stride = 8; }
#pragma unroll
	for (int j = 0; j < 4; j++)
		*(uint32_t *)(dst + j * stride) = ((uint32_t)out[j][0] & 0xffffu) | (uint32_t)out[j][1] << 16;
}

// 8x8: lanes 0..31, (block b, lane index j).  int16 arithmetic with wraparound (edge264_residual.c:250-316).
E264_DEV void idct8_1d(int16_t d[8])
{
	int16_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5], d6 = d[6], d7 = d[7];
	int16_t e0 = (int16_t)(d0 + d4);
	int16_t e1 = (int16_t)(d5 - d3 - (int16_t)((d7 >> 1) + d7));
	int16_t e2 = (int16_t)(d0 - d4);
	int16_t e3 = (int16_t)(d1 + d7 - (int16_t)((d3 >> 1) + d3));
	int16_t e4 = (int16_t)((d2 >> 1) - d6);
	int16_t e5 = (int16_t)(d7 - d1 + (int16_t)((d5 >> 1) + d5));
	int16_t e6 = (int16_t)((d6 >> 1) + d2);
	int16_t e7 = (int16_t)(d3 + d5 + (int16_t)((d1 >> 1) + d1));
	int16_t f0 = (int16_t)(e0 + e6);
	int16_t f1 = (int16_t)((e7 >> 2) + e1);
	int16_t f2 = (int16_t)(e2 + e4);
	int16_t f3 = (int16_t)((e5 >> 2) + e3);
	int16_t f4 = (int16_t)(e2 - e4);
	int16_t f5 = (int16_t)((e3 >> 2) - e5);
	int16_t f6 = (int16_t)(e0 - e6);
	int16_t f7 = (int16_t)(e7 - (e1 >> 2));
	d[0] = (int16_t)(f0 + f7); d[1] = (int16_t)(f2 + f5); d[2] = (int16_t)(f4 + f3); d[3] = (int16_t)(f6 + f1);
	d[4] = (int16_t)(f6 - f1); d[5] = (int16_t)(f4 - f3); d[6] = (int16_t)(f2 - f5); d[7] = (int16_t)(f0 - f7);
}
E264_DEV int level_at(const uint8_t *base, int idx, bool l8)
{
	return l8 ? (int)((const int8_t *)base)[idx] : (int)((const int16_t *)base)[idx];
}
E264_DEV void res8_pass1(IntraWave &L, const ResPlan &p, int lane)
{
	const int b = lane >> 3, j = lane & 7;
	if (!(p.luma8 && lane < 32 && (p.coded >> (b * 4) & 1)))
		return;
	int16_t *t16 = (int16_t *)L.tmp;
	const uint8_t *wS = L.ws + 96 + (p.inter ? 64 : 0);
	int nb = 0;
	for (int i = 0; i < b; i++) nb += p.coded >> (i * 4) & 1;
	const int qP = p.qp[0], div = qP / 6, m = qP - div * 6;
	int16_t d[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		int pos = i * 8 + j;
		int LS = wS[pos] * norm8(m, pos);
		const int lev = level_at(p.co, nb * 64 + pos, p.l8);
		if (div < 6)
			d[i] = (int16_t)sat16((lev * LS + (1 << (5 - div))) >> (6 - div));
		else
			d[i] = (int16_t)(lev * (int)(int16_t)(LS << (div - 6)));
	}
	idct8_1d(d);
	// transposed read in pass 2: element [i][j]; +32 lands on the new vector 0 = all elements with j == 0
#pragma unroll
	for (int i = 0; i < 8; i++)
		t16[b * 64 + i * 8 + j] = (int16_t)(d[i] + (j == 0 ? 32 : 0));
}
E264_DEV void res8_pass2(IntraWave &L, const ResPlan &p, int lane)
{
	const int b = lane >> 3, i = lane & 7; // this lane now owns pixel column i
	if (!(p.luma8 && lane < 32 && (p.coded >> (b * 4) & 1)))
		return;
	const int16_t *t16 = (const int16_t *)L.tmp;
	int16_t d[8];
#pragma unroll
	for (int jj = 0; jj < 8; jj++)
		d[jj] = t16[b * 64 + i * 8 + jj];
	idct8_1d(d);
	int16_t *r = L.res + BYf(b * 4) * 16 + BXf(b * 4) + i;
#pragma unroll
	for (int jj = 0; jj < 8; jj++)
		r[jj * 16] = (int16_t)(d[jj] >> 6);
}

// ---------------------------------------------------------------------------------------------------------------------
// neighbours of the macroblock from the frame (reconstructed, not yet deblocked) into the tiles.  Out-of-frame positions
// are never dereferenced; the remapped modes never use them.
// ---------------------------------------------------------------------------------------------------------------------
// ONE predicated load per plane group and lane (address and predicate selected by the lane's role), nothing cleared first:
// role branches writing the same register one after the other serialise into one memory round trip each.
E264_DEV IntraNb issue_intra_neighbours(const FrameCtx &f, int mbx, int mby, int lane)
{
	IntraNb n;
#ifdef E264_HOST_INTRINSICS
	n.y = n.c = 0; // (the device build clears nothing in front of a conditional load)
#endif
	const bool left = lane >= 32 && lane < 48;
	// luma: top row x = -1..23 (lanes 0..24), left column (lanes 32..47)
	const bool topY = lane < 25;
	const int x = lane - 1, gx = mbx * 16 + x;
	n.oky = topY ? (mby > 0 && gx >= 0 && gx < f.W) : (left && mbx > 0);
	const gu8 *Y = f.cur + (size_t)(mby * 16) * f.sY + mbx * 16;
	const ptrdiff_t offY = topY ? (ptrdiff_t)x - f.sY : (ptrdiff_t)(lane - 32) * f.sY - 1;
	if (n.oky) n.y = Y[offY];
	// chroma: top rows of both planes (lanes 0..17: plane lane / 9, x = lane % 9 - 1), left columns (lanes 32..47)
	const bool topC = lane < 18;
	const int xc = lane % 9 - 1;
	const int pl = topC ? lane / 9 : (lane - 32) >> 3;
	n.okc = topC ? (mby > 0 && mbx * 8 + xc >= 0) : (left && mbx > 0);
	const gu8 *C = plane_base(f, f.cur, 1 + (pl & 1)) + (size_t)(mby * 8) * f.sC + mbx * 8;
	const ptrdiff_t offC = topC ? (ptrdiff_t)xc - f.sC : (ptrdiff_t)(lane & 7) * f.sC - 1;
	if (n.okc) n.c = C[offC];
	return n;
}
E264_DEV void commit_intra_neighbours(IntraWave &L, const IntraNb &n, int lane)
{ // unavailable neighbours read as 0 (never used: the parser resolved the modes against availability)
	const bool left = lane >= 32 && lane < 48;
	const uint8_t vy = n.oky ? (uint8_t)n.y : 0, vc = n.okc ? (uint8_t)n.c : 0;
	if (lane < 25) L.YT(-1, lane - 1) = vy;
	else if (left) L.YT(lane - 32, -1) = vy;
	if (lane < 18) L.CT(lane / 9, -1, lane % 9 - 1) = vc;
	else if (left) L.CT((lane - 32) >> 3, lane & 7, -1) = vc;
	// the sums the DC modes want
	if (lane >= 1 && lane <= 16) lds_add(&L.sum[0], vy);
	else if (left) lds_add(&L.sum[1], vy);
	if (lane < 18 && lane % 9 != 0) lds_add(&L.csum[lane / 9][(lane % 9 - 1) >> 2], vc);
	else if (left) lds_add(&L.csum[(lane - 32) >> 3][2 + ((lane & 7) >> 2)], vc);
}

// ---------------------------------------------------------------------------------------------------------------------
// intra prediction (modes: edge264_internal.h:564-634; U(navailable) suffixes A left, B top, C top-right, D top-left)
// ---------------------------------------------------------------------------------------------------------------------
#define LP(l, m, r) (((l) + 2 * (m) + (r) + 2) >> 2)
#include "e264_intra_tab.h"

// One sample of a 4x4 block from the tap table (tools/gen_intra4x4_table.py): one table read and three sample reads per
// lane, the same instructions for every directional mode.  e: the table word tab[mode * 16 + y * 4 + x], fetched by the
// caller one step ahead (it depends on the mode only, not on samples).
E264_DEV int intra4x4_tab(const IntraWave &L, uint32_t e, int X0, int Y0, int mode)
{
	const uint8_t *org = &L.YT(Y0 - 1, X0 - 1);
	const bool dc = mode >= 2 && mode <= 5;
	int v;
	{
		const int a = org[e & 255], b = org[e >> 8 & 255], c = org[e >> 16 & 255];
		const int ty = (int)(e >> 24), sh = ty >> 3;
		v = (a + (ty & 3) * b + ((ty & 4) ? c : 0) + ((1 << sh) >> 1)) >> sh;
	}
	if (IR_ANY(dc)) { // DC variants: top and left, top, left, none; no branch inside
		const int st = (int)v_sad_u8(*(const uint32_t *)(org + 1), 0, 0);
		const int sl = org[YT_STRIDE] + org[2 * YT_STRIDE] + org[3 * YT_STRIDE] + org[4 * YT_STRIDE];
		const bool useT = mode == 2 || mode == 3, useL = mode == 2 || mode == 4;
		const int s = (useT ? st : 0) + (useL ? sl : 0), shd = 1 + (int)useT + (int)useL;
		const int vd = mode == 5 ? 128 : (s + (1 << (shd - 1))) >> shd;
		v = dc ? vd : v;
	}
	return v;
}

// Intra4x4, edge264_slice.c:615-635: predict, add residual, next block.  The 16 blocks are decoded in 10 steps instead of
// 16: block (x,y) of the 4x4 grid only needs its left, top, top-left and (when the standard counts it as available, i.e.
// when it precedes in zig-zag order) top-right neighbours, all of which belong to earlier anti-diagonals x + 2y.  Lanes
// 0..15 take the first block of a diagonal, lanes 16..31 the second one; the modes are already resolved against
// availability by the parser.  zig-zag indices per step: (0,-)(1,-)(4,2)(5,3)(6,8)(7,9)(12,10)(13,11)(14,-)(15,-)
E264_DEV bool i4_on(int lane, int t) { return lane < 32 && !((lane >> 4) == 1 && (t < 2 || t > 7)); }
E264_DEV int i4_mode(const MbInfo &m, int lane, int t, int &bb)
{ // the lane's block and mode at step t (lanes that idle take block 0: valid addresses, nothing stored)
	const uint64_t firsts = 0xfedc765410ull, seconds = 0xffba9832ffull; // one nibble per step
	const uint64_t order = (lane >> 4) ? seconds : firsts;
	bb = i4_on(lane, t) ? (int)(order >> (4 * t) & 15) : 0;
	return (int)(((bb < 8 ? m.modes_lo : m.modes_hi) >> (4 * (bb & 7))) & 15);
}
E264_DEV void i4_first(const MbInfo &m, const uint32_t *i4tab, IrLane &s, int lane)
{
	s.mode_n = i4_mode(m, lane, 0, s.bb_n);
	s.e_n = i4tab[s.mode_n * 16 + (lane & 15)];
}
E264_DEV void i4_read(const IntraWave &L, const MbInfo &m, const uint32_t *i4tab, IrLane &s, int t, int &bb_out, int lane)
{
	const int bb = s.bb_n, mode = s.mode_n;
	const uint32_t e = s.e_n;
	if (t < 9) { // next step's table word: in flight during this step's sample reads
		s.mode_n = i4_mode(m, lane, t + 1, s.bb_n);
		s.e_n = i4tab[s.mode_n * 16 + (lane & 15)];
	}
	bb_out = bb;
	if (i4_on(lane, t)) {
		const int X0 = BXf(bb), Y0 = BYf(bb), x = lane & 3, y = (lane & 15) >> 2;
		const int v = intra4x4_tab(L, e, X0, Y0, mode);
		s.v = clip255(w16(v + L.res[(Y0 + y) * 16 + X0 + x]));
	}
}
E264_DEV void i4_write(IntraWave &L, const IrLane &s, int t, int bb, int lane)
{
	if (i4_on(lane, t))
		L.YT(BYf(bb) + ((lane & 15) >> 2), BXf(bb) + (lane & 3)) = (uint8_t)s.v;
}

__constant__ int8_t c_i8spec[32] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 5, 6, 7, 7, 7, 7, 8, 8};
__constant__ int8_t c_i8unav[32] = {0, 4, 8, 12, 0, 8, 0, 1, 5, 9, 13, 2, 10, 4, 8, 12, 3, 0, 4, 8, 12, 0, 4, 0, 4, 0, 0, 4, 8, 12, 0, 8};

// one 8x8 block: all 64 lanes = (y = lane>>3, x = lane&7).  8.3.2.2.1 filtering into L.ftop/L.fleft first.
E264_DEV void i8_filter(IntraWave &L, int X0, int Y0, int mode, int lane)
{
	const int sm = c_i8spec[mode], un = c_i8unav[mode];
	const bool useA = !(un & 1) && (sm == 1 || sm == 2 || sm == 4 || sm == 5 || sm == 6 || sm == 8);
	const bool useB = !(un & 2) && (sm == 0 || sm == 2 || sm == 3 || sm == 4 || sm == 5 || sm == 6 || sm == 7);
	const bool useC = useB && !(un & 4) && sm != 6;
	const bool cornerAvail = !(un & 8) && (useA || useB);
#define T(i) ((int)L.YT(Y0 - 1, X0 + (i)))
#define TC(i) ((i) < 8 || useC ? T(i) : T(7))
#define Lf(i) ((int)L.YT(Y0 + (i), X0 - 1))
	if (lane < 16) { // filtered top ft[0..15]
		int i = lane, v = 0;
		if (useB) {
			if (i == 0) v = cornerAvail ? LP(T(-1), T(0), T(1)) : (3 * T(0) + T(1) + 2) >> 2;
			else if (i == 15) v = (TC(14) + 3 * TC(15) + 2) >> 2;
			else v = LP(TC(i - 1), TC(i), TC(i + 1));
		}
		L.ftop[i + 1] = (uint8_t)v;
	} else if (lane < 24) { // filtered left
		int i = lane - 16, v = 0;
		if (useA) {
			if (i == 0) v = cornerAvail ? LP(T(-1), Lf(0), Lf(1)) : (3 * Lf(0) + Lf(1) + 2) >> 2;
			else if (i == 7) v = (Lf(6) + 3 * Lf(7) + 2) >> 2;
			else v = LP(Lf(i - 1), Lf(i), Lf(i + 1));
		}
		L.fleft[i] = (uint8_t)v;
	} else if (lane == 24) {
		int v = 0;
		if (cornerAvail) {
			if (!useB) v = (3 * T(-1) + Lf(0) + 2) >> 2;
			else if (!useA) v = (3 * T(-1) + T(0) + 2) >> 2;
			else v = LP(T(0), T(-1), Lf(0));
		}
		L.ftop[0] = (uint8_t)v;
	}
#undef T
#undef TC
#undef Lf
}
E264_DEV int i8_pred(const IntraWave &L, int X0, int Y0, int mode, int lane)
{
	const int sm = c_i8spec[mode], un = c_i8unav[mode];
	const bool useA = !(un & 1) && (sm == 1 || sm == 2 || sm == 4 || sm == 5 || sm == 6 || sm == 8);
	const bool useB = !(un & 2) && (sm == 0 || sm == 2 || sm == 3 || sm == 4 || sm == 5 || sm == 6 || sm == 7);
	const int x = lane & 7, y = lane >> 3;
#define FT(i) ((int)L.ftop[(i) + 1])
#define FL(i) ((i) < 0 ? (int)L.ftop[0] : (int)L.fleft[i])
	int v;
	if (mode == 16) v = 128;
	else switch (sm) {
	default:
	case 0: v = FT(x); break;
	case 1: v = FL(y); break;
	case 2: {
		int st = 0, sl = 0;
#pragma unroll
		for (int i = 0; i < 8; i++) { st += FT(i); sl += FL(i); }
		v = useA && useB ? (st + sl + 8) >> 4 : useB ? (st + 4) >> 3 : useA ? (sl + 4) >> 3 : 128;
		} break;
	case 3: v = (x == 7 && y == 7) ? (FT(14) + 3 * FT(15) + 2) >> 2 : LP(FT(x + y), FT(x + y + 1), FT(x + y + 2)); break;
	case 4:
		if (x > y) v = LP(FT(x - y - 2), FT(x - y - 1), FT(x - y));
		else if (x < y) v = LP(FL(y - x - 2), FL(y - x - 1), FL(y - x));
		else v = LP(FT(0), FT(-1), FL(0));
		break;
	case 5: {
		int z = 2 * x - y, i = x - (y >> 1);
		if (z >= 0 && !(z & 1)) v = (FT(i - 1) + FT(i) + 1) >> 1;
		else if (z >= 0) v = LP(FT(i - 2), FT(i - 1), FT(i));
		else if (z == -1) v = LP(FL(0), FT(-1), FT(0));
		else v = LP(FL(y - 2 * x - 1), FL(y - 2 * x - 2), FL(y - 2 * x - 3));
		} break;
	case 6: {
		int z = 2 * y - x, i = y - (x >> 1);
		if (z >= 0 && !(z & 1)) v = (FL(i - 1) + FL(i) + 1) >> 1;
		else if (z >= 0) v = LP(FL(i - 2), FL(i - 1), FL(i));
		else if (z == -1) v = LP(FL(0), FT(-1), FT(0));
		else v = LP(FT(x - 2 * y - 1), FT(x - 2 * y - 2), FT(x - 2 * y - 3));
		} break;
	case 7: {
		int i = x + (y >> 1);
		v = (y & 1) ? LP(FT(i), FT(i + 1), FT(i + 2)) : (FT(i) + FT(i + 1) + 1) >> 1;
		} break;
	case 8: {
		int z = x + 2 * y, i = y + (x >> 1);
		if (z > 13) v = FL(7);
		else if (z == 13) v = (FL(6) + 3 * FL(7) + 2) >> 2;
		else if (z & 1) v = LP(FL(i), FL(i + 1), FL(i + 2));
		else v = (FL(i) + FL(i + 1) + 1) >> 1;
		} break;
	}
#undef FT
#undef FL
	// add residual (add_idct8x8 tail, edge264_residual.c:318-342)
	return clip255(w16(v + L.res[(Y0 + y) * 16 + X0 + x]));
}

// Intra 16x16 in the pixel layout: 4 predicted samples for (row Yr, cols X..X+3)
E264_DEV void intra16x16_pred(const IntraWave &L, int mode, int X, int Yr, int out[4])
{
#define T(i) ((int)L.YT(-1, (i)))
#define Lf(i) ((int)L.YT((i), -1))
	switch (mode) {
	default:
	case 0: {
		const uint32_t w = *(const uint32_t *)&L.YT(-1, X);
		out[0] = w & 255; out[1] = w >> 8 & 255; out[2] = w >> 16 & 255; out[3] = w >> 24;
		return; }
	case 1: for (int i = 0; i < 4; i++) out[i] = Lf(Yr); return;
	case 2: case 3: case 4: case 5: {
		const int st = L.sum[0], sl = L.sum[1];
		int v = mode == 2 ? (st + sl + 16) >> 5 : mode == 3 ? (st + 8) >> 4 : mode == 4 ? (sl + 8) >> 4 : 128;
		for (int i = 0; i < 4; i++) out[i] = v;
		return; }
	case 6: {
		int Hh = 0, V = 0;
		for (int i = 0; i < 8; i++) {
			Hh += (i + 1) * (T(8 + i) - (i == 7 ? T(-1) : T(6 - i)));
			V += (i + 1) * (Lf(8 + i) - (i == 7 ? T(-1) : Lf(6 - i)));
		}
		int a = 16 * (Lf(15) + T(15)), b = (5 * Hh + 32) >> 6, c = (5 * V + 32) >> 6;
		for (int i = 0; i < 4; i++) out[i] = clip255((a + b * (X + i - 7) + c * (Yr - 7) + 16) >> 5);
		return; }
	}
#undef T
#undef Lf
}

// Intra chroma for plane p, samples (x, y) and (x + 1, y) (x even: both in the same 4x4 block)
E264_DEV void intra_chroma_px2(const IntraWave &L, int p, int mode, int x, int y, int out[2])
{
#define T(i) ((int)L.CT(p, -1, (i)))
#define Lf(i) ((int)L.CT(p, (i), -1))
	switch (mode) {
	default:
	case 0: case 1: case 2: case 3: {
		const int bx = x >> 2, by = y >> 2;
		const int t = L.csum[p][bx], l = L.csum[p][2 + by];
		int v;
		if (mode == 3) v = 128;
		else if (mode == 1) v = (t + 2) >> 2;
		else if (mode == 2) v = (l + 2) >> 2;
		else if (bx == by) v = (t + l + 4) >> 3;
		else v = bx ? (t + 2) >> 2 : (l + 2) >> 2;
		out[0] = out[1] = v;
		return; }
	case 4: out[0] = out[1] = Lf(y); return;
	case 5: out[0] = T(x); out[1] = T(x + 1); return;
	case 6: {
		int Hh = 0, V = 0;
		for (int i = 0; i < 4; i++) {
			Hh += (i + 1) * (T(4 + i) - (i == 3 ? T(-1) : T(2 - i)));
			V += (i + 1) * (Lf(4 + i) - (i == 3 ? T(-1) : Lf(2 - i)));
		}
		int a = 16 * (Lf(7) + T(7)), b = (34 * Hh + 32) >> 6, c = (34 * V + 32) >> 6;
		out[0] = clip255((a + b * (x - 3) + c * (y - 3) + 16) >> 5);
		out[1] = clip255((a + b * (x - 2) + c * (y - 3) + 16) >> 5);
		return; }
	}
#undef T
#undef Lf
}

#ifndef E264_I4_UNROLL
#define E264_I4_UNROLL 5 // steps of the Intra4x4 anti-diagonal loop unrolled together
#endif

// ---------------------------------------------------------------------------------------------------------------------
// One intra macroblock by one wave.  Before: st.nb = issue_intra_neighbours (loads in flight), coefs = the macroblock's
// payload in LDS, L.ws = the slice's scaling lists.  After: the macroblock's samples are in the frame.
// ---------------------------------------------------------------------------------------------------------------------
E264_DEV void intra_recon_mb(IntraWave &L, const FrameCtx &f, const MbInfo &m, int mbx, int mby, const uint32_t *i4tab, const int16_t *coefs,
	IR_STATE_PARAM IR_LANE_PARAM)
{
	const ResPlan rp = res_plan(m, coefs);
	IR_LANES res_begin(L, rp, lane);
	IR_SYNC;
	if (rp.coded) { // (uniform)
		if (rp.has_ldc || rp.has_cdc) {
			IR_LANES res_dc(L, rp, lane);
			IR_SYNC;
		}
		IR_LANES { res4_pass1(L, rp, lane); res8_pass1(L, rp, lane); }
		IR_SYNC;
		IR_LANES { res4_pass2(L, rp, lane); res8_pass2(L, rp, lane); }
		IR_SYNC;
	}
	IR_LANES commit_intra_neighbours(L, IR_S(st).nb, lane);
	IR_SYNC;
	bool tile_luma = false;
	if (m.kind == E264_MB_I16x16) {
		IR_LANES {
			const int k = lane >> 2, r = lane & 3;
			intra16x16_pred(L, m.i16_mode, BXf(k), BYf(k) + r, IR_S(st).pY);
		}
	} else if (m.kind == E264_MB_I4x4) {
		tile_luma = true;
		IR_LANES i4_first(m, i4tab, IR_S(st), lane);
#pragma unroll E264_I4_UNROLL
		for (int t = 0; t < 10; t++) {
#ifdef E264_HOST_INTRINSICS
			int bbs[64];
			IR_LANES i4_read(L, m, i4tab, IR_S(st), t, bbs[lane], lane);
			IR_LANES i4_write(L, IR_S(st), t, bbs[lane], lane);
#else
			int bb;
			i4_read(L, m, i4tab, st, t, bb, lane);
			IR_SYNC;
			i4_write(L, st, t, bb, lane);
			IR_SYNC;
#endif
		}
	} else { // I8x8, edge264_slice.c:645-668
		tile_luma = true;
#pragma unroll 1
		for (int b = 0; b < 4; b++) {
			const int X0 = BXf(b * 4), Y0 = BYf(b * 4), mode = (int)(m.modes_lo >> (8 * b) & 255);
			IR_LANES i8_filter(L, X0, Y0, mode, lane);
			IR_SYNC;
			IR_LANES IR_S(st).v = i8_pred(L, X0, Y0, mode, lane);
			IR_SYNC;
			IR_LANES L.YT(Y0 + (lane >> 3), X0 + (lane & 7)) = (uint8_t)IR_S(st).v;
			IR_SYNC;
		}
	}
	// chroma prediction; add residual, clip, store (int16 wrap add then packus: edge264_residual.c:160-171)
	IR_LANES {
		const int k = lane >> 2, r = lane & 3;
		const int X = BXf(k), Yr = BYf(k) + r;
		const int cpl = lane >> 5, cy = (lane >> 2) & 7, cx = (lane & 3) * 2;
		int pC[2];
		intra_chroma_px2(L, cpl, m.chroma_mode, cx, cy, pC);
		uint32_t outw;
		if (tile_luma) {
			outw = *(const uint32_t *)&L.YT(Yr, X);
		} else {
			const int *pY = IR_S(st).pY;
			const int16_t *rr = L.res + Yr * 16 + X;
			outw = (uint32_t)clip255(w16(pY[0] + rr[0])) | (uint32_t)clip255(w16(pY[1] + rr[1])) << 8 |
				(uint32_t)clip255(w16(pY[2] + rr[2])) << 16 | (uint32_t)clip255(w16(pY[3] + rr[3])) << 24;
		}
		gu8 *dY = f.cur + (size_t)(mby * 16 + Yr) * f.sY + mbx * 16 + X;
		gu8 *dC = plane_base(f, f.cur, 1 + cpl) + (size_t)(mby * 8 + cy) * f.sC + mbx * 8 + cx;
		*(gu32 *)dY = outw;
		const int16_t *rc = L.res + 256 + cpl * 64 + cy * 8 + cx;
		*(gu16 *)dC = (uint16_t)(clip255(w16(pC[0] + rc[0])) | clip255(w16(pC[1] + rc[1])) << 8);
	}
}

} // namespace
#endif
