// e264_intra.h -- e264_intra_kernel: residual + intra prediction of the intra macroblocks of one picture, ONE WORKGROUP PER
// PICTURE, ONE WAVE PER MACROBLOCK ROW: row y may reconstruct macroblock x once row y-1 has finished macroblock x+1 (progress
// counters in LDS).  Device restatement of (file:line in /root/reference/src):
//   residual   edge264_residual.c:108-538   (dequant + 4x4 / 8x8 integer IDCT, DC transforms)
//   intra      edge264_intra.c:291-765      (14 + 32 + 7 + 7 internal modes), block order edge264_slice.c:573-668
//
// The whole kernel lives here (moved out of e264_kernels.hip in round 3) so that tests/emu can compile it for the HOST as it
// is: the 64 lanes of a wave run as 64 fibres that meet at the collectives (wave_sync, E264_BALLOT, E264_FIRST), the wavefront of
// rows is one wave taking the rows in order (tests/emu/intra_emu.cpp, tests/test_intra_emu.py).  The hooks below are the
// only difference between the two builds.
#ifndef E264_INTRA_H
#define E264_INTRA_H
#include "e264_dev.h"

#ifndef E264_HOST_INTRINSICS
#define E264_FIRST(x) __builtin_amdgcn_readfirstlane(x)          // a value every lane of the wave holds, as a scalar
#define E264_BALLOT(x) __ballot(x)
#define E264_WG_SYNC() __syncthreads()
#define E264_SLEEP() __builtin_amdgcn_s_sleep(1)
#define E264_FENCE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")
#define E264_FENCE_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#define E264_PROGRESS_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define E264_PROGRESS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif

namespace {
#ifndef E264_HOST_INTRINSICS
E264_DEV uint32_t v_sad_u8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_u8(a, b, c); } // c + sum of |a_i - b_i| over the four bytes
E264_DEV int relane(int lane) { asm volatile("" : "+v"(lane)); return lane; } // see E264_INTRA_LAUNDER
#endif

// ---------------------------------------------------------------------------------
// per-wave LDS scratch
// ---------------------------------------------------------------------------------
// luma tile: rows -1..15, columns -8..23 (top-right of an 8x8 block reaches x=23), 32-byte rows
#define YT_STRIDE 32
#define YT(y, x) ytile[((y) + 1) * YT_STRIDE + (x) + 8]
// chroma tiles: rows -1..7, columns -4..11
#define CT_STRIDE 16
#define CT(p, y, x) ctile[p][((y) + 1) * CT_STRIDE + (x) + 4]

// 4x4 transform: block k's 16 intermediate values at tmp[k * T4_STRIDE]: 20 dwords apart, the 64 lanes of a pass (16 blocks x 4 lanes on consecutive
// dwords) fall on the 32 banks twice each; 16 apart they fell on 8 banks, eight times each
#define T4_STRIDE 20
struct __attribute__((aligned(16))) WaveLds { // reconstruction scratch of one wave
	int16_t res[384];          // residual: luma [y*16+x], Cb at 256 [y*8+x], Cr at 320
	int32_t tmp[16 * T4_STRIDE]; // IDCT intermediate (4x4: 16 blocks x 16 int32, T4_STRIDE apart; 8x8: int16 view of the first 512 bytes)
	int32_t dc[24];            // 16 luma DC (zig order), 4 Cb, 4 Cr
	uint8_t ytile[17 * YT_STRIDE];
	uint8_t ctile[2][9 * CT_STRIDE];
	__attribute__((aligned(4))) uint8_t fz[32]; // intra 8x8: the filtered edge, FL(j) at j, the corner at 8, FT(i) at 12 + i
	uint8_t pad_[8];
	// the LEFT neighbour columns once more, contiguous (round 5): luma 16 samples, Cb 8, Cr 8.  In the tiles a column is 32 / 16 bytes apart per sample: the
	// DC and plane modes of Intra16x16 / chroma read it with 16 (8) byte accesses per lane, each batch behind a full LDS drain (the ISA showed 14 + 8 + 6 + 6
	// `s_waitcnt lgkmcnt(0)` in those four places); as dwords it is 4 (2) reads and the sums are v_sad_u8
	__attribute__((aligned(4))) uint8_t lcol[3][16];
	// residual inputs, staged so that the transforms never wait for memory (see coef_dma / slice_cache; the payload buffers live
	// in IntraLds.coef)
	__attribute__((aligned(4))) uint8_t ws[224];   // scaling lists of the cached slice: weightScale4x4[6][16], weightScale8x8[0..1][64]
	int ws_slice;              // slice index the cache holds (-1: none)
	int ws_idc;                // its weighted_bipred_idc
};

// register copy of one macroblock header (uniform: fetched with scalar loads)
struct MbInfo {
	int kind, flags, chroma_mode, i16_mode, slice;
	uint8_t qp[3];
	uint32_t coded, payload_off, modes_lo, modes_hi;
};
// The same header out of an LDS copy of the record (8 dwords), all lanes reading the same words (intra kernel: the
// headers of 64 macroblocks of the row are fetched with two vector loads per lane instead of one scalar-memory round
// trip per macroblock in the middle of the dependency chain)
E264_DEV MbInfo mb_from_lds(const uint32_t *rec)
{
	MbInfo m;
	const uint32_t d0 = E264_FIRST(rec[0]), d1 = E264_FIRST(rec[1]);
	const uint32_t d2 = E264_FIRST(rec[2]);
	m.kind = d0 & 255; m.flags = d0 >> 8 & 255; m.qp[0] = d0 >> 16 & 255; m.qp[1] = d0 >> 24;
	m.qp[2] = d1 & 255; m.chroma_mode = d1 >> 8 & 255; m.i16_mode = d1 >> 16 & 255;
	m.slice = d2 >> 16;
	m.coded = E264_FIRST(rec[3]); m.payload_off = E264_FIRST(rec[4]);
	m.modes_lo = E264_FIRST(rec[5]); m.modes_hi = E264_FIRST(rec[6]);
	return m;
}

// ---------------------------------------------------------------------------------
// residual: fills lds.res for the whole macroblock
// ---------------------------------------------------------------------------------
// One 4x4 block per 4 lanes.  pass 1 (lane = block k, row y): dequant + horizontal butterfly
// (edge264_residual.c:118-134); pass 2 (lane = block k, column x'): vertical butterfly, >>6,
// saturate to int16 (residual.c:141-158).  dc_only blocks get the add_dc4x4 value (residual.c:174-187).
// level `idx` of the coefficient area at `base` (LDS): int16, or int8 with E264_MBF_LEV8
E264_DEV int level_at(const uint8_t *base, int idx, bool l8)
{
	return l8 ? (int)((const int8_t *)base)[idx] : (int)((const int16_t *)base)[idx];
}
E264_DEV void idct4x4_blocks(WaveLds &L, int nblk, uint32_t codedmask, bool use_dc, bool dc_valid,
	const uint8_t *coef_base, bool l8, const uint8_t *wS, int qP, int dc_off, int res_off, int res_stride, int lane)
{
	int k = lane >> 2, y = lane & 3;
	bool active = k < nblk;
	bool coded = active && (codedmask >> k & 1);
	if (coded) {
		// coefficient blocks are packed in increasing k: offset = popcount of lower coded bits
		const int cb = __builtin_popcount(codedmask & ((1u << k) - 1)) * 16; // levels before this block
		int sh = qP / 6, m = qP - sh * 6;
		int lev[4], d[4];
		// ONE test of the level size around the four reads (the compiler left `l8 ? int8 read : int16 read` as a scalar branch around each)
		if (l8) {
#pragma unroll
			for (int x = 0; x < 4; x++) lev[x] = ((const int8_t *)coef_base)[cb + x * 4 + y];
		} else {
#pragma unroll
			for (int x = 0; x < 4; x++) lev[x] = ((const int16_t *)coef_base)[cb + x * 4 + y];
		}
#pragma unroll
		for (int x = 0; x < 4; x++) {
			int pos = x * 4 + y;
			int LS = wS[pos] * norm4(m, pos);
			d[x] = (int)(((uint32_t)mul24(lev[x], LS) << sh) + 8u) >> 4; // (|level| < 2^15, LS < 2^13)
		}
		if (use_dc && y == 0)
			d[0] = L.dc[dc_off + k];
		int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = (d[3] >> 1) + d[1];
		int32_t *t = L.tmp + k * T4_STRIDE;
		int add = (y == 0) ? 32 : 0;
		t[0 * 4 + y] = e0 + e3 + add;
		t[1 * 4 + y] = e1 + e2 + add;
		t[2 * 4 + y] = e1 - e2 + add;
		t[3 * 4 + y] = e0 - e3 + add;
	}
	wave_sync();
	if (active) {
		int x = y; // second role of the low lane bits: column x'
		int16_t *r;
		if (nblk == 16) // luma: block k in zig order inside the 16x16 tile
			r = L.res + res_off + BYf(k) * res_stride + BXf(k) + x;
		else            // chroma: plane k>>2 (64 samples each), 2x2 blocks of an 8x8 tile
			r = L.res + res_off + (k >> 2) * 64 + ((k >> 1) & 1) * 4 * res_stride + (k & 1) * 4 + x;
		if (coded) {
			const int32_t *t = L.tmp + k * T4_STRIDE + x * 4;
			int f0 = t[0], f1 = t[1], f2 = t[2], f3 = t[3];
			int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = (f3 >> 1) + f1;
			r[0 * res_stride] = (int16_t)sat16((g0 + g3) >> 6);
			r[1 * res_stride] = (int16_t)sat16((g1 + g2) >> 6);
			r[2 * res_stride] = (int16_t)sat16((g1 - g2) >> 6);
			r[3 * res_stride] = (int16_t)sat16((g0 - g3) >> 6);
		} else if (dc_valid) {
			int16_t v = (int16_t)((L.dc[dc_off + k] + 32) >> 6);
			r[0 * res_stride] = v; r[1 * res_stride] = v; r[2 * res_stride] = v; r[3 * res_stride] = v;
		}
	}
	wave_sync();
}

// 8x8: lanes 0..31, (block b, lane index j).  int16 arithmetic with wraparound (residual.c:250-316).
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

E264_DEV void idct8x8_blocks(WaveLds &L, uint32_t coded, const uint8_t *coef_base, bool l8, const uint8_t *wS, int qP, int lane)
{
	int b = lane >> 3, j = lane & 7;
	bool on = lane < 32 && (coded >> (b * 4) & 1);
	int16_t *t16 = (int16_t *)L.tmp;
	if (on) {
		const int nb = __builtin_popcount(coded & 0x1111u & ((1u << (b * 4)) - 1)); // coded 8x8 blocks before this one
		int div = qP / 6, m = qP - div * 6;
		// residual.c:214-247 has two forms: qP < 36: saturate((level * LS + 2^(5 - div)) >> (6 - div)) to int16; else (level * int16(LS << (div - 6)))
		// wrapped to int16.  One flow for both (all scalars): the shifts and the rounding term are 0 on the side that has none, the
		// clamp bounds are int16's for the first form and int32's (no clamp; the store wraps) for the second.
		const int shl = max(div - 6, 0), shr = max(6 - div, 0), rnd = div < 6 ? 1 << (5 - div) : 0;
		const int lo = div < 6 ? -32768 : (int)0x80000000, hi = div < 6 ? 32767 : 0x7fffffff;
		int16_t d[8];
		int lev[8];
		if (l8) { // (one test of the level size around the eight reads)
#pragma unroll
			for (int i = 0; i < 8; i++) lev[i] = ((const int8_t *)coef_base)[nb * 64 + i * 8 + j];
		} else {
#pragma unroll
			for (int i = 0; i < 8; i++) lev[i] = ((const int16_t *)coef_base)[nb * 64 + i * 8 + j];
		}
#pragma unroll
		for (int i = 0; i < 8; i++) {
			int pos = i * 8 + j;
			int LS = wS[pos] * norm8(m, pos);
			d[i] = (int16_t)min(max((mul24(lev[i], (int)(int16_t)(LS << shl)) + rnd) >> shr, lo), hi);
		}
		idct8_1d(d);
		// transposed read in pass 2: element [i][j]
#pragma unroll
		for (int i = 0; i < 8; i++)
			t16[b * 64 + i * 8 + j] = d[i];
	}
	wave_sync();
	if (on) {
		int i = j; // this lane now owns vector-lane i = pixel column i
		int16_t d[8];
#pragma unroll
		for (int jj = 0; jj < 8; jj++)
			d[jj] = t16[b * 64 + i * 8 + jj];
		d[0] = (int16_t)(d[0] + 32); // the rounding term of the >> 6 below, on the new vector 0 (residual.c:288: int16 wrap-around as there)
		idct8_1d(d);
		int16_t *r = L.res + BYf(b * 4) * 16 + BXf(b * 4) + i;
#pragma unroll
		for (int jj = 0; jj < 8; jj++)
			r[jj * 16] = (int16_t)(d[jj] >> 6);
	}
	wave_sync();
}

// Residual inputs without a memory round trip inside the transforms:
//   coef_dma      the macroblock's payload (<= 816 bytes) goes from the packet straight into L.coef[buf] by LDS-DMA
//                 (global_load_lds_dword: 4 instructions of 64 dwords, no register in between), requested while the
//                 PREVIOUS macroblock of the wave is reconstructed; the consumer waits with vmcnt(0) at the top of its
//                 macroblock, thousands of cycles later.  (Round 2: four dword loads per lane into registers at the top of the
//                 same macroblock, then a copy into LDS: the latency stood at the head of every macroblock's chain, and
//                 keeping a second set of registers for the next macroblock spilled.)
//   slice_cache   scaling lists + weighted_bipred_idc of the current slice in LDS, reloaded when the
//                 slice index changes.
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
E264_DEV void coef_dma(int16_t *dst, const FrameCtx &f, const MbInfo &m, int lane)
{ // dst: one of the wave's two payload buffers (uniform address); lanes beyond the payload are masked out and write nothing
	const int ndw = coef_dwords(m); // uniform
	if (ndw == 0)
		return;
	const gu32 *src = (const gu32 *)(f.payload + m.payload_off) + lane;
#ifndef E264_HOST_INTRINSICS
	typedef __attribute__((address_space(3))) void *lds_p;
	if (lane < ndw) __builtin_amdgcn_global_load_lds(src, (lds_p)dst, 4, 0, 0);
	if (ndw > 64 && 64 + lane < ndw) __builtin_amdgcn_global_load_lds(src + 64, (lds_p)(dst + 128), 4, 0, 0);
	if (ndw > 128 && 128 + lane < ndw) __builtin_amdgcn_global_load_lds(src + 128, (lds_p)(dst + 256), 4, 0, 0);
	if (ndw > 192 && 192 + lane < ndw) __builtin_amdgcn_global_load_lds(src + 192, (lds_p)(dst + 384), 4, 0, 0);
#else // host build: the same dwords, copied by the lanes
	for (int k = 0; k < 4; k++)
		if (64 * k + lane < ndw) ((uint32_t *)dst)[64 * k + lane] = src[64 * k];
#endif
}
// every LDS-DMA transfer this wave has requested has landed (they were requested a macroblock ago: no stall in steady state)
E264_DEV void coef_dma_wait()
{
#ifndef E264_HOST_INTRINSICS
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	wave_sync();
}
E264_DEV void slice_cache(WaveLds &L, const FrameCtx &f, int slice, int lane)
{
	if (E264_FIRST(L.ws_slice) == slice) // uniform
		return;
	cslice_t s = f.slices + slice;
	wave_sync();
	const gu32 *g4 = (const gu32 *)((const gu8 *)s + offsetof(E264SliceParams, weightScale4x4));
	const gu32 *g8 = (const gu32 *)((const gu8 *)s + offsetof(E264SliceParams, weightScale8x8));
	if (lane < 24) ((uint32_t *)L.ws)[lane] = g4[lane];
	else if (lane < 56) ((uint32_t *)L.ws)[lane] = g8[lane - 24];
	if (lane == 0) { L.ws_slice = slice; L.ws_idc = s->weighted_bipred_idc; }
	wave_sync();
}

// coefs holds the macroblock's payload (coef_dma + coef_dma_wait), the slice cache is valid (slice_cache)
// inter: the macroblock is an inter one (its own scaling lists); the intra kernel never sees one and passes a constant
// PLANES: 1 = luma only, 2 = chroma only, 3 = both (e264_intra_planes_kernel: an I picture's luma and chroma on two workgroups)
template <int PLANES = 3>
E264_DEV void compute_residual(WaveLds &L, const int16_t *coefs, const FrameCtx &f, const MbInfo &m, bool inter, int lane)
{
	// zero the residual tile (384 int16 = 192 dwords)
	uint32_t *rz = (uint32_t *)L.res;
	rz[lane] = 0; rz[lane + 64] = 0; rz[lane + 128] = 0;
	const uint32_t coded = m.coded;
	const int16_t *pl = coefs;
	const int16_t *ldc = nullptr, *cdc = nullptr;
	if (coded & E264_CODED_LUMA_DC) { ldc = pl; pl += 16; }
	if (coded & E264_CODED_CHROMA_DC) { cdc = pl; pl += 8; }
	const uint8_t *co = (const uint8_t *)pl;          // the AC blocks: int16 levels, or int8 (E264_MBF_LEV8)
	const bool l8 = m.flags & E264_MBF_LEV8;
	const int lsz = l8 ? 1 : 2;                         // bytes per level
	const uint8_t *ws4 = L.ws, *ws8 = L.ws + 96;
	if (lane < 24) L.dc[lane] = 0;
	wave_sync();
	if (!coded) return;

	// DC transforms (residual.c:352-399 and :456-480): one output per lane
	if ((PLANES & 1) && ldc && lane < 16) {
		int r = lane >> 2, l = lane & 3, acc = 0;
		// f[r][l] = sum_m sum_i A[r][m] A[l][i] c[4i+m], A = rows {++++, ++--, +--+, +-+-}
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int mm = 0; mm < 4; mm++) {
				int v = ldc[4 * i + mm];
				bool neg = (((0xA6C0 >> (4 * r)) >> mm) ^ ((0xA6C0 >> (4 * l)) >> i)) & 1; // rows of A: bit set => negative
				acc += neg ? -v : v;
			}
		int qP = m.qp[0];
		int LS = (ws4[0] * norm4(qP % 6, 0)) << (qP / 6);
		int k = (r >> 1) * 8 + (l >> 1) * 4 + (r & 1) * 2 + (l & 1);
		L.dc[k] = (int)((uint32_t)acc * (uint32_t)LS + 32u) >> 6;
	}
	if ((PLANES & 2) && cdc && lane >= 16 && lane < 24) {
		int n = lane & 3, pc = (lane >> 2) & 1; // pc: 0 Cb, 1 Cr
		int c0 = cdc[pc], c4 = cdc[4 + pc], c2 = cdc[2 + pc], c6 = cdc[6 + pc];
		int v = n == 0 ? c0 + c4 + c2 + c6 : n == 1 ? c0 - c4 + c2 - c6 : n == 2 ? c0 + c4 - c2 - c6 : c0 - c4 - c2 + c6;
		int qP = pc ? m.qp[2] : m.qp[1];
		int LS = (ws4[(1 + pc + (inter ? 3 : 0)) * 16] * norm4(qP % 6, 0)) << (qP / 6);
		L.dc[16 + pc * 4 + n] = (int)((uint32_t)v * (uint32_t)LS) >> 5;
	}
	wave_sync();

	// luma
	if (m.kind == E264_MB_I16x16) {
		if ((PLANES & 1) && (coded & (0xffff | E264_CODED_LUMA_DC)))
			idct4x4_blocks(L, 16, coded & 0xffff, true, ldc != nullptr, co, l8, ws4, m.qp[0], 0, 0, 16, lane);
		co += __builtin_popcount(coded & 0xffff) * 16 * lsz;
	} else if (m.flags & E264_MBF_T8x8) {
		if ((PLANES & 1) && (coded & 0x1111))
			idct8x8_blocks(L, coded, co, l8, ws8 + (inter ? 64 : 0), m.qp[0], lane);
		co += __builtin_popcount(coded & 0x1111) * 64 * lsz;
	} else {
		if ((PLANES & 1) && (coded & 0xffff))
			idct4x4_blocks(L, 16, coded & 0xffff, false, false, co, l8, ws4 + (inter ? 3 : 0) * 16, m.qp[0], 0, 0, 16, lane);
		co += __builtin_popcount(coded & 0xffff) * 16 * lsz;
	}
	// chroma: 8 blocks, Cb 0..3 then Cr 4..7; different QP / scaling list per plane
	if ((PLANES & 2) && (coded & (0xff0000 | E264_CODED_CHROMA_DC))) {
		int k = lane >> 2;
		int pc = (k >> 2) & 1;
		idct4x4_blocks(L, 8, coded >> 16 & 0xff, true, cdc != nullptr, co, l8, ws4 + (1 + pc + (inter ? 3 : 0)) * 16, pc ? m.qp[2] : m.qp[1], 16, 256, 8, lane);
	}
}

// ---------------------------------------------------------------------------------
// intra prediction (modes: edge264_internal.h:564-634; U(navailable) suffixes A left, B top,
// C top-right, D top-left)
// ---------------------------------------------------------------------------------
#define LP(l, m, r) (((l) + 2 * (m) + (r) + 2) >> 2)
#ifndef E264_I4_UNROLL
#define E264_I4_UNROLL 10 // (history: the Intra4x4 anti-diagonal loop is always fully unrolled now) all ten (what a lane does in a step is then a handful of
                          // loop-invariant registers; affordable since E264_INTRA_LAUNDER freed the registers: round 2 spilled with it)
#endif
#include "e264_intra_tab.h"

// neighbours of the macroblock from the frame (un-deblocked, pass R) into the tiles.
// Out-of-frame positions are never dereferenced; the remapped modes never use them.
// In two steps, so that the frame reads are in flight while the residual is computed: issue (loads only, nothing may
// use the values) and commit (values -> tiles).
struct IntraNb { uint32_t y, c; bool oky, okc; };
// ONE predicated load per plane group and lane (address and predicate selected by the lane's role), nothing cleared first:
// the four role branches used to write the same two registers one after the other, which serialised them into two extra
// memory round trips per intra macroblock (s_waitcnt vmcnt(0) between the loads).
E264_DEV IntraNb issue_intra_neighbours(const FrameCtx &f, int mbx, int mby, int lane)
{
	IntraNb n;
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
E264_DEV void commit_intra_neighbours(WaveLds &L, const IntraNb &n, int lane)
{ // unavailable neighbours read as 0 (never used: the parser resolved the modes against availability)
	const bool left = lane >= 32 && lane < 48;
	const uint8_t vy = n.oky ? (uint8_t)n.y : 0, vc = n.okc ? (uint8_t)n.c : 0;
	if (lane < 25) L.YT(-1, lane - 1) = vy;
	else if (left) { L.YT(lane - 32, -1) = vy; L.lcol[0][lane - 32] = vy; }
	if (lane < 18) L.CT(lane / 9, -1, lane % 9 - 1) = vc;
	else if (left) { L.CT((lane - 32) >> 3, lane & 7, -1) = vc; L.lcol[1 + ((lane - 32) >> 3)][lane & 7] = vc; }
	wave_sync();
}

// One sample of a 4x4 block from the tap table (tools/gen_intra4x4_table.py): one table read and three sample reads per
// lane, the same instructions for every directional mode.  The 14-way switch below (kept as the readable statement of the
// modes; the generator checks the table against the same formulas) ran as two divergent paths per step -- one per block
// of the step -- each a chain of dependent byte reads: 46 % of the kernel's time on I pictures.
// tab: c_i4tab in LDS; p = y * 4 + x.
// e: the table word tab[mode * 16 + p], fetched by the caller ONE STEP AHEAD (it depends on the mode only, not on samples: read
// where it is used it was a third LDS round trip in every step of the chain table word -> samples -> store)
E264_DEV int intra4x4_tab(const WaveLds &L, uint32_t e, int X0, int Y0, int mode)
{
	const uint8_t *org = &L.YT(Y0 - 1, X0 - 1);
	const bool dc = mode >= 2 && mode <= 5;
	// the three taps of the directional modes are requested by every lane (DC lanes: table word 0, a valid position) BEFORE the
	// DC modes' branch, so that their latency runs under it instead of behind it
	const int a = org[e & 255], b = org[e >> 8 & 255], c = org[e >> 16 & 255];
	int vd = 128;
	if (dc) { // DC variants: top and left, top, left, none (edge264_intra.c DC modes)
		const uint32_t w = *(const uint32_t *)(org + 1);
		const int st = (int)v_sad_u8(w, 0, 0); // the four samples above, summed
		const int sl = org[YT_STRIDE] + org[2 * YT_STRIDE] + org[3 * YT_STRIDE] + org[4 * YT_STRIDE];
		vd = mode == 2 ? (st + sl + 4) >> 3 : mode == 3 ? (st + 2) >> 2 : mode == 4 ? (sl + 2) >> 2 : 128;
	}
	const int ty = (int)(e >> 24), sh = ty >> 3;
	const int v = (a + (ty & 3) * b + ((ty & 4) ? c : 0) + ((1 << sh) >> 1)) >> sh;
	return dc ? vd : v;
}

// one 4x4 block, lanes 0..15 = (y = lane>>2, x = lane&3); reads/writes the luma tile
E264_DEV int intra4x4_px(const WaveLds &L, int X0, int Y0, int mode, int x, int y)
{
#define T(i) ((int)L.YT(Y0 - 1, X0 + (i)))
#define Lf(i) ((int)L.YT(Y0 + (i), X0 - 1))
#define TR(i) ((i) < 4 || has_tr ? T(i) : T(3))
	const bool has_tr = (mode == 6 || mode == 11);
	switch (mode) {
	default:
	case 0: return T(x);
	case 1: return Lf(y);
	case 2: return (T(0) + T(1) + T(2) + T(3) + Lf(0) + Lf(1) + Lf(2) + Lf(3) + 4) >> 3;
	case 3: return (T(0) + T(1) + T(2) + T(3) + 2) >> 2;
	case 4: return (Lf(0) + Lf(1) + Lf(2) + Lf(3) + 2) >> 2;
	case 5: return 128;
	case 6: case 7:
		return (x == 3 && y == 3) ? (TR(6) + 3 * TR(7) + 2) >> 2 : LP(TR(x + y), TR(x + y + 1), TR(x + y + 2));
	case 8:
		if (x > y) return LP(T(x - y - 2), T(x - y - 1), T(x - y));
		if (x < y) return LP(y - x - 2 < 0 ? T(-1) : Lf(y - x - 2), Lf(y - x - 1), Lf(y - x));
		return LP(T(0), T(-1), Lf(0));
	case 9: {
		int z = 2 * x - y, i = x - (y >> 1);
		if (z >= 0 && !(z & 1)) return (T(i - 1) + T(i) + 1) >> 1;
		if (z >= 0) return LP(T(i - 2), T(i - 1), T(i));
		if (z == -1) return LP(Lf(0), T(-1), T(0));
		return LP(Lf(y - 1), Lf(y - 2), y - 3 < 0 ? T(-1) : Lf(y - 3)); }
	case 10: {
		int z = 2 * y - x, i = y - (x >> 1);
#define L_(j) ((j) < 0 ? T(-1) : Lf(j))
		if (z >= 0 && !(z & 1)) return (L_(i - 1) + L_(i) + 1) >> 1;
		if (z >= 0) return LP(L_(i - 2), L_(i - 1), L_(i));
		if (z == -1) return LP(Lf(0), T(-1), T(0));
		return LP(T(x - 1), T(x - 2), T(x - 3));
#undef L_
		}
	case 11: case 12: {
		int i = x + (y >> 1);
		return (y & 1) ? LP(TR(i), TR(i + 1), TR(i + 2)) : (TR(i) + TR(i + 1) + 1) >> 1; }
	case 13: {
		int z = x + 2 * y, i = y + (x >> 1);
		if (z > 5) return Lf(3);
		if (z == 5) return (Lf(2) + 3 * Lf(3) + 2) >> 2;
		if (z & 1) return LP(Lf(i), Lf(i + 1), Lf(i + 2));
		return (Lf(i) + Lf(i + 1) + 1) >> 1; }
	}
#undef T
#undef Lf
#undef TR
}

// shape (0..8, c_i8tab) of the 32 internal Intra8x8 modes, one nibble each
// (immediates: the tables used to be two scalar memory reads at the head of every block)
E264_DEV int i8_nibble(unsigned long long lo, unsigned long long hi, int mode) { return (int)(((mode & 16) ? hi : lo) >> (4 * (mode & 15)) & 15); }
#define I8_SPEC_LO 0x2222222222110000ull
#define I8_SPEC_HI 0x8877776554433332ull
#define I8_FZ 1 // the wave's filtered edge: FL(j) at fz[j], the corner at fz[8], FT(i) at fz[12 + i] (c_i8tab's tap offsets)

// one 8x8 block by the whole wave.  Round 4: two branch-free phases (the first version walked three lane ranges with their special
// cases, then a nine-way switch of per-lane conditions: 3.3 x the time of an Intra4x4 macroblock, profiles/r04_ablations.txt item 16).
//   filter   8.3.2.2.1 on the 25 edge samples L7..L0, corner, T0..T15 as ONE line: lane k < 25 filters sample k from its two
//            neighbours on the line; ends, missing corner / top right and the corner's own rules are index replacements.
//            The line's address in the luma tile is piecewise linear in k (up the left column, then along the top row).
//   predict  lane = sample (y = lane >> 3, x = lane & 7): table word e (c_i8tab, fetched by the caller's previous step) = three taps on
//            the filtered edge + filter type, the same instructions for all eight directional shapes; DC from four v_sad_u8.
E264_DEV void intra8x8_block(WaveLds &L, uint32_t e, int X0, int Y0, int mode, int lane)
{
	// which neighbours the mode reads, one bit per internal mode (from the shape and the unavailability bits: left = shape in {1, 2, 4, 5, 6, 8}
	// and available; top = shape in {0, 2, 3, 4, 5, 6, 7} and available; top right = top, available, shape != 6; corner = available and
	// left or top): five scalar bit tests instead of thirty compares per block
	const bool useA = 0xc3e0f870u >> mode & 1, useB = 0x3ffee7cfu >> mode & 1, useC = 0x14aa42c5u >> mode & 1, cornerAvail = 0x4fe629d3u >> mode & 1;
	const bool isDC = 0x0001ffc0u >> mode & 1; // shape 2 (mode 16: nothing available, 128)
	const int x = lane & 7, y = lane >> 3;
	const int rres = L.res[(Y0 + y) * 16 + X0 + x]; // requested in front of everything else
	{
		const int k = min(lane, 24), hi = useC ? 24 : 16; // without a top right, T8..T15 read as T7
		int ia = max(k - 1, 0), ib = k, ic = k + 1;
		if (k == 7 && !cornerAvail) ic = 7;           // L0 without a corner: (3 L0 + L1 + 2) >> 2
		if (k == 9 && !cornerAvail) ia = 9;           // T0 likewise
		if (k == 8) { if (!useB) ic = 8; if (!useA) ia = 8; } // the corner with one side only: (3 corner + side + 2) >> 2
		ia = min(ia, hi); ib = min(ib, hi); ic = min(ic, hi);
		const uint8_t *A0 = &L.YT(Y0 + 7, X0 - 1);    // sample 0 of the line (L7); sample i: 32 bytes up per step until the corner (8), then to the right
		const int a = A0[max(ia - 8, 0) - 32 * min(ia, 8)], b = A0[max(ib - 8, 0) - 32 * min(ib, 8)], c = A0[max(ic - 8, 0) - 32 * min(ic, 8)];
		if (lane < 25)
			L.fz[lane < 8 ? 7 - lane : lane == 8 ? 8 : lane + 3] = (uint8_t)LP(a, b, c);
	}
	wave_sync();
	int v;
	if (isDC) { // (uniform) DC from the filtered edge (mode 16: nothing available)
		const uint32_t *w = (const uint32_t *)L.fz;
		const int sl = (int)v_sad_u8(w[0], 0, v_sad_u8(w[1], 0, 0)), st = (int)v_sad_u8(w[3], 0, v_sad_u8(w[4], 0, 0));
		v = useA && useB ? (st + sl + 8) >> 4 : useB ? (st + 4) >> 3 : useA ? (sl + 4) >> 3 : 128;
	} else {
		const int a = L.fz[e & 255], b = L.fz[e >> 8 & 255], c = L.fz[e >> 16 & 255];
		const int ty = (int)(e >> 24), sh = ty >> 3;
		v = (a + (ty & 3) * b + ((ty & 4) ? c : 0) + ((1 << sh) >> 1)) >> sh;
	}
	// add residual (add_idct8x8 tail, residual.c:318-342) and write the tile: every read of the tile by this block happened before the
	// wave_sync above
	L.YT(Y0 + y, X0 + x) = (uint8_t)clip255(w16(v + rres));
	wave_sync();
}

// Intra 16x16 in the pixel layout: returns 4 predicted samples for (row Yr, cols X..X+3)
E264_DEV int byte_of(const uint32_t *w, int j) { return (int)(w[j >> 2] >> (8 * (j & 3)) & 255u); } // sample j of a row / column held as dwords (j: a constant of an unrolled loop)
E264_DEV void intra16x16_pred(const WaveLds &L, int mode, int X, int Yr, int out[4])
{
#define T(i) ((int)L.YT(-1, (i)))
#define Lf(i) ((int)L.YT((i), -1))
	switch (mode) {
	default:
	case 0: for (int i = 0; i < 4; i++) out[i] = T(X + i); return;
	case 1: for (int i = 0; i < 4; i++) out[i] = Lf(Yr); return;
	case 2: case 3: case 4: case 5: {
		int st = 0, sl = 0;
		if (mode == 2 || mode == 3) for (int i = 0; i < 4; i++) st = (int)v_sad_u8(*(const uint32_t *)&L.YT(-1, 4 * i), 0, (uint32_t)st); // the row above, four samples per instruction
		if (mode == 2 || mode == 4) for (int i = 0; i < 4; i++) sl = (int)v_sad_u8(((const uint32_t *)L.lcol[0])[i], 0, (uint32_t)sl); // the left column out of its contiguous copy
		int v = mode == 2 ? (st + sl + 16) >> 5 : mode == 3 ? (st + 8) >> 4 : mode == 4 ? (sl + 8) >> 4 : 128;
		for (int i = 0; i < 4; i++) out[i] = v;
		return; }
	case 6: {
		// the row above and the left column as four dwords each, the corner as one byte: 9 LDS reads, the 32 samples extracted in registers
		uint32_t tw[4], lw[4];
		for (int i = 0; i < 4; i++) { tw[i] = *(const uint32_t *)&L.YT(-1, 4 * i); lw[i] = ((const uint32_t *)L.lcol[0])[i]; }
		const int corner = T(-1);
		int Hh = 0, V = 0;
		for (int i = 0; i < 8; i++) {
			Hh += (i + 1) * (byte_of(tw, 8 + i) - (i == 7 ? corner : byte_of(tw, 6 - i)));
			V += (i + 1) * (byte_of(lw, 8 + i) - (i == 7 ? corner : byte_of(lw, 6 - i)));
		}
		int a = 16 * (byte_of(lw, 15) + byte_of(tw, 15)), b = (5 * Hh + 32) >> 6, c = (5 * V + 32) >> 6;
		for (int i = 0; i < 4; i++) out[i] = clip255((a + b * (X + i - 7) + c * (Yr - 7) + 16) >> 5);
		return; }
	}
#undef T
#undef Lf
}

// Intra chroma for plane p, sample (x,y)
E264_DEV int intra_chroma_px(const WaveLds &L, int p, int mode, int x, int y)
{
#define T(i) ((int)L.CT(p, -1, (i)))
#define Lf(i) ((int)L.CT(p, (i), -1))
	switch (mode) {
	default:
	case 0: case 1: case 2: case 3: {
		if (mode == 3) return 128;
		int bx = x >> 2, by = y >> 2;
		int t = 0, l = 0;
		t = (int)v_sad_u8(*(const uint32_t *)&L.CT(p, -1, bx * 4), 0, 0);
		l = (int)v_sad_u8(((const uint32_t *)L.lcol[1 + p])[by], 0, 0); // four samples of the left column out of its contiguous copy
		if (mode == 1) return (t + 2) >> 2;
		if (mode == 2) return (l + 2) >> 2;
		if (bx == by) return (t + l + 4) >> 3;
		return bx ? (t + 2) >> 2 : (l + 2) >> 2; }
	case 4: return Lf(y);
	case 5: return T(x);
	case 6: {
		const uint32_t tw[2] = {*(const uint32_t *)&L.CT(p, -1, 0), *(const uint32_t *)&L.CT(p, -1, 4)};
		const uint32_t lw[2] = {((const uint32_t *)L.lcol[1 + p])[0], ((const uint32_t *)L.lcol[1 + p])[1]};
		const int corner = T(-1);
		int Hh = 0, V = 0;
		for (int i = 0; i < 4; i++) {
			Hh += (i + 1) * (byte_of(tw, 4 + i) - (i == 3 ? corner : byte_of(tw, 2 - i)));
			V += (i + 1) * (byte_of(lw, 4 + i) - (i == 3 ? corner : byte_of(lw, 2 - i)));
		}
		int a = 16 * (byte_of(lw, 7) + byte_of(tw, 7)), b = (34 * Hh + 32) >> 6, c = (34 * V + 32) >> 6;
		return clip255((a + b * (x - 3) + c * (y - 3) + 16) >> 5); }
	}
#undef T
#undef Lf
}

// ---------------------------------------------------------------------------------
// reconstruction of one macroblock by one wave
// ---------------------------------------------------------------------------------
// WHICH: 1 = inter and PCM macroblocks only (no dependency inside the frame), 2 = intra only, 3 = all
// coefs: this macroblock's payload (the caller has waited for it: coef_dma_wait).
// next_rec (may be null): LDS copy of the record of the macroblock this wave reconstructs next; its header goes to mn and its
// payload is requested into coefs_next -- the transfer then has the whole reconstruction (thousands of cycles) to land instead of
// standing at the head of the next macroblock's chain.
// What a lane is in every part of recon_mb (pixel, block, tile and frame addresses) is hoisted out of the macroblock loop by the
// compiler: ~130 registers where a 16-wave workgroup has 128.  Hiding the lane number from it in ONE part makes that part derive its
// addresses again per macroblock (a few VALU instructions) and the rest fit: bit 0 = the Intra8x8 blocks, 1 = the neighbour fetch,
// 2 = the residual, 3 = the neighbour commit.  Default 5: no spill left even with the Intra4x4 loop fully unrolled
// (profiles/r03_ablations.txt item 14: the bench GOP's intra time unchanged, 4x4-only pictures +12 %).
#ifndef E264_INTRA_LAUNDER
#define E264_INTRA_LAUNDER 5
#endif
template <int WHICH, int PLANES = 3>
E264_DEV void recon_mb(WaveLds &L, const FrameCtx &f, const MbInfo &m, int mbx, int mby, int lane, const uint32_t *i4tab, const int16_t *coefs,
	int16_t *coefs_next, const uint32_t *next_rec, MbInfo &mn PH_PARAMS)
{
	if (m.kind == E264_MB_ABSENT)
		return;
	const bool par = m.kind == E264_MB_INTER || m.kind == E264_MB_PCM;
	if ((WHICH == 1 && !par) || (WHICH == 2 && par))
		return;
	cslice_t s = f.slices + m.slice;
	const gu8 *pl = f.payload + m.payload_off;
	// pixel layout
	const int k = lane >> 2, r = lane & 3;
	const int X = BXf(k), Yr = BYf(k) + r;
	gu8 *dY = f.cur + (size_t)(mby * 16 + Yr) * f.sY + mbx * 16 + X;
	const int cpl = lane >> 5, cy = (lane >> 2) & 7, cx = (lane & 3) * 2;
	gu8 *dC = plane_base(f, f.cur, 1 + cpl) + (size_t)(mby * 8 + cy) * f.sC + mbx * 8 + cx;

	if (m.kind == E264_MB_PCM) { // edge264_slice.c:914-935
		*(gu32 *)dY = *(const gu32 *)(pl + Yr * 16 + X);
		*(gu16 *)dC = *(const gu16 *)(pl + 256 + cpl * 64 + cy * 8 + cx);
		return;
	}
	const uint32_t modes_lo = m.modes_lo, modes_hi = m.modes_hi;
	IntraNb nbv;
	nbv.oky = nbv.okc = false;
	if (WHICH != 1 && m.kind != E264_MB_INTER)
		nbv = issue_intra_neighbours(f, mbx, mby, (E264_INTRA_LAUNDER & 2) ? relane(lane) : lane); // in flight during the residual
	PH(2);
	if (next_rec) { // (uniform)
		mn = mb_from_lds(next_rec);
		coef_dma(coefs_next, f, mn, lane);
	}
	slice_cache(L, f, m.slice, lane);
	PH(3);
	compute_residual<PLANES>(L, coefs, f, m, WHICH == 2 ? false : m.kind == E264_MB_INTER, (E264_INTRA_LAUNDER & 4) ? relane(lane) : lane);
	PH(4);

	int pY[4], pC[2];
	bool tile_luma = false;
	if (WHICH != 1 && m.kind != E264_MB_INTER) {
		commit_intra_neighbours(L, nbv, (E264_INTRA_LAUNDER & 8) ? relane(lane) : lane);
		PH(5);
		if (!(PLANES & 1)) {
			// (chroma only: the luma samples are another workgroup's)
		} else if (m.kind == E264_MB_I16x16) {
			intra16x16_pred(L, m.i16_mode, X, Yr, pY);
		} else if (m.kind == E264_MB_I4x4) { // edge264_slice.c:615-635: predict, add residual, next block
			tile_luma = true;
			// The 16 blocks are decoded in 10 steps instead of 16: block (x,y) of the 4x4 grid only needs its left, top,
			// top-left and (when the standard counts it as available, i.e. when it precedes in zig-zag order) top-right
			// neighbours, all of which belong to earlier anti-diagonals x + 2y.  Lanes 0..15 take the first block of a
			// diagonal, lanes 16..31 the second one; the modes are already resolved against availability by the parser.
			// zig-zag indices per step: (0,-)(1,-)(4,2)(5,3)(6,8)(7,9)(12,10)(13,11)(14,-)(15,-)
			// Which block a half takes at step t is a CONSTANT of the unrolled loop, so its mode is a scalar field extract of the header and
			// the lane only selects between the two halves' values (round 4; the per-lane 64-bit shifts of an order word cost ~8 VALU per step).
			static constexpr int F4[10] = {0, 1, 4, 5, 6, 7, 12, 13, 14, 15}, S4[10] = {0, 0, 2, 3, 8, 9, 10, 11, 0, 0}; // (second half idle: block 0, valid addresses, nothing stored)
			const bool second = lane >= 16;
			const int hl16 = lane & 15, px = hl16 & 3, py = hl16 >> 2;
			auto mode_of = [&](int blk) { return (int)(((blk < 8 ? modes_lo : modes_hi) >> (4 * (blk & 7))) & 15); };
			int mode_n = second ? mode_of(S4[0]) : mode_of(F4[0]);
			uint32_t e_n = i4tab[mode_n * 16 + hl16];
#pragma unroll
			for (int t = 0; t < 10; t++) {
				const bool on = (t >= 2 && t <= 7) ? lane < 32 : lane < 16;
				const int mode = mode_n;
				const uint32_t e = e_n;
				if (t < 9) { // next step's table word: in flight during this step's sample reads
					mode_n = second ? mode_of(S4[t + 1]) : mode_of(F4[t + 1]);
					e_n = i4tab[mode_n * 16 + hl16];
				}
				const int X0 = second ? BXf(S4[t]) : BXf(F4[t]), Y0 = second ? BYf(S4[t]) : BYf(F4[t]);
				int v = 0;
				if (on) {
					const int rres = L.res[(Y0 + py) * 16 + X0 + px]; // requested in front of the prediction's reads, not behind its branches
					v = intra4x4_tab(L, e, X0, Y0, mode);
					v = clip255(w16(v + rres));
				}
				wave_sync();
				if (on)
					L.YT(Y0 + py, X0 + px) = (uint8_t)v;
				wave_sync();
			}
		} else { // I8x8, edge264_slice.c:645-668
			tile_luma = true;
			const uint32_t *i8tab = i4tab + 14 * 16;
			const int l8 = (E264_INTRA_LAUNDER & 1) ? relane(lane) : lane;
			uint32_t e_n = i8tab[i8_nibble(I8_SPEC_LO, I8_SPEC_HI, (int)(modes_lo & 255)) * 64 + l8];
#pragma unroll 1
			for (int b = 0; b < 4; b++) {
				const uint32_t e = e_n;
				if (b < 3) // the next block's table word depends on its mode only: in flight during this block
					e_n = i8tab[i8_nibble(I8_SPEC_LO, I8_SPEC_HI, (int)(modes_lo >> (8 * b + 8) & 255)) * 64 + l8];
				intra8x8_block(L, e, BXf(b * 4), BYf(b * 4), (int)(modes_lo >> (8 * b) & 255), l8);
			}
		}
		PH(6);
		if (PLANES & 2) {
			pC[0] = intra_chroma_px(L, cpl, m.chroma_mode, cx, cy);
			pC[1] = intra_chroma_px(L, cpl, m.chroma_mode, cx + 1, cy);
		}
	}
	// add residual, clip, store (int16 wrap add then packus: residual.c:160-171)
	if (PLANES & 1) {
		uint32_t outw;
		if (tile_luma) {
			outw = *(const uint32_t *)&L.YT(Yr, X);
		} else {
			const int16_t *rr = L.res + Yr * 16 + X;
			outw = (uint32_t)clip255(w16(pY[0] + rr[0])) | (uint32_t)clip255(w16(pY[1] + rr[1])) << 8 |
				(uint32_t)clip255(w16(pY[2] + rr[2])) << 16 | (uint32_t)clip255(w16(pY[3] + rr[3])) << 24;
		}
		PH(7);
		*(gu32 *)dY = outw;
	}
	if (PLANES & 2) {
		const int16_t *rc = L.res + 256 + cpl * 64 + cy * 8 + cx;
		*(gu16 *)dC = (uint16_t)(clip255(w16(pC[0] + rc[0])) | clip255(w16(pC[1] + rc[1])) << 8);
	}
}

// ---------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------
E264_DEV int lds_load_relaxed(const int *p)
{
	return E264_PROGRESS_LOAD(p);
}


#define E264_MAX_ROWS 1056
// A row may reconstruct macroblock x once the row above has finished x + 1: rows that start the moment they may run exactly two
// macroblocks behind each other, with no slack -- whenever any row of the wavefront is late (a slower kind of macroblock, a lost
// arbitration) every row below it waits, and the picture advances at the pace of the slowest row of each step.  A row's FIRST
// dependent macroblock therefore waits for this many macroblocks more: distance that later absorbs the jitter.
#ifndef E264_INTRA_SLACK
#define E264_INTRA_SLACK 0
#endif

// the workgroup's LDS: one object, the LDS-DMA targets first (lowest LDS addresses)
template <int NW>
struct __attribute__((aligned(16))) IntraLds {
	int16_t coef[NW][2][512];            // per wave: the payload of its macroblock and of the next one: [luma DC 16][chroma DC 8][coded blocks], as in the packet
	uint32_t hdrs[NW][64 * 8];           // E264Mb records of the 64 macroblocks being scanned, per wave
	WaveLds w[NW];
	int progress[E264_MAX_ROWS];         // macroblocks finished per row
	int next_row;                        // the next macroblock row nobody has taken yet (rows beyond the first NW are handed out as waves finish theirs)
	uint32_t i4tab[14 * 16 + 9 * 64];    // c_i4tab, then c_i8tab: read with a per-lane index
};

// Which row a wave takes after row y.  Round 6: FIRST COME, FIRST SERVED (an LDS counter) instead of y + NW.  On an I picture every row costs the same
// and nothing changes; on a P / B picture the intra macroblocks are unevenly spread over the rows (bench GOP: 6 +- 2.4 per row), a wave with a static
// share of 4 - 5 rows is 30 % over the mean often enough that the picture waits for it.  Rows are still STARTED in increasing order (the counter), so
// the row a wave may have to wait for (y - 1) has always been taken by somebody: no deadlock.
#ifndef E264_INTRA_DYNROWS
#define E264_INTRA_DYNROWS 1
#endif
#ifndef E264_ROW_TAKE
#define E264_ROW_TAKE(p) __hip_atomic_fetch_add((p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif
template <int NW>
E264_DEV int intra_next_row(int *next_row, int y, int lane)
{
	if (!E264_INTRA_DYNROWS || NW == 1)
		return y + NW;
	int ny = 0;
	if (lane == 0) ny = E264_ROW_TAKE(next_row);
	return E264_FIRST(ny);
}

// what thread tid of the picture's workgroup does
template <int NW, int PLANES = 3>
E264_DEV void intra_kernel_body(IntraLds<NW> &S, const E264Job &job, const int tid, const bool use_bitmap = true)
{
	WaveLds *const lds = S.w;
	uint32_t (*const hdrs)[64 * 8] = S.hdrs;
	int *const progress = S.progress;
	uint32_t *const i4tab = S.i4tab;
	const int lane = tid & 63;
	const int wave = E264_FIRST(tid >> 6);
	FrameCtx f;
	if (!open_frame(f, job))
		return;
	if (f.h->n_coded_mbs == f.h->n_inter_mbs)
		return; // nothing intra in this frame (PCM is handled by the parallel kernel but counted as coded: rare)
#if defined(E264_ABL_INTRA_STOP) && E264_ABL_INTRA_STOP == 1 // timing ablation: what the launch alone costs
	if (f.wm > 0) return;
#endif
	for (int i = tid; i < f.hm; i += NW * 64)
		progress[i] = 0;
	if (tid == 0) S.next_row = NW;
	for (int i = tid; i < 14 * 16 + 9 * 64; i += NW * 64)
		i4tab[i] = i < 14 * 16 ? c_i4tab[i] : c_i8tab[i - 14 * 16];
	E264_WG_SYNC();
#if defined(E264_ABL_INTRA_STOP) && E264_ABL_INTRA_STOP == 2 // timing ablation: launch + tables + barrier
	if (f.wm > 0) return;
#endif
	WaveLds &L = lds[wave];
	if (lane == 0) L.ws_slice = -1;
	wave_sync();
	const gu8 *mbs_g = f.mbs_g; // the E264Mb array through a per-lane (global) pointer
	const gu16 *bitmap = (f.dbk && use_bitmap) ? (const gu16 *)(f.dbk + E264_BITMAP_OFF(f.wm * f.hm)) : nullptr;
	const int ntx16 = (f.wm + 15) >> 4;
	PH_DECL;
#pragma unroll 1
	for (int y = wave; y < f.hm; y = intra_next_row<NW>(&S.next_row, y, lane)) {
		bool row_start = true; // (uniform) the row's first dependent macroblock has not started yet: see E264_INTRA_SLACK
		// the row is scanned 64 macroblocks at a time (one vector load + ballot) instead of one scalar load per
		// macroblock: in P/B frames, where few macroblocks are intra, the scan WAS the kernel's run time
#pragma unroll 1
		for (int x0 = 0; x0 < f.wm; x0 += 64) {
			const int xl = x0 + lane;
			// Round 6: the prediction kernel of this submission has left one bit per macroblock "intra, and this packet's" in the stream's
			// scratch (e264_kernels.h E264_BITMAP_OFF; it reads every record anyway).  A chunk without any is left alone after one 2-byte load
			// per lane -- on P / B pictures of encoder-made streams (1 - 4 intra macroblocks per picture) the scan of 8 160 records per picture,
			// two 16-byte loads per lane and chunk, WAS this kernel: 0.11 ms per launch for nothing.  No scratch (host tests): every chunk is scanned.
			if (bitmap) {
				const bool mine = xl < f.wm && (bitmap[(size_t)y * ntx16 + (xl >> 4)] >> (xl & 15) & 1);
				if (E264_BALLOT(mine) == 0) {
					if (lane == 0)
						E264_PROGRESS_STORE(&progress[y], min(x0 + 64, f.wm));
					continue;
				}
			}
			// whole records of the chunk -> LDS (2 x 16 bytes per lane); `kind` for the ballot comes out of the first dword
			v4u h0v = {0, 0, 0, 0}, h1v = {0, 0, 0, 0};
			uint32_t kup = E264_MB_ABSENT; // kind of the macroblock above
			if (xl < f.wm) {
				const gv4u *rp = (const gv4u *)(mbs_g + (size_t)(y * f.wm + xl) * sizeof(E264Mb));
				h0v = rp[0]; h1v = rp[1];
				if (y > 0) kup = *(const gu32 *)(mbs_g + (size_t)((y - 1) * f.wm + xl) * sizeof(E264Mb));
			}
			wave_sync(); // the previous chunk's records are no longer read
			*(v4u *)&hdrs[wave][lane * 8] = h0v;
			*(v4u *)&hdrs[wave][lane * 8 + 4] = h1v;
			wave_sync();
			if (kup >> 8 & E264_MBF_DONE) kup = E264_MB_ABSENT; // written by an earlier packet of the picture: stable
			kup &= 255u;
			const int kind = (xl < f.wm && !(h0v.x >> 8 & E264_MBF_DONE)) ? (int)(h0v.x & 255) : E264_MB_ABSENT;
			unsigned long long todo = E264_BALLOT(kind == E264_MB_I4x4 || kind == E264_MB_I8x8 || kind == E264_MB_I16x16);
			// Which macroblocks of the row above does THIS kernel write?  Only those make a macroblock below wait: inter and
			// PCM neighbours were finished by the prediction kernel before this launch (P / B pictures: an isolated intra
			// macroblock starts at once instead of queueing behind every intra macroblock up and to the left of it).
			const unsigned long long upi = E264_BALLOT(kup == E264_MB_I4x4 || kup == E264_MB_I8x8 || kup == E264_MB_I16x16);
			const int xe = min(x0 + 64, f.wm);
			// bit rx: one of the three macroblocks above macroblock x0 + rx -- x-1 (corner), x, x+1 (top right) -- is intra; outside the chunk: assume so
			const unsigned long long depmask = upi | upi << 1 | upi >> 1 | (x0 > 0 ? 1ull : 0ull) | (x0 + 64 < f.wm ? 1ull << 63 : 0ull);
			if (todo == 0 || (int)__builtin_ctzll(todo) > 0) { // macroblocks before the first intra one need nothing from this kernel
				const int upto = todo ? x0 + (int)__builtin_ctzll(todo) : xe;
				if (lane == 0)
					E264_PROGRESS_STORE(&progress[y], upto);
			}
			// software pipeline over the intra macroblocks of the chunk: the header and payload of the next one are fetched
			// inside recon_mb of the current one
			MbInfo mi, mn;
			int buf = 0;
			if (todo) {
				mi = mb_from_lds(&hdrs[wave][__builtin_ctzll(todo) * 8]);
				coef_dma(S.coef[wave][buf], f, mi, lane); // the payload does not depend on the neighbours: in flight during the wait below
			}
#pragma unroll 1
			while (todo) {
				const int x = x0 + (int)__builtin_ctzll(todo);
				todo &= todo - 1;
				const uint32_t *next_rec = todo ? &hdrs[wave][__builtin_ctzll(todo) * 8] : nullptr;
				PH(0);
				const bool dep = depmask >> (x - x0) & 1;
#ifdef E264_ABL_INTRA_NOWAIT // timing ablation: nobody waits for the row above (wrong samples): what the wavefront order itself costs
				if (false) {
#else
				if (y > 0 && dep) {
#endif
					int want = min(x + 2 + (row_start ? E264_INTRA_SLACK : 0), f.wm);
					row_start = false;
					while (lds_load_relaxed(&progress[y - 1]) < want)
						E264_SLEEP();
					E264_FENCE_ACQUIRE();
				}
				PH(1);
				coef_dma_wait();
				recon_mb<2, PLANES>(L, f, mi, x, y, lane, i4tab, S.coef[wave][buf], S.coef[wave][buf ^ 1], next_rec, mn PH_ARGS);
				PH(8);
#ifndef E264_ABL_INTRA_NOFENCE // timing ablation: progress published without waiting for the stores
				E264_FENCE_RELEASE();
#endif
				PH(9);
				// finished: everything up to the next intra macroblock of the chunk (or the chunk's end)
				const int upto = todo ? x0 + (int)__builtin_ctzll(todo) : xe;
				if (lane == 0)
					E264_PROGRESS_STORE(&progress[y], upto);
				if (next_rec) { mi = mn; buf ^= 1; }
			}
		}
	}
#ifdef E264_PHASE_INTRA
	PH_FLUSH_DBK(lane);
#endif
}

} // namespace
#endif