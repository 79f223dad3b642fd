// e264_kernels.hip -- gfx950 (MI355X) macroblock reconstruction kernels.
//
// Consumes the command packet of include/edge264_cmd.h and writes the planar YUV DPB.
// Device restatement of the reference's sample path (file:line in /root/reference/src):
//   residual   edge264_residual.c:108-538   (dequant + 4x4 / 8x8 integer IDCT, DC transforms)
//   intra      edge264_intra.c:291-765      (14 + 32 + 7 + 7 internal modes)
//   inter      edge264_inter.c:416-1251     (6-tap luma, bilinear chroma, 5 weighting schemes)
//   deblock    edge264_deblock.c:927-1123   (bS / alpha / beta / tC0) and :284-895 (filters)
//
// Execution model, four launches per batch of frames (one frame of each of n streams):
//   e264_dbkparam_kernel one wave64 per 4 macroblocks: deblocking parameters (bS, alpha, beta, indexA) of EVERY
//                        macroblock from the command packet alone (nothing in the frame is read).
//   e264_mbpar_kernel    one wave64 per STRIP of E264_MBPAR_STRIP consecutive macroblocks, every strip of every frame
//                        in parallel: inter prediction + residual (+ PCM), which depend on nothing inside the frame.
//                        Software pipelined over the strip: motion two macroblocks ahead, reference windows and
//                        coefficients one ahead; output staged in LDS and written as whole 128-byte rows.
//   e264_intra_kernel    ONE WORKGROUP PER FRAME, ONE WAVE PER MACROBLOCK ROW: intra MBs only, row y
//                        may reconstruct macroblock x once row y-1 has finished macroblock x+1.
//   e264_deblock_kernel  one workgroup per frame, one HALF-wave per macroblock row, same wavefront over every MB:
//                        the 2-MB lag is exactly what H.264 in-loop deblocking requires.
// Progress counters live in LDS, so the hand-off between rows never leaves the CU: no
// agent-scope fences, no cross-XCD traffic, no placement assumption.  Chip-level parallelism
// comes from many independent streams (one frame of each per launch), the north-star workload
// (>=1000 concurrent streams).  Integer / byte work throughout, no MFMA.
//
// Pipeline rule (learnt with the phase profiler, -DE264_PHASE_TIMING / tools/gpu_phase.sh): a stage that ISSUES loads for
// a later stage must not read, clear or copy any register that may still have a load in flight -- each of those is an
// s_waitcnt vmcnt(0), i.e. a wait for the loads it has just issued.  Hence: prefetch helpers contain loads only, nothing
// is zero-initialised in front of a conditional load, one code path fills the pipeline registers, and the place where
// the prefetches are consumed says so with an explicit vmcnt(0).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/edge264_cmd.h"
#include "e264_kernels.h"
#include "e264_dev.h"
#include "e264_pred.h"
#include "e264_dbkp.h"

namespace {
// ---------------------------------------------------------------------------------
// per-wave LDS scratch
// ---------------------------------------------------------------------------------
// luma tile: rows -1..15, columns -8..23 (top-right of an 8x8 block reaches x=23), 32-byte rows
#define YT_STRIDE 32
#define YT(y, x) ytile[((y) + 1) * YT_STRIDE + (x) + 8]
// chroma tiles: rows -1..7, columns -4..11
#define CT_STRIDE 16
#define CT(p, y, x) ctile[p][((y) + 1) * CT_STRIDE + (x) + 4]
// deblock tiles: luma rows -4..15, cols -4..15 (stride 20 bytes, dword aligned); chroma rows -4..7, cols -4..7
#define DY_STRIDE 20
#define DYT(y, x) dytile[((y) + 4) * DY_STRIDE + (x) + 4]
#define DC_STRIDE 12
#define DCT(p, y, x) dctile[p][((y) + 4) * DC_STRIDE + (x) + 4]

struct __attribute__((aligned(16))) WaveLds { // reconstruction scratch of one wave
	int16_t res[384];          // residual: luma [y*16+x], Cb at 256 [y*8+x], Cr at 320
	int32_t tmp[256];          // IDCT intermediate (4x4: 16 blocks x 16 int32; 8x8: int16 view)
	int32_t dc[24];            // 16 luma DC (zig order), 4 Cb, 4 Cr
	uint8_t ytile[17 * YT_STRIDE];
	uint8_t ctile[2][9 * CT_STRIDE];
	uint8_t ftop[32];          // intra 8x8 filtered top  ft[-1..15] at [i+1]
	uint8_t fleft[8];          // intra 8x8 filtered left
	uint32_t win[432];         // reference windows of inter prediction (one list at a time): 1 x 21x24, 4 x 13x16 or 16 x 9x12 bytes
	// residual inputs, staged so that the transforms never wait for memory (see coef_issue / slice_cache)
	__attribute__((aligned(4))) int16_t coef[408]; // the macroblock's payload: [luma DC 16][chroma DC 8][coded blocks], as in the packet
	__attribute__((aligned(4))) uint8_t ws[224];   // scaling lists of the cached slice: weightScale4x4[6][16], weightScale8x8[0..1][64]
	int ws_slice;              // slice index the cache holds (-1: none)
	int ws_idc;                // its weighted_bipred_idc
};

#ifndef DBK_RING
#define DBK_RING 8 // macroblocks of bottom rows each row keeps in LDS for the row below (power of two)
#endif
#ifndef DBK_LAG
#define DBK_LAG 2  // the second row of a wave trails the first by this many macroblocks (>= 2; measured 2: 2.03 ms, 3: 2.09, 4: 2.06)
#endif
struct __attribute__((aligned(16))) DbkTile {
	uint8_t dytile[20 * DY_STRIDE];
	uint8_t dctile[2][12 * DC_STRIDE];
	uint8_t prm[E264_DBK_BYTES]; // deblocking parameters of the current macroblock
};
// Hand-off to the macroblock row below: the last 4 luma rows / 2 chroma rows of the most recent
// DBK_RING macroblocks, final for the row below once this row is 2 macroblocks further.
struct __attribute__((aligned(16))) DbkRing {
	uint32_t y[4][DBK_RING * 4];      // [row 12..15][mb slot * 4 + dword]
	uint32_t c[2][2][DBK_RING * 2];   // [plane][row 6..7][mb slot * 2 + dword]
};
// Samples that have become FINAL, collected per group of 4 macroblocks and written as whole 64-byte row
// pieces (16 bytes per lane, 4 consecutive lanes per row).  Writing every macroblock as it is filtered (16-byte
// rows, then its top rows and left columns again one step later) cost 5.7x the frame size in partial
// 32/64-byte memory-side write requests (TCC_EA0_WRREQ, profiles/r01_pmc_calibration.txt).
// What is final after macroblock x of a row: the rows above it (-4..-1, filtered by its top edge) over its 16
// columns, and its rows 0..11 over columns -4..11 (the left neighbour's last 4 columns now have their right
// edge filtered).  Rows 12..15 belong to the row below, except where that row cannot take them from the
// LDS ring (last row of the frame; last row of a round, handed to wave 0 through memory).
struct __attribute__((aligned(16))) DbkStage {
	uint32_t y[20][16];    // luma rows -4..15 (index row + 4) x 64 columns
	uint32_t c[2][10][8];  // chroma planes, rows -2..7 (index row + 2) x 32 columns
};
// Input of a PAIR of macroblocks of a row (unfiltered samples, deblocking parameters, header words), fetched with one
// 16-byte load per lane and piece -- 2 load instructions per pair -- instead of six 4-byte loads per lane and MACROBLOCK:
// the vector memory path spends 1.5 - 2.5 cycles on every lane of a non-contiguous load whatever its width
// (tools/calib/load_rate.hip), and with 384 lane-loads per step that, not the filter arithmetic, set the pace.
struct __attribute__((aligned(16))) DbkIn {
	uint32_t y[16][8];    // luma rows 0..15 x 2 macroblocks
	uint32_t c[2][8][4];  // chroma planes, rows 0..7
	uint32_t prm[2][16];  // E264_DBK_BYTES per macroblock
	uint32_t hdr[2][4];   // first 16 bytes of the E264Mb records
};
struct __attribute__((aligned(16))) DbkLds { // deblocking scratch of one wave = two macroblock rows
	DbkTile tile[2];     // [half-wave]
	DbkRing ring[2][2];  // [parity of the row-pair round][half-wave]
	DbkStage stage[2];   // [half-wave]
	DbkIn in[2];         // [half-wave]
};

// -DE264_PHASE_TIMING: wall cycles of the mbpar kernel's phases, summed over all waves (tools/gpu_phase.sh reads them back
// through e264_debug_phase_cycles).  s_memtime at the phase boundaries drains the LGKM counter, so the numbers are a
// profile, not a benchmark.
#ifdef E264_PHASE_TIMING
__device__ unsigned long long g_phase[32]; // [0..13] mbpar kernel, [16..29] deblock kernel
#define PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PH(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); ph_acc[k] += t_ - ph_t; ph_t = t_; } while (0)
#define PH_FLUSH(lane) do { if ((lane) == 0) for (int k_ = 0; k_ < 14; k_++) atomicAdd(&g_phase[k_], ph_acc[k_]); } while (0)
#define PH_FLUSH_DBK(lane) do { if ((lane) == 0) for (int k_ = 0; k_ < 14; k_++) atomicAdd(&g_phase[16 + k_], ph_acc[k_]); } while (0)
#define PH_PARAMS , unsigned long long &ph_t, unsigned long long (&ph_acc)[14]
#define PH_ARGS , ph_t, ph_acc
#else
#define PH_DECL
#define PH(k)
#define PH_FLUSH(lane)
#define PH_FLUSH_DBK(lane)
#define PH_PARAMS
#define PH_ARGS
#endif

// register copy of one macroblock header (uniform: fetched with scalar loads)
struct MbInfo {
	int kind, flags, chroma_mode, i16_mode, slice;
	uint8_t qp[3];
	uint32_t coded, payload_off, modes_lo, modes_hi;
};
__device__ __forceinline__ MbInfo load_mb(cmb_t p)
{
	MbInfo m;
	m.kind = p->kind; m.flags = p->flags; m.chroma_mode = p->chroma_mode; m.i16_mode = p->i16_mode; m.slice = p->slice;
	m.qp[0] = p->qp[0]; m.qp[1] = p->qp[1]; m.qp[2] = p->qp[2];
	m.coded = p->coded; m.payload_off = p->payload_off;
	m.modes_lo = *(const uint32_t __attribute__((address_space(4))) *)&p->modes[0];
	m.modes_hi = *(const uint32_t __attribute__((address_space(4))) *)&p->modes[4];
	return m;
}

// The same header out of a lane-distributed copy of 8 consecutive E264Mb records (lane = mb * 8 + dword, one
// vector load per strip): v_readlane instead of scalar-memory loads.  Scalar loads share the LGKM counter with
// LDS and return out of order, so every LDS wait of the inner loop also waited for the headers "prefetched" for
// later macroblocks (ablation: 40% of the kernel's time was spent with all filters switched off).
__device__ __forceinline__ MbInfo mb_from_lanes(uint32_t hv, int i)
{
	MbInfo m;
	const uint32_t d0 = __builtin_amdgcn_readlane(hv, i * 8), d1 = __builtin_amdgcn_readlane(hv, i * 8 + 1);
	const uint32_t d2 = __builtin_amdgcn_readlane(hv, i * 8 + 2);
	m.kind = d0 & 255; m.flags = d0 >> 8 & 255; m.qp[0] = d0 >> 16 & 255; m.qp[1] = d0 >> 24;
	m.qp[2] = d1 & 255; m.chroma_mode = d1 >> 8 & 255; m.i16_mode = d1 >> 16 & 255;
	m.slice = d2 >> 16;
	m.coded = __builtin_amdgcn_readlane(hv, i * 8 + 3); m.payload_off = __builtin_amdgcn_readlane(hv, i * 8 + 4);
	m.modes_lo = __builtin_amdgcn_readlane(hv, i * 8 + 5); m.modes_hi = __builtin_amdgcn_readlane(hv, i * 8 + 6);
	return m;
}

// The same header out of an LDS copy of the record (8 dwords), all lanes reading the same words (intra kernel: the
// headers of 64 macroblocks of the row are fetched with two vector loads per lane instead of one scalar-memory round
// trip per macroblock in the middle of the dependency chain)
__device__ __forceinline__ MbInfo mb_from_lds(const uint32_t *rec)
{
	MbInfo m;
	const uint32_t d0 = __builtin_amdgcn_readfirstlane(rec[0]), d1 = __builtin_amdgcn_readfirstlane(rec[1]);
	const uint32_t d2 = __builtin_amdgcn_readfirstlane(rec[2]);
	m.kind = d0 & 255; m.flags = d0 >> 8 & 255; m.qp[0] = d0 >> 16 & 255; m.qp[1] = d0 >> 24;
	m.qp[2] = d1 & 255; m.chroma_mode = d1 >> 8 & 255; m.i16_mode = d1 >> 16 & 255;
	m.slice = d2 >> 16;
	m.coded = __builtin_amdgcn_readfirstlane(rec[3]); m.payload_off = __builtin_amdgcn_readfirstlane(rec[4]);
	m.modes_lo = __builtin_amdgcn_readfirstlane(rec[5]); m.modes_hi = __builtin_amdgcn_readfirstlane(rec[6]);
	return m;
}

// ---------------------------------------------------------------------------------
// residual: fills lds.res for the whole macroblock
// ---------------------------------------------------------------------------------
// One 4x4 block per 4 lanes.  pass 1 (lane = block k, row y): dequant + horizontal butterfly
// (edge264_residual.c:118-134); pass 2 (lane = block k, column x'): vertical butterfly, >>6,
// saturate to int16 (residual.c:141-158).  dc_only blocks get the add_dc4x4 value (residual.c:174-187).
__device__ __forceinline__ void idct4x4_blocks(WaveLds &L, int nblk, uint32_t codedmask, bool use_dc, bool dc_valid,
	const int16_t *coef_base, const uint8_t *wS, int qP, int dc_off, int res_off, int res_stride, int lane)
{
	int k = lane >> 2, y = lane & 3;
	bool active = k < nblk;
	bool coded = active && (codedmask >> k & 1);
	if (coded) {
		// coefficient blocks are packed in increasing k: offset = popcount of lower coded bits
		const int16_t *c = coef_base + __builtin_popcount(codedmask & ((1u << k) - 1)) * 16;
		int sh = qP / 6, m = qP - sh * 6;
		int d[4];
#pragma unroll
		for (int x = 0; x < 4; x++) {
			int pos = x * 4 + y;
			int LS = wS[pos] * norm4(m, pos);
			d[x] = (int)(((uint32_t)((int)c[pos] * LS) << sh) + 8u) >> 4;
		}
		if (use_dc && y == 0)
			d[0] = L.dc[dc_off + k];
		int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = (d[3] >> 1) + d[1];
		int32_t *t = L.tmp + k * 16;
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
			const int32_t *t = L.tmp + k * 16 + x * 4;
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
__device__ __forceinline__ void idct8_1d(int16_t d[8])
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

__device__ __forceinline__ void idct8x8_blocks(WaveLds &L, uint32_t coded, const int16_t *coef_base, const uint8_t *wS, int qP, int lane)
{
	int b = lane >> 3, j = lane & 7;
	bool on = lane < 32 && (coded >> (b * 4) & 1);
	int16_t *t16 = (int16_t *)L.tmp;
	if (on) {
		int nb = 0;
		for (int i = 0; i < b; i++) nb += coded >> (i * 4) & 1;
		const int16_t *c = coef_base + nb * 64;
		int div = qP / 6, m = qP - div * 6;
		int16_t d[8];
#pragma unroll
		for (int i = 0; i < 8; i++) {
			int pos = i * 8 + j;
			int LS = wS[pos] * norm8(m, pos);
			if (div < 6)
				d[i] = (int16_t)sat16(((int)c[pos] * LS + (1 << (5 - div))) >> (6 - div));
			else
				d[i] = (int16_t)((int)c[pos] * (int)(int16_t)(LS << (div - 6)));
		}
		idct8_1d(d);
		// transposed read in pass 2: element [i][j]; +32 lands on the new vector 0 = all elements with j == 0
#pragma unroll
		for (int i = 0; i < 8; i++)
			t16[b * 64 + i * 8 + j] = (int16_t)(d[i] + (j == 0 ? 32 : 0));
	}
	wave_sync();
	if (on) {
		int i = j; // this lane now owns vector-lane i = pixel column i
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
	wave_sync();
}

// LDS copy of the weighting part of E264SliceParams (the three tables are contiguous there), filled by slice_cache when the
// slice weights anything: select_weights runs in the middle of a macroblock, after the next macroblock's loads have been
// issued -- a table read from memory at that point waits for all of them.
struct __attribute__((aligned(4))) SliceW {
	int16_t explicit_weights[3][64];
	int8_t explicit_offsets[3][64];
	uint8_t implicit_weights[32][32];
	int8_t weighted_bipred_idc, luma_log2_weight_denom, chroma_log2_weight_denom, pad;
};
static_assert(offsetof(E264SliceParams, explicit_offsets) == offsetof(E264SliceParams, explicit_weights) + 384 &&
              offsetof(E264SliceParams, implicit_weights) == offsetof(E264SliceParams, explicit_weights) + 576 &&
              offsetof(E264SliceParams, explicit_weights) % 4 == 0, "SliceW mirrors a contiguous range of E264SliceParams");
// Residual inputs without a memory round trip inside the transforms:
//   coef_issue / coef_commit  the macroblock's payload (<= 816 bytes) as 4 coalesced dword loads per lane, issued
//                             while something else runs (mbpar: one macroblock ahead; intra: before the wait for the
//                             row above), then dropped into L.coef.  (Before: 2-byte loads at transform positions,
//                             used at once: one exposed round trip per transform call.)
//   slice_cache               scaling lists + weighted_bipred_idc of the current slice in LDS, reloaded when the
//                             slice index changes.
struct CoefPf { uint32_t v0, v1, v2, v3; };
__device__ __forceinline__ int coef_dwords(const MbInfo &m)
{
	if (m.kind == E264_MB_ABSENT || m.kind == E264_MB_PCM || m.coded == 0)
		return 0;
	const uint32_t c = m.coded;
	int bytes = ((c & E264_CODED_LUMA_DC) ? 32 : 0) + ((c & E264_CODED_CHROMA_DC) ? 16 : 0) + __builtin_popcount(c >> 16 & 0xff) * 32;
	bytes += (m.kind != E264_MB_I16x16 && (m.flags & E264_MBF_T8x8)) ? __builtin_popcount(c & 0x1111) * 128 : __builtin_popcount(c & 0xffff) * 32;
	return bytes >> 2;
}
__device__ __forceinline__ void coef_issue(const FrameCtx &f, const MbInfo &m, int lane, CoefPf &pf)
{ // loads only: nothing here may use a loaded value
	const int ndw = coef_dwords(m);
	pf.v0 = pf.v1 = pf.v2 = pf.v3 = 0;
	if (ndw == 0)
		return;
	const gu32 *src = (const gu32 *)(f.payload + m.payload_off);
	if (lane < ndw) pf.v0 = src[lane];
	if (64 + lane < ndw) pf.v1 = src[64 + lane];
	if (128 + lane < ndw) pf.v2 = src[128 + lane];
	if (192 + lane < ndw) pf.v3 = src[192 + lane];
}
__device__ __forceinline__ void coef_commit(WaveLds &L, const MbInfo &m, int lane, const CoefPf &pf)
{ // the caller synchronises the wave before the transforms read L.coef (compute_residual does)
	const int ndw = coef_dwords(m);
	if (ndw == 0)
		return;
	uint32_t *c = (uint32_t *)L.coef;
	if (lane < ndw) c[lane] = pf.v0;
	if (64 + lane < ndw) c[64 + lane] = pf.v1;
	if (128 + lane < ndw) c[128 + lane] = pf.v2;
	if (192 + lane < ndw) c[192 + lane] = pf.v3;
}
__device__ __forceinline__ void slice_cache(WaveLds &L, const FrameCtx &f, int slice, int lane, SliceW *W = nullptr)
{
	if (__builtin_amdgcn_readfirstlane(L.ws_slice) == slice) // uniform
		return;
	cslice_t s = f.slices + slice;
	wave_sync();
	if (W && s->weighted_bipred_idc != 0) { // 400 dwords: explicit weights, offsets, implicit weights
		const gu32 *gw = (const gu32 *)((const gu8 *)s + offsetof(E264SliceParams, explicit_weights));
#pragma unroll
		for (int it = 0; it < 7; it++)
			if (it * 64 + lane < 400) ((uint32_t *)W)[it * 64 + lane] = gw[it * 64 + lane];
	}
	if (W && lane == 0) {
		W->weighted_bipred_idc = s->weighted_bipred_idc;
		W->luma_log2_weight_denom = s->luma_log2_weight_denom;
		W->chroma_log2_weight_denom = s->chroma_log2_weight_denom;
	}
	const gu32 *g4 = (const gu32 *)((const gu8 *)s + offsetof(E264SliceParams, weightScale4x4));
	const gu32 *g8 = (const gu32 *)((const gu8 *)s + offsetof(E264SliceParams, weightScale8x8));
	if (lane < 24) ((uint32_t *)L.ws)[lane] = g4[lane];
	else if (lane < 56) ((uint32_t *)L.ws)[lane] = g8[lane - 24];
	if (lane == 0) { L.ws_slice = slice; L.ws_idc = s->weighted_bipred_idc; }
	wave_sync();
}

// L.coef holds the macroblock's payload (coef_commit), the slice cache is valid (slice_cache)
__device__ __forceinline__ void compute_residual(WaveLds &L, const FrameCtx &f, const MbInfo &m, int lane)
{
	// zero the residual tile (384 int16 = 192 dwords)
	uint32_t *rz = (uint32_t *)L.res;
	rz[lane] = 0; rz[lane + 64] = 0; rz[lane + 128] = 0;
	const uint32_t coded = m.coded;
	const bool inter = m.kind == E264_MB_INTER;
	const int16_t *pl = L.coef;
	const int16_t *ldc = nullptr, *cdc = nullptr;
	if (coded & E264_CODED_LUMA_DC) { ldc = pl; pl += 16; }
	if (coded & E264_CODED_CHROMA_DC) { cdc = pl; pl += 8; }
	const int16_t *co = pl;
	const uint8_t *ws4 = L.ws, *ws8 = L.ws + 96;
	if (lane < 24) L.dc[lane] = 0;
	wave_sync();
	if (!coded) return;

	// DC transforms (residual.c:352-399 and :456-480): one output per lane
	if (ldc && lane < 16) {
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
	if (cdc && lane >= 16 && lane < 24) {
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
		if (coded & (0xffff | E264_CODED_LUMA_DC))
			idct4x4_blocks(L, 16, coded & 0xffff, true, ldc != nullptr, co, ws4, m.qp[0], 0, 0, 16, lane);
		co += __builtin_popcount(coded & 0xffff) * 16;
	} else if (m.flags & E264_MBF_T8x8) {
		if (coded & 0x1111)
			idct8x8_blocks(L, coded, co, ws8 + (inter ? 64 : 0), m.qp[0], lane);
		co += __builtin_popcount(coded & 0x1111) * 64;
	} else {
		if (coded & 0xffff)
			idct4x4_blocks(L, 16, coded & 0xffff, false, false, co, ws4 + (inter ? 3 : 0) * 16, m.qp[0], 0, 0, 16, lane);
		co += __builtin_popcount(coded & 0xffff) * 16;
	}
	// chroma: 8 blocks, Cb 0..3 then Cr 4..7; different QP / scaling list per plane
	if (coded & (0xff0000 | E264_CODED_CHROMA_DC)) {
		int k = lane >> 2;
		int pc = (k >> 2) & 1;
		idct4x4_blocks(L, 8, coded >> 16 & 0xff, true, cdc != nullptr, co, ws4 + (1 + pc + (inter ? 3 : 0)) * 16, pc ? m.qp[2] : m.qp[1], 16, 256, 8, lane);
	}
}

// ---------------------------------------------------------------------------------
// intra prediction (modes: edge264_internal.h:564-634; U(navailable) suffixes A left, B top,
// C top-right, D top-left)
// ---------------------------------------------------------------------------------
#define LP(l, m, r) (((l) + 2 * (m) + (r) + 2) >> 2)

// neighbours of the macroblock from the frame (un-deblocked, pass R) into the tiles.
// Out-of-frame positions are never dereferenced; the remapped modes never use them.
// In two steps, so that the frame reads are in flight while the residual is computed: issue (loads only, nothing may
// use the values) and commit (values -> tiles).
struct IntraNb { uint32_t y, c; bool oky, okc; };
// ONE predicated load per plane group and lane (address and predicate selected by the lane's role), nothing cleared first:
// the four role branches used to write the same two registers one after the other, which serialised them into two extra
// memory round trips per intra macroblock (s_waitcnt vmcnt(0) between the loads).
__device__ __forceinline__ IntraNb issue_intra_neighbours(const FrameCtx &f, int mbx, int mby, int lane)
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
__device__ __forceinline__ void commit_intra_neighbours(WaveLds &L, const IntraNb &n, int lane)
{ // unavailable neighbours read as 0 (never used: the parser resolved the modes against availability)
	const bool left = lane >= 32 && lane < 48;
	const uint8_t vy = n.oky ? (uint8_t)n.y : 0, vc = n.okc ? (uint8_t)n.c : 0;
	if (lane < 25) L.YT(-1, lane - 1) = vy;
	else if (left) L.YT(lane - 32, -1) = vy;
	if (lane < 18) L.CT(lane / 9, -1, lane % 9 - 1) = vc;
	else if (left) L.CT((lane - 32) >> 3, lane & 7, -1) = vc;
	wave_sync();
}

// one 4x4 block, lanes 0..15 = (y = lane>>2, x = lane&3); reads/writes the luma tile
__device__ __forceinline__ int intra4x4_px(const WaveLds &L, int X0, int Y0, int mode, int x, int y)
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

__constant__ int8_t c_i8spec[32] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 5, 6, 7, 7, 7, 7, 8, 8};
__constant__ int8_t c_i8unav[32] = {0, 4, 8, 12, 0, 8, 0, 1, 5, 9, 13, 2, 10, 4, 8, 12, 3, 0, 4, 8, 12, 0, 4, 0, 4, 0, 0, 4, 8, 12, 0, 8};

// one 8x8 block: all 64 lanes = (y = lane>>3, x = lane&7).  8.3.2.2.1 filtering into L.ftop/L.fleft first.
__device__ __forceinline__ void intra8x8_block(WaveLds &L, int X0, int Y0, int mode, int lane)
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
	wave_sync();
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
	// add residual (add_idct8x8 tail, residual.c:318-342) and write the tile
	int rres = L.res[(Y0 + y) * 16 + X0 + x];
	wave_sync();
	L.YT(Y0 + y, X0 + x) = (uint8_t)clip255(w16(v + rres));
	wave_sync();
}

// Intra 16x16 in the pixel layout: returns 4 predicted samples for (row Yr, cols X..X+3)
__device__ __forceinline__ void intra16x16_pred(const WaveLds &L, int mode, int X, int Yr, int out[4])
{
#define T(i) ((int)L.YT(-1, (i)))
#define Lf(i) ((int)L.YT((i), -1))
	switch (mode) {
	default:
	case 0: for (int i = 0; i < 4; i++) out[i] = T(X + i); return;
	case 1: for (int i = 0; i < 4; i++) out[i] = Lf(Yr); return;
	case 2: case 3: case 4: case 5: {
		int st = 0, sl = 0;
		if (mode == 2 || mode == 3) for (int i = 0; i < 16; i++) st += T(i);
		if (mode == 2 || mode == 4) for (int i = 0; i < 16; i++) sl += Lf(i);
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

// Intra chroma for plane p, sample (x,y)
__device__ __forceinline__ int intra_chroma_px(const WaveLds &L, int p, int mode, int x, int y)
{
#define T(i) ((int)L.CT(p, -1, (i)))
#define Lf(i) ((int)L.CT(p, (i), -1))
	switch (mode) {
	default:
	case 0: case 1: case 2: case 3: {
		if (mode == 3) return 128;
		int bx = x >> 2, by = y >> 2;
		int t = 0, l = 0;
		for (int i = 0; i < 4; i++) { t += T(bx * 4 + i); l += Lf(by * 4 + i); }
		if (mode == 1) return (t + 2) >> 2;
		if (mode == 2) return (l + 2) >> 2;
		if (bx == by) return (t + l + 4) >> 3;
		return bx ? (t + 2) >> 2 : (l + 2) >> 2; }
	case 4: return Lf(y);
	case 5: return T(x);
	case 6: {
		int Hh = 0, V = 0;
		for (int i = 0; i < 4; i++) {
			Hh += (i + 1) * (T(4 + i) - (i == 3 ? T(-1) : T(2 - i)));
			V += (i + 1) * (Lf(4 + i) - (i == 3 ? T(-1) : Lf(2 - i)));
		}
		int a = 16 * (Lf(7) + T(7)), b = (34 * Hh + 32) >> 6, c = (34 * V + 32) >> 6;
		return clip255((a + b * (x - 3) + c * (y - 3) + 16) >> 5); }
	}
#undef T
#undef Lf
}

// ---------------------------------------------------------------------------------
// reconstruction of one macroblock by one wave
// ---------------------------------------------------------------------------------
// WHICH: 1 = inter and PCM macroblocks only (no dependency inside the frame), 2 = intra only, 3 = all
template <int WHICH>
__device__ __forceinline__ void recon_mb(WaveLds &L, const FrameCtx &f, const MbInfo &m, int mbx, int mby, int lane, const CoefPf &pf PH_PARAMS)
{ // pf: the macroblock's payload, issued by the caller (coef_issue) as early as it could
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
		nbv = issue_intra_neighbours(f, mbx, mby, lane); // in flight during the residual
	PH(2);
	if (!(f.dbg & 1024)) {
		slice_cache(L, f, m.slice, lane);
		coef_commit(L, m, lane, pf);
		PH(3);
		compute_residual(L, f, m, lane);
	}
	PH(4);

	int pY[4], pC[2];
	bool tile_luma = false;
	if (WHICH != 1 && m.kind != E264_MB_INTER) {
		commit_intra_neighbours(L, nbv, lane);
		PH(5);
		if (m.kind == E264_MB_I16x16) {
			intra16x16_pred(L, m.i16_mode, X, Yr, pY);
		} else if (m.kind == E264_MB_I4x4) { // edge264_slice.c:615-635: predict, add residual, next block
			tile_luma = true;
			// The 16 blocks are decoded in 10 steps instead of 16: block (x,y) of the 4x4 grid only needs its left, top,
			// top-left and (when the standard counts it as available, i.e. when it precedes in zig-zag order) top-right
			// neighbours, all of which belong to earlier anti-diagonals x + 2y.  Lanes 0..15 take the first block of a
			// diagonal, lanes 16..31 the second one; the modes are already resolved against availability by the parser.
			// zig-zag indices per step: (0,-)(1,-)(4,2)(5,3)(6,8)(7,9)(12,10)(13,11)(14,-)(15,-)
			const uint64_t firsts = 0xfedc765410ull, seconds = 0xffba9832ffull; // one nibble per step
			const int half = lane >> 4, hl16 = lane & 15;
			for (int t = 0; t < 10; t++) {
				const int b = (int)((half ? seconds : firsts) >> (4 * t) & 15);
				const bool on = lane < 32 && !(half == 1 && (t < 2 || t > 7));
				const int bb = on ? b : 0;
				int mode = (int)(((bb < 8 ? modes_lo : modes_hi) >> (4 * (bb & 7))) & 15);
				int X0 = BXf(bb), Y0 = BYf(bb);
				int v = 0;
				if (on) {
					int x = hl16 & 3, y = hl16 >> 2;
					v = intra4x4_px(L, X0, Y0, mode, x, y);
					v = clip255(w16(v + L.res[(Y0 + y) * 16 + X0 + x]));
				}
				wave_sync();
				if (on)
					L.YT(Y0 + (hl16 >> 2), X0 + (hl16 & 3)) = (uint8_t)v;
				wave_sync();
			}
		} else { // I8x8, edge264_slice.c:645-668
			tile_luma = true;
			for (int b = 0; b < 4; b++)
				intra8x8_block(L, BXf(b * 4), BYf(b * 4), (int)(modes_lo >> (8 * b) & 255), lane);
		}
		PH(6);
		pC[0] = intra_chroma_px(L, cpl, m.chroma_mode, cx, cy);
		pC[1] = intra_chroma_px(L, cpl, m.chroma_mode, cx + 1, cy);
	}
	// add residual, clip, store (int16 wrap add then packus: residual.c:160-171)
	uint32_t outw;
	if (tile_luma) {
		outw = *(const uint32_t *)&L.YT(Yr, X);
	} else {
		const int16_t *rr = L.res + Yr * 16 + X;
		outw = (uint32_t)clip255(w16(pY[0] + rr[0])) | (uint32_t)clip255(w16(pY[1] + rr[1])) << 8 |
			(uint32_t)clip255(w16(pY[2] + rr[2])) << 16 | (uint32_t)clip255(w16(pY[3] + rr[3])) << 24;
	}
	PH(7);
	if (f.dbg & 4096) return;
	*(gu32 *)dY = outw;
	const int16_t *rc = L.res + 256 + cpl * 64 + cy * 8 + cx;
	*(gu16 *)dC = (uint16_t)(clip255(w16(pC[0] + rc[0])) | clip255(w16(pC[1] + rc[1])) << 8);
}

// ---------------------------------------------------------------------------------
// deblocking wavefront (edge264_deblock.c:284-895): ONE HALF-WAVE PER MACROBLOCK ROW.
// A wave owns two consecutive rows; lanes 0..31 walk the upper row, lanes 32..63 the lower row
// DBK_LAG macroblocks behind, so that all 64 lanes filter (16 luma + 8 Cb + 8 Cr lines per row).
// ---------------------------------------------------------------------------------
// One edge on 8 values p3 p2 p1 p0 | q0 q1 q2 q3 held in registers; chroma lines use the same
// code with ap/aq/strong forced off and tc = tC0+1 (deblock.c:95-152, 213-276).  Branch-free per
// lane; `strong_somewhere` is WAVE-UNIFORM: the bS==4 arithmetic (deblock.c:213-276) is only emitted
// for macroblock edges (EDGE0) and only executed when some line of the wave has bS 4.
template <bool EDGE0>
__device__ __forceinline__ void edge_filter(int &p3, int &p2, int &p1, int &p0, int &q0, int &q1, int &q2, int &q3,
	int bS, int alpha, int beta, int tc0, bool chroma, bool strong_somewhere)
{
	const int dpq = abs(p0 - q0);
	const bool go = (bS != 0) & (dpq < alpha) & (abs(p1 - p0) < beta) & (abs(q1 - q0) < beta);
	if (!go) // filterSamplesFlag (8.7.2.3): per-lane; the region below is skipped when no line of the wave passes
		return;
	const bool ap = !chroma & (abs(p2 - p0) < beta), aq = !chroma & (abs(q2 - q0) < beta);
	// bS < 4
	const int tc = tc0 + (chroma ? 1 : (int)ap + (int)aq);
	const int delta = clip3i(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3);
	const int avg = (p0 + q0 + 1) >> 1;
	const int w_p0 = clip255(p0 + delta), w_q0 = clip255(q0 - delta);
	const int w_p1 = p1 + clip3i(-tc0, tc0, (p2 + avg - 2 * p1) >> 1);
	const int w_q1 = q1 + clip3i(-tc0, tc0, (q2 + avg - 2 * q1) >> 1);
	if (EDGE0 && strong_somewhere) {
		// bS == 4
		const bool strong = bS == 4;
		const bool small = dpq < (alpha >> 2) + 2;
		const bool sp = ap & small, sq = aq & small;
		const int s_p0 = sp ? (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3 : (2 * p1 + p0 + q1 + 2) >> 2;
		const int s_p1 = (p2 + p1 + p0 + q0 + 2) >> 2;
		const int s_p2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
		const int s_q0 = sq ? (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3 : (2 * q1 + q0 + p1 + 2) >> 2;
		const int s_q1 = (p0 + q0 + q1 + q2 + 2) >> 2;
		const int s_q2 = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
		p0 = strong ? s_p0 : w_p0;
		q0 = strong ? s_q0 : w_q0;
		p1 = (strong ? sp : ap) ? (strong ? s_p1 : w_p1) : p1;
		q1 = (strong ? sq : aq) ? (strong ? s_q1 : w_q1) : q1;
		p2 = (strong & sp) ? s_p2 : p2;
		q2 = (strong & sq) ? s_q2 : q2;
	} else {
		p0 = w_p0;
		q0 = w_q0;
		p1 = ap ? w_p1 : p1;
		q1 = aq ? w_q1 : q1;
	}
}

// A "line" of 20 samples (positions -4..15) crossing the four luma edges at positions 0,4,8,12 lives
// in v[0..19].  Chroma lines (positions -4..7, edges at 0 and 4) are parked so that their two edges
// coincide with luma edges 0 and 2: v[0..5] = pos -4..1, v[10..15] = pos 2..7 (v[6..9] unused).
// An edge whose bS is 0 on every line of the wave is skipped (wave-uniform branch).
__device__ __forceinline__ void filter_line(int v[20], const int bS[4], int a_edge0, int a_in, int b_edge0, int b_in, const int tc0[4], bool chroma)
{
	if (__any(bS[0] != 0))
		edge_filter<true>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], bS[0], a_edge0, b_edge0, tc0[0], chroma, __any(bS[0] == 4));
	if (__any(bS[1] != 0))
		edge_filter<false>(v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], bS[1], a_in, b_in, tc0[1], chroma, false);
	if (__any(bS[2] != 0))
		edge_filter<false>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], bS[2], a_in, b_in, tc0[2], chroma, false);
	if (__any(bS[3] != 0))
		edge_filter<false>(v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19], bS[3], a_in, b_in, tc0[3], chroma, false);
}

// per-lane state of the macroblock a half-wave is about to filter
struct DbkRegs { uint32_t va0, va1, vc, vp, vt, hdr; bool act; };

// Loads of macroblock (mbx,mby) that do not depend on the row above.  hl = lane within the half-wave:
// own luma (2 dwords per lane), own chroma (1), parameters (hl < 16), first header dword (kind, flags).
// LOADS ONLY: nothing here uses a loaded value (the header used to be decoded here: a memory round trip in the middle
// of every step) and no register is cleared first (clearing a register that may have a load in flight is a wait too);
// lanes / cases that do not load keep stale values that dbk_process never looks at (same predicates).
// gtop: the row above belongs to the previous ROUND of row pairs (last wave -> first wave); that one
// hand-off goes through global memory (a bounded LDS ring there would close a dependency cycle).
__device__ __forceinline__ void dbk_prefetch(const FrameCtx &f, int mbx, int mby, int hl, bool act, bool gtop, DbkRegs &r)
{
	r.act = act;
	if (!act)
		return;
	r.hdr = *(const gu32 *)(f.mbs + mby * f.wm + mbx); // kind, flags, qp0, qp1
	gu8 *Yb = f.cur + (size_t)(mby * 16) * f.sY + mbx * 16;
	gu8 *Cb0 = plane_base(f, f.cur, 1) + (size_t)(mby * 8) * f.sC + mbx * 8;
	r.va0 = *(const gu32 *)(Yb + (size_t)(hl >> 2) * f.sY + (hl & 3) * 4);
	r.va1 = *(const gu32 *)(Yb + (size_t)(8 + (hl >> 2)) * f.sY + (hl & 3) * 4);
	r.vc = *(const gu32 *)(Cb0 + (hl >> 4) * (f.sC >> 1) + (size_t)((hl >> 1) & 7) * f.sC + (hl & 1) * 4);
	if (hl < 16)
		r.vp = ((const gu32 *)(f.dbk + (size_t)(mby * f.wm + mbx) * E264_DBK_BYTES))[hl];
	if (gtop) { // (gtop implies mby > 0) needed even without a top edge to filter: this row writes those rows back
		if (hl < 16) r.vt = *(const gu32 *)(Yb + (ptrdiff_t)(-4 + (hl >> 2)) * f.sY + (hl & 3) * 4);
		else if (hl < 24) { int i = hl - 16; r.vt = *(const gu32 *)(Cb0 + (i >> 2) * (f.sC >> 1) + (ptrdiff_t)(-2 + ((i >> 1) & 1)) * f.sC + (i & 1) * 4); }
	}
}

// ---- pair input: loads only (registers), commit to LDS, per-macroblock pick-up -------------------------------------
struct DbkPf { v4u a, b; };
__device__ __forceinline__ void dbk_prefetch_pair(const FrameCtx &f, int x0, int mby, int hl, DbkPf &p)
{ // nothing here uses a loaded value; pieces beyond the row's last macroblock are not fetched (and never looked at)
#ifdef E264_ABL_DBK_NOLOAD // timing ablation
	if (f.wm > 0) { p.a.x = x0; p.b.x = mby; return; }
#endif
	const int left = f.wm - x0; // macroblocks of the pair inside the row: min(left, 2)
	const gu8 *Yb = f.cur + (size_t)(mby * 16) * f.sY + x0 * 16;
	if ((hl & 1) < left) // luma piece hl: row hl >> 1, macroblock hl & 1
		p.a = *(const gv4u *)(Yb + (size_t)(hl >> 1) * f.sY + (hl & 1) * 16);
	const int a = mby * f.wm + x0;
	if (hl < 16) { // chroma piece hl: plane hl >> 3, row hl & 7, both macroblocks (8 bytes each)
		const gu8 *cp = plane_base(f, f.cur, 1 + (hl >> 3)) + (size_t)(mby * 8 + (hl & 7)) * f.sC + x0 * 8;
		if (left > 1) p.b = *(const gv4u *)cp;
		else { const v2u h = *(const gv2u *)cp; p.b.x = h.x; p.b.y = h.y; }
	} else if (hl < 24) { // parameter records: macroblock (hl - 16) >> 2, piece hl & 3
		if (((hl - 16) >> 2) < left) p.b = *(const gv4u *)(f.dbk + (size_t)(a + ((hl - 16) >> 2)) * E264_DBK_BYTES + (hl & 3) * 16);
	} else if (hl < 26) { // header records: macroblock hl - 24
		if (hl - 24 < left) p.b = *(const gv4u *)((const gu8 *)(f.payload - f.h->payload_off + f.h->mbs_off) + (size_t)(a + hl - 24) * sizeof(E264Mb));
	}
}
__device__ __forceinline__ void dbk_commit_pair(DbkIn &I, int hl, const DbkPf &p)
{
	*(v4u *)&I.y[hl >> 1][(hl & 1) * 4] = p.a;
	if (hl < 16) *(v4u *)&I.c[hl >> 3][hl & 7][0] = p.b;
	else if (hl < 24) *(v4u *)&I.prm[(hl - 16) >> 2][(hl & 3) * 4] = p.b;
	else if (hl < 26) *(v4u *)&I.hdr[hl - 24][0] = p.b;
}
__device__ __forceinline__ void dbk_fetch_mb(const DbkIn &I, int k, int hl, DbkRegs &r)
{ // macroblock k (0 / 1) of the committed pair -> the per-lane registers dbk_process consumes
	r.va0 = I.y[hl >> 2][k * 4 + (hl & 3)];
	r.va1 = I.y[8 + (hl >> 2)][k * 4 + (hl & 3)];
	r.vc = I.c[hl >> 4][(hl >> 1) & 7][k * 2 + (hl & 1)];
	r.vp = I.prm[k][hl & 15];
	r.hdr = I.hdr[k][0];
}
// top rows of macroblock (mbx, mby) from memory (first wave of a round: the row above belongs to the previous round)
__device__ __forceinline__ void dbk_prefetch_top(const FrameCtx &f, int mbx, int mby, int hl, uint32_t &vt)
{
	gu8 *Yb = f.cur + (size_t)(mby * 16) * f.sY + mbx * 16;
	gu8 *Cb0 = plane_base(f, f.cur, 1) + (size_t)(mby * 8) * f.sC + mbx * 8;
	if (hl < 16) vt = *(const gu32 *)(Yb + (ptrdiff_t)(-4 + (hl >> 2)) * f.sY + (hl & 3) * 4);
	else if (hl < 24) { int i = hl - 16; vt = *(const gu32 *)(Cb0 + (i >> 2) * (f.sC >> 1) + (ptrdiff_t)(-2 + ((i >> 1) & 1)) * f.sC + (i & 1) * 4); }
}

// top rows from the LDS ring of the row above: hl 0..15 luma rows -4..-1 (4 dwords each),
// hl 16..23 chroma rows -2..-1 (2 planes x 2 rows x 2 dwords)
__device__ __forceinline__ void dbk_load_top(const DbkRing &up, int mbx, int hl, DbkRegs &r)
{
	if (!r.act)
		return;
	const int slot = mbx & (DBK_RING - 1);
	if (hl < 16) r.vt = up.y[hl >> 2][slot * 4 + (hl & 3)];
	else if (hl < 24) { int i = hl - 16; r.vt = up.c[i >> 2][(i >> 1) & 1][slot * 2 + (i & 1)]; }
}

// Group g (macroblocks 4g .. 4g+nmb-1) of row mby: staged samples -> frame.
__device__ __forceinline__ void dbk_flush(const DbkStage &S, const FrameCtx &f, int g, int nmb, int mby, int hl, bool has_top, int nrow, int ncrow)
{
	// Runs every 4th step only: its per-lane index arithmetic is kept out of the set of loop invariants (opaque lane
	// index), where it competed for registers with the per-step code and pushed other invariants into scratch.
	asm volatile("" : "+v"(hl));
#ifdef E264_ABL_DBK_NOSTORE // timing ablation
	if (f.wm > 0) return;
#endif
	// all LDS reads first (unconditional, clamped indices), then the predicated stores: one LDS round trip instead of five
	gu8 *Yb = f.cur + (size_t)(mby * 16) * f.sY + g * 64;
	v4u yv[3], cv[2];
#pragma unroll
	for (int it = 0; it < 3; it++) { // 20 rows x 4 pieces of 16 bytes
		const int idx = min(it * 32 + hl, 79);
		yv[it] = *(const v4u *)&S.y[idx >> 2][(idx & 3) * 4];
	}
#pragma unroll
	for (int it = 0; it < 2; it++) { // 2 planes x 10 rows x 2 pieces of 16 bytes (= 2 macroblocks each)
		const int idx = min(it * 32 + hl, 39), pc = idx >= 20, rem = idx - pc * 20;
		cv[it] = *(const v4u *)&S.c[pc][rem >> 1][(rem & 1) * 4];
	}
#pragma unroll
	for (int it = 0; it < 3; it++) {
		const int idx = it * 32 + hl, row = (idx >> 2) - 4, c = idx & 3;
		if (idx < 80 && c < nmb && (row < 0 ? has_top : row < nrow))
			*(gv4u *)(Yb + (ptrdiff_t)row * f.sY + c * 16) = yv[it];
	}
#pragma unroll
	for (int it = 0; it < 2; it++) {
		const int idx = it * 32 + hl, pc = idx >= 20, rem = idx - pc * 20, row = (rem >> 1) - 2, c = rem & 1;
		if (idx < 40 && (row < 0 ? has_top : row < ncrow)) {
			gu8 *dst = plane_base(f, f.cur, 1 + pc) + (ptrdiff_t)(mby * 8 + row) * f.sC + g * 32 + c * 16;
			if (nmb >= 2 * c + 2) *(gv4u *)dst = cv[it];
			else if (nmb == 2 * c + 1) { v2u h = {cv[it].x, cv[it].y}; *(gv2u *)dst = h; }
		}
	}
}

// Filter the macroblock whose samples are in r, publish its bottom rows in `ring`, store it.
// S: staging of final samples; self_bottom: this row also writes its rows 12..15 (nobody below takes them from the ring)
__device__ __forceinline__ void dbk_process(DbkTile &L, DbkRing &ring, DbkStage &S, const uint8_t *tc0tab, const FrameCtx &f, int mbx, int mby, int hl,
	const DbkRegs &r, bool carry, bool last, bool self_bottom PH_PARAMS)
{
	const bool has_top = mby > 0;
	const uint32_t mkind = r.hdr & 255, mflags = r.hdr >> 8 & 255;
	const bool r_on = r.act && (mflags & E264_MBF_DEBLOCK) && mkind != E264_MB_ABSENT;
	const bool r_hasL = r_on && (mflags & E264_MBF_EDGE_LEFT), r_hasT = r_on && (mflags & E264_MBF_EDGE_TOP), r_t8 = mflags & E264_MBF_T8x8;
	const int pl = hl < 16 ? 0 : hl < 24 ? 1 : 2; // line roles: 0..15 luma, 16..23 Cb, 24..31 Cr
	const int li = hl < 16 ? hl : (hl & 7);
	const bool chroma = pl != 0;
	const int cpl = chroma ? pl - 1 : 0;
	// ---- carry the previous macroblock's right 4 columns, then drop the new samples in the tile
	uint32_t cv = 0;
	if (r.act && carry)
		cv = hl < 16 ? *(const uint32_t *)&L.DYT(li, 12) : *(const uint32_t *)&L.DCT(cpl, li, 4);
	wave_sync();
	if (r.act) {
		if (carry) {
			if (hl < 16) *(uint32_t *)&L.DYT(li, -4) = cv;
			else *(uint32_t *)&L.DCT(cpl, li, -4) = cv;
		}
		*(uint32_t *)&L.DYT(hl >> 2, (hl & 3) * 4) = r.va0;
		*(uint32_t *)&L.DYT(8 + (hl >> 2), (hl & 3) * 4) = r.va1;
		*(uint32_t *)&L.DCT(hl >> 4, (hl >> 1) & 7, (hl & 1) * 4) = r.vc;
		if (hl < 16) ((uint32_t *)L.prm)[hl] = r.vp;
		if (has_top && hl < 24) {
			if (hl < 16) *(uint32_t *)&L.DYT(-4 + (hl >> 2), (hl & 3) * 4) = r.vt;
			else { int i = hl - 16; *(uint32_t *)&L.DCT(i >> 2, -2 + ((i >> 1) & 1), (i & 1) * 4) = r.vt; }
		}
	}
	wave_sync();
	PH(4);
	int v[20];
	const int seg = chroma ? li >> 1 : li >> 2;
	if (r_on) {
		// ---- per-lane parameters of the VERTICAL edges crossing this line (those of the horizontal edges are
		// fetched after the vertical pass: fewer live registers) -------------------------------------------
		int bV[4], tV[4];
#pragma unroll
		for (int e = 0; e < 4; e++) bV[e] = L.prm[e * 4 + seg];
		if (!r_hasL) bV[0] = 0;
		if (chroma || r_t8) bV[1] = bV[3] = 0;
		const int a0 = L.prm[32 + pl * 3], a1 = L.prm[32 + pl * 3 + 1];
		const int b0 = L.prm[41 + pl * 3], b1 = L.prm[41 + pl * 3 + 1];
		const int i0 = L.prm[50 + pl * 3], i1 = L.prm[50 + pl * 3 + 1];
#pragma unroll
		for (int e = 0; e < 4; e++) // tc0tab has a zero row for bS 0 and 4 (index bS & 3; see kernel prologue)
			tV[e] = tc0tab[(bV[e] & 3) * 52 + (e ? i0 : i1)] & (bV[e] < 4 ? 255 : 0);
		// ---- vertical edges: this lane owns ROW li ----------------------------------------
		if (!chroma) {
#pragma unroll
			for (int d = 0; d < 5; d++) {
				uint32_t w = *(const uint32_t *)&L.DYT(li, d * 4 - 4);
				v[d * 4] = w & 255; v[d * 4 + 1] = w >> 8 & 255; v[d * 4 + 2] = w >> 16 & 255; v[d * 4 + 3] = w >> 24;
			}
		} else {
			uint32_t w0 = *(const uint32_t *)&L.DCT(cpl, li, -4), w1 = *(const uint32_t *)&L.DCT(cpl, li, 0), w2 = *(const uint32_t *)&L.DCT(cpl, li, 4);
			v[0] = w0 & 255; v[1] = w0 >> 8 & 255; v[2] = w0 >> 16 & 255; v[3] = w0 >> 24;
			v[4] = w1 & 255; v[5] = w1 >> 8 & 255; v[10] = w1 >> 16 & 255; v[11] = w1 >> 24;
			v[12] = w2 & 255; v[13] = w2 >> 8 & 255; v[14] = w2 >> 16 & 255; v[15] = w2 >> 24;
			v[6] = v[7] = v[8] = v[9] = v[16] = v[17] = v[18] = v[19] = 0;
		}
#ifndef E264_ABL_DBK_NOFILTER
		filter_line(v, bV, a1, a0, b1, b0, tV, chroma);
#endif
		if (!chroma) {
#pragma unroll
			for (int d = 0; d < 5; d++)
				*(uint32_t *)&L.DYT(li, d * 4 - 4) = (uint32_t)v[d * 4] | (uint32_t)v[d * 4 + 1] << 8 | (uint32_t)v[d * 4 + 2] << 16 | (uint32_t)v[d * 4 + 3] << 24;
		} else {
			*(uint32_t *)&L.DCT(cpl, li, -4) = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
			*(uint32_t *)&L.DCT(cpl, li, 0) = (uint32_t)v[4] | (uint32_t)v[5] << 8 | (uint32_t)v[10] << 16 | (uint32_t)v[11] << 24;
			*(uint32_t *)&L.DCT(cpl, li, 4) = (uint32_t)v[12] | (uint32_t)v[13] << 8 | (uint32_t)v[14] << 16 | (uint32_t)v[15] << 24;
		}
	}
	wave_sync();
	PH(5);
	if (r_on) {
		// ---- horizontal edges: this lane owns COLUMN li -----------------------------------
		int bH[4], tH[4];
#pragma unroll
		for (int e = 0; e < 4; e++) bH[e] = L.prm[16 + e * 4 + seg];
		if (!r_hasT) bH[0] = 0;
		if (chroma || r_t8) bH[1] = bH[3] = 0;
		const int a0 = L.prm[32 + pl * 3], a2 = L.prm[32 + pl * 3 + 2];
		const int b0 = L.prm[41 + pl * 3], b2 = L.prm[41 + pl * 3 + 2];
		const int i0 = L.prm[50 + pl * 3], i2 = L.prm[50 + pl * 3 + 2];
#pragma unroll
		for (int e = 0; e < 4; e++)
			tH[e] = tc0tab[(bH[e] & 3) * 52 + (e ? i0 : i2)] & (bH[e] < 4 ? 255 : 0);
		if (!chroma) {
#pragma unroll
			for (int i = 0; i < 20; i++) v[i] = L.DYT(i - 4, li);
		} else {
#pragma unroll
			for (int i = 0; i < 6; i++) { v[i] = L.DCT(cpl, i - 4, li); v[10 + i] = L.DCT(cpl, i + 2, li); }
		}
#ifndef E264_ABL_DBK_NOFILTER
		filter_line(v, bH, a2, a0, b2, b0, tH, chroma);
#endif
		if (!chroma) {
#pragma unroll
			for (int i = 1; i < 19; i++) L.DYT(i - 4, li) = (uint8_t)v[i];
		} else {
			L.DCT(cpl, -1, li) = (uint8_t)v[3]; L.DCT(cpl, 0, li) = (uint8_t)v[4];
			L.DCT(cpl, 3, li) = (uint8_t)v[11]; L.DCT(cpl, 4, li) = (uint8_t)v[12];
		}
	}
	wave_sync();
	PH(6);
	if (!r.act)
		return;
	// ---- publish the bottom rows for the row below: this macroblock's columns 0..11 (chroma 0..3)
	// and, now that the left edge has been filtered, the previous macroblock's columns 12..15 (4..7)
	{ // branch-free: every lane makes ONE 4-byte copy tile -> ring (source, destination and predicate selected by its role).
	  // The three-way role branch with its inner cases ran as six serialised divergent paths, each with its own LDS round trip.
		const int slot = mbx & (DBK_RING - 1), pslot = (mbx - 1) & (DBK_RING - 1);
		const bool ra = hl < 16, rb = !ra && hl < 24;
		const int ia = hl & 3, ib = hl - 16, ic = hl - 24, jc = ic - 4;
		// source: byte offset inside the DbkTile
		const int srcA = (12 + (hl >> 2) + 4) * DY_STRIDE + 4 + (ia == 0 ? -4 : (ia - 1) * 4);
		const int srcB = (int)offsetof(DbkTile, dctile) + (ib >> 2) * 12 * DC_STRIDE + (6 + ((ib >> 1) & 1) + 4) * DC_STRIDE + 4 + ((ib & 1) ? 0 : -4);
		const int srcC = ic < 4 ? (12 + ic + 4) * DY_STRIDE + 4 + 12
		                        : (int)offsetof(DbkTile, dctile) + (jc >> 1) * 12 * DC_STRIDE + (6 + (jc & 1) + 4) * DC_STRIDE + 4 + 4;
		// destination: dword index inside the DbkRing (y[4][DBK_RING*4] then c[2][2][DBK_RING*2]) = base + slot * mul
		const int dstA = (hl >> 2) * DBK_RING * 4 + (ia == 0 ? 3 : ia - 1);
		const int dstB = 4 * DBK_RING * 4 + ((ib >> 2) * 2 + ((ib >> 1) & 1)) * DBK_RING * 2 + ((ib & 1) ? 0 : 1);
		const int dstC = ic < 4 ? ic * DBK_RING * 4 + 3 : 4 * DBK_RING * 4 + ((jc >> 1) * 2 + (jc & 1)) * DBK_RING * 2 + 1;
		const int src = ra ? srcA : rb ? srcB : srcC;
		const int dst0 = ra ? dstA : rb ? dstB : dstC;
		const int mul = ra ? 4 : rb ? 2 : (ic < 4 ? 4 : 2);
		const bool prev = ra ? ia == 0 : rb ? !(ib & 1) : false; // the previous macroblock's last columns (valid once carried)
		const bool pred = (ra || rb) ? (prev ? carry : true) : last;
		const uint32_t val = *(const uint32_t *)((const uint8_t *)&L + src);
		if (pred) ((uint32_t *)&ring)[dst0 + (prev ? pslot : slot) * mul] = val;
	}
	PH(7);
	// ---- stage what has become final; groups of 4 macroblocks leave as whole 64-byte row pieces ----------
	const int gxm = mbx & 3;
	const int nrow = self_bottom ? 16 : 12, ncrow = self_bottom ? 8 : 6;
	{ // the left neighbour's last 4 columns (tile columns -4..-1), rows 0..nrow-1: ONE predicated copy per lane
	  // (lanes 0..15 luma row hl, lanes 16..31 chroma plane (hl-16)>>3 row (hl-16)&7), no divergent paths
		const bool lu = hl < 16;
		const int cr = (hl - 16) & 7, cp = (hl - 16) >> 3;
		const int src = lu ? (hl + 4) * DY_STRIDE : (int)offsetof(DbkTile, dctile) + cp * 12 * DC_STRIDE + (cr + 4) * DC_STRIDE;
		const uint32_t val = *(const uint32_t *)((const uint8_t *)&L + src);
		uint32_t *dst = lu ? &S.y[hl + 4][gxm ? gxm * 4 - 1 : 15] : &S.c[cp][cr + 2][gxm ? gxm * 2 - 1 : 7];
		if (carry && (lu ? hl < nrow : cr < ncrow)) *dst = val;
		if (carry && gxm == 0) { // ... which completes the previous group
			wave_sync();
			dbk_flush(S, f, (mbx >> 2) - 1, 4, mby, hl, has_top, nrow, ncrow);
		}
	}
	PH(10);
	wave_sync();
	{ // this macroblock: luma rows -4..15 x 4 dwords, chroma 2 planes x rows -2..7 x 2 dwords.  All LDS reads first
	  // (unconditional, clamped), then the predicated writes.
		uint32_t yv[3], cv[2];
#pragma unroll
		for (int it = 0; it < 3; it++) {
			const int idx = min(it * 32 + hl, 79);
			yv[it] = *(const uint32_t *)&L.DYT((idx >> 2) - 4, (idx & 3) * 4);
		}
#pragma unroll
		for (int it = 0; it < 2; it++) {
			const int idx = min(it * 32 + hl, 39), pc = idx >= 20, rem = idx - pc * 20;
			cv[it] = *(const uint32_t *)&L.DCT(pc, (rem >> 1) - 2, (rem & 1) * 4);
		}
#pragma unroll
		for (int it = 0; it < 3; it++) {
			const int idx = it * 32 + hl, row = (idx >> 2) - 4, dw = idx & 3;
			if (idx < 80 && (row < 0 ? has_top : (row < nrow && (dw < 3 || last))))
				S.y[row + 4][gxm * 4 + dw] = yv[it];
		}
#pragma unroll
		for (int it = 0; it < 2; it++) {
			const int idx = it * 32 + hl, pc = idx >= 20, rem = idx - pc * 20, row = (rem >> 1) - 2, dw = rem & 1;
			if (idx < 40 && (row < 0 ? has_top : (row < ncrow && (dw < 1 || last))))
				S.c[pc][row + 2][gxm * 2 + dw] = cv[it];
		}
	}
	PH(11);
	if (last) {
		wave_sync();
		dbk_flush(S, f, mbx >> 2, gxm + 1, mby, hl, has_top, nrow, ncrow);
	}
}

// ---------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int lds_load_relaxed(const int *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


#define E264_MAX_ROWS 1056
} // namespace

// XCD-aware workgroup order.  The dispatcher places linear workgroup b on XCD b % 8, each with a private
// 4 MiB L2; in launch order the strips that share reference rows (vertical neighbours of one frame,
// 15 strips = 4 workgroups apart) land on different XCDs and every one of them fetches the shared
// halo lines from HBM again.  This bijective remap gives each XCD a contiguous range of (frame, strip)
// pairs, so a frame is walked by ONE XCD and the halo rows are L2 hits.  Pure speed choice.
static __device__ __forceinline__ void xcd_tile(int &bx, int &by)
{
	const unsigned gx = gridDim.x, nwg = gx * gridDim.y;
	const unsigned lin = blockIdx.y * gx + blockIdx.x;
	const unsigned q = nwg >> 3, r = nwg & 7, xcd = lin & 7;
	const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
	by = (int)(v / gx);
	bx = (int)(v - (unsigned)by * gx);
}

// Inter prediction + residual of every inter / PCM macroblock: one workgroup per tile of 16 x 8 macroblocks, one thread
// per 8x8 block; the phases are in e264_pred.h (and run on the host by tests/emu).
#ifndef E264_PRED_WAVES_PER_EU
#define E264_PRED_WAVES_PER_EU 4 // 128 VGPRs: two 512-thread workgroups per CU, so that one tile's barriers and first loads hide behind the other's arithmetic
#endif
__attribute__((amdgpu_waves_per_eu(E264_PRED_WAVES_PER_EU, E264_PRED_WAVES_PER_EU))) __global__ __launch_bounds__(PT_NT) void e264_pred_kernel(const E264Job *jobs, int mode)
{
	__shared__ PredLds L;
	const int tid = (int)threadIdx.x;
	FrameCtx f;
	int bx, by;
	xcd_tile(bx, by);
	if (!open_frame(f, jobs[by]))
		return;
	const int ntx = (f.wm + PT_W - 1) / PT_W, nty = (f.hm + PT_H - 1) / PT_H;
	if (bx >= ntx * nty)
		return;
	const PredTile t = {(bx % ntx) * PT_W, (bx / ntx) * PT_H};
	PH_DECL;
	pred_phase_setup(L, f, t, tid);
	PH(0);
	__syncthreads();
	{ // nothing more for this kernel in the tile (every tile of an I frame)? leave at once
		const int kind = tid < PT_MBS ? (int)(L.hdr[tid][0] & 255) : 0;
		if (!__syncthreads_or(kind == E264_MB_INTER || kind == E264_MB_PCM))
			return;
	}
	PH(1);
	pred_phase_classify(L, f, t, 0, tid);
	PH(2);
	__syncthreads();
	PH(3);
	pred_phase_items(L, f, t, 0, tid);
	PH(4);
	__syncthreads();
	PH(5);
	if (L.any_l1) { // uniform: written before the barrier above
		pred_phase_reset(L, tid);
		__syncthreads();
		pred_phase_classify(L, f, t, 1, tid);
		__syncthreads();
		pred_phase_items(L, f, t, 1, tid);
		__syncthreads();
	}
	PH(6);
	pred_phase_reslist(L, tid);
	PH(7);
	__syncthreads();
	PH(8);
	pred_phase_residual(L, f, tid);
	PH(9);
	__syncthreads();
	PH(10);
	pred_phase_flush(L, f, t, tid);
	PH(11);
	PH_FLUSH(tid & 63);
}

// Deblocking parameters of every macroblock, 64 consecutive macroblocks per workgroup: records in through LDS with
// contiguous 16-byte loads, parameters out as contiguous 16-byte stores (e264_dbkp.h; the phases run on the host in tests/emu).
__global__ __launch_bounds__(DP_NT) void e264_dbkparam2_kernel(const E264Job *jobs)
{
	__shared__ DbkpLds L;
	const int tid = (int)threadIdx.x;
	FrameCtx f;
	int bx, by;
	xcd_tile(bx, by);
	if (!open_frame(f, jobs[by]) || !f.dbk)
		return;
	const int a0 = bx * DP_MBS;
	if (a0 >= f.wm * f.hm)
		return;
	dbkp_phase_load(L, f, a0, tid);
	__syncthreads();
	dbkp_phase_slices(L, f, tid);
	__syncthreads();
	dbkp_phase_compute(L, f, a0, tid);
	__syncthreads();
	dbkp_phase_store(L, f, a0, tid);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void e264_intra_kernel(const E264Job *jobs)
{
	__shared__ WaveLds lds[NW];
	__shared__ __attribute__((aligned(16))) uint32_t hdrs[NW][64 * 8]; // E264Mb records of the 64 macroblocks being scanned, per wave
	__shared__ int progress[E264_MAX_ROWS]; // macroblocks finished per row
	const int lane = lane_id();
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	FrameCtx f;
	if (!open_frame(f, jobs[blockIdx.x]))
		return;
	if (f.h->n_coded_mbs == f.h->n_inter_mbs)
		return; // nothing intra in this frame (PCM is handled by the parallel kernel but counted as coded: rare)
	for (int i = threadIdx.x; i < f.hm; i += NW * 64)
		progress[i] = 0;
	__syncthreads();
	WaveLds &L = lds[wave];
	if (lane == 0) L.ws_slice = -1;
	wave_sync();
	const gu8 *mbs_g = f.payload - f.h->payload_off + f.h->mbs_off; // the E264Mb array through a per-lane (global) pointer
	PH_DECL;
#pragma unroll 1
	for (int y = wave; y < f.hm; y += NW) {
		// the row is scanned 64 macroblocks at a time (one vector load + ballot) instead of one scalar load per
		// macroblock: in P/B frames, where few macroblocks are intra, the scan WAS the kernel's run time
#pragma unroll 1
		for (int x0 = 0; x0 < f.wm; x0 += 64) {
			const int xl = x0 + lane;
			// whole records of the chunk -> LDS (2 x 16 bytes per lane); `kind` for the ballot comes out of the first dword
			v4u h0v = {0, 0, 0, 0}, h1v = {0, 0, 0, 0};
			if (xl < f.wm) {
				const gv4u *rp = (const gv4u *)(mbs_g + (size_t)(y * f.wm + xl) * sizeof(E264Mb));
				h0v = rp[0]; h1v = rp[1];
			}
			wave_sync(); // the previous chunk's records are no longer read
			*(v4u *)&hdrs[wave][lane * 8] = h0v;
			*(v4u *)&hdrs[wave][lane * 8 + 4] = h1v;
			wave_sync();
			const int kind = xl < f.wm ? (int)(h0v.x & 255) : E264_MB_ABSENT;
			unsigned long long todo = __ballot(kind == E264_MB_I4x4 || kind == E264_MB_I8x8 || kind == E264_MB_I16x16);
			const int xe = min(x0 + 64, f.wm);
			if (todo == 0 || (int)__builtin_ctzll(todo) > 0) { // macroblocks before the first intra one need nothing from this kernel
				const int upto = todo ? x0 + (int)__builtin_ctzll(todo) : xe;
				if (lane == 0)
					__hip_atomic_store(&progress[y], upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
#pragma unroll 1
			while (todo) {
				const int x = x0 + (int)__builtin_ctzll(todo);
				todo &= todo - 1;
				PH(0);
				const MbInfo mi = mb_from_lds(&hdrs[wave][(x - x0) * 8]);
				CoefPf pf;
				coef_issue(f, mi, lane, pf); // the payload does not depend on the neighbours: in flight during the wait below
				if (y > 0) {
					int want = min(x + 2, f.wm);
					while (lds_load_relaxed(&progress[y - 1]) < want)
						__builtin_amdgcn_s_sleep(1);
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				}
				PH(1);
				recon_mb<2>(L, f, mi, x, y, lane, pf PH_ARGS);
				PH(8);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				PH(9);
				// finished: everything up to the next intra macroblock of the chunk (or the chunk's end)
				const int upto = todo ? x0 + (int)__builtin_ctzll(todo) : xe;
				if (lane == 0)
					__hip_atomic_store(&progress[y], upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
	}
#ifdef E264_PHASE_INTRA
	PH_FLUSH_DBK(lane);
#endif
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void e264_deblock_kernel(const E264Job *jobs)
{
	__shared__ DbkLds lds[NW];
	__shared__ int progress[E264_MAX_ROWS];
	__shared__ uint8_t tc0tab[4 * 52]; // row (bS & 3): row 0 is all zero (bS 0 and 4 have no tC0)
	const int lane = lane_id();
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int half = lane >> 5, hl = lane & 31;
	FrameCtx f;
	if (!open_frame(f, jobs[blockIdx.x]) || !f.dbk)
		return;
	for (int i = threadIdx.x; i < f.hm; i += NW * 64)
		progress[i] = 0;
	for (int i = threadIdx.x; i < 4 * 52; i += NW * 64)
		tc0tab[i] = i < 52 ? 0 : c_tc0[i / 52 - 1][i % 52];
	__syncthreads();
	DbkLds &L = lds[wave];
	const int upw = (wave + NW - 1) % NW; // the wave that owns the row pair above
#pragma unroll 1
	for (int pr = wave; 2 * pr < f.hm; pr += NW) { // row pair: rows 2*pr (lanes 0..31) and 2*pr+1 (lanes 32..63)
		const int yA = 2 * pr, yB = yA + 1;
		const int my_y = yA + half;
		const bool row_ok = my_y < f.hm;
		const int par = (pr / NW) & 1, uppar = ((pr - 1 + NW) / NW + 1) & 1; // ring buffers alternate per round
		DbkRing &myring = L.ring[par][half];
		const DbkRing &upring = half ? L.ring[par][0] : lds[upw].ring[(wave == 0) ? par ^ 1 : par][1];
		(void)uppar;
		// wave 0 takes its top rows from global memory (written by the last wave one round earlier)
		const bool gtop_wave = wave == 0 && yA > 0;
		const bool gtop = gtop_wave && half == 0;
		// the lower row of the last wave is read by wave 0 of the next round from memory, not from the ring
		const bool handoff = wave == NW - 1 && half == 1 && my_y + 1 < f.hm;
		const bool self_bottom = handoff || my_y == f.hm - 1;
		DbkRegs cur = {0, 0, 0, 0, 0, 0, false};
		DbkPf pf;
		uint32_t vt_next = 0;
		PH_DECL;
		if (gtop_wave) {
			while (lds_load_relaxed(&progress[yA - 1]) < min(2, f.wm))
				__builtin_amdgcn_s_sleep(1);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		}
		if (row_ok) dbk_prefetch_pair(f, 0, my_y, hl, pf); // the rows' own samples depend on nothing in this kernel
		if (gtop && row_ok) dbk_prefetch_top(f, 0, my_y, hl, vt_next);
#pragma unroll 1
		for (int t = 0; t < f.wm + DBK_LAG; t++) {
			const int xB = t - DBK_LAG;
			const int my_x = half ? xB : t;
			PH(0);
			if (yA > 0 && t < f.wm) { // the row above must be 2 macroblocks ahead
				int want = min(t + (gtop_wave ? 3 : 2), f.wm); // wave 0 also prefetches the next macroblock's top rows
				while (lds_load_relaxed(&progress[yA - 1]) < want)
					__builtin_amdgcn_s_sleep(1);
			}
			if (wave != NW - 1 && yB + 1 < f.hm && xB >= DBK_RING - 2) { // ring back-pressure: the slot must have been consumed
				while (lds_load_relaxed(&progress[yB + 1]) < xB - (DBK_RING - 2))
					__builtin_amdgcn_s_sleep(1);
			}
			if (gtop_wave) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			PH(1);
			cur.act = row_ok && my_x >= 0 && my_x < f.wm;
			if (cur.act && (my_x & 1) == 0) { // a new pair: the prefetched registers -> LDS, the next pair's loads go out
				__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the pair was requested 2 macroblocks ago
				dbk_commit_pair(L.in[half], hl, pf);
				if (my_x + 2 < f.wm) dbk_prefetch_pair(f, my_x + 2, my_y, hl, pf);
			}
			wave_sync();
			if (cur.act) dbk_fetch_mb(L.in[half], my_x & 1, hl, cur);
			if (gtop) {
				cur.vt = vt_next;
				if (row_ok && my_x + 1 < f.wm) dbk_prefetch_top(f, my_x + 1, my_y, hl, vt_next);
			} else if (my_y > 0)
				dbk_load_top(upring, my_x, hl, cur);
			PH(3);
			dbk_process(L.tile[half], myring, L.stage[half], tc0tab, f, my_x, my_y, hl, cur, my_x > 0, my_x == f.wm - 1, self_bottom PH_ARGS);
			PH(8);
			// LDS operations of a wave execute in order: the ring is written before the counter.  The last
			// wave hands its lower row to the next round through global memory: its stores must be visible.
			if (wave == NW - 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			// a row handed over through memory counts the macroblocks whose samples have LEFT the staging buffer
			const int done = handoff ? (my_x == f.wm - 1 ? f.wm : (my_x & ~3)) : my_x + 1;
			if (hl == 0 && cur.act)
				__hip_atomic_store(&progress[my_y], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			PH(9);
		}
#ifndef E264_PHASE_INTRA
		PH_FLUSH_DBK(lane);
#endif
	}
}

extern "C" hipError_t e264_launch_frames(const E264Job *jobs, int n_jobs, int max_mbs, int max_tiles, int mode, int waves, hipStream_t stream, hipEvent_t *marks,
	const E264Fork *fork)
{
	if (n_jobs <= 0)
		return hipSuccess;
	// marks (optional): 5 events recorded before / between / after the four launches
	if (marks) hipEventRecord(marks[0], stream);
	const bool dbkp = (mode & 2) && !(mode & 2048);
	// fork (optional): the parameter kernel reads nothing but the packet and is needed only by the deblocking kernel, so it
	// runs on a second queue NEXT TO the macroblock-parallel kernel -- 28 VGPRs per wave, its waves fit beside the two
	// 215-VGPR waves per SIMD and use issue slots those leave idle.  Ordered after everything enqueued before (the packet
	// copies, the previous batch's deblocking that still reads the parameter buffer) by `forked`, before deblocking by `joined`.
	const bool side = dbkp && fork && fork->aux;
	// side queue, second form (mode bit 16): the fork sits AFTER the prediction kernel, so that the parameter kernel (bound by
	// packet reads) runs beside the intra wavefront kernel (bound by dependency latency) instead of beside the prediction kernel
	const bool side_late = side && (mode & 65536);
	auto launch_side = [&]() {
		hipEventRecord(fork->forked, stream);
		hipStreamWaitEvent(fork->aux, fork->forked, 0);
		if (marks) hipEventRecord(fork->amarks[0], fork->aux);
		hipLaunchKernelGGL(e264_dbkparam2_kernel, dim3((max_mbs + DP_MBS - 1) / DP_MBS, n_jobs), dim3(DP_NT), 0, fork->aux, jobs);
		if (marks) hipEventRecord(fork->amarks[1], fork->aux);
		hipEventRecord(fork->joined, fork->aux);
	};
	if (side) {
		if (!side_late) launch_side();
	} else if (dbkp) // (mode bit 17: the prediction kernel computes the parameters of its tiles itself)
		hipLaunchKernelGGL(e264_dbkparam2_kernel, dim3((max_mbs + DP_MBS - 1) / DP_MBS, n_jobs), dim3(DP_NT), 0, stream, jobs);
	if (marks) hipEventRecord(marks[1], stream);
	if (mode & 1)
		hipLaunchKernelGGL(e264_pred_kernel, dim3(max_tiles, n_jobs), dim3(PT_NT), 0, stream, jobs, mode);
	if (marks) hipEventRecord(marks[2], stream);
	if (side_late) launch_side();
	const int intra_waves = waves >> 8 ? waves >> 8 : waves & 255;
	waves &= 255;
	if (mode & 1) {
		switch (intra_waves) {
		case 4: hipLaunchKernelGGL(e264_intra_kernel<4>, dim3(n_jobs), dim3(256), 0, stream, jobs); break;
		case 16: hipLaunchKernelGGL(e264_intra_kernel<16>, dim3(n_jobs), dim3(1024), 0, stream, jobs); break;
		default: hipLaunchKernelGGL(e264_intra_kernel<8>, dim3(n_jobs), dim3(512), 0, stream, jobs); break;
		}
	}
	if (marks) hipEventRecord(marks[3], stream);
	if (side) hipStreamWaitEvent(stream, fork->joined, 0);
	if (mode & 2) {
		switch (waves) {
		case 9: hipLaunchKernelGGL(e264_deblock_kernel<9>, dim3(n_jobs), dim3(576), 0, stream, jobs); break;
		case 10: hipLaunchKernelGGL(e264_deblock_kernel<10>, dim3(n_jobs), dim3(640), 0, stream, jobs); break;
		case 12: case 16: hipLaunchKernelGGL(e264_deblock_kernel<12>, dim3(n_jobs), dim3(768), 0, stream, jobs); break; // 16 waves no longer fit the LDS
		case 4: hipLaunchKernelGGL(e264_deblock_kernel<4>, dim3(n_jobs), dim3(256), 0, stream, jobs); break;
		default: hipLaunchKernelGGL(e264_deblock_kernel<8>, dim3(n_jobs), dim3(512), 0, stream, jobs); break;
		}
	}
	if (marks) hipEventRecord(marks[4], stream);
	return hipGetLastError();
}

#ifdef E264_PHASE_TIMING
extern "C" __attribute__((visibility("default"))) int e264_debug_phase_cycles(unsigned long long *out32, int reset)
{
	hipDeviceSynchronize();
	if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_phase), sizeof(g_phase)) != hipSuccess) return -1;
	if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return -1; }
	return 0;
}
#endif
