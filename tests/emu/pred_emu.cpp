// tests/emu/pred_emu.cpp -- TEST INFRASTRUCTURE.  Compiles the product's kernel source (edge264_amd/csrc/e264_pred.h) for
// the host and runs e264_pred_kernel's phases thread by thread, tile by tile: the same arithmetic, the same LDS layout,
// the same work lists, with the four VALU byte instructions the source names restated below.  tests/test_pred_emu.py
// compares the result with the CPU oracle, so that a logic error is found here and not on the GPU box.
#include <stdlib.h>
#include "emu_shims.h"
#include "../../edge264_amd/csrc/e264_pred.h"
#include "../../edge264_amd/csrc/e264_dbkp.h"
#include "../../edge264_amd/csrc/e264_expand.h"

// e264_expand_kernel over one wire packet (include/edge264_compact.h) with a grid of nt threads: area = e264_expand_area_bytes(wire)
extern "C" __attribute__((visibility("default"))) int e264emu_expand(const uint8_t *wire, uint8_t *area, int nt)
{
	E264Job job = {wire, nullptr, nullptr, area};
	for (int t = 0; t < nt; t++) expand_thread(job, (uint32_t)t, (uint32_t)nt);
	return 0;
}
static uint8_t *g_expand; // the expansion buffer the frame entry points below hand to the kernels (NULL: version-4 packets)
extern "C" __attribute__((visibility("default"))) void e264emu_set_expand(uint8_t *area) { g_expand = area; }

// dbk: NULL, or the stream's scratch (E264_SCRATCH_BYTES(macroblocks)): the kernel then also writes the intra bitmap of its tiles
extern "C" __attribute__((visibility("default"))) int e264emu_pred_frame2(const uint8_t *pkt, uint8_t *const *dpb, uint8_t *dbk)
{
	E264Job job = {pkt, dpb, dbk, g_expand};
	FrameCtx f;
	if (!open_frame(f, job))
		return -1;
	static PredLds L;
	const int ntx = (f.wm + PT_W - 1) / PT_W, nty = (f.hm + PT_H - 1) / PT_H;
	for (int ti = 0; ti < ntx * nty; ti++) {
		PredTile t = {(ti % ntx) * PT_W, (ti / ntx) * PT_H};
		memset(&L, 0xA5, sizeof(L)); // LDS is not zeroed on the device either
		for (int tid = 0; tid < PT_NT; tid++) pred_phase_setup(L, f, t, tid);
		for (int tid = 0; tid < PT_NT; tid++) pred_phase_bitmap(L, f, t, tid);
		for (int list = 0; list < 2; list++) {
			if (list == 1 && !L.any_l1) break;
			if (list == 1) for (int tid = 0; tid < PT_NT; tid++) pred_phase_reset(L, tid);
			for (int tid = 0; tid < PT_NT; tid++) pred_phase_classify(L, f, t, list, tid);
			for (int tid = 0; tid < PT_NT; tid++) pred_phase_items(L, f, t, list, tid);
		}
		for (int tid = 0; tid < PT_NT; tid++) pred_phase_reslist(L, tid);
		for (int tid = 0; tid < PT_NT; tid++) pred_phase_residual(L, f, tid);
		for (int tid = 0; tid < PT_NT; tid++) pred_phase_flush(L, f, t, tid);
	}
	return 0;
}

extern "C" __attribute__((visibility("default"))) int e264emu_pred_frame(const uint8_t *pkt, uint8_t *const *dpb)
{
	return e264emu_pred_frame2(pkt, dpb, nullptr);
}

// e264_dbkparam2_kernel: out = E264_DBK_BYTES (256) per macroblock, the pieces of the deblocking lanes' layout; raw (may be NULL) = the 64-byte
// raw records (bS, alpha, beta, indexA) the pieces are made of, as they stand in LDS between the kernel's phases
template <bool HAS_L1> static int emu_dbkparam(const uint8_t *pkt, uint8_t *out, uint8_t *raw);
extern "C" __attribute__((visibility("default"))) int e264emu_dbkparam_frame2(const uint8_t *pkt, uint8_t *out, uint8_t *raw) { return emu_dbkparam<true>(pkt, out, raw); }
// the kernel's small form (no room for list 1 in LDS): for pictures that do not predict from list 1 only -- the launcher's choice on the device
extern "C" __attribute__((visibility("default"))) int e264emu_dbkparam_frame2_nol1(const uint8_t *pkt, uint8_t *out, uint8_t *raw) { return emu_dbkparam<false>(pkt, out, raw); }
template <bool HAS_L1> static int emu_dbkparam(const uint8_t *pkt, uint8_t *out, uint8_t *raw)
{
	uint8_t dummy = 0;
	uint8_t *dpb[E264_MAX_SLOTS];
	for (int i = 0; i < E264_MAX_SLOTS; i++) dpb[i] = &dummy;
	E264Job job = {pkt, dpb, out, g_expand};
	FrameCtx f;
	if (!open_frame(f, job))
		return -1;
	static DbkpLdsT<HAS_L1> L;
	const int n = f.wm * f.hm;
	for (int a0 = 0; a0 < n; a0 += DP_MBS) {
		memset(&L, 0xA5, sizeof(L));
		for (int tid = 0; tid < DP_NT; tid++) dbkp_phase_load(L, f, a0, tid);
		for (int tid = 0; tid < DP_NT; tid++) dbkp_phase_slices(L, f, tid);
		for (int tid = 0; tid < DP_NT; tid++) dbkp_phase_compute(L, f, a0, tid);
		if (raw) for (int i = 0; i < DP_MBS && a0 + i < n; i++) memcpy(raw + (size_t)(a0 + i) * DP_RAW, L.out[i], DP_RAW);
		memset(L.mo, 0xA5, sizeof(L.mo)); // (the pieces reuse the motion area: nothing of it may be read any more)
		for (int tid = 0; tid < DP_NT; tid++) dbkp_phase_pieces(L, tid);
		for (int tid = 0; tid < DP_NT; tid++) dbkp_phase_store(L, f, a0, tid);
	}
	return 0;
}
// the pieces alone: what e264emu_deblock_frame2 consumes
extern "C" __attribute__((visibility("default"))) int e264emu_dbkparam_frame(const uint8_t *pkt, uint8_t *out)
{
	return e264emu_dbkparam_frame2(pkt, out, nullptr);
}
// the raw records alone (64 bytes per macroblock)
extern "C" __attribute__((visibility("default"))) int e264emu_dbkparam_raw(const uint8_t *pkt, uint8_t *raw)
{
	FrameCtx f;
	uint8_t dummy = 0;
	uint8_t *dpb[E264_MAX_SLOTS];
	for (int i = 0; i < E264_MAX_SLOTS; i++) dpb[i] = &dummy;
	E264Job job = {pkt, dpb, &dummy, g_expand};
	if (!open_frame(f, job))
		return -1;
	uint8_t *out = (uint8_t *)malloc((size_t)f.wm * f.hm * E264_DBK_BYTES);
	const int r = e264emu_dbkparam_frame2(pkt, out, raw);
	free(out);
	return r;
}
// a raw record -> the E264_DBK_BYTES of the lanes' layout (what the kernel's piece phase does), by the slot-by-slot definition
extern "C" __attribute__((visibility("default"))) void e264emu_dbk_pieces(const uint8_t *raw, uint8_t *out)
{
	uint8_t tc0tab[4 * 52];
	for (int i = 0; i < 4 * 52; i++) tc0tab[i] = i < 52 ? 0 : c_tc0[i / 52 - 1][i % 52];
	for (int c = 0; c < 2; c++)
		for (int dir = 0; dir < 2; dir++)
			for (int sgm = 0; sgm < 4; sgm++) {
				const v2u p = dbkp_piece(raw, tc0tab, c != 0, dir, sgm);
				memcpy(out + c * 64 + sgm * 16 + dir * 8, &p, 8);
			}
	const v4u w = dbkp_mbwide(raw);
	memcpy(out + 128, &w, 16);
}
extern "C" __attribute__((visibility("default"))) int e264emu_dbk_bytes(void) { return E264_DBK_BYTES; }

// e264_deblock_kernel / e264_deblock2_kernel: dbk = the parameter records (e264emu_dbkparam_frame's output); the picture in
// dpb[dst_slot] is filtered in place.  Groups of rows are run one after the other (a group only ever waits for the group above it
// of its own kind), the lanes of a wave phase by phase.  K: the kind of wave (DkGeom): 2 mixed, 0 luma only, 1 chroma only.
#include "../../edge264_amd/csrc/e264_dbk.h"
static long g_zero_steps, g_filter_steps; // steps that took the copy-only path / the filter path (tests check that both are exercised)
extern "C" __attribute__((visibility("default"))) void e264emu_deblock_step_counts(long *zero, long *filt, int reset)
{
	*zero = g_zero_steps; *filt = g_filter_steps;
	if (reset) g_zero_steps = g_filter_steps = 0;
}
template <int K>
static void emu_walk_group(const FrameCtx &f, int q)
{
	typedef DkGeom<K> G;
	static DkWaveT<K> W;
	DkRole R[64];
	for (int lane = 0; lane < 64; lane++) R[lane] = dk_role<K>(lane);
	memset(&W, 0xA5, sizeof(W));
	const int y0 = q * G::ROWS;
	const bool top = q > 0;
	static v4u N[64][2 * DK_GS], K2a[64], K2b[64], K3a[64], K3b[64], tt[64], ra[64], rb[64];
	static DkRaw np[2][64];
	memset(N, 0x5A, sizeof(N)); memset(K2a, 0x5A, sizeof(K2a)); memset(K2b, 0x5A, sizeof(K2b)); memset(K3a, 0x5A, sizeof(K3a)); memset(K3b, 0x5A, sizeof(K3b));
	memset(np, 0x5A, sizeof(np)); memset(tt, 0x5A, sizeof(tt));
	for (int t4 = DK_FIRST_STEP; t4 <= dk_last_step<K>(f.wm); t4 += DK_GS) // (the kernel's loop: whole groups of four (two) steps)
	for (int t = t4; t < t4 + DK_GS; t++) {
		const int par = t & 1, k = (t + 2) & (DK_GS - 1); // the parameter register set of this step; which macroblock of its group it filters
		DkPlan p[64];
		for (int lane = 0; lane < 64; lane++) {
			const int y = y0 + R[lane].g;
			p[lane] = dk_plan(t, R[lane], !R[lane].idle && y < f.hm, top, f.wm);
			if (p[lane].top_commit >= 0) dk_top_commit<K>(W, f, lane, p[lane].top_commit, y0, tt[lane]);
			if (DK_GS == 4) {
				if (k < 2) dk_pick<K>(N[lane], R[lane], k, ra[lane], rb[lane]);
				else { ra[lane] = k == 2 ? K2a[lane] : K3a[lane]; rb[lane] = k == 2 ? K2b[lane] : K3b[lane]; }
				if (k == 1) { dk_pick<K>(N[lane], R[lane], 2, K2a[lane], K2b[lane]); dk_pick<K>(N[lane], R[lane], 3, K3a[lane], K3b[lane]); }
			} else if (k == 0) { dk_pick<K>(N[lane], R[lane], 0, ra[lane], rb[lane]); dk_pick<K>(N[lane], R[lane], 1, K3a[lane], K3b[lane]); }
			else { ra[lane] = K3a[lane]; rb[lane] = K3b[lane]; }
			if (p[lane].flush >= 0) dk_flush<K>(W, f, R[lane], p[lane].flush, y);
			if (p[lane].top_flush >= 0) dk_top_flush<K>(W, f, lane, p[lane].top_flush, y0);
			if (p[lane].top_fetch >= 0) dk_top_fetch<K>(f, lane, p[lane].top_fetch, y0, tt[lane]);
			if (p[lane].prm_fetch) dk_fetch_prm<K>(f, R[lane], p[lane].x + 1, y, np[par ^ 1][lane]);
			if (k == (DK_GS == 4 ? 2 : 0) && p[lane].grp_fetch) dk_fetch4<K>(dk_src<K>(f, R[lane], y), R[lane], p[lane].x + 2, f.wm, N[lane]);
		}
		static DkPrm P[64][2];
		bool any_edge = !E264_DBK_ZEROSKIP; // (the kernel's wave-uniform test: no macroblock of the wave has an edge to filter -> samples only move into the strips)
		for (int lane = 0; lane < 64; lane++)
			if (p[lane].act && dk_any_bs(np[par][lane]) != 0) any_edge = true;
		if (!any_edge) {
			for (int lane = 0; lane < 64; lane++)
				if (p[lane].act) dk_vcopy<K>(W, R[lane], ra[lane], rb[lane], p[lane].x);
			g_zero_steps++;
			continue;
		}
		g_filter_steps++;
		for (int lane = 0; lane < 64; lane++)
			if (p[lane].act) {
				dk_params<K>(np[par][lane], R[lane], P[lane]);
				dk_vpass<K>(W, P[lane][0], R[lane], ra[lane], rb[lane], p[lane].x);
			}
		for (int lane = 0; lane < 64; lane++)
			if (p[lane].act) dk_hpass<K>(W, P[lane][1], R[lane], p[lane].x);
	}
}
// split: 0 = mixed waves (e264_deblock_kernel), 1 = luma waves + chroma waves (e264_deblock2_kernel)
extern "C" __attribute__((visibility("default"))) int e264emu_deblock_frame2(const uint8_t *pkt, uint8_t *const *dpb, uint8_t *dbk, int split)
{
	E264Job job = {pkt, dpb, dbk, g_expand};
	FrameCtx f;
	if (!open_frame(f, job) || !f.dbk)
		return -1;
	if (!split) {
		for (int q = 0; q < (f.hm + DK_ROWS_OF(2) - 1) / DK_ROWS_OF(2); q++) emu_walk_group<2>(f, q);
	} else {
		for (int q = 0; q < (f.hm + DK_ROWS_OF(1) - 1) / DK_ROWS_OF(1); q++) emu_walk_group<1>(f, q); // (the two chains are independent: any order)
		for (int q = 0; q < (f.hm + DK_ROWS_OF(0) - 1) / DK_ROWS_OF(0); q++) emu_walk_group<0>(f, q);
	}
	return 0;
}
extern "C" __attribute__((visibility("default"))) int e264emu_deblock_frame(const uint8_t *pkt, uint8_t *const *dpb, uint8_t *dbk)
{
	return e264emu_deblock_frame2(pkt, dpb, dbk, 0);
}

// the four edge slots of one lane: lines[2][20] (positions -4..15 of the lane's two lines) filtered in place; prm: a RAW 64-byte record
extern "C" __attribute__((visibility("default"))) void e264emu_dk_filter(uint8_t *lines, const uint8_t *prm, int lane, int dir)
{
	const DkRole R = dk_role<2>(lane);
	s16x2 v[20];
	for (int k = 0; k < 20; k++) v[k] = (s16x2){(short)lines[k], (short)lines[20 + k]};
	uint8_t pieces[E264_DBK_BYTES];
	e264emu_dbk_pieces(prm, pieces);
	DkRaw raw;
	memcpy(&raw.v, pieces + (R.chroma ? 64 : 0) + R.seg * 16, 8);
	memcpy(&raw.h, pieces + (R.chroma ? 64 : 0) + R.seg * 16 + 8, 8);
	memcpy(&raw.w, pieces + 128 + (R.chroma ? 8 : 0), 8);
	DkPrm P[2];
	dk_params<2>(raw, R, P);
	dk_filter<2>(v, P[dir], R);
	for (int k = 0; k < 20; k++) { // (the pack back to bytes, as dk_vpass / dk_hpass do it: p0 / q0 may arrive unclipped, E264_DBK_SATPACK)
		const uint32_t b = (E264_DBK_SATPACK && dk_is_p0q0(k)) ? v_sat_pk_u8_i16(as_u(v[k])) : v_perm(0, as_u(v[k]), 0x0c0c0200u);
		lines[k] = (uint8_t)b; lines[20 + k] = (uint8_t)(b >> 8);
	}
}
