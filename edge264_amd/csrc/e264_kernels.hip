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
//   e264_dbkparam2_kernel (e264_dbkp.h)  one workgroup per 64 macroblocks: deblocking parameters (bS, alpha, beta, indexA)
//                        of EVERY macroblock from the command packet alone (nothing in the frame is read).
//   e264_pred_kernel (e264_pred.h)       one workgroup per tile of 16 x 4 macroblocks, one lane per 8x8 block: inter
//                        prediction + residual (+ PCM), which depend on nothing inside the frame.
//   e264_intra_kernel (e264_intra.h)     ONE WORKGROUP PER FRAME, ONE WAVE PER MACROBLOCK ROW: intra MBs only, row y
//                        may reconstruct macroblock x once row y-1 has finished macroblock x+1.
//   e264_deblock_kernel (e264_dbk.h)     one workgroup per frame, five macroblock rows per wave in lockstep, two lines
//                        per lane: the raster dependency order of H.264 in-loop deblocking.
// Progress counters live in LDS, so the hand-off between rows never leaves the CU: no
// agent-scope fences, no cross-XCD traffic, no placement assumption.  Chip-level parallelism
// comes from many independent streams (one frame of each per launch), the north-star workload
// (>=1000 concurrent streams).  Integer / byte work throughout, no MFMA.
//
// Pipeline rule (learnt with the phase profiler, -DE264_PHASE_TIMING / tools/visits/gpu_phase.sh): a stage that ISSUES loads for
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
#include "e264_dbk.h"
#include "e264_expand.h"

namespace {
// -DE264_PHASE_TIMING: wall cycles of the mbpar kernel's phases, summed over all waves (tools/visits/gpu_phase.sh reads them back
// through e264_debug_phase_cycles).  s_memtime at the phase boundaries drains the LGKM counter, so the numbers are a
// profile, not a benchmark.
#if defined(E264_DBK_TIMELINE) && !defined(E264_PHASE_TIMING) // only the start / end stamps of the deblocking kernel's groups of rows (workgroup 0): two s_memtime per 130 steps
__device__ unsigned long long g_phase[32];
__device__ unsigned long long g_timeline[128];
#endif
#ifdef E264_PHASE_TIMING
__device__ unsigned long long g_phase[32]; // [0..13] mbpar kernel, [16..29] deblock kernel
__device__ unsigned long long g_timeline[128]; // deblock kernel, workgroup 0: start / end of every group of five rows
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

} // namespace

#include "e264_intra.h"


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

// Inter prediction + residual of every inter / PCM macroblock: one workgroup per tile of 16 x PT_H macroblocks, one thread
// per 8x8 block; the phases are in e264_pred.h (and run on the host by tests/emu).
#ifndef E264_PRED_WAVES_PER_EU
#define E264_PRED_WAVES_PER_EU 4 // 128 VGPRs: 16 waves per CU (four 256-thread workgroups), so that one tile's barriers and first loads hide behind the others' arithmetic
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
	pred_phase_bitmap(L, f, t, tid);
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
template <bool HAS_L1>
static __device__ __forceinline__ void dbkparam2_body(DbkpLdsT<HAS_L1> &L, const E264Job *jobs)
{
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
#ifndef E264_ABL_DBKP_NOPIECES // timing ablation: the pieces are not built (whatever the LDS holds is written)
	dbkp_phase_pieces(L, tid);
#endif
	__syncthreads();
	dbkp_phase_store(L, f, a0, tid);
}
// Two forms of one kernel (e264_dbkp.h): <false> for batches the launcher knows to be without list-1 motion (18.6 KB of LDS, 52 VGPRs as the compiler likes them: eight
// workgroups per CU), <true> the general one -- 19.9 KB, and held to 64 VGPRs (two of them spill) so that the register file, too, takes eight workgroups
template <bool HAS_L1> __global__ void e264_dbkparam2_kernel(const E264Job *jobs);
template <> __global__ __launch_bounds__(DP_NT) void e264_dbkparam2_kernel<false>(const E264Job *jobs)
{
	__shared__ DbkpLdsT<false> L;
	dbkparam2_body<false>(L, jobs);
}
template <> __attribute__((amdgpu_waves_per_eu(8, 8))) __global__ __launch_bounds__(DP_NT) void e264_dbkparam2_kernel<true>(const E264Job *jobs)
{
	__shared__ DbkpLdsT<true> L;
	dbkparam2_body<true>(L, jobs);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void e264_intra_kernel(const E264Job *jobs, int use_bitmap)
{
	__shared__ IntraLds<NW> S;
	intra_kernel_body<NW>(S, jobs[blockIdx.x], (int)threadIdx.x, use_bitmap != 0);
}
// The same pass with a picture's LUMA and CHROMA on two workgroups (blockIdx.y): intra prediction and residual of the two never meet (a chroma block predicts from
// chroma samples only), so an I picture whose intra pass stands alone on the critical path -- a submission that mixes it with P / B pictures (E264Fork.n_nopred), one
// stream by itself -- gets two CUs instead of one.  Sixteen waves each; no bitmap (pictures without prediction work).
__global__ __launch_bounds__(1024) void e264_intra_planes_kernel(const E264Job *jobs)
{
	__shared__ IntraLds<16> S;
	if (blockIdx.y == 0) intra_kernel_body<16, 1>(S, jobs[blockIdx.x], (int)threadIdx.x, false);
	else intra_kernel_body<16, 2>(S, jobs[blockIdx.x], (int)threadIdx.x, false);
}


// In-loop deblocking: one workgroup per picture (e264_dbk.h; the phases run on the host in tests/emu).  A wave walks a GROUP of
// macroblock rows of kind K (DkGeom: mixed 5 rows luma + chroma, luma-only 8 rows, chroma-only 16 rows) from left to right; it waits for
// the wave that walks the group above through progress[] (macroblocks of that group's last row that have reached memory).
// dk_walk_group: one group q of kind K, by the calling wave.  progress: the counters of this kind's chain of groups.
template <int K>
static __device__ __forceinline__ void dk_walk_group(DkWaveT<K> &W, const FrameCtx &f, int *progress, const int q, const int lane, const int tl_slot)
{
	typedef DkGeom<K> G;
	const DkRole R = dk_role<K>(lane);
	const int wm = f.wm, last_step = dk_last_step<K>(wm);
	const int y0 = q * G::ROWS, y = y0 + R.g;
	const bool row_ok = !R.idle && y < f.hm, top = q > 0;
	const int lastg = min(G::ROWS, f.hm - y0) - 1; // the row the wave below waits for
	const DkSrc src = dk_src<K>(f, R, y);
#if defined(E264_PHASE_TIMING) || defined(E264_DBK_TIMELINE)
	if (blockIdx.x == 0 && lane == 0 && tl_slot < 64) g_timeline[2 * tl_slot] = __builtin_amdgcn_s_memtime();
#endif
	v4u tt = {0, 0, 0, 0};
	DkRaw p0 = {{0, 0}, {0, 0}, {0, 0}}, p1 = p0; // the lane's parameter pieces: the set this step uses and the set it requests for the next one
	v4u N[2 * DK_GS];       // samples of four (two) macroblocks of the lane's two rows (dk_fetch4), requested at steps t = 0 mod 4 (2)
	v4u K2a = tt, K2b = tt, K3a = tt, K3b = tt; // the last two (groups of 2: K3, the last one) of them, kept while the next group is on its way
	PH_DECL;
	// one step; k = (t + 2) & 3: which macroblock of its group the step filters (k = 2, 3: of the group before, out of K2 / K3);
	// groups of 2: k = t & 1, k = 1 out of K3, the step with k = 0 requests the next group;
	// sp: the parameter pieces of x, requested by the step before; sn: where this step requests those of x + 1
	auto step = [&](const int t, const int k, DkRaw &sp, DkRaw &sn) __attribute__((always_inline)) {
		const DkPlan p = dk_plan(t, R, row_ok, top, wm);
		// what earlier steps requested is picked up BEFORE this step's stores are issued: the compiler cannot count
		// conditional stores, any use of a loaded register after them is an s_waitcnt vmcnt(0) = a full drain
		if (p.top_commit >= 0) dk_top_commit<K>(W, f, lane, p.top_commit, y0, tt);
		v4u ra, rb;
		asm volatile("" :: "v"(sp.v), "v"(sp.h), "v"(sp.w)); // (the parameters of this step have landed)
		if (DK_GS == 4) {
			if (k < 2) dk_pick<K>(N, R, k, ra, rb);
			else { ra = k == 2 ? K2a : K3a; rb = k == 2 ? K2b : K3b; }
			if (k == 1) { dk_pick<K>(N, R, 2, K2a, K2b); dk_pick<K>(N, R, 3, K3a, K3b); } // N is overwritten in the next step
			asm volatile("" :: "v"(ra), "v"(rb), "v"(K2a), "v"(K2b), "v"(K3a), "v"(K3b)); // the copies happen here
		} else {
			if (k == 0) { dk_pick<K>(N, R, 0, ra, rb); dk_pick<K>(N, R, 1, K3a, K3b); } // N is overwritten further down in this step
			else { ra = K3a; rb = K3b; }
			asm volatile("" :: "v"(ra), "v"(rb), "v"(K3a), "v"(K3b));
		}
		if (p.flush >= 0) dk_flush<K>(W, f, R, p.flush, y);
		if (p.top_flush >= 0) dk_top_flush<K>(W, f, lane, p.top_flush, y0);
		PH(0);
		if (p.top_fetch >= 0) { // (wave-uniform) the rows above this group of 4 must have reached memory
			const int need = min(p.top_fetch * DK_GS + DK_GS, wm);
#ifndef E264_ABL_DBK_NOWAIT // timing ablation: the group above is not waited for (wrong samples along the seams): what the hand-off lag costs
			while (lds_load_relaxed(&progress[q - 1]) < need)
				__builtin_amdgcn_s_sleep(1);
#endif
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			dk_top_fetch<K>(f, lane, p.top_fetch, y0, tt);
		}
		PH(1);
		if (p.prm_fetch) dk_fetch_prm<K>(f, R, p.x + 1, y, sn);
		if (k == (DK_GS == 4 ? 2 : 0) && p.grp_fetch) dk_fetch4<K>(src, R, p.x + 2, wm, N);
		wave_sync();
		PH(2);
		DkPrm P[2];
#if E264_DBK_ZEROSKIP // (wave-uniform) a step in which no macroblock of the wave has an edge to filter only moves its samples into the strips
		if (!__any(p.act && dk_any_bs(sp) != 0)) {
			if (p.act) dk_vcopy<K>(W, R, ra, rb, p.x);
			wave_sync();
		} else
#endif
		{
			if (p.act) {
				dk_params<K>(sp, R, P);
				PH(3);
				dk_vpass<K>(W, P[0], R, ra, rb, p.x);
			}
			wave_sync();
			PH(4);
			if (p.act) dk_hpass<K>(W, P[1], R, p.x);
			wave_sync();
		}
		PH(5);
		if (p.publish) {
			// the stores at the top of this step must be visible to the wave below before the counter moves
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == lastg * G::LANES)
				__hip_atomic_store(&progress[q], dk_progress(t, lastg, wm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		PH(6);
	};
#pragma unroll 1
	for (int t = DK_FIRST_STEP; t <= last_step; t += DK_GS) { // unrolled by four (two): a group of macroblocks per fetch, registers by name
		if (DK_GS == 4) {
			step(t, 2, p0, p1);
			step(t + 1, 3, p1, p0);
			step(t + 2, 0, p0, p1);
			step(t + 3, 1, p1, p0);
		} else {
			step(t, 0, p0, p1);
			step(t + 1, 1, p1, p0);
		}
	}
#if defined(E264_PHASE_TIMING) || defined(E264_DBK_TIMELINE)
	if (blockIdx.x == 0 && lane == 0 && tl_slot < 64) g_timeline[2 * tl_slot + 1] = __builtin_amdgcn_s_memtime();
#endif
#ifndef E264_PHASE_INTRA
	PH_FLUSH_DBK(lane);
#endif
}

// (a) mixed waves (rounds 2 and 3): NW waves take the groups of five rows round-robin
template <int NW>
__global__ __launch_bounds__(NW * 64) void e264_deblock_kernel(const E264Job *jobs)
{
	__shared__ DkWaveT<2> lds[NW];
	__shared__ int progress[(E264_MAX_ROWS + DK_ROWS_OF(2) - 1) / DK_ROWS_OF(2)];
	const int lane = lane_id();
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	FrameCtx f;
	if (!open_frame(f, jobs[blockIdx.x]) || !f.dbk)
		return;
	const int nquint = (f.hm + DK_ROWS_OF(2) - 1) / DK_ROWS_OF(2);
	for (int i = threadIdx.x; i < nquint; i += NW * 64)
		progress[i] = 0;
	__syncthreads();
#pragma unroll 1
	for (int q = wave; q < nquint; q += NW) {
#ifndef E264_DBK_NO_PRIO // (1.045 - 1.059 -> 1.039 ms: profiles/r04_ablations.txt item 5)
		// Groups of rows form ONE dependency chain (a wave waits for the group above), and two waves share a SIMD: with equal priority the
		// OLDER wave wins the issue slots -- in its second pass (group w + NW) that is the wave whose work depends on its partner's
		// first-pass group (w + NW / 2).  Earlier passes get the higher priority, whatever the wave's age.
		{ const int pass = q / NW; if (pass == 0) __builtin_amdgcn_s_setprio(3); else if (pass == 1) __builtin_amdgcn_s_setprio(2); else if (pass == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
		dk_walk_group<2>(lds[wave], f, progress, q, lane, q);
	}
}

// (b) luma waves and chroma waves (round 4).  Luma and chroma are two independent chains of groups (8 / 16 rows each); the NW waves
// take the groups of BOTH chains from one list, luma and chroma interleaved in proportion, through a counter in LDS: a wave that
// finishes takes the next group of the list, whatever its kind, so the SIMDs stay evenly loaded (the mixed kernel's 14 groups of a
// 1080p picture fall 4 / 4 / 3 / 3 on the four SIMDs).  A group's predecessor is always earlier in the list: it has been taken.
union DkWaveAny { DkWaveT<0> l; DkWaveT<1> c; };
template <int NW>
__global__ __launch_bounds__(NW * 64) void e264_deblock2_kernel(const E264Job *jobs)
{
	__shared__ DkWaveAny lds[NW];
	__shared__ int progress_l[(E264_MAX_ROWS + DK_ROWS_OF(0) - 1) / DK_ROWS_OF(0)];
	__shared__ int progress_c[(E264_MAX_ROWS + DK_ROWS_OF(1) - 1) / DK_ROWS_OF(1)];
	__shared__ int next_task;
	const int lane = lane_id();
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	FrameCtx f;
	if (!open_frame(f, jobs[blockIdx.x]) || !f.dbk)
		return;
	const int nl = (f.hm + DK_ROWS_OF(0) - 1) / DK_ROWS_OF(0), nc = (f.hm + DK_ROWS_OF(1) - 1) / DK_ROWS_OF(1), total = nl + nc;
	for (int i = threadIdx.x; i < nl; i += NW * 64) progress_l[i] = 0;
	for (int i = threadIdx.x; i < nc; i += NW * 64) progress_c[i] = 0;
	if (threadIdx.x == 0) next_task = 0;
	__syncthreads();
#pragma unroll 1
	for (;;) {
		int task = 0;
		if (lane == 0) task = atomicAdd(&next_task, 1);
		task = __builtin_amdgcn_readfirstlane(task);
		if (task >= total)
			break;
#ifndef E264_DBK_ORDER
#define E264_DBK_ORDER 0 // the task list: 0 = luma and chroma groups interleaved in proportion (round 4), 1 = all luma groups first, 2 = all chroma groups first
#endif
		if (E264_DBK_ORDER == 1) {
			if (task < nl) dk_walk_group<0>(lds[wave].l, f, progress_l, task, lane, task);
			else dk_walk_group<1>(lds[wave].c, f, progress_c, task - nl, lane, 32 + task - nl);
		} else if (E264_DBK_ORDER == 2) {
			if (task < nc) dk_walk_group<1>(lds[wave].c, f, progress_c, task, lane, 32 + task);
			else dk_walk_group<0>(lds[wave].l, f, progress_l, task - nc, lane, task - nc);
		} else {
			// luma groups among the first i tasks of the list: (i * nl) / total; task i is a luma group iff that count grows at i + 1
			const int lb = task * nl / total, la = (task + 1) * nl / total;
			if (la > lb) dk_walk_group<0>(lds[wave].l, f, progress_l, lb, lane, lb);
			else dk_walk_group<1>(lds[wave].c, f, progress_c, task - lb, lane, 32 + task - lb);
		}
	}
}

// The same walk with a picture's luma groups on one workgroup and its chroma groups on another (blockIdx.y): the two chains never meet (separate samples, the same
// read-only parameters), so a picture that has the device to itself -- one stream, a small batch -- gets two CUs for the kernel that is most of its latency
// (one workgroup per picture: 0.9 ms of a P picture's 0.9).  The back end's call (E264Fork.planes bit 2: few pictures).
template <int NW>
__global__ __launch_bounds__(NW * 64) void e264_deblock2_planes_kernel(const E264Job *jobs)
{
	__shared__ DkWaveAny lds[NW];
	__shared__ int progress[(E264_MAX_ROWS + DK_ROWS_OF(0) - 1) / DK_ROWS_OF(0)]; // (the longer of the two chains: luma groups are the shorter ones)
	__shared__ int next_task;
	const int lane = lane_id();
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	FrameCtx f;
	if (!open_frame(f, jobs[blockIdx.x]) || !f.dbk)
		return;
	const bool chroma = blockIdx.y != 0;
	const int n = chroma ? (f.hm + DK_ROWS_OF(1) - 1) / DK_ROWS_OF(1) : (f.hm + DK_ROWS_OF(0) - 1) / DK_ROWS_OF(0);
	for (int i = threadIdx.x; i < n; i += NW * 64) progress[i] = 0;
	if (threadIdx.x == 0) next_task = 0;
	__syncthreads();
#pragma unroll 1
	for (;;) {
		int task = 0;
		if (lane == 0) task = atomicAdd(&next_task, 1);
		task = __builtin_amdgcn_readfirstlane(task);
		if (task >= n)
			break;
		if (chroma) dk_walk_group<1>(lds[wave].c, f, progress, task, lane, 32 + task);
		else dk_walk_group<0>(lds[wave].l, f, progress, task, lane, task);
	}
}

// Build-time switches this code object was compiled with, as a space-separated list ("" = the product build).  The timing
// ablations (E264_ABL_*, E264_PHASE_*) produce WRONG SAMPLES on purpose: a library that reports one is refused by every loader
// (e264hip_device_open, edge264_amd/backend.py, the front end) unless E264_ALLOW_ABLATION=1 is set -- the A/B tooling sets it.
extern "C" const char *e264_kernel_build_flags(void)
{
	return ""
#if E264_DBK_GS == 2 // (not an ablation: a bit-exact variant; named so that the back end knows which wave counts exist)
		" E264_DBK_GS=2"
#endif
#ifdef E264_ABL_NOBH
		" E264_ABL_NOBH"
#endif
#ifdef E264_ABL_NOEDGE
		" E264_ABL_NOEDGE"
#endif
#ifdef E264_ABL_NOLOAD
		" E264_ABL_NOLOAD"
#endif
#ifdef E264_ABL_NOLUMA
		" E264_ABL_NOLUMA"
#endif
#ifdef E264_ABL_DBKP_NOMOT
		" E264_ABL_DBKP_NOMOT"
#endif
#ifdef E264_ABL_DBK_NOFILTER
		" E264_ABL_DBK_NOFILTER"
#endif
#ifdef E264_ABL_DBK_NOLOAD
		" E264_ABL_DBK_NOLOAD"
#endif
#ifdef E264_ABL_DBK_NOSTORE
		" E264_ABL_DBK_NOSTORE"
#endif
#ifdef E264_ABL_INTRA_NOWAIT
		" E264_ABL_INTRA_NOWAIT"
#endif
#ifdef E264_ABL_INTRA_STOP
		" E264_ABL_INTRA_STOP"
#endif
#ifdef E264_ABL_DBKP_STORE1
		" E264_ABL_DBKP_STORE1"
#endif
#ifdef E264_ABL_DBKP_NOPIECES
		" E264_ABL_DBKP_NOPIECES"
#endif

#ifdef E264_ABL_INTRA_NOFENCE
		" E264_ABL_INTRA_NOFENCE"
#endif
#ifdef E264_ABL_DBK_NOWAIT
		" E264_ABL_DBK_NOWAIT"
#endif
#ifdef E264_PHASE_TIMING
		" E264_PHASE_TIMING"
#endif
#ifdef E264_DBK_TIMELINE
		" E264_DBK_TIMELINE"
#endif
#ifdef E264_PHASE_INTRA
		" E264_PHASE_INTRA"
#endif
#ifdef E264_PRED_HPAIR
		" E264_PRED_HPAIR"
#endif
#ifdef E264_PRED_CHROMA_LAST
		" E264_PRED_CHROMA_LAST"
#endif
#ifdef E264_PRED_CLASS_PACKED
		" E264_PRED_CLASS_PACKED"
#endif
		;
}

// ---------------------------------------------------------------------------------
// Kernel 0 (only for batches with wire packets): version 5 -> the record array and motion section of version 4 (e264_expand.h)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(XP_NT) void e264_expand_kernel(const E264Job *jobs)
{
	expand_thread(jobs[blockIdx.y], blockIdx.x * XP_NT + threadIdx.x, gridDim.x * XP_NT);
}

extern "C" int e264_pred_tiles(int width_mbs, int height_mbs)
{
	return ((width_mbs + PT_W - 1) / PT_W) * ((height_mbs + PT_H - 1) / PT_H);
}

extern "C" hipError_t e264_launch_expand(const E264Job *jobs, int n_jobs, int max_mbs, hipStream_t stream)
{
	if (n_jobs > 0)
		hipLaunchKernelGGL(e264_expand_kernel, dim3((max_mbs + XP_NT - 1) / XP_NT, n_jobs), dim3(XP_NT), 0, stream, jobs);
	return hipGetLastError();
}

extern "C" hipError_t e264_launch_frames(const E264Job *jobs, int n_jobs, int max_mbs, int max_tiles, int mode, int waves, hipStream_t stream, hipEvent_t *marks,
	const E264Fork *fork)
{
	if (n_jobs <= 0)
		return hipSuccess;
	// wire packets first (before the marks: they bracket the four kernels; the whole-run clocks contain this one)
	if (mode & E264_RUN_EXPAND)
		e264_launch_expand(jobs, n_jobs, max_mbs, stream);
	// marks (optional): 5 events recorded before / between / after the four launches
	if (marks) hipEventRecord(marks[0], stream);
	const bool dbkp = (mode & 2) != 0;
	const bool no_l1 = (mode & E264_RUN_NO_L1) != 0; // no packet of the batch predicts from list 1: the parameter kernel's small form (eight workgroups per CU)
	const int intra_waves_ = waves >> 8 ? waves >> 8 : waves & 255;
	auto launch_intra = [&](const E264Job *j, int n, hipStream_t q, int use_bitmap) {
		switch (intra_waves_) {
		case 4: hipLaunchKernelGGL(e264_intra_kernel<4>, dim3(n), dim3(256), 0, q, j, use_bitmap); break;
		case 16: hipLaunchKernelGGL(e264_intra_kernel<16>, dim3(n), dim3(1024), 0, q, j, use_bitmap); break;
		default: hipLaunchKernelGGL(e264_intra_kernel<8>, dim3(n), dim3(512), 0, q, j, use_bitmap); break;
		}
	};
	// a submission that mixes pictures without prediction work (the table's last n_nopred jobs: I pictures) with others: their intra pass starts NOW on the
	// second queue (E264Fork.n_nopred, e264_kernels.h); the rest of this function then sees the other jobs only, up to deblocking
	const int n_split = (fork && fork->aux && (mode & 1) && !(mode & E264_RUN_NO_PRED) && fork->n_nopred > 0 && fork->n_nopred < n_jobs) ? fork->n_nopred : 0;
	const int n_front = n_jobs - n_split;
	if (n_split) {
		hipEventRecord(fork->forked, stream);
		hipStreamWaitEvent(fork->aux, fork->forked, 0);
		if (marks) hipEventRecord(fork->amarks[0], fork->aux);
		if ((fork->planes & 1) && intra_waves_ == 16) hipLaunchKernelGGL(e264_intra_planes_kernel, dim3(n_split, 2), dim3(1024), 0, fork->aux, jobs + n_front);
		else launch_intra(jobs + n_front, n_split, fork->aux, 0);
		if (marks) hipEventRecord(fork->amarks[1], fork->aux);
		hipEventRecord(fork->joined, fork->aux);
	}
	// fork->where (optional): the parameter kernel reads nothing but the packet and is needed only by the deblocking kernel, so it can run on the second queue beside
	// the prediction kernel (1; rounds 1 - 4: no gain, that kernel fills every CU) or beside the intra kernel (2; round 6: no gain either, profiles/r06_ablations.txt item 1)
	const bool side = dbkp && fork && fork->aux && fork->where && !n_split; // (not when the second queue is taken)
	const int where = side ? (fork->where == 2 ? 2 : 1) : 0;
	auto launch_side = [&]() {
		hipEventRecord(fork->forked, stream);
		hipStreamWaitEvent(fork->aux, fork->forked, 0);
		if (marks) hipEventRecord(fork->amarks[0], fork->aux);
		if (no_l1) hipLaunchKernelGGL(e264_dbkparam2_kernel<false>, dim3((max_mbs + DP_MBS - 1) / DP_MBS, n_jobs), dim3(DP_NT), 0, fork->aux, jobs);
		else hipLaunchKernelGGL(e264_dbkparam2_kernel<true>, dim3((max_mbs + DP_MBS - 1) / DP_MBS, n_jobs), dim3(DP_NT), 0, fork->aux, jobs);
		if (marks) hipEventRecord(fork->amarks[1], fork->aux);
		hipEventRecord(fork->joined, fork->aux);
	};
	if (where == 1)
		launch_side();
	else if (dbkp && !side) {
		if (no_l1) hipLaunchKernelGGL(e264_dbkparam2_kernel<false>, dim3((max_mbs + DP_MBS - 1) / DP_MBS, n_jobs), dim3(DP_NT), 0, stream, jobs);
		else hipLaunchKernelGGL(e264_dbkparam2_kernel<true>, dim3((max_mbs + DP_MBS - 1) / DP_MBS, n_jobs), dim3(DP_NT), 0, stream, jobs);
	}
	if (marks) hipEventRecord(marks[1], stream);
	// an all-intra batch (every picture of an I launch: E264_RUN_NO_PRED) has nothing for the prediction kernel: 34 816 workgroups that load their records and
	// leave cost 0.12 ms per launch of 256 pictures; the intra kernel then scans without the bitmap those workgroups would have written
	const bool no_pred = (mode & E264_RUN_NO_PRED) != 0;
	if ((mode & 1) && !no_pred)
		hipLaunchKernelGGL(e264_pred_kernel, dim3(max_tiles, n_front), dim3(PT_NT), 0, stream, jobs, mode);
	if (marks) hipEventRecord(marks[2], stream);
	if (where == 2)
		launch_side();
	waves &= 255;
	if ((mode & 1) && no_pred && fork && (fork->planes & 2) && intra_waves_ == 16) // a FEW pictures, all without prediction work (one stream's I picture): two CUs each
		hipLaunchKernelGGL(e264_intra_planes_kernel, dim3(n_jobs, 2), dim3(1024), 0, stream, jobs);
	else if (mode & 1)
		launch_intra(jobs, n_front, stream, no_pred ? 0 : 1);
	if (n_split) hipStreamWaitEvent(stream, fork->joined, 0);
	if (side) hipStreamWaitEvent(stream, fork->joined, 0); // (before the mark: with the parameter kernel beside it, "intra" is the phase both share)
	if (marks) hipEventRecord(marks[3], stream);
	if ((mode & 2) && waves == 108 && fork && (fork->planes & 4)) // few pictures: two workgroups each (luma groups, chroma groups)
		hipLaunchKernelGGL(e264_deblock2_planes_kernel<8>, dim3(n_jobs, 2), dim3(512), 0, stream, jobs);
	else if (mode & 2) {
		switch (waves) { // waves per picture (default 8, set by the back end); 100 + n: luma / chroma waves (e264_deblock2_kernel)
#if E264_DBK_GS == 2 // strips of four macroblocks: 12.6 KB of LDS per wave, twelve waves (three per SIMD) fit the CU
		case 112: hipLaunchKernelGGL(e264_deblock2_kernel<12>, dim3(n_jobs), dim3(768), 0, stream, jobs); break;
		case 110: hipLaunchKernelGGL(e264_deblock2_kernel<10>, dim3(n_jobs), dim3(640), 0, stream, jobs); break;
#else
		case 112: case 110:
#endif
		case 108: hipLaunchKernelGGL(e264_deblock2_kernel<8>, dim3(n_jobs), dim3(512), 0, stream, jobs); break;
		case 107: hipLaunchKernelGGL(e264_deblock2_kernel<7>, dim3(n_jobs), dim3(448), 0, stream, jobs); break;
		case 106: hipLaunchKernelGGL(e264_deblock2_kernel<6>, dim3(n_jobs), dim3(384), 0, stream, jobs); break;
		case 2: hipLaunchKernelGGL(e264_deblock_kernel<2>, dim3(n_jobs), dim3(128), 0, stream, jobs); break;
		case 4: hipLaunchKernelGGL(e264_deblock_kernel<4>, dim3(n_jobs), dim3(256), 0, stream, jobs); break;
		case 8: hipLaunchKernelGGL(e264_deblock_kernel<8>, dim3(n_jobs), dim3(512), 0, stream, jobs); break;
		default: hipLaunchKernelGGL(e264_deblock_kernel<7>, dim3(n_jobs), dim3(448), 0, stream, jobs); break;
		}
	}
	if (marks) hipEventRecord(marks[4], stream);
	return hipGetLastError();
}

#if defined(E264_PHASE_TIMING) || defined(E264_DBK_TIMELINE)
extern "C" __attribute__((visibility("default"))) int e264_debug_phase_cycles(unsigned long long *out32, int reset)
{
	hipDeviceSynchronize();
	if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_phase), sizeof(g_phase)) != hipSuccess) return -1;
	if (out32 && hipMemcpyFromSymbol(out32 + 32, HIP_SYMBOL(g_timeline), sizeof(g_timeline)) != hipSuccess) return -1; // the caller has room for 32 + 128
	if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return -1; }
	return 0;
}
#endif
