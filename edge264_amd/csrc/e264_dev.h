// e264_dev.h -- device-side definitions shared by the gfx950 kernels (e264_kernels.hip, e264_pred.h):
// address-space pointer types, small integer helpers, the spec tables, FrameCtx / open_frame.
//
// The kernels' source is also compiled for the HOST by the test suite (tests/emu/: every thread of a workgroup is run
// phase by phase by a plain C++ loop and the result is compared with the CPU oracle before any GPU time is spent).  That
// build pre-defines the three hooks below (qualifiers, address spaces, the handful of byte-permute intrinsics); the
// product build never does.
#ifndef E264_DEV_H
#define E264_DEV_H
#include <stdint.h>
#include <stddef.h>
#include "../../include/edge264_cmd.h"
#include "e264_kernels.h"

#ifndef E264_DEV
#define E264_DEV __device__ __forceinline__
#endif
#ifndef E264_AS_GLOBAL
#define E264_AS_GLOBAL __attribute__((address_space(1)))
#define E264_AS_CONST __attribute__((address_space(4)))
#endif

namespace {


// ---------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------
E264_DEV int clip3i(int lo, int hi, int v) { return min(max(v, lo), hi); }
E264_DEV int clip255(int v) { return min(max(v, 0), 255); }
E264_DEV int sat16(int v) { return min(max(v, -32768), 32767); }
E264_DEV int w16(int v) { return (int)(int16_t)v; }
// build-time experiment switches (`make variant NAME=.. DEFS=-D..` builds one library per setting, tools/visits/gpu_ab.sh compares them)
#ifdef E264_ABL_NOBH
#define E264_ABL_BH && false
#else
#define E264_ABL_BH
#endif
#ifndef E264_LUMA_PACKED
#define E264_LUMA_PACKED 1 // luma interpolation in packed 16-bit arithmetic (two samples per VALU instruction)
#endif
typedef uint8_t E264_AS_GLOBAL gu8;   // global memory, so that loads/stores are global_* not flat_*
typedef uint32_t E264_AS_GLOBAL gu32;
typedef uint16_t E264_AS_GLOBAL gu16;
typedef uint32_t E264_AS_GLOBAL __attribute__((aligned(1))) gu32u;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef v4u E264_AS_GLOBAL __attribute__((aligned(4))) gv4u; // dword-aligned is all the strides guarantee (stride_C/2 of a 4096-wide frame)
typedef v2u E264_AS_GLOBAL __attribute__((aligned(4))) gv2u; // unaligned dword (global memory allows it on gfx9+)
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef v3u E264_AS_GLOBAL __attribute__((aligned(4))) gv3u;
typedef int16_t E264_AS_GLOBAL gi16;
// the command packet is read-only for every kernel: constant address space => uniform reads become
// scalar loads (s_load) and the values live in SGPRs
typedef const E264FrameHdr E264_AS_CONST *chdr_t;
typedef const E264SliceParams E264_AS_CONST *cslice_t;
typedef const E264Mb E264_AS_CONST *cmb_t;
typedef const E264Motion E264_AS_GLOBAL *gmotion_t;
typedef const uint8_t E264_AS_CONST *cu8p;
typedef uint8_t *generic_u8p;
typedef const generic_u8p E264_AS_GLOBAL *gdpb_t;
#ifndef E264_HOST_INTRINSICS
E264_DEV int lane_id() { return (int)(threadIdx.x & 63); }
// All LDS scratch is private to one wave; LDS operations of a wave execute in order, so a
// compiler-level fence is all that is needed between producer and consumer lanes.
E264_DEV void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
#endif

__constant__ uint8_t c_BX[16] = {0, 4, 0, 4, 8, 12, 8, 12, 0, 4, 0, 4, 8, 12, 8, 12};
__constant__ uint8_t c_BY[16] = {0, 0, 4, 4, 0, 0, 4, 4, 8, 8, 12, 12, 8, 8, 12, 12};
E264_DEV int BXf(int k) { return ((k & 1) << 2) | ((k & 4) << 1); }
E264_DEV int BYf(int k) { return ((k & 2) << 1) | ((k & 8)); }
E264_DEV int blk_of(int bx, int by) { return (by >> 1) * 8 + (bx >> 1) * 4 + (by & 1) * 2 + (bx & 1); }

// normAdjust4x4 / normAdjust8x8 (edge264_residual.c:77-98) as arithmetic on immediates: byte m of a 64-bit constant per
// position class.  (A table in constant memory read with a per-lane index is a vector-memory round trip in the middle
// of every transform.)
E264_DEV int na_byte(uint64_t t, int m) { return (int)(t >> (8 * m)) & 255; }
#define NA4_0 0x12100e0d0b0aull /* class 0 (even,even): 10 11 13 14 16 18 */
#define NA4_1 0x1d1917141210ull /* class 1 (odd,odd)  : 16 18 20 23 25 29 */
E264_DEV int norm4(int m, int pos)
{
	int i = pos >> 2, j = pos & 3;
	const int a = na_byte(NA4_0, m), b = na_byte(NA4_1, m), c = na_byte(0x171412100e0dull, m);
	return (i & j & 1) ? b : ((i | j) & 1) ? c : a;
}
E264_DEV int norm8(int m, int pos)
{
	int i = pos >> 3, j = pos & 7, k;
	const int v0 = na_byte(0x24201c1a1614ull, m), v1 = na_byte(0x201c19171312ull, m), v2 = na_byte(0x3a332d2a2320ull, m);
	const int v3 = na_byte(0x221e1a181513ull, m), v4 = na_byte(0x2e2823211c19ull, m), v5 = na_byte(0x2b26211f1a18ull, m);
	if ((i & 3) == 0 && (j & 3) == 0) k = v0;
	else if (i & j & 1) k = v1;
	else if ((i & 3) == 2 && (j & 3) == 2) k = v2;
	else if (((i & 3) == 0 && (j & 1)) || ((i & 1) && (j & 3) == 0)) k = v3;
	else if (((i & 3) == 0 && (j & 3) == 2) || ((i & 3) == 2 && (j & 3) == 0)) k = v4;
	else k = v5;
	return k;
}

__constant__ uint8_t c_alpha[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28,
	32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
__constant__ uint8_t c_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8,
	9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
__constant__ uint8_t c_tc0[3][52] = {
	{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13},
	{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17},
	{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25}};

// ---- the byte-permute / funnel-shift VALU instructions the kernels use by name -------------------------------------
#ifndef E264_HOST_INTRINSICS
E264_DEV uint32_t v_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }          // v_perm_b32
E264_DEV uint32_t v_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); } // ({hi,lo} >> 8*(sh&3))
E264_DEV uint32_t v_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }   // ({hi,lo} >> (sh&31))
E264_DEV uint32_t v_lerp_u8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_lerp(a, b, c); }              // per byte (a + b + (c & 1)) >> 1
E264_DEV int mul24(int a, int b) { return a * b; } // operands within 24 bits signed.  (As inline asm v_mul_i32_i24 it bought nothing -- v_mul_lo_u32 issues at the VOP3 rate on gfx950, profiles/r02_valu_rate.txt -- and cost s_nops the compiler puts around asm it cannot see into.)
E264_DEV uint32_t v_sat_pk_u8_i16(uint32_t v) // two int16 -> two bytes with unsigned saturation in bits 15:0; no builtin
{
	uint32_t r;
	asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(v));
	return r;
}
#endif
// ---- packed 16-bit arithmetic (v_pk_*_i16: two samples per VALU instruction) ------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
E264_DEV s16x2 as_s2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
E264_DEV uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// pair (byte i, byte i+1) of the 8 bytes {hi, lo}, zero-extended to 16 bits each (v_perm_b32; selector 0x0c = 0x00)
template <int I>
E264_DEV s16x2 pair_at(uint32_t hi, uint32_t lo)
{
	return as_s2(v_perm(hi, lo, 0x0c000c00u | (uint32_t)I | (uint32_t)(I + 1) << 16));
}
// the 8 pairs Q_i = (p_i, p_i+1), i = 0..7, of the 9 samples of one row
E264_DEV void pairs9(const uint32_t w[3], s16x2 Q[8])
{
	Q[0] = pair_at<0>(w[1], w[0]); Q[1] = pair_at<1>(w[1], w[0]); Q[2] = pair_at<2>(w[1], w[0]); Q[3] = pair_at<3>(w[1], w[0]);
	Q[4] = pair_at<0>(w[2], w[1]); Q[5] = pair_at<1>(w[2], w[1]); Q[6] = pair_at<2>(w[2], w[1]); Q[7] = pair_at<3>(w[2], w[1]);
}
// the same taps on pairs of SAMPLES (0..255 each): the three sums cannot carry from one half into the other, so they are
// plain 32-bit adds (v_add_u32 issues at twice the rate of the packed and three-operand instructions on gfx950:
// tools/calib/valu_rate.hip)
E264_DEV s16x2 tap6u(s16x2 a, s16x2 b, s16x2 c, s16x2 d, s16x2 e, s16x2 f)
{
	const s16x2 k5 = {-5, -5}, k20 = {20, 20};
	const s16x2 af = as_s2(as_u(a) + as_u(f)), be = as_s2(as_u(b) + as_u(e)), cd = as_s2(as_u(c) + as_u(d));
	return be * k5 + (cd * k20 + af);
}
// the same + 16 (the rounding of (x + 16) >> 5), folded into the first sum: one three-operand add instead of an add and a packed add
E264_DEV s16x2 tap6u16(s16x2 a, s16x2 b, s16x2 c, s16x2 d, s16x2 e, s16x2 f)
{
	const s16x2 k5 = {-5, -5}, k20 = {20, 20};
	const s16x2 af = as_s2(as_u(a) + as_u(f) + 0x00100010u), be = as_s2(as_u(b) + as_u(e)), cd = as_s2(as_u(c) + as_u(d));
	return be * k5 + (cd * k20 + af);
}
// two pairs of int16 -> 4 bytes, each clipped to 0..255 (packus)
E264_DEV uint32_t packus4(s16x2 lo, s16x2 hi) { return v_perm(v_sat_pk_u8_i16(as_u(hi)), v_sat_pk_u8_i16(as_u(lo)), 0x05040100u); }
E264_DEV s16x2 tap6p(s16x2 a, s16x2 b, s16x2 c, s16x2 d, s16x2 e, s16x2 f)
{
	const s16x2 k5 = {5, 5}, k20 = {20, 20};
	return (a + f) - (b + e) * k5 + (c + d) * k20;
}
// sixtapHV + (x + 32) >> 6, clipped: 16-bit lanes wrap exactly like the reference's int16 vectors (inter.c:4-9,14)
E264_DEV s16x2 centre6p(s16x2 t0, s16x2 t1, s16x2 t2, s16x2 t3, s16x2 t4, s16x2 t5)
{
	const s16x2 s2 = {2, 2}, s6 = {6, 6}, r32 = {32, 32}, z = {0, 0}, m = {255, 255};
	const s16x2 af = t0 + t5, be = t1 + t4, cd = t2 + t3;
	const s16x2 x1 = af - be;
	const s16x2 x2 = (x1 >> s2) + (cd - be);
	const s16x2 x3 = (x2 >> s2) + cd;
	return __builtin_elementwise_min(__builtin_elementwise_max((x3 + r32) >> s6, z), m);
}
E264_DEV s16x2 half5p(s16x2 v) // clip255((v + 16) >> 5)
{
	const s16x2 s5 = {5, 5}, r16 = {16, 16}, z = {0, 0}, m = {255, 255};
	return __builtin_elementwise_min(__builtin_elementwise_max((v + r16) >> s5, z), m);
}
// One quadrant (list l, 8x8 block q) of a macroblock's compact motion record (edge264_cmd.h): its reference bytes
// {refPic, refIdx, 0, 0} and the vectors of its four 4x4 blocks (zig order inside the quadrant).  mot_off / mot_hdr: the
// directory words of E264Mb.modes.  Returns false when the list does not predict the quadrant.
E264_DEV uint32_t mot_nmv(uint32_t sub) { return 0x4221u >> (sub * 4) & 15u; } // vectors of a quadrant: 8x8, 8x4, 4x8, 4x4
E264_DEV bool mot_quadrant(const gu8 *motion, uint32_t mot_off, uint32_t h, int l, int q, uint32_t &refword, uint32_t mv[4])
{
	const bool uni = E264_MOT_UNI(h, l);
	if (!uni && !E264_MOT_USED(h, l * 4 + q))
		return false;
	uint32_t off = mot_off;
	if (l) { // skip the list-0 part
		uint32_t n0 = 0;
#pragma unroll
		for (int k = 0; k < 4; k++)
			n0 += E264_MOT_USED(h, k) ? 4 + 4 * mot_nmv(E264_MOT_SUB(h, k)) : 0;
		off += E264_MOT_UNI(h, 0) ? 8 : n0;
	}
	const gu32 *rec = (const gu32 *)(motion + off);
	if (uni) {
		refword = rec[0];
		mv[0] = mv[1] = mv[2] = mv[3] = rec[1];
		return true;
	}
	uint32_t skip = 0; // dwords of the used quadrants before q
#pragma unroll
	for (int k = 0; k < 3; k++)
		if (k < q && E264_MOT_USED(h, l * 4 + k)) skip += 1 + mot_nmv(E264_MOT_SUB(h, l * 4 + k));
	rec += skip;
	refword = rec[0];
	const uint32_t sub = E264_MOT_SUB(h, l * 4 + q);
	const uint32_t v0 = rec[1], v1 = sub ? rec[2] : v0;
	if (sub == 3) { mv[0] = v0; mv[1] = v1; mv[2] = rec[3]; mv[3] = rec[4]; }
	else if (sub == 2) { mv[0] = v0; mv[1] = v1; mv[2] = v0; mv[3] = v1; }
	else { mv[0] = v0; mv[1] = v0; mv[2] = v1; mv[3] = v1; }
	return true;
}

struct FrameCtx {
	chdr_t h;
	cslice_t slices;
	cmb_t mbs;
	const gu8 *mbs_g;  // the same array through a per-lane (global) pointer
	const gu8 *motion; // compact motion records (edge264_cmd.h E264_MOT_*), NULL if the frame has no inter macroblock
	const gu8 *payload;
	gdpb_t dpb;
	const generic_u8p *dpb_lds; // the same table staged in LDS (mbpar kernel): no dependent global round trip per reference
	gu8 *cur;
	int W, H;          // luma samples
	int wm, hm;        // macroblocks
	int sY, sC;        // strides
	uint32_t psY;      // plane_size_Y
	gu8 *dbk;          // per-MB deblocking parameters (E264_DBK_BYTES each), written by e264_dbkparam2_kernel
};

E264_DEV gu8 *plane_base(const FrameCtx &f, gu8 *base, int pl)
{
	return pl == 0 ? base : base + f.psY + (pl == 2 ? (f.sC >> 1) : 0);
}

E264_DEV bool open_frame(FrameCtx &f, const E264Job &job)
{
	const uint8_t *pkt = job.packet;
	chdr_t h = (chdr_t)pkt;
	if (h->magic != E264_MAGIC)
		return false;
	f.h = h;
	f.slices = (cslice_t)(pkt + h->slices_off);
	f.payload = (const gu8 *)(pkt + h->payload_off);
	if (h->version == E264_VERSION) {
		f.mbs = (cmb_t)(pkt + h->mbs_off);
		f.mbs_g = (const gu8 *)(pkt + h->mbs_off);
		f.motion = h->motion_off ? (const gu8 *)(pkt + h->motion_off) : nullptr;
	} else if (h->version == 5u && job.expand) {
		// a wire packet (include/edge264_compact.h): records and motion section are where e264_expand_kernel has put them, the rest where it lies
		f.mbs = (cmb_t)job.expand;
		f.mbs_g = (const gu8 *)job.expand;
		const uint32_t n_compact = *(const uint32_t E264_AS_CONST *)(pkt + h->mbs_off); // E264CompactHdr.n_compact
		f.motion = (h->motion_off || n_compact) ? (const gu8 *)job.expand + 32u * (uint32_t)h->width_mbs * (uint32_t)h->height_mbs : nullptr;
	} else
		return false;
	f.dpb_lds = nullptr;
	f.dpb = (gdpb_t)job.dpb;
	f.cur = (gu8 *)job.dpb[h->dst_slot];
	f.wm = h->width_mbs; f.hm = h->height_mbs;
	f.W = f.wm * 16; f.H = f.hm * 16;
	f.sY = (int)h->stride_Y; f.sC = (int)h->stride_C;
	f.psY = h->plane_size_Y;
	f.dbk = (gu8 *)job.dbk;
	return f.cur != nullptr;
}

} // namespace
#endif
