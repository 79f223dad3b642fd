/*
 * edge264_cmd.h -- the per-frame "command packet" that crosses the drop-in
 * boundary between the edge264 C front end (bitstream side: CAVLC/CABAC,
 * mvpred, DPB bookkeeping -- stays on the host) and the MI355X macroblock
 * reconstruction back end (sample side: residual, intra, inter, deblock).
 *
 * One packet == one coded frame.  It is plain C, position independent (only
 * byte offsets, no pointers) so the same bytes are consumed by
 *   - the HIP kernels   (edge264_amd/csrc/, after one hipMemcpyAsync),
 *   - the CPU oracle    (oracle/e264_oracle.c, test infrastructure only),
 *   - the on-disk capture/replay files (tools/, tests/golden/).
 *
 * Information content follows what the reference's leaf functions read
 * (/root/reference citations, file:line):
 *   slice constants ......... src/edge264_internal.h:223-261 (Edge264Task),
 *                             :310 (implicit_weights)
 *   per-MB metadata ......... src/edge264_internal.h:128-143 (Edge264Macroblock)
 *   intra modes ............. post-remap internal modes, src/edge264_slice.c:573-594,
 *                             619, 649, 881; enums src/edge264_internal.h:564-634
 *   coefficient blocks ...... ctx->c[] as handed to add_idct4x4/add_idct8x8/
 *                             transform_dc4x4/transform_dc2x2
 *                             (src/edge264_residual.c:108, 194, 352, 456): NOT
 *                             dequantised, in the reference's transposed order
 *                             c[x*4+y] / c[x*8+y]
 *   motion .................. mb->refPic[8], mb->refIdx[8], mb->mvs[64] as read
 *                             by decode_inter (src/edge264_inter.c:1108-1135) and
 *                             deblock_mb (src/edge264_deblock.c:927-1123)
 *   PCM ..................... raw samples, src/edge264_slice.c:914-935
 *
 * Layout of a packet (all offsets from the first byte of E264FrameHdr):
 *   [E264FrameHdr][E264SliceParams x n_slices][E264Mb x n_mbs][motion records of the inter macroblocks
 *   (only if the frame has any)][payload]                        n_mbs = width_mbs*height_mbs
 * Every section starts on a 16-byte boundary.  E264Mb is indexed by macroblock address; the motion record of an
 * inter macroblock is variable-sized (one vector for a 16x16 partition ... 16 for 4x4 partitions, per list) and
 * found through the directory words its E264Mb carries (see E264Mb.modes).
 */
#ifndef EDGE264_CMD_H
#define EDGE264_CMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E264_MAGIC   0x34363245u /* "E264" little endian */
#define E264_VERSION 4u      /* 3: motion sized by partition (a 16x16 macroblock carries one vector, not 32);
                                4: E264_MBF_LEV8 (one byte per AC level where they fit), E264_MBF_DONE */
#define E264_MAX_SLOTS 32        /* DPB slots per decoder, src/edge264_internal.h:402 */

/* Macroblock kinds (what the reconstruction pass has to do). */
enum {
	E264_MB_ABSENT = 0, /* never decoded (lost slice): samples left untouched */
	E264_MB_I4x4   = 1, /* I_NxN, transform 4x4: 16 x (predict, add residual) in zig order */
	E264_MB_I8x8   = 2, /* I_NxN, transform 8x8: 4 x (predict, add residual) */
	E264_MB_I16x16 = 3,
	E264_MB_PCM    = 4,
	E264_MB_INTER  = 5, /* P/B incl. skip/direct: mvs already expanded by the host */
};

/* E264Mb.flags */
#define E264_MBF_T8x8        0x01 /* transform_size_8x8_flag (luma residual is 8x8, deblock skips edges 1,3) */
#define E264_MBF_EDGE_LEFT   0x02 /* filter_edges bit0: deblock the left MB edge */
#define E264_MBF_EDGE_TOP    0x04 /* filter_edges bit1: deblock the top MB edge */
#define E264_MBF_DEBLOCK     0x08 /* filter_edges != 0: this MB is deblocked at all */
#define E264_MBF_DONE        0x10 /* the macroblock's samples were written by an EARLIER packet of the same picture: the record is
                                     here for its neighbours (bS, QP averages), nothing reconstructs it again and it carries no payload.
                                     Pictures are split into several packets only around a slice that failed (src/edge264_headers.c:
                                     486-529: the reference deblocks what it decoded, conceals, and a later copy of the slice decodes
                                     the macroblocks again on top) */

#define E264_MBF_LEV8        0x20 /* every level of the macroblock's AC blocks (4x4 luma / chroma, 8x8 luma) fits a signed byte and is
                                     stored as one: 16 / 64 bytes per block instead of 32 / 128.  The DC blocks stay int16.  (Almost
                                     every macroblock of a real stream: the coefficient payload, the largest part of a packet that
                                     crosses PCIe, halves.) */

/* E264Mb.coded bit positions */
#define E264_CODED_LUMA(k)    (1u << (k))        /* k = 4x4 block 0..15 in zig order; for T8x8 only k=0,4,8,12 */
#define E264_CODED_CHROMA(k)  (1u << (16 + (k))) /* k = 0..3 Cb, 4..7 Cr: AC block present */
#define E264_CODED_LUMA_DC    (1u << 24)         /* I16x16 DC block present (transform_dc4x4 was called) */
#define E264_CODED_CHROMA_DC  (1u << 25)         /* chroma DC block(s) present (transform_dc2x2 was called) */

typedef struct E264FrameHdr { /* 80 bytes */
	uint32_t magic;
	uint32_t version;
	uint32_t total_bytes;     /* whole packet */
	uint16_t width_mbs;
	uint16_t height_mbs;
	uint32_t stride_Y;        /* bytes between luma rows (src/edge264_headers.c:2032) */
	uint32_t stride_C;        /* bytes between chroma rows; a row is [Cb | Cr], Cr = Cb + stride_C/2 (:2041, :182) */
	uint32_t plane_size_Y;
	uint32_t plane_size_C;
	uint32_t n_slices;
	uint32_t slices_off;
	uint32_t mbs_off;
	uint32_t payload_off;
	uint32_t payload_bytes;
	int32_t  dst_slot;        /* DPB slot written by this frame */
	uint32_t ref_slots;       /* bitmask of DPB slots read by inter prediction */
	int32_t  frame_id;
	uint32_t n_coded_mbs;     /* macroblocks with kind != ABSENT */
	uint32_t n_inter_mbs;
	uint32_t motion_off;      /* motion records (up to payload_off), 0 if the frame has no inter macroblock */
	uint32_t stream_id;       /* capture files (a concatenation of packets): which decoder the packet belongs to;
	                             ignored by the kernels */
} E264FrameHdr;

typedef struct E264SliceParams { /* 2112 bytes */
	int8_t   slice_type;                  /* 0 P, 1 B, 2 I */
	int8_t   weighted_bipred_idc;         /* as seen by decode_inter (P slices alias weighted_pred_flag, headers.c:711-712) */
	int8_t   luma_log2_weight_denom;
	int8_t   chroma_log2_weight_denom;
	int8_t   FilterOffsetA;
	int8_t   FilterOffsetB;
	int8_t   disable_deblocking_filter_idc;
	int8_t   cabac;                       /* entropy_coding_mode_flag (only for diagnostics) */
	uint32_t first_mb;
	uint32_t reserved;
	uint8_t  weightScale4x4[6][16];       /* transposed-scan order, as pps.weightScale4x4 */
	uint8_t  weightScale8x8[6][64];
	int16_t  explicit_weights[3][64];     /* [Y,Cb,Cr][LX*32 + refIdx] */
	int8_t   explicit_offsets[3][64];
	uint8_t  implicit_weights[32][32];    /* [refIdxL0][refIdxL1], w1 + 64 */
	uint8_t  pad[16];
} E264SliceParams;

typedef struct E264Mb { /* 32 bytes, one per macroblock in raster order */
	uint8_t  kind;
	uint8_t  flags;
	uint8_t  qp[3];          /* QP_Y, QP_Cb, QP_Cr (mb->QP): dequant and deblock */
	uint8_t  chroma_mode;    /* IC8x8_* internal mode (intra kinds except PCM) */
	uint8_t  i16_mode;       /* I16x16_* internal mode */
	uint8_t  reserved0;
	uint16_t nz_mask;        /* bit k: luma 4x4 block k (zig order) has nonzero coefficients (mb->nC) */
	uint16_t slice;          /* index into the slice table */
	uint32_t coded;          /* E264_CODED_* */
	uint32_t payload_off;    /* byte offset from payload start, multiple of 8 */
	uint8_t  modes[8];       /* I4x4: 16 x 4-bit internal modes (block k in modes[k>>1] >> 4*(k&1));
	                            I8x8: 4 x 8-bit internal modes in modes[0..3];
	                            INTER: the motion directory: uint32 mot_off (modes[0..3]) = byte offset of the macroblock's
	                            motion record from motion_off, multiple of 4; uint32 mot_hdr (modes[4..7]) = its shape,
	                            E264_MOT_* below */
	uint16_t dbk_slice;      /* slice whose FilterOffsetA/B deblock this macroblock.  Normally == slice; the reference filters
	                            macroblocks whose deblocking had to wait for other slices (arbitrary slice order with
	                            disable_deblocking_filter_idc 0) with the constants of the slice that COMPLETES the picture
	                            (src/edge264_headers.c:538-567 run deblock_mb on the last task's context): kept bit-exact */
	uint16_t reserved1;
} E264Mb;

/* Payload of one macroblock, in this order (each item only if present):
 *   PCM   : 256 B luma (16 rows x 16), 64 B Cb, 64 B Cr (384 B)
 *   coded & LUMA_DC   : int16_t[16]  (c[0..15] at transform_dc4x4)
 *   coded & CHROMA_DC : int16_t[8]   (c[0..7]  at transform_dc2x2, Cb/Cr interleaved)
 *   luma blocks   : T8x8 ? int16_t[64] per coded 8x8 (bits 0,4,8,12) : int16_t[16] per coded 4x4
 *   chroma blocks : int16_t[16] per coded block k=0..7
 * With E264_MBF_LEV8 the luma and chroma blocks are int8_t[64] / int8_t[16].
 */
/* Motion of one macroblock in EXPANDED form (what mb->refPic / refIdx / mvs hold in the reference,
 * src/edge264_internal.h:139-142): the in-memory form of emitters, checkers and tools.  Packets carry the compact
 * record below; e264_motion_compact / e264_motion_expand convert (lossless for the used lists; vectors of unused
 * lists read as zero, which is what the reference stores for them). */
typedef struct E264Motion {
	int8_t  refPic[8];   /* [LX*4 + i8x8] DPB slot, -1 if the list is unused */
	int8_t  refIdx[8];   /* [LX*4 + i8x8] index used to look up weights, -1 if unused */
	int16_t mvs[64];     /* [LX*32 + i4x4*2 + {x,y}] quarter-pel, i4x4 in zig order */
} E264Motion;

/* mot_hdr (E264Mb.modes[4..7] of an inter macroblock):
 *   bits 0..7    used[LX*4 + q]   list LX predicts 8x8 quadrant q
 *   bits 8..9    uni[LX]          the macroblock is ONE 16x16 partition in list LX (all quadrants used, one reference,
 *                                 one vector): the list's part of the record is {int8 refPic, int8 refIdx, 0, 0, mv}
 *   bits 10..25  sub[LX*4 + q]    2 bits each: partition of the quadrant: 0 8x8 (1 vector), 1 8x4 (2: top, bottom),
 *                                 2 4x8 (2: left, right), 3 4x4 (4, zig order)
 * Motion record = list 0 part, then list 1 part.  A part is the uniform form (8 bytes) or, for every used quadrant in
 * increasing q, {int8 refPic, int8 refIdx, 0, 0} followed by its 1 / 2 / 4 vectors (int16 x, int16 y). */
#define E264_MOT_USED(h, lq)  ((h) >> (lq) & 1u)
#define E264_MOT_UNI(h, l)    ((h) >> (8 + (l)) & 1u)
#define E264_MOT_SUB(h, lq)   ((h) >> (10 + 2 * (lq)) & 3u)
static inline uint32_t e264_mot_nmv(uint32_t sub) { return sub == 0 ? 1u : sub == 3 ? 4u : 2u; }
static inline uint32_t e264_mot_part_bytes(uint32_t h, int l)
{
	if (E264_MOT_UNI(h, l)) return 8;
	uint32_t n = 0;
	for (int q = 0; q < 4; q++)
		if (E264_MOT_USED(h, l * 4 + q)) n += 4 + 4 * e264_mot_nmv(E264_MOT_SUB(h, l * 4 + q));
	return n;
}
static inline uint32_t e264_mot_record_bytes(uint32_t h) { return e264_mot_part_bytes(h, 0) + e264_mot_part_bytes(h, 1); }

/* expanded -> compact: writes the record (at most 2 * 4 * 20 = 160 bytes) and its shape word; returns the record's size */
static inline uint32_t e264_motion_compact(const E264Motion *m, uint8_t *rec, uint32_t *mot_hdr)
{
	uint32_t h = 0, n = 0;
	for (int l = 0; l < 2; l++) {
		const int32_t *mv = (const int32_t *)(const void *)&m->mvs[l * 32];
		int uni = 1;
		for (int q = 0; q < 4; q++) {
			uni &= m->refPic[l * 4 + q] >= 0 && m->refPic[l * 4 + q] == m->refPic[l * 4] && m->refIdx[l * 4 + q] == m->refIdx[l * 4];
			for (int j = 0; j < 4; j++) uni &= mv[q * 4 + j] == mv[0];
		}
		if (uni) {
			h |= 15u << (l * 4) | 1u << (8 + l);
			rec[n] = (uint8_t)m->refPic[l * 4]; rec[n + 1] = (uint8_t)m->refIdx[l * 4]; rec[n + 2] = rec[n + 3] = 0;
			*(int32_t *)(void *)(rec + n + 4) = mv[0];
			n += 8;
			continue;
		}
		for (int q = 0; q < 4; q++) {
			if (m->refPic[l * 4 + q] < 0)
				continue;
			const int32_t *v = mv + q * 4;
			const uint32_t sub = (v[0] == v[1] && v[0] == v[2] && v[0] == v[3]) ? 0 : (v[0] == v[1] && v[2] == v[3]) ? 1 : (v[0] == v[2] && v[1] == v[3]) ? 2 : 3;
			h |= 1u << (l * 4 + q) | sub << (10 + 2 * (l * 4 + q));
			rec[n] = (uint8_t)m->refPic[l * 4 + q]; rec[n + 1] = (uint8_t)m->refIdx[l * 4 + q]; rec[n + 2] = rec[n + 3] = 0;
			n += 4;
			const int idx[4][4] = {{0, 0, 0, 0}, {0, 2, 0, 0}, {0, 1, 0, 0}, {0, 1, 2, 3}};
			for (uint32_t j = 0; j < e264_mot_nmv(sub); j++, n += 4)
				*(int32_t *)(void *)(rec + n) = v[idx[sub][j]];
		}
	}
	*mot_hdr = h;
	return n;
}
/* compact -> expanded */
static inline void e264_motion_expand(uint32_t h, const uint8_t *rec, E264Motion *m)
{
	uint32_t n = 0;
	for (int l = 0; l < 2; l++) {
		int32_t *mv = (int32_t *)(void *)&m->mvs[l * 32];
		if (E264_MOT_UNI(h, l)) {
			for (int q = 0; q < 4; q++) { m->refPic[l * 4 + q] = (int8_t)rec[n]; m->refIdx[l * 4 + q] = (int8_t)rec[n + 1]; }
			for (int k = 0; k < 16; k++) mv[k] = *(const int32_t *)(const void *)(rec + n + 4);
			n += 8;
			continue;
		}
		for (int q = 0; q < 4; q++) {
			if (!E264_MOT_USED(h, l * 4 + q)) {
				m->refPic[l * 4 + q] = m->refIdx[l * 4 + q] = -1;
				mv[q * 4] = mv[q * 4 + 1] = mv[q * 4 + 2] = mv[q * 4 + 3] = 0;
				continue;
			}
			const uint32_t sub = E264_MOT_SUB(h, l * 4 + q);
			m->refPic[l * 4 + q] = (int8_t)rec[n]; m->refIdx[l * 4 + q] = (int8_t)rec[n + 1];
			const int32_t *v = (const int32_t *)(const void *)(rec + n + 4);
			n += 4 + 4 * e264_mot_nmv(sub);
			const int sel[4][4] = {{0, 0, 0, 0}, {0, 0, 1, 1}, {0, 1, 0, 1}, {0, 1, 2, 3}};
			for (int j = 0; j < 4; j++) mv[q * 4 + j] = v[sel[sub][j]];
		}
	}
}

#define E264_ALIGN16(x) (((x) + 15u) & ~15u)

/* Size in bytes of the payload of one macroblock (used by writers and checkers). */
static inline uint32_t e264_mb_payload_bytes(const E264Mb *m)
{
	uint32_t n = 0;
	if (m->flags & E264_MBF_DONE) return 0;
	if (m->kind == E264_MB_PCM) n += 384;
	if (m->coded & E264_CODED_LUMA_DC) n += 32;
	if (m->coded & E264_CODED_CHROMA_DC) n += 16;
	uint32_t ac = 0;
	if (m->flags & E264_MBF_T8x8) {
		for (int b = 0; b < 4; b++) ac += (m->coded >> (b * 4) & 1) * 128;
	} else {
		ac += (uint32_t)__builtin_popcount(m->coded & 0xffff) * 32;
	}
	ac += (uint32_t)__builtin_popcount(m->coded >> 16 & 0xff) * 32;
	return n + ((m->flags & E264_MBF_LEV8) ? ac >> 1 : ac);
}

/* the layout is an ABI shared by C (front end), HIP (back end) and numpy (edge264_amd/packet.py) */
#ifdef __cplusplus
#define E264_SIZE_CHECK(c, m) static_assert(c, m)
#else
#define E264_SIZE_CHECK(c, m) _Static_assert(c, m)
#endif
E264_SIZE_CHECK(sizeof(E264FrameHdr) == 80, "E264FrameHdr");
E264_SIZE_CHECK(sizeof(E264SliceParams) == 2112, "E264SliceParams");
E264_SIZE_CHECK(sizeof(E264Mb) == 32, "E264Mb");
E264_SIZE_CHECK(sizeof(E264Motion) == 144, "E264Motion (expanded form)");

#ifdef __cplusplus
}
#endif
#endif
