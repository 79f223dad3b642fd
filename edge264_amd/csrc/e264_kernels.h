// e264_kernels.h -- launch interface between the C-ABI back end and the gfx950 kernels.
#ifndef E264_KERNELS_H
#define E264_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

// One job = one coded frame of one stream: its command packet (already in HBM) and the
// table of the stream's DPB slots (device pointers, E264_MAX_SLOTS entries, NULL if unallocated).
struct E264Job {
	const uint8_t *packet;
	uint8_t *const *dpb;
	uint8_t *dbk; // per-stream scratch, E264_SCRATCH_BYTES(macroblocks) (NULL: no deblocking, no intra bitmap -- host tests only, the back end always has one)
	uint8_t *expand; // per-stream expansion buffer of a WIRE packet (version 5, include/edge264_compact.h: e264_expand_area_bytes), NULL for a version-4 packet
};
#define E264_DBK_BYTES 144 // sixteen 8-byte pieces in the layout of the deblocking kernel's lanes + 16 bytes for the whole macroblock (e264_dbkp.h)
// The scratch of a stream: the parameter records of n_mbs macroblocks, then the INTRA BITMAP of the picture being decoded: one uint16_t per
// (macroblock row, group of 16 macroblocks) = per row of a prediction tile, bit i = macroblock 16 g + i is Intra4x4 / 8x8 / 16x16 and this
// packet's to reconstruct.  Written by e264_pred_kernel (which reads every record anyway), read by e264_intra_kernel in the same submission
// to leave rows and 64-macroblock chunks without intra macroblocks alone (P / B pictures).  2 bytes per macroblock cover the worst shape (one column).
#define E264_SCRATCH_BYTES(n_mbs) ((size_t)(n_mbs) * (E264_DBK_BYTES + 2) + 64)
#define E264_BITMAP_OFF(n_mbs) ((size_t)(n_mbs) * E264_DBK_BYTES)

#define E264_RUN_NO_PRED 4 // (launcher-internal) no job of the batch has an inter or PCM macroblock: e264_pred_kernel is not launched and the intra kernel scans without its bitmap
#define E264_RUN_EXPAND 16 // (launcher-internal) some job is a wire packet that has not been unfolded yet: e264_expand_kernel runs first, on the same queue
#define E264_RUN_NO_L1 8   // (launcher-internal) no job of the batch predicts from list 1 (validated packets of I / P pictures): e264_dbkparam2_kernel<false>
// mode: bit0 reconstruction, bit1 deblocking, bit2 E264_RUN_NO_PRED, bit3 E264_RUN_NO_L1, bit4 E264_RUN_EXPAND.  waves: 4, 8 or 16 macroblock rows in flight per frame.
// max_mbs: largest macroblock count among the jobs; max_tiles: largest e264_pred_tiles() among the jobs.  marks: NULL or 5 events (boundaries of the 4 kernels).
// fork: NULL, or a second queue + events on which the parameter kernel runs beside the macroblock-parallel kernel
// (amarks: 2 events bracketing it there, recorded when marks != NULL).
struct E264Fork { hipStream_t aux; hipEvent_t forked, joined; hipEvent_t *amarks; int where; int n_nopred; int planes; }; // planes: bit 0 = e264_intra_planes_kernel for the split-off pictures, bit 1 = for an all-intra batch (E264_RUN_NO_PRED), bit 2 = e264_deblock2_planes_kernel for the whole batch: the back end's calls (few enough pictures for two CUs each) // where: 1 = beside the prediction kernel, 2 = beside the intra kernel
// n_nopred (round 6): the LAST n_nopred jobs of the table hold no inter / PCM macroblock (I pictures: known from their validation).  In a submission that
// mixes them with P / B pictures -- streams whose GOPs are not in phase -- their intra pass (one workgroup per picture, 2.7 ms for a 1080p I picture) runs on
// `aux` from the start of the submission, beside the parameter and prediction kernels of the others, and deblocking waits for both:
// max(intra of the I pictures, parameters + prediction + intra of the rest) instead of their sum.
// workgroups e264_pred_kernel needs for a picture of this size (its tile geometry is a build-time choice of the kernels)
extern "C" int e264_pred_tiles(int width_mbs, int height_mbs);
// build-time switches of the kernels ("" = product build; e264hip_build_flags hands it out)
extern "C" const char *e264_kernel_build_flags(void);
extern "C" hipError_t e264_launch_frames(const E264Job *jobs, int n_jobs, int max_mbs, int max_tiles, int mode, int waves, hipStream_t stream, hipEvent_t *marks,
	const E264Fork *fork);

// e264_expand_kernel alone (a batch's wire packets, on the queue of its upload); E264_RUN_EXPAND in e264_launch_frames' mode runs it in front of the four instead
extern "C" hipError_t e264_launch_expand(const E264Job *jobs, int n_jobs, int max_mbs, hipStream_t stream);

#endif
