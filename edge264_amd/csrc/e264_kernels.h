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
	uint8_t *dbk; // per-stream scratch: E264_DBK_BYTES per macroblock (bS, alpha, beta, indexA), NULL = no deblocking
};
#define E264_DBK_BYTES 256 // sixteen 16-byte pieces in the layout of the deblocking kernel's lanes (e264_dbkp.h)

// mode: bit0 reconstruction, bit1 deblocking.  waves: 4, 8 or 16 macroblock rows in flight per frame.
// max_mbs: largest macroblock count among the jobs; max_tiles: largest e264_pred_tiles() among the jobs.  marks: NULL or 5 events (boundaries of the 4 kernels).
// fork: NULL, or a second queue + events on which the parameter kernel runs beside the macroblock-parallel kernel
// (amarks: 2 events bracketing it there, recorded when marks != NULL).
struct E264Fork { hipStream_t aux; hipEvent_t forked, joined; hipEvent_t *amarks; int where; }; // where: 1 = beside the prediction kernel, 2 = beside the intra kernel
// workgroups e264_pred_kernel needs for a picture of this size (its tile geometry is a build-time choice of the kernels)
extern "C" int e264_pred_tiles(int width_mbs, int height_mbs);
// build-time switches of the kernels ("" = product build; e264hip_build_flags hands it out)
extern "C" const char *e264_kernel_build_flags(void);
extern "C" hipError_t e264_launch_frames(const E264Job *jobs, int n_jobs, int max_mbs, int max_tiles, int mode, int waves, hipStream_t stream, hipEvent_t *marks,
	const E264Fork *fork);

#endif
