/*
 * edge264_hip.h -- C ABI of the MI355X macroblock-reconstruction back end
 * (libedge264_hip.so, built from edge264_amd/csrc/ with hipcc for gfx950).
 *
 * This is the drop-in boundary of BASELINE.json's north_star: everything that
 * touches the BITSTREAM (CAVLC/CABAC, mvpred, DPB bookkeeping) stays in the
 * edge264 C front end; everything that touches SAMPLES is behind these entry
 * points.  Plain pointers and sizes only -- what a C front end (or cgo / JNI /
 * ctypes) binds.  Each function names the reference interface it replaces
 * (/root/reference, file:line).  INTEGRATION.md shows the reference-side glue.
 *
 * Threading: a E264Device may be shared by threads; one E264Stream (= one
 * Edge264Decoder) is driven by one thread at a time, like the reference's
 * n_threads==0 mode (src/edge264_headers.c:1285-1286).
 * Errors: 0 or a positive errno value like the reference (src/edge264_internal.h:1259-1263):
 * ENOMEM (device/pinned allocation), EINVAL (bad packet/slot), EIO (HIP runtime failure),
 * ENODEV (no gfx950 device).  Nothing aborts.
 */
#ifndef EDGE264_HIP_H
#define EDGE264_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "edge264_cmd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct E264Device E264Device; /* one per GPU: compute lanes (HIP queues), upload / download queues, kernels */
typedef struct E264Stream E264Stream; /* one per decoder: device DPB slots + host mirrors */

/* ---- device ------------------------------------------------------------ */
/* Selected in edge264_alloc where the reference picks its ISA variant
 * (src/edge264.c:161-216): open once per GPU, share between decoders. */
int  e264hip_device_open(int ordinal, E264Device **out);
void e264hip_device_close(E264Device *dev);
int  e264hip_device_sync(E264Device *dev);
const char *e264hip_last_error(void);

/* ---- decoder-side objects ---------------------------------------------- */
int  e264hip_stream_open(E264Device *dev, E264Stream **out);
void e264hip_stream_close(E264Stream *s);
/* Compute lanes.  A device has E264_MAX_LANES HIP queues for kernels; a stream is bound to one (default 0) and everything
 * that touches its DPB is ordered there, like the frames of one Edge264Decoder are ordered by its task queue
 * (src/edge264_internal.h:396-400).  Streams of different lanes are independent (src/edge264_internal.h:144-151: no shared
 * state between decoders), so their submissions overlap on the GPU: the tile-parallel kernels of one lane use the compute
 * units the wavefront kernels of another leave idle.  All streams of one batch must share a lane.  Rebinding waits for
 * what the stream still has queued on its old lane. */
#define E264_MAX_LANES 4
int  e264hip_stream_bind_lane(E264Stream *s, int lane);

/* Frame memory.  Replaces Edge264AllocCb / Edge264FreeCb (edge264.h:42-43; called from
 * alloc_frame, src/edge264_headers.c:113-133): the samples of DPB slot `slot` live in
 * HBM; *host_mirror (pinned, samples_bytes) is what Edge264Frame.samples will point
 * into (src/edge264.c:385-387) once e264hip_frame_download has run. */
int  e264hip_frame_alloc(E264Stream *s, int slot, size_t samples_bytes, void **host_mirror);
void e264hip_frame_free(E264Stream *s, int slot);
/* "non-existing" frames of frame_num gaps (src/edge264_headers.c:1122-1144) and tests */
int  e264hip_frame_fill(E264Stream *s, int slot, int value);
int  e264hip_frame_upload(E264Stream *s, int slot, const void *src, size_t bytes);

/* Submit one coded frame.  Replaces the sample-writing half of the per-slice worker
 * (worker_loop, src/edge264_headers.c:450-603: every decode_intra / decode_inter / add_idct /
 * deblock_mb call issued by parse_slice_data, src/edge264_slice.c:1651-1849).  `packet` is a
 * host buffer (ideally from e264hip_packet_buffer, pinned); the call enqueues the H2D copy
 * and the kernels on the device queue and returns without waiting. */
int  e264hip_frame_submit(E264Stream *s, const void *packet, size_t bytes);
/* Pinned staging memory for the emitters, recycled when the copy has been consumed. */
void *e264hip_packet_buffer(E264Stream *s, size_t max_bytes);

/* Readiness test of edge264_get_frame (next_deblock_addr[pic]==INT_MAX, src/edge264.c:373)
 * becomes "the kernels that write this slot have completed". */
int  e264hip_frame_wait(E264Stream *s, int slot);
/* Copy the finished frame to its host mirror (what get_frame hands out). */
int  e264hip_frame_download(E264Stream *s, int slot, void *dst, size_t bytes);
/* Decode-to-device use (the consumer of the pictures is another GPU stage -- scaler, encoder, inference): the
 * HBM address of slot `slot` in the reference's frame layout (Edge264Frame: Y plane, then rows of Cb | Cr, strides as in the
 * packet header), valid until e264hip_frame_free; call e264hip_frame_wait first.  NULL when the slot is not allocated.  Nothing
 * crosses PCIe.  The reference has no counterpart (its frames are host memory, edge264.h:45-62). */
void *e264hip_frame_device_ptr(E264Stream *s, int slot);
/* edge264_flush (src/edge264.c:261-270): wait for what THIS stream has submitted, keep allocations.  (frame_alloc,
 * frame_free, stream_close and frame_download likewise never wait for other decoders' work: freed memory is parked and
 * recycled by the device object instead of hipFree'd, which would drain every queue.) */
int  e264hip_stream_flush(E264Stream *s);

/* ---- batched, device-resident replay (multi-stream front end, benchmarking) ---- */
/* Uploads a packet once; the handle can then be replayed any number of times with the
 * command bytes already resident in HBM (bench.py's timed region). */
typedef struct E264Packet E264Packet;
int  e264hip_packet_upload(E264Device *dev, const void *packet, size_t bytes, E264Packet **out);
void e264hip_packet_free(E264Packet *p);
/* One launch over n (stream, packet) pairs: frame i of every stream.  All streams must be
 * distinct.  mode: E264_RUN_* bits. */
#define E264_RUN_RECON   1
#define E264_RUN_DEBLOCK 2
#define E264_RUN_ALL     3
int  e264hip_submit_batch(E264Device *dev, E264Stream *const *streams, E264Packet *const *packets, int n, int mode);
/* Host-only validation of a command packet (layout, per-macroblock offsets and indices): what every host-packet entry
 * point below runs before the packet may reach the device.  0 or EINVAL (e264hip_last_error() names the field). */
/* Build-time switches of the library, space separated ("" = the product build).  E264_ABL_* and E264_PHASE_* entries are timing
 * ablations whose samples are wrong by design: e264hip_device_open refuses such a build (ENOTSUP) unless E264_ALLOW_ABLATION=1 is in
 * the environment, and so do the loaders (edge264_amd/backend.py, the front end).  No counterpart in the reference (its variants are
 * all correct decoders, src/edge264.c:161-216). */
const char *e264hip_build_flags(void);
int  e264hip_packet_check(const void *packet, size_t bytes);
/* The WIRE form of a packet (version 5, include/edge264_compact.h: fewer bytes over PCIe for pictures full of skipped / plain 16x16 macroblocks -- what the
 * stream itself spends a run length on, src/edge264_slice.c:1651-1849).  Every host-packet entry point (e264hip_frame_submit, e264hip_submit_batch_host /
 * _pinned, e264hip_packet_upload, e264hip_packet_check) takes either form; a wire packet is unfolded on the device by one more kernel in front of the four
 * (into a buffer of the stream's) and means exactly what its expansion means.  C callers use the inline functions of edge264_compact.h (the front end does);
 * these three are the same functions for callers that do not compile C:
 *   e264hip_packet_compact_bound   capacity e264hip_packet_compact needs for this version-4 packet (0: not one)
 *   e264hip_packet_compact         version 4 (must pass e264hip_packet_check) -> version 5; returns the bytes written, 0 on error
 *   e264hip_packet_expand          version 5 -> the canonical version-4 packet; out == NULL: the size needed; 0 on error */
size_t e264hip_packet_compact_bound(const void *packet, size_t bytes);
size_t e264hip_packet_compact(const void *packet, size_t bytes, void *out, size_t cap);
size_t e264hip_packet_expand(const void *packet, size_t bytes, void *out, size_t cap);
/* Same for packets that still live in HOST memory (the finished frames of many decoders, src/edge264_headers.c:532-568,
 * one per stream): staged through each stream's pinned ring, copied and launched on the device queue without any
 * synchronisation; the host buffers may be reused on return.  This is what a multi-stream front end calls once per
 * round (edge264_amd/driver/e264_multi.cpp). */
int  e264hip_submit_batch_host(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode);
/* Same for packets assembled IN PLACE in page-locked memory (e264hip_host_alloc): no staging copy, the H2D transfer reads
 * the caller's buffer, which must stay untouched until the submission has retired (e264hip_event_record after the call, then
 * e264hip_event_query / e264hip_frame_wait / e264hip_device_sync).  This is the path of a front end whose
 * emitters write the finished frame straight into pinned memory (the reference's per-frame hand-over point,
 * src/edge264_headers.c:532-568).  flags: E264_SUBMIT_TRUSTED = these exact bytes have already passed
 * e264hip_packet_check (on the parser thread that produced them), the per-macroblock walk is not repeated here; the slots
 * the vetted header names (dst_slot, ref_slots -- packet_check verifies both against the records) are still checked
 * against the stream's allocations. */
#define E264_SUBMIT_TRUSTED 1
int  e264hip_submit_batch_pinned(E264Device *dev, E264Stream *const *streams, const void *const *packets, const size_t *bytes, int n, int mode, int flags);
void *e264hip_host_alloc(E264Device *dev, size_t bytes);
void e264hip_host_free(E264Device *dev, void *p);
/* Same, with the job table built once and kept in HBM: the launch itself moves no bytes
 * over PCIe (a persistent multi-stream front end re-submits frame i of every stream). */
typedef struct E264Batch E264Batch;
int  e264hip_batch_create(E264Device *dev, E264Stream *const *streams, E264Packet *const *packets, int n, E264Batch **out);
int  e264hip_batch_submit(E264Batch *b, int mode);
void e264hip_batch_free(E264Batch *b);

/* Timing on the queues the kernels run on (hipEvents; torch.cuda.Event would only see
 * torch's own stream).  Slots 0..15.  Recorded on lane 0 behind everything queued on the other lanes so far. */
int  e264hip_event_record(E264Device *dev, int idx);
int  e264hip_event_elapsed_ms(E264Device *dev, int idx_start, int idx_stop, float *ms);
/* 0 when everything queued before e264hip_event_record(dev, idx) has left the GPU, EBUSY while it has not; never blocks.
 * (How a front end that submits packets in place, e264hip_submit_batch_pinned, learns when their buffers may be reused.) */
int  e264hip_event_query(E264Device *dev, int idx);
/* Accumulated time of each of the four kernels of a submission (ms4[0] deblock-parameter kernel,
 * ms4[1] parallel MB kernel, ms4[2] intra wavefront, ms4[3] deblocking wavefront), measured with
 * events recorded on the queue between the launches (only when enabled). */
int  e264hip_kernel_timing(E264Device *dev, int enable);
int  e264hip_kernel_time_ms(E264Device *dev, double *ms4, int *launches);

/* Tunables; returns the previous value, -1 if unknown: "waves" / "intra_waves" (waves per frame workgroup of the two
 * wavefront kernels), "side_queue" (1 = the deblock-parameter kernel runs on a second HIP queue beside the parallel MB
 * kernel; its ms4[0] is then measured on that queue), "upload_queue" (default 1: the H2D copies of host batches run on
 * the device's upload queue beside the kernels of earlier batches; 0: on the lane itself, in order with them).
 * Timing ablations that skip work are separate builds of the library (csrc/Makefile `variant`, -DE264_ABL_*), never an
 * option of the product. */
int  e264hip_set_option(E264Device *dev, const char *name, int value);

#ifdef __cplusplus
}
#endif
#endif
