// tools/calib/load_rate.hip -- how fast the vector memory path serves lane-private 13-row windows (gfx950).
// Every lane fetches 13 rows x 16 bytes of a "frame" (1920-byte rows) at a position derived from its id and an offset
// pattern; variants: alignment of the 16 bytes (16 / 4 / 1 byte), lane -> position mapping (neighbouring lanes 8 pixels
// apart = spatial order; scattered over a 256 x 128 tile = class-sorted order), width (dwordx4 vs 4 x dword).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(4))) v4u4;
typedef v4u __attribute__((aligned(1))) v4u1;
#define STRIDE 1920
template <int MODE, int SCATTER>
__global__ __launch_bounds__(512) void k(const uint8_t *frames, uint32_t *out, int frame_bytes)
{
	const int tid = threadIdx.x, tile = blockIdx.x;
	const uint8_t *f = frames + (size_t)(tile % 256) * frame_bytes; // 256 "streams"
	const int t2 = (tile / 256) % 56; // tile of the frame: 7 x 8 tiles of 256 x 128
	const int tx = (t2 % 7) * 256, ty = (t2 / 7) * 128 + 16;
	// quadrant of the tile: 32 x 16 quadrants
	int qi = tid;
	if (SCATTER) qi = (tid * 197 + 13) & 511; // pseudo-random permutation of the 512 quadrants
	const int qx = qi & 31, qy = qi >> 5;
	uint32_t h = (uint32_t)(tile * 512 + qi) * 2654435761u;
	const int mvx = (int)(h >> 8 & 31) - 16, mvy = (int)(h >> 16 & 31) - 16;
	int X = tx + qx * 8 + mvx + 16, Y = ty + qy * 8 + mvy;
	if (MODE == 0) X &= ~15;      // 16-byte aligned
	else if (MODE == 1 || MODE == 3) X &= ~3; // dword aligned
	const uint8_t *p = f + (size_t)Y * STRIDE + X;
	uint32_t acc = 0;
#pragma unroll
	for (int r = 0; r < 13; r++) {
		if (MODE == 3) {
			const uint32_t *q = (const uint32_t *)(p + r * STRIDE);
			acc ^= q[0] ^ q[1] ^ q[2] ^ q[3];
		} else if (MODE == 2) {
			const v4u v = *(const v4u1 *)(p + r * STRIDE);
			acc ^= v.x ^ v.y ^ v.z ^ v.w;
		} else {
			const v4u v = *(const v4u4 *)(p + r * STRIDE);
			acc ^= v.x ^ v.y ^ v.z ^ v.w;
		}
	}
	out[(size_t)tile * 512 + tid] = acc;
}
int main()
{
	const int frame_bytes = STRIDE * 1088, n_tiles = 256 * 56;
	uint8_t *frames; uint32_t *out;
	hipMalloc((void **)&frames, (size_t)256 * frame_bytes + 65536); hipMemset(frames, 1, (size_t)256 * frame_bytes + 65536);
	hipMalloc((void **)&out, (size_t)n_tiles * 512 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct { const char *name; void (*fn)(const uint8_t *, uint32_t *, int); } ks[] = {
		{"dwordx4 16B-aligned, spatial", k<0, 0>}, {"dwordx4 16B-aligned, scattered", k<0, 1>},
		{"dwordx4 4B-aligned, spatial", k<1, 0>}, {"dwordx4 4B-aligned, scattered", k<1, 1>},
		{"dwordx4 1B-aligned, spatial", k<2, 0>}, {"dwordx4 1B-aligned, scattered", k<2, 1>},
		{"4 x dword 4B-aligned, spatial", k<3, 0>}, {"4 x dword 4B-aligned, scattered", k<3, 1>}};
	for (auto &kk : ks)
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(kk.fn, dim3(n_tiles), dim3(512), 0, 0, frames, out, frame_bytes);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (rep) printf("%-36s %7.3f ms for %d tiles x 512 windows  (%.1f G lane-rows/s)\n", kk.name, ms, n_tiles, (double)n_tiles * 512 * 13 / ms / 1e6);
		}
	return 0;
}
