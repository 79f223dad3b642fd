// tools/calib/stage_rate.hip -- what staging a tile's reference region in LDS costs on gfx950, against the lane-private
// window fetch of tools/calib/load_rate.hip (same frames, same tiles, same windows).
//
// A workgroup of 512 threads = one tile of 256 x 128 luma samples of one of 256 "frames" (1920-byte rows).  Variants:
//   stage_reg    region (320 x 168 bytes around the tile) global -> registers -> LDS: coalesced dwordx4 loads, ds_write_b128
//   stage_dma    the same through LDS-DMA (global_load_lds_dwordx4: no register round trip)
//   + windows    every lane then reads its 13-row x 16-byte window out of the region (4 dword LDS reads per row, per-lane
//                position: what the prediction items would do), folded into one word
//   windows_mem  the round-2 scheme for comparison: the same windows straight from memory (13 lane-private dwordx4 loads)
// EXTRA_LDS pads the workgroup's LDS so that one (1) or two (0) workgroups fit a CU, like the kernel variants would.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(4))) v4u4;
#define STRIDE 1920
#define RW 320            // region row bytes (tile 256 + 32 left + 32 right)
#define RH 168            // region rows (tile 128 + 20 above + 20 below)
#define RPIECES (RW / 16 * RH)  // 16-byte pieces: 3360

__device__ __forceinline__ void window_pos(int tile, int tid, int &wx, int &wy)
{ // position of the lane's 13 x 16 window inside the region: its 8x8 block + a vector of +-16 samples, dword aligned
	const int qi = (tid * 197 + 13) & 511; // class-sorted order scatters the lanes over the tile
	const int qx = qi & 31, qy = qi >> 5;
	const uint32_t h = (uint32_t)(tile * 512 + qi) * 2654435761u;
	const int mvx = (int)(h >> 8 & 31) - 16, mvy = (int)(h >> 16 & 31) - 16;
	wx = (32 + qx * 8 + mvx - 2) & ~3;
	wy = 20 + qy * 8 + mvy - 2;
}

template <int MODE, int WINDOWS, int EXTRA_LDS>
__global__ __launch_bounds__(512) void k(const uint8_t *frames, uint32_t *out, int frame_bytes)
{
	__shared__ __attribute__((aligned(16))) uint8_t reg[RH * RW + EXTRA_LDS];
	const int tid = threadIdx.x, tile = blockIdx.x;
	const uint8_t *f = frames + (size_t)(tile % 256) * frame_bytes;
	const int t2 = (tile / 256) % 42; // tile of the frame: 6 x 7 tiles with room for the margins
	const int tx = 32 + (t2 % 6) * 256, ty = 32 + (t2 / 6) * 128;
	const uint8_t *src = f + (size_t)(ty - 20) * STRIDE + (tx - 32);
	uint32_t acc = 0;
	if (MODE == 0) { // registers
		v4u v[7];
#pragma unroll
		for (int i = 0; i < 7; i++) {
			const int p = i * 512 + tid;
			if (p < RPIECES) v[i] = *(const v4u *)(src + (size_t)(p / 20) * STRIDE + (p % 20) * 16);
		}
#pragma unroll
		for (int i = 0; i < 7; i++) {
			const int p = i * 512 + tid;
			if (p < RPIECES) *(v4u *)(reg + p * 16) = v[i];
		}
	} else if (MODE == 1) { // LDS-DMA: a wave's 64 pieces land at consecutive 16-byte slots from the (uniform) LDS address
		const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
		for (int i = 0; i < 7; i++) {
			const int p0 = (i * 8 + wave) * 64, p = p0 + lane;
			if (p0 < RPIECES) {
				const int pc = p < RPIECES ? p : RPIECES - 1;
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)(pc / 20) * STRIDE + (pc % 20) * 16),
					(__attribute__((address_space(3))) void *)(reg + p0 * 16), 16, 0, 0);
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	if (MODE != 2) __syncthreads();
	if (WINDOWS) {
		int wx, wy;
		window_pos(tile, tid, wx, wy);
		if (MODE == 2) { // straight from memory
			const uint8_t *p = src + (size_t)wy * STRIDE + wx;
#pragma unroll
			for (int r = 0; r < 13; r++) {
				const v4u v = *(const v4u4 *)(p + (size_t)r * STRIDE);
				acc ^= v.x ^ v.y ^ v.z ^ v.w;
			}
		} else {
			const uint32_t *p = (const uint32_t *)(reg + wy * RW + wx);
#pragma unroll
			for (int r = 0; r < 13; r++)
				acc ^= p[r * (RW / 4)] ^ p[r * (RW / 4) + 1] ^ p[r * (RW / 4) + 2] ^ p[r * (RW / 4) + 3];
		}
	} else {
		acc = reg[(tid * 37) % (RH * RW)];
	}
	out[(size_t)tile * 512 + tid] = acc;
}

int main()
{
	const int frame_bytes = STRIDE * 1088, n_tiles = 256 * 72;
	uint8_t *frames; uint32_t *out;
	hipMalloc((void **)&frames, (size_t)256 * frame_bytes + 65536); hipMemset(frames, 1, (size_t)256 * frame_bytes + 65536);
	hipMalloc((void **)&out, (size_t)n_tiles * 512 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct { const char *name; void (*fn)(const uint8_t *, uint32_t *, int); } ks[] = {
		{"stage via registers, 2 WG/CU", k<0, 0, 20000>}, {"stage via registers + windows, 2 WG/CU", k<0, 1, 20000>},
		{"stage via LDS-DMA, 2 WG/CU", k<1, 0, 20000>}, {"stage via LDS-DMA + windows, 2 WG/CU", k<1, 1, 20000>},
		{"stage via registers + windows, 1 WG/CU", k<0, 1, 100000>}, {"stage via LDS-DMA + windows, 1 WG/CU", k<1, 1, 100000>},
		{"windows from memory (round 2), 2 WG/CU", k<2, 1, 20000>}};
	for (auto &kk : ks)
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(kk.fn, dim3(n_tiles), dim3(512), 0, 0, frames, out, frame_bytes);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (rep) printf("%-44s %7.3f ms for %d tiles (%.0f GB/s of region bytes, %.2f us per tile and CU)\n", kk.name, ms, n_tiles,
				(double)n_tiles * RH * RW / ms / 1e6, ms * 1e3 * 256 / n_tiles);
		}
	return 0;
}
