// PMC calibration for FETCH_SIZE / WRITE_SIZE on gfx950 in OUR access patterns (MI355X_MICROARCH.md "HBM":
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every kernel moves exactly BYTES (printed) so that counter / BYTES gives the correction factor.
//   k_write_full : 16 B per lane, fully coalesced (1 KiB per wave instruction)
//   k_write_mbrow: macroblock pattern of e264_mbpar_kernel: a wave writes a 16x16 luma block as 64 x 4 B,
//                  (16 rows of 16 B at stride 1920), 8 horizontally adjacent blocks one after the other
//   k_write_row16: deblock pattern: 16 lanes each write one 16-B row of a block
//   k_read_full  : 16 B per lane coalesced read
//   k_read_window: 21 rows x 21(+3) B window per 16x16 block (motion compensation halo), blocks adjacent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int W = 1920, H = 1088, NF = 256;   // 256 luma planes of 1080p = 535 MB (> 256 MiB Infinity Cache)
constexpr size_t PLANE = (size_t)W * H;

__global__ void k_write_full(uint4 *dst, size_t n16)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i < n16; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = make_uint4((unsigned)i, 1, 2, 3);
}
__global__ void k_read_full(const uint4 *src, size_t n16, unsigned *sink)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned acc = 0;
	for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read_dword(const uint32_t *src, size_t n4, unsigned *sink)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned acc = 0;
	for (; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
	if (acc == 0x12345678u) *sink = acc;
}
// grid (H/16 * W/128, NF), block 64: one wave = strip of 8 blocks
__global__ void k_write_mbrow(uint8_t *dst)
{
	const int lane = threadIdx.x, strip = blockIdx.x, f = blockIdx.y;
	const int sy = strip / (W / 128), sx = strip % (W / 128);
	const int k = lane >> 2, r = lane & 3;
	const int X = ((k & 1) + ((k >> 2) & 1) * 2) * 4, Y = (((k >> 1) & 1) + (k >> 3) * 2) * 4 + r;
	uint8_t *p = dst + (size_t)f * PLANE + (size_t)(sy * 16 + Y) * W + sx * 128 + X;
	for (int i = 0; i < 8; i++) {
		*(uint32_t *)(p + i * 16) = lane * 0x01010101u + i;
		__builtin_amdgcn_s_sleep(20);
	}
}
// grid as above; 16 lanes active per block row pattern: 4 blocks per wave instruction
__global__ void k_write_row16(uint8_t *dst)
{
	const int lane = threadIdx.x, strip = blockIdx.x, f = blockIdx.y;
	const int sy = strip / (W / 128), sx = strip % (W / 128);
	uint8_t *p = dst + (size_t)f * PLANE + (size_t)(sy * 16 + (lane & 15)) * W + sx * 128 + (lane >> 4) * 16;
	for (int i = 0; i < 2; i++) {
		*(uint4 *)(p + i * 64) = make_uint4(lane, i, 2, 3);
		__builtin_amdgcn_s_sleep(20);
	}
}
__global__ void k_read_window(const uint8_t *src, unsigned *sink)
{
	const int lane = threadIdx.x, strip = blockIdx.x, f = blockIdx.y;
	const int sy = strip / (W / 128), sx = strip % (W / 128);
	unsigned acc = 0;
	for (int i = 0; i < 8; i++) {
		int x0 = sx * 128 + i * 16 - 2, y0 = sy * 16 - 2;
		// 21 rows x 6 dwords (24 B) = 126 lanes -> two rounds
		for (int t = lane; t < 126; t += 64) {
			int row = t / 6, c = t % 6;
			int y = min(max(y0 + row, 0), H - 1), x = min(max(x0 + c * 4, 0), W - 4);
			acc += *(const uint32_t *)(src + (size_t)f * PLANE + (size_t)y * W + (x & ~3));
		}
	}
	if (acc == 0x12345678u) *sink = acc;
}

int main()
{
	uint8_t *buf; unsigned *sink;
	const size_t bytes = PLANE * NF;
	CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 4));
	CHK(hipMemset(buf, 1, bytes));
	CHK(hipDeviceSynchronize());
	dim3 g(H / 16 * (W / 128), NF);
	for (int rep = 0; rep < 2; rep++) {
		hipLaunchKernelGGL(k_write_full, dim3(8192), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
		hipLaunchKernelGGL(k_read_full, dim3(8192), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
		hipLaunchKernelGGL(k_read_dword, dim3(8192), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, sink);
		hipLaunchKernelGGL(k_write_mbrow, g, dim3(64), 0, 0, buf);
		hipLaunchKernelGGL(k_read_window, g, dim3(64), 0, 0, buf, sink);
		hipLaunchKernelGGL(k_write_row16, g, dim3(64), 0, 0, buf);
	}
	CHK(hipDeviceSynchronize());
	printf("bytes_per_kernel %zu  (k_read_window useful bytes: %zu = 21x21 per block)\n", bytes, (size_t)NF * (H / 16) * (W / 16) * 441);
	return 0;
}
