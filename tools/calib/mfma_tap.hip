// tools/calib/mfma_tap.hip -- is the matrix pipe worth using for H.264's six-tap sums?  (gfx950)
//
// The luma half-sample filter of /root/reference/src/edge264_inter.c:416-968 is a banded Toeplitz product: 16 output columns of a
// row are (1, -5, 20, 20, -5, 1) slid over 21 samples.  e264_pred_kernel does it in packed 16-bit VALU arithmetic (e264_pred.h brow /
// htaps8); this program prices the same 256 sums (a 16 x 16 block) on v_mfma_i32_16x16x64_i8 and v_mfma_i32_16x16x32_i8 INCLUDING what the
// kernel would have to do around the instruction: operands out of LDS, samples made signed (x ^ 0x80; the taps sum to 32, so 4096 comes
// back through the accumulator together with the rounding 16: exact in int32), the int32 sums shifted, clipped and packed to bytes in
// the layout the kernel's tile sink wants (a dword = four samples of one row), and written to LDS.
//
//   C[m][n] = sum_k A[m][k] B[k][n]     m = output column, n = picture row, k = window column
//   A[m][k] = tap[k - m]  (a constant of the kernel, in registers)      B[k][n] = sample(row n, column k) - 128
//   operand layout (checked against a scalar product on the host before anything is timed; the program says which hypothesis held):
//     x64: lane l holds 16 bytes: row / column l % 16, k = 16 (l / 16) + j      x32: 8 bytes, k = 8 (l / 16) + j
//     C:   lane l holds n = l % 16, m = 4 (l / 16) + j: four consecutive output columns of one picture row = one dword of samples
//
// Every variant runs on 4 waves per SIMD of every CU (as e264_pred_kernel does) and is timed with s_memtime around ITER blocks per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../edge264_amd/csrc/e264_pred.h"

typedef int v4i __attribute__((ext_vector_type(4)));
#define ITER 512
#define WBUF 2048 // bytes of window samples per wave in LDS (ITER blocks walk through it at 16-byte steps)

// ---- (a) the product's way: one lane = 8 rows x 8 outputs, packed 16-bit (brow of e264_pred.h): 64 lanes x 64 outputs = 16 blocks of 256 ----
__global__ __launch_bounds__(256) void k_valu(const uint8_t *src, uint32_t *out, unsigned long long *cyc)
{
	__shared__ __attribute__((aligned(16))) uint8_t win[4][WBUF + 2048];
	__shared__ uint32_t tile[4][64 * 16];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int i = lane; i < (WBUF + 2048) / 4; i += 64) ((uint32_t *)win[w])[i] = ((const uint32_t *)src)[i];
	__syncthreads();
	uint32_t acc = 0;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int it = 0; it < ITER / 16; it++) { // one pass = 64 lanes x 8 rows x 8 outputs = 16 blocks of 256 sums
		const uint8_t *p = win[w] + ((it * 16) & (WBUF - 1)) + lane * 16;
#pragma unroll
		for (int r = 0; r < 8; r++) {
			const v4u v = *(const v4u *)(p + r * 64);
			Row4 a = {v.x, v.y, v.z, v.w};
			uint32_t b[2];
			brow(a, b);
			tile[w][(r * 64 + lane) * 2 & 1023] = b[0];
			tile[w][((r * 64 + lane) * 2 + 1) & 1023] = b[1];
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	__syncthreads();
	acc = tile[w][lane] ^ tile[w][lane + 64];
	out[blockIdx.x * 256 + threadIdx.x] = acc;
	if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

// the constant operand: A[m][k] = tap[k - m]; lane l holds row m = l % 16, bytes k0 .. k0 + n - 1
__device__ __forceinline__ int tap_at(int d) { return d == 0 || d == 5 ? 1 : d == 1 || d == 4 ? -5 : d == 2 || d == 3 ? 20 : 0; }
__device__ __forceinline__ uint32_t tap_dword(int m, int k)
{
	uint32_t v = 0;
	for (int j = 0; j < 4; j++) v |= (uint32_t)(uint8_t)tap_at(k + j - m) << (8 * j);
	return v;
}
// four int32 sums (+ 4096 + 16 already inside) -> (x >> 5) clipped to 0..255, packed to one dword
__device__ __forceinline__ uint32_t pack_sums(v4i c)
{
	// two saturating packs to int16, packed shift, packed unsigned saturation to bytes: 2 + 2 + 3 instructions
	const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(c.x, c.y)), hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(c.z, c.w));
	const s16x2 s5 = {5, 5};
	return packus4(as_s2(lo) >> s5, as_s2(hi) >> s5);
}

// ---- (b) v_mfma_i32_16x16x64_i8: one instruction per block; lanes 0..31 carry the 21 (32) window bytes of 16 rows, lanes 32..63 zeros ----
template <bool CHECK>
__global__ __launch_bounds__(256) void k_mfma64(const uint8_t *src, uint32_t *out, unsigned long long *cyc)
{
	__shared__ __attribute__((aligned(16))) uint8_t win[4][WBUF + 2048];
	__shared__ uint32_t tile[4][64 * 16];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int i = lane; i < (WBUF + 2048) / 4; i += 64) ((uint32_t *)win[w])[i] = ((const uint32_t *)src)[i];
	__syncthreads();
	const int m = lane & 15, kg = lane >> 4;
	v4i A;
	A.x = (int)tap_dword(m, kg * 16); A.y = (int)tap_dword(m, kg * 16 + 4); A.z = (int)tap_dword(m, kg * 16 + 8); A.w = (int)tap_dword(m, kg * 16 + 12);
	const v4i cinit = {4096 + 16, 4096 + 16, 4096 + 16, 4096 + 16};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int it = 0; it < (CHECK ? 1 : ITER); it++) {
		// row n = lane % 16 of the block: 32 bytes at row stride 32 in this buffer (the kernel: the lane's own 16-byte global load)
		const uint8_t *p = win[w] + ((it * 16) & (WBUF - 1)) + m * 32 + (kg & 1) * 16;
		v4i B = {0, 0, 0, 0}; // k >= 32 (lanes 32..63): the taps are zero there, nothing is loaded
		if (kg < 2) {
			const v4u v = *(const v4u *)p;
			B.x = (int)(v.x ^ 0x80808080u); B.y = (int)(v.y ^ 0x80808080u); B.z = (int)(v.z ^ 0x80808080u); B.w = (int)(v.w ^ 0x80808080u);
		}
		const v4i c = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, cinit, 0, 0, 0);
		const uint32_t px = pack_sums(c);
		if (CHECK) out[(blockIdx.x * 4 + w) * 256 + m * 4 + kg] = px; // row n = m (lane % 16), dword kg of the row
		else tile[w][(it * 64 + m * 4 + kg) & 1023] = px;
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (!CHECK) {
		__syncthreads();
		out[blockIdx.x * 256 + threadIdx.x] = tile[w][lane] ^ tile[w][lane + 64];
		if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
	}
}

// ---- (c) v_mfma_i32_16x16x32_i8: K = 32 holds the 21-byte window exactly; every lane carries 8 bytes ----
template <bool CHECK>
__global__ __launch_bounds__(256) void k_mfma32(const uint8_t *src, uint32_t *out, unsigned long long *cyc)
{
	__shared__ __attribute__((aligned(16))) uint8_t win[4][WBUF + 2048];
	__shared__ uint32_t tile[4][64 * 16];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int i = lane; i < (WBUF + 2048) / 4; i += 64) ((uint32_t *)win[w])[i] = ((const uint32_t *)src)[i];
	__syncthreads();
	const int m = lane & 15, kg = lane >> 4;
	const long A = (long)((unsigned long long)tap_dword(m, kg * 8) | (unsigned long long)tap_dword(m, kg * 8 + 4) << 32);
	const v4i cinit = {4096 + 16, 4096 + 16, 4096 + 16, 4096 + 16};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int it = 0; it < (CHECK ? 1 : ITER); it++) {
		const uint8_t *p = win[w] + ((it * 16) & (WBUF - 1)) + m * 32 + kg * 8;
		const v2u v = *(const v2u *)p;
		const long B = (long)((unsigned long long)(v.x ^ 0x80808080u) | (unsigned long long)(v.y ^ 0x80808080u) << 32);
		const v4i c = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, B, cinit, 0, 0, 0);
		const uint32_t px = pack_sums(c);
		if (CHECK) out[(blockIdx.x * 4 + w) * 256 + m * 4 + kg] = px;
		else tile[w][(it * 64 + m * 4 + kg) & 1023] = px;
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (!CHECK) {
		__syncthreads();
		out[blockIdx.x * 256 + threadIdx.x] = tile[w][lane] ^ tile[w][lane + 64];
		if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
	}
}

// ---- (d) the instruction alone, back to back on independent accumulators: the pipe's own rate ----
__global__ __launch_bounds__(256) void k_mfma64_bare(const uint8_t *src, uint32_t *out, unsigned long long *cyc)
{
	const int lane = threadIdx.x & 63;
	v4i A = {(int)src[lane], (int)src[lane + 1], 3, 4}, B = {5, 6, (int)src[lane + 2], 8};
	v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int it = 0; it < ITER / 4; it++) {
		c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c0, 0, 0, 0);
		c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c1, 0, 0, 0);
		c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c2, 0, 0, 0);
		c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, c3, 0, 0, 0);
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(c0.x ^ c1.y ^ c2.z ^ c3.w);
	if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

static int clip255h(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

int main()
{
	const int blocks = 256 * 4; // 4 workgroups of 4 waves per CU = 4 waves per SIMD
	const size_t nsrc = WBUF + 2048;
	uint8_t *hsrc = (uint8_t *)malloc(nsrc);
	srand(7);
	for (size_t i = 0; i < nsrc; i++) hsrc[i] = (uint8_t)(rand() >> 7);
	for (int i = 0; i < 64; i++) hsrc[i] = i & 1 ? 255 : 0; // extremes: sums below 0 and above 255 * 32
	uint8_t *src; uint32_t *out; unsigned long long *cyc;
	hipMalloc((void **)&src, nsrc); hipMalloc((void **)&out, (size_t)blocks * 256 * 4); hipMalloc((void **)&cyc, 8);
	hipMemcpy(src, hsrc, nsrc, hipMemcpyHostToDevice);
	// ---- layout / exactness check: block 0 of the buffer (it = 0): rows n = 0..15 at stride 32, output (n, m) = clip((sum + 16) >> 5) ----
	uint32_t expect[64];
	for (int n = 0; n < 16; n++)
		for (int d = 0; d < 4; d++) {
			uint32_t wv = 0;
			for (int j = 0; j < 4; j++) {
				const int mm = d * 4 + j;
				const uint8_t *s = hsrc + n * 32 + mm;
				const int sum = s[0] - 5 * s[1] + 20 * s[2] + 20 * s[3] - 5 * s[4] + s[5];
				wv |= (uint32_t)clip255h((sum + 16) >> 5) << (8 * j);
			}
			expect[n * 4 + d] = wv;
		}
	uint32_t got[64];
	int ok64 = 0, ok32 = 0;
	hipMemset(out, 0, 1024);
	hipLaunchKernelGGL(k_mfma64<true>, dim3(1), dim3(256), 0, 0, src, out, cyc);
	hipMemcpy(got, out, sizeof(got), hipMemcpyDeviceToHost);
	ok64 = memcmp(got, expect, sizeof(got)) == 0;
	hipMemset(out, 0, 1024);
	hipLaunchKernelGGL(k_mfma32<true>, dim3(1), dim3(256), 0, 0, src, out, cyc);
	hipMemcpy(got, out, sizeof(got), hipMemcpyDeviceToHost);
	ok32 = memcmp(got, expect, sizeof(got)) == 0;
	printf("exactness against the scalar six-tap filter on a 16 x 16 block (layout k = 16 (l / 16) + j resp. 8 (l / 16) + j): x64 %s, x32 %s\n",
	       ok64 ? "IDENTICAL" : "DIFFERENT", ok32 ? "IDENTICAL" : "DIFFERENT");
	struct { const char *name; void (*k)(const uint8_t *, uint32_t *, unsigned long long *); double blocks_per_wave; } ks[] = {
		{"packed 16-bit VALU (e264_pred.h brow), 8 x 8 per lane", k_valu, ITER},
		{"v_mfma_i32_16x16x64_i8 + marshalling + pack", k_mfma64<false>, ITER},
		{"v_mfma_i32_16x16x32_i8 + marshalling + pack", k_mfma32<false>, ITER},
		{"v_mfma_i32_16x16x64_i8 alone (4 accumulators)", k_mfma64_bare, ITER}};
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	double base = 0;
	for (auto &k : ks)
		for (int rep = 0; rep < 2; rep++) {
			hipMemset(cyc, 0, 8);
			hipEventRecord(e0);
			hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, src, out, cyc);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
			if (rep == 1) {
				const double per_wave = (double)c / blocks;                 // ticks of one wave for its ITER blocks, 4 waves sharing the SIMD
				const double per_block_simd = per_wave / (k.blocks_per_wave * 4); // SIMD cycles per block of 256 sums
				if (base == 0) base = per_block_simd;
				printf("%-56s %7.3f ms   SIMD cycles per 256 sums %7.1f   x%.2f against the VALU form\n", k.name, ms, per_block_simd, base / per_block_simd);
			}
		}
	return 0;
}
