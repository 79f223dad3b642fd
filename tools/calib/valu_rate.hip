// tools/calib/valu_rate.hip -- issue rate of the integer / byte VALU instructions the kernels are made of (gfx950).
// Each kernel runs N iterations of 32 independent copies of one instruction per wave, 4 waves per SIMD on every CU;
// prints cycles per wave-instruction per SIMD (at the clock measured with s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
#define REP8(x) x x x x x x x x
#define KERNEL(name, ASM) \
__global__ __launch_bounds__(256) void name(uint32_t *out, unsigned long long *cyc) { \
	uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
	uint32_t b = threadIdx.x * 0x01010101u + blockIdx.x, c = 0x00030002u; \
	asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[10:11], 0x3333" ::: "vcc", "s10", "s11"); \
	unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
	for (int i = 0; i < ITER; i++) { \
		REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c)); \
		     asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) \
	} \
	unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
	out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
	if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0); \
}
#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %4\n"
#define A_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %4\n"
#define A_PKMAD(n) "v_pk_mad_u16 %" #n ", %" #n ", %5, %4\n"
#define A_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %5\n"
#define A_PKASHR(n) "v_pk_ashrrev_i16 %" #n ", %5, %" #n "\n"
#define A_PKMAX(n) "v_pk_max_i16 %" #n ", %" #n ", %4\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %4, %5\n"
#define A_ALIGNB(n) "v_alignbyte_b32 %" #n ", %" #n ", %4, %5\n"
#define A_LERP(n) "v_lerp_u8 %" #n ", %" #n ", %4, %5\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %4, vcc\n"
#define A_CNDMASK64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %4, s[10:11]\n"
#define A_MAX32(n) "v_max_i32 %" #n ", %" #n ", %4\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %4\n"
#define A_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %5\n"
#define A_ADDU16(n) "v_add_u16 %" #n ", %" #n ", %4\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %4\n"
#define A_MAD24(n) "v_mad_i32_i24 %" #n ", %" #n ", %5, %4\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %5\n"
#define A_FMA(n) "v_fma_f32 %" #n ", %" #n ", %5, %4\n"
#define A_MED3(n) "v_med3_i32 %" #n ", %" #n ", %4, %5\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %4, %5\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %4\n"
#define A_SDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define A_DOT4(n) "v_dot4_i32_i8 %" #n ", %4, %5, %" #n "\n"
#define A_SAD(n) "v_sad_u8 %" #n ", %4, %5, %" #n "\n"
#define A_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
// round 5: which of v_cndmask_b32's forms is the slow one (0.105 G/s above), and what a compare + select pair or a bit-field insert costs instead
#define A_CND64V(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %4, vcc\n"
#define A_CMPCND(n) "v_cmp_lt_u32_e32 vcc, %" #n ", %5\n v_cndmask_b32_e32 %" #n ", %" #n ", %4, vcc\n"
#define A_CMPCND64(n) "v_cmp_lt_u32_e64 s[10:11], %" #n ", %5\n v_cndmask_b32_e64 %" #n ", %" #n ", %4, s[10:11]\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %5, %" #n ", %4\n"
KERNEL(k_add, A_ADD) KERNEL(k_pkadd, A_PKADD) KERNEL(k_pkmad, A_PKMAD) KERNEL(k_pkmul, A_PKMUL) KERNEL(k_pkashr, A_PKASHR)
KERNEL(k_pkmax, A_PKMAX) KERNEL(k_perm, A_PERM) KERNEL(k_alignb, A_ALIGNB) KERNEL(k_lerp, A_LERP) KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_mad24, A_MAD24) KERNEL(k_mullo, A_MULLO) KERNEL(k_fma, A_FMA) KERNEL(k_med3, A_MED3) KERNEL(k_add3, A_ADD3)
KERNEL(k_cnd64, A_CNDMASK64) KERNEL(k_max32, A_MAX32) KERNEL(k_lshr, A_LSHR) KERNEL(k_and, A_AND) KERNEL(k_mul24, A_MUL24) KERNEL(k_addu16, A_ADDU16) KERNEL(k_mov, A_MOV)
KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_sdwa, A_SDWA) KERNEL(k_dot4, A_DOT4) KERNEL(k_sad, A_SAD) KERNEL(k_bfe, A_BFE)
KERNEL(k_cnd64v, A_CND64V) KERNEL(k_cmpcnd, A_CMPCND) KERNEL(k_cmpcnd64, A_CMPCND64) KERNEL(k_bfi, A_BFI)
typedef void (*kern_t)(uint32_t *, unsigned long long *);
int main()
{
	struct { const char *name; kern_t k; } ks[] = {{"v_add_u32", k_add}, {"v_pk_add_u16", k_pkadd}, {"v_pk_mad_u16", k_pkmad}, {"v_pk_mul_lo_u16", k_pkmul},
		{"v_pk_ashrrev_i16", k_pkashr}, {"v_pk_max_i16", k_pkmax}, {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_alignb}, {"v_lerp_u8", k_lerp},
		{"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32_e64 sgpr", k_cnd64}, {"v_max_i32", k_max32}, {"v_lshrrev_b32", k_lshr}, {"v_and_b32", k_and}, {"v_mul_u32_u24", k_mul24}, {"v_add_u16", k_addu16}, {"v_mov_b32", k_mov}, {"v_mad_i32_i24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_fma_f32", k_fma}, {"v_med3_i32", k_med3},
		{"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd}, {"v_add_u32_sdwa", k_sdwa}, {"v_dot4_i32_i8", k_dot4}, {"v_sad_u8", k_sad}, {"v_bfe_u32", k_bfe},
		{"v_cndmask_b32_e64 vcc", k_cnd64v}, {"v_cmp_e32 + v_cndmask_e32 (vcc), PAIRS", k_cmpcnd}, {"v_cmp_e64 + v_cndmask_e64 (sgpr), PAIRS", k_cmpcnd64}, {"v_bfi_b32", k_bfi}};
	const int blocks = 256 * 4; // 4 workgroups of 4 waves per CU = 4 waves per SIMD
	uint32_t *out; unsigned long long *cyc;
	hipMalloc((void **)&out, blocks * 256 * 4); hipMalloc((void **)&cyc, 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (auto &k : ks) {
		for (int rep = 0; rep < 2; rep++) {
			hipMemset(cyc, 0, 8);
			hipEventRecord(e0);
			hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, cyc);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
			if (rep == 1) {
				const double n_inst = (double)ITER * 64; // wave-instructions per wave
				const double per_wave_cycles = (double)c / blocks; // s_memtime ticks per wave (wave 0 of each block)
				// 4 waves share a SIMD: cycles per wave-instruction per SIMD = per-wave cycles / (n_inst * 4)
				printf("%-20s %8.3f ms   memtime ticks/wave %.0f   ticks per wave-instr per SIMD %.3f   Ginstr/s/SIMD %.3f\n", k.name, ms, per_wave_cycles,
				       per_wave_cycles / (n_inst * 4), n_inst * 4 / (ms * 1e6));
			}
		}
	}
	return 0;
}
