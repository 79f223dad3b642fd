// tests/emu/emu_shims.h -- TEST INFRASTRUCTURE: what the kernels' source needs to compile for the host (qualifiers, address spaces,
// the VALU byte instructions the source names, LDS atomics of a sequentially emulated workgroup).  Shared by pred_emu.cpp and intra_emu.cpp.
#ifndef E264_EMU_SHIMS_H
#define E264_EMU_SHIMS_H
#include <stdint.h>
#include <string.h>
#define E264_HOST_INTRINSICS
#define E264_DEV static inline
#define E264_AS_GLOBAL
#define E264_AS_CONST
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int mul24(int a, int b) { return (int)((uint32_t)(a << 8 >> 8) * (uint32_t)(b << 8 >> 8)); } // v_mul_i32_i24: operands truncated to 24 bits signed, as the instruction does
static inline uint32_t v_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{ // v_perm_b32: selector byte 0..3 -> lo, 4..7 -> hi, 0x0c -> 0x00, >= 0x0d -> 0xff (8..11: sign replication, unused here)
	uint64_t src = (uint64_t)hi << 32 | lo;
	uint32_t out = 0;
	for (int i = 0; i < 4; i++) {
		uint32_t s = sel >> (8 * i) & 255, b;
		if (s < 8) b = (uint32_t)(src >> (8 * s)) & 255;
		else if (s == 0x0c) b = 0;
		else if (s >= 0x0d) b = 255;
		else b = ((src >> (16 * (s - 8) + 15)) & 1) ? 255 : 0;
		out |= b << (8 * i);
	}
	return out;
}
static inline uint32_t v_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((uint64_t)hi << 32 | lo) >> (8 * (sh & 3))); }
static inline uint32_t v_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((uint64_t)hi << 32 | lo) >> (sh & 31)); }
static inline uint32_t v_lerp_u8(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t out = 0;
	for (int i = 0; i < 4; i++)
		out |= (((a >> (8 * i) & 255) + (b >> (8 * i) & 255) + (c >> (8 * i) & 1)) >> 1) << (8 * i);
	return out;
}
static inline uint32_t v_sat_pk_u8_i16(uint32_t v)
{
	int a = (int16_t)(v & 0xffff), b = (int16_t)(v >> 16);
	a = a < 0 ? 0 : a > 255 ? 255 : a; b = b < 0 ? 0 : b > 255 ? 255 : b;
	return (uint32_t)a | (uint32_t)b << 8;
}
static inline int lds_add(int *p, int v) { int o = *p; *p += v; return o; }
static inline void lds_or(uint32_t *p, uint32_t v) { *p |= v; }
#endif
