/* tests/emu: stand-in for <hip/hip_runtime.h> when the kernels' source is compiled for the HOST (test infrastructure) */
#ifndef E264_EMU_HIP_RUNTIME_H
#define E264_EMU_HIP_RUNTIME_H
#include <stdint.h>
#include <stddef.h>
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
#define __constant__ static const
#endif
