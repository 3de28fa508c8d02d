// Shared device/host helpers for the gfx950 kernels (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/promptir_hip.h"

#define PIR_WAVE 64
#define PIR_NUM_XCD 8
#define PIR_NUM_CU 256

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PIR_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return PIR_EINVAL; \
  } while (0)

static inline int pir_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PIR_OK : (int)e;
}

static inline long pir_cdiv(long a, long b) { return (a + b - 1) / b; }

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Remap the
// linear block id so that each XCD owns a CONTIGUOUS range of logical work items:
// neighbouring tiles (which share an operand panel) then hit the same L2.
// Bijective for any grid size (cdna_hip_programming.md §5, "XCD swizzle must be bijective").
__device__ __forceinline__ int pir_xcd_remap(int bid, int nwg) {
  const int q = nwg / PIR_NUM_XCD, r = nwg % PIR_NUM_XCD;
  const int xcd = bid % PIR_NUM_XCD, idx = bid / PIR_NUM_XCD;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float pir_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float pir_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` holds >= 16 floats of LDS.
__device__ __forceinline__ float pir_block_sum(float v, float* red) {
  v = pir_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
