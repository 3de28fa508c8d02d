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

// Division of a small index by a runtime divisor without the ~30-instruction integer-divide sequence:
// magic = floor(2^32 / d) + 1 (host side, pir_magic), exact for n * d < 2^32.
// d == 1 has no 32-bit magic: encoded as 0 and handled by the select.
static inline unsigned pir_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }
__device__ __forceinline__ int pir_fastdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

// Standard normal CDF Phi(v) = 0.5 * erfc(-v / sqrt(2)) for the erf form of GELU (F.gelu default, net/model.py:97).
// Branch-free erfc: t = 1 / (1 + |x| / 2), erfc(|x|) = t * exp(-x^2 + P(t)) with the degree-9 Chebyshev fit of
// Numerical Recipes (`erfcc`, fractional error < 1.2e-7 everywhere, i.e. fp32-class and relative in the tail) -
// one reciprocal, one exponential and ten FMAs instead of libm erff's two-branch evaluation (measured: the fused
// GDFN backward spends ~18 % of its time in erff).
__device__ __forceinline__ float pir_norm_cdf(float v) {
  const float x = v * 0.70710678118654752440f;
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.f + 0.5f * ax);
  float p = 0.17087277f;
  p = fmaf(p, t, -0.82215223f);
  p = fmaf(p, t, 1.48851587f);
  p = fmaf(p, t, -1.13520398f);
  p = fmaf(p, t, 0.27886807f);
  p = fmaf(p, t, -0.18628806f);
  p = fmaf(p, t, 0.09678418f);
  p = fmaf(p, t, 0.37409196f);
  p = fmaf(p, t, 1.00002368f);
  p = fmaf(p, t, -1.26551223f);
  const float half_erfc = 0.5f * t * __expf(fmaf(-ax, ax, p));   // 0.5 * erfc(|x|)
  return x >= 0.f ? 1.f - half_erfc : half_erfc;
}

// gelu_erf(v) and its derivative from one CDF evaluation
__device__ __forceinline__ void pir_gelu_both(float v, float& g, float& dg) {
  const float cdf = pir_norm_cdf(v);
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
  g = v * cdf;
  dg = cdf + v * pdf;
}

// Buffer descriptor over [base, base + bytes): the pointer halves go through readfirstlane so that the
// descriptor is provably wave-uniform (otherwise every buffer op is wrapped in a waterfall loop).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pir_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
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

// Reduce NV per-thread values over the workgroup with ONE barrier: wave64 shuffle reduction of each
// value, one LDS row per wave, then threads 0..NV-1 add the rows in fixed order (deterministic).
// `red` holds >= 4*NV floats.  Returns the total in threads 0..NV-1 (value index = threadIdx.x).
template <int NV>
__device__ __forceinline__ float pir_block_sum_many(const float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const float s = pir_wave_sum(v[t]);
    if (lane == 0) red[wid * NV + t] = s;
  }
  __syncthreads();
  float tot = 0.f;
  if (threadIdx.x < NV)
    for (int w = 0; w < nw; ++w) tot += red[w * NV + threadIdx.x];
  return tot;
}
