// Shared device/host helpers for the gfx950 kernels (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/promptir_hip.h"

// Diagnostic builds of round 2 (per-phase clocks, one pipeline component removed per bit: results garbage) lived behind
// these macros; the product sources no longer contain them, and a build that defines one is refused.
#if defined(X3_ABLATE) || defined(NT_ABLATE) || defined(X3_TRACE) || defined(PIR_DIAG)
#error "diagnostic macros are not part of the product build (see tools/patches/ for the round-2 experiment diffs)"
#endif

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

// Sum of the S split-K partials of element e (partials `stride` floats apart) in EXACTLY the order of the stand-alone
// second stage (reduce_batch.hip: reduce_wide_body with 4 groups below 64 splits, 16 from there on: every group walks
// its splits with four accumulators, the groups are added in order): a consumer that reads the partials itself - the
// MDTA softmax kernels do, for the gram matrix and for dattn - gets the very bits the reduction launch would have written.
__device__ __forceinline__ float pir_split_sum(const float* __restrict__ p, long stride, int S, long e) {
  if (S < 64) {
    // four groups x four accumulators = sixteen independent chains: the loads of a round are all in flight together (walked
    // group by group the sum is a chain of S / 4 dependent load latencies - as long as the launch it replaces).  Every chain
    // sees its elements in the order of reduce_wide_body<4>: full rounds while k + 12 < S, the rest of a group into s0.
    float s[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int a = 0; a < 4; ++a) s[g][a] = 0.f;
    int kk[4] = {0, 1, 2, 3};
    for (int base = 0; base + 12 < S; base += 16) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = base + g;
        if (k + 12 < S) {
#pragma unroll
          for (int a = 0; a < 4; ++a) s[g][a] += p[(long)(k + 4 * a) * stride + e];
          kk[g] = k + 16;
        }
      }
    }
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      for (int k = kk[g]; k < S; k += 4) s[g][0] += p[(long)k * stride + e];
      tot += (s[g][0] + s[g][1]) + (s[g][2] + s[g][3]);
    }
    return tot;
  }
  const int GR = 16;
  float tot = 0.f;
  for (int grp = 0; grp < GR; ++grp) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = grp;
    for (; k + 3 * GR < S; k += 4 * GR) {
      s0 += p[(long)k * stride + e];
      s1 += p[(long)(k + GR) * stride + e];
      s2 += p[(long)(k + 2 * GR) * stride + e];
      s3 += p[(long)(k + 3 * GR) * stride + e];
    }
    for (; k < S; k += GR) s0 += p[(long)k * stride + e];
    tot += (s0 + s1) + (s2 + s3);
  }
  return tot;
}

// Division of a small index by a runtime divisor without the ~30-instruction integer-divide sequence:
// magic = floor(2^32 / d) + 1 (host side, pir_magic), exact for n * d < 2^32.
// d == 1 has no 32-bit magic: encoded as 0 and handled by the select.
static inline unsigned pir_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }
__device__ __forceinline__ int pir_fastdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

// gelu_erf(v) and its derivative from ONE erf evaluation (F.gelu default, net/model.py:97)
__device__ __forceinline__ void pir_gelu_both(float v, float& g, float& dg) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
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
