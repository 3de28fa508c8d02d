// Building blocks of the register-only ("wave-autonomous") stencil kernels: a group of adjacent lanes owns VEC
// columns each of one image row, slides down the rows with its 3-row windows in registers, and takes the one-column
// halos from the neighbouring lanes through DPP wave shifts.  No LDS, no barriers.  (gdfn_bwd.hip, stencil_wave.hip)
#pragma once
#include "pir_common.h"

__device__ __forceinline__ float dpp_from_lower(float v) {   // value held by lane - 1 (0 for lane 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_upper(float v) {   // value held by lane + 1 (0 for lane 63)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

template <int VEC>
__device__ __forceinline__ void row_load(const float* __restrict__ p, bool ok, float (&out)[VEC]) {
  if (VEC == 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f32x4*>(p);
    out[0] = v[0]; out[1 % VEC] = v[1]; out[2 % VEC] = v[2]; out[VEC - 1] = v[3];
  } else if (VEC == 2) {
    float2 v = {0.f, 0.f};
    if (ok) v = *reinterpret_cast<const float2*>(p);
    out[0] = v.x; out[VEC - 1] = v.y;
  } else {
    out[0] = ok ? p[0] : 0.f;
  }
}
template <int VEC>
__device__ __forceinline__ void row_store(float* __restrict__ p, const float (&v)[VEC]) {
  if (VEC == 4) { f32x4 t = {v[0], v[1 % VEC], v[2 % VEC], v[VEC - 1]}; *reinterpret_cast<f32x4*>(p) = t; }
  else if (VEC == 2) { float2 t = {v[0], v[VEC - 1]}; *reinterpret_cast<float2*>(p) = t; }
  else p[0] = v[0];
}


// [left halo, own columns, right halo] of a row held VEC columns per lane; lanes at the edges of their unit (or
// outside the image) get the convolution's zero padding.
template <int VEC>
__device__ __forceinline__ void pir_widen(const float (&raw)[VEC], float (&wide)[VEC + 2], bool has_left, bool has_right) {
  const float l = dpp_from_lower(raw[VEC - 1]), r = dpp_from_upper(raw[0]);
#pragma unroll
  for (int j = 0; j < VEC; ++j) wide[j + 1] = raw[j];
  wide[0] = has_left ? l : 0.f;
  wide[VEC + 1] = has_right ? r : 0.f;
}

// sum over the 2^shift lanes of a unit (aligned lane groups); every lane of the unit receives the total
__device__ __forceinline__ float pir_unit_sum(float v, int lanes) {
  for (int off = lanes >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
