// Probes of the raw-buffer range check on gfx950:
//  (1) it includes the scalar offset (soffset); (2) a dwordx4 load that straddles the end of the
//  descriptor is checked per dword (in-range dwords are returned, the rest read 0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, float* out, int records, int soff) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, records, 0x00020000);
  unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4, soff, 0);
  out[threadIdx.x] = __builtin_bit_cast(float, v);
}
__global__ void probe4(const float* src, float* out, int records, int soff, int shift) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, records, 0x00020000);
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16 + shift, soff, 0));   // shift 4: dword-aligned only
  out[threadIdx.x * 4 + 0] = v.x;
  out[threadIdx.x * 4 + 1] = v.y;
  out[threadIdx.x * 4 + 2] = v.z;
  out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
  float h[256], *d, *o, ho[256];
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const int cases[][2] = {{256, 0}, {256, 128}, {256, 240}, {256, 256}, {256, 512}, {128, 64}, {128, 120}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, c[0], c[1]);
    hipMemcpy(ho, o, 64 * 4, hipMemcpyDeviceToHost);
    printf("b32  records=%d soffset=%d:", c[0], c[1]);
    for (int i = 0; i < 64; i += 7) printf(" [%d]=%g", i, ho[i]);
    printf("\n");
  }
  const int cases4[][3] = {{1024, 0, 0}, {1024, 0, 4}, {100, 0, 4}, {108, 0, 4}, {104, 0, 0}, {200, 64, 4}};
  for (auto& c : cases4) {
    hipLaunchKernelGGL(probe4, dim3(1), dim3(16), 0, 0, d, o, c[0], c[1], c[2]);
    hipMemcpy(ho, o, 64 * 4, hipMemcpyDeviceToHost);
    printf("b128 records=%d soffset=%d shift=%d:", c[0], c[1], c[2]);
    for (int i = 0; i < 36; ++i) printf(" %g", ho[i]);
    printf("\n");
  }
  return 0;
}
