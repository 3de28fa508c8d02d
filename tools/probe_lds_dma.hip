// Probe: buffer_load ... lds (LDS-DMA) with 16 bytes per lane on gfx950: where does lane L's data land, and what
// do out-of-range lanes write?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void probe(const float* src, float* out, int records) {
  __shared__ __attribute__((aligned(16))) float sm[512];
  for (int i = threadIdx.x; i < 512; i += 64) sm[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, records, 0x00020000);
  // lane L loads 16 bytes from src + L*16 (permuted: lane L reads chunk (L*7)%64 to see the lane->LDS map)
  const int chunk = (threadIdx.x * 7) % 64;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sm + 64), 16, chunk * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = sm[i];
}
int main() {
  float h[512], *d, *o, ho[512];
  for (int i = 0; i < 512; ++i) h[i] = (float)i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int records : {1024, 512}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, records);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    printf("records=%d: sm[60..] =", records);
    for (int i = 60; i < 64 + 4 * 6; ++i) printf(" %g", ho[i]);
    printf(" ... lane63 area:");
    for (int i = 64 + 63 * 4; i < 64 + 64 * 4 + 4; ++i) printf(" %g", ho[i]);
    printf("\n   lane L's first float for L=0..15:");
    for (int L = 0; L < 16; ++L) printf(" %g", ho[64 + L * 4]);
    printf("\n");
  }
  return 0;
}
