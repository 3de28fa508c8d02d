// Loss, plane copies, partial-sum reduction and the AdamW step (gfx950), all HBM-bound streams.
#include "pir_common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ grad, float gval, float* __restrict__ ws,
                                                         long count) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    s += fabsf(d);
    if (grad) grad[i] = d > 0.f ? gval : (d < 0.f ? -gval : 0.f);
  }
  const float t = pir_block_sum(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ ws, int n, float inv_count, float lscale,
                                                       float* __restrict__ loss) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += ws[i];
  const float t = pir_block_sum(s, red);
  // (t * inv_count) first: with lscale == 1 the value is the plain mean, bit for bit
  if (threadIdx.x == 0) loss[0] = (t * inv_count) * lscale;
}

// grad = sign(a-b) * (*dloss) / count : backward of the mean-absolute-error with the upstream
// gradient read from device memory (no host sync).
__global__ __launch_bounds__(256) void l1_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ dloss, float inv_count,
                                                      float* __restrict__ grad, long count) {
  const float gval = dloss[0] * inv_count;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    grad[i] = d > 0.f ? gval : (d < 0.f ? -gval : 0.f);
  }
}

// dst (contiguous NCHW) = src read through four free strides: the layout repair of the operator layer (a permuted or
// channels-last view handed to the module surface) as a library kernel instead of an ATen copy.
__global__ __launch_bounds__(256) void copy_strided4_kernel(const float* __restrict__ x, long s0, long s1, long s2, long s3,
                                                            float* __restrict__ y, int C, int H, int W, long total) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long w = e % W, t1 = e / W, h = t1 % H, t2 = t1 / H, c = t2 % C, b = t2 / C;
    y[e] = x[b * s0 + c * s1 + h * s2 + w * s3];
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void copy_planes_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ y,
                                                          long y_bs, int accumulate, int B, long n) {
  const long per = n / VEC, total = (long)B * per;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / per, i = (e % per) * VEC;
    if (VEC == 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + b * x_bs + i);
      f32x4* d = reinterpret_cast<f32x4*>(y + b * y_bs + i);
      if (accumulate) v += *d;
      *d = v;
    } else {
      float v = x[b * x_bs + i];
      if (accumulate) v += y[b * y_bs + i];
      y[b * y_bs + i] = v;
    }
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long count) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i];
}

// torch.optim.AdamW single-tensor update, same operation order:
//   p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = b2*v + (1-b2) g^2;
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long count,
                                                    float decay, float beta1, float beta2, float eps,
                                                    float step_size, float inv_bc2_sqrt, float grad_scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * decay;
    float mi = m[i];
    mi = mi + (1.f - beta1) * (gi - mi);
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// sigma-noise in the uint8 domain (reference utils/degradation_utils.py:21-27 / utils/dataset_utils.py:195-198:
// np.clip(clean + randn * sigma, 0, 255).astype(np.uint8), then ToTensor) with the repo's counter-based
// generator (promptir_amd/weights.py: splitmix64 -> uniform -> Box-Muller), so host and device agree.
__device__ __forceinline__ float u01_splitmix(unsigned long long key, unsigned long long idx1) {
  unsigned long long x = key + idx1 * 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  x = x ^ (x >> 31);
  return (float)(x >> 40) * 5.9604644775390625e-08f;  // 2^-24
}

__global__ __launch_bounds__(256) void degrade_gaussian_kernel(const float* __restrict__ clean, float* __restrict__ out,
                                                               const float* __restrict__ sigma,
                                                               const unsigned long long* __restrict__ keys, long per_image,
                                                               int B) {
  const long total = (long)B * per_image;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e / per_image);
    const unsigned long long i1 = (unsigned long long)(e % per_image) + 1ULL;
    const double u1 = (double)u01_splitmix(keys[2 * b], i1), u2 = (double)u01_splitmix(keys[2 * b + 1], i1);
    const float n = (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586476925 * u2));
    const double img255 = floor((double)clean[e] * 255.0);
    double v = img255 + (double)n * (double)sigma[b];
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    out[e] = (float)(unsigned char)v / 255.0f;
  }
}

// Training patches cut from whole decoded images ON THE DEVICE (reference: PromptTrainDataset.__getitem__,
// utils/dataset_utils.py:133-172 - random crop :102-111 / RandomCrop :32-35, random_augmentation utils/image_utils.py:177-182
// -> data_augmentation :133-160, Degradation.single_degrade utils/degradation_utils.py:21-27, ToTensor).  The host picks
// the crop window and the augmentation mode (its RNG, like the reference) and ships the uint8 HWC images as they were
// decoded; this kernel does the index work: out[c][i][j] of sample b is source pixel aug^-1(i, j) of the P x P window.
// np.rot90 / np.flipud on a square HWC patch m (P x P), out = data_augmentation(m, mode):
//   0: m[i][j]        1: m[P-1-i][j]      2: m[j][P-1-i]      3: m[j][i]
//   4: m[P-1-i][P-1-j] 5: m[i][P-1-j]     6: m[P-1-j][i]      7: m[P-1-j][P-1-i]
// meta[b] = {offset of the clean image, offset of the paired degraded image or -1, H, W, top, left, mode, unused}.
// Unpaired samples (de_id 0..2) get sigma[b] Gaussian noise in the uint8 domain: uint8(clip(k + n * sigma, 0, 255)).
__global__ __launch_bounds__(256) void crop_augment_u8_kernel(const unsigned char* __restrict__ images, const long* __restrict__ meta,
                                                              const float* __restrict__ sigma,
                                                              const unsigned long long* __restrict__ keys,
                                                              float* __restrict__ degraded, float* __restrict__ clean, int B, int P) {
  const long per = 3L * P * P, total = (long)B * per;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e / per);
    const long r = e - (long)b * per;
    const int c = (int)(r / ((long)P * P)), i = (int)((r / P) % P), j = (int)(r % P);
    const long* m = meta + 8L * b;
    const long W = m[3], top = m[4], left = m[5];
    const int mode = (int)m[6], q = P - 1;
    int si, sj;
    switch (mode) {
      case 1: si = q - i; sj = j; break;
      case 2: si = j; sj = q - i; break;
      case 3: si = j; sj = i; break;
      case 4: si = q - i; sj = q - j; break;
      case 5: si = i; sj = q - j; break;
      case 6: si = q - j; sj = i; break;
      case 7: si = q - j; sj = q - i; break;
      default: si = i; sj = j; break;
    }
    const long src = ((top + si) * W + (left + sj)) * 3 + c;
    const unsigned char k = images[m[0] + src];
    clean[e] = (float)k / 255.0f;
    if (m[1] >= 0) {
      degraded[e] = (float)images[m[1] + src] / 255.0f;
    } else {
      const unsigned long long i1 = (unsigned long long)r + 1ULL;
      const double u1 = (double)u01_splitmix(keys[2 * b], i1), u2 = (double)u01_splitmix(keys[2 * b + 1], i1);
      const float n = (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586476925 * u2));
      double v = (double)k + (double)n * (double)sigma[b];
      v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
      degraded[e] = (float)(unsigned char)v / 255.0f;
    }
  }
}

inline int grid_for(long total, int cap = 4096) { long g = pir_cdiv(total, 256); if (g < 1) g = 1; return (int)(g < cap ? g : cap); }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pir_abi_version(void) { return 8; }
extern "C" const char* pir_arch(void) { return "gfx950"; }

extern "C" int pir_l1_loss(const float* restored, const float* clean, float* loss, float* grad, float gscale, float lscale,
                           float* ws, long count, pir_stream_t stream) {
  PIR_CHECK_ARG(restored && clean && loss && ws && count > 0);
  hipStream_t s = (hipStream_t)stream;
  const int blocks = grid_for(count, 1024);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(blocks), dim3(256), 0, s, restored, clean, grad, gscale / (float)count, ws, count);
  int st = pir_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, s, ws, blocks, 1.f / (float)count, lscale, loss);
  return pir_launch_status();
}

extern "C" int pir_l1_loss_grad(const float* restored, const float* clean, const float* dloss, float scale, float* grad,
                                long count, pir_stream_t stream) {
  PIR_CHECK_ARG(restored && clean && dloss && grad && count > 0);
  hipLaunchKernelGGL(l1_grad_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream,
                     restored, clean, dloss, scale / (float)count, grad, count);
  return pir_launch_status();
}

extern "C" int pir_copy_strided4(const float* x, long s0, long s1, long s2, long s3, float* y, int B, int C, int H, int W,
                                 pir_stream_t stream) {
  PIR_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0);
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(copy_strided4_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, s0, s1, s2, s3, y, C, H, W, total);
  return pir_launch_status();
}

extern "C" int pir_degrade_gaussian(const float* clean, float* out, const float* sigma, const unsigned long long* keys,
                                    long per_image, int B, pir_stream_t stream) {
  PIR_CHECK_ARG(clean && out && sigma && keys && per_image > 0 && B > 0);
  hipLaunchKernelGGL(degrade_gaussian_kernel, dim3(grid_for((long)B * per_image)), dim3(256), 0, (hipStream_t)stream,
                     clean, out, sigma, keys, per_image, B);
  return pir_launch_status();
}

extern "C" int pir_crop_augment_u8(const unsigned char* images, const long* meta, const float* sigma, const unsigned long long* keys,
                                   float* degraded, float* clean, int B, int P, pir_stream_t stream) {
  PIR_CHECK_ARG(images && meta && sigma && keys && degraded && clean && B > 0 && P > 0);
  hipLaunchKernelGGL(crop_augment_u8_kernel, dim3(grid_for(3L * B * P * P)), dim3(256), 0, (hipStream_t)stream,
                     images, meta, sigma, keys, degraded, clean, B, P);
  return pir_launch_status();
}

extern "C" int pir_copy_planes(const float* x, long x_bs, float* y, long y_bs, int accumulate,
                               int B, long plane_floats, pir_stream_t stream) {
  PIR_CHECK_ARG(x && y && B > 0 && plane_floats > 0);
  const bool v4 = plane_floats % 4 == 0 && x_bs % 4 == 0 && y_bs % 4 == 0 && al16(x) && al16(y);
  hipStream_t s = (hipStream_t)stream;
  if (v4) hipLaunchKernelGGL((copy_planes_kernel<4>), dim3(grid_for((long)B * plane_floats / 4)), dim3(256), 0, s, x, x_bs, y, y_bs, accumulate, B, plane_floats);
  else hipLaunchKernelGGL((copy_planes_kernel<1>), dim3(grid_for((long)B * plane_floats)), dim3(256), 0, s, x, x_bs, y, y_bs, accumulate, B, plane_floats);
  return pir_launch_status();
}

extern "C" int pir_add(const float* a, const float* b, float* out, long count, pir_stream_t stream) {
  PIR_CHECK_ARG(a && b && out && count > 0);
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, a, b, out, count);
  return pir_launch_status();
}

extern "C" int pir_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count,
                              float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                              float grad_scale, pir_stream_t stream) {
  PIR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && count > 0 && step > 0);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const float decay = 1.f - lr * weight_decay;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(count, 8192)), dim3(256), 0, (hipStream_t)stream,
                     param, grad, exp_avg, exp_avg_sq, count, decay, beta1, beta2, eps, step_size, inv_bc2_sqrt, grad_scale);
  return pir_launch_status();
}

// Build self-description of the LOADED library, derived from compile-time state (tests/test_cabi.py compares it with
// what the sources say): bit 0 = a diagnostic / ablation macro was defined (pir_common.h refuses those, so a product
// build reports 0 here), bits 8..23 = the ABI version this object was compiled with, bit 24 = the device pass of this
// translation unit targeted gfx950 (read from a kernel-side constant the host copy of which is patched by the compiler's
// own target macro).
#if defined(X3_ABLATE) || defined(NT_ABLATE) || defined(X3_TRACE) || defined(PIR_DIAG)
#define PIR_FLAG_DIAG 1
#else
#define PIR_FLAG_DIAG 0
#endif
#ifndef PIR_BUILD_ARCH_GFX950   /* the Makefile passes -DPIR_BUILD_ARCH_GFX950=1 together with --offload-arch=gfx950 */
#define PIR_BUILD_ARCH_GFX950 0
#endif
extern "C" int pir_build_flags(void) {
  return PIR_FLAG_DIAG | (8 << 8) | (PIR_BUILD_ARCH_GFX950 ? (1 << 24) : 0);
}
