// PromptGenBlock (net/model.py:218-235) stages other than its 3x3 conv and spatial mean:
// prompt-mixture weights, mixture + bilinear resize, and their adjoints.  All tiny / HBM-bound.
#include "pir_common.h"

extern "C" int pir_reduce_partials(const float* parts, long stride, int S, float alpha, int accumulate,
                                   float* out, long count, pir_stream_t stream);

namespace {

// PyTorch upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0,
// scale = in/out; taps i0 = floor(src), i1 = i0 + (i0 < in-1), weights (1-l, l).
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap bilinear_tap(int dst, float scale, int in_size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Tap t;
  t.i0 = (int)src;
  if (t.i0 > in_size - 1) t.i0 = in_size - 1;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

// mix[b][l] = softmax_l( emb[b] . Wl[l] + bl[l] );  one wave per image
__global__ __launch_bounds__(64) void prompt_mix_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ Wl,
                                                           const float* __restrict__ bl, float* __restrict__ mix,
                                                           int C, int L) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float logit[8];
  float mx = -INFINITY;
  for (int l = 0; l < L; ++l) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += emb[(long)b * C + c] * Wl[(long)l * C + c];
    s = pir_wave_sum(s) + bl[l];
    logit[l] = s;
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
  for (int l = 0; l < L; ++l) { logit[l] = expf(logit[l] - mx); den += logit[l]; }
  if (lane == 0)
    for (int l = 0; l < L; ++l) mix[(long)b * L + l] = logit[l] / den;
}

__global__ __launch_bounds__(256) void prompt_resize_fwd_kernel(const float* __restrict__ mix, const float* __restrict__ P,
                                                                float* __restrict__ out, long out_bs,
                                                                int B, int L, int D, int S, int H, int W) {
  const long total = (long)B * D * H * W;
  const float sh = (float)S / (float)H, sw = (float)S / (float)W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % W);
    long t = e / W;
    const int y = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const int b = (int)(t / D);
    const Tap ty = bilinear_tap(y, sh, S), tx = bilinear_tap(x, sw, S);
    float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
    for (int l = 0; l < L; ++l) {
      const float wl = mix[(long)b * L + l];
      const float* pl = P + ((long)l * D + d) * S * S;
      v00 += wl * pl[ty.i0 * S + tx.i0];
      v01 += wl * pl[ty.i0 * S + tx.i1];
      v10 += wl * pl[ty.i1 * S + tx.i0];
      v11 += wl * pl[ty.i1 * S + tx.i1];
    }
    out[b * out_bs + ((long)d * H + y) * W + x] = ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
  }
}

// dQ[b][d][s][t] = sum_{y,x} wy(y->s) wx(x->t) dout[b][d][y][x]   (gather form of the adjoint; deterministic)
__global__ __launch_bounds__(256) void prompt_resize_adjoint_kernel(const float* __restrict__ dout, long dout_bs,
                                                                    float* __restrict__ dQ, int B, int D, int S, int H, int W) {
  const long total = (long)B * D * S * S;
  const float sh = (float)S / (float)H, sw = (float)S / (float)W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e % S);
    long r = e / S;
    const int s = (int)(r % S); r /= S;
    const int d = (int)(r % D);
    const int b = (int)(r / D);
    int ylo = (int)floorf(((float)s - 0.5f) / sh - 0.5f) - 1, yhi = (int)ceilf(((float)s + 1.5f) / sh - 0.5f) + 1;
    int xlo = (int)floorf(((float)t - 0.5f) / sw - 0.5f) - 1, xhi = (int)ceilf(((float)t + 1.5f) / sw - 0.5f) + 1;
    if (ylo < 0) ylo = 0; if (yhi > H - 1) yhi = H - 1;
    if (xlo < 0) xlo = 0; if (xhi > W - 1) xhi = W - 1;
    const float* src = dout + b * dout_bs + (long)d * H * W;
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
      const Tap ty = bilinear_tap(y, sh, S);
      const float wy = (ty.i0 == s ? ty.l0 : 0.f) + (ty.i1 == s ? ty.l1 : 0.f);
      if (wy == 0.f) continue;
      float rowacc = 0.f;
      for (int x = xlo; x <= xhi; ++x) {
        const Tap tx = bilinear_tap(x, sw, S);
        const float wx = (tx.i0 == t ? tx.l0 : 0.f) + (tx.i1 == t ? tx.l1 : 0.f);
        if (wx != 0.f) rowacc += wx * src[(long)y * W + x];
      }
      acc += wy * rowacc;
    }
    dQ[e] = acc;
  }
}

// dP[l][d][s][t] = sum_b mix[b][l] * dQ[b][d][s][t]
__global__ __launch_bounds__(256) void prompt_dparam_kernel(const float* __restrict__ mix, const float* __restrict__ dQ,
                                                            float* __restrict__ dP, int B, int L, long DSS) {
  const long total = (long)L * DSS;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int l = (int)(e / DSS);
    const long r = e % DSS;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += mix[(long)b * L + l] * dQ[(long)b * DSS + r];
    dP[e] = acc;
  }
}

// dmix[b][l] = < dQ[b], P[l] >: a [B x DSS] . [L x DSS]^T product with DSS up to 262144 and only B * L <= 256 results.
// One workgroup per (image, chunk of DSS) reads its dQ chunk once and all L prompt chunks, and leaves L partial sums in
// part[chunk][b * L + l]; pir_reduce_partials sums the chunks (fixed order).  (One workgroup per (b, l) walking all of DSS
// alone took 218 us per call at 0.58 TB/s: 160 workgroups, 1024 dependent iterations each.)
constexpr int DMIX_CHUNKS = 32, DMIX_LMAX = 8;
__global__ __launch_bounds__(256) void prompt_dmix_kernel(const float* __restrict__ dQ, const float* __restrict__ P,
                                                          float* __restrict__ part, int B, int L, long DSS) {
  __shared__ float red[16];
  const int b = blockIdx.x, ch = blockIdx.y;
  const long per = (DSS + DMIX_CHUNKS - 1) / DMIX_CHUNKS;
  const long lo = ch * per, hi = lo + per < DSS ? lo + per : DSS;
  const float* q = dQ + (long)b * DSS;
  float s[DMIX_LMAX];
#pragma unroll
  for (int l = 0; l < DMIX_LMAX; ++l) s[l] = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float qv = q[i];
#pragma unroll
    for (int l = 0; l < DMIX_LMAX; ++l)
      if (l < L) s[l] += qv * P[(long)l * DSS + i];
  }
#pragma unroll
  for (int l = 0; l < DMIX_LMAX; ++l) {
    if (l < L) {
      const float t = pir_block_sum(s[l], red);
      if (threadIdx.x == 0) part[(long)ch * B * L + (long)b * L + l] = t;
      __syncthreads();
    }
  }
}

__device__ __forceinline__ void softmax_bwd_small(const float* dmix_b, const float* mix_b, int L, float* dlogit) {
  float dot = 0.f;
  for (int l = 0; l < L; ++l) dot += dmix_b[l] * mix_b[l];
  for (int l = 0; l < L; ++l) dlogit[l] = mix_b[l] * (dmix_b[l] - dot);
}

// dWl[l][c] = sum_b dlogit[b][l] emb[b][c];  dbl[l] = sum_b dlogit[b][l]
__global__ __launch_bounds__(256) void prompt_dlinear_kernel(const float* __restrict__ dmix, const float* __restrict__ mix,
                                                             const float* __restrict__ emb, float* __restrict__ dWl,
                                                             float* __restrict__ dbl, int B, int C, int L) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= L * (C + 1)) return;
  const int l = e / (C + 1), c = e % (C + 1);
  float acc = 0.f;
  float dlogit[8];
  for (int b = 0; b < B; ++b) {
    softmax_bwd_small(dmix + (long)b * L, mix + (long)b * L, L, dlogit);
    acc += dlogit[l] * (c < C ? emb[(long)b * C + c] : 1.f);
  }
  if (c < C) dWl[(long)l * C + c] = acc; else dbl[l] = acc;
}

// dx[b][c][:] (+)= demb[b][c] / HW,  demb = dlogit[b] . Wl[:, c]
__global__ __launch_bounds__(256) void prompt_demb_kernel(const float* __restrict__ dmix, const float* __restrict__ mix,
                                                          const float* __restrict__ Wl, float* __restrict__ dx, long dx_bs,
                                                          int C, int L, int HW, int accumulate) {
  const int b = blockIdx.x / C, c = blockIdx.x % C;
  float dlogit[8];
  softmax_bwd_small(dmix + (long)b * L, mix + (long)b * L, L, dlogit);
  float demb = 0.f;
  for (int l = 0; l < L; ++l) demb += dlogit[l] * Wl[(long)l * C + c];
  const float v = demb / (float)HW;
  float* plane = dx + b * dx_bs + (long)c * HW;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) plane[i] = accumulate ? plane[i] + v : v;
}

inline int grid_for(long total) { long g = pir_cdiv(total, 256); return (int)(g < 4096 ? g : 4096); }

}  // namespace

extern "C" int pir_prompt_mix_fwd(const float* emb, const float* Wl, const float* bl, float* mix,
                                  int B, int C, int L, pir_stream_t stream) {
  PIR_CHECK_ARG(emb && Wl && bl && mix && B > 0 && C > 0 && L > 0 && L <= 8);
  hipLaunchKernelGGL(prompt_mix_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, emb, Wl, bl, mix, C, L);
  return pir_launch_status();
}

extern "C" int pir_prompt_resize_fwd(const float* mix, const float* P, float* out, long out_bs,
                                     int B, int L, int D, int S, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(mix && P && out && B > 0 && L > 0 && D > 0 && S > 0 && H > 0 && W > 0);
  const long total = (long)B * D * H * W;
  hipLaunchKernelGGL(prompt_resize_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     mix, P, out, out_bs, B, L, D, S, H, W);
  return pir_launch_status();
}

extern "C" size_t pir_prompt_resize_bwd_ws_floats(int B, int L, int D, int S, int H, int W) {
  (void)H; (void)W;
  if (B <= 0 || D <= 0 || S <= 0 || L <= 0) return 0;
  return (size_t)B * D * S * S + (size_t)DMIX_CHUNKS * B * L;   // the resized gradient + the dmix partial sums
}

extern "C" int pir_prompt_resize_bwd(const float* dout, long dout_bs, const float* mix, const float* P,
                                     float* dP, float* dmix, float* ws, size_t ws_floats,
                                     int B, int L, int D, int S, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(dout && mix && P && dP && dmix && ws && B > 0 && L > 0 && D > 0 && S > 0 && H > 0 && W > 0);
  const long DSS = (long)D * S * S;
  PIR_CHECK_ARG(L <= DMIX_LMAX);
  if ((size_t)B * DSS + (size_t)DMIX_CHUNKS * B * L > ws_floats) return PIR_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(prompt_resize_adjoint_kernel, dim3(grid_for(B * DSS)), dim3(256), 0, s, dout, dout_bs, ws, B, D, S, H, W);
  int st = pir_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(prompt_dparam_kernel, dim3(grid_for(L * DSS)), dim3(256), 0, s, mix, ws, dP, B, L, DSS);
  st = pir_launch_status();
  if (st) return st;
  float* part = ws + (long)B * DSS;
  hipLaunchKernelGGL(prompt_dmix_kernel, dim3(B, DMIX_CHUNKS), dim3(256), 0, s, ws, P, part, B, L, DSS);
  st = pir_launch_status();
  if (st) return st;
  return pir_reduce_partials(part, (long)B * L, DMIX_CHUNKS, 1.f, 0, dmix, (long)B * L, stream);
}

extern "C" int pir_prompt_mix_bwd(const float* dmix, const float* mix, const float* emb, const float* Wl,
                                  float* dWl, float* dbl, float* dx, long dx_bs, int accumulate,
                                  int B, int C, int L, int HW, pir_stream_t stream) {
  PIR_CHECK_ARG(dmix && mix && emb && Wl && dWl && dbl && dx && B > 0 && C > 0 && L > 0 && L <= 8 && HW > 0);
  hipStream_t s = (hipStream_t)stream;
  const int n = L * (C + 1);
  hipLaunchKernelGGL(prompt_dlinear_kernel, dim3((unsigned)pir_cdiv(n, 256)), dim3(256), 0, s, dmix, mix, emb, dWl, dbl, B, C, L);
  int st = pir_launch_status();
  if (st) return st;
  const int threads = HW >= 1024 ? 256 : 64;
  hipLaunchKernelGGL(prompt_demb_kernel, dim3((unsigned)(B * C)), dim3(threads), 0, s, dmix, mix, Wl, dx, dx_bs, C, L, HW, accumulate);
  return pir_launch_status();
}
