// Small-matrix stages of MDTA (net/model.py:127-131): the c x c channel-attention map per
// (image, head).  c <= 176 in PromptIR, so one workgroup owns one map; rows go to waves,
// columns to lanes, reductions are wave64 shuffles.  These are launch-latency-sized kernels;
// the heavy lifting (q k^T over pixels, attn @ v) is in gemm.hip.
#include "pir_common.h"

namespace {

constexpr float NORM_EPS = 1e-12f;  // F.normalize default eps

// squared norm of one q / k row = sum of the partial sums the depthwise kernel left per row band
// (pir_dwconv3x3_sumsq: [image][part][2C]); fixed order, so deterministic
__device__ __forceinline__ float sum_parts(const float* __restrict__ p, int nparts, int stride) {
  float s = p[0];
  for (int k = 1; k < nparts; ++k) s += p[(long)k * stride];
  return s;
}

// attn = softmax_j( gram[i][j] / (nq_i nk_j) * temperature ),  n* = max(sqrt(sumsq), eps)
// 16 waves per workgroup, ONE row per wave: the kernel is bound by the latency of its dependent global loads
// (norm partials -> row -> exp -> store), so the rows of a map are spread over gridDim.y workgroups instead of being
// walked six deep by one (10.7 -> ~5 us per launch at c = 96; B * heads is only 16 workgroups at the 128^2 level);
// a row (c <= 256 columns) is loaded once into registers, the column norms once per wave.
// PARTS: `gram` still holds the split-K partial sums of q k^T (`splits` slices, `pstride` floats apart): the kernel sums
// them in the reduction's own order (pir_split_sum), writes the gram matrix the backward needs to `gram_out` and goes on -
// the reduction launch between the product and the softmax is gone.
template <bool PARTS>
__global__ __launch_bounds__(1024) void mdta_softmax_fwd_kernel(const float* __restrict__ gram,
                                                                const float* __restrict__ sumsq,
                                                                const float* __restrict__ temperature,
                                                                float* __restrict__ attn, int heads, int c, int nparts,
                                                                float* __restrict__ gram_out, long pstride, int splits) {
  constexpr int MAXJ = 4;
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, C = heads * c;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float* G = gram + (long)bh * c * c;
  float* A = attn + (long)bh * c * c;
  const float* sq = sumsq + (long)b * nparts * 2 * C + h * c;
  const float* sk = sq + C;
  const float t = temperature[h];
  float inv_k[MAXJ];
#pragma unroll
  for (int u = 0; u < MAXJ; ++u) {
    const int j = lane + 64 * u;
    inv_k[u] = j < c ? 1.f / fmaxf(sqrtf(sum_parts(sk + j, nparts, 2 * C)), NORM_EPS) : 0.f;
  }
  for (int i = blockIdx.y * nw + wid; i < c; i += nw * gridDim.y) {
    const float inv_q = 1.f / fmaxf(sqrtf(sum_parts(sq + i, nparts, 2 * C)), NORM_EPS);
    float sv[MAXJ];
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < MAXJ; ++u) {
      const int j = lane + 64 * u;
      float gij = 0.f;
      if (j < c) {
        if constexpr (PARTS) {
          gij = pir_split_sum(gram, pstride, splits, (long)bh * c * c + i * c + j);
          gram_out[(long)bh * c * c + i * c + j] = gij;
        } else {
          gij = G[i * c + j];
        }
      }
      sv[u] = j < c ? gij * inv_q * inv_k[u] * t : -INFINITY;
      m = fmaxf(m, sv[u]);
    }
    m = pir_wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < MAXJ; ++u) {
      sv[u] = lane + 64 * u < c ? expf(sv[u] - m) : 0.f;
      sum += sv[u];
    }
    sum = pir_wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int u = 0; u < MAXJ; ++u) {
      const int j = lane + 64 * u;
      if (j < c) A[i * c + j] = sv[u] * inv;
    }
  }
}

// Backward through softmax, temperature and both L2 normalisations (see include/promptir_hip.h).
// MAXJ: 64-column groups per lane (c <= 64 * MAXJ); ROWS: rows of a wave whose loads are in flight together.
// PARTS: `dattn` holds split-K partial sums (see the forward kernel); dattn itself is never written.
template <int MAXJ, int ROWS, bool PARTS = false>
__global__ __launch_bounds__(1024) void mdta_softmax_bwd_kernel(
    const float* __restrict__ dattn, const float* __restrict__ attn, const float* __restrict__ gram,
    const float* __restrict__ sumsq, const float* __restrict__ temperature, float* __restrict__ dgram,
    float* __restrict__ alpha_q, float* __restrict__ alpha_k, float* __restrict__ dtemp_partial, int heads, int c,
    int nparts, long pstride, int splits) {
  __shared__ float colred[16][64 * MAXJ];
  __shared__ float red[16];
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, C = heads * c;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float* G = gram + (long)bh * c * c;
  const float* A = attn + (long)bh * c * c;
  const float* dA = dattn + (long)bh * c * c;
  float* dG = dgram + (long)bh * c * c;
  const float* sq = sumsq + (long)b * nparts * 2 * C + h * c;
  const float* sk = sq + C;
  const float t = temperature[h];

  float inv_k[MAXJ], col[MAXJ];
#pragma unroll
  for (int u = 0; u < MAXJ; ++u) {
    const int j = lane + 64 * u;
    inv_k[u] = j < c ? 1.f / fmaxf(sqrtf(sum_parts(sk + j, nparts, 2 * C)), NORM_EPS) : 0.f;
    col[u] = 0.f;
  }
  float dt_acc = 0.f;
  // ROWS rows per trip: the loads of all of them (norm partials, attn, dattn, gram) are in flight before the first
  // reduction - the walk over a wave's rows is a chain of dependent load latencies otherwise
  for (int i0 = wid; i0 < c; i0 += ROWS * nw) {
    float sqi[ROWS], av[ROWS][MAXJ], dav[ROWS][MAXJ], gv[ROWS][MAXJ], dot[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int i = i0 + r * nw, ic = i < c ? i : i0;
      sqi[r] = sum_parts(sq + ic, nparts, 2 * C);
      dot[r] = 0.f;
#pragma unroll
      for (int u = 0; u < MAXJ; ++u) {
        const int j = lane + 64 * u;
        av[r][u] = j < c ? A[ic * c + j] : 0.f;
        if constexpr (PARTS) dav[r][u] = j < c ? pir_split_sum(dattn, pstride, splits, (long)bh * c * c + ic * c + j) : 0.f;
        else dav[r][u] = j < c ? dA[ic * c + j] : 0.f;
        gv[r][u] = j < c ? G[ic * c + j] : 0.f;
        dot[r] += dav[r][u] * av[r][u];
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int i = i0 + r * nw;
      if (i >= c) break;
      const float nq = fmaxf(sqrtf(sqi[r]), NORM_EPS);
      const float inv_q = 1.f / nq;
      const float d = pir_wave_sum(dot[r]);
      float rowterm = 0.f;
#pragma unroll
      for (int u = 0; u < MAXJ; ++u) {
        const int j = lane + 64 * u;
        if (j < c) {
          const float dS = av[r][u] * (dav[r][u] - d);
          const float R = gv[r][u] * inv_q * inv_k[u];
          dt_acc += dS * R;
          const float dR = t * dS;
          dG[i * c + j] = dR * inv_q * inv_k[u];
          rowterm += dR * R;
          col[u] += dR * R;
        }
      }
      rowterm = pir_wave_sum(rowterm);
      if (lane == 0) alpha_q[(long)b * C + h * c + i] = sqrtf(sqi[r]) > NORM_EPS ? -rowterm * inv_q * inv_q : 0.f;
    }
  }
#pragma unroll
  for (int u = 0; u < MAXJ; ++u) colred[wid][lane + 64 * u] = col[u];
  __syncthreads();
  for (int j = threadIdx.x; j < c; j += blockDim.x) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += colred[w][j];
    const float skj = sum_parts(sk + j, nparts, 2 * C);
    const float nk = fmaxf(sqrtf(skj), NORM_EPS);
    alpha_k[(long)b * C + h * c + j] = sqrtf(skj) > NORM_EPS ? -s / (nk * nk) : 0.f;
  }
  const float dt = pir_block_sum(dt_acc, red);
  if (threadIdx.x == 0) dtemp_partial[bh] = dt;
}

}  // namespace

extern "C" int pir_mdta_softmax_fwd(const float* gram, const float* sumsq, int nparts, const float* temperature,
                                    float* attn, int B, int heads, int c, pir_stream_t stream) {
  PIR_CHECK_ARG(gram && sumsq && temperature && attn && B > 0 && heads > 0 && c > 0 && c <= 256 && nparts > 0);
  hipLaunchKernelGGL(mdta_softmax_fwd_kernel<false>, dim3((unsigned)(B * heads), (unsigned)pir_cdiv(c, 16)), dim3(1024), 0,
                     (hipStream_t)stream, gram, sumsq, temperature, attn, heads, c, nparts, (float*)nullptr, 0L, 0);
  return pir_launch_status();
}

extern "C" int pir_mdta_softmax_fwd_parts(const float* gram_parts, int splits, const float* sumsq, int nparts,
                                          const float* temperature, float* gram, float* attn, int B, int heads, int c,
                                          pir_stream_t stream) {
  PIR_CHECK_ARG(gram_parts && splits > 0 && gram && sumsq && temperature && attn && B > 0 && heads > 0 && c > 0 && c <= 256 && nparts > 0);
  hipLaunchKernelGGL(mdta_softmax_fwd_kernel<true>, dim3((unsigned)(B * heads), (unsigned)pir_cdiv(c, 16)), dim3(1024), 0,
                     (hipStream_t)stream, gram_parts, sumsq, temperature, attn, heads, c, nparts, gram, (long)B * heads * c * c, splits);
  return pir_launch_status();
}

extern "C" int pir_mdta_softmax_bwd(const float* dattn, const float* attn, const float* gram, const float* sumsq,
                                    int nparts, const float* temperature, float* dgram,
                                    float* alpha_q, float* alpha_k, float* dtemp_partial,
                                    int B, int heads, int c, pir_stream_t stream) {
  PIR_CHECK_ARG(dattn && attn && gram && sumsq && temperature && dgram && alpha_q && alpha_k && dtemp_partial);
  PIR_CHECK_ARG(B > 0 && heads > 0 && c > 0 && c <= 256 && nparts > 0);
#define PIR_SMB(MJ, RW) hipLaunchKernelGGL((mdta_softmax_bwd_kernel<MJ, RW>), dim3((unsigned)(B * heads)), dim3(1024), 0, \
      (hipStream_t)stream, dattn, attn, gram, sumsq, temperature, dgram, alpha_q, alpha_k, dtemp_partial, heads, c, nparts, 0L, 0)
  if (c <= 64) PIR_SMB(1, 4); else if (c <= 128) PIR_SMB(2, 6); else PIR_SMB(4, 3);
#undef PIR_SMB
  return pir_launch_status();
}

extern "C" int pir_mdta_softmax_bwd_parts(const float* dattn_parts, int splits, const float* attn, const float* gram,
                                          const float* sumsq, int nparts, const float* temperature, float* dgram,
                                          float* alpha_q, float* alpha_k, float* dtemp_partial,
                                          int B, int heads, int c, pir_stream_t stream) {
  PIR_CHECK_ARG(dattn_parts && splits > 0 && attn && gram && sumsq && temperature && dgram && alpha_q && alpha_k && dtemp_partial);
  PIR_CHECK_ARG(B > 0 && heads > 0 && c > 0 && c <= 256 && nparts > 0);
  const long ps = (long)B * heads * c * c;
#define PIR_SMB(MJ, RW) hipLaunchKernelGGL((mdta_softmax_bwd_kernel<MJ, RW, true>), dim3((unsigned)(B * heads)), dim3(1024), 0, \
      (hipStream_t)stream, dattn_parts, attn, gram, sumsq, temperature, dgram, alpha_q, alpha_k, dtemp_partial, heads, c, nparts, ps, splits)
  if (c <= 64) PIR_SMB(1, 4); else if (c <= 128) PIR_SMB(2, 6); else PIR_SMB(4, 3);
#undef PIR_SMB
  return pir_launch_status();
}
