// Channel LayerNorm and the pixel-axis reductions of the PromptIR path (gfx950), HBM-bound.
//
// NCHW keeps the normalised axis (C) OUTERMOST, so the reference's to_3d/to_4d permutes
// (net/model.py:21-25) are never materialised: a workgroup owns 64 consecutive pixels (one
// 256-byte coalesced row segment per channel) and its 4 waves split the channels; per-pixel
// statistics are combined through LDS.  For C <= 256 the wave's channel slice stays in
// registers between the statistics and the normalisation, so x is read from HBM once.
#include "pir_common.h"

namespace {

constexpr int LN_PIX = 64;    // pixels per workgroup (one per lane)
constexpr int LN_WAVES = 4;
constexpr float LN_EPS = 1e-5f;

// NREG: compile-time bound on channels per wave kept in registers (0 = re-read from memory/L2)
template <int NREG, int WAVES>
__global__ __launch_bounds__(LN_PIX* WAVES) void ln_fwd_kernel(
    const float* __restrict__ x, long x_bs, const float* __restrict__ weight, const float* __restrict__ bias,
    float* __restrict__ y, long y_bs, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    int C, int HW, int tiles) {
  __shared__ float red[WAVES][LN_PIX];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x / tiles, p = (blockIdx.x % tiles) * LN_PIX + lane;
  const bool ok = p < HW;
  const float* xb = x + b * x_bs + p;
  float v[NREG > 0 ? NREG : 1];

  float s = 0.f;
  if (NREG > 0) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int c = wid + i * WAVES;
      v[i] = (ok && c < C) ? xb[(long)c * HW] : 0.f;
      s += v[i];
    }
  } else {
    for (int c = wid; c < C; c += WAVES) s += ok ? xb[(long)c * HW] : 0.f;
  }
  red[wid][lane] = s;
  __syncthreads();
  float tsum = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) tsum += red[w][lane];
  const float mu = tsum / (float)C;
  __syncthreads();

  float ss = 0.f;
  if (NREG > 0) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int c = wid + i * WAVES;
      const float d = (c < C) ? v[i] - mu : 0.f;
      ss += d * d;
    }
  } else {
    for (int c = wid; c < C; c += WAVES) {
      const float d = ok ? xb[(long)c * HW] - mu : 0.f;
      ss += d * d;
    }
  }
  red[wid][lane] = ss;
  __syncthreads();
  float vsum = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) vsum += red[w][lane];
  const float var = vsum / (float)C;
  const float rstd = 1.f / sqrtf(var + LN_EPS);
  if (wid == 0 && ok) {
    mean_out[(long)b * HW + p] = mu;
    rstd_out[(long)b * HW + p] = rstd;
  }
  if (!ok) return;
  float* yb = y + b * y_bs + p;
  const float shift = bias ? mu : 0.f;  // BiasFree keeps the un-centred numerator (net/model.py:41)
  if (NREG > 0) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int c = wid + i * WAVES;
      if (c < C) yb[(long)c * HW] = (v[i] - shift) * rstd * weight[c] + (bias ? bias[c] : 0.f);
    }
  } else {
    for (int c = wid; c < C; c += WAVES)
      yb[(long)c * HW] = (xb[(long)c * HW] - shift) * rstd * weight[c] + (bias ? bias[c] : 0.f);
  }
}

// dx for both variants.  g = dy*w.
//  WithBias : dx = rstd * (g - mean_c(g) - xhat * mean_c(g*xhat)),           xhat = (x-mu)*rstd
//  BiasFree : y = x*rstd*w  ->  dx = rstd*g - rstd^3 * (x-mu) * mean_c(g*x)
__global__ __launch_bounds__(LN_PIX* LN_WAVES) void ln_bwd_dx_kernel(
    const float* __restrict__ dy, long dy_bs, const float* __restrict__ x, long x_bs,
    const float* __restrict__ weight, int with_bias, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ dx, long dx_bs, const float* __restrict__ dres, long dres_bs, int C, int HW, int tiles) {
  __shared__ float red[2][LN_WAVES][LN_PIX];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x / tiles, p = (blockIdx.x % tiles) * LN_PIX + lane;
  const bool ok = p < HW;
  const float* xb = x + b * x_bs + p;
  const float* gb = dy + b * dy_bs + p;
  const float mu = ok ? mean[(long)b * HW + p] : 0.f;
  const float rs = ok ? rstd[(long)b * HW + p] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  for (int c = wid; c < C; c += LN_WAVES) {
    if (ok) {
      const float g = gb[(long)c * HW] * weight[c];
      const float xv = xb[(long)c * HW];
      s1 += g;
      s2 += g * (with_bias ? (xv - mu) * rs : xv);
    }
  }
  red[0][wid][lane] = s1;
  red[1][wid][lane] = s2;
  __syncthreads();
  const float m1 = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) / (float)C;
  const float m2 = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) / (float)C;
  if (!ok) return;
  float* db = dx + b * dx_bs + p;
  for (int c = wid; c < C; c += LN_WAVES) {
    const float g = gb[(long)c * HW] * weight[c];
    const float xv = xb[(long)c * HW];
    float r;
    if (with_bias) r = rs * (g - m1 - (xv - mu) * rs * m2);
    else r = rs * g - rs * rs * rs * (xv - mu) * m2;
    if (dres) r += dres[b * dres_bs + (long)c * HW + p];
    db[(long)c * HW] = r;
  }
}

// One pass for C <= 4*NREG: a persistent workgroup walks (image, 64-pixel tile) pairs with its
// channel slice of dy and x in registers, writes dx (+ optional residual gradient) and keeps the
// per-channel dweight / dbias sums of its pixels in registers; one shuffle reduction per channel at
// the end gives this workgroup's partial row ws[block][2][C].  dy and x are read exactly once.
template <int NREG, int WAVES>
__global__ __launch_bounds__(LN_PIX* WAVES) __attribute__((amdgpu_waves_per_eu(WAVES == 8 && NREG == 12 ? 4 : 1, 8)))
void ln_bwd_fused_kernel(
    const float* __restrict__ dy, long dy_bs, const float* __restrict__ x, long x_bs,
    const float* __restrict__ weight, int with_bias, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ dx, long dx_bs, const float* __restrict__ dres, long dres_bs,
    float* __restrict__ ws, int B, int C, int HW, int tiles) {
  __shared__ float red[2][WAVES][LN_PIX];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: channel
  // ONE_TILE (the 16-wave variant, 128-register cap): exactly one pixel tile per workgroup, so the per-channel
  // dweight / dbias sums need no accumulators across tiles - each is reduced over the wave and written at once.
  // (With accumulators and the residual staged, 5 * NREG live values per lane spilled 105 registers at NREG = 24.)
  constexpr bool ONE_TILE = WAVES >= 16 && NREG >= 24;
  float accw[ONE_TILE ? 1 : NREG], accb[ONE_TILE ? 1 : NREG];        // tests and weight loads go scalar
#pragma unroll
  for (int i = 0; i < (ONE_TILE ? 1 : NREG); ++i) { accw[i] = 0.f; accb[i] = 0.f; }
  float* row = ws + (long)blockIdx.x * 2 * C;
  const long total = (long)B * tiles;
  for (long t = blockIdx.x; t < total; t += gridDim.x) {
    const int b = (int)(t / tiles), p = (int)(t % tiles) * LN_PIX + lane;
    const bool ok = p < HW;
    const int pc = ok ? p : HW - 1;                 // clamped: loads are unconditional, stores masked
    const int off0 = wid * HW + pc;                 // 32-bit offset inside one image (C*HW < 2^31)
    // wave-uniform per-image bases: the per-channel step is added on the scalar side
    const float* __restrict__ dyb = dy + b * dy_bs;
    const float* __restrict__ xb = x + b * x_bs;
    const float* __restrict__ rb = dres ? dres + b * dres_bs : nullptr;
    float* __restrict__ ob = dx + b * dx_bs;
    const float mu = mean[(long)b * HW + pc];
    const float rs = rstd[(long)b * HW + pc];
    // phase A: issue every load of this tile (channel index clamped -> no branches, all in flight)
    // The 16-wave variant runs under a 128-register cap (1024 threads): with the residual gradient staged too,
    // 5 * NREG values per lane spill (105 registers at NREG = 24); it reads the residual in the final loop instead.
    constexpr bool LATE_RES = WAVES >= 16 && NREG >= 24;
    float g[NREG], xh[NREG], rr[LATE_RES ? 1 : NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int c = wid + i * WAVES;             // wave-uniform
      const int ic = c < C ? i : 0;                 // clamp to a valid channel of this wave
      const long step = (long)ic * WAVES * HW;
      g[i] = (dyb + step)[off0];
      xh[i] = (xb + step)[off0];
      if (!LATE_RES) rr[i] = rb ? (rb + step)[off0] : 0.f;
    }
    // phase B: arithmetic
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int c = wid + i * WAVES;
      const bool live = c < C;
      const float d = live ? g[i] : 0.f;
      const float xv = xh[i];
      const float xn = with_bias ? (xv - mu) * rs : xv * rs;   // what multiplies the weight in forward
      if (ONE_TILE) {
        const float sw = pir_wave_sum(ok ? d * xn : 0.f), sb = pir_wave_sum(ok ? d : 0.f);
        if (lane == 0 && live) { row[c] = sw; row[C + c] = sb; }
      } else if (ok) { accw[i] += d * xn; accb[i] += d; }
      g[i] = d * weight[live ? c : 0];
      const float second = with_bias ? xn : xv;                // BiasFree: second sum is over g*x
      s1 += g[i];
      s2 += g[i] * second;
      xh[i] = with_bias ? xn : xv - mu;                        // factor of m2 in dx
    }
    __syncthreads();
    red[0][wid][lane] = s1;
    red[1][wid][lane] = s2;
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { m1 += red[0][w][lane]; m2 += red[1][w][lane]; }
    m1 /= (float)C; m2 /= (float)C;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int c = wid + i * WAVES;
      if (c < C) {
        const long step = (long)i * WAVES * HW;
        float r;
        if (with_bias) r = rs * (g[i] - m1 - xh[i] * m2);
        else r = rs * g[i] - rs * rs * rs * xh[i] * m2;
        r += LATE_RES ? (rb ? (rb + step)[off0] : 0.f) : rr[i];
        if (ok) (ob + step)[off0] = r;
      }
    }
  }
  if (!ONE_TILE) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int c = wid + i * WAVES;
      const float sw = pir_wave_sum(accw[i]), sb = pir_wave_sum(accb[i]);
      if (lane == 0 && c < C) { row[c] = sw; row[C + c] = sb; }
    }
  }
}

// dweight[c] = sum_{b,p} dy * xhat ; dbias[c] = sum dy.  grid (C, S); partials [S][2][C].
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(
    const float* __restrict__ dy, long dy_bs, const float* __restrict__ x, long x_bs, int with_bias,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ ws, int B, int C, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x, S = gridDim.y, sidx = blockIdx.y;
  const long total = (long)B * HW;
  const long per = (total + S - 1) / S;
  const long e0 = sidx * per, e1 = (e0 + per < total) ? e0 + per : total;
  float sw = 0.f, sb = 0.f;
  for (long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
    const long b = e / HW, p = e % HW;
    const float g = dy[b * dy_bs + (long)c * HW + p];
    const float xv = x[b * x_bs + (long)c * HW + p];
    const float rs = rstd[e];
    sw += g * (with_bias ? (xv - mean[e]) * rs : xv * rs);
    sb += g;
  }
  const float tw = pir_block_sum(sw, red);
  const float tb = pir_block_sum(sb, red);
  if (threadIdx.x == 0) {
    ws[((long)sidx * 2 + 0) * C + c] = tw;
    ws[((long)sidx * 2 + 1) * C + c] = tb;
  }
}

// out[row] = reduce over n of f(x[row][n]); one wave per row when N is small, a block otherwise.
template <bool SQUARE>
__global__ __launch_bounds__(256) void row_reduce_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ out,
                                                         int C, int HW, float scale) {
  __shared__ float red[16];
  const int b = blockIdx.x / C, c = blockIdx.x % C;
  const float* row = x + b * x_bs + (long)c * HW;
  float s = 0.f;
  if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
    for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
      const f32x4 v = r4[i];
      s += SQUARE ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += SQUARE ? row[i] * row[i] : row[i];
  }
  const float t = pir_block_sum(s, red);
  if (threadIdx.x == 0) out[(long)b * C + c] = t * scale;
}

int g_ln_small = 1;   // pir_tune_set knob 13 (development A/B)

// ---------------------------------------------------------------------------------------------------------------
// Low-resolution levels (16 x 16 and 32 x 32 planes, 192 ... 704 channels): with 64 pixels per workgroup a batch of
// 8 images gives 32 ... 128 workgroups of up to 1024 threads - half the chip idle and every wave walking 24+ strided
// channels.  Here a workgroup owns 16 pixels: a wave = 16 pixels x 4 channel slots, so four times as many
// workgroups, each lane holding <= 12 channels; per-pixel sums go over the slots by shuffles and over the waves
// through LDS, per-channel sums (dweight / dbias) over the 16 pixel lanes by shuffles, one partial row per workgroup.
constexpr int LS_PIX = 16, LS_SLOTS = 64 / LS_PIX;

__device__ __forceinline__ float ls_sum_slots(float v) {    // over the 4 channel slots of a pixel (lanes p, p+16, p+32, p+48)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float ls_sum_pixels(float v) {   // over the 16 pixel lanes of a slot
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int NREG, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ln_fwd_small_kernel(
    const float* __restrict__ x, long x_bs, const float* __restrict__ weight, const float* __restrict__ bias,
    float* __restrict__ y, long y_bs, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    int C, int HW, int tiles) {
  __shared__ float red[2][WAVES][LS_PIX];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int px = lane & (LS_PIX - 1), slot = lane / LS_PIX;
  const int b = blockIdx.x / tiles, p = (blockIdx.x % tiles) * LS_PIX + px;
  const bool ok = p < HW;
  const int c0 = wid * LS_SLOTS + slot;
  constexpr int CSTEP = WAVES * LS_SLOTS;
  const float* xb = x + b * x_bs + (ok ? p : HW - 1);
  float v[NREG];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int c = c0 + i * CSTEP;
    v[i] = c < C ? xb[(long)c * HW] : 0.f;
    s += v[i];
  }
  s = ls_sum_slots(s);
  if (slot == 0) red[0][wid][px] = s;
  __syncthreads();
  float tsum = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) tsum += red[0][w][px];
  const float mu = tsum / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const float d = (c0 + i * CSTEP < C) ? v[i] - mu : 0.f;
    ss += d * d;
  }
  ss = ls_sum_slots(ss);
  if (slot == 0) red[1][wid][px] = ss;
  __syncthreads();
  float vsum = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) vsum += red[1][w][px];
  const float rstd = 1.f / sqrtf(vsum / (float)C + LN_EPS);
  if (wid == 0 && slot == 0 && ok) {
    mean_out[(long)b * HW + p] = mu;
    rstd_out[(long)b * HW + p] = rstd;
  }
  if (!ok) return;
  float* yb = y + b * y_bs + p;
  const float shift = bias ? mu : 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int c = c0 + i * CSTEP;
    if (c < C) yb[(long)c * HW] = (v[i] - shift) * rstd * weight[c] + (bias ? bias[c] : 0.f);
  }
}

// dx (+ residual gradient) and one partial row [2][C] of dweight / dbias per workgroup; same formulas as ln_bwd_fused_kernel
template <int NREG, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ln_bwd_small_kernel(
    const float* __restrict__ dy, long dy_bs, const float* __restrict__ x, long x_bs,
    const float* __restrict__ weight, int with_bias, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ dx, long dx_bs, const float* __restrict__ dres, long dres_bs,
    float* __restrict__ ws, int C, int HW, int tiles) {
  __shared__ float red[2][WAVES][LS_PIX];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int px = lane & (LS_PIX - 1), slot = lane / LS_PIX;
  const int b = blockIdx.x / tiles, p = (blockIdx.x % tiles) * LS_PIX + px;
  const bool ok = p < HW;
  const int pc = ok ? p : HW - 1;
  const int c0 = wid * LS_SLOTS + slot;
  constexpr int CSTEP = WAVES * LS_SLOTS;
  const float* __restrict__ dyb = dy + b * dy_bs + pc;
  const float* __restrict__ xb = x + b * x_bs + pc;
  const float* __restrict__ rb = dres ? dres + b * dres_bs + pc : nullptr;
  float* __restrict__ ob = dx + b * dx_bs + pc;
  float* __restrict__ row = ws + (long)blockIdx.x * 2 * C;
  const float mu = mean[(long)b * HW + pc], rs = rstd[(long)b * HW + pc];
  float g[NREG], xh[NREG], rr[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {           // every load of the tile in flight at once (clamped channel: no branches)
    const int c = c0 + i * CSTEP, cc = c < C ? c : C - 1;
    g[i] = dyb[(long)cc * HW];
    xh[i] = xb[(long)cc * HW];
    rr[i] = rb ? rb[(long)cc * HW] : 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int c = c0 + i * CSTEP;
    const bool live = c < C;
    const float d = live ? g[i] : 0.f;
    const float xv = xh[i];
    const float xn = with_bias ? (xv - mu) * rs : xv * rs;
    const float sw = ls_sum_pixels(ok ? d * xn : 0.f), sb = ls_sum_pixels(ok ? d : 0.f);
    if (px == 0 && live) { row[c] = sw; row[C + c] = sb; }
    g[i] = d * weight[live ? c : 0];
    s1 += g[i];
    s2 += g[i] * (with_bias ? xn : xv);
    xh[i] = with_bias ? xn : xv - mu;
  }
  s1 = ls_sum_slots(s1);
  s2 = ls_sum_slots(s2);
  if (slot == 0) { red[0][wid][px] = s1; red[1][wid][px] = s2; }
  __syncthreads();
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) { m1 += red[0][w][px]; m2 += red[1][w][px]; }
  m1 /= (float)C; m2 /= (float)C;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int c = c0 + i * CSTEP;
    if (c < C && ok) {
      float r;
      if (with_bias) r = rs * (g[i] - m1 - xh[i] * m2);
      else r = rs * g[i] - rs * rs * rs * xh[i] * m2;
      ob[(long)c * HW] = r + rr[i];
    }
  }
}

// small-plane variants apply when the 64-pixel tiling would leave the chip underfilled and <= 12 channels per lane suffice
static bool ln_small(int B, int C, int HW) { return (long)B * pir_cdiv(HW, LN_PIX) < 2L * PIR_NUM_CU && C > 64 && C <= 768 && g_ln_small; }
static int ln_small_waves(int C) { return C <= 192 ? 4 : (C <= 384 ? 8 : 16); }

int ln_threads_for(int HW) { return HW >= 4096 ? 256 : (HW >= 1024 ? 128 : 64); }

}  // namespace

int g_ln_bwd8 = -1;   // knob 16: eight waves per 64-pixel tile in the fused backward (half the registers per wave):
                      // -1 automatic (C <= 96, or planes of <= 4096 pixels: -5..-15 %; at C = 96 only once the variant is
                      // held to 128 registers = two 8-wave workgroups per CU), 0 never, 1 always
int pir_ln_tune(int knob, int value) {
  if (knob == 13) { g_ln_small = value; return PIR_OK; }
  if (knob == 16) { g_ln_bwd8 = value; return PIR_OK; }
  return PIR_EINVAL;
}

extern "C" int pir_reduce_partials(const float* parts, long stride, int S, float alpha, int accumulate,
                                   float* out, long count, pir_stream_t stream);
int pir_reduce_partials_to2(const float* parts, long stride, int S, float alpha, int accumulate, float* out, float* out2,
                            long split, long count, pir_stream_t stream);   // misc.hip

extern "C" int pir_layernorm_fwd(const float* x, long x_bs, const float* weight, const float* bias,
                                 float* y, long y_bs, float* mean, float* rstd,
                                 int B, int C, int HW, pir_stream_t stream) {
  PIR_CHECK_ARG(x && weight && y && mean && rstd && B > 0 && C > 0 && HW > 0);
  hipStream_t s = (hipStream_t)stream;
  if (ln_small(B, C, HW)) {
    const int tl = (int)pir_cdiv(HW, LS_PIX);
    const dim3 gr((unsigned)((long)B * tl));
    switch (ln_small_waves(C)) {
      case 4: hipLaunchKernelGGL((ln_fwd_small_kernel<12, 4>), gr, dim3(256), 0, s, x, x_bs, weight, bias, y, y_bs, mean, rstd, C, HW, tl); break;
      case 8: hipLaunchKernelGGL((ln_fwd_small_kernel<12, 8>), gr, dim3(512), 0, s, x, x_bs, weight, bias, y, y_bs, mean, rstd, C, HW, tl); break;
      default: hipLaunchKernelGGL((ln_fwd_small_kernel<12, 16>), gr, dim3(1024), 0, s, x, x_bs, weight, bias, y, y_bs, mean, rstd, C, HW, tl); break;
    }
    return pir_launch_status();
  }
  const int tiles = (int)pir_cdiv(HW, LN_PIX);
  dim3 grid((unsigned)((long)B * tiles));
#define PIR_LN(NR, WV) hipLaunchKernelGGL((ln_fwd_kernel<NR, WV>), grid, dim3(LN_PIX * WV), 0, s, x, x_bs, weight, bias, y, y_bs, mean, rstd, C, HW, tiles)
  const bool few = (long)B * tiles < 4L * PIR_NUM_CU;   // low-resolution levels: 16 waves per pixel tile
  if (few && C > 64 && C <= 1024) {
    if (C <= 256) PIR_LN(16, 16); else if (C <= 512) PIR_LN(32, 16); else PIR_LN(64, 16);
  } else if (C <= 64) PIR_LN(16, 4);
  else if (C <= 128) PIR_LN(32, 4);
  else if (C <= 256) PIR_LN(64, 4);
  else PIR_LN(0, 4);
#undef PIR_LN
  return pir_launch_status();
}

static int ln_param_splits(int B, int C, int HW) {
  const long total = (long)B * HW;
  long s = pir_cdiv(4L * PIR_NUM_CU, C);
  const long max_s = pir_cdiv(total, 2048);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}

static bool ln_use16(int B, int C, int HW);
static int ln_fused_blocks(int B, int HW, bool one_tile = false) {
  const long total = (long)B * pir_cdiv(HW, LN_PIX);
  if (one_tile) return (int)total;
  long g = pir_cdiv(total, 4);                      // >= 4 pixel tiles per workgroup amortise its tail
  if (g > 8L * PIR_NUM_CU) g = 8L * PIR_NUM_CU;
  if (total < 2L * PIR_NUM_CU) g = total;           // tiny tensors: one tile per workgroup
  return (int)(g < 1 ? 1 : g);
}
static bool ln_use16(int B, int C, int HW) { return (long)B * pir_cdiv(HW, LN_PIX) < 4L * PIR_NUM_CU && C > 64 && C <= 512; }

extern "C" size_t pir_layernorm_bwd_ws_floats(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return 0;
  const size_t a = (size_t)ln_param_splits(B, C, HW) * 2 * C, b = (size_t)ln_fused_blocks(B, HW, ln_use16(B, C, HW) && C > 192) * 2 * C;
  const size_t c = (size_t)B * pir_cdiv(HW, LS_PIX) * 2 * C;   // small-plane kernel: one partial row per 16-pixel tile
  return (a > b ? a : b) > c ? (a > b ? a : b) : c;
}

extern "C" int pir_layernorm_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* weight,
                                 int with_bias, const float* mean, const float* rstd,
                                 float* dx, long dx_bs, const float* dres, long dres_bs,
                                 float* dweight, float* dbias,
                                 float* ws, size_t ws_floats, int B, int C, int HW, pir_stream_t stream) {
  PIR_CHECK_ARG(dy && x && weight && mean && rstd && dx && dweight && ws && B > 0 && C > 0 && HW > 0);
  PIR_CHECK_ARG(!with_bias || dbias);
  hipStream_t s = (hipStream_t)stream;
  const int tiles = (int)pir_cdiv(HW, LN_PIX);
  int S;
  const bool w16 = ln_use16(B, C, HW);
  if (ln_small(B, C, HW)) {
    const int tl = (int)pir_cdiv(HW, LS_PIX);
    S = B * tl;
    if ((size_t)S * 2 * C > ws_floats) return PIR_ENOMEM;
    const dim3 gr((unsigned)S);
#define PIR_LNS(WV) hipLaunchKernelGGL((ln_bwd_small_kernel<12, WV>), gr, dim3(64 * WV), 0, s, dy, dy_bs, x, x_bs, weight, with_bias, \
      mean, rstd, dx, dx_bs, dres, dres_bs, ws, C, HW, tl)
    switch (ln_small_waves(C)) { case 4: PIR_LNS(4); break; case 8: PIR_LNS(8); break; default: PIR_LNS(16); break; }
#undef PIR_LNS
    int st = pir_launch_status();
    if (st) return st;
  } else if (C <= 128 || w16) {
    S = ln_fused_blocks(B, HW, w16 && C > 192);   // the NREG >= 24 variants take one tile per workgroup
    if ((size_t)S * 2 * C > ws_floats) return PIR_ENOMEM;
#define PIR_LNB(NR, WV) hipLaunchKernelGGL((ln_bwd_fused_kernel<NR, WV>), dim3((unsigned)S), dim3(LN_PIX * WV), 0, s, \
      dy, dy_bs, x, x_bs, weight, with_bias, mean, rstd, dx, dx_bs, dres, dres_bs, ws, B, C, HW, tiles)
    if (w16) { if (C <= 192) PIR_LNB(12, 16); else if (C <= 384) PIR_LNB(24, 16); else PIR_LNB(32, 16); }
    else if ((g_ln_bwd8 < 0 ? (C <= 96 || HW <= 4096) : g_ln_bwd8 != 0) && C <= 128) { if (C <= 48) PIR_LNB(6, 8); else if (C <= 64) PIR_LNB(8, 8); else if (C <= 96) PIR_LNB(12, 8); else PIR_LNB(16, 8); }
    else if (C <= 48) PIR_LNB(12, 4); else if (C <= 64) PIR_LNB(16, 4); else if (C <= 96) PIR_LNB(24, 4); else PIR_LNB(32, 4);
#undef PIR_LNB
    int st = pir_launch_status();
    if (st) return st;
  } else {
    S = ln_param_splits(B, C, HW);
    if ((size_t)S * 2 * C > ws_floats) return PIR_ENOMEM;
    hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3((unsigned)((long)B * tiles)), dim3(LN_PIX * LN_WAVES), 0, s,
                       dy, dy_bs, x, x_bs, weight, with_bias, mean, rstd, dx, dx_bs, dres, dres_bs, C, HW, tiles);
    int st = pir_launch_status();
    if (st) return st;
    hipLaunchKernelGGL(ln_bwd_param_kernel, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s,
                       dy, dy_bs, x, x_bs, with_bias, mean, rstd, ws, B, C, HW);
    st = pir_launch_status();
    if (st) return st;
  }
  // partials are [S][2][C]: one launch sums both halves, dweight from columns [0, C), dbias from [C, 2C)
  if (with_bias) return pir_reduce_partials_to2(ws, 2L * C, S, 1.f, 0, dweight, dbias, C, 2L * C, stream);
  return pir_reduce_partials(ws, 2L * C, S, 1.f, 0, dweight, C, stream);
}

extern "C" int pir_row_sumsq(const float* x, long x_bs, float* out, int B, int C, int HW, pir_stream_t stream) {
  PIR_CHECK_ARG(x && out && B > 0 && C > 0 && HW > 0);
  hipLaunchKernelGGL((row_reduce_kernel<true>), dim3((unsigned)((long)B * C)), dim3(ln_threads_for(HW)), 0,
                     (hipStream_t)stream, x, x_bs, out, C, HW, 1.f);
  return pir_launch_status();
}

extern "C" int pir_spatial_mean(const float* x, long x_bs, float* out, int B, int C, int HW, pir_stream_t stream) {
  PIR_CHECK_ARG(x && out && B > 0 && C > 0 && HW > 0);
  hipLaunchKernelGGL((row_reduce_kernel<false>), dim3((unsigned)((long)B * C)), dim3(ln_threads_for(HW)), 0,
                     (hipStream_t)stream, x, x_bs, out, C, HW, 1.f / (float)HW);
  return pir_launch_status();
}
