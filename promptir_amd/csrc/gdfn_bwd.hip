// Fused backward of the GDFN depthwise tail (net/model.py:96-97) for gfx950, HBM-bound.
//
//   forward : t = dw3x3(x) on 2*hid channels;  g = gelu_erf(t[:hid]) * t[hid:]
//   backward: dt1 = dg * t2 * gelu'(t1), dt2 = dg * gelu(t1);  dx = dw3x3^T(dt);  dw = sum dt * shift(x)
//
// Unfused this is three stencil passes (gate-backward, transposed conv, weight gradient) moving 13
// hid-channel-planes per pixel; here one workgroup stages x (2-pixel halo) and dg (1-pixel halo) of a
// (plane pair, tile) in LDS, recomputes t and dt on the 1-pixel-halo-extended tile into LDS, and then
// produces dx and the weight-gradient partial sums from LDS: 3 planes read + 2 written per hid channel.
#include "pir_common.h"

namespace {

struct GArgs {
  const float* x; long x_bs;
  const float* w;
  const float* dg; long dg_bs;
  float* dx; long dx_bs;
  float* ws;
  int B, hid, H, W;
  int CT, LPR, RT, strips, SR, tiles_r, tiles_c;
  unsigned magic_spr, magic_lpr;  // fast division by LPR+2 and LPR
};

// rows x (LPR+2) chunks of 4 floats; chunk j covers image columns w0 + 4*(j-1) ...; zero outside the image
__device__ __forceinline__ void stage(float* lds, const float* __restrict__ plane, int H, int W, int row0, int w0,
                                      int rows, int LPR, int LS, unsigned magic_spr) {
  const int spr = LPR + 2, total = rows * spr;
  for (int s = threadIdx.x; s < total; s += blockDim.x) {
    const int lr = pir_fastdiv(s, magic_spr), j = s - lr * spr;
    const int h = row0 + lr, col = w0 + (j - 1) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (h >= 0 && h < H && col >= 0 && col < W) v = *reinterpret_cast<const f32x4*>(plane + (long)h * W + col);
    *reinterpret_cast<f32x4*>(lds + lr * LS + j * 4) = v;
  }
}

// Register-staged variant: `stage_load` issues a tile's global loads into registers (zero for slots
// outside the image), `stage_store` writes them to LDS later.  A persistent workgroup prefetches row tile
// t+1 while it computes tile t.  MAXS bounds the slots per thread (checked on the host).
template <int MAXS>
__device__ __forceinline__ void stage_load(f32x4 (&regs)[MAXS], const float* __restrict__ plane, int H, int W, int row0,
                                           int w0, int rows, int LPR, unsigned magic_spr) {
  const int spr = LPR + 2, total = rows * spr;
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    const int s = threadIdx.x + i * blockDim.x;
    const int lr = pir_fastdiv(s, magic_spr), j = s - lr * spr;
    const int h = row0 + lr, col = w0 + (j - 1) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s < total && h >= 0 && h < H && col >= 0 && col < W) v = *reinterpret_cast<const f32x4*>(plane + (long)h * W + col);
    regs[i] = v;
  }
}
template <int MAXS>
__device__ __forceinline__ void stage_store(float* lds, const f32x4 (&regs)[MAXS], int rows, int LPR, int LS,
                                            unsigned magic_spr) {
  const int spr = LPR + 2, total = rows * spr;
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    const int s = threadIdx.x + i * blockDim.x;
    const int lr = pir_fastdiv(s, magic_spr), j = s - lr * spr;
    if (s < total) *reinterpret_cast<f32x4*>(lds + lr * LS + j * 4) = regs[i];
  }
}

__device__ __forceinline__ void read6(const float* row, int o, float (&v)[6]) {
  v[0] = row[o - 1];
  const f32x4 m = *reinterpret_cast<const f32x4*>(row + o);
  v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
  v[5] = row[o + 4];
}

__device__ __forceinline__ void conv4(const float (&r0)[6], const float (&r1)[6], const float (&r2)[6],
                                      const float (&k)[9], float (&out)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) s += k[d] * r0[j + d] + k[3 + d] * r1[j + d] + k[6 + d] * r2[j + d];
    out[j] = s;
  }
}

// PREFETCH: register-staged row tiles (pays when a plane has several row tiles); otherwise direct staging.
template <bool PREFETCH>
__global__ __launch_bounds__(256) void gdfn_dw_bwd_kernel(GArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[4 * 18];
  const int LS = (a.LPR + 2) * 4, RT = a.RT, LPR = a.LPR;
  float* X1 = lds;
  float* X2 = X1 + (RT + 4) * LS;
  float* DG = X2 + (RT + 4) * LS;
  float* DT1 = DG + (RT + 2) * LS;
  float* DT2 = DT1 + (RT + 2) * LS;

  // one workgroup walks every row tile of its (plane pair, column tile): the weight-gradient sums stay
  // in registers across tiles and are reduced over the workgroup once.
  int bid = blockIdx.x;
  const int tile_c = bid % a.tiles_c; bid /= a.tiles_c;
  const int b = bid / a.hid, c = bid % a.hid;
  const int w0 = tile_c * a.CT;
  const long HW = (long)a.H * a.W;
  const float* __restrict__ x1p = a.x + b * a.x_bs + c * HW;
  const float* __restrict__ x2p = x1p + a.hid * HW;
  float k1[9], k2[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { k1[t] = a.w[c * 9 + t]; k2[t] = a.w[(c + a.hid) * 9 + t]; }
  const int s = pir_fastdiv(threadIdx.x, a.magic_lpr), q = threadIdx.x - s * LPR;
  const int col = w0 + q * 4, o = 4 * (q + 1);
  float ws1[9], ws2[9], f1[9], f2[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { ws1[t] = 0.f; ws2[t] = 0.f; f1[t] = k1[8 - t]; f2[t] = k2[8 - t]; }

  constexpr int MAXS = 3;   // staging slots per thread and tile (host guarantees rows*(LPR+2) <= 3*threads)
  const float* __restrict__ dgp = a.dg + b * a.dg_bs + c * HW;
  f32x4 px1[MAXS], px2[MAXS], pdg[MAXS];
  if (PREFETCH) {
    stage_load<MAXS>(px1, x1p, a.H, a.W, -2, w0, RT + 4, LPR, a.magic_spr);
    stage_load<MAXS>(px2, x2p, a.H, a.W, -2, w0, RT + 4, LPR, a.magic_spr);
    stage_load<MAXS>(pdg, dgp, a.H, a.W, -1, w0, RT + 2, LPR, a.magic_spr);
  }

  for (int tile_r = 0; tile_r < a.tiles_r; ++tile_r) {
  const int h0 = tile_r * RT;
  if (tile_r) __syncthreads();  // previous tile's phase 3 has finished reading LDS
  if (!PREFETCH) {
    stage(X1, x1p, a.H, a.W, h0 - 2, w0, RT + 4, LPR, LS, a.magic_spr);
    stage(X2, x2p, a.H, a.W, h0 - 2, w0, RT + 4, LPR, LS, a.magic_spr);
    stage(DG, dgp, a.H, a.W, h0 - 1, w0, RT + 2, LPR, LS, a.magic_spr);
  } else {
    stage_store<MAXS>(X1, px1, RT + 4, LPR, LS, a.magic_spr);
    stage_store<MAXS>(X2, px2, RT + 4, LPR, LS, a.magic_spr);
    stage_store<MAXS>(DG, pdg, RT + 2, LPR, LS, a.magic_spr);
  }
  if (PREFETCH && tile_r + 1 < a.tiles_r) {  // next row tile's loads fly during phases 2 and 3
    stage_load<MAXS>(px1, x1p, a.H, a.W, h0 + RT - 2, w0, RT + 4, LPR, a.magic_spr);
    stage_load<MAXS>(px2, x2p, a.H, a.W, h0 + RT - 2, w0, RT + 4, LPR, a.magic_spr);
    stage_load<MAXS>(pdg, dgp, a.H, a.W, h0 + RT - 1, w0, RT + 2, LPR, a.magic_spr);
  }
  __syncthreads();

  // ---- phase 2: t and dt on the halo-extended tile (rows h0-1 .. h0+RT, columns w0-1 .. w0+CT)
  const int ipr = LPR + 2;  // items per row: LPR vector chunks + the two halo columns
  for (int it = threadIdx.x; it < (RT + 2) * ipr; it += blockDim.x) {
    const int lr = pir_fastdiv(it, a.magic_spr), k = it - lr * ipr;
    const float* x1r = X1 + lr * LS;  // X row lr <-> image row (h0-1+lr)-1
    const float* x2r = X2 + lr * LS;
    if (k < LPR) {
      const int o = 4 * (k + 1);
      float r0[6], r1[6], r2[6], t1[4], t2[4];
      read6(x1r, o, r0); read6(x1r + LS, o, r1); read6(x1r + 2 * LS, o, r2);
      conv4(r0, r1, r2, k1, t1);
      read6(x2r, o, r0); read6(x2r + LS, o, r1); read6(x2r + 2 * LS, o, r2);
      conv4(r0, r1, r2, k2, t2);
      const f32x4 g = *reinterpret_cast<const f32x4*>(DG + lr * LS + o);
      f32x4 d1, d2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float ge, gd;
        pir_gelu_both(t1[j], ge, gd);
        d1[j] = g[j] * t2[j] * gd;
        d2[j] = g[j] * ge;
      }
      *reinterpret_cast<f32x4*>(DT1 + lr * LS + o) = d1;
      *reinterpret_cast<f32x4*>(DT2 + lr * LS + o) = d2;
    } else {
      const int o = (k == LPR) ? 3 : 4 * (LPR + 1);  // column w0-1 or w0+CT
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) {
          t1 += k1[dr * 3 + dc] * x1r[dr * LS + o - 1 + dc];
          t2 += k2[dr * 3 + dc] * x2r[dr * LS + o - 1 + dc];
        }
      const float g = DG[lr * LS + o];
      float ge, gd;
      pir_gelu_both(t1, ge, gd);
      DT1[lr * LS + o] = g * t2 * gd;
      DT2[lr * LS + o] = g * ge;
    }
  }
  __syncthreads();

  // ---- phase 3: dx = dw^T(dt) and weight-gradient partial sums on the interior tile; the two planes of the pair
  // one after the other (their register windows are then live one at a time: 36 instead of 72 registers)
  if (s < a.strips && col < a.W) {
    const int r0i = s * a.SR;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const float* DT = pl ? DT2 : DT1;
      const float* XP = pl ? X2 : X1;
      const float (&ff)[9] = pl ? f2 : f1;
      float (&wsum)[9] = pl ? ws2 : ws1;
      float d0[6], d1[6], d2[6], x0[6], x1[6], x2[6];   // dt windows (rows r, r+1, r+2), x windows (rows r+1, r+2, r+3)
      read6(DT + r0i * LS, o, d0); read6(DT + (r0i + 1) * LS, o, d1);
      read6(XP + (r0i + 1) * LS, o, x0); read6(XP + (r0i + 2) * LS, o, x1);
      for (int i = 0; i < a.SR; ++i) {
        const int r = r0i + i, h = h0 + r;
        if (h >= a.H) break;
        read6(DT + (r + 2) * LS, o, d2);
        read6(XP + (r + 3) * LS, o, x2);
        float o1[4];
        conv4(d0, d1, d2, ff, o1);
        float* p1 = a.dx + b * a.dx_bs + (c + pl * a.hid) * HW + (long)h * a.W + col;
        f32x4 v1 = {o1[0], o1[1], o1[2], o1[3]};
        *reinterpret_cast<f32x4*>(p1) = v1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            wsum[d] += d1[j + 1] * x0[j + d]; wsum[3 + d] += d1[j + 1] * x1[j + d]; wsum[6 + d] += d1[j + 1] * x2[j + d];
          }
#pragma unroll
        for (int j = 0; j < 6; ++j) { d0[j] = d1[j]; d1[j] = d2[j]; x0[j] = x1[j]; x1[j] = x2[j]; }
      }
    }
  }
  }  // row tiles
  float all[18];
#pragma unroll
  for (int t = 0; t < 9; ++t) { all[t] = ws1[t]; all[9 + t] = ws2[t]; }
  const float tot = pir_block_sum_many<18>(all, red);
  const long part = (long)b * a.tiles_c + tile_c;
  float* wp = a.ws + part * (2L * a.hid * 9);
  if (threadIdx.x < 9) wp[c * 9 + threadIdx.x] = tot;
  else if (threadIdx.x < 18) wp[(c + a.hid) * 9 + threadIdx.x - 9] = tot;
}

struct GPlan { int CT, LPR, threads, strips, SR, RT, tiles_r, tiles_c; size_t lds_bytes; };

GPlan gplan(int H, int W) {
  GPlan p;
  p.CT = W < 128 ? W : 128;
  p.LPR = p.CT / 4;
  const long want = (long)H * p.LPR;
  p.threads = want >= 256 ? 256 : (int)(pir_cdiv(want, 64) * 64);
  if (p.threads < p.LPR) p.threads = (int)(pir_cdiv(p.LPR, 64) * 64);
  p.strips = p.threads / p.LPR;
  if (p.strips > H) p.strips = H;
  const int LS = (p.LPR + 2) * 4;
  const int budget_rows = 13000 / LS;            // ~52 KB of LDS -> 3 workgroups per CU
  int rt_max = (budget_rows - 14) / 5;
  if (rt_max < 1) rt_max = 1;
  int sr = rt_max / p.strips;
  if (sr < 1) sr = 1;
  if (sr > 8) sr = 8;
  const int need = (int)pir_cdiv(H, p.strips);
  if (sr > need) sr = need;
  // register staging holds at most 3 slots per thread: (RT+4)*(LPR+2) <= 3*threads
  while (sr > 1 && (p.strips * sr + 4) * (p.LPR + 2) > 3 * p.threads) --sr;
  p.SR = sr;
  p.RT = p.strips * p.SR;
  p.tiles_r = (int)pir_cdiv(H, p.RT);
  p.tiles_c = (int)pir_cdiv(W, p.CT);
  p.lds_bytes = (size_t)(5 * p.RT + 14) * LS * sizeof(float);
  return p;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" size_t pir_gdfn_dwconv_bwd_ws_floats(int B, int hid, int H, int W) {
  if (B <= 0 || hid <= 0 || H <= 0 || W <= 0) return 0;
  // fused path: partial sums; fallback path (W % 4 != 0 / unaligned): a dt buffer + the wgrad partials
  GPlan p = gplan(H, W % 4 == 0 ? W : 4);
  const size_t fused = W % 4 == 0 ? (size_t)B * p.tiles_c * 2 * hid * 9 : 0;
  const size_t fallback = (size_t)B * 2 * hid * H * W + pir_dwconv3x3_wgrad_ws_floats(B, 2 * hid, H, W);
  return fused > fallback ? fused : fallback;
}

extern "C" int pir_gdfn_dwconv_bwd(const float* x, long x_bs, const float* w, const float* dg, long dg_bs,
                                   float* dx, long dx_bs, float* dw, float* ws, size_t ws_floats,
                                   int B, int hid, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(x && w && dg && dx && dw && ws && B > 0 && hid > 0 && H > 0 && W > 0);
  const bool fast = W % 4 == 0 && al16(x) && al16(dg) && al16(dx) && x_bs % 4 == 0 && dg_bs % 4 == 0 && dx_bs % 4 == 0;
  if (!fast) {  // three-pass path through the generic stencils
    const size_t dt_floats = (size_t)B * 2 * hid * H * W;
    if (dt_floats + pir_dwconv3x3_wgrad_ws_floats(B, 2 * hid, H, W) > ws_floats) return PIR_ENOMEM;
    float* dt = ws;
    const long dt_bs = 2L * hid * H * W;
    int st = pir_dwconv3x3_gate_bwd(x, x_bs, w, dg, dg_bs, dt, dt_bs, B, hid, H, W, stream);
    if (st) return st;
    st = pir_dwconv3x3(dt, dt_bs, w, 1, dx, dx_bs, B, 2 * hid, H, W, stream);
    if (st) return st;
    return pir_dwconv3x3_wgrad(dt, dt_bs, x, x_bs, dw, ws + dt_floats, ws_floats - dt_floats, B, 2 * hid, H, W, stream);
  }
  GPlan p = gplan(H, W);
  const long parts = (long)B * p.tiles_c;
  if ((size_t)parts * 2 * hid * 9 > ws_floats) return PIR_ENOMEM;
  const bool prefetch = p.tiles_r > 1 && (p.RT + 4) * (p.LPR + 2) <= 3 * p.threads;
  if (p.lds_bytes > 64 * 1024) return PIR_EINVAL;
  GArgs a;
  a.x = x; a.x_bs = x_bs; a.w = w; a.dg = dg; a.dg_bs = dg_bs; a.dx = dx; a.dx_bs = dx_bs; a.ws = ws;
  a.B = B; a.hid = hid; a.H = H; a.W = W;
  a.magic_spr = pir_magic(p.LPR + 2); a.magic_lpr = pir_magic(p.LPR);
  a.CT = p.CT; a.LPR = p.LPR; a.RT = p.RT; a.strips = p.strips; a.SR = p.SR; a.tiles_r = p.tiles_r; a.tiles_c = p.tiles_c;
  const long blocks = (long)B * hid * p.tiles_c;
  if (blocks > 2147483647L) return PIR_EINVAL;
  if (prefetch)
    hipLaunchKernelGGL(gdfn_dw_bwd_kernel<true>, dim3((unsigned)blocks), dim3(p.threads), p.lds_bytes, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(gdfn_dw_bwd_kernel<false>, dim3((unsigned)blocks), dim3(p.threads), p.lds_bytes, (hipStream_t)stream, a);
  int st = pir_launch_status();
  if (st) return st;
  return pir_reduce_partials(ws, 2L * hid * 9, (int)parts, 1.f, 0, dw, 2L * hid * 9, stream);
}
