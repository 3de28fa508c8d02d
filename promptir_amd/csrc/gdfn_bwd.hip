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
#include "wave_rows.h"

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


// ------------------------------------------------------------------------------------------------------------
// Wave-autonomous variant (no LDS, no barriers) for power-of-two image widths.
//
// A "unit" is one (image, hidden-channel pair, band of RB rows); W/VEC adjacent lanes own VEC columns each of the
// unit and slide down its rows with everything in registers: the x window (3 rows), the dt window (3 rows), the
// taps and the 18 weight-gradient sums.  The one-column halos come from the neighbouring lanes through DPP wave
// shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) = the convolution's zero padding at the unit's edges.
// Every global access is a coalesced VEC*4-byte load/store of a whole image row segment; rows are prefetched three
// ahead into registers.  UNI: the unit fills the wave (W = 64*VEC) - taps are wave-uniform (SGPRs) and the DPP
// bound control supplies the edge zeros; otherwise 64*VEC/W units share a wave, taps sit in VGPRs and the unit
// edges are masked.  Per output row and plane pair: 3 row loads, 2 row stores - the algorithmic 5 planes - plus the
// band halos (x rows r0-2, r0-1, r0+RB, r0+RB+1 and dg rows r0-1, r0+RB are read by two bands).
struct GWArgs {
  const float* x; long x_bs;
  const float* w;
  const float* dg; long dg_bs;
  float* dx; long dx_bs;
  float* ws;
  int B, hid, H, W;
  int lpu_shift, RB, nbands;
  long npairs;
};

// gelu_erf(v) and its derivative, branch-free: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, the size of
// fp32 rounding of the cdf), sharing its exp(-v^2/2) with the pdf term.  ~17 VALU + v_rcp + v_exp instead of libm's
// erff + expf; only used in backward kernels (the forward keeps libm erff).  Checked against fp64 in tests.
__device__ __forceinline__ void gelu_both_fast(float v, float& g, float& dg) {
  const float ax = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float e = __expf(-ax * ax);                     // = exp(-v^2 / 2)
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erf_abs = fmaf(-p * t, e, 1.f);
  const float cdf = fmaf(0.5f, copysignf(erf_abs, v), 0.5f);
  g = v * cdf;
  dg = fmaf(v * 0.39894228040143267794f, e, cdf);
}

template <int VEC, bool UNI>
__global__ __launch_bounds__(256, (VEC == 2 ? 3 : 4)) void gdfn_dw_bwd_wave_kernel(GWArgs a) {
  constexpr int WD = VEC + 2;
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lpu = 1 << a.lpu_shift, upw = 64 >> a.lpu_shift;
  const int band = (int)(wave % a.nbands);
  const long pair = (wave / a.nbands) * upw + (UNI ? 0 : (lane >> a.lpu_shift));
  const int q = UNI ? lane : (lane & (lpu - 1));
  const bool lane_ok = pair < a.npairs;
  const long pc = lane_ok ? pair : 0;
  const int b = (int)(pc / a.hid), c = (int)(pc % a.hid);
  const bool has_left = q != 0, has_right = q != lpu - 1;
  const int W = a.W, H = a.H;
  const long HW = (long)H * W;
  const int r0 = band * a.RB;
  const int rb = (r0 + a.RB <= H) ? a.RB : H - r0;      // rows of this band (wave-uniform)
  const float* __restrict__ x1p = a.x + b * a.x_bs + c * HW + q * VEC;
  const float* __restrict__ x2p = x1p + a.hid * HW;
  const float* __restrict__ dgp = a.dg + b * a.dg_bs + c * HW + q * VEC;
  float* __restrict__ o1p = a.dx + b * a.dx_bs + c * HW + q * VEC;
  float* __restrict__ o2p = o1p + a.hid * HW;

  float k1[9], k2[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    k1[t] = a.w[c * 9 + t]; k2[t] = a.w[(c + a.hid) * 9 + t];
    if (UNI) {   // one unit per wave: the taps are wave-uniform, keep them in scalar registers
      k1[t] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, k1[t])));
      k2[t] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, k2[t])));
    }
  }
  float ws1[9], ws2[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { ws1[t] = 0.f; ws2[t] = 0.f; }

  // windows, slot = (row - (r0 - 2)) % 3 for x rows; dt row y = r0 - 1 + i lives in slot (i + 1) % 3
  float X1[3][WD], X2[3][WD], D1[3][WD], D2[3][WD];
  float P1[3][VEC], P2[3][VEC], PG[3][VEC];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < WD; ++j) { X1[s][j] = 0.f; X2[s][j] = 0.f; D1[s][j] = 0.f; D2[s][j] = 0.f; }

  const int x_last = r0 + rb + 1 < H ? r0 + rb + 1 : H - 1;   // last x row this band touches
  const int g_last = r0 + rb < H ? r0 + rb : H - 1;           // last dg row
  auto load_x = [&](int row, float (&d1)[VEC], float (&d2)[VEC]) {
    const bool ok = lane_ok && row >= 0 && row <= x_last;
    const long off = (long)row * W;
    row_load<VEC>(x1p + off, ok, d1);
    row_load<VEC>(x2p + off, ok, d2);
  };
  auto load_g = [&](int row, float (&d)[VEC]) {
    const bool ok = lane_ok && row >= 0 && row <= g_last;
    row_load<VEC>(dgp + (long)row * W, ok, d);
  };
  auto widen = [&](const float (&raw)[VEC], float (&wide)[WD]) {   // [left halo, own columns, right halo]
    const float l = dpp_from_lower(raw[VEC - 1]), r = dpp_from_upper(raw[0]);
#pragma unroll
    for (int j = 0; j < VEC; ++j) wide[j + 1] = raw[j];
    wide[0] = (UNI || has_left) ? l : 0.f;       // UNI: lanes 0 / 63 get the DPP bound-control zero
    wide[WD - 1] = (UNI || has_right) ? r : 0.f;
  };

  {  // prologue: x rows r0-2, r0-1 into slots 0, 1; rows r0, r0+1, r0+2 and dg rows r0-1, r0, r0+1 in flight
    float t1[VEC], t2[VEC];
    load_x(r0 - 2, t1, t2); widen(t1, X1[0]); widen(t2, X2[0]);
    load_x(r0 - 1, t1, t2); widen(t1, X1[1]); widen(t2, X2[1]);
    load_x(r0, P1[2], P2[2]); load_x(r0 + 1, P1[0], P2[0]); load_x(r0 + 2, P1[1], P2[1]);
    load_g(r0 - 1, PG[0]); load_g(r0, PG[1]); load_g(r0 + 1, PG[2]);
  }

  // one dt row per step; PH = i % 3 fixes every window slot at compile time
#define PIR_GW_STEP(PH)                                                                                         \
  if (i <= rb + 1) {                                                                                            \
    constexpr int S0 = (PH) % 3, S1 = ((PH) + 1) % 3, S2 = ((PH) + 2) % 3;                                      \
    const int y = r0 - 1 + i;                                                                                   \
    widen(P1[S2], X1[S2]); widen(P2[S2], X2[S2]);              /* x row y + 1 */                                \
    float g[VEC];                                                                                               \
    _Pragma("unroll") for (int j = 0; j < VEC; ++j) g[j] = PG[S0][j];                                           \
    load_x(y + 4, P1[S2], P2[S2]);                              /* refill the slots just consumed */            \
    load_g(y + 3, PG[S0]);                                                                                      \
    float d1[VEC], d2[VEC];                                                                                     \
    _Pragma("unroll") for (int j = 0; j < VEC; ++j) {                                                           \
      float t1 = 0.f, t2 = 0.f;                                                                                 \
      _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                           \
        t1 += k1[d] * X1[S0][j + d] + k1[3 + d] * X1[S1][j + d] + k1[6 + d] * X1[S2][j + d];                    \
        t2 += k2[d] * X2[S0][j + d] + k2[3 + d] * X2[S1][j + d] + k2[6 + d] * X2[S2][j + d];                    \
      }                                                                                                         \
      float ge, gd;                                                                                             \
      gelu_both_fast(t1, ge, gd);                                                                               \
      d1[j] = g[j] * t2 * gd;                                                                                   \
      d2[j] = g[j] * ge;                                                                                        \
    }                                                                                                           \
    widen(d1, D1[S1]); widen(d2, D2[S1]);                       /* dt row y */                                  \
    if (i >= 1 && i <= rb) {                                    /* rows of this band: weight-gradient sums */   \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j)                                                           \
        _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                         \
          ws1[d] += d1[j] * X1[S0][j + d]; ws1[3 + d] += d1[j] * X1[S1][j + d]; ws1[6 + d] += d1[j] * X1[S2][j + d]; \
          ws2[d] += d2[j] * X2[S0][j + d]; ws2[3 + d] += d2[j] * X2[S1][j + d]; ws2[6 + d] += d2[j] * X2[S2][j + d]; \
        }                                                                                                       \
    }                                                                                                           \
    if (i >= 2) {                                               /* dx row y - 1 from dt rows y-2, y-1, y */     \
      float o1[VEC], o2[VEC];                                                                                   \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) {                                                         \
        float s1 = 0.f, s2 = 0.f;                                                                               \
        _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                         \
          s1 += k1[8 - d] * D1[S2][j + d] + k1[5 - d] * D1[S0][j + d] + k1[2 - d] * D1[S1][j + d];              \
          s2 += k2[8 - d] * D2[S2][j + d] + k2[5 - d] * D2[S0][j + d] + k2[2 - d] * D2[S1][j + d];              \
        }                                                                                                       \
        o1[j] = s1; o2[j] = s2;                                                                                 \
      }                                                                                                         \
      if (lane_ok) {                                                                                            \
        const long off = (long)(y - 1) * W;                                                                     \
        row_store<VEC>(o1p + off, o1);                                                                          \
        row_store<VEC>(o2p + off, o2);                                                                          \
      }                                                                                                         \
    }                                                                                                           \
  }                                                                                                             \
  ++i;

  for (int i = 0; i <= rb + 1;) {
    PIR_GW_STEP(0)
    PIR_GW_STEP(1)
    PIR_GW_STEP(2)
  }
#undef PIR_GW_STEP

  // weight-gradient sums of the unit: butterfly over its lanes, lane 0 of the unit writes the band's partial
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v1 = ws1[t], v2 = ws2[t];
    for (int off = lpu >> 1; off > 0; off >>= 1) { v1 += __shfl_xor(v1, off, 64); v2 += __shfl_xor(v2, off, 64); }
    ws1[t] = v1; ws2[t] = v2;
  }
  if (lane_ok && q == 0) {
    float* wp = a.ws + ((long)b * a.nbands + band) * (2L * a.hid * 9);
#pragma unroll
    for (int t = 0; t < 9; ++t) { wp[c * 9 + t] = ws1[t]; wp[(c + a.hid) * 9 + t] = ws2[t]; }
  }
}

struct GWPlan { int vec, lpu_shift, RB, nbands; long waves; bool ok; };

int g_gw_rb = 0, g_gw_off = 0;   // development overrides (pir_tune_set knobs 6, 7)

// W = 128 / 64: one unit per wave (2 / 1 columns per lane); narrower planes: one column per lane, 64 / W units per
// wave.  (Four columns per lane need > 168 registers: W = 256 stays on the LDS-tiled kernel.)  Bands of 32 rows,
// halved while the launch has too few waves to fill the chip.
GWPlan gwplan(int B, int hid, int H, int W) {
  GWPlan p = {0, 0, 0, 0, 0, false};
  if (g_gw_off || W < 4 || W > 128 || (W & (W - 1)) != 0) return p;
  const long npairs = (long)B * hid;
  const int vec = W == 128 ? 2 : 1;
  p.vec = vec;
  const int lpu = W / vec;
  int sh = 0;
  while ((1 << sh) < lpu) ++sh;
  p.lpu_shift = sh;
  const long groups = pir_cdiv(npairs, 64 / lpu);
  int rb = 32;
  while (rb > 8 && groups * pir_cdiv(H, rb) < 3072) rb >>= 1;
  if (g_gw_rb) rb = g_gw_rb;
  if (rb > H) rb = H;
  p.RB = rb;
  p.nbands = (int)pir_cdiv(H, rb);
  p.waves = groups * p.nbands;
  p.ok = true;
  return p;
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

extern "C" int pir_reduce_partials(const float* parts, long stride, int S, float alpha, int accumulate,
                                   float* out, long count, pir_stream_t stream);
int pir_gdfn_wave_tune(int knob, int value) {
  if (knob == 6) g_gw_rb = value; else if (knob == 7) g_gw_off = value; else return PIR_EINVAL;
  return PIR_OK;
}

extern "C" size_t pir_gdfn_dwconv_bwd_ws_floats(int B, int hid, int H, int W) {
  if (B <= 0 || hid <= 0 || H <= 0 || W <= 0) return 0;
  // fused path: partial sums; fallback path (W % 4 != 0 / unaligned): a dt buffer + the wgrad partials
  GPlan p = gplan(H, W % 4 == 0 ? W : 4);
  size_t fused = W % 4 == 0 ? (size_t)B * p.tiles_c * 2 * hid * 9 : 0;
  const size_t wave = (size_t)B * pir_cdiv(H, 8 < H ? 8 : H) * 2 * hid * 9;   // most bands the wave kernel ever uses
  if (wave > fused) fused = wave;
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
  const GWPlan gw = gwplan(B, hid, H, W);
  // (a workspace below the wave plan's need - band-height override, caller-sized buffer - falls through to the tiled kernel)
  if (gw.ok && (size_t)B * gw.nbands * 2 * hid * 9 <= ws_floats && (gw.vec == 1 || (x_bs % gw.vec == 0 && dg_bs % gw.vec == 0 && dx_bs % gw.vec == 0 &&
                               (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dg) | reinterpret_cast<uintptr_t>(dx)) % (4 * gw.vec) == 0))) {   // register-only sliding-window kernel
    const long parts = (long)B * gw.nbands;
    GWArgs g;
    g.x = x; g.x_bs = x_bs; g.w = w; g.dg = dg; g.dg_bs = dg_bs; g.dx = dx; g.dx_bs = dx_bs; g.ws = ws;
    g.B = B; g.hid = hid; g.H = H; g.W = W; g.lpu_shift = gw.lpu_shift; g.RB = gw.RB; g.nbands = gw.nbands;
    g.npairs = (long)B * hid;
    const long blocks = pir_cdiv(gw.waves, 4);
    if (blocks > 2147483647L) return PIR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool uni = gw.lpu_shift == 6;
    const dim3 grid((unsigned)blocks), blk(256);
    if (gw.vec == 2) hipLaunchKernelGGL((gdfn_dw_bwd_wave_kernel<2, true>), grid, blk, 0, s, g);
    else if (uni) hipLaunchKernelGGL((gdfn_dw_bwd_wave_kernel<1, true>), grid, blk, 0, s, g);
    else hipLaunchKernelGGL((gdfn_dw_bwd_wave_kernel<1, false>), grid, blk, 0, s, g);
    int st = pir_launch_status();
    if (st) return st;
    return pir_reduce_partials(ws, 2L * hid * 9, (int)parts, 1.f, 0, dw, 2L * hid * 9, stream);
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
