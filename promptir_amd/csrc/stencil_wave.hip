// Register-only depthwise 3x3 stencils for power-of-two image widths (gfx950), HBM-bound.
//
// Same structure as the wave kernel of gdfn_bwd.hip (wave_rows.h): a "unit" is one (image, channel, band of RB
// rows); W/VEC adjacent lanes own VEC columns each and slide down the band with the 3-row windows in registers;
// halos come from neighbouring lanes by DPP; every global access is a coalesced whole-row segment, rows are
// prefetched three ahead.  No LDS, no barriers, no per-tile staging phases: the LDS-tiled kernels of stencil.hip
// stage -> barrier -> compute and reach 3.4-4.4 TB/s where these stream continuously.
//
//   SW_FWD   y = dw3x3(x)  (Attention.qkv_dwconv, net/model.py:112,120)  + optional sum of y^2 per (image, channel)
//            for the first `nsq` channels = the squared L2 norms F.normalize needs for q and k (:127-128), so the
//            separate pass over q and k (pir_row_sumsq) disappears
//   SW_GATE  g = gelu_erf(dw(x)[:hid]) * dw(x)[hid:]  (FeedForward, :96-97)
//   SW_BWD   dx = dw3x3^T(dy) and the weight-gradient sums sum dy * shift(x) in one pass (autograd of :112)
#include "wave_rows.h"

namespace {

enum { SW_FWD = 0, SW_GATE = 1, SW_BWD = 2 };

struct SWArgs {
  const float* x; long x_bs;      // FWD / GATE: input; BWD: the forward's input (weight-gradient operand)
  const float* w;                 // [C][9] (GATE: [2*hid][9])
  const float* dy; long dy_bs;    // BWD: upstream gradient
  float* y; long y_bs;            // FWD: y; GATE: g; BWD: dx
  float* part;                    // FWD: sumsq partials [B*nbands][nsq]; BWD: weight-gradient partials [B*nbands][C][9]
  int B, C, H, W;                 // C = units per image (hid for GATE)
  int hid;                        // GATE: channel offset of the second half
  int nsq;                        // FWD: channels [0, nsq) get their sum of squares
  int flip;                       // FWD: 180-degree rotated taps (the transposed convolution)
  int lpu_shift, RB, nbands;
  long nunits;                    // B * C
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

template <int VEC>
__device__ __forceinline__ void conv_row(const float (&r0)[VEC + 2], const float (&r1)[VEC + 2], const float (&r2)[VEC + 2],
                                         const float (&k)[9], float (&out)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) s += k[d] * r0[j + d] + k[3 + d] * r1[j + d] + k[6 + d] * r2[j + d];
    out[j] = s;
  }
}

template <int MODE, int VEC>
__global__ __launch_bounds__(256, (MODE == SW_BWD && VEC == 4) ? 3 : 4) void stencil_wave_kernel(SWArgs a) {
  constexpr int WD = VEC + 2;
  constexpr bool TWO = MODE != SW_FWD;     // two sliding planes (GATE: x1, x2; BWD: dy, x)
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lpu = 1 << a.lpu_shift, upw = 64 >> a.lpu_shift;
  const int band = (int)(wave % a.nbands);
  const long unit = (wave / a.nbands) * upw + (lane >> a.lpu_shift);
  const int q = lane & (lpu - 1);
  const bool lane_ok = unit < a.nunits;
  const long uc = lane_ok ? unit : 0;
  const int b = (int)(uc / a.C), c = (int)(uc % a.C);
  const bool has_left = q != 0, has_right = q != lpu - 1;
  const int W = a.W, H = a.H;
  const long HW = (long)H * W;
  const int r0 = band * a.RB;
  const int rb = (r0 + a.RB <= H) ? a.RB : H - r0;

  // plane A slides through the stencil (FWD / GATE: x or x1; BWD: dy); plane B is x2 (GATE) or x (BWD)
  const float* __restrict__ pa = (MODE == SW_BWD ? a.dy + b * a.dy_bs : a.x + b * a.x_bs) + c * HW + q * VEC;
  const float* __restrict__ pb = MODE == SW_GATE ? a.x + b * a.x_bs + (c + a.hid) * HW + q * VEC
                                                 : a.x + b * a.x_bs + c * HW + q * VEC;
  float* __restrict__ po = a.y + b * a.y_bs + c * HW + q * VEC;

  float ka[9], kb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    ka[t] = a.w[c * 9 + ((MODE == SW_BWD || (MODE == SW_FWD && a.flip)) ? 8 - t : t)];   // transposed convolution = flipped taps
    kb[t] = MODE == SW_GATE ? a.w[(c + a.hid) * 9 + t] : 0.f;
  }
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;

  float A[3][WD], Bw[3][WD], PA[3][VEC], PB[3][VEC];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < WD; ++j) { A[s][j] = 0.f; Bw[s][j] = 0.f; }

  const int last = r0 + rb < H ? r0 + rb : H - 1;   // last input row this band touches
  auto load_row = [&](int row, float (&da)[VEC], float (&db)[VEC]) {
    const bool ok = lane_ok && row >= 0 && row <= last;
    const long off = (long)row * W;
    row_load<VEC>(pa + off, ok, da);
    if (TWO) row_load<VEC>(pb + off, ok, db);
  };

  {  // rows r0-1 and r0 into window slots 0 and 1; rows r0+1 .. r0+3 in flight (slot = (row - (r0-1)) % 3)
    float ta[VEC], tb[VEC];
    load_row(r0 - 1, ta, tb);
    pir_widen<VEC>(ta, A[0], has_left, has_right);
    if (TWO) pir_widen<VEC>(tb, Bw[0], has_left, has_right);
    load_row(r0, ta, tb);
    pir_widen<VEC>(ta, A[1], has_left, has_right);
    if (TWO) pir_widen<VEC>(tb, Bw[1], has_left, has_right);
    load_row(r0 + 1, PA[2], PB[2]); load_row(r0 + 2, PA[0], PB[0]); load_row(r0 + 3, PA[1], PB[1]);
  }

#define PIR_SW_STEP(PH)                                                                                          \
  if (i < rb) {                                                                                                  \
    constexpr int S0 = (PH) % 3, S1 = ((PH) + 1) % 3, S2 = ((PH) + 2) % 3;                                       \
    const int y = r0 + i;                                                                                        \
    pir_widen<VEC>(PA[S2], A[S2], has_left, has_right);              /* input row y + 1 */                       \
    if (TWO) pir_widen<VEC>(PB[S2], Bw[S2], has_left, has_right);                                                \
    load_row(y + 4, PA[S2], PB[S2]);                                 /* refill the slot just consumed */         \
    float o[VEC];                                                                                                \
    conv_row<VEC>(A[S0], A[S1], A[S2], ka, o);                                                                   \
    if (MODE == SW_GATE) {                                                                                       \
      float t2[VEC];                                                                                             \
      conv_row<VEC>(Bw[S0], Bw[S1], Bw[S2], kb, t2);                                                             \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) o[j] = gelu_erf(o[j]) * t2[j];                             \
    }                                                                                                            \
    if (MODE == SW_FWD) {                                                                                        \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[0] += o[j] * o[j];                                     \
    }                                                                                                            \
    if (MODE == SW_BWD) {                                            /* dw[dr][dc] += dy[y][p] * x[y+dr][p+dc] */ \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j)                                                            \
        _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                          \
          acc[d] += A[S1][j + 1] * Bw[S0][j + d];                                                                \
          acc[3 + d] += A[S1][j + 1] * Bw[S1][j + d];                                                            \
          acc[6 + d] += A[S1][j + 1] * Bw[S2][j + d];                                                            \
        }                                                                                                        \
    }                                                                                                            \
    if (lane_ok) row_store<VEC>(po + (long)y * W, o);                                                            \
  }                                                                                                              \
  ++i;

  for (int i = 0; i < rb;) {
    PIR_SW_STEP(0)
    PIR_SW_STEP(1)
    PIR_SW_STEP(2)
  }
#undef PIR_SW_STEP

  if (MODE == SW_FWD) {
    if (a.part) {   // wave-uniform
      const float s = pir_unit_sum(acc[0], lpu);
      if (lane_ok && q == 0 && c < a.nsq) a.part[((long)b * a.nbands + band) * a.nsq + c] = s;
    }
  } else if (MODE == SW_BWD) {
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = pir_unit_sum(acc[t], lpu);
    if (lane_ok && q == 0) {
      float* wp = a.part + (((long)b * a.nbands + band) * a.C + c) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) wp[t] = acc[t];
    }
  }
}

struct SWPlan { int vec, lpu_shift, RB, nbands; long waves; bool ok; };

int g_sw_off = 0, g_sw_rb = 0, g_sw_vec = 0;   // development overrides (pir_tune_set knobs 8, 9, 10)

// Columns per lane: the widest access the operands' alignment allows (16 bytes when possible).  Bands of 32 rows,
// halved while the launch has too few waves to fill the chip.
SWPlan swplan(long nunits, int H, int W, int max_vec) {
  SWPlan p = {0, 0, 0, 0, 0, false};
  if (g_sw_off || W < 4 || W > 256 || (W & (W - 1)) != 0) return p;
  int vec = 0;
  for (int v = max_vec; v >= 1; v >>= 1) {
    if (W % v || W / v > 64) continue;
    if (g_sw_vec) { if (v == g_sw_vec) vec = v; continue; }
    vec = v;
    break;   // the widest access wins at every level measured (tools/sw_ab.py), also when it leaves fewer waves
  }
  if (!vec) return p;
  p.vec = vec;
  const int lpu = W / vec;
  int sh = 0;
  while ((1 << sh) < lpu) ++sh;
  p.lpu_shift = sh;
  const long groups = pir_cdiv(nunits, 64 / lpu);
  int rb = 32;
  while (rb > 8 && groups * pir_cdiv(H, rb) < 3072) rb >>= 1;
  if (g_sw_rb) rb = g_sw_rb;
  if (rb > H) rb = H;
  p.RB = rb;
  p.nbands = (int)pir_cdiv(H, rb);
  p.waves = groups * p.nbands;
  p.ok = true;
  return p;
}

inline int max_vec_for(std::initializer_list<const void*> ptrs, std::initializer_list<long> strides) {
  uintptr_t bits = 0;
  for (const void* p : ptrs) bits |= reinterpret_cast<uintptr_t>(p);
  long sb = 0;
  for (long s : strides) sb |= s;
  if (bits % 16 == 0 && sb % 4 == 0) return 4;
  if (bits % 8 == 0 && sb % 2 == 0) return 2;
  return 1;
}

template <int MODE>
int launch_sw(const SWArgs& a, const SWPlan& p, hipStream_t s) {
  const long blocks = pir_cdiv(p.waves, 4);
  if (blocks <= 0 || blocks > 2147483647L) return PIR_EINVAL;
  const dim3 grid((unsigned)blocks), blk(256);
  if (p.vec == 4) hipLaunchKernelGGL((stencil_wave_kernel<MODE, 4>), grid, blk, 0, s, a);
  else if (p.vec == 2) hipLaunchKernelGGL((stencil_wave_kernel<MODE, 2>), grid, blk, 0, s, a);
  else hipLaunchKernelGGL((stencil_wave_kernel<MODE, 1>), grid, blk, 0, s, a);
  return pir_launch_status();
}

}  // namespace

int pir_stencil_wave_tune(int knob, int value) {
  if (knob == 8) g_sw_off = value; else if (knob == 9) g_sw_rb = value; else if (knob == 10) g_sw_vec = value; else return PIR_EINVAL;
  return PIR_OK;
}

// internal entry points used by stencil.hip's public functions; return 1000 when the shape is not served
// sq_parts (optional): [B][*nparts][nsq] partial sums of squares of the first nsq channels, *nparts = bands used
int pir_sw_try_fwd(const float* x, long x_bs, const float* w, int flip, float* y, long y_bs, float* sq_parts, int nsq,
                   int* nparts, int B, int C, int H, int W, hipStream_t s) {
  const SWPlan p = swplan((long)B * C, H, W, max_vec_for({x, y}, {x_bs, y_bs}));
  if (!p.ok) return 1000;
  if (nparts) *nparts = p.nbands;
  SWArgs a = {};
  a.x = x; a.x_bs = x_bs; a.w = w; a.flip = flip; a.y = y; a.y_bs = y_bs; a.part = sq_parts; a.nsq = nsq;
  a.B = B; a.C = C; a.H = H; a.W = W; a.lpu_shift = p.lpu_shift; a.RB = p.RB; a.nbands = p.nbands; a.nunits = (long)B * C;
  return launch_sw<SW_FWD>(a, p, s);
}

int pir_sw_try_gate(const float* x, long x_bs, const float* w, float* g, long g_bs, int B, int hid, int H, int W,
                    hipStream_t s) {
  const SWPlan p = swplan((long)B * hid, H, W, max_vec_for({x, g}, {x_bs, g_bs}));
  if (!p.ok) return 1000;
  SWArgs a = {};
  a.x = x; a.x_bs = x_bs; a.w = w; a.y = g; a.y_bs = g_bs; a.hid = hid;
  a.B = B; a.C = hid; a.H = H; a.W = W; a.lpu_shift = p.lpu_shift; a.RB = p.RB; a.nbands = p.nbands; a.nunits = (long)B * hid;
  return launch_sw<SW_GATE>(a, p, s);
}

// ws: [B * nbands][C][9] partial sums (the caller reduces them); *parts_out = B * nbands
int pir_sw_try_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* w, float* dx, long dx_bs,
                   float* ws, size_t ws_floats, int B, int C, int H, int W, int* parts_out, hipStream_t s) {
  const SWPlan p = swplan((long)B * C, H, W, max_vec_for({dy, x, dx}, {dy_bs, x_bs, dx_bs}));
  if (!p.ok) return 1000;
  const long parts = (long)B * p.nbands;
  // a band-count override (knob 8) or a caller-sized workspace below this plan's need: not served here, the caller
  // falls through to the LDS-tiled kernel (whose own need it checks) instead of failing the whole backward
  if ((size_t)parts * C * 9 > ws_floats) return 1000;
  SWArgs a = {};
  a.x = x; a.x_bs = x_bs; a.w = w; a.dy = dy; a.dy_bs = dy_bs; a.y = dx; a.y_bs = dx_bs; a.part = ws;
  a.B = B; a.C = C; a.H = H; a.W = W; a.lpu_shift = p.lpu_shift; a.RB = p.RB; a.nbands = p.nbands; a.nunits = (long)B * C;
  *parts_out = (int)parts;
  return launch_sw<SW_BWD>(a, p, s);
}

size_t pir_sw_bwd_ws_floats(int B, int C, int H) {
  return (size_t)B * pir_cdiv(H, 8 < H ? 8 : H) * C * 9;   // most bands the plan ever uses
}
