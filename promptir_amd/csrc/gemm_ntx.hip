// Weight gradients of the 1x1 convolutions with one SMALL operand (M2 <= 96 rows: the layer's input channels) on the
// bf16x3 matrix cores - "X private, Y shared" (gfx950).
//
//   G[i][j] = sum_{r < BR} sum_n X[r][i][n] * Y[r][j][n]        (contraction over pixels, split-K over workgroups)
//
// The tiled kernel (gemm.hip, gemm_nt_x3_kernel) stages BOTH operands through LDS per 128 x 96 output tile: every
// 128-row tile splits the small operand again, every element is written to and read back from LDS, and the conversion
// work (VALU active 0.46) co-limits the matrix pipe (MFMA busy 0.39; round-2 counters).  Here ONE workgroup owns the
// whole M1 x M2 output of its pixel range:
//   * a wave owns RBW 32-row blocks of the tall operand X PRIVATELY: the MFMA A-operand layout (lane = row, 8
//     consecutive pixels per lane) is exactly what 16-byte row loads deliver, so X goes global -> registers -> split ->
//     MFMA without touching LDS, each element split exactly once;
//   * only the small operand Y (<= 96 rows) is shared: each thread loads 8 pixels of one row, splits them and writes
//     three 16-byte fragments into a double-buffered LDS stage (18 KB per 32 pixels), one barrier per 32 pixels;
//   * per 32 pixels a wave issues 2 x TN x RBW x 6 MFMAs against RBW x 16 + <= 8 split elements per lane: a third of
//     the tiled kernel's VALU work per MFMA, and a sixth of its LDS traffic.
// The k index of an MFMA is free as long as both operands agree: lane half h takes pixels 16h .. 16h + 15 of a
// 32-pixel step (one 64-byte run per lane, a full 128-byte line per row), MFMA m = 0, 1 of the step uses its first /
// second eight.  Partial tiles go to the split-K workspace and through the deterministic nt_reduce of gemm.hip.
#include "gemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct XFrag3 { bf16x8 hi, mid, lo; };

__device__ __forceinline__ XFrag3 xp_split8(const f32x4& a, const f32x4& b) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  XFrag3 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    f.hi[j] = h; f.mid[j] = m; f.lo[j] = (__bf16)r2;
  }
  return f;
}

struct XpArgs {
  pir_gemm_nt_t g;
  int splits;        // workgroups = split-K slices
  int steps_per_r;   // N / 32
  int rowblocks;     // ceil(M1 / 32)
  // YLN: the shared operand is normalised as it is staged - Y <- (Y - mean[pixel]) rstd[pixel] gamma[row] + beta[row], the
  // WithBias channel LayerNorm (net/model.py:60-63) of the convolution's input, which then never has to be materialised
  const float* y_mean; const float* y_rstd;   // [BR][N]
  const float* y_gamma; const float* y_beta;  // [M2]
};

// RBW: 32-row blocks of X per wave; TN: 32-column blocks of the output (ceil(M2 / 32)); the number of waves is the
// launch's (rowblocks / RBW rounded up)
// GRP: the four 16-byte loads that make up a lane's 64-byte run of a row (and, with the partner lane half, the row's whole
// 128-byte line of the step) are issued BACK TO BACK instead of half a step apart.  Issued apart, every line is touched at
// two times ~1 us apart with 64 - 128 KB of other lines per CU in between - rows lie a power of two apart (one image
// plane), so they compete for a few cache sets - and the counters showed 1.6 - 2.2x the algorithmic bytes FETCHED from
// memory (profiles/r04_pmc_summary.txt: 2 x 557 MB against 635 MB for the 510 x 96 weight gradient, which ran at 6.2 TB/s
// of real traffic).  One row block per wave: a second register set, loaded a whole step ahead.  Two row blocks: the units
// run block-major, (e, m) instead of (m, e), so that all registers of a block are free at its second unit and are
// refilled there in one go, two units ahead of their use.
template <int RBW, int TN, bool YLN = false, bool GRP = false>
__global__ __launch_bounds__(RBW == 2 ? 512 : 576) __attribute__((amdgpu_waves_per_eu(2)))
void gemm_nt_xp_kernel(XpArgs p) {
  constexpr int YR = TN * 32;             // rows of the shared operand held per stage (padded)
  constexpr int YU = 4 * YR;              // 16-byte units per part and stage: (m, h) x rows
  constexpr int STAGE = 3 * YU;
  __shared__ bf16x8 smem[2 * STAGE];
  const pir_gemm_nt_t& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwv = (int)(blockDim.x >> 6);
  const int h = lane >> 5, r = lane & 31;

  const int total = g.BR * p.steps_per_r;
  const int per = (total + p.splits - 1) / p.splits;
  const int s_begin = (int)blockIdx.x * per, s_end = s_begin + per < total ? s_begin + per : total;

  // ---- X: the wave's own rows.  Lane (r, h) reads pixels 16h .. 16h + 15 of row 32 rb + r: four 16-byte loads.
  int xoff[RBW];
#pragma unroll
  for (int e = 0; e < RBW; ++e) {
    const int rb = wid + e * nwv;
    const int row = rb * 32 + r, rc = row < g.M1 ? row : g.M1 - 1;     // rows beyond M1 only feed discarded outputs
    xoff[e] = rc * (int)g.ldx + 16 * h;
  }
  // ---- Y: thread -> (row j, pixel group q of 8): h' = q >> 1, m = q & 1
  const bool yact = tid < YR * 4;
  const int yj = tid >> 2, yq = tid & 3;
  const int yoff = (yj < g.M2 ? yj : g.M2 - 1) * (int)g.ldy + 8 * yq;
  const int ydst = ((yq & 1) * 2 + (yq >> 1)) * YR + yj;             // unit (m, h', row) inside a part

  long stat_off = 0;                                                  // YLN: offset of the step's pixels in mean / rstd
  auto base = [&](int s, const float*& xp, const float*& yp) {       // image and pixel of flattened step s
    const int sc = s < total ? s : total - 1;
    const int img = sc / p.steps_per_r, st = sc - img * p.steps_per_r;
    xp = g.X + (long)img * g.x_sr + st * 32;
    yp = g.Y + (long)img * g.y_sr + st * 32;
    if constexpr (YLN) stat_off = (long)img * g.N + st * 32 + 8 * yq;
  };
  const float y_gam = YLN ? p.y_gamma[yj < g.M2 ? yj : g.M2 - 1] : 1.f, y_bet = YLN ? p.y_beta[yj < g.M2 ? yj : g.M2 - 1] : 0.f;
  f32x4 ym[2], ys[2];
  auto load_stats = [&]() {
    if constexpr (YLN) {
      ym[0] = *reinterpret_cast<const f32x4*>(p.y_mean + stat_off); ym[1] = *reinterpret_cast<const f32x4*>(p.y_mean + stat_off + 4);
      ys[0] = *reinterpret_cast<const f32x4*>(p.y_rstd + stat_off); ys[1] = *reinterpret_cast<const f32x4*>(p.y_rstd + stat_off + 4);
    }
  };
  auto ysplit = [&](const f32x4& a, const f32x4& b) {
    if constexpr (YLN) return xp_split8((a - ym[0]) * ys[0] * y_gam + y_bet, (b - ym[1]) * ys[1] * y_gam + y_bet);
    else return xp_split8(a, b);
  };

  f32x4 xr[RBW][4];                 // raw X of the current step, refilled half by half for the next one
  f32x4 yr[2];
  const float *xp, *yp;
  base(s_begin, xp, yp);
#pragma unroll
  for (int e = 0; e < RBW; ++e)
#pragma unroll
    for (int q = 0; q < 4; ++q) xr[e][q] = *reinterpret_cast<const f32x4*>(xp + xoff[e] + 4 * q);
  __builtin_amdgcn_sched_barrier(0);
  yr[0] = *reinterpret_cast<const f32x4*>(yp + yoff);
  yr[1] = *reinterpret_cast<const f32x4*>(yp + yoff + 4);
  load_stats();
  __builtin_amdgcn_sched_barrier(0);
  {
    const XFrag3 f = ysplit(yr[0], yr[1]);
    if (yact) { smem[ydst] = f.hi; smem[YU + ydst] = f.mid; smem[2 * YU + ydst] = f.lo; }
  }
  __syncthreads();

  f32x16 acc[RBW][TN];
#pragma unroll
  for (int e = 0; e < RBW; ++e)
#pragma unroll
    for (int c = 0; c < TN; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[e][c][q] = 0.f;

  if constexpr (RBW == 1 && GRP) {
    // one row block per wave, two register sets: the whole next step is requested at the top of the current one
    f32x4 xb[4];
    auto step = [&](int s, f32x4 (&cur)[4], f32x4 (&nxt)[4]) {
      const int buf = (s - s_begin) & 1;
      const float *xn, *yn;
      base(s + 1, xn, yn);            // (the step after the last one re-reads the last: unused)
      const bf16x8* bp = smem + buf * STAGE + h * YR + r + ((s - s_begin) >> 30);   // opaque zero: see gemm_res.hip
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) nxt[q] = *reinterpret_cast<const f32x4*>(xn + xoff[0] + 4 * q);
      yr[0] = *reinterpret_cast<const f32x4*>(yn + yoff);
      yr[1] = *reinterpret_cast<const f32x4*>(yn + yoff + 4);
      load_stats();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const XFrag3 cf = xp_split8(cur[2 * m], cur[2 * m + 1]);
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          const bf16x8 bh = bp[m * 2 * YR + c * 32], bm = bp[YU + m * 2 * YR + c * 32], bl = bp[2 * YU + m * 2 * YR + c * 32];
          acc[0][c] = pir_mfma_x3(cf.hi, cf.mid, cf.lo, bh, bm, bl, acc[0][c]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {   // the next step's slice of the shared operand into the other buffer
        const XFrag3 f = ysplit(yr[0], yr[1]);
        bf16x8* dst = smem + (buf ^ 1) * STAGE;
        if (yact) { dst[ydst] = f.hi; dst[YU + ydst] = f.mid; dst[2 * YU + ydst] = f.lo; }
      }
      __syncthreads();
    };
    int s = s_begin;
    for (; s + 1 < s_end; s += 2) { step(s, xr[0], xb); step(s + 1, xb, xr[0]); }
    if (s < s_end) step(s, xr[0], xb);
  } else if constexpr (RBW == 1) {
    // one row block per wave: conversion, refill of the consumed registers with the NEXT step's pixels (a whole step
    // ahead), multiply
    for (int s = s_begin; s < s_end; ++s) {
      const int buf = (s - s_begin) & 1;
      const float *xn, *yn;
      base(s + 1, xn, yn);            // (the step after the last one re-reads the last: unused)
      const bf16x8* bp = smem + buf * STAGE + h * YR + r + ((s - s_begin) >> 30);   // opaque zero: see gemm_res.hip
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        __builtin_amdgcn_sched_barrier(0);
        const XFrag3 cur = xp_split8(xr[0][2 * m], xr[0][2 * m + 1]);
        xr[0][2 * m] = *reinterpret_cast<const f32x4*>(xn + xoff[0] + 8 * m);
        xr[0][2 * m + 1] = *reinterpret_cast<const f32x4*>(xn + xoff[0] + 8 * m + 4);
        if (m == 0) {
          yr[0] = *reinterpret_cast<const f32x4*>(yn + yoff);
          yr[1] = *reinterpret_cast<const f32x4*>(yn + yoff + 4);
          load_stats();
        }
#pragma unroll
        for (int c = 0; c < TN; ++c) {
          const bf16x8 bh = bp[m * 2 * YR + c * 32], bm = bp[YU + m * 2 * YR + c * 32], bl = bp[2 * YU + m * 2 * YR + c * 32];
          acc[0][c] = pir_mfma_x3(cur.hi, cur.mid, cur.lo, bh, bm, bl, acc[0][c]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {   // the next step's slice of the shared operand into the other buffer
        const XFrag3 f = ysplit(yr[0], yr[1]);
        bf16x8* dst = smem + (buf ^ 1) * STAGE;
        if (yact) { dst[ydst] = f.hi; dst[YU + ydst] = f.mid; dst[2 * YU + ydst] = f.lo; }
      }
      __syncthreads();
    }
  } else if constexpr (GRP) {
  // block-major units u = (step, e, m): conversion of unit u + 1 in the shadow of unit u's MFMAs as below; at a block's
  // second unit (m = 1) all of its raw registers have been converted and the block's whole next step is requested
  XFrag3 xf = xp_split8(xr[0][0], xr[0][1]);
  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    const float *xn, *yn;
    base(s + 1, xn, yn);            // (the step after the last one re-reads the last: unused)
    const bf16x8* bp = smem + buf * STAGE + h * YR + r + ((s - s_begin) >> 30);   // opaque zero: see gemm_res.hip
#pragma unroll
    for (int e = 0; e < RBW; ++e) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int c = 0; c < TN; ++c) { bh[c] = bp[m * 2 * YR + c * 32]; bm[c] = bp[YU + m * 2 * YR + c * 32]; bl[c] = bp[2 * YU + m * 2 * YR + c * 32]; }
        const XFrag3 cur = xf;
        if (m == 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) xr[e][q] = *reinterpret_cast<const f32x4*>(xn + xoff[e] + 4 * q);
        }
        if (m == 0 && e == 0) {
          yr[0] = *reinterpret_cast<const f32x4*>(yn + yoff);
          yr[1] = *reinterpret_cast<const f32x4*>(yn + yoff + 4);
          load_stats();
        }
        // next unit: (e, 1), (e + 1, 0) or the next step's (0, 0) - whose registers were refilled two units ago
        const int ne = m == 0 ? e : (e + 1 < RBW ? e + 1 : 0), nm = m ^ 1;
        xf = xp_split8(xr[ne][2 * nm], xr[ne][2 * nm + 1]);
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[e][c] = pir_mfma_x3(cur.hi, cur.mid, cur.lo, bh[c], bm[c], bl[c], acc[e][c]);
        constexpr int NV = 0;
        (void)NV;
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * TN, 0);
        const int nvm = (m == 1 ? 4 : 0) + ((m == 0 && e == 0) ? (YLN ? 6 : 2) : 0);
#pragma unroll
        for (int q = 0; q < 6 * TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (q < nvm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (96 + 6 * TN - 1) / (6 * TN), 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // the next step's slice of the shared operand into the other buffer
      const XFrag3 f = ysplit(yr[0], yr[1]);
      bf16x8* dst = smem + (buf ^ 1) * STAGE;
      if (yact) { dst[ydst] = f.hi; dst[YU + ydst] = f.mid; dst[2 * YU + ydst] = f.lo; }
    }
    __syncthreads();
  }
  } else {
  // Software pipeline over units u = (step, m, row block e): while the 6 x TN MFMAs of unit u run, the conversion of
  // unit u + 1's eight pixels per lane is issued in their shadow (the waves of a workgroup run in lockstep from the
  // per-step barrier: without the interleave both waves of a SIMD convert, then both multiply), and the registers the
  // previous conversion emptied are refilled with the same pixels of the NEXT step.
  XFrag3 xf = xp_split8(xr[0][0], xr[0][1]);
  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    const float *xn, *yn;
    base(s + 1, xn, yn);            // (the step after the last one re-reads the last: unused)
    const bf16x8* bp = smem + buf * STAGE + h * YR + r + ((s - s_begin) >> 30);   // opaque zero: see gemm_res.hip
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x8 bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int c = 0; c < TN; ++c) { bh[c] = bp[m * 2 * YR + c * 32]; bm[c] = bp[YU + m * 2 * YR + c * 32]; bl[c] = bp[2 * YU + m * 2 * YR + c * 32]; }
#pragma unroll
      for (int e = 0; e < RBW; ++e) {
        __builtin_amdgcn_sched_barrier(0);
        const XFrag3 cur = xf;
        // refill what unit u's conversion (issued one unit ago) consumed
        xr[e][2 * m] = *reinterpret_cast<const f32x4*>(xn + xoff[e] + 8 * m);
        xr[e][2 * m + 1] = *reinterpret_cast<const f32x4*>(xn + xoff[e] + 8 * m + 4);
        if (m == 0 && e == 0) {
          yr[0] = *reinterpret_cast<const f32x4*>(yn + yoff);
          yr[1] = *reinterpret_cast<const f32x4*>(yn + yoff + 4);
          load_stats();
        }
        // next unit: (m, e + 1), (m + 1, 0) or the next step's (0, 0) - whose registers were refilled a step ago
        constexpr int dummy = 0; (void)dummy;
        const int ne = e + 1 < RBW ? e + 1 : 0, nm = e + 1 < RBW ? m : (m + 1) & 1;
        xf = xp_split8(xr[ne][2 * nm], xr[ne][2 * nm + 1]);
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[e][c] = pir_mfma_x3(cur.hi, cur.mid, cur.lo, bh[c], bm[c], bl[c], acc[e][c]);
        // interleave: per MFMA a share of the ~90 conversion operations (and the loads among the first)
        if (e == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3 * TN, 0);
#pragma unroll
        for (int q = 0; q < 6 * TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (q < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (96 + 6 * TN - 1) / (6 * TN), 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // the next step's slice of the shared operand into the other buffer
      const XFrag3 f = ysplit(yr[0], yr[1]);
      bf16x8* dst = smem + (buf ^ 1) * STAGE;
      if (yact) { dst[ydst] = f.hi; dst[YU + ydst] = f.mid; dst[2 * YU + ydst] = f.lo; }
    }
    __syncthreads();
  }

  }

  // partial tile of this slice
  float* __restrict__ P = g.ws + (long)blockIdx.x * ((long)g.M1 * g.M2);
#pragma unroll
  for (int e = 0; e < RBW; ++e) {
    const int rb = wid + e * nwv;
    if (rb >= p.rowblocks) continue;
#pragma unroll
    for (int c = 0; c < TN; ++c) {
      const int jj = c * 32 + r;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ii = rb * 32 + pir_c_row(q, lane);
        if (ii < g.M1 && jj < g.M2) P[(long)ii * g.M2 + jj] = acc[e][c][q];
      }
    }
  }
}

int g_xp_mode = -1;   // knob 25: -1 automatic, 0 never, 1 whenever the shape is served
int g_xp_rbw2_min = 10;   // knob 40: row blocks from which a wave takes two (the software-pipelined kernel)
int g_xp_grp = 1;     // knob 38: a row's loads of a step issued back to back (GRP): 1 two-row-block kernels only (the one-block kernels
                      // are not bound by bytes: same-box A/B 1.02 - 1.11 of the half-step refill's time), 2 all, 0 none

struct XpPlan { int rbw, tn, nwv, splits, rowblocks; };

bool xp_plan(const pir_gemm_nt_t& g, XpPlan& pl, bool yln = false) {
  if (g_xp_mode == 0) return false;
  if (g.O1 * g.O2 != 1 || g.H != 0 || g.N % 32 != 0 || g.N < 32) return false;
  if (g.M2 > 96 || g.M2 < 24 || g.M1 < 127) return false;    // (square 96 x 96 / 48 x 48 products: the tiled kernel wins, tools/ntx_ab.py)
  auto al = [](const float* q, long sr, long ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && sr % 4 == 0 && ld % 4 == 0; };
  if (!al(g.X, g.x_sr, g.ldx) || !al(g.Y, g.y_sr, g.ldy)) return false;
  if ((long)g.M1 * g.ldx >= (1L << 31) || (long)g.M2 * g.ldy >= (1L << 31)) return false;   // 32-bit row offsets
  pl.rowblocks = (int)pir_cdiv(g.M1, 32);
  if (pl.rowblocks > 16) return false;
  pl.rbw = pl.rowblocks >= g_xp_rbw2_min ? 2 : 1;
  pl.nwv = (int)pir_cdiv(pl.rowblocks, pl.rbw);
  if (pl.rbw == 2 && pl.nwv > 8) return false;            // (RBW = 2 kernels are built for <= 512 threads)
  pl.tn = (int)pir_cdiv(g.M2, 32);
  if (pl.tn < 2) pl.tn = 2;
  if (pl.nwv * 64 < pl.tn * 32 * 4) pl.nwv = pl.tn * 2;      // enough threads for the shared operand's stage (4 per row)
  const long total = (long)g.BR * (g.N / 32);
  if (total >= (1L << 31)) return false;
  long splits = (long)PIR_NUM_CU * (pl.nwv <= 4 ? 2 : 1);
  const long by_work = total / 8 > 0 ? total / 8 : 1;          // at least 256 pixels per slice
  if (splits > by_work) splits = by_work;
  if (g_xp_mode < 0 && total / splits < (yln ? 8 : 16)) return false;   // short pixel ranges: the tiled kernel's finer split wins
  pl.splits = (int)splits;
  return true;
}

}  // namespace

int pir_nt_xp_tune(int knob, int value) {
  if (knob == 25) { g_xp_mode = value; return PIR_OK; }
  if (knob == 38) { g_xp_grp = value; return PIR_OK; }
  if (knob == 40) { g_xp_rbw2_min = value; return PIR_OK; }
  return PIR_EINVAL;
}

// split-K slices the kernel would use for this call (0: not served) - workspace sizing
int pir_nt_xp_splits(const pir_gemm_nt_t* a) {
  XpPlan pl;
  return xp_plan(*a, pl) ? pl.splits : 0;
}

// Launches the split-K kernel only (partials into g.ws, `*splits` slices); the caller runs the reduction.  1000: not served.
int pir_nt_xp_launch(const pir_gemm_nt_t* a, int* splits, hipStream_t s, const float* y_mean, const float* y_rstd,
                     const float* y_gamma, const float* y_beta) {
  XpPlan pl;
  if (!xp_plan(*a, pl, y_mean != nullptr)) return 1000;
  if ((size_t)pl.splits * a->M1 * a->M2 > a->ws_floats) return 1000;
  XpArgs xa;
  xa.g = *a;
  xa.y_mean = y_mean; xa.y_rstd = y_rstd; xa.y_gamma = y_gamma; xa.y_beta = y_beta;
  xa.splits = pl.splits;
  xa.steps_per_r = a->N / 32;
  xa.rowblocks = pl.rowblocks;
  const dim3 grid((unsigned)pl.splits), block((unsigned)pl.nwv * 64);
#define PIR_XP_GO(RBW_, TN_, YLN_)                                                                         \
  do {                                                                                                     \
    if (g_xp_grp >= 2 || (g_xp_grp == 1 && RBW_ == 2)) hipLaunchKernelGGL((gemm_nt_xp_kernel<RBW_, TN_, YLN_, true>), grid, block, 0, s, xa);   \
    else hipLaunchKernelGGL((gemm_nt_xp_kernel<RBW_, TN_, YLN_, false>), grid, block, 0, s, xa);           \
  } while (0)
  if (y_mean) {
    if (pl.rbw == 2 && pl.tn == 3) PIR_XP_GO(2, 3, true);
    else if (pl.rbw == 2) PIR_XP_GO(2, 2, true);
    else if (pl.tn == 3) PIR_XP_GO(1, 3, true);
    else PIR_XP_GO(1, 2, true);
  } else {
    if (pl.rbw == 2 && pl.tn == 3) PIR_XP_GO(2, 3, false);
    else if (pl.rbw == 2) PIR_XP_GO(2, 2, false);
    else if (pl.tn == 3) PIR_XP_GO(1, 3, false);
    else PIR_XP_GO(1, 2, false);
  }
#undef PIR_XP_GO
  *splits = pl.splits;
  return pir_launch_status();
}

int pir_nt_reduce_launch(const float* ws, int splits, int M1, int M2, float* G, long g_so, long g_si, long g_sj, float alpha,
                         int accumulate, hipStream_t s);   // gemm.hip: the deterministic second stage of the split-K products

// dW[co][ci] = sum_{b,p} dy[b][co][p] * LayerNorm(x[b])[ci][p] with the normalisation applied as x is staged: the weight
// gradient of a 1x1 convolution whose input was the WithBias LayerNorm of x (net/model.py:60-63, 192-196) without that
// input in memory.  1000 = shape not served (nothing launched; the caller materialises LayerNorm(x) and calls pir_gemm_nt).
extern "C" int pir_conv1x1_wgrad_ln(const float* dy, long dy_bs, const float* x, long x_bs, const float* mean, const float* rstd,
                                    const float* ln_w, const float* ln_b, float* dw, float* ws, size_t ws_floats,
                                    int B, int Cout, int Cin, int HW, pir_stream_t stream) {
  PIR_CHECK_ARG(dy && x && mean && rstd && ln_w && ln_b && dw && ws && B > 0 && Cout > 0 && Cin > 0 && HW > 0);
  if (Cout < Cin) return 1000;                       // the normalised operand must be the small (shared) one
  if ((reinterpret_cast<uintptr_t>(mean) & 15) || (reinterpret_cast<uintptr_t>(rstd) & 15)) return 1000;
  pir_gemm_nt_t g;
  g.X = dy; g.x_s1 = g.x_s2 = 0; g.x_sr = dy_bs; g.ldx = HW;
  g.Y = x; g.y_s1 = g.y_s2 = 0; g.y_sr = x_bs; g.ldy = HW;
  g.G = dw; g.g_so = 0; g.g_si = Cin; g.g_sj = 1;
  g.M1 = Cout; g.M2 = Cin; g.N = HW; g.O1 = 1; g.O2 = 1; g.BR = B;
  g.shift_dh = g.shift_dw = 0; g.H = 0; g.W = 0;
  g.ws = ws; g.ws_floats = ws_floats; g.alpha = 1.f; g.accumulate = 0;
  int splits = 0;
  const int st = pir_nt_xp_launch(&g, &splits, (hipStream_t)stream, mean, rstd, ln_w, ln_b);
  if (st) return st;
  return pir_nt_reduce_launch(ws, splits, Cout, Cin, dw, 0, Cin, 1, 1.f, 0, (hipStream_t)stream);
}
