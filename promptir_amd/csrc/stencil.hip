// Depthwise 3x3 stencils of the PromptIR path (gfx950), HBM-bound.
//
// One workgroup = one (plane, row-tile, column-tile).  The tile plus a one-pixel halo is
// staged in LDS with 16-byte coalesced loads (zero-filled outside the image = the conv's
// padding=1), then every thread slides a 3-row register window down a strip of SR rows and
// produces VEC(=4) horizontally adjacent outputs per row, stored as one 16-byte write.
// Algorithmic traffic: 1 read + 1 write of the plane; the halo re-reads are L2 hits.
//
// Modes:  FWD       y = dw3x3(x)                      (Attention.qkv_dwconv, net/model.py:112,120)
//         GATE_FWD  g = gelu(dw(x)[:hid]) * dw(x)[hid:] (FeedForward.dwconv + gate, :96-97)
//         GATE_BWD  dt from dg, recomputing t = dw(x)
//         WGRAD     dw[c][tap] partial sums           (weight gradient of either dwconv)
//         BWD       dx = dw3x3^T(dy) AND the dw partial sums in one pass over dy and x
#include "pir_common.h"

namespace {

enum { MODE_FWD = 0, MODE_GATE_FWD = 1, MODE_GATE_BWD = 2, MODE_WGRAD = 3, MODE_BWD = 4 };

struct DwArgs {
  const float* x; long x_bs;
  const float* w;
  float* y; long y_bs;       // FWD: y; GATE_FWD: g; GATE_BWD: dt
  const float* dz; long dz_bs;  // GATE_BWD: dg; WGRAD: dy
  float* ws;                 // WGRAD partials [part][C][9]
  int B, C, H, W;            // C = planes per image handled by the grid (hid for gate modes)
  int hid;                   // gate modes: channel offset of the second half
  int flip;
  int CT, LPR, strips, SR, tiles_r, tiles_c;
  unsigned magic_spr, magic_lpr;  // fast division by LPR+2 and LPR
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
  return cdf + v * pdf;
}

template <int VEC>
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ plane, int H, int W, int h0, int w0,
                                           int rows, int LPR, int LS, unsigned magic_spr) {
  const int slots_per_row = LPR + 2;
  const int total = rows * slots_per_row;
  for (int s = threadIdx.x; s < total; s += blockDim.x) {
    const int lr = pir_fastdiv(s, magic_spr), j = s - lr * slots_per_row;
    const int h = h0 - 1 + lr, col = w0 + (j - 1) * VEC;
    float* dst = lds + lr * LS + j * VEC;
    if (VEC == 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (h >= 0 && h < H && col >= 0 && col < W) v = *reinterpret_cast<const f32x4*>(plane + (long)h * W + col);
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
      dst[0] = (h >= 0 && h < H && col >= 0 && col < W) ? plane[(long)h * W + col] : 0.f;
    }
  }
}

template <int VEC>
__device__ __forceinline__ void read_row(const float* lds_row, int q, float (&v)[VEC + 2]) {
  const float* p = lds_row + (q + 1) * VEC;
  v[0] = p[-1];
  if (VEC == 4) {
    const f32x4 m = *reinterpret_cast<const f32x4*>(p);
    v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
  } else {
    v[1] = p[0];
  }
  v[VEC + 1] = p[VEC];
}

template <int VEC>
__device__ __forceinline__ void stencil_row(const float (&r0)[VEC + 2], const float (&r1)[VEC + 2],
                                            const float (&r2)[VEC + 2], const float (&k)[9], float (&out)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) s += k[d] * r0[j + d] + k[3 + d] * r1[j + d] + k[6 + d] * r2[j + d];
    out[j] = s;
  }
}

template <int VEC, int MODE>
__global__ void dwconv_kernel(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool GATE = (MODE == MODE_GATE_FWD || MODE == MODE_GATE_BWD);
  constexpr bool TWO = GATE || MODE == MODE_BWD;   // two staged tiles
  constexpr bool SUMS = MODE == MODE_WGRAD || MODE == MODE_BWD;
  const int LS = (a.LPR + 2) * VEC;
  const int RT = a.strips * a.SR;
  // FWD / gate modes: one workgroup per (plane, row tile, column tile).  Modes that produce weight-gradient
  // sums walk all row tiles of their (plane, column tile) so the sums are reduced once per workgroup.
  int bid = blockIdx.x;
  const int tile_c = bid % a.tiles_c; bid /= a.tiles_c;
  int tr_begin = 0, tr_end = a.tiles_r;
  if (!SUMS) { tr_begin = bid % a.tiles_r; tr_end = tr_begin + 1; bid /= a.tiles_r; }
  const int b = bid / a.C, c = bid % a.C;
  const int w0 = tile_c * a.CT;
  const long HW = (long)a.H * a.W;
  const float* __restrict__ xp = a.x + b * a.x_bs + c * HW;
  float* lds2 = lds + (RT + 2) * LS;

  float k1[9], k2[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    k1[t] = a.w[c * 9 + (a.flip ? 8 - t : t)];
    if (GATE) k2[t] = a.w[(c + a.hid) * 9 + t];
  }
  const int s = pir_fastdiv(threadIdx.x, a.magic_lpr), q = threadIdx.x - s * a.LPR;
  const int col = w0 + q * VEC;
  const bool active = s < a.strips && col < a.W;
  float wsum[9];
  if (SUMS) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wsum[t] = 0.f;
  }

  for (int tile_r = tr_begin; tile_r < tr_end; ++tile_r) {
  const int h0 = tile_r * RT;
  if (tile_r != tr_begin) __syncthreads();
  if (MODE == MODE_BWD) {  // tile 1 = dy (stencil input), tile 2 = x (weight-gradient operand)
    stage_tile<VEC>(lds, a.dz + b * a.dz_bs + c * HW, a.H, a.W, h0, w0, RT + 2, a.LPR, LS, a.magic_spr);
    stage_tile<VEC>(lds2, xp, a.H, a.W, h0, w0, RT + 2, a.LPR, LS, a.magic_spr);
  } else {
    stage_tile<VEC>(lds, xp, a.H, a.W, h0, w0, RT + 2, a.LPR, LS, a.magic_spr);
    if (GATE) stage_tile<VEC>(lds2, xp + a.hid * HW, a.H, a.W, h0, w0, RT + 2, a.LPR, LS, a.magic_spr);
  }
  __syncthreads();

  if (active) {
    float r0[VEC + 2], r1[VEC + 2], r2[VEC + 2];
    float u0[VEC + 2], u1[VEC + 2], u2[VEC + 2];
    const int lr0 = s * a.SR;
    read_row<VEC>(lds + lr0 * LS, q, r0);
    read_row<VEC>(lds + (lr0 + 1) * LS, q, r1);
    if (TWO) {
      read_row<VEC>(lds2 + lr0 * LS, q, u0);
      read_row<VEC>(lds2 + (lr0 + 1) * LS, q, u1);
    }
    for (int i = 0; i < a.SR; ++i) {
      const int h = h0 + lr0 + i;
      if (h >= a.H) break;
      read_row<VEC>(lds + (lr0 + i + 2) * LS, q, r2);
      if (TWO) read_row<VEC>(lds2 + (lr0 + i + 2) * LS, q, u2);
      const long off = (long)h * a.W + col;
      if (MODE == MODE_BWD) {
        float o[VEC];
        stencil_row<VEC>(r0, r1, r2, k1, o);   // k1 holds the flipped taps (a.flip = 1)
        float* yp = a.y + b * a.y_bs + c * HW + off;
        if (VEC == 4) { f32x4 v = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4*>(yp) = v; } else yp[0] = o[0];
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            wsum[d] += r1[j + 1] * u0[j + d];
            wsum[3 + d] += r1[j + 1] * u1[j + d];
            wsum[6 + d] += r1[j + 1] * u2[j + d];
          }
      } else if (MODE == MODE_FWD) {
        float o[VEC];
        stencil_row<VEC>(r0, r1, r2, k1, o);
        float* yp = a.y + b * a.y_bs + c * HW + off;
        if (VEC == 4) { f32x4 v = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4*>(yp) = v; } else yp[0] = o[0];
      } else if (MODE == MODE_GATE_FWD) {
        float t1[VEC], t2[VEC], o[VEC];
        stencil_row<VEC>(r0, r1, r2, k1, t1);
        stencil_row<VEC>(u0, u1, u2, k2, t2);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = gelu_erf(t1[j]) * t2[j];
        float* yp = a.y + b * a.y_bs + c * HW + off;
        if (VEC == 4) { f32x4 v = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4*>(yp) = v; } else yp[0] = o[0];
      } else if (MODE == MODE_GATE_BWD) {
        float t1[VEC], t2[VEC], d1[VEC], d2[VEC], dg[VEC];
        stencil_row<VEC>(r0, r1, r2, k1, t1);
        stencil_row<VEC>(u0, u1, u2, k2, t2);
        const float* gp = a.dz + b * a.dz_bs + c * HW + off;
        if (VEC == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(gp); dg[0] = v[0]; dg[1] = v[1]; dg[2] = v[2]; dg[3] = v[3]; }
        else dg[0] = gp[0];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float ge, gd;
          pir_gelu_both(t1[j], ge, gd);
          d1[j] = dg[j] * t2[j] * gd;
          d2[j] = dg[j] * ge;
        }
        float* p1 = a.y + b * a.y_bs + c * HW + off;
        float* p2 = p1 + a.hid * HW;
        if (VEC == 4) {
          f32x4 v1 = {d1[0], d1[1], d1[2], d1[3]}, v2 = {d2[0], d2[1], d2[2], d2[3]};
          *reinterpret_cast<f32x4*>(p1) = v1;
          *reinterpret_cast<f32x4*>(p2) = v2;
        } else { p1[0] = d1[0]; p2[0] = d2[0]; }
      } else {  // WGRAD
        float dy[VEC];
        const float* gp = a.dz + b * a.dz_bs + c * HW + off;
        if (VEC == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(gp); dy[0] = v[0]; dy[1] = v[1]; dy[2] = v[2]; dy[3] = v[3]; }
        else dy[0] = gp[0];
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            wsum[d] += dy[j] * r0[j + d];
            wsum[3 + d] += dy[j] * r1[j + d];
            wsum[6 + d] += dy[j] * r2[j + d];
          }
      }
#pragma unroll
      for (int j = 0; j < VEC + 2; ++j) { r0[j] = r1[j]; r1[j] = r2[j]; if (TWO) { u0[j] = u1[j]; u1[j] = u2[j]; } }
    }
  }
  }  // row tiles
  if (SUMS) {
    __shared__ float red[4 * 9];
    const float tot = pir_block_sum_many<9>(wsum, red);
    const int part = b * a.tiles_c + tile_c;
    if (threadIdx.x < 9) a.ws[((long)part * a.C + c) * 9 + threadIdx.x] = tot;
  }
}

struct DwPlan { int vec, CT, LPR, threads, strips, SR, tiles_r, tiles_c; size_t lds_bytes; };

DwPlan dw_plan(int H, int W, bool aligned, int ntiles_lds) {
  DwPlan p;
  p.vec = (W % 4 == 0 && aligned) ? 4 : 1;
  const int maxct = p.vec == 4 ? 256 : 128;
  p.CT = W < maxct ? W : maxct;
  p.LPR = (p.CT + p.vec - 1) / p.vec;
  long want = (long)H * p.LPR;
  p.threads = want >= 256 ? 256 : (int)(pir_cdiv(want, 64) * 64);
  if (p.threads < p.LPR) p.threads = (int)(pir_cdiv(p.LPR, 64) * 64);
  p.strips = p.threads / p.LPR;
  int sr = (int)pir_cdiv(H, p.strips);
  const int max_sr = ntiles_lds > 1 ? 4 : 8;  // keeps the dynamic LDS under 64 KiB
  p.SR = sr < max_sr ? sr : max_sr;
  p.tiles_r = (int)pir_cdiv(H, p.strips * p.SR);
  p.tiles_c = (int)pir_cdiv(W, p.CT);
  p.lds_bytes = (size_t)ntiles_lds * (p.strips * p.SR + 2) * (p.LPR + 2) * p.vec * sizeof(float);
  return p;
}

template <int MODE>
int launch_dw(DwArgs a, bool aligned, hipStream_t s) {
  constexpr bool GATE = (MODE == MODE_GATE_FWD || MODE == MODE_GATE_BWD || MODE == MODE_BWD);
  DwPlan p = dw_plan(a.H, a.W, aligned, GATE ? 2 : 1);
  a.magic_spr = pir_magic(p.LPR + 2); a.magic_lpr = pir_magic(p.LPR);
  a.CT = p.CT; a.LPR = p.LPR; a.strips = p.strips; a.SR = p.SR; a.tiles_r = p.tiles_r; a.tiles_c = p.tiles_c;
  constexpr bool SUMS = MODE == MODE_WGRAD || MODE == MODE_BWD;
  const long blocks = (long)a.B * a.C * (SUMS ? 1 : p.tiles_r) * p.tiles_c;
  if (blocks <= 0 || blocks > 2147483647L) return PIR_EINVAL;
  if (p.lds_bytes > 64 * 1024) return PIR_EINVAL;
  if (p.vec == 4)
    hipLaunchKernelGGL((dwconv_kernel<4, MODE>), dim3((unsigned)blocks), dim3(p.threads), p.lds_bytes, s, a);
  else
    hipLaunchKernelGGL((dwconv_kernel<1, MODE>), dim3((unsigned)blocks), dim3(p.threads), p.lds_bytes, s, a);
  return pir_launch_status();
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ------------------------------------------------------------------------------------ row reductions for the
// fallback of pir_dwconv3x3_sumsq: out[b][c] = sum_p x[b][c][p]^2, one workgroup per (image, channel)
__global__ __launch_bounds__(256) void plane_sumsq_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ out,
                                                          int C, int HW, int nsq) {
  __shared__ float red[16];
  const int b = blockIdx.x / nsq, c = blockIdx.x % nsq;
  const float* __restrict__ p = x + b * x_bs + (long)c * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) s += p[i] * p[i];
  const float t = pir_block_sum(s, red);
  if (threadIdx.x == 0) out[(long)b * nsq + c] = t;
}

// ------------------------------------------------------------------------------------ pixel (un)shuffle
// lo side: [B][4C][H][W]; hi side: [B][C][2H][2W].  One thread per hi-side float4 (4 columns).
template <bool TO_LO>
__global__ void pixel_shuffle_kernel(const float* __restrict__ src, long s_bs, float* __restrict__ dst, long d_bs,
                                     int B, int C, int H, int W) {
  const long total = (long)B * C * 2 * H * W;  // hi-side pairs of columns (2 hi pixels = 1 lo column)
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int w = (int)(e % W);
    long t = e / W;
    const int hh = (int)(t % (2 * H)); t /= (2 * H);
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const int h = hh >> 1, i = hh & 1;
    const long hi = (long)c * 4 * H * W + (long)hh * 2 * W + 2 * w;             // hi[b][c][hh][2w + j]
    const long lo0 = ((long)(c * 4 + i * 2 + 0) * H + h) * W + w;               // lo[b][c*4+i*2+j][h][w]
    const long lo1 = ((long)(c * 4 + i * 2 + 1) * H + h) * W + w;
    if (TO_LO) {
      const float2 v = *reinterpret_cast<const float2*>(src + b * s_bs + hi);
      dst[b * d_bs + lo0] = v.x;
      dst[b * d_bs + lo1] = v.y;
    } else {
      float2 v;
      v.x = src[b * s_bs + lo0];
      v.y = src[b * s_bs + lo1];
      *reinterpret_cast<float2*>(dst + b * d_bs + hi) = v;
    }
  }
}

}  // namespace

// register-only kernels for power-of-two widths (stencil_wave.hip); 1000 = shape not served, use the LDS-tiled kernels
int pir_sw_try_fwd(const float* x, long x_bs, const float* w, int flip, float* y, long y_bs, float* sq_parts, int nsq,
                   int* nparts, int B, int C, int H, int W, hipStream_t s);
int pir_sw_try_gate(const float* x, long x_bs, const float* w, float* g, long g_bs, int B, int hid, int H, int W, hipStream_t s);
int pir_sw_try_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* w, float* dx, long dx_bs,
                   float* ws, size_t ws_floats, int B, int C, int H, int W, int* parts_out, hipStream_t s);
size_t pir_sw_bwd_ws_floats(int B, int C, int H);

extern "C" int pir_dwconv3x3(const float* x, long x_bs, const float* w, int flip, float* y, long y_bs,
                             int B, int C, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(x && w && y && B > 0 && C > 0 && H > 0 && W > 0);
  const int st = pir_sw_try_fwd(x, x_bs, w, flip, y, y_bs, nullptr, 0, nullptr, B, C, H, W, (hipStream_t)stream);
  if (st != 1000) return st;
  DwArgs a = {};
  a.x = x; a.x_bs = x_bs; a.w = w; a.y = y; a.y_bs = y_bs; a.B = B; a.C = C; a.H = H; a.W = W; a.flip = flip;
  const bool aligned = al16(x) && al16(y) && x_bs % 4 == 0 && y_bs % 4 == 0;
  return launch_dw<MODE_FWD>(a, aligned, (hipStream_t)stream);
}

extern "C" size_t pir_dwconv3x3_sumsq_floats(int B, int nsq, int H) {
  if (B <= 0 || nsq <= 0 || H <= 0) return 0;
  return (size_t)B * pir_cdiv(H, 8 < H ? 8 : H) * nsq;
}

extern "C" int pir_dwconv3x3_sumsq(const float* x, long x_bs, const float* w, float* y, long y_bs, float* sq_parts,
                                   size_t sq_floats, int nsq, int* nparts, int B, int C, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(x && w && y && sq_parts && nparts && B > 0 && C > 0 && H > 0 && W > 0 && nsq > 0 && nsq <= C);
  if (sq_floats < pir_dwconv3x3_sumsq_floats(B, nsq, H)) return PIR_ENOMEM;
  int st = pir_sw_try_fwd(x, x_bs, w, 0, y, y_bs, sq_parts, nsq, nparts, B, C, H, W, (hipStream_t)stream);
  if (st != 1000) return st;
  st = pir_dwconv3x3(x, x_bs, w, 0, y, y_bs, B, C, H, W, stream);   // LDS-tiled stencil, then one pass over q and k
  if (st) return st;
  *nparts = 1;
  hipLaunchKernelGGL(plane_sumsq_kernel, dim3((unsigned)((long)B * nsq)), dim3(256), 0, (hipStream_t)stream, y, y_bs,
                     sq_parts, C, H * W, nsq);
  return pir_launch_status();
}

extern "C" int pir_dwconv3x3_gate(const float* x, long x_bs, const float* w, float* g, long g_bs,
                                  int B, int hid, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(x && w && g && B > 0 && hid > 0 && H > 0 && W > 0);
  const int sw = pir_sw_try_gate(x, x_bs, w, g, g_bs, B, hid, H, W, (hipStream_t)stream);
  if (sw != 1000) return sw;
  DwArgs a = {};
  a.x = x; a.x_bs = x_bs; a.w = w; a.y = g; a.y_bs = g_bs; a.B = B; a.C = hid; a.hid = hid; a.H = H; a.W = W;
  const bool aligned = al16(x) && al16(g) && x_bs % 4 == 0 && g_bs % 4 == 0;
  return launch_dw<MODE_GATE_FWD>(a, aligned, (hipStream_t)stream);
}

extern "C" int pir_dwconv3x3_gate_bwd(const float* x, long x_bs, const float* w, const float* dg, long dg_bs,
                                      float* dt, long dt_bs, int B, int hid, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(x && w && dg && dt && B > 0 && hid > 0 && H > 0 && W > 0);
  DwArgs a = {};
  a.x = x; a.x_bs = x_bs; a.w = w; a.y = dt; a.y_bs = dt_bs; a.dz = dg; a.dz_bs = dg_bs;
  a.B = B; a.C = hid; a.hid = hid; a.H = H; a.W = W;
  const bool aligned = al16(x) && al16(dg) && al16(dt) && x_bs % 4 == 0 && dg_bs % 4 == 0 && dt_bs % 4 == 0;
  return launch_dw<MODE_GATE_BWD>(a, aligned, (hipStream_t)stream);
}

extern "C" size_t pir_dwconv3x3_wgrad_ws_floats(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  // worst case over the two vector widths
  DwPlan p4 = dw_plan(H, W, W % 4 == 0, 1), p1 = dw_plan(H, W, false, 1);
  const size_t t4 = (size_t)p4.tiles_c, t1 = (size_t)p1.tiles_c;
  return (size_t)B * (t4 > t1 ? t4 : t1) * C * 9;
}

extern "C" int pir_reduce_partials(const float* parts, long stride, int S, float alpha, int accumulate,
                                   float* out, long count, pir_stream_t stream);

extern "C" int pir_dwconv3x3_wgrad(const float* dy, long dy_bs, const float* x, long x_bs, float* dw,
                                   float* ws, size_t ws_floats, int B, int C, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(dy && x && dw && ws && B > 0 && C > 0 && H > 0 && W > 0);
  DwArgs a = {};
  a.x = x; a.x_bs = x_bs; a.w = dw /*unused taps read; any valid [C][9] buffer*/; a.dz = dy; a.dz_bs = dy_bs; a.ws = ws;
  a.B = B; a.C = C; a.H = H; a.W = W;
  const bool aligned = al16(x) && al16(dy) && x_bs % 4 == 0 && dy_bs % 4 == 0;
  DwPlan p = dw_plan(H, W, aligned && W % 4 == 0, 1);
  const long parts = (long)B * p.tiles_c;
  if ((size_t)parts * C * 9 > ws_floats) return PIR_ENOMEM;
  int st = launch_dw<MODE_WGRAD>(a, aligned, (hipStream_t)stream);
  if (st) return st;
  return pir_reduce_partials(ws, (long)C * 9, (int)parts, 1.f, 0, dw, (long)C * 9, stream);
}

// dx = dw3x3^T(dy) and dw in ONE pass over dy and x (2 reads + 1 write instead of 3 + 1)
extern "C" size_t pir_dwconv3x3_bwd_ws_floats(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  DwPlan p4 = dw_plan(H, W, W % 4 == 0, 2), p1 = dw_plan(H, W, false, 2);
  const size_t t4 = (size_t)p4.tiles_c, t1 = (size_t)p1.tiles_c;
  const size_t lds = (size_t)B * (t4 > t1 ? t4 : t1) * C * 9, wave = pir_sw_bwd_ws_floats(B, C, H);
  return lds > wave ? lds : wave;
}

extern "C" int pir_dwconv3x3_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* w,
                                 float* dx, long dx_bs, float* dw, float* ws, size_t ws_floats,
                                 int B, int C, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(dy && x && w && dx && dw && ws && B > 0 && C > 0 && H > 0 && W > 0);
  {
    int wparts = 0;
    const int sw = pir_sw_try_bwd(dy, dy_bs, x, x_bs, w, dx, dx_bs, ws, ws_floats, B, C, H, W, &wparts, (hipStream_t)stream);
    if (sw == 0) return pir_reduce_partials(ws, (long)C * 9, wparts, 1.f, 0, dw, (long)C * 9, stream);
    if (sw != 1000) return sw;
  }
  DwArgs a = {};
  a.x = x; a.x_bs = x_bs; a.w = w; a.flip = 1; a.y = dx; a.y_bs = dx_bs; a.dz = dy; a.dz_bs = dy_bs; a.ws = ws;
  a.B = B; a.C = C; a.H = H; a.W = W;
  const bool aligned = al16(x) && al16(dy) && al16(dx) && x_bs % 4 == 0 && dy_bs % 4 == 0 && dx_bs % 4 == 0;
  DwPlan p = dw_plan(H, W, aligned && W % 4 == 0, 2);
  const long parts = (long)B * p.tiles_c;
  if ((size_t)parts * C * 9 > ws_floats) return PIR_ENOMEM;
  int st = launch_dw<MODE_BWD>(a, aligned, (hipStream_t)stream);
  if (st) return st;
  return pir_reduce_partials(ws, (long)C * 9, (int)parts, 1.f, 0, dw, (long)C * 9, stream);
}

extern "C" int pir_pixel_unshuffle2(const float* x, long x_bs, float* y, long y_bs, int B, int C, int H, int W,
                                    pir_stream_t stream) {
  PIR_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && x_bs % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0);
  const long total = (long)B * C * 2 * H * W;
  const int blocks = (int)(pir_cdiv(total, 256) < 4096 ? pir_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL((pixel_shuffle_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_bs, y, y_bs, B, C, H, W);
  return pir_launch_status();
}

extern "C" int pir_pixel_shuffle2(const float* x, long x_bs, float* y, long y_bs, int B, int C, int H, int W,
                                  pir_stream_t stream) {
  PIR_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && y_bs % 2 == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0);
  const long total = (long)B * C * 2 * H * W;
  const int blocks = (int)(pir_cdiv(total, 256) < 4096 ? pir_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL((pixel_shuffle_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_bs, y, y_bs, B, C, H, W);
  return pir_launch_status();
}
