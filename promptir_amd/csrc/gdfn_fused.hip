// GDFN forward without its 2 x hid-channel intermediate in memory (gfx950):
//
//     g = gelu_erf(dw3x3(W_in LN(x))[:hid]) * dw3x3(W_in LN(x))[hid:]          net/model.py:94-97 behind :195
//
// The unfused forward writes h0 = W_in LN(x) (2 hid = 510 planes at C = 96) and reads it back for the depthwise stencil:
// 1020 of the block's ~3260 forward floats per pixel.  Here h0 only ever exists as MFMA accumulators:
//
//  pass 1, pir-internal (ln_split_rows_kernel): the channel LayerNorm (:60-63) of every pixel, split ONCE into the
//      three bf16 pieces and stored as ready-made MFMA fragments (6 bytes per element; one 16-byte unit = 8 channels
//      of one pixel), laid out so that a wave's fragment of (image row, 32-pixel block, k-step, piece) is 1 KB contiguous;
//  pass 2 (gdfn_fused_kernel): a workgroup = one image x 32 gate pairs (64 rows of W_in: 32 of the first half and their
//      32 partners of the second half, resident in LDS as bf16x3), NWV waves side by side covering the image row (one
//      32-pixel block each), walking DOWN the rows.  Per row a wave multiplies its pixel fragments with the 64 weight
//      rows - as D = pixels x channels, so that a lane ends up with ONE channel (lane & 31) of each half and 16 pixels
//      of it (4 runs of 4 consecutive pixels): the depthwise taps are then 9 + 9 per-lane registers, horizontal
//      neighbours are mostly the lane's own registers (run ends: the partner lane half; block ends: the neighbouring
//      wave through a 1 KB LDS exchange), and vertical neighbours never exist at once - each arriving row adds its three
//      horizontally filtered contributions into the two pending output rows (out[r-1] += w[2] * row r, out[r] += w[1],
//      out[r+1] = w[0]).  A finished row goes through the gate (libm erff, as the unfused forward) and leaves as
//      16-byte stores.  Every input row's product is computed once: no halo recompute.
// The normalised, split activations are re-read by the ceil(hid / 32) workgroups of an image (from L2); h0 is never
// written or read.  Served: W = 128 or 64 (4 or 2 waves per row), C = 48 or 96, bias-free, WithBias LayerNorm.
#include "gemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
// pass 1: LayerNorm + bf16x3 split into fragment order.  One thread per pixel; the C channel values live in registers.
// unit index of (image b, row y, block blk, k-step ks, piece, lane): ((((b H + y) NB + blk) KS + ks) 3 + piece) 64 + lane
template <int C>
__global__ __launch_bounds__(256) void ln_split_rows_kernel(const float* __restrict__ x, long x_bs, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16x8* __restrict__ out,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int H, int W, int B) {
  constexpr int KS = C / 16;
  const long HW = (long)H * W;
  const long pix = blockIdx.x * 256L + threadIdx.x;
  if (pix >= (long)B * HW) return;
  const int b = (int)(pix / HW);
  const long p = pix - (long)b * HW;
  const int y = (int)(p / W), xx = (int)(p - (long)y * W);
  const float* __restrict__ src = x + (long)b * x_bs + p;
  float v[C];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { v[c] = src[(long)c * HW]; s += v[c]; }
  const float mu = s / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { v[c] -= mu; s2 += v[c] * v[c]; }
  const float rstd = 1.f / sqrtf(s2 / (float)C + 1e-5f);      // biased variance, eps inside the root (net/model.py:62-63)
  if (mean_out) { mean_out[pix] = mu; rstd_out[pix] = rstd; }
  const int NB = W / 32, blk = xx / 32, r = xx & 31;
  bf16x8* __restrict__ dst = out + ((((long)b * H + y) * NB + blk) * KS) * 3 * 64;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 16 * ks + 8 * h + e;
        f[e] = v[c] * rstd * gamma[c] + beta[c];
      }
      const pir_frag3 fr = pir_split8(f);
      bf16x8* u = dst + (long)ks * 3 * 64 + 32 * h + r;
      u[0] = fr.hi; u[64] = fr.mid; u[128] = fr.lo;
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct FusedArgs {
  const bf16x8* xn3;          // pass-1 fragments
  const bf16x8* w3;           // pir_split_bf16x3 pieces of W_in: unit ((part KS + ks) M + m) 2 + h, M = 2 hid
  const float* wd;            // depthwise weights [2 hid][9]
  float* g; long g_bs;        // [B][hid][H][W]
  int hid, H, W;
  int band;                   // output rows per workgroup; each band re-multiplies one input row above and below
  int nchunks, nbands, B;     // ceil(hid / 32), bands per image, images
  float* dump;                // 16 KB the pipelined kernel's non-storing lanes write to (keeps its row step branch-free)
};

// FAST = false: libm erff, the arithmetic of the unfused forward (pir_dwconv3x3_gate); FAST = true: Abramowitz & Stegun
// 7.1.26 (|error| <= 1.5e-7, what the backward kernels use; checked against fp64 in tests) - a third of the instructions
template <bool FAST>
__device__ __forceinline__ float gelu_erf_f(float v) {
  if (!FAST) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  const float ax = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float e = __expf(-ax * ax);
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erf_abs = fmaf(-p * t, e, 1.f);
  return v * fmaf(0.5f, copysignf(erf_abs, v), 0.5f);
}

// NWV = W / 32 waves per workgroup, KS = C / 16; two waves per SIMD (one workgroup's MFMAs run under another's filter /
// gate arithmetic: a row costs 72 MFMAs and ~1200 vector instructions, serial inside one wave)
template <int KS, int NWV, bool FAST>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(2)))
void gdfn_fused_kernel(FusedArgs a) {
  constexpr int T = NWV * 64;
  constexpr int PUNITS = 3 * KS * 2 * 64;                 // weight panel: [piece][ks][h][64 rows]
  __shared__ bf16x8 panel[PUNITS];
  __shared__ float edge[2][2][2][NWV][32];                // [row parity][half nb][side: 0 = first pixel, 1 = last pixel][wave][channel]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, cl = lane & 31;
  // XCD-aware order (workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MB L2): the nchunks
  // workgroups that read the SAME fragments - one (image, band) unit - get linear ids of equal residue mod 8 and
  // consecutive rank inside it, so the unit's 1.3 MB of fragments come from HBM once per XCD and from that L2 otherwise.
  // (In (chunk, image, band) grid order the chunks of a unit landed on eight different XCDs and every one of them
  // streamed the whole tensor: the first versions of this kernel were bound by exactly that.)
  const int lin = (int)blockIdx.x, xcd = lin & 7, t = lin >> 3;
  const int q = t % a.nchunks, unit = (t / a.nchunks) * 8 + xcd;
  if (unit >= a.B * a.nbands) return;
  const int b = unit / a.nbands, band_i = unit - b * a.nbands;
  const int hid = a.hid, H = a.H, W = a.W, M = 2 * hid;
  const int npairs = hid - 32 * q < 32 ? hid - 32 * q : 32;

  // ---- weight panel: rows 0..31 = channels 32 q + i of the first half, rows 32..63 = their partners hid + 32 q + i
  for (int u = tid; u < PUNITS; u += T) {
    const int row = u & 63, hh = (u >> 6) & 1, rest = u >> 7;      // rest = piece * KS + ks
    const int i = row & 31;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
    if (i < npairs) {
      const int m = (row < 32 ? 0 : hid) + 32 * q + i;
      v = a.w3[((long)rest * M + m) * 2 + hh];
    }
    panel[u] = v;
  }
  // ---- this lane's channel pair and its depthwise taps
  const bool live = cl < npairs;
  const int c1 = 32 * q + (live ? cl : 0), c2 = hid + c1;
  float t1[9], t2[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) { t1[e] = live ? a.wd[c1 * 9 + e] : 0.f; t2[e] = live ? a.wd[c2 * 9 + e] : 0.f; }
  __syncthreads();

  const bf16x8* __restrict__ xbase = a.xn3 + ((long)b * H * NWV + wid) * KS * 3 * 64 + lane;
  const long row_units = (long)NWV * KS * 3 * 64;
  float* __restrict__ gout = a.g + (long)b * a.g_bs + (long)c1 * H * W + 32 * wid + 4 * h;

  // ---- this workgroup's band of output rows [r0, r1); input rows r0 - 1 .. r1 (clipped to the image)
  const int r0 = band_i * a.band, r1 = r0 + a.band < H ? r0 + a.band : H;
  const int rs = r0 > 0 ? r0 - 1 : 0;
  const int rend = r1 == H ? H : r1;                         // last iteration: input row r1, or the flush (no input) at the image's end

  // A row's fragments arrive in two halves: k-steps [0, KSL) are fetched a whole filter / gate phase ahead, k-steps
  // [KSL, KS) at the start of the row's own MFMA phase (their latency hides under the first half's MFMAs) - they are
  // not live during the filter phase, where the register pressure peaks.
  constexpr int KSL = KS / 2;
  bf16x8 cur[KS][3];
  auto load_part = [&](int r, int k0, int k1) {
    const bf16x8* __restrict__ p = xbase + (long)r * row_units;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      if (ks >= k0 && ks < k1) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) cur[ks][pc] = p[(ks * 3 + pc) * 64];
      }
  };
  load_part(rs, 0, KSL);

  // Pending output rows in registers.  Input row r adds its three horizontally filtered contributions: taps row 2 completes
  // out[r - 1] (held in P), taps row 1 goes into out[r] (held in Q), taps row 0 STARTS out[r + 1] - written into P's
  // registers once the finished row has left them.  P and Q swap roles every row (the loop is unrolled by two): no moves.
  float accA[2][16], accB[2][16];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int p = 0; p < 16; ++p) { accA[nb][p] = 0.f; accB[nb][p] = 0.f; }
  const bool hb = h != 0;

  auto row_step = [&](int r, float (&P)[2][16], float (&Q)[2][16]) {
    const bool have = r < H;                                 // row H is the flush iteration: no input, out[H - 1] leaves
    f32x16 d[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int p = 0; p < 16; ++p) d[nb][p] = 0.f;
    if (have) {
      load_part(r, KSL, KS);
      const bf16x8* wp = panel + h * 64 + cl + (r >> 30);    // opaque zero: keeps the panel reads inside the loop
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 bw[2][3];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) bw[nb][pc] = wp[((pc * KS + ks) * 2) * 64 + nb * 32];
        // the two halves' chains interleaved term by term (same per-accumulator term order as pir_mfma_x3)
#define PIR_GF_TERM(A_, B_) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
        d[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[ks][A_], bw[nb][B_], d[nb], 0, 0, 0);
        PIR_GF_TERM(2, 0) PIR_GF_TERM(0, 2) PIR_GF_TERM(1, 1) PIR_GF_TERM(1, 0) PIR_GF_TERM(0, 1) PIR_GF_TERM(0, 0)
#undef PIR_GF_TERM
      }
    }
    // the fragments are consumed: the first half of the next row's flies during the filter / gate phase below
    if (r + 1 <= rend && r + 1 < H) load_part(r + 1, 0, KSL);
    // ---- block-edge pixels of the new row for the neighbouring waves (pixel 0: lane half 0, register 0; pixel 31: half 1, register 15)
    const int par = r & 1;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) edge[par][nb][h][wid][cl] = hb ? d[nb][15] : d[nb][0];
    __syncthreads();
    // ---- horizontal neighbours of the run ends: partner lane half, block ends from the neighbouring waves (zero at the border)
    float lft[2][4], rgt[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float rf[4], rl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rf[j] = __shfl_xor(d[nb][4 * j], 32, 64);
        rl[j] = __shfl_xor(d[nb][4 * j + 3], 32, 64);
      }
      const float eL = wid > 0 ? edge[par][nb][1][wid > 0 ? wid - 1 : 0][cl] : 0.f;                 // pixel -1 (lane half 0 uses it)
      const float eR = wid + 1 < NWV ? edge[par][nb][0][wid + 1 < NWV ? wid + 1 : wid][cl] : 0.f;    // pixel 32 (lane half 1)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lft[nb][j] = hb ? rl[j] : (j > 0 ? rl[j > 0 ? j - 1 : 0] : eL);
        rgt[nb][j] = hb ? (j < 3 ? rf[j < 3 ? j + 1 : 3] : eR) : rf[j];
      }
    }
    const bool store = r - 1 >= r0 && r - 1 < r1 && live;
    float* __restrict__ dst = gout + (long)(r - 1) * W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // the four pixels of run j as two pairs; cross-correlation: input row r feeds out[r - 1] through taps row 2, out[r]
      // through row 1, out[r + 1] through row 0
      f32x2 l2[2][2], m2[2][2], r2[2][2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float a0 = d[nb][4 * j], a1 = d[nb][4 * j + 1], a2 = d[nb][4 * j + 2], a3 = d[nb][4 * j + 3];
        l2[nb][0] = f32x2{lft[nb][j], a0}; m2[nb][0] = f32x2{a0, a1}; r2[nb][0] = f32x2{a1, a2};
        l2[nb][1] = f32x2{a1, a2};         m2[nb][1] = f32x2{a2, a3}; r2[nb][1] = f32x2{a3, rgt[nb][j]};
      }
      f32x2 fin[2][2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float (&tp)[9] = nb == 0 ? t1 : t2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int p = 4 * j + 2 * k;
          f32x2 o = f32x2{P[nb][p], P[nb][p + 1]};
          o = __builtin_elementwise_fma(f32x2{tp[6], tp[6]}, l2[nb][k], o);
          o = __builtin_elementwise_fma(f32x2{tp[7], tp[7]}, m2[nb][k], o);
          o = __builtin_elementwise_fma(f32x2{tp[8], tp[8]}, r2[nb][k], o);
          fin[nb][k] = o;
          f32x2 q2 = f32x2{Q[nb][p], Q[nb][p + 1]};
          q2 = __builtin_elementwise_fma(f32x2{tp[3], tp[3]}, l2[nb][k], q2);
          q2 = __builtin_elementwise_fma(f32x2{tp[4], tp[4]}, m2[nb][k], q2);
          q2 = __builtin_elementwise_fma(f32x2{tp[5], tp[5]}, r2[nb][k], q2);
          Q[nb][p] = q2[0]; Q[nb][p + 1] = q2[1];
          f32x2 n2 = f32x2{tp[0], tp[0]} * l2[nb][k];
          n2 = __builtin_elementwise_fma(f32x2{tp[1], tp[1]}, m2[nb][k], n2);
          n2 = __builtin_elementwise_fma(f32x2{tp[2], tp[2]}, r2[nb][k], n2);
          P[nb][p] = n2[0]; P[nb][p + 1] = n2[1];          // out[r + 1] starts in the registers out[r - 1] just left
        }
      }
      if (store) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f<FAST>(fin[0][e >> 1][e & 1]) * fin[1][e >> 1][e & 1];
        *reinterpret_cast<f32x4*>(dst + 8 * j) = v;          // pixels 32 wid + 8 j + 4 h .. + 3
      }
    }
  };

  int r = rs;
  for (; r + 1 <= rend; r += 2) {
    row_step(r, accA, accB);
    row_step(r + 1, accB, accA);
  }
  if (r <= rend) row_step(r, accA, accB);
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined form (knob 37): ONE wave per SIMD with the whole register file.  The product of row r + 1 is issued INSIDE the
// filter / gate arithmetic of row r (same basic block, interleaved by scheduling groups: one MFMA per ~8 vector
// instructions), the fragments of row r + 2 are in flight meanwhile: nothing waits on memory or on the matrix pipe in
// steady state.  Fragment sets, products and pending rows swap roles every row (loop unrolled by two).
template <int KS, int NWV, bool FAST>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(1)))
void gdfn_fused_pipe_kernel(FusedArgs a) {
  constexpr int T = NWV * 64;
  constexpr int PUNITS = 3 * KS * 2 * 64;
  __shared__ bf16x8 panel[PUNITS];
  __shared__ float edge[2][2][2][NWV][32];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, cl = lane & 31;
  const int lin = (int)blockIdx.x, xcd = lin & 7, t = lin >> 3;
  const int q = t % a.nchunks, unit = (t / a.nchunks) * 8 + xcd;
  if (unit >= a.B * a.nbands) return;
  const int b = unit / a.nbands, band_i = unit - b * a.nbands;
  const int hid = a.hid, H = a.H, W = a.W, M = 2 * hid;
  const int npairs = hid - 32 * q < 32 ? hid - 32 * q : 32;
  for (int u = tid; u < PUNITS; u += T) {
    const int row = u & 63, hh = (u >> 6) & 1, rest = u >> 7;
    const int i = row & 31;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
    if (i < npairs) {
      const int m = (row < 32 ? 0 : hid) + 32 * q + i;
      v = a.w3[((long)rest * M + m) * 2 + hh];
    }
    panel[u] = v;
  }
  const bool live = cl < npairs;
  const int c1 = 32 * q + (live ? cl : 0), c2 = hid + c1;
  float t1[9], t2[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) { t1[e] = live ? a.wd[c1 * 9 + e] : 0.f; t2[e] = live ? a.wd[c2 * 9 + e] : 0.f; }
  __syncthreads();

  const bf16x8* __restrict__ xbase = a.xn3 + ((long)b * H * NWV + wid) * KS * 3 * 64 + lane;
  const long row_units = (long)NWV * KS * 3 * 64;
  float* __restrict__ gout = a.g + (long)b * a.g_bs + (long)c1 * H * W + 32 * wid + 4 * h;
  const int r0 = band_i * a.band, r1 = r0 + a.band < H ? r0 + a.band : H;
  const int rs = r0 > 0 ? r0 - 1 : 0;
  const int rend = r1 == H ? H : r1;
  const int rlast = rend < H ? rend : H - 1;                 // last input row of the band
  const bool hb = h != 0;
  const float maskL = wid > 0 ? 1.f : 0.f, maskR = wid + 1 < NWV ? 1.f : 0.f;     // zero padding at the image's left / right border

  auto load_row = [&](int r, bf16x8 (&f)[KS][3]) {
    const bf16x8* __restrict__ p = xbase + (long)r * row_units;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) f[ks][pc] = p[(ks * 3 + pc) * 64];
  };
  auto product = [&](const bf16x8 (&f)[KS][3], f32x16 (&d)[2], int opaque) {
    const bf16x8* wp = panel + h * 64 + cl + (opaque >> 30);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int p = 0; p < 16; ++p) d[nb][p] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 bw[2][3];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bw[nb][pc] = wp[((pc * KS + ks) * 2) * 64 + nb * 32];
#define PIR_GF_TERM(A_, B_) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
      d[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][A_], bw[nb][B_], d[nb], 0, 0, 0);
      PIR_GF_TERM(2, 0) PIR_GF_TERM(0, 2) PIR_GF_TERM(1, 1) PIR_GF_TERM(1, 0) PIR_GF_TERM(0, 1) PIR_GF_TERM(0, 0)
#undef PIR_GF_TERM
    }
  };

  bf16x8 fa[KS][3], fb[KS][3];
  f32x16 da[2], db[2];
  float accA[2][16], accB[2][16];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int p = 0; p < 16; ++p) { accA[nb][p] = 0.f; accB[nb][p] = 0.f; }
  load_row(rs, fa);
  load_row(rs + 1 <= rlast ? rs + 1 : rs, fb);
  product(fa, da, rs);                                       // the band's first row: nothing to hide it under

  // D: product of row r (zeros in the flush iteration); DN <- product of row r + 1 from FN; FL <- fragments of row r + 2
  auto step = [&](int r, f32x16 (&D)[2], f32x16 (&DN)[2], const bf16x8 (&FN)[KS][3], bf16x8 (&FL)[KS][3],
                  float (&P)[2][16], float (&Q)[2][16]) {
    const int par = r & 1;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) edge[par][nb][h][wid][cl] = hb ? D[nb][15] : D[nb][0];
    __syncthreads();
    load_row(r + 2 <= rlast ? r + 2 : rlast, FL);            // (past the band: the last row again, unused)
    __builtin_amdgcn_sched_barrier(0);
    product(FN, DN, r);                                      // row r + 1 (past the band: discarded below)
    float lft[2][4], rgt[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float rf[4], rl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rf[j] = __shfl_xor(D[nb][4 * j], 32, 64);
        rl[j] = __shfl_xor(D[nb][4 * j + 3], 32, 64);
      }
      // (multiplied by a 0 / 1 factor instead of selected: a wave-uniform condition would become a scalar BRANCH around the
      // load and cut the row step into several scheduling regions - the MFMAs could then not be interleaved with the filter)
      const float eL = edge[par][nb][1][wid > 0 ? wid - 1 : 0][cl] * maskL;
      const float eR = edge[par][nb][0][wid + 1 < NWV ? wid + 1 : wid][cl] * maskR;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lft[nb][j] = hb ? rl[j] : (j > 0 ? rl[j > 0 ? j - 1 : 0] : eL);
        rgt[nb][j] = hb ? (j < 3 ? rf[j < 3 ? j + 1 : 3] : eR) : rf[j];
      }
    }
    const bool store = r - 1 >= r0 && r - 1 < r1 && live;
    // lanes that must not store (rows outside the band, lanes beyond the last gate pair) write into a dump area instead: no
    // branch inside the interleaved region
    float* __restrict__ dst = store ? gout + (long)(r - 1) * W : a.dump + tid * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 l2[2][2], m2[2][2], r2[2][2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float a0 = D[nb][4 * j], a1 = D[nb][4 * j + 1], a2 = D[nb][4 * j + 2], a3 = D[nb][4 * j + 3];
        l2[nb][0] = f32x2{lft[nb][j], a0}; m2[nb][0] = f32x2{a0, a1}; r2[nb][0] = f32x2{a1, a2};
        l2[nb][1] = f32x2{a1, a2};         m2[nb][1] = f32x2{a2, a3}; r2[nb][1] = f32x2{a3, rgt[nb][j]};
      }
      f32x2 fin[2][2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float (&tp)[9] = nb == 0 ? t1 : t2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int p = 4 * j + 2 * k;
          f32x2 o = f32x2{P[nb][p], P[nb][p + 1]};
          o = __builtin_elementwise_fma(f32x2{tp[6], tp[6]}, l2[nb][k], o);
          o = __builtin_elementwise_fma(f32x2{tp[7], tp[7]}, m2[nb][k], o);
          o = __builtin_elementwise_fma(f32x2{tp[8], tp[8]}, r2[nb][k], o);
          fin[nb][k] = o;
          f32x2 q2 = f32x2{Q[nb][p], Q[nb][p + 1]};
          q2 = __builtin_elementwise_fma(f32x2{tp[3], tp[3]}, l2[nb][k], q2);
          q2 = __builtin_elementwise_fma(f32x2{tp[4], tp[4]}, m2[nb][k], q2);
          q2 = __builtin_elementwise_fma(f32x2{tp[5], tp[5]}, r2[nb][k], q2);
          Q[nb][p] = q2[0]; Q[nb][p + 1] = q2[1];
          f32x2 n2 = f32x2{tp[0], tp[0]} * l2[nb][k];
          n2 = __builtin_elementwise_fma(f32x2{tp[1], tp[1]}, m2[nb][k], n2);
          n2 = __builtin_elementwise_fma(f32x2{tp[2], tp[2]}, r2[nb][k], n2);
          P[nb][p] = n2[0]; P[nb][p + 1] = n2[1];
        }
      }
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f<FAST>(fin[0][e >> 1][e & 1]) * fin[1][e >> 1][e & 1];
      *reinterpret_cast<f32x4*>(dst + (store ? 8 * j : 4 * j)) = v;
    }
    // one MFMA per ~8 vector instructions, the panel reads ahead of their MFMAs
#pragma unroll
    for (int m = 0; m < 6 * KS; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, FAST ? 7 : 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, FAST ? 7 : 12, 0);
    }
    if (!(r + 1 <= rlast)) {                                 // no row r + 1 in this band: the flush iteration sees zeros
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int p = 0; p < 16; ++p) DN[nb][p] = 0.f;
    }
  };

  int r = rs;
  for (; r + 1 <= rend; r += 2) {
    step(r, da, db, fb, fa, accA, accB);
    step(r + 1, db, da, fa, fb, accB, accA);
  }
  if (r <= rend) step(r, da, db, fb, fa, accA, accB);
}

int g_fused_fast_erf = 0;   // knob 35: 1 = the backward kernels' erf approximation in the fused forward gate (default: libm erff)
int g_fused_pipe = 0;       // knob 37: 1 = the pipelined one-wave-per-SIMD kernel

}  // namespace

int pir_gdfn_fused_tune(int knob, int value) {
  if (knob == 35) { g_fused_fast_erf = value; return PIR_OK; }
  if (knob == 37) { g_fused_pipe = value; return PIR_OK; }
  return PIR_EINVAL;
}

extern "C" size_t pir_gdfn_fused_ws_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * H * W * C * 6 + 16384;      // fragments + the pipelined kernel's dump area
}

// g = gelu(dw3x3(W_in LN(x))[:hid]) * dw3x3(W_in LN(x))[hid:] with h0 never in memory.  w3: pir_split_bf16x3 pieces of W_in
// [2 hid][C] (forward orientation, k padded to kp); ws: pir_gdfn_fused_ws_bytes bytes.  mean / rstd [B][HW]: optional outputs
// (the statistics of the LayerNorm, for a backward pass).  1000 = shape not served (nothing launched).
extern "C" int pir_gdfn_fused_fwd(const float* x, long x_bs, const float* ln_w, const float* ln_b, const void* w3, int kp,
                                  const float* wd, float* g, long g_bs, void* ws, size_t ws_bytes, float* mean, float* rstd,
                                  int B, int C, int hid, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(x && ln_w && ln_b && w3 && wd && g && ws && B > 0 && C > 0 && hid > 0 && H > 0 && W > 0);
  PIR_CHECK_ARG((mean == nullptr) == (rstd == nullptr));
  if (!((C == 48 || C == 96) && (W == 128 || W == 64) && kp == C)) return 1000;
  if ((reinterpret_cast<uintptr_t>(g) & 15) || g_bs % 4 || (reinterpret_cast<uintptr_t>(ws) & 15) || (reinterpret_cast<uintptr_t>(w3) & 15)) return 1000;
  if (B > 65535 || (long)hid * H * W >= (1L << 31)) return 1000;
  if (ws_bytes < pir_gdfn_fused_ws_bytes(B, C, H, W)) return PIR_ENOMEM;
  hipStream_t s = (hipStream_t)stream;
  bf16x8* xn3 = reinterpret_cast<bf16x8*>(ws);
  const long pixels = (long)B * H * W;
  const dim3 g1((unsigned)pir_cdiv(pixels, 256));
  if (C == 96) hipLaunchKernelGGL((ln_split_rows_kernel<96>), g1, dim3(256), 0, s, x, x_bs, ln_w, ln_b, xn3, mean, rstd, H, W, B);
  else hipLaunchKernelGGL((ln_split_rows_kernel<48>), g1, dim3(256), 0, s, x, x_bs, ln_w, ln_b, xn3, mean, rstd, H, W, B);
  int st = pir_launch_status();
  if (st) return st;
  FusedArgs a;
  a.xn3 = xn3; a.w3 = reinterpret_cast<const bf16x8*>(w3); a.wd = wd; a.g = g; a.g_bs = g_bs; a.hid = hid; a.H = H; a.W = W;
  a.dump = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)B * H * W * C * 6);
  // bands of output rows: enough workgroups for two per CU (a workgroup is one wave per SIMD); each band re-multiplies
  // two halo rows, so no band is shorter than 16 rows (12 % recompute)
  const long base = pir_cdiv(hid, 32) * (long)B;
  long nbands = pir_cdiv(4L * PIR_NUM_CU, base);
  if (nbands > H / 16) nbands = H / 16;
  if (nbands < 1) nbands = 1;
  a.band = (int)pir_cdiv(H, nbands);
  a.nchunks = (int)pir_cdiv(hid, 32); a.nbands = (int)pir_cdiv(H, a.band); a.B = B;
  const long units = (long)B * a.nbands;
  const dim3 grid((unsigned)(pir_cdiv(units, 8) * 8 * a.nchunks));
#define PIR_GF(KS_, NW_) do { \
    if (g_fused_pipe && g_fused_fast_erf) hipLaunchKernelGGL((gdfn_fused_pipe_kernel<KS_, NW_, true>), grid, dim3(NW_ * 64), 0, s, a); \
    else if (g_fused_pipe) hipLaunchKernelGGL((gdfn_fused_pipe_kernel<KS_, NW_, false>), grid, dim3(NW_ * 64), 0, s, a); \
    else if (g_fused_fast_erf) hipLaunchKernelGGL((gdfn_fused_kernel<KS_, NW_, true>), grid, dim3(NW_ * 64), 0, s, a); \
    else hipLaunchKernelGGL((gdfn_fused_kernel<KS_, NW_, false>), grid, dim3(NW_ * 64), 0, s, a); } while (0)
  if (C == 96 && W == 128) PIR_GF(6, 4);
  else if (C == 96) PIR_GF(6, 2);
  else if (W == 128) PIR_GF(3, 4);
  else PIR_GF(3, 2);
#undef PIR_GF
  return pir_launch_status();
}
