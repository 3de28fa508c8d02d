// fp32 MFMA contractions of the PromptIR path (gfx950).
//
// Two kernels cover every dense contraction of net/model.py:
//   gemm_nn : Y[m][n] = sum_k A(m,k) X[k][n]      (pixels n contiguous in X and Y)
//             1x1 convs, their input gradients, attn@v, dense 3x3 as 9 shifted GEMMs
//   gemm_nt : G[i][j] = sum_n X[i][n] Y[j][n]     (contraction over pixels, split-K)
//             q k^T, every 1x1 / 3x3 weight gradient, dOut v^T
//
// Both stage k-major operand tiles in LDS and feed v_mfma_f32_32x32x2_f32
// (exact fp32, 64 FLOP/clk/SIMD = the chip's 157 TFLOP/s fp32 matrix peak; there is no
// xf32/TF32 on gfx950, and the 1e-4 parity bar rules out bf16 inputs).
// Operand lane maps (cdna_hip_programming.md §3): A: lane l holds A[i=l&31][k=l>>5],
// B: B[k=l>>5][j=l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#include "gemm_common.h"
#include <stdlib.h>


namespace {

#ifndef PIR_BK
#define PIR_BK 16
#endif
constexpr int BK = PIR_BK;  // k-depth of one LDS stage (PIR_BK/2 MFMA k-steps)
#ifndef PIR_PIN_SCHED
#define PIR_PIN_SCHED 0  /* A/B on MI355X: pinning the ds_read/MFMA order is 1.5% slower overall */
#endif
constexpr bool PIN_SCHED = PIR_PIN_SCHED;

// tuning overrides (pir_tune_set): -1 / 0 = automatic
int g_nn_cfg = -1, g_nt_cfg = -1, g_nt_splits = 0, g_nn_x3 = -1, g_nt_x3 = -1;
int g_nt_want_half = 5;   // knob 19: split-K workgroups per CU aimed at, in halves (bench sweep with two part-batch streams:
                          // 2: 100.5 ms, 4: 99.4, 5: 99.4, 6: 99.7-100.3, 8: 101.0, 12: 102.2; one stream: 6 beats 4 by 0.5 ms)
int g_nt_group_wide = 0;   // knob 34: 128 x 192 tiles in the grouped low-resolution weight gradients (measured neutral in the two-stream step, round 4: off)
int g_nt_tile96 = 0;  // knob 33: 96 x 96 three-wave tile for outputs of at most 96 x 96 (measured neutral in the step, round 4: off; 0 = the 128 x 96 tile)
int g_nt_quad = -1;   // knob 14: gemm_nt_x3 four-lanes-per-row stage loads (-1 automatic, 0 never, 1 always)

__device__ __forceinline__ int c_row(int reg, int lane) { return pir_c_row(reg, lane); }

// ---------------------------------------------------------------------------------------------
// gemm_nn
// ---------------------------------------------------------------------------------------------
struct NNParams {
  pir_gemm_nn_t g;
  // dense 3x3 mode (taps == 9): N == H*W, X rows are image planes
  int taps, flip, H, W;
  long a_st;
};

template <int NA, int NB, int VEC>
struct NNStage {  // one k-stage of both operands held in registers between global load and LDS store
  float a[NA];
  float b[NB][VEC];
};

template <int TM, int TN, int WM, int WN, bool A_MFAST, int VEC, bool CONV, int DEPTH>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nn_kernel(NNParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, T = WM * WN * 64;
  constexpr int AS = BM + 2;  // LDS row strides (floats); +2 keeps the k-fast A staging conflict-free
  constexpr int BS = BN + 4;
  constexpr int STAGE = BK * (AS + BS);
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const pir_gemm_nn_t& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int wg = pir_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg % tiles_m) * BM, n0 = (wg / tiles_m) * BN;
  const int o1 = blockIdx.y / g.O2, o2 = blockIdx.y % g.O2;

  const float* __restrict__ A = g.A + o1 * g.a_s1 + o2 * g.a_s2;
  const float* __restrict__ X = g.X + o1 * g.x_s1 + o2 * g.x_s2;
  float* __restrict__ Y = g.Y + o1 * g.y_s1 + o2 * g.y_s2;

  constexpr int A_EL = BK * BM, NA = (A_EL + T - 1) / T;
  constexpr int B_CH = BK * BN / VEC, NB = (B_CH + T - 1) / T;
  using Stage = NNStage<NA, NB, VEC>;

  const int ktiles = (g.K + BK - 1) / BK;
  const int iters = ktiles * (CONV ? 9 : 1);

  // Loads are unconditional from clamped (always valid) addresses: no exec-mask branches, and
  // nothing consumes the values until stash(), so the loads stay in flight across compute().
  // Out-of-range elements are zeroed when the stage is written to LDS (stash recomputes the
  // predicates).  Addresses are (wave-uniform stage base) + (32-bit per-thread offset).
  auto load = [&](int it, Stage& st) {
    const int tap = CONV ? it / ktiles : 0;
    const int k0 = (CONV ? it % ktiles : it) * BK;
    const int klast = g.K - 1 - k0;  // last valid k offset inside this stage (>= 0)
    const float* __restrict__ At = A + (CONV ? (long)(p.flip ? 8 - tap : tap) * p.a_st : 0) + (long)k0 * g.a_sk;
    const float* __restrict__ Xt = X + (long)k0 * g.ldx;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * T;
      int kk, mm;
      if (A_MFAST) { mm = e % BM; kk = e / BM; } else { kk = e % BK; mm = e / BK; }
      const int m = m0 + mm;
      const int kc = kk <= klast ? kk : klast, mc = m < g.M ? m : g.M - 1;
      st.a[i] = At[mc * (int)g.a_sm + kc * (int)g.a_sk];
    }
    int dh = 0, dw = 0;
    if (CONV) { dh = tap / 3 - 1; dw = tap % 3 - 1; }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + i * T;
      const int kk = c / (BN / VEC), nn = (c % (BN / VEC)) * VEC;
      const int n = n0 + nn;
      const int kc = kk <= klast ? kk : klast;
      if (!CONV) {
        if (VEC == 4) {
          const int nc = n < g.N ? n : g.N - 4;  // N % 4 == 0 and N >= 4 on this path
          const f32x4 v = *reinterpret_cast<const f32x4*>(Xt + (kc * (int)g.ldx + nc));
          st.b[i][0] = v[0]; st.b[i][1] = v[1]; st.b[i][2] = v[2]; st.b[i][3] = v[3];
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const int nj = n + j, nc = nj < g.N ? nj : g.N - 1;
            st.b[i][j] = Xt[kc * (int)g.ldx + nc];
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int nj = n + j, nc = nj < g.N ? nj : g.N - 1;
          const int h = nc / p.W + dh, w = nc % p.W + dw;
          const int hc = h < 0 ? 0 : (h >= p.H ? p.H - 1 : h), wc = w < 0 ? 0 : (w >= p.W ? p.W - 1 : w);
          st.b[i][j] = Xt[kc * (int)g.ldx + hc * p.W + wc];
        }
      }
    }
  };

  auto stash = [&](int buf, int it, const Stage& st) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * AS;
    const int tap = CONV ? it / ktiles : 0;
    const int k0 = (CONV ? it % ktiles : it) * BK;
    const int klast = g.K - 1 - k0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * T;
      int kk, mm;
      if (A_MFAST) { mm = e % BM; kk = e / BM; } else { kk = e % BK; mm = e / BK; }
      const bool ok = kk <= klast && m0 + mm < g.M;
      if (A_EL % T == 0 || e < A_EL) As[kk * AS + mm] = ok ? st.a[i] : 0.f;
    }
    int dh = 0, dw = 0;
    if (CONV) { dh = tap / 3 - 1; dw = tap % 3 - 1; }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + i * T;
      const int kk = c / (BN / VEC), nn = (c % (BN / VEC)) * VEC;
      const int n = n0 + nn;
      const bool rowok = kk <= klast;
      if (B_CH % T == 0 || c < B_CH) {
        if (VEC == 4 && !CONV) {
          const bool ok = rowok && n < g.N;
          f32x4 v = {ok ? st.b[i][0] : 0.f, ok ? st.b[i][1] : 0.f, ok ? st.b[i][2] : 0.f, ok ? st.b[i][3] : 0.f};
          *reinterpret_cast<f32x4*>(Bs + kk * BS + nn) = v;
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const int nj = n + j;
            bool ok = rowok && nj < g.N;
            if (CONV) {
              const int nc = nj < g.N ? nj : g.N - 1;
              const int h = nc / p.W + dh, w = nc % p.W + dw;
              ok = ok && h >= 0 && h < p.H && w >= 0 && w < p.W;
            }
            Bs[kk * BS + nn + j] = ok ? st.b[i][j] : 0.f;
          }
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS operand reads are software-pipelined one MFMA k-step ahead (register double buffer),
  // so a lone wave on a SIMD does not stall on ds_read latency between MFMA groups.
  auto compute = [&](int buf) {
    const float* As = smem + buf * STAGE + (lane >> 5) * AS + wm * TM * 32 + (lane & 31);
    const float* Bs = smem + buf * STAGE + BK * AS + (lane >> 5) * BS + wn * TN * 32 + (lane & 31);
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = As[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = Bs[j * 32];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
      if (kk + 2 < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[nxt][i] = As[(kk + 2) * AS + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[nxt][j] = Bs[(kk + 2) * BS + j * 32];
      }
      if (PIN_SCHED) __builtin_amdgcn_sched_barrier(0);  // keep the next step's ds_reads AHEAD of this step's MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      if (PIN_SCHED) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Global prefetch runs TWO stages ahead through two register sets (loop unrolled by two so the
  // set index is static): 2 stages x (A+B) bytes per block stay in flight, which is what the
  // small-K, HBM-bound 1x1 convs of the full-resolution levels need.
  if (DEPTH == 2) {
    Stage s0, s1;
    load(0, s0);
    if (iters > 1) load(1, s1);
    stash(0, 0, s0);
    __syncthreads();
    int it = 0;
    for (; it + 1 < iters; it += 2) {
      if (it + 2 < iters) load(it + 2, s0);
      compute(0);
      stash(1, it + 1, s1);
      __syncthreads();
      if (it + 3 < iters) load(it + 3, s1);
      compute(1);
      if (it + 2 < iters) stash(0, it + 2, s0);
      __syncthreads();
    }
    if (it < iters) compute(0);  // odd tail: its stage was stashed into buffer 0 by the last iteration
  } else {  // one stage ahead, one register set: fewer VGPRs -> more resident blocks
    Stage s0;
    load(0, s0);
    stash(0, 0, s0);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      if (it + 1 < iters) load(it + 1, s0);
      compute(it & 1);
      if (it + 1 < iters) stash((it & 1) ^ 1, it + 1, s0);
      __syncthreads();
    }
  }

  pir_nn_epilogue<TM, TN>(acc, g, Y, o1, o2, m0, n0, wm, wn, lane);
}

template <int TM, int TN, int WM, int WN>
int launch_nn_cfg(const NNParams& p, hipStream_t s) {
  const pir_gemm_nn_t& g = p.g;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const long tiles = pir_cdiv(g.M, BM) * pir_cdiv(g.N, BN);
  dim3 grid((unsigned)tiles, (unsigned)(g.O1 * g.O2)), block(WM * WN * 64);
  const bool conv = p.taps == 9;
  const bool mfast = g.a_sm == 1;
  static const int depth_env = getenv("PIR_NN_DEPTH") ? atoi(getenv("PIR_NN_DEPTH")) : 0;
  const bool depth2 = depth_env ? depth_env == 2 : true;  // two-stage prefetch: 4% faster over the step's shapes
  const bool vec4 = !conv && g.N % 4 == 0 && g.ldx % 4 == 0 && g.x_s1 % 4 == 0 && g.x_s2 % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(g.X) & 15) == 0;
#define PIR_NN_LAUNCH(MF, V, C) \
  do { if (depth2) hipLaunchKernelGGL((gemm_nn_kernel<TM, TN, WM, WN, MF, V, C, 2>), grid, block, 0, s, p); \
       else hipLaunchKernelGGL((gemm_nn_kernel<TM, TN, WM, WN, MF, V, C, 1>), grid, block, 0, s, p); } while (0)
  if (conv) {
    if (mfast) PIR_NN_LAUNCH(true, 1, true); else PIR_NN_LAUNCH(false, 1, true);
  } else if (vec4) {
    if (mfast) PIR_NN_LAUNCH(true, 4, false); else PIR_NN_LAUNCH(false, 4, false);
  } else {
    if (mfast) PIR_NN_LAUNCH(true, 1, false); else PIR_NN_LAUNCH(false, 1, false);
  }
#undef PIR_NN_LAUNCH
  return pir_launch_status();
}

int launch_nn(const NNParams& p, hipStream_t s) {
  const int M = p.g.M;
  const long batch = (long)p.g.O1 * p.g.O2;
  // pick the M-tile height (multiple of 32) that wastes the fewest padded rows
  if (g_nn_cfg < 0 && M <= 32) return launch_nn_cfg<1, 2, 1, 4>(p, s);   // 32 x 256
  if (g_nn_cfg < 0 && M <= 64) return launch_nn_cfg<2, 2, 1, 4>(p, s);   // 64 x 256
  const long pad96 = pir_cdiv(M, 96) * 96, pad128 = pir_cdiv(M, 128) * 128;
  if (g_nn_cfg >= 0) {
    switch (g_nn_cfg) {
      case 0: return launch_nn_cfg<1, 2, 1, 4>(p, s);
      case 1: return launch_nn_cfg<2, 2, 1, 4>(p, s);
      case 2: return launch_nn_cfg<3, 2, 1, 4>(p, s);
      case 3: return launch_nn_cfg<2, 2, 2, 2>(p, s);
      default: return launch_nn_cfg<1, 2, 2, 2>(p, s);
    }
  }
  const bool use96 = pad96 < pad128;
  const long blocks = use96 ? pir_cdiv(M, 96) * pir_cdiv(p.g.N, 256) * batch : pir_cdiv(M, 128) * pir_cdiv(p.g.N, 128) * batch;
  // low-resolution levels: too few 128-row tiles to give every CU two blocks -> 64 x 128 tiles
  if (blocks < 2L * PIR_NUM_CU && pir_cdiv(M, 64) * 64 <= pad128) return launch_nn_cfg<1, 2, 2, 2>(p, s);
  if (use96) return launch_nn_cfg<3, 2, 1, 4>(p, s);  // 96 x 256
  return launch_nn_cfg<2, 2, 2, 2>(p, s);              // 128 x 128
}

// ---------------------------------------------------------------------------------------------
// gemm_nt (split-K over pixels, deterministic two-stage reduction)
// ---------------------------------------------------------------------------------------------
constexpr int NT_BK = 32;  // pixels per LDS stage

struct NTParams {
  pir_gemm_nt_t g;
  int splits;        // split-K factor over the flattened (r, n-chunk) axis
  int chunks_per_r;  // ceil(N / NT_BK)
  // dense-3x3 weight gradient in ONE launch (gemm_nt_x3_kernel<..., true>): the Y operand has M2 = 9*C virtual
  // rows, row j = channel j/9 shifted by tap j%9 (times tap_sign); g.H x g.W is the image, W % 8 == 0
  int tap_sign;
  unsigned magic_w;
};

// WK waves of a block share an output tile and each takes a slice of every stage's k-range.
// VEC=4: 16-byte loads along the pixel axis (needs 16-byte aligned rows), transposed into the
// k-major LDS image by four conflict-free ds_write_b32 (odd row stride).
template <int TM, int TN, int WM, int WN, int WK, int VEC>
__global__ __launch_bounds__(WM* WN* WK * 64) void gemm_nt_kernel(NTParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, T = WM * WN * WK * 64;
  constexpr int XS = BM + 1, YS = BN + 1;  // odd strides: conflict-free transposing stores
  constexpr int STAGE = NT_BK * (XS + YS);
  constexpr int RED = (WK > 1) ? WM * WN * (WK / 2) * TM * TN * 16 * 64 : 0;  // pairwise tree: half the waves park
  constexpr int SMEM = (2 * STAGE > RED) ? 2 * STAGE : RED;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const pir_gemm_nt_t& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid % WK, wmn = wid / WK, wm = wmn / WN, wn = wmn % WN;
  const int tiles_i = (g.M1 + BM - 1) / BM;
  // XCD-aware order: the output tiles of one split (same pixel range => same operand rows) are neighbours in the
  // logical order, so they run on one XCD and share its L2 instead of each XCD fetching the rows from HBM
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int rr = pir_xcd_remap(lin, gridDim.x * gridDim.y * gridDim.z);
  const int tile = rr % (int)gridDim.x, rest = rr / (int)gridDim.x;
  const int i0 = (tile % tiles_i) * BM, j0 = (tile / tiles_i) * BN;
  const int split = rest % (int)gridDim.y;
  const int o = rest / (int)gridDim.y, o1 = o / g.O2, o2 = o % g.O2;

  const float* __restrict__ Xb = g.X + o1 * g.x_s1 + o2 * g.x_s2;
  const float* __restrict__ Yb = g.Y + o1 * g.y_s1 + o2 * g.y_s2;

  const int total = g.BR * p.chunks_per_r;
  const int per = (total + p.splits - 1) / p.splits;
  const int c_begin = split * per, c_end = (c_begin + per < total) ? c_begin + per : total;
  int l_r = c_begin / p.chunks_per_r, l_ch = c_begin - l_r * p.chunks_per_r;   // load cursor (no per-stage division)

  constexpr int CPR = NT_BK / VEC;                       // vector chunks per row of a stage
  constexpr int X_CH = CPR * BM, NX = (X_CH + T - 1) / T;
  constexpr int Y_CH = CPR * BN, NY = (Y_CH + T - 1) / T;
  float rx[NX][VEC], ry[NY][VEC];

  auto load = [&]() {
    const int r = l_r;
    const int nb = l_ch * NT_BK;
    if (++l_ch == p.chunks_per_r) { l_ch = 0; ++l_r; }
    const float* Xp = Xb + r * g.x_sr;
    const float* Yp = Yb + r * g.y_sr;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int e = tid + q * T;
      const int nn = (e % CPR) * VEC, ii = e / CPR;
      const int n = nb + nn, i = i0 + ii;
      const bool ok = (X_CH % T == 0 || e < X_CH) && i < g.M1;
      if (VEC == 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok && n < g.N) v = *reinterpret_cast<const f32x4*>(Xp + (long)i * g.ldx + n);
        rx[q][0] = v[0]; rx[q][1] = v[1]; rx[q][2] = v[2]; rx[q][3] = v[3];
      } else {
        rx[q][0] = (ok && n < g.N) ? Xp[(long)i * g.ldx + n] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < NY; ++q) {
      const int e = tid + q * T;
      const int nn = (e % CPR) * VEC, jj = e / CPR;
      const int n = nb + nn, j = j0 + jj;
      const bool ok = (Y_CH % T == 0 || e < Y_CH) && j < g.M2;
      if (VEC == 4) {  // never used with a tap shift (launcher picks VEC=1 then)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok && n < g.N) v = *reinterpret_cast<const f32x4*>(Yp + (long)j * g.ldy + n);
        ry[q][0] = v[0]; ry[q][1] = v[1]; ry[q][2] = v[2]; ry[q][3] = v[3];
      } else {
        bool okn = ok && n < g.N;
        long off = (long)j * g.ldy + n;
        if (g.H > 0) {  // 3x3 tap shift on the Y operand (dense 3x3 weight gradient)
          const int h = n / g.W + g.shift_dh, w = n % g.W + g.shift_dw;
          okn = okn && h >= 0 && h < g.H && w >= 0 && w < g.W;
          off = (long)j * g.ldy + (long)h * g.W + w;
        }
        ry[q][0] = okn ? Yp[off] : 0.f;
      }
    }
  };
  auto stash = [&](int buf) {
    float* Xs = smem + buf * STAGE;
    float* Ys = Xs + NT_BK * XS;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int e = tid + q * T;
      if (X_CH % T == 0 || e < X_CH) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) Xs[((e % CPR) * VEC + v) * XS + e / CPR] = rx[q][v];
      }
    }
#pragma unroll
    for (int q = 0; q < NY; ++q) {
      const int e = tid + q * T;
      if (Y_CH % T == 0 || e < Y_CH) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) Ys[((e % CPR) * VEC + v) * YS + e / CPR] = ry[q][v];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (c_begin < c_end) {
    load();
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c, buf ^= 1) {
      if (c + 1 < c_end) load();
      constexpr int KW = NT_BK / WK;  // k-range of this wave inside the stage
      const float* Xs = smem + buf * STAGE + (wk * KW + (lane >> 5)) * XS + wm * TM * 32 + (lane & 31);
      const float* Ys = smem + buf * STAGE + NT_BK * XS + (wk * KW + (lane >> 5)) * YS + wn * TN * 32 + (lane & 31);
#pragma unroll
      for (int kk = 0; kk < KW; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = Xs[kk * XS + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Ys[kk * YS + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (c + 1 < c_end) stash(buf ^ 1);
      __syncthreads();
    }
  }

  // cross-wave (WK) reduction: pairwise tree through LDS, fixed order -> deterministic
  if (WK > 1) {
#pragma unroll
    for (int half = WK / 2; half >= 1; half >>= 1) {
      __syncthreads();
      if (wk >= half && wk < 2 * half) {
        float* dst = smem + ((wmn * half + (wk - half)) * TM * TN * 16) * 64;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
      }
      __syncthreads();
      if (wk < half) {
        const float* src = smem + ((wmn * half + wk) * TM * TN * 16) * 64;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * TN + j) * 16 + r) * 64 + lane];
      }
    }
  }
  if (wk != 0) return;

  // partial tile -> workspace [split][o][M1][M2]
  float* __restrict__ P = g.ws + ((long)split * (g.O1 * g.O2) + o) * ((long)g.M1 * g.M2);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int jj = j0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = i0 + (wm * TM + i) * 32 + c_row(r, lane);
        if (ii < g.M1 && jj < g.M2) P[(long)ii * g.M2 + jj] = acc[i][j][r];
      }
    }
}

// gemm_nt on the bf16 matrix cores (bf16x3 split, see gemm_x3.hip): both operands are rows with the
// contraction index (pixels) contiguous, so an MFMA fragment = 8 consecutive pixels of one row = two 16-byte
// global loads, split into three bf16 pieces when the stage is written to LDS ([k-group][row][8 x bf16]).
// One stage = 16 pixels = one v_mfma_f32_32x32x16_bf16 k-step; split-K, partial layout and the reduction
// are those of gemm_nt_kernel.  Needs 16-byte aligned rows and N % 4 == 0 (else the fp32 kernel runs).
constexpr int X3_BK = 16;
// waves_per_eu(3): three workgroups per CU (LDS allows it for every tile); the 128 x 128 tile would otherwise be
// allocated 172 registers, four too many
// QUAD (plain operands only): four lanes read one row's 64 bytes of a stage (16 bytes each) and each lane splits and
// stores its 4 pixels (8 bytes per piece), instead of two lanes x two 16-byte loads per row: one L1 request per row and
// stage instead of two.
// (lin, ntiles, nsplits, nbatch): the workgroup's linear index inside ITS problem and that problem's grid extents - the
// stand-alone kernel passes its own grid, the grouped kernel (several problems in one launch) the problem's slice of it
template <int TM, int TN, int WM, int WN, bool TAPS, bool QUAD>
__device__ __forceinline__ void gemm_nt_x3_body(const NTParams& p, pir_bf16x8* smem, int lin, int ntiles, int nsplits, int nbatch) {
  static_assert(!(TAPS && QUAD), "the tap-shifted operand keeps the fragment mapping");
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, T = WM * WN * 64;
  constexpr int XKS = BM + 4, YKS = BN + 4;            // 16-byte units between the two k-groups
  constexpr int XU = 2 * XKS, YU = 2 * YKS, PART = XU + YU, STAGE = 3 * PART;

  const pir_gemm_nt_t& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_i = (g.M1 + BM - 1) / BM;
  // XCD-aware order: the output tiles of one split (same pixel range => same operand rows) are neighbours in the
  // logical order, so they run on one XCD and share its L2 instead of each XCD fetching the rows from HBM
  const int rr = pir_xcd_remap(lin, ntiles * nsplits * nbatch);
  const int tile = rr % ntiles, rest = rr / ntiles;
  const int i0 = (tile % tiles_i) * BM, j0 = (tile / tiles_i) * BN;
  const int split = rest % nsplits;
  const int o = rest / nsplits, o1 = o / g.O2, o2 = o % g.O2;
  const float* __restrict__ Xb = g.X + o1 * g.x_s1 + o2 * g.x_s2;
  const float* __restrict__ Yb = g.Y + o1 * g.y_s1 + o2 * g.y_s2;

  // the launcher sets chunks_per_r = ceil(N / 16) for this kernel
  // 32-bit stage counters and an incremental (image, chunk) cursor: a 64-bit division per stage on the scalar
  // unit costs more than the stage's MFMAs (the host checks BR * chunks_per_r < 2^31)
  const int total = g.BR * p.chunks_per_r;
  const int per = (total + p.splits - 1) / p.splits;
  const int c_begin = split * per, c_end = (c_begin + per < total) ? c_begin + per : total;
  int l_r = c_begin / p.chunks_per_r, l_ch = c_begin - l_r * p.chunks_per_r;   // load cursor

  constexpr int XF = 2 * BM, NX = (XF + T - 1) / T;    // fragments (8 pixels of one row) per stage
  constexpr int YF = 2 * BN, NY = (YF + T - 1) / T;
  constexpr int XQ = 4 * BM, NXQ = (XQ + T - 1) / T;   // QUAD: 4-pixel pieces per stage
  constexpr int YQ = 4 * BN, NYQ = (YQ + T - 1) / T;
  struct Stage { f32x4 x[QUAD ? 1 : NX][2]; f32x4 y[QUAD ? 1 : NY][2]; f32x4 xq[QUAD ? NXQ : 1]; f32x4 yq[QUAD ? NYQ : 1];
                 float ye[TAPS ? NY : 1]; int nb; };

  auto load = [&](Stage& st) {   // loads the stage under the cursor, then advances it
    const int r = l_r;
    const int nb = l_ch * X3_BK;
    st.nb = nb;
    if (++l_ch == p.chunks_per_r) { l_ch = 0; ++l_r; }
    const float* __restrict__ Xp = Xb + r * g.x_sr;
    const float* __restrict__ Yp = Yb + r * g.y_sr;
    if constexpr (QUAD) {
#pragma unroll
      for (int q = 0; q < NXQ; ++q) {
        const int f = tid + q * T;
        int qd = f & 3, ii = f >> 2;
        if (ii >= BM) { ii = 0; qd = 0; }
        const int i = i0 + ii, ic = i < g.M1 ? i : g.M1 - 1;
        const int n = nb + 4 * qd, nc = n < g.N ? n : g.N - 4;
        st.xq[q] = *reinterpret_cast<const f32x4*>(Xp + ((long)ic * g.ldx + nc));
      }
#pragma unroll
      for (int q = 0; q < NYQ; ++q) {
        const int f = tid + q * T;
        int qd = f & 3, jj = f >> 2;
        if (jj >= BN) { jj = 0; qd = 0; }
        const int j = j0 + jj, jc = j < g.M2 ? j : g.M2 - 1;
        const int n = nb + 4 * qd, nc = n < g.N ? n : g.N - 4;
        st.yq[q] = *reinterpret_cast<const f32x4*>(Yp + ((long)jc * g.ldy + nc));
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int f = tid + q * T;
      int kg = f & 1, ii = f >> 1;                      // the two k-groups of a row on adjacent lanes: 64 B per row
      if (ii >= BM) { ii = 0; kg = 0; }
      const int i = i0 + ii, ic = i < g.M1 ? i : g.M1 - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int n = nb + 8 * kg + 4 * h, nc = n < g.N ? n : g.N - 4;
        st.x[q][h] = *reinterpret_cast<const f32x4*>(Xp + ((long)ic * g.ldx + nc));
      }
    }
#pragma unroll
    for (int q = 0; q < NY; ++q) {
      const int f = tid + q * T;
      int kg = f & 1, jj = f >> 1;
      if (jj >= BN) { jj = 0; kg = 0; }
      const int j = j0 + jj, jc = j < g.M2 ? j : g.M2 - 1;
      if (TAPS) {
        // virtual row jc = channel jc/9 seen through tap jc%9: the 8 pixels of a fragment lie in one image row
        // (W % 8 == 0); two aligned float4 of the shifted row plus the one element that slides in from the side
        const int ch = jc / 9, tap = jc - 9 * ch;
        const int dh = p.tap_sign * (tap / 3 - 1), dw = p.tap_sign * (tap % 3 - 1);
        const int n = nb + 8 * kg, nc = n < g.N ? n : g.N - 8;
        const int hh = pir_fastdiv(nc, p.magic_w), w0 = nc - hh * g.W;
        const int hs = hh + dh;
        const bool rok = hs >= 0 && hs < g.H;
        const float* __restrict__ row = Yp + ((long)ch * g.ldy + (rok ? hs : hh) * g.W);
        st.y[q][0] = *reinterpret_cast<const f32x4*>(row + w0);
        st.y[q][1] = *reinterpret_cast<const f32x4*>(row + w0 + 4);
        const int we = dw < 0 ? w0 - 1 : w0 + 8;
        const bool eok = we >= 0 && we < g.W;
        st.ye[q] = row[eok ? we : w0];
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int n = nb + 8 * kg + 4 * h, nc = n < g.N ? n : g.N - 4;
          st.y[q][h] = *reinterpret_cast<const f32x4*>(Yp + ((long)jc * g.ldy + nc));
        }
      }
    }
  };
  auto stash = [&](int buf, const Stage& st) {
    pir_bf16x8* base = smem + buf * STAGE;
    const int nb = st.nb;
    if constexpr (QUAD) {
      pir_bf16x4* base4 = reinterpret_cast<pir_bf16x4*>(base);
#pragma unroll
      for (int q = 0; q < NXQ; ++q) {
        const int f = tid + q * T;
        const int qd = f & 3, ii = f >> 2;
        if (XQ % T == 0 || f < XQ) {
          const bool ok = i0 + ii < g.M1 && nb + 4 * qd < g.N;   // N % 4 == 0: a float4 is all-or-nothing
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ok ? st.xq[q][e] : 0.f;
          const pir_frag3h fr = pir_split4(v);
          const int u = 2 * ((qd >> 1) * XKS + ii) + (qd & 1);
          base4[u] = fr.hi; base4[2 * PART + u] = fr.mid; base4[4 * PART + u] = fr.lo;
        }
      }
#pragma unroll
      for (int q = 0; q < NYQ; ++q) {
        const int f = tid + q * T;
        const int qd = f & 3, jj = f >> 2;
        if (YQ % T == 0 || f < YQ) {
          const bool ok = j0 + jj < g.M2 && nb + 4 * qd < g.N;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ok ? st.yq[q][e] : 0.f;
          const pir_frag3h fr = pir_split4(v);
          const int u = 2 * (XU + (qd >> 1) * YKS + jj) + (qd & 1);
          base4[u] = fr.hi; base4[2 * PART + u] = fr.mid; base4[4 * PART + u] = fr.lo;
        }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int f = tid + q * T;
      const int kg = f & 1, ii = f >> 1;
      if (XF % T == 0 || f < XF) {
        const bool rok = i0 + ii < g.M1;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool ok = rok && nb + 8 * kg + 4 * h < g.N;   // N % 4 == 0: a float4 is all-or-nothing
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * h + e] = ok ? st.x[q][h][e] : 0.f;
        }
        const pir_frag3 fr = pir_split8(v);
        const int u = kg * XKS + ii;
        base[u] = fr.hi; base[PART + u] = fr.mid; base[2 * PART + u] = fr.lo;
      }
    }
#pragma unroll
    for (int q = 0; q < NY; ++q) {
      const int f = tid + q * T;
      const int kg = f & 1, jj = f >> 1;
      if (YF % T == 0 || f < YF) {
        const bool rok = j0 + jj < g.M2;
        float v[8];
        if (TAPS) {
          const int j = j0 + jj, jc = j < g.M2 ? j : g.M2 - 1;
          const int ch = jc / 9, tap = jc - 9 * ch;
          const int dh = p.tap_sign * (tap / 3 - 1), dw = p.tap_sign * (tap % 3 - 1);
          const int n = nb + 8 * kg;
          const int hh = pir_fastdiv(n < g.N ? n : g.N - 8, p.magic_w), w0 = (n < g.N ? n : g.N - 8) - hh * g.W;
          const bool ok = rok && n < g.N && hh + dh >= 0 && hh + dh < g.H;
          const int we = dw < 0 ? w0 - 1 : w0 + 8;
          const float ext = (we >= 0 && we < g.W) ? st.ye[q] : 0.f;
          float a[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { a[e] = st.y[q][0][e]; a[4 + e] = st.y[q][1][e]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float left = e == 0 ? ext : a[e - 1], right = e == 7 ? ext : a[e + 1];
            const float val = dw == 0 ? a[e] : (dw < 0 ? left : right);
            v[e] = ok ? val : 0.f;
          }
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const bool ok = rok && nb + 8 * kg + 4 * h < g.N;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * h + e] = ok ? st.y[q][h][e] : 0.f;
          }
        }
        const pir_frag3 fr = pir_split8(v);
        const int u = XU + kg * YKS + jj;
        base[u] = fr.hi; base[PART + u] = fr.mid; base[2 * PART + u] = fr.lo;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    const pir_bf16x8* base = smem + buf * STAGE;
    const int h = lane >> 5, r = lane & 31;
    const pir_bf16x8* xp = base + h * XKS + wm * TM * 32 + r;
    const pir_bf16x8* yp = base + XU + h * YKS + wn * TN * 32 + r;
    pir_bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { ah[i] = xp[i * 32]; am[i] = xp[PART + i * 32]; al[i] = xp[2 * PART + i * 32]; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { bh[j] = yp[j * 32]; bm[j] = yp[PART + j * 32]; bl[j] = yp[2 * PART + j * 32]; }
    // term-major: consecutive MFMAs hit different accumulators (same per-accumulator term order as pir_mfma_x3)
#define PIR_X3_TERM(A_, B_)                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_[j], acc[i][j], 0, 0, 0);
    PIR_X3_TERM(al, bh)
    PIR_X3_TERM(ah, bl)
    PIR_X3_TERM(am, bm)
    PIR_X3_TERM(am, bh)
    PIR_X3_TERM(ah, bm)
    PIR_X3_TERM(ah, bh)
#undef PIR_X3_TERM
  };

  if (c_begin < c_end) {
    Stage s0, s1;
    load(s0);
    if (c_begin + 1 < c_end) load(s1);
    stash(0, s0);
    __syncthreads();
    int c = c_begin;
    // main part: both prefetch loads in range, so they are unconditional and the compiler keeps exact vmcnt counts
    // (the loads of the stage after next stay in flight across the split: +8-12 % on every weight gradient;
    // a third register stage (168 registers, no spills after the quad-load rewrite) measured +1.5-4 %: round 2)
    for (; c + 3 < c_end; c += 2) {
      load(s0);
      compute(0);
      stash(1, s1);
      __syncthreads();
      load(s1);
      compute(1);
      stash(0, s0);
      __syncthreads();
    }
    for (; c + 1 < c_end; c += 2) {   // tail: loads guarded
      if (c + 2 < c_end) load(s0);
      compute(0);
      stash(1, s1);
      __syncthreads();
      if (c + 3 < c_end) load(s1);
      compute(1);
      if (c + 2 < c_end) stash(0, s0);
      __syncthreads();
    }
    if (c < c_end) compute(0);
  }

  float* __restrict__ P = g.ws + ((long)split * (g.O1 * g.O2) + o) * ((long)g.M1 * g.M2);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int jj = j0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = i0 + (wm * TM + i) * 32 + c_row(r, lane);
        if (ii < g.M1 && jj < g.M2) P[(long)ii * g.M2 + jj] = acc[i][j][r];
      }
    }
}

template <int TM, int TN, int WM, int WN, bool TAPS = false, bool QUAD = false>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(WM * WN >= 8 ? 4 : 3)))
void gemm_nt_x3_kernel(NTParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  __shared__ pir_bf16x8 smem[2 * 3 * (2 * (BM + 4) + 2 * (BN + 4))];
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  gemm_nt_x3_body<TM, TN, WM, WN, TAPS, QUAD>(p, smem, lin, (int)gridDim.x, (int)gridDim.y, (int)gridDim.z);
}

// Several weight gradients of ONE transformer block in one launch (the low-resolution levels: four launches of 16 - 64
// output tiles each had to split the pixel axis 40 - 160 ways to fill the chip and wrote as many partial tiles; together
// they need a quarter of the splits).  The problems share the tile shape; each keeps its own workspace and reduction.
constexpr int NT_GROUP_MAX = 4;
struct NTGroup {
  NTParams p[NT_GROUP_MAX];
  int first[NT_GROUP_MAX + 1];   // first workgroup of every problem
  int ntiles[NT_GROUP_MAX];
  int n;
};

// (the 128 x 192 tile: one wave = 32 rows x 192 columns, 96 accumulator registers, two workgroups per CU)
template <int TM, int TN, int WM, int WN, bool QUAD>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(TM * TN >= 6 ? 2 : 3)))
void gemm_nt_x3_group_kernel(NTGroup grp) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  __shared__ pir_bf16x8 smem[2 * 3 * (2 * (BM + 4) + 2 * (BN + 4))];
  int k = 0;
  while (k + 1 < grp.n && (int)blockIdx.x >= grp.first[k + 1]) ++k;
  k = __builtin_amdgcn_readfirstlane(k);
  gemm_nt_x3_body<TM, TN, WM, WN, false, QUAD>(grp.p[k], smem, (int)blockIdx.x - grp.first[k], grp.ntiles[k], grp.p[k].splits, 1);
}

// second stage of every split-K product: reduce_batch.hip (launched at once, or queued inside a deferral scope)
}  // namespace
int pir_nt_reduce_submit(const float* ws, int splits, long O, int M1, int M2, float* G, long g_so, long g_si, long g_sj, long g_st,
                         float alpha, int accumulate, hipStream_t s);
namespace {

// tile configurations of gemm_nt: 0: 64x64 (4 k-slices)  1: 128x64 (2 k-slices)  2: 128x96 (2 k-slices)  3: 128x128
struct NTPlan {
  int cfg, bm, bn;
  int splits, chunks_per_r;
};

NTPlan nt_plan(int M1, int M2, int N, int O, int BR, int bk = NT_BK, bool wide_ok = false, bool taps = false) {
  NTPlan pl;
  if (taps && g_nt_cfg < 0 && M1 <= 64 && M2 > 64) {
    // dense 3x3 weight gradients with few channels on the row side (down1_2: 48 x 216 virtual rows, prompt1: 64 x 576):
    // 128-row tiles would be half padding
    pl.cfg = 6; pl.bm = 64; pl.bn = 128;
  } else if (g_nt_cfg >= 0) {
    pl.cfg = g_nt_cfg;
    const int c = (g_nt_cfg >= 4 && !(wide_ok && bk == X3_BK)) ? 3 : g_nt_cfg;   // the eight-wave tiles exist for the plain bf16x3 kernel only
    pl.cfg = c;
    pl.bm = c == 0 ? 64 : (c >= 4 ? 256 : 128);
    pl.bn = c == 0 || c == 1 ? 64 : (c == 2 || c == 4 ? 96 : 128);
  } else if (M1 <= 64 && M2 <= 64) { pl.cfg = 0; pl.bm = 64; pl.bn = 64; }
  else if (M2 <= 64) { pl.cfg = 1; pl.bm = 128; pl.bn = 64; }
  else if (bk == X3_BK && !taps && g_nt_tile96 && M1 <= 96 && M2 <= 96) {
    // 96 x 96 outputs (the per-image gram and dW_eff of the 96-channel levels, net/model.py:129,133 and adjoint): three
    // waves, no padded rows (the 128 x 96 tile loaded, split and multiplied 32 rows of nothing: a quarter of its work)
    pl.cfg = 7; pl.bm = 96; pl.bn = 96;
  }
  else {
    // (eight-wave 256 x 96 / 256 x 128 tiles exist behind knob 1 = 4 / 5: measured slower at every level, round 3)
    const long pad96 = pir_cdiv(M2, 96) * 96, pad128 = pir_cdiv(M2, 128) * 128;
    if (pad96 < pad128) { pl.cfg = 2; pl.bm = 128; pl.bn = 96; } else { pl.cfg = 3; pl.bm = 128; pl.bn = 128; }
  }
  const long tiles = pir_cdiv(M1, pl.bm) * pir_cdiv(M2, pl.bn) * O;
  pl.chunks_per_r = (int)pir_cdiv(N, bk);
  const long total = (long)BR * pl.chunks_per_r;
  long want = (long)g_nt_want_half * PIR_NUM_CU / (2 * tiles);   // all blocks resident at once (2.5 per CU), no second round
  if (want < 1) want = 1;
  const long min_stages = 128 / bk;                       // at least 128 pixels per split (sweeps at batch 8 and 32, tools/ktune.py)
  long max_by_work = total / min_stages > 0 ? total / min_stages : 1;
  long s = want < max_by_work ? want : max_by_work;
  if (g_nt_splits > 0) s = g_nt_splits < total ? g_nt_splits : total;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  pl.splits = (int)s;
  return pl;
}

template <int TM, int TN, int WM, int WN, int WK>
void launch_nt_cfg(const NTParams& p, dim3 grid, bool vec4, hipStream_t s) {
  if (vec4) hipLaunchKernelGGL((gemm_nt_kernel<TM, TN, WM, WN, WK, 4>), grid, dim3(WM * WN * WK * 64), 0, s, p);
  else hipLaunchKernelGGL((gemm_nt_kernel<TM, TN, WM, WN, WK, 1>), grid, dim3(WM * WN * WK * 64), 0, s, p);
}

// shared launcher of pir_gemm_nt and pir_conv3x3_wgrad: tile plan, split-K kernel, deterministic reduction
int launch_nt(NTParams& p, size_t ws_floats, int tap_sign, long g_st, hipStream_t s, int* partials_only = nullptr) {
  pir_gemm_nt_t& g = p.g;
  const int O = g.O1 * g.O2;
  auto al = [](const float* q, long s1, long s2, long sr, long ld) {
    return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && s1 % 4 == 0 && s2 % 4 == 0 && sr % 4 == 0 && ld % 4 == 0;
  };
  const bool vec4 = (g.H == 0 || tap_sign != 0) && g.N % 4 == 0 && al(g.X, g.x_s1, g.x_s2, g.x_sr, g.ldx) &&
                    al(g.Y, g.y_s1, g.y_s2, g.y_sr, g.ldy);
  const bool x3 = vec4 && (g_nt_x3 != 0 || tap_sign != 0);   // bf16x3 matrix-core path whenever rows are 16-byte aligned
  if (tap_sign != 0 && !x3) return PIR_EINVAL;
  if ((long)g.BR * pir_cdiv(g.N, x3 ? X3_BK : NT_BK) >= 2147483647L) return PIR_EINVAL;   // 32-bit stage counters
  NTPlan pl = nt_plan(g.M1, g.M2, g.N, O, g.BR, x3 ? X3_BK : NT_BK, x3 && tap_sign == 0, tap_sign != 0);
  int xp_splits = 0;
  if (x3 && tap_sign == 0 && g_nt_cfg < 0 && g_nt_splits == 0) {
    // tall x small weight gradients: the tall operand stays private to its wave, only the small one goes through LDS
    const int st = pir_nt_xp_launch(&g, &xp_splits, s);
    if (st != 1000 && st != PIR_OK) return st;
    if (st == 1000) xp_splits = 0;
  }
  if (xp_splits > 0) pl.splits = xp_splits;
  if ((size_t)pl.splits * O * g.M1 * g.M2 > ws_floats) return PIR_ENOMEM;
  p.splits = pl.splits;
  p.chunks_per_r = pl.chunks_per_r;
  p.tap_sign = tap_sign;
  p.magic_w = pir_magic(g.W > 0 ? (unsigned)g.W : 1u);
  dim3 grid((unsigned)(pir_cdiv(g.M1, pl.bm) * pir_cdiv(g.M2, pl.bn)), (unsigned)pl.splits, (unsigned)O);
  if (xp_splits > 0) {
    // partials already written by gemm_nt_xp_kernel
  } else if (tap_sign != 0) {
    switch (pl.cfg) {
      case 0: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 1, 2, 2, true>), grid, dim3(256), 0, s, p); break;
      case 1: hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 1, 2, 2, true>), grid, dim3(256), 0, s, p); break;
      case 2: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 3, 4, 1, true>), grid, dim3(256), 0, s, p); break;
      case 6: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 2, 2, 2, true>), grid, dim3(256), 0, s, p); break;   // 64 x 128
      default: hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 2, 2, 2, true>), grid, dim3(256), 0, s, p); break;
    }
  } else if (x3 && (g_nt_quad < 0 ? (g.N >= 1024 && pl.cfg != 1) : g_nt_quad != 0)) {   // sweep: tools/nt_quad_ab.py
    switch (pl.cfg) {
      case 0: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 1, 2, 2, false, true>), grid, dim3(256), 0, s, p); break;
      case 1: hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 1, 2, 2, false, true>), grid, dim3(256), 0, s, p); break;
      case 2: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 3, 4, 1, false, true>), grid, dim3(256), 0, s, p); break;
      case 4: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 3, 8, 1, false, true>), grid, dim3(512), 0, s, p); break;   // 256 x 96
      case 5: hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 2, 4, 2, false, true>), grid, dim3(512), 0, s, p); break;   // 256 x 128
      case 7: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 3, 3, 1, false, true>), grid, dim3(192), 0, s, p); break;   // 96 x 96
      default: hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 2, 2, 2, false, true>), grid, dim3(256), 0, s, p); break;
    }
  } else if (x3) {
    switch (pl.cfg) {
      case 0: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 1, 2, 2>), grid, dim3(256), 0, s, p); break;   // 64 x 64
      case 1: hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 1, 2, 2>), grid, dim3(256), 0, s, p); break;   // 128 x 64
      case 2: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 3, 4, 1>), grid, dim3(256), 0, s, p); break;   // 128 x 96
      case 4: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 3, 8, 1>), grid, dim3(512), 0, s, p); break;   // 256 x 96
      case 5: hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 2, 4, 2>), grid, dim3(512), 0, s, p); break;   // 256 x 128
      case 7: hipLaunchKernelGGL((gemm_nt_x3_kernel<1, 3, 3, 1>), grid, dim3(192), 0, s, p); break;   // 96 x 96
      default: hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 2, 2, 2>), grid, dim3(256), 0, s, p); break;  // 128 x 128
    }
  } else {
    switch (pl.cfg) {
      case 0: launch_nt_cfg<2, 2, 1, 1, 4>(p, grid, vec4, s); break;
      case 1: launch_nt_cfg<2, 2, 2, 1, 2>(p, grid, vec4, s); break;
      case 2: launch_nt_cfg<2, 3, 2, 1, 2>(p, grid, vec4, s); break;
      default: launch_nt_cfg<2, 2, 2, 2, 1>(p, grid, vec4, s); break;
    }
  }
  int st = pir_launch_status();
  if (st) return st;
  if (partials_only) { *partials_only = pl.splits; return PIR_OK; }   // the consumer sums the slices itself (pir_split_sum)
  return pir_nt_reduce_submit(g.ws, pl.splits, O, g.M1, g.M2, g.G, g.g_so, g.g_si, g.g_sj, g_st, g.alpha, g.accumulate, s);
}

}  // namespace

// second stage of a split-K product whose partials another kernel wrote (gemm_ntx.hip)
int pir_nt_reduce_launch(const float* ws, int splits, int M1, int M2, float* G, long g_so, long g_si, long g_sj, float alpha,
                         int accumulate, hipStream_t s) {
  return pir_nt_reduce_submit(ws, splits, 1, M1, M2, G, g_so, g_si, g_sj, 0L, alpha, accumulate, s);
}

int pir_gdfn_wave_tune(int knob, int value);   // gdfn_bwd.hip
int pir_stencil_wave_tune(int knob, int value);   // stencil_wave.hip
int pir_ln_tune(int knob, int value);             // norm.hip
int pir_gdfn_fused_tune(int knob, int value);     // gdfn_fused.hip
int pir_mdta_dqk_tune(int knob, int value);       // mdta_dqk.hip

extern "C" int pir_tune_set(int knob, int value) {
  switch (knob) {
    case 0: g_nn_cfg = value; return PIR_OK;
    case 1: g_nt_cfg = value; return PIR_OK;
    case 2: g_nt_splits = value; return PIR_OK;
    case 3: g_nn_x3 = value; return PIR_OK;
    case 4: g_nt_x3 = value; return PIR_OK;
    case 5: case 18: case 29: case 43: case 44: case 45: return pir_nn_x3_tune(knob, value);
    case 6: case 7: return pir_gdfn_wave_tune(knob, value);
    case 8: case 9: case 10: return pir_stencil_wave_tune(knob, value);
    case 13: case 16: return pir_ln_tune(knob, value);
    case 14: g_nt_quad = value; return PIR_OK;
    case 33: g_nt_tile96 = value; return PIR_OK;
    case 34: g_nt_group_wide = value; return PIR_OK;
    case 35: case 37: return pir_gdfn_fused_tune(knob, value);
    case 36: return pir_mdta_dqk_tune(knob, value);
    case 19: g_nt_want_half = value; return PIR_OK;
    case 20: return pir_nn_res_tune(knob, value);
    case 21: case 22: case 23: case 24: return pir_nn_res_tune2(knob, value);
    case 25: case 38: case 40: return pir_nt_xp_tune(knob, value);
    case 26: case 27: case 32: return pir_nn_cst_tune(knob, value);
    case 30: return pir_conv_rows_tune(knob, value);
    default: return PIR_EINVAL;
  }
}

extern "C" int pir_gemm_nn_plan(const pir_gemm_nn_t* a) {
  if (!a || a->M <= 0 || a->K <= 0 || a->N <= 0 || a->O1 <= 0 || a->O2 <= 0) return PIR_EINVAL;
  if (!pir_nn_x3_wanted(a, g_nn_x3)) return 0;
  if (g_nn_cfg < 0) {   // persistent kernels: 9000 resident weight panel, 9100 B-stationary, 9200 C-stationary
    if (pir_nn_cst_serves(a)) return 9200;
    const int kind = pir_nn_res_kind(a);
    if (kind) return kind == 2 ? 9100 : 9000;
  }
  return pir_nn_x3_plan(a, g_nn_cfg);
}

int pir_reduce_partials_now(const float* parts, long stride, int S, float alpha, int accumulate, float* out, long count,
                            pir_stream_t stream);   // reduce_batch.hip: launched at once, never queued

// which kernel family serves the call without a split: 0 tiled bf16x3 (the only one that splits over k)
static int nn_x3_tiled(const pir_gemm_nn_t* a) {
  if (!pir_nn_x3_wanted(a, g_nn_x3)) return 0;
  if (g_nn_cfg < 0 && (pir_nn_cst_serves(a) || pir_nn_res_kind(a))) return 0;
  return 1;
}

// scratch floats pir_gemm_nn_ws wants for this call (0: it would not split over k)
extern "C" size_t pir_gemm_nn_ws_floats(const pir_gemm_nn_t* a) {
  if (!a || a->M <= 0 || a->K <= 0 || a->N <= 0 || a->O1 <= 0 || a->O2 <= 0 || !nn_x3_tiled(a)) return 0;
  const int sp = pir_nn_x3_ksplit(a, g_nn_cfg);
  return sp > 1 ? (size_t)sp * a->O1 * a->M * a->N : 0;
}

static int gemm_nn_impl(const pir_gemm_nn_t* a, float* ws, size_t ws_floats, pir_stream_t stream) {
  PIR_CHECK_ARG(a && a->A && a->X && a->Y);
  PIR_CHECK_ARG(a->M > 0 && a->K > 0 && a->N > 0 && a->O1 > 0 && a->O2 > 0);
  PIR_CHECK_ARG((long)a->O1 * a->O2 <= 65535);
  PIR_CHECK_ARG(a->rowscale == nullptr || a->R != nullptr);
  // kernels index one batch item with 32-bit offsets
  PIR_CHECK_ARG((long)a->M * a->ldy < 2147483647L && (long)a->K * a->ldx < 2147483647L);
  PIR_CHECK_ARG(a->R == nullptr || (long)a->M * a->ldr < 2147483647L);
  PIR_CHECK_ARG((long)(a->M - 1) * a->a_sm + (long)(a->K - 1) * a->a_sk < 2147483647L);
  if (pir_nn_x3_wanted(a, g_nn_x3)) {
    if (g_nn_cfg < 0) {   // long pixel streams: the persistent kernels (gemm_cst.hip, gemm_res.hip)
      int st = pir_nn_cst_launch(a, (hipStream_t)stream);
      if (st != 1000) return st;
      st = pir_nn_res_launch(a, (hipStream_t)stream);
      if (st != 1000) return st;
    }
    // Underfilled deep-k product (the 384-row products of the 16^2 level: 32 - 128 workgroups walking 64 - 128 k-steps): the k
    // loop is cut into slices that run side by side; partial sums (residual and row scale in slice 0) go to the scratch
    // buffer and the deterministic second stage adds them in order.
    const int sp = ws ? pir_nn_x3_ksplit(a, g_nn_cfg) : 1;
    const long out_floats = (long)a->O1 * a->M * a->N;
    if (sp > 1 && (size_t)sp * out_floats <= ws_floats && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
      pir_gemm_nn_t gs = *a;
      gs.Y = ws;
      const int st = pir_nn_x3_launch(&gs, g_nn_cfg, (hipStream_t)stream, sp, out_floats);
      if (st) return st;
      return pir_reduce_partials_now(ws, out_floats, sp, 1.f, 0, a->Y, out_floats, stream);
    }
    return pir_nn_x3_launch(a, g_nn_cfg, (hipStream_t)stream);
  }
  NNParams p;
  p.g = *a;
  p.taps = 1; p.flip = 0; p.H = 0; p.W = 0; p.a_st = 0;
  return launch_nn(p, (hipStream_t)stream);
}

extern "C" int pir_gemm_nn(const pir_gemm_nn_t* a, pir_stream_t stream) { return gemm_nn_impl(a, nullptr, 0, stream); }

// pir_gemm_nn with a scratch buffer of pir_gemm_nn_ws_floats(a) floats: the same product, split over k where the launch would
// leave most CUs idle behind a long k loop (same result to fp32 rounding: another grouping of the same sum; deterministic)
extern "C" int pir_gemm_nn_ws(const pir_gemm_nn_t* a, float* ws, size_t ws_floats, pir_stream_t stream) {
  return gemm_nn_impl(a, ws, ws_floats, stream);
}

extern "C" int pir_conv3x3(const float* A, long a_st, long a_sm, long a_sk, int flip,
                           const float* X, long x_bs, float* Y, long y_bs,
                           const float* R, long r_bs,
                           int B, int M, int K, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(A && X && Y && B > 0 && M > 0 && K > 0 && H > 0 && W > 0 && B <= 65535);
  PIR_CHECK_ARG((long)(M > K ? M : K) * H * W < 2147483647L);
  NNParams p;
  pir_gemm_nn_t& g = p.g;
  g.A = A; g.a_s1 = 0; g.a_s2 = 0; g.a_sm = a_sm; g.a_sk = a_sk;
  g.X = X; g.x_s1 = x_bs; g.x_s2 = 0; g.ldx = (long)H * W;
  g.Y = Y; g.y_s1 = y_bs; g.y_s2 = 0; g.ldy = (long)H * W;
  g.R = R; g.r_s1 = r_bs; g.r_s2 = 0; g.ldr = (long)H * W;
  g.rowscale = nullptr; g.rs_s1 = 0; g.rs_s2 = 0;
  g.M = M; g.K = K; g.N = H * W; g.O1 = B; g.O2 = 1; g.A3 = nullptr; g.a3_kp = 0;
  p.taps = 9; p.flip = flip; p.H = H; p.W = W; p.a_st = a_st;
  return launch_nn(p, (hipStream_t)stream);
}

extern "C" size_t pir_gemm_nt_ws_floats(int M1, int M2, int N, int O, int BR) {
  if (M1 <= 0 || M2 <= 0 || N <= 0 || O <= 0 || BR <= 0) return 0;
  // the launcher may swap the operands so that the larger extent plays M1: size for the worst of both
  NTPlan a = nt_plan(M1, M2, N, O, BR), b = nt_plan(M2, M1, N, O, BR);
  NTPlan c = nt_plan(M1, M2, N, O, BR, X3_BK), d = nt_plan(M2, M1, N, O, BR, X3_BK);
  NTPlan e = nt_plan(M1, M2, N, O, BR, X3_BK, true), f = nt_plan(M2, M1, N, O, BR, X3_BK, true);
  int s = a.splits > b.splits ? a.splits : b.splits;
  if (c.splits > s) s = c.splits;
  if (d.splits > s) s = d.splits;
  if (e.splits > s) s = e.splits;
  if (f.splits > s) s = f.splits;
  if (g_nt_splits > s) s = g_nt_splits;
  if (O == 1 && s < 2 * PIR_NUM_CU) s = 2 * PIR_NUM_CU;   // gemm_nt_xp_kernel: up to two slices per CU
  return (size_t)s * O * M1 * M2;
}

// the larger extent on the tile rows (free output strides), unless a tap shift pins the second operand
static void nt_orient(pir_gemm_nt_t& g) {
  if (g.H == 0 && g.M2 > g.M1) {
    const float* t = g.X; g.X = g.Y; g.Y = t;
    long v;
    v = g.x_s1; g.x_s1 = g.y_s1; g.y_s1 = v;
    v = g.x_s2; g.x_s2 = g.y_s2; g.y_s2 = v;
    v = g.x_sr; g.x_sr = g.y_sr; g.y_sr = v;
    v = g.ldx; g.ldx = g.ldy; g.ldy = v;
    v = g.g_si; g.g_si = g.g_sj; g.g_sj = v;
    int m = g.M1; g.M1 = g.M2; g.M2 = m;
  }
}

// split count launch_nt would use for this call (mirrors its decisions; 0 = invalid arguments)
static int nt_splits_for(pir_gemm_nt_t g) {
  if (!(g.X && g.Y && g.M1 > 0 && g.M2 > 0 && g.N > 0 && g.O1 > 0 && g.O2 > 0 && g.BR > 0)) return 0;
  nt_orient(g);
  auto al = [](const float* q, long s1, long s2, long sr, long ld) {
    return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && s1 % 4 == 0 && s2 % 4 == 0 && sr % 4 == 0 && ld % 4 == 0;
  };
  const bool vec4 = g.H == 0 && g.N % 4 == 0 && al(g.X, g.x_s1, g.x_s2, g.x_sr, g.ldx) && al(g.Y, g.y_s1, g.y_s2, g.y_sr, g.ldy);
  const bool x3 = vec4 && g_nt_x3 != 0;
  const NTPlan pl = nt_plan(g.M1, g.M2, g.N, g.O1 * g.O2, g.BR, x3 ? X3_BK : NT_BK, x3, false);
  if (x3 && g_nt_cfg < 0 && g_nt_splits == 0) {
    g.ws_floats = (size_t)1 << 60;
    const int xs = pir_nt_xp_splits(&g);
    if (xs > 0) return xs;
  }
  return pl.splits;
}

// Workspace floats THIS call needs (its actual split count; pir_gemm_nt_ws_floats is the worst case over every plan).
// Callers that give every call a workspace piece of its own (deferred reductions) size the pieces with it.
extern "C" size_t pir_gemm_nt_ws_needed(const pir_gemm_nt_t* a) {
  if (!a) return 0;
  return (size_t)nt_splits_for(*a) * a->O1 * a->O2 * a->M1 * a->M2;
}

extern "C" int pir_gemm_nt(const pir_gemm_nt_t* a, pir_stream_t stream) {
  PIR_CHECK_ARG(a && a->X && a->Y && a->G && a->ws);
  PIR_CHECK_ARG(a->M1 > 0 && a->M2 > 0 && a->N > 0 && a->O1 > 0 && a->O2 > 0 && a->BR > 0);
  const int O = a->O1 * a->O2;
  PIR_CHECK_ARG(O <= 65535);
  PIR_CHECK_ARG(a->H == 0 || (long)a->H * a->W == a->N);
  NTParams p;
  p.g = *a;
  pir_gemm_nt_t& g = p.g;
  nt_orient(g);   // G^T[j][i] = sum_n Y[j][n] X[i][n]: the output strides are free, so the larger extent goes on M1 (tile rows)
  return launch_nt(p, a->ws_floats, 0, 0L, (hipStream_t)stream);
}

// The split-K product WITHOUT its second stage: ws[s][o][i][j], s < *splits, holds the partial sums (alpha, accumulate and
// the output strides are the consumer's business; `G` is not touched).  For consumers that read the partials themselves
// in the reduction's order (pir_mdta_softmax_fwd_parts / _bwd_parts).  M1 >= M2 (no operand swap), no tap shift.
extern "C" int pir_gemm_nt_partials(const pir_gemm_nt_t* a, int* splits, pir_stream_t stream) {
  PIR_CHECK_ARG(a && splits && a->X && a->Y && a->ws);
  PIR_CHECK_ARG(a->M1 > 0 && a->M2 > 0 && a->M1 >= a->M2 && a->N > 0 && a->O1 > 0 && a->O2 > 0 && a->BR > 0 && a->H == 0);
  PIR_CHECK_ARG(a->O1 * a->O2 <= 65535);
  NTParams p;
  p.g = *a;
  return launch_nt(p, a->ws_floats, 0, 0L, (hipStream_t)stream, splits);
}

// Up to four split-K products in ONE launch (the 1x1 weight gradients of a transformer block at the 32^2 / 16^2 levels,
// net/model.py:88,92,111,113): same tile shape for all, one split count for the whole launch chosen for the SUM of the
// tiles, every problem with its own workspace and its own (deferrable) second stage.  Problems the grouped kernel does
// not serve (unaligned rows, different tile shapes, the "X private" shapes) are launched one by one: the results are
// those of pir_gemm_nt either way, up to the order of the split-K sum.
// plan of a grouped launch: fills grp (oriented problems, tile counts), tile shape and the common split count; false = not served
static bool nt_group_plan(const pir_gemm_nt_t* probs, int n, NTGroup& grp, int& cfg, int& quad, long& sp) {
  auto al = [](const float* q, long s1, long s2, long sr, long ld) {
    return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && s1 % 4 == 0 && s2 % 4 == 0 && sr % 4 == 0 && ld % 4 == 0;
  };
  grp.n = 0;
  cfg = -1; quad = -1;
  long tiles_total = 0, total_chunks = -1;
  if (!(n > 1 && n <= NT_GROUP_MAX && g_nt_cfg < 0 && g_nt_splits == 0 && g_nt_x3 != 0)) return false;
  for (int k = 0; k < n; ++k) {
    const pir_gemm_nt_t* a = &probs[k];
    if (!(a->X && a->Y && a->M1 > 0 && a->M2 > 0 && a->N > 0 && a->BR > 0)) return false;
    NTParams& p = grp.p[k];
    p.g = *a;
    pir_gemm_nt_t& g = p.g;
    if (g.O1 * g.O2 != 1 || g.H != 0) return false;
    nt_orient(g);
    if (!(g.N % 4 == 0 && al(g.X, 0, 0, g.x_sr, g.ldx) && al(g.Y, 0, 0, g.y_sr, g.ldy))) return false;
    { pir_gemm_nt_t probe = g; probe.ws_floats = (size_t)1 << 60; if (pir_nt_xp_splits(&probe) > 0) return false; }
    NTPlan pl = nt_plan(g.M1, g.M2, g.N, 1, g.BR, X3_BK, true, false);
    const int q = (g.N >= 1024 && pl.cfg != 1) ? 1 : 0;
    if (pl.cfg != 2 && pl.cfg != 3) return false;
    // 128 x 192 tiles where every problem's columns come in 192s (the 192- and 384-channel levels): the conversion work
    // per MFMA of the 128 x 96 tile (VALU active 0.45 beside MFMA busy 0.39, profiles/r04_pmc_summary.txt) drops by a
    // third - each staged row meets twice as many columns
    if (g_nt_group_wide && g.M2 % 192 == 0) { pl.cfg = 8; pl.bm = 128; pl.bn = 192; }
    if (cfg < 0) { cfg = pl.cfg; quad = q; }
    if (pl.cfg != cfg || q != quad) return false;
    const long chunks = (long)g.BR * pl.chunks_per_r;
    if (total_chunks < 0) total_chunks = chunks;
    if (chunks != total_chunks || chunks >= 2147483647L) return false;
    p.chunks_per_r = pl.chunks_per_r;
    p.tap_sign = 0;
    p.magic_w = pir_magic(1u);
    grp.ntiles[k] = (int)(pir_cdiv(g.M1, pl.bm) * pir_cdiv(g.M2, pl.bn));
    tiles_total += grp.ntiles[k];
    grp.n = k + 1;
  }
  // one split count for the launch: all workgroups resident at once (as nt_plan), at least 128 pixels per split
  long want = (long)g_nt_want_half * PIR_NUM_CU / (2 * tiles_total);
  if (want < 1) want = 1;
  const long max_by_work = total_chunks / (128 / X3_BK) > 0 ? total_chunks / (128 / X3_BK) : 1;
  sp = want < max_by_work ? want : max_by_work;
  if (sp > 1024) sp = 1024;
  return true;
}

// workspace floats problem k of this group needs (its actual split count)
extern "C" size_t pir_gemm_nt_group_ws_needed(const pir_gemm_nt_t* probs, int n, int k) {
  if (!probs || k < 0 || k >= n) return 0;
  NTGroup grp; int cfg, quad; long sp;
  if (nt_group_plan(probs, n, grp, cfg, quad, sp)) return (size_t)sp * probs[k].M1 * probs[k].M2;
  return pir_gemm_nt_ws_needed(&probs[k]);
}

extern "C" int pir_gemm_nt_group(const pir_gemm_nt_t* probs, int n, pir_stream_t stream) {
  PIR_CHECK_ARG(probs && n > 0 && n <= NT_GROUP_MAX);
  hipStream_t s = (hipStream_t)stream;
  for (int k = 0; k < n; ++k) PIR_CHECK_ARG(probs[k].X && probs[k].Y && probs[k].G && probs[k].ws);
  NTGroup grp;
  int cfg = -1, quad = -1;
  long sp = 1;
  if (!nt_group_plan(probs, n, grp, cfg, quad, sp)) {
    for (int k = 0; k < n; ++k) {
      const int st = pir_gemm_nt(&probs[k], stream);
      if (st) return st;
    }
    return PIR_OK;
  }
  long blocks = 0;
  for (int k = 0; k < grp.n; ++k) {
    NTParams& p = grp.p[k];
    if ((size_t)sp * p.g.M1 * p.g.M2 > p.g.ws_floats) return PIR_ENOMEM;
    p.splits = (int)sp;
    grp.first[k] = (int)blocks;
    blocks += (long)grp.ntiles[k] * sp;
  }
  for (int k = grp.n; k <= NT_GROUP_MAX; ++k) grp.first[k] = (int)blocks;
  const dim3 grid((unsigned)blocks), block(256);
  if (cfg == 8 && quad) hipLaunchKernelGGL((gemm_nt_x3_group_kernel<1, 6, 4, 1, true>), grid, block, 0, s, grp);
  else if (cfg == 8) hipLaunchKernelGGL((gemm_nt_x3_group_kernel<1, 6, 4, 1, false>), grid, block, 0, s, grp);
  else if (cfg == 2 && quad) hipLaunchKernelGGL((gemm_nt_x3_group_kernel<1, 3, 4, 1, true>), grid, block, 0, s, grp);
  else if (cfg == 2) hipLaunchKernelGGL((gemm_nt_x3_group_kernel<1, 3, 4, 1, false>), grid, block, 0, s, grp);
  else if (quad) hipLaunchKernelGGL((gemm_nt_x3_group_kernel<2, 2, 2, 2, true>), grid, block, 0, s, grp);
  else hipLaunchKernelGGL((gemm_nt_x3_group_kernel<2, 2, 2, 2, false>), grid, block, 0, s, grp);
  int st = pir_launch_status();
  if (st) return st;
  for (int k = 0; k < grp.n; ++k) {
    const pir_gemm_nt_t& g = grp.p[k].g;
    st = pir_nt_reduce_submit(g.ws, (int)sp, 1, g.M1, g.M2, g.G, g.g_so, g.g_si, g.g_sj, 0L, g.alpha, g.accumulate, s);
    if (st) return st;
  }
  return PIR_OK;
}

// Dense 3x3 weight gradient in one launch: dW[co][ci][tap] = sum_{b,p} dy[b][co][p] * x[b][ci][p + s(tap)].
// The operand with fewer channels is expanded to 9 virtual (shifted) rows per channel inside the bf16x3
// gemm_nt kernel, so both tensors are read once from HBM instead of nine times.
extern "C" size_t pir_conv3x3_wgrad_ws_floats(int Cout, int Cin, int H, int W, int B) {
  if (Cout <= 0 || Cin <= 0 || H <= 0 || W <= 0 || B <= 0) return 0;
  const int big = Cout > Cin ? Cout : Cin, small = Cout > Cin ? Cin : Cout;
  NTPlan a = nt_plan(big, 9 * small, H * W, 1, B, X3_BK, false, true);
  const size_t fused = (size_t)a.splits * big * 9 * small;
  const size_t per_tap = pir_gemm_nt_ws_floats(Cout, Cin, H * W, 1, B);
  return fused > per_tap ? fused : per_tap;
}

extern "C" int pir_conv3x3_wgrad(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, int B, int Cout, int Cin,
                                 int H, int W, float* ws, size_t ws_floats, int accumulate, pir_stream_t stream) {
  PIR_CHECK_ARG(dy && x && dw && ws && B > 0 && Cout > 0 && Cin > 0 && H > 0 && W > 0);
  const int HW = H * W;
  NTParams p;
  pir_gemm_nt_t& g = p.g;
  g.O1 = 1; g.O2 = 1; g.BR = B; g.N = HW; g.H = H; g.W = W; g.shift_dh = 0; g.shift_dw = 0;
  g.x_s1 = g.x_s2 = g.y_s1 = g.y_s2 = 0; g.g_so = 0;
  g.G = dw; g.ws = ws; g.ws_floats = ws_floats; g.alpha = 1.f; g.accumulate = accumulate;
  const bool fusable = W % 8 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 &&
                       dy_bs % 4 == 0 && x_bs % 4 == 0;
  if (fusable) {
    long g_st = 0;
    int sign;
    if (Cin <= Cout) {   // rows: dy channels; virtual rows: (ci, tap) reading x[p + s]
      g.X = dy; g.x_sr = dy_bs; g.ldx = HW; g.M1 = Cout;
      g.Y = x; g.y_sr = x_bs; g.ldy = HW; g.M2 = 9 * Cin;
      g.g_si = (long)Cin * 9; g.g_sj = 1; sign = 1;
    } else {             // rows: x channels; virtual rows: (co, tap) reading dy[q - s]
      g.X = x; g.x_sr = x_bs; g.ldx = HW; g.M1 = Cin;
      g.Y = dy; g.y_sr = dy_bs; g.ldy = HW; g.M2 = 9 * Cout;
      g.g_si = 9; g.g_sj = (long)Cin * 9; g_st = 1; sign = -1;
    }
    return launch_nt(p, ws_floats, sign, g_st, (hipStream_t)stream);
  }
  // general fallback (W % 8 != 0 or unaligned planes): nine shifted gemm_nt launches
  for (int tap = 0; tap < 9; ++tap) {
    g.X = dy; g.x_sr = dy_bs; g.ldx = HW; g.M1 = Cout;
    g.Y = x; g.y_sr = x_bs; g.ldy = HW; g.M2 = Cin;
    g.G = dw + tap; g.g_si = (long)Cin * 9; g.g_sj = 9;
    g.shift_dh = tap / 3 - 1; g.shift_dw = tap % 3 - 1;
    const int st = launch_nt(p, ws_floats, 0, 0L, (hipStream_t)stream);
    if (st) return st;
  }
  return PIR_OK;
}
