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
#include "pir_common.h"

namespace {

constexpr int BK = 16;  // k-depth of one LDS stage (8 MFMA k-steps)

__device__ __forceinline__ int c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------
// gemm_nn
// ---------------------------------------------------------------------------------------------
struct NNParams {
  pir_gemm_nn_t g;
  // dense 3x3 mode (taps == 9): N == H*W, X rows are image planes
  int taps, flip, H, W;
  long a_st;
};

template <int TM, int TN, int WM, int WN, bool A_MFAST, int VEC, bool CONV>
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
  float ra[NA];
  float rb[NB][VEC];

  const int ktiles = (g.K + BK - 1) / BK;
  const int iters = ktiles * (CONV ? 9 : 1);

  auto load = [&](int it) {
    const int tap = CONV ? it / ktiles : 0;
    const int k0 = (CONV ? it % ktiles : it) * BK;
    const float* At = A + (CONV ? (long)(p.flip ? 8 - tap : tap) * p.a_st : 0);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * T;
      int kk, mm;
      if (A_MFAST) { mm = e % BM; kk = e / BM; } else { kk = e % BK; mm = e / BK; }
      const int k = k0 + kk, m = m0 + mm;
      float v = 0.f;
      if ((A_EL % T == 0 || e < A_EL) && k < g.K && m < g.M) v = At[(long)m * g.a_sm + (long)k * g.a_sk];
      ra[i] = v;
    }
    int dh = 0, dw = 0;
    if (CONV) { dh = tap / 3 - 1; dw = tap % 3 - 1; }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + i * T;
      const int kk = c / (BN / VEC), nn = (c % (BN / VEC)) * VEC;
      const int k = k0 + kk, n = n0 + nn;
      const bool rowok = (B_CH % T == 0 || c < B_CH) && k < g.K;
      if (!CONV) {
        if (VEC == 4) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (rowok && n < g.N) v = *reinterpret_cast<const f32x4*>(X + (long)k * g.ldx + n);  // N%4==0 here
          rb[i][0] = v[0]; rb[i][1] = v[1]; rb[i][2] = v[2]; rb[i][3] = v[3];
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) rb[i][j] = (rowok && n + j < g.N) ? X[(long)k * g.ldx + n + j] : 0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int nj = n + j;
          const int h = nj / p.W + dh, w = nj % p.W + dw;
          const bool ok = rowok && nj < g.N && h >= 0 && h < p.H && w >= 0 && w < p.W;
          rb[i][j] = ok ? X[(long)k * g.ldx + (long)h * p.W + w] : 0.f;
        }
      }
    }
  };

  auto stash = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BK * AS;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * T;
      int kk, mm;
      if (A_MFAST) { mm = e % BM; kk = e / BM; } else { kk = e % BK; mm = e / BK; }
      if (A_EL % T == 0 || e < A_EL) As[kk * AS + mm] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + i * T;
      const int kk = c / (BN / VEC), nn = (c % (BN / VEC)) * VEC;
      if (B_CH % T == 0 || c < B_CH) {
        if (VEC == 4) {
          f32x4 v = {rb[i][0], rb[i][1], rb[i][2], rb[i][3]};
          *reinterpret_cast<f32x4*>(Bs + kk * BS + nn) = v;
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) Bs[kk * BS + nn + j] = rb[i][j];
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

  load(0);
  stash(0);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (it + 1 < iters) load(it + 1);
    const float* As = smem + buf * STAGE + (lane >> 5) * AS + wm * TM * 32 + (lane & 31);
    const float* Bs = smem + buf * STAGE + BK * AS + (lane >> 5) * BS + wn * TN * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kk * AS + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kk * BS + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < iters) stash(buf ^ 1);
    __syncthreads();
  }

  // epilogue: each store instruction writes two 128-byte row segments
  const float* __restrict__ R = g.R ? g.R + o1 * g.r_s1 + o2 * g.r_s2 : nullptr;
  const float* __restrict__ RS = g.rowscale ? g.rowscale + o1 * g.rs_s1 + o2 * g.rs_s2 : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + c_row(r, lane);
        if (m < g.M && n < g.N) {
          float v = acc[i][j][r];
          if (R) v += (RS ? RS[m] : 1.f) * R[(long)m * g.ldr + n];
          Y[(long)m * g.ldy + n] = v;
        }
      }
    }
}

template <int TM, int TN, int WM, int WN>
int launch_nn_cfg(const NNParams& p, hipStream_t s) {
  const pir_gemm_nn_t& g = p.g;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const long tiles = pir_cdiv(g.M, BM) * pir_cdiv(g.N, BN);
  dim3 grid((unsigned)tiles, (unsigned)(g.O1 * g.O2)), block(WM * WN * 64);
  const bool conv = p.taps == 9;
  const bool mfast = g.a_sm == 1;
  const bool vec4 = !conv && g.N % 4 == 0 && g.ldx % 4 == 0 && g.x_s1 % 4 == 0 && g.x_s2 % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(g.X) & 15) == 0;
#define PIR_NN_LAUNCH(MF, V, C) \
  hipLaunchKernelGGL((gemm_nn_kernel<TM, TN, WM, WN, MF, V, C>), grid, block, 0, s, p)
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
  // pick the M-tile height (multiple of 32) that wastes the fewest padded rows
  if (M <= 32) return launch_nn_cfg<1, 2, 1, 4>(p, s);   // 32 x 256
  if (M <= 64) return launch_nn_cfg<2, 2, 1, 4>(p, s);   // 64 x 256
  const long pad96 = pir_cdiv(M, 96) * 96, pad128 = pir_cdiv(M, 128) * 128;
  if (pad96 < pad128) return launch_nn_cfg<3, 2, 1, 4>(p, s);  // 96 x 256
  return launch_nn_cfg<2, 2, 2, 2>(p, s);                       // 128 x 128
}

// ---------------------------------------------------------------------------------------------
// gemm_nt (split-K over pixels, deterministic two-stage reduction)
// ---------------------------------------------------------------------------------------------
constexpr int NT_BK = 32;  // pixels per LDS stage

struct NTParams {
  pir_gemm_nt_t g;
  int splits;       // split-K factor over the flattened (r, n-chunk) axis
  int chunks_per_r; // ceil(N / NT_BK)
};

// WK waves of a block share an output tile and each takes a quarter of every stage's k-range.
template <int TM, int TN, int WM, int WN, int WK>
__global__ __launch_bounds__(WM* WN* WK * 64) void gemm_nt_kernel(NTParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, T = WM * WN * WK * 64;
  constexpr int XS = BM + 1, YS = BN + 1;  // odd strides: conflict-free transposing stores
  constexpr int STAGE = NT_BK * (XS + YS);
  constexpr int RED = (WK > 1) ? (WK - 1) * TM * TN * 16 * 64 : 0;
  constexpr int SMEM = (2 * STAGE > RED) ? 2 * STAGE : RED;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const pir_gemm_nt_t& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid % WK, wmn = wid / WK, wm = wmn / WN, wn = wmn % WN;
  const int tiles_i = (g.M1 + BM - 1) / BM;
  const int i0 = (blockIdx.x % tiles_i) * BM, j0 = (blockIdx.x / tiles_i) * BN;
  const int split = blockIdx.y;
  const int o = blockIdx.z, o1 = o / g.O2, o2 = o % g.O2;

  const float* __restrict__ Xb = g.X + o1 * g.x_s1 + o2 * g.x_s2;
  const float* __restrict__ Yb = g.Y + o1 * g.y_s1 + o2 * g.y_s2;

  // this block's slice of the flattened (r, chunk) axis
  const long total = (long)g.BR * p.chunks_per_r;
  const long per = (total + p.splits - 1) / p.splits;
  const long c_begin = split * per, c_end = (c_begin + per < total) ? c_begin + per : total;

  constexpr int X_EL = NT_BK * BM, NX = (X_EL + T - 1) / T;
  constexpr int Y_EL = NT_BK * BN, NY = (Y_EL + T - 1) / T;
  float rx[NX], ry[NY];

  auto load = [&](long c) {
    const int r = (int)(c / p.chunks_per_r);
    const int nb = (int)(c % p.chunks_per_r) * NT_BK;
    const float* Xp = Xb + r * g.x_sr;
    const float* Yp = Yb + r * g.y_sr;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int e = tid + q * T;
      const int nn = e % NT_BK, ii = e / NT_BK;
      const int n = nb + nn, i = i0 + ii;
      rx[q] = ((X_EL % T == 0 || e < X_EL) && n < g.N && i < g.M1) ? Xp[(long)i * g.ldx + n] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NY; ++q) {
      const int e = tid + q * T;
      const int nn = e % NT_BK, jj = e / NT_BK;
      const int n = nb + nn, j = j0 + jj;
      bool ok = (Y_EL % T == 0 || e < Y_EL) && n < g.N && j < g.M2;
      long off = (long)j * g.ldy + n;
      if (g.H > 0) {  // 3x3 tap shift on the Y operand (dense 3x3 weight gradient)
        const int h = n / g.W + g.shift_dh, w = n % g.W + g.shift_dw;
        ok = ok && h >= 0 && h < g.H && w >= 0 && w < g.W;
        off = (long)j * g.ldy + (long)h * g.W + w;
      }
      ry[q] = ok ? Yp[off] : 0.f;
    }
  };
  auto stash = [&](int buf) {
    float* Xs = smem + buf * STAGE;
    float* Ys = Xs + NT_BK * XS;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int e = tid + q * T;
      if (X_EL % T == 0 || e < X_EL) Xs[(e % NT_BK) * XS + e / NT_BK] = rx[q];
    }
#pragma unroll
    for (int q = 0; q < NY; ++q) {
      const int e = tid + q * T;
      if (Y_EL % T == 0 || e < Y_EL) Ys[(e % NT_BK) * YS + e / NT_BK] = ry[q];
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
    load(c_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (long c = c_begin; c < c_end; ++c, buf ^= 1) {
      if (c + 1 < c_end) load(c + 1);
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

  // cross-wave (WK) reduction through LDS, fixed order -> deterministic
  if (WK > 1) {
    __syncthreads();
    if (wk > 0) {
      float* dst = smem + ((wk - 1) * TM * TN * 16) * 64;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll 1
      for (int w = 1; w < WK; ++w) {
        const float* src = smem + ((w - 1) * TM * TN * 16) * 64;
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

// G[o][i][j] (strided) = alpha * sum_s ws[s][o][i][j]  (+ G)
__global__ void nt_reduce_kernel(const float* __restrict__ ws, int splits, long per_split, int M1, int M2,
                                 float* __restrict__ G, long g_so, long g_si, long g_sj, long total,
                                 float alpha, int accumulate) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[k * per_split + e];
    const long o = e / ((long)M1 * M2);
    const long ij = e % ((long)M1 * M2);
    const long i = ij / M2, j = ij % M2;
    float* dst = G + o * g_so + i * g_si + j * g_sj;
    const float v = alpha * s;
    *dst = accumulate ? *dst + v : v;
  }
}

struct NTPlan {
  bool small;
  int splits, chunks_per_r;
};

NTPlan nt_plan(int M1, int M2, int N, int O, int BR) {
  NTPlan pl;
  pl.small = (long)M1 * M2 <= 96L * 96L || M1 <= 64 || M2 <= 64;
  const int bm = pl.small ? 64 : 128;
  const long tiles = pir_cdiv(M1, bm) * pir_cdiv(M2, bm) * O;
  pl.chunks_per_r = (int)pir_cdiv(N, NT_BK);
  const long total = (long)BR * pl.chunks_per_r;
  long want = pir_cdiv(4L * PIR_NUM_CU, tiles);        // ~4 blocks per CU overall
  long max_by_work = total / 8 > 0 ? total / 8 : 1;      // at least 8 stages (256 pixels) per split
  long s = want < max_by_work ? want : max_by_work;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  pl.splits = (int)s;
  return pl;
}

}  // namespace

extern "C" int pir_gemm_nn(const pir_gemm_nn_t* a, pir_stream_t stream) {
  PIR_CHECK_ARG(a && a->A && a->X && a->Y);
  PIR_CHECK_ARG(a->M > 0 && a->K > 0 && a->N > 0 && a->O1 > 0 && a->O2 > 0);
  PIR_CHECK_ARG((long)a->O1 * a->O2 <= 65535);
  PIR_CHECK_ARG(a->rowscale == nullptr || a->R != nullptr);
  NNParams p;
  p.g = *a;
  p.taps = 1; p.flip = 0; p.H = 0; p.W = 0; p.a_st = 0;
  return launch_nn(p, (hipStream_t)stream);
}

extern "C" int pir_conv3x3(const float* A, long a_st, long a_sm, long a_sk, int flip,
                           const float* X, long x_bs, float* Y, long y_bs,
                           const float* R, long r_bs,
                           int B, int M, int K, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(A && X && Y && B > 0 && M > 0 && K > 0 && H > 0 && W > 0 && B <= 65535);
  NNParams p;
  pir_gemm_nn_t& g = p.g;
  g.A = A; g.a_s1 = 0; g.a_s2 = 0; g.a_sm = a_sm; g.a_sk = a_sk;
  g.X = X; g.x_s1 = x_bs; g.x_s2 = 0; g.ldx = (long)H * W;
  g.Y = Y; g.y_s1 = y_bs; g.y_s2 = 0; g.ldy = (long)H * W;
  g.R = R; g.r_s1 = r_bs; g.r_s2 = 0; g.ldr = (long)H * W;
  g.rowscale = nullptr; g.rs_s1 = 0; g.rs_s2 = 0;
  g.M = M; g.K = K; g.N = H * W; g.O1 = B; g.O2 = 1;
  p.taps = 9; p.flip = flip; p.H = H; p.W = W; p.a_st = a_st;
  return launch_nn(p, (hipStream_t)stream);
}

extern "C" size_t pir_gemm_nt_ws_floats(int M1, int M2, int N, int O, int BR) {
  if (M1 <= 0 || M2 <= 0 || N <= 0 || O <= 0 || BR <= 0) return 0;
  NTPlan pl = nt_plan(M1, M2, N, O, BR);
  return (size_t)pl.splits * O * M1 * M2;
}

extern "C" int pir_gemm_nt(const pir_gemm_nt_t* a, pir_stream_t stream) {
  PIR_CHECK_ARG(a && a->X && a->Y && a->G && a->ws);
  PIR_CHECK_ARG(a->M1 > 0 && a->M2 > 0 && a->N > 0 && a->O1 > 0 && a->O2 > 0 && a->BR > 0);
  const int O = a->O1 * a->O2;
  PIR_CHECK_ARG(O <= 65535);
  PIR_CHECK_ARG(a->H == 0 || (long)a->H * a->W == a->N);
  NTPlan pl = nt_plan(a->M1, a->M2, a->N, O, a->BR);
  if ((size_t)pl.splits * O * a->M1 * a->M2 > a->ws_floats) return PIR_ENOMEM;
  NTParams p;
  p.g = *a;
  p.splits = pl.splits;
  p.chunks_per_r = pl.chunks_per_r;
  hipStream_t s = (hipStream_t)stream;
  if (pl.small) {
    dim3 grid((unsigned)(pir_cdiv(a->M1, 64) * pir_cdiv(a->M2, 64)), (unsigned)pl.splits, (unsigned)O);
    hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 1, 1, 4>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid((unsigned)(pir_cdiv(a->M1, 128) * pir_cdiv(a->M2, 128)), (unsigned)pl.splits, (unsigned)O);
    hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 2, 2, 1>), grid, dim3(256), 0, s, p);
  }
  int st = pir_launch_status();
  if (st) return st;
  const long per_split = (long)O * a->M1 * a->M2;
  const int blocks = (int)(pir_cdiv(per_split, 256) < 2048 ? pir_cdiv(per_split, 256) : 2048);
  hipLaunchKernelGGL(nt_reduce_kernel, dim3(blocks), dim3(256), 0, s, a->ws, pl.splits, per_split, a->M1, a->M2,
                     a->G, a->g_so, a->g_si, a->g_sj, per_split, a->alpha, a->accumulate);
  return pir_launch_status();
}
