// Shared pieces of the gemm_nn kernels (fp32 MFMA and bf16x3 split MFMA): C/D fragment map and epilogue.
#pragma once
#include "pir_common.h"

// row of accumulator register `reg` of a 32x32 MFMA result (col = lane & 31); same for every dtype on gfx950
__device__ __forceinline__ int pir_c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <int TM, int TN>
__device__ __forceinline__ void pir_nn_epilogue(const f32x16 (&acc)[TM][TN], const pir_gemm_nn_t& g, float* __restrict__ Y,
                                                int o1, int o2, int m0, int n0, int wm, int wn, int lane) {
  // epilogue.  Each store instruction writes two 128-byte row segments.  Offsets are 32-bit
  // (per-image tensors are < 2^31 elements, checked on the host) = one VALU add per element on top
  // of scalar row strides; a wave-uniform test selects an unguarded path for interior tiles.
  // Residual loads of one 32x32 tile are issued back-to-back so their latencies overlap.
  const float* __restrict__ R = g.R ? g.R + o1 * g.r_s1 + o2 * g.r_s2 : nullptr;
  const float* __restrict__ RS = g.rowscale ? g.rowscale + o1 * g.rs_s1 + o2 * g.rs_s2 : nullptr;
  const int ldy = (int)g.ldy, ldr = (int)g.ldr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + (wn * TN + j) * 32, mb = m0 + (wm * TM + i) * 32;  // wave-uniform
      const int n = nb + (lane & 31), mrow = mb + 4 * (lane >> 5);
      const bool full = mb + 32 <= g.M && nb + 32 <= g.N;
      if (full) {
        const int offy = mrow * ldy + n;
        float res[16];
        if (R) {
          const int offr = mrow * ldr + n;
#pragma unroll
          for (int r = 0; r < 16; ++r) res[r] = R[offr + ((r & 3) + 8 * (r >> 2)) * ldr];
          if (RS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] *= RS[mrow + (r & 3) + 8 * (r >> 2)];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) Y[offy + ((r & 3) + 8 * (r >> 2)) * ldy] = acc[i][j][r] + res[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) Y[offy + ((r & 3) + 8 * (r >> 2)) * ldy] = acc[i][j][r];
        }
      } else {
        const int nc = n < g.N ? n : g.N - 1;
        float res[16];
        if (R) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow + (r & 3) + 8 * (r >> 2);
            const int mc = m < g.M ? m : g.M - 1;
            res[r] = R[mc * ldr + nc] * (RS ? RS[mc] : 1.f);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mrow + (r & 3) + 8 * (r >> 2);
          if (m < g.M && n < g.N) Y[m * ldy + n] = R ? acc[i][j][r] + res[r] : acc[i][j][r];
        }
      }
    }
}

// ---- bf16x3 split helpers (shared by gemm_x3.hip and the split-K gemm_nt in gemm.hip)
typedef __bf16 pir_bf16x8 __attribute__((ext_vector_type(8)));
struct pir_frag3 { pir_bf16x8 hi, mid, lo; };

// exact three-way split of 8 fp32 values: x = hi + mid + lo (+ <= 2^-27 |x|), every subtraction exact
__device__ __forceinline__ pir_frag3 pir_split8(const float (&v)[8]) {
  pir_frag3 f;
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

// the same split for 4 values (8 bytes per piece)
typedef __bf16 pir_bf16x4 __attribute__((ext_vector_type(4)));
struct pir_frag3h { pir_bf16x4 hi, mid, lo; };
__device__ __forceinline__ pir_frag3h pir_split4(const float (&v)[4]) {
  pir_frag3h f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x = v[j];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    f.hi[j] = h; f.mid[j] = m; f.lo[j] = (__bf16)r2;
  }
  return f;
}

// six-term product block: acc += (hi+mid+lo)_a x (hi+mid+lo)_b without the three <= 2^-27 terms
__device__ __forceinline__ f32x16 pir_mfma_x3(const pir_bf16x8& ah, const pir_bf16x8& am, const pir_bf16x8& al,
                                              const pir_bf16x8& bh, const pir_bf16x8& bm, const pir_bf16x8& bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  return c;
}

// bf16x3 split path (gemm_x3.hip): exact-fp32-class results from six bf16 MFMAs per product block.
bool pir_nn_x3_wanted(const pir_gemm_nn_t* a, int knob);
int pir_nn_x3_launch(const pir_gemm_nn_t* a, int cfg, hipStream_t stream, int ksplit = 1, long part_stride = 0);
int pir_nn_x3_ksplit(const pir_gemm_nn_t* a, int cfg);
int pir_nn_x3_plan(const pir_gemm_nn_t* a, int cfg);
int pir_nn_x3_tune(int knob, int value);

// resident-weight-panel persistent kernel (gemm_res.hip): 1000 = shape not served
int pir_nn_res_kind(const pir_gemm_nn_t* a);   // 0 not served, 1 resident-panel kernel, 2 B-stationary kernel
int pir_nn_res_launch(const pir_gemm_nn_t* a, hipStream_t stream);
int pir_nn_res_tune(int knob, int value);
int pir_nn_res_tune2(int knob, int value);

// dense 3x3 convolution with whole image rows per tile (conv_rows.hip): 1000 = shape not served
int pir_conv_rows_launch(const pir_gemm_nn_t* g, int H, int W, int tile, hipStream_t stream, int splits = 1, long part_stride = 0);
bool pir_conv_rows_serves(const pir_gemm_nn_t* g, int W, int tile);
int pir_conv_rows_tune(int knob, int value);

// C-stationary persistent kernel for few rows against a long k (gemm_cst.hip): 1000 = shape not served
bool pir_nn_cst_serves(const pir_gemm_nn_t* a);
int pir_nn_cst_launch(const pir_gemm_nn_t* a, hipStream_t stream);
int pir_nn_cst_tune(int knob, int value);

// tall x small weight gradients with the tall operand private to its wave (gemm_ntx.hip): 1000 = shape not served
int pir_nt_xp_launch(const pir_gemm_nt_t* a, int* splits, hipStream_t stream, const float* y_mean = nullptr, const float* y_rstd = nullptr,
                     const float* y_gamma = nullptr, const float* y_beta = nullptr);   // y_*: LayerNorm applied to Y as it is staged
int pir_nt_xp_splits(const pir_gemm_nt_t* a);
int pir_nt_xp_tune(int knob, int value);
