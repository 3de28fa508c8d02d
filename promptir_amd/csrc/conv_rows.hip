// Dense 3x3 convolution on the bf16x3 matrix cores, whole image rows per tile (gfx950).
//
//   Y[b][m][p] = sum_{tap, k} W(tap, m, k) X[b][k][p + s(tap)]  (+ R)        (contract of pir_conv3x3_x3; zero padding)
//
// gemm_nn_x3_kernel<CONV> runs the nine taps as nine k ranges: every tap loads its own shifted copy of the activation
// tile with 4-byte loads and converts it again - nine passes over the input through the vector-memory path, which is
// what bounds it (round-3 counters: TA busy 0.82, MFMA busy 0.36; tools/conv3x3_bench.py: 0.10-0.45 of the roofline).
// Here a tile is BN = 128 or 256 pixels of WHOLE image rows, and a stage is (row shift dy, 16 input channels):
//   * the stage's activations are loaded ONCE, with 16-byte loads (a row shift keeps the alignment), transposed in
//     registers (wide_tiles.h), split to bf16x3 once and written to LDS with one zero column on either side of every image
//     row - the three horizontal taps are then the SAME fragments read one column to the left / right, the padding
//     columns supplying the zeros of the image border: a third of the loads, a twelfth of the load instructions, a
//     third of the conversion work;
//   * the pre-split weights of the stage's three taps go through LDS as before; 3 x 6 x TM x TN MFMAs per barrier pair.
// Per (m, p) the taps are summed in the order (dy, k, dx) instead of (tap, k): results agree with the nine-pass kernel to
// fp32 rounding, not bit for bit.
#include "gemm_common.h"
#include "wide_tiles.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct RowsArgs {
  pir_gemm_nn_t g;
  int H, W, wshift;   // W = 1 << wshift
  int ksteps;         // a3_kp / 16
  // split over the stages (row shift, k-step): gridDim.z workgroups share an output tile, workgroup z multiplies stages
  // [z per, (z + 1) per) and writes its partial sums to `Y + z * part_stride` (the residual goes into slice 0); the caller
  // adds the slices in order (pir_reduce_partials).  For launches that leave most CUs idle behind k loops of 72 - 432 stages.
  int splits; long part_stride;
};

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(2)))
void conv3x3_rows_kernel(RowsArgs p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, T = WM * WN * 64, NWV = WM * WN;
  constexpr int NBLK = BN / 32, BPW = NBLK / NWV;       // 32-column blocks staged per wave and stage
  static_assert(NBLK % NWV == 0, "blocks per wave");
  constexpr int BNP = BN + 2 * (BN / 16);               // columns + two padding columns per image row (W >= 16)
  constexpr int AKS = BM + 4;                           // 16-byte units between the two k-groups of A (+4: bank shift)
  constexpr int SB = 6 * BNP;                           // [part][k-group][padded column]
  constexpr int SA = 18 * AKS;                          // [dx][part][k-group][row]
  constexpr int AUN = 18 * BM, NLA = (AUN + T - 1) / T; // weight units per stage / per thread
  // two LDS buffers (one barrier per stage, the next stage's conversion in the shadow of this stage's MFMAs) for the
  // 64 x 128 tile, which serves the underfilled low-resolution launches (one workgroup per CU anyway: -8 %); the other tiles
  // keep one buffer and two barriers (the 32 x 256 tile measured 10-15 % slower with two: fewer workgroups per CU)
  constexpr bool DB = WM == 2 && WN == 2 && TM == 1;
  __shared__ bf16x8 sB[(DB ? 2 : 1) * SB];
  __shared__ bf16x8 sA[(DB ? 2 : 1) * SA];
  const pir_gemm_nn_t& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int wg = pir_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg % tiles_m) * BM, n0 = (wg / tiles_m) * BN;
  const int o1 = blockIdx.y;
  const int zs = blockIdx.z;
  const float* __restrict__ X = g.X + o1 * g.x_s1;
  float* __restrict__ Y = g.Y + o1 * g.y_s1 + zs * p.part_stride;
  const int h = lane >> 5, r = lane & 31;
  const int qk = ((r >> 4) << 2) | (r & 3), qj = (r >> 2) & 3;   // lane -> (pixel quad, row in the quad group): gemm_res.hip
  const int ldx4 = (int)g.ldx * 4;
  const __amdgpu_buffer_rsrc_t xrs = pir_make_rsrc(X, (unsigned)((((long)g.K - 1) * g.ldx + g.N) * 4));
  const __amdgpu_buffer_rsrc_t ars = pir_make_rsrc(g.A3, (unsigned)(54L * g.M * g.a3_kp));
  const int a3_part_bytes = 9 * g.M * g.a3_kp * 2, a3_step_bytes = g.M * 32;

  // the padding columns (and everything else) of the activation buffer start as zeros and are never written again
  {
    const bf16x8 z = {};
    for (int u = tid; u < (DB ? 2 : 1) * SB; u += T) sB[u] = z;
  }

  // ---- activations: block i of this wave; lane (h, qj, qk) loads rows 16 ks + 8 h + 4 t + qj, pixels 4 qk .. 4 qk + 3
  int b_vo[BPW], b_y[BPW], b_dst[BPW];
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    const int c0 = 32 * (wid * BPW + i);
    const int cq = n0 + c0 + 4 * qk;                      // first pixel of the lane's quad
    b_y[i] = cq >> p.wshift;                              // its image row (a quad never straddles rows: W % 4 == 0)
    b_vo[i] = ((8 * h + qj) * (int)g.ldx + cq) * 4;
    const int cs = c0 + 4 * qk + qj;                      // the column this lane holds after the transpose
    b_dst[i] = h * BNP + cs + 2 * (cs >> p.wshift) + 1;
  }
  auto load_b = [&](int dy, int ks, f32x4 (&raw)[BPW][2]) {
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      // a row shift that leaves the image gets an out-of-range offset: the loads return 0
      const int vo = (unsigned)(b_y[i] + dy) < (unsigned)p.H ? b_vo[i] + dy * p.W * 4 : 0x7ffffff0;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        raw[i][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo + 4 * t * ldx4, ks * 16 * ldx4, 0));
    }
  };
  auto stash_b = [&](f32x4 (&raw)[BPW][2], int buf) {
    bf16x8* sB_ = sB + buf * SB;
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      float v[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float a0 = raw[i][t][0], a1 = raw[i][t][1], a2 = raw[i][t][2], a3 = raw[i][t][3];
        res_transpose4(a0, a1, a2, a3);
        v[4 * t] = a0; v[4 * t + 1] = a1; v[4 * t + 2] = a2; v[4 * t + 3] = a3;
      }
      const pir_frag3 fr = pir_split8(v);                 // channels beyond K are zeros (range-checked loads)
      sB_[b_dst[i]] = fr.hi; sB_[2 * BNP + b_dst[i]] = fr.mid; sB_[4 * BNP + b_dst[i]] = fr.lo;
    }
  };

  // ---- weights of the stage's three taps: unit u = (dx, part, row, k-group), rows contiguous in memory per (part, tap, k-step)
  int a_vo[NLA], a_dst[NLA];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int u0 = tid + i * T, u = u0 < AUN ? u0 : 0;
    const int kg = u & 1, row = (u >> 1) % BM, rest = (u >> 1) / BM;
    const int part = rest % 3, dxi = rest / 3;
    const int m = m0 + row, mc = m < g.M ? m : g.M - 1;   // rows beyond M only feed masked outputs
    a_vo[i] = (mc * 16 + 8 * kg) * 2 + part * a3_part_bytes + dxi * p.ksteps * a3_step_bytes;
    a_dst[i] = u0 < AUN ? ((dxi * 3 + part) * 2 + kg) * AKS + row : -1;
  }
  auto load_a = [&](int dyi, int ks, bf16x8 (&a)[NLA]) {
    const int so = (dyi * 3 * p.ksteps + ks) * a3_step_bytes;
#pragma unroll
    for (int i = 0; i < NLA; ++i) a[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, a_vo[i], so, 0));
  };
  auto stash_a = [&](const bf16x8 (&a)[NLA], int buf) {
#pragma unroll
    for (int i = 0; i < NLA; ++i)
      if (AUN % T == 0 || a_dst[i] >= 0) sA[buf * SA + a_dst[i]] = a[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  int colp[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c = wn * TN * 32 + j * 32 + r;
    colp[j] = h * BNP + c + 2 * (c >> p.wshift) + 1;
  }
  auto compute = [&](int buf) {
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const bf16x8* ap = sA + buf * SA + dxi * 6 * AKS + h * AKS + wm * TM * 32 + r;
      bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) { ah[i] = ap[i * 32]; am[i] = ap[2 * AKS + i * 32]; al[i] = ap[4 * AKS + i * 32]; }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const bf16x8* bp = sB + buf * SB + colp[j] + (dxi - 1);   // the horizontal tap: the neighbouring column (or a zero column)
        bh[j] = bp[0]; bm[j] = bp[2 * BNP]; bl[j] = bp[4 * BNP];
      }
#define PIR_ROWS_TERM(A_, B_)                                                                 \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_[j], acc[i][j], 0, 0, 0);
      PIR_ROWS_TERM(al, bh)
      PIR_ROWS_TERM(ah, bl)
      PIR_ROWS_TERM(am, bm)
      PIR_ROWS_TERM(am, bh)
      PIR_ROWS_TERM(ah, bm)
      PIR_ROWS_TERM(ah, bh)
#undef PIR_ROWS_TERM
    }
  };

  // stages s = (dy + 1) * ksteps + ks; the next stage's loads fly while this one is multiplied
  const int stages_all = 3 * p.ksteps;
  const int per = (stages_all + p.splits - 1) / p.splits;
  const int s_lo = zs * per < stages_all ? zs * per : stages_all, s_hi = s_lo + per < stages_all ? s_lo + per : stages_all;
  const int stages = s_hi - s_lo;                         // this workgroup's share (all of them without a split)
  f32x4 raw[BPW][2];
  bf16x8 areg[NLA];
  int dyi = s_lo / p.ksteps, ks = s_lo - dyi * p.ksteps;
  if (dyi > 2) { dyi = 2; ks = 0; }                       // (an empty share: loads stay in range, nothing is multiplied)
  auto advance = [&](int s) {      // (dyi, ks) -> the stage after local stage s (behind the last one: its own again, unused)
    if (s + 1 >= stages) return;
    if (++ks == p.ksteps) { ks = 0; ++dyi; }
  };
  load_b(dyi - 1, ks, raw);
  load_a(dyi, ks, areg);
  __syncthreads();                                        // the zero fill is complete
  if constexpr (DB) {
    stash_b(raw, 0);
    stash_a(areg, 0);
    advance(0);
    load_b(dyi - 1, ks, raw);
    load_a(dyi, ks, areg);
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
      compute(s & 1);
      if (s + 1 < stages) {                               // stage s + 1 into the other buffer (last read during s - 1)
        stash_b(raw, (s + 1) & 1);
        stash_a(areg, (s + 1) & 1);
      }
      __syncthreads();
      advance(s + 1);
      load_b(dyi - 1, ks, raw);
      load_a(dyi, ks, areg);
    }
  } else {
    for (int s = 0; s < stages; ++s) {
      stash_b(raw, 0);
      stash_a(areg, 0);
      __syncthreads();
      advance(s);
      load_b(dyi - 1, ks, raw);
      load_a(dyi, ks, areg);
      compute(0);
      __syncthreads();
    }
  }
  if (zs != 0) {                                          // the residual belongs to slice 0 only
    pir_gemm_nn_t gz = g;
    gz.R = nullptr;
    pir_nn_epilogue<TM, TN>(acc, gz, Y, o1, 0, m0, n0, wm, wn, lane);
    return;
  }
  pir_nn_epilogue<TM, TN>(acc, g, Y, o1, 0, m0, n0, wm, wn, lane);
}

int g_rows_mode = 1;   // knob 30: 1 whole-row conv kernel where it serves the shape, 0 the nine-pass kernel

template <int TM, int TN, int WM, int WN>
int rows_launch(const RowsArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  if (BN % a.W != 0 || a.g.N % BN != 0) return 1000;
  dim3 grid((unsigned)(pir_cdiv(a.g.M, BM) * (a.g.N / BN)), (unsigned)a.g.O1, (unsigned)a.splits), block(WM * WN * 64);
  hipLaunchKernelGGL((conv3x3_rows_kernel<TM, TN, WM, WN>), grid, block, 0, s, a);
  return pir_launch_status();
}

}  // namespace

int pir_conv_rows_tune(int knob, int value) {
  if (knob != 30) return PIR_EINVAL;
  g_rows_mode = value;
  return PIR_OK;
}

// tile = the plan code of pir_conv3x3_x3's own choice (TM TN WM WN digits); 1000: shape not served (nothing launched)
// does the whole-row kernel serve the shape (the split over the stages exists in that kernel only)?
bool pir_conv_rows_serves(const pir_gemm_nn_t* g, int W, int tile) {
  if (!g_rows_mode) return false;
  if (W < 16 || W > 256 || (W & (W - 1)) != 0 || g->O2 != 1) return false;
  if ((reinterpret_cast<uintptr_t>(g->X) & 15) || g->x_s1 % 4 || g->ldx % 4) return false;
  const int bn = (tile / 100 % 10) * (tile % 10) * 32;
  return bn % W == 0 && g->N % bn == 0;
}

// splits > 1: g->Y is the slice buffer ([splits][B][M][N], y_s1 = M N); part_stride = B M N
int pir_conv_rows_launch(const pir_gemm_nn_t* g, int H, int W, int tile, hipStream_t s, int splits, long part_stride) {
  if (!g_rows_mode) return 1000;
  if (W < 16 || W > 256 || (W & (W - 1)) != 0 || g->O2 != 1) return 1000;
  if ((reinterpret_cast<uintptr_t>(g->X) & 15) || g->x_s1 % 4 || g->ldx % 4) return 1000;
  RowsArgs a;
  a.g = *g; a.H = H; a.W = W; a.ksteps = g->a3_kp / 16;
  a.splits = splits; a.part_stride = part_stride;
  a.wshift = 0;
  while ((1 << a.wshift) < W) ++a.wshift;
  switch (tile) {
    case 1214: return rows_launch<1, 2, 1, 4>(a, s);
    case 2214: return rows_launch<2, 2, 1, 4>(a, s);
    case 3214: return rows_launch<3, 2, 1, 4>(a, s);
    case 3114: return rows_launch<3, 1, 1, 4>(a, s);
    case 1222: return rows_launch<1, 2, 2, 2>(a, s);
    case 2222: return rows_launch<2, 2, 2, 2>(a, s);
    default: return 1000;
  }
}
