// gemm_nn on the bf16 matrix cores with fp32-class accuracy ("bf16x3 split"), gfx950.
//
// The 1e-4 parity bar forbids bf16 *inputs*, and gfx950 has no TF32: plain fp32 MFMA runs at 1/16 of the
// bf16 MFMA rate.  Every fp32 operand x is split exactly into three bf16 pieces
//     hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid),   |x - (hi+mid+lo)| <= 2^-27 |x|
// (each subtraction is exact in fp32), and a product block is accumulated in fp32 from the six bf16 MFMAs
//     lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi            (dropped terms <= 2^-27 relative)
// = 6 x v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16-deep k-step instead of 8 x v_mfma_f32_32x32x2_f32
// (64 cycles each): 2.67x the fp32 MFMA throughput at the same (fp32-accumulate) accuracy.
// Same interface, tiling, XCD-aware tile order and epilogue as gemm_nn_kernel (gemm.hip).
//
// Operand lane maps of v_mfma_f32_32x32x16_bf16 (cdna_hip_programming.md §3): lane l (r = l&31, h = l>>5)
// holds A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7  => one 16-byte LDS read per fragment from a
// [k-group][row][8 x bf16] image.  The split happens when a stage is written to LDS.
#include "gemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int XK = 16;  // k-depth of one LDS stage = one bf16 MFMA k-step

struct Frag3 { bf16x8 hi, mid, lo; };

__device__ __forceinline__ Frag3 split8(const float (&v)[8], bool ok) {
  Frag3 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = ok ? v[j] : 0.f;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    f.hi[j] = h; f.mid[j] = m; f.lo[j] = (__bf16)r2;
  }
  return f;
}

// dense 3x3 convolution as 9 x ceil(K/16) stages: stage it = (tap, k-step); the tap shifts the activation columns
// (ksplit, part_stride: plain products only - gridDim.z workgroups share an output tile, workgroup z multiplies its share of
// the k-steps and writes partial sums to Y + z * part_stride, the residual / row scale ride in slice 0; see pir_gemm_nn_ws)
struct X3Conv { int H, W, ksteps; unsigned magic_ks, magic_w; int ksplit; long part_stride; };

// A_PRE: A comes pre-split (pir_split_bf16x3): a fragment is three 16-byte loads, no conversion work.
// CONV (needs A_PRE, weights from pir_split_bf16x3_taps): the k loop also runs over the nine taps.
// waves_per_eu: the 128-column tiles (<= 50 KB of LDS) fit three workgroups per CU once the compiler is told to
// stay within 168 registers (it then also keeps the accumulators in VGPRs); the 256-column tiles run two.
template <int TM, int TN, int WM, int WN, bool A_MFAST, bool A_PRE, bool CONV = false, bool BREG = false>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu((WM == 2 && WN == 2 && TM == 1 && A_PRE) ? 4 : ((WM == 2 && WN == 2 && (A_PRE || TM * TN <= 2)) || (TM == 3 && TN == 1 && A_PRE)) ? 3 : 2)))
void gemm_nn_x3_kernel(pir_gemm_nn_t g, X3Conv cv) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, T = WM * WN * 64;
  constexpr int AKS = BM + 4;              // 16-byte units between the two k-groups of A (+4: bank shift)
  constexpr int AU = 2 * AKS, BU = 2 * BN; // units per part
  constexpr int PART = AU + BU, STAGE = 3 * PART;
  __shared__ bf16x8 smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int wg = pir_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg % tiles_m) * BM, n0 = (wg / tiles_m) * BN;
  const int o1 = blockIdx.y / g.O2, o2 = blockIdx.y % g.O2;
  const float* __restrict__ A = g.A + o1 * g.a_s1 + o2 * g.a_s2;
  const float* __restrict__ X = g.X + o1 * g.x_s1 + o2 * g.x_s2;
  float* __restrict__ Y = g.Y + o1 * g.y_s1 + o2 * g.y_s2 + (CONV ? 0L : (long)blockIdx.z * cv.part_stride);

  constexpr int AF = 2 * BM, NA = (AF + T - 1) / T;   // 8-deep k fragments per stage
  constexpr int BF = 2 * BN, NB = (BF + T - 1) / T;
  struct Stage { float a[A_PRE ? 1 : NA][8]; bf16x8 a3[A_PRE ? NA : 1][3]; float b[NB][8]; };
  const int iters_all = CONV ? 9 * cv.ksteps : (g.K + XK - 1) / XK;
  // split over k: this workgroup's k-steps are it_base .. it_base + iters - 1 (the host gives every slice at least one)
  const int ks_per = (!CONV && cv.ksplit > 1) ? (iters_all + cv.ksplit - 1) / cv.ksplit : iters_all;
  const int it_base = (!CONV && cv.ksplit > 1) ? (int)blockIdx.z * ks_per : 0;
  const int iters = iters_all - it_base < ks_per ? iters_all - it_base : ks_per;

  // fragment -> (row, k-group).  k-fast A (forward weights): the two k-groups of a row sit on adjacent lanes
  auto a_map = [&](int f, int& mm, int& kg) { if (A_MFAST) { mm = f % BM; kg = f / BM; } else { kg = f & 1; mm = f >> 1; } };

  // Buffer descriptors (wave-uniform): B rows beyond K and everything past the last valid element read as 0
  // through the hardware range check, so the k tail needs no masks (the check covers the scalar row offset too on
  // gfx950: tests/test_kernels_gpu.py::test_k_tail_never_multiplies_what_lies_behind_the_operand puts NaNs behind the
  // operand); per-lane offsets are computed once and
  // each load adds a scalar row offset (no vector address arithmetic inside the k loop).
  const __amdgpu_buffer_rsrc_t xrs = pir_make_rsrc(X, (unsigned)((((long)g.K - 1) * g.ldx + g.N) * 4));
  const __amdgpu_buffer_rsrc_t ars = pir_make_rsrc(g.A3, A_PRE ? (unsigned)((CONV ? 54L : 6L) * g.M * g.a3_kp) : 0u);
  static_assert(BF % T == 0, "B fragments must tile the workgroup");
  int b_voff[NB], b_kg[NB], a_voff[NA], b_taps[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int f = tid + i * T;
    // BREG (one wave owns all rows of its 32 columns): every lane loads its own MFMA B fragment - column lane & 31 of
    // the wave's block, k-group lane >> 5 (folded into the lane offset) - which then never passes through LDS
    const int n = BREG ? n0 + wn * 32 + (lane & 31) : n0 + f % BN;
    b_voff[i] = BREG ? n * 4 + (lane >> 5) * 8 * (int)g.ldx * 4 : n * 4;
    b_kg[i] = BREG ? 0 : __builtin_amdgcn_readfirstlane(f / BN);   // BN is a multiple of 64: uniform per wave
    b_taps[i] = 0;
    if (CONV) {  // bit t set: tap t of this pixel column lies inside the image
      const int hh = pir_fastdiv(n, cv.magic_w), ww = n - hh * cv.W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hs = hh + t / 3 - 1, ws = ww + t % 3 - 1;
        if (n < g.N && hs >= 0 && hs < cv.H && ws >= 0 && ws < cv.W) b_taps[i] |= 1 << t;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {   // pre-split A: lanes (2r, 2r+1) = the two k-groups of row r -> 1 KB contiguous per wave
    const int f = tid + i * T;
    int mm = f >> 1, kg = f & 1;
    if (mm >= BM) { mm = 0; kg = 0; }
    const int m = m0 + mm, mc = m < g.M ? m : g.M - 1;
    a_voff[i] = (mc * 16 + 8 * kg) * 2;
  }
  const int a3_part_bytes = (CONV ? 9 : 1) * g.M * g.a3_kp * 2, a3_step_bytes = g.M * 32, ldx4 = (int)g.ldx * 4;

  auto load = [&](int it_raw, Stage& st) {
    const int it = it_base + (it_raw < iters ? it_raw : iters - 1);
    int tap = 0, ks = it;
    if (CONV) { tap = pir_fastdiv(it, cv.magic_ks); ks = it - tap * cv.ksteps; }
    const int k0 = ks * XK, klast = g.K - 1 - k0;
    const int tap_shift = CONV ? ((tap / 3 - 1) * cv.W + tap % 3 - 1) * 4 : 0;
    const float* __restrict__ At = A + (long)k0 * g.a_sk;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (A_PRE) {  // [part][m][kp]: k is zero-padded to a multiple of 16, so no k clamp is needed
#pragma unroll
        for (int part = 0; part < 3; ++part)
          st.a3[i][part] = __builtin_bit_cast(
              bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, a_voff[i], part * a3_part_bytes + it * a3_step_bytes, 0));
      } else {
        int mm, kg;
        a_map(tid + i * T, mm, kg);
        if (mm >= BM || kg > 1) { mm = 0; kg = 0; }
        const int m = m0 + mm, mc = m < g.M ? m : g.M - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kk = 8 * kg + j, kc = kk <= klast ? kk : klast;
          st.a[i][j] = At[mc * (int)g.a_sm + kc * (int)g.a_sk];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      // a column whose tap falls outside the image gets an out-of-range offset: the loads return 0
      const int voff = CONV ? (((b_taps[i] >> tap) & 1) ? b_voff[i] + tap_shift : 0x7ffffff0) : b_voff[i];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        st.b[i][j] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(xrs, voff, (k0 + 8 * b_kg[i] + j) * ldx4, 0));
    }
  };

  auto stash_a = [&](int buf, int it, const Stage& st) {
    bf16x8* base = smem + buf * STAGE;
    const int klast = g.K - 1 - it * XK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * T;
      int mm, kg;
      if (A_PRE) { mm = f >> 1; kg = f & 1; } else a_map(f, mm, kg);
      if (AF % T == 0 || f < AF) {
        const int u = kg * AKS + mm;
        if (A_PRE) {  // rows beyond M only feed masked outputs: no zeroing needed
          base[u] = st.a3[i][0]; base[PART + u] = st.a3[i][1]; base[2 * PART + u] = st.a3[i][2];
        } else {
          // a fragment is all-or-nothing in m; k beyond K is zeroed element-wise
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (8 * kg + j <= klast) ? st.a[i][j] : 0.f;
          const Frag3 fr = split8(v, m0 + mm < g.M);
          base[u] = fr.hi; base[PART + u] = fr.mid; base[2 * PART + u] = fr.lo;
        }
      }
    }
  };
  static_assert(!BREG || (WM == 1 && TN == 1 && NB == 1 && !CONV), "B in registers: one wave per column block");
  Frag3 bfr[2];   // BREG: the split B fragment of the stage in each LDS buffer's turn
  auto stash_b = [&](int buf, const Stage& st) {
    if constexpr (BREG) { bfr[buf] = split8(st.b[0], true); return; }
    bf16x8* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * T;
      const int nn = f % BN, kg = f / BN;
      const Frag3 fr = split8(st.b[i], true);   // k tail already zero (range-checked loads)
      const int u = AU + kg * BN + nn;
      base[u] = fr.hi; base[PART + u] = fr.mid; base[2 * PART + u] = fr.lo;
    }
  };
  auto stash = [&](int buf, int it, const Stage& st) {
    stash_a(buf, it, st);
    stash_b(buf, st);
  };
  // compute(cur) and the split of the next stage's activations share one basic block; ask the scheduler for
  // LDS reads first, then a few conversion VALU ops in the shadow of every MFMA, LDS writes last (left alone it
  // clusters the VALU work behind the MFMAs, where nothing hides it)
  auto interleave = [&]() {
    __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);     // fragment reads first
#pragma unroll
    for (int q = 0; q < TM * TN * 6; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA ...
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               // ... one global load in its shadow
    }
    __builtin_amdgcn_sched_group_barrier(0x200, 3 * NB, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    const bf16x8* base = smem + buf * STAGE;
    const int h = lane >> 5, r = lane & 31;
    const bf16x8* ap = base + h * AKS + wm * TM * 32 + r;
    const bf16x8* bp = base + AU + h * BN + wn * TN * 32 + r;
    bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { ah[i] = ap[i * 32]; am[i] = ap[PART + i * 32]; al[i] = ap[2 * PART + i * 32]; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (BREG) { bh[j] = bfr[buf].hi; bm[j] = bfr[buf].mid; bl[j] = bfr[buf].lo; }
      else { bh[j] = bp[j * 32]; bm[j] = bp[PART + j * 32]; bl[j] = bp[2 * PART + j * 32]; }
    }
    // term-major order: consecutive MFMAs go to DIFFERENT accumulators (the per-accumulator order of the six
    // terms, hence the result, is unchanged).  Left accumulator-major, the compiler emits six back-to-back
    // dependent MFMAs per accumulator.
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

  // Two stages are loaded ahead into registers.  (Measured: making these loads unconditional so that the
  // compiler can keep exact vmcnt counts is slower - the two extra stage loads per tile cost more than the
  // deeper prefetch gains, K is often only 3-6 stages.)
  Stage s0, s1;
  load(0, s0);
  if (iters > 1) load(1, s1);
  stash(0, 0, s0);
  __syncthreads();
  int it = 0;
  // Main part: both prefetch loads are in range, so they sit in the same basic block as the MFMAs and the
  // scheduler can be told to issue one vector-memory instruction behind each MFMA.  In-kernel cycle counters
  // showed the load issue of a stage (19 instructions, ~40 cycles each in the wave's in-order stream) costing
  // ~820 cycles per k-step in front of 1152 cycles of MFMAs.
  for (; it + 3 < iters; it += 2) {
    load(it + 2, s0);
    compute(0);
    stash_b(1, s1);
    interleave();
    stash_a(1, it + 1, s1);
    __syncthreads();
    load(it + 3, s1);
    compute(1);
    stash_b(0, s0);
    interleave();
    stash_a(0, it + 2, s0);
    __syncthreads();
  }
  for (; it + 1 < iters; it += 2) {   // tail: at most three stages left, loads guarded
    if (it + 2 < iters) load(it + 2, s0);
    compute(0);
    stash_b(1, s1);
    stash_a(1, it + 1, s1);
    __syncthreads();
    if (it + 3 < iters) load(it + 3, s1);
    compute(1);
    stash_b(0, s0);   // harmless past the end (slot 0 is not read again)
    if (it + 2 < iters) stash_a(0, it + 2, s0);
    __syncthreads();
  }
  if (it < iters) compute(0);

  if (!CONV && blockIdx.z != 0) {      // residual and row scale belong to slice 0
    pir_gemm_nn_t gz = g;
    gz.R = nullptr; gz.rowscale = nullptr;
    pir_nn_epilogue<TM, TN>(acc, gz, Y, o1, o2, m0, n0, wm, wn, lane);
    return;
  }
  pir_nn_epilogue<TM, TN>(acc, g, Y, o1, o2, m0, n0, wm, wn, lane);
}

int g_x3_fill32 = 45;  // knob 43: 32 x 128 tiles when the 96 x 128 plan would give fewer workgroups than this percentage of the CUs (0: never);
                       // tools/deep_ab.py at part batches of 1 / 4: 0.66 - 0.86 of the 96 x 128 tile's time up to 69 % fill, 1.16 at 75 %
int g_x3_breg = -1;   // knob 18: activations stay in registers in the one-wave-per-column-block tile (96 x 128): -1 automatic
                      // (planes of <= 4096 pixels: -2..-9 %, bit-identical; neutral to worse at 128^2), 0 never, 1 always

int g_x3_conv_fill = 1;   // pir_tune_set knob 29: narrower dense-3x3 tiles when 96 x 256 leaves the chip underfilled

template <int TM, int TN, int WM, int WN>
int launch_cfg(const pir_gemm_nn_t& g, hipStream_t s, const X3Conv* conv = nullptr, int ksplit = 1, long part_stride = 0) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const long tiles = pir_cdiv(g.M, BM) * pir_cdiv(g.N, BN);
  dim3 grid((unsigned)tiles, (unsigned)(g.O1 * g.O2), (unsigned)(conv ? 1 : ksplit)), block(WM * WN * 64);
  X3Conv cv = {0, 0, 0, 0u, 0u, ksplit, part_stride};
  if (conv) hipLaunchKernelGGL((gemm_nn_x3_kernel<TM, TN, WM, WN, true, true, true>), grid, block, 0, s, g, *conv);
  else if (g.A3 && (g_x3_breg < 0 ? g.N <= 4096 : g_x3_breg != 0) && WM == 1 && TN == 1) {
    if constexpr (WM == 1 && TN == 1) hipLaunchKernelGGL((gemm_nn_x3_kernel<TM, TN, WM, WN, true, true, false, true>), grid, block, 0, s, g, cv);
  }
  else if (g.A3) hipLaunchKernelGGL((gemm_nn_x3_kernel<TM, TN, WM, WN, true, true>), grid, block, 0, s, g, cv);
  else if (g.a_sm == 1) hipLaunchKernelGGL((gemm_nn_x3_kernel<TM, TN, WM, WN, true, false>), grid, block, 0, s, g, cv);
  else hipLaunchKernelGGL((gemm_nn_x3_kernel<TM, TN, WM, WN, false, false>), grid, block, 0, s, g, cv);
  return pir_launch_status();
}

// out[part][k/16][m][k%16] bf16 pieces of W(m,k), zero-padded in k to kp: the 16 k-values of one MFMA
// k-step are contiguous per row and the rows of a tile are contiguous per k-step, so a tile's stage is ONE
// contiguous block (BM x 32 bytes per part) that a wave reads with fully coalesced 16-byte loads.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ W, int M, int K, long sm, long sk,
                                                           __bf16* __restrict__ out, int kp) {
  const long total = (long)M * kp;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int m = (int)(e / kp), k = (int)(e % kp);
    const float x = k < K ? W[m * sm + k * sk] : 0.f;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    const long d = ((long)(k >> 4) * M + m) * 16 + (k & 15);
    out[d] = h; out[total + d] = mid; out[2 * total + d] = (__bf16)r2;
  }
}

// nine taps: out[part][tap][k/16][m][k%16] of W(tap, m, k) = W[(flip ? 8 - tap : tap) * st + m * sm + k * sk]
__global__ __launch_bounds__(256) void split_bf16x3_taps_kernel(const float* __restrict__ W, int M, int K, long st, long sm,
                                                                long sk, int flip, __bf16* __restrict__ out, int kp) {
  const long per_tap = (long)M * kp, total = 9 * per_tap;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(e / per_tap);
    const long r = e - tap * per_tap;
    const int m = (int)(r / kp), k = (int)(r % kp);
    const float x = k < K ? W[(flip ? 8 - tap : tap) * st + m * sm + k * sk] : 0.f;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    const long d = tap * per_tap + ((long)(k >> 4) * M + m) * 16 + (k & 15);
    out[d] = h; out[total + d] = mid; out[2 * total + d] = (__bf16)r2;
  }
}

// All weights of a model in one launch: block b handles OUTPUT elements [begin, begin + 4096) of descriptor blk[b].x
// (blk[b].y = begin / 4096); layout per descriptor as written by the two kernels above.  Walking the output order
// (k % 16 fastest, then the row, then the k-step, then the tap) makes the 2-byte stores contiguous and keeps a block's
// reads inside a 16 x 256 (k x row) patch of the weight, so every fetched line is used from L1 whatever the
// orientation (walking the input order cost 1.7 GB of fetches per step for 142 MB of weights in the transposed,
// input-gradient orientation).
__global__ __launch_bounds__(256) void split_bf16x3_batch_kernel(const pir_split_desc_t* __restrict__ descs,
                                                                 const int2* __restrict__ blk) {
  const int2 bd = blk[blockIdx.x];
  const pir_split_desc_t d = descs[bd.x];
  const int kp = (d.K + 15) / 16 * 16, ntap = d.taps ? 9 : 1;
  const long per_tap = (long)d.M * kp, total = ntap * per_tap;
  const long per_step = (long)d.M * 16;
  __bf16* __restrict__ out = reinterpret_cast<__bf16*>(d.out);
  const long base = (long)bd.y * 4096;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const long o = base + i * 256 + threadIdx.x;
    if (o >= total) break;
    const int tap = (int)(o / per_tap);
    const long r = o - tap * per_tap;
    const int ks = (int)(r / per_step);
    const int r2 = (int)(r - ks * per_step);
    const int m = r2 >> 4, k = ks * 16 + (r2 & 15);
    const float x = k < d.K ? d.W[(d.flip ? 8 - tap : tap) * d.st + m * d.sm + k * d.sk] : 0.f;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mid = (__bf16)r1;
    const float rr = r1 - (float)mid;
    out[o] = h; out[total + o] = mid; out[2 * total + o] = (__bf16)rr;
  }
}

}  // namespace

extern "C" int pir_split_bf16x3_batch(const pir_split_desc_t* descs, const int* blocks, int nblocks, pir_stream_t stream) {
  PIR_CHECK_ARG(descs && blocks && nblocks > 0);
  hipLaunchKernelGGL(split_bf16x3_batch_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, descs,
                     reinterpret_cast<const int2*>(blocks));
  return pir_launch_status();
}

extern "C" int pir_split_bf16x3_taps(const float* W, int M, int K, long st, long sm, long sk, int flip, void* out,
                                     pir_stream_t stream) {
  PIR_CHECK_ARG(W && out && M > 0 && K > 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const int kp = (int)(pir_cdiv(K, 16) * 16);
  const long total = 9L * M * kp;
  const int blocks = (int)(pir_cdiv(total, 256) < 2048 ? pir_cdiv(total, 256) : 2048);
  hipLaunchKernelGGL(split_bf16x3_taps_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, M, K, st, sm, sk, flip,
                     reinterpret_cast<__bf16*>(out), kp);
  return pir_launch_status();
}

int g_x3_ksplit = 1;   // knob 45: split of underfilled deep-k products over k (pir_gemm_nn_ws; 0: never)
int g_x3_conv_split = 1;   // knob 44: split of the underfilled dense 3x3 convolutions over their stages (0: never)

// tile of the dense convolution (rows = output channels, often few)
static int conv_tile(int M, int N, int B) {
  const long pad96 = pir_cdiv(M, 96) * 96, pad128 = pir_cdiv(M, 128) * 128;
  const long blocks128 = pir_cdiv(M, 128) * pir_cdiv(N, 128) * B;
  if (M <= 32) return 1214;
  if (M <= 64) return 2214;
  if (pad96 <= pad128 && N >= 256) {
    // the low-resolution convolutions (16^2, 32^2 planes, a part batch of images) give 96 x 256 tiles only 64-128 workgroups
    // with k loops of 200-430 steps: narrower tiles until the chip is about filled (tools/conv3x3_bench.py)
    const long wg256 = pir_cdiv(M, 96) * pir_cdiv(N, 256) * B, wg128 = pir_cdiv(M, 96) * pir_cdiv(N, 128) * B;
    if (g_x3_conv_fill && wg256 < PIR_NUM_CU) return wg128 >= 3L * PIR_NUM_CU / 4 ? 3114 : 1222;
    return 3214;
  }
  if (blocks128 < 2L * PIR_NUM_CU && pir_cdiv(M, 64) * 64 <= pad128) return 1222;
  return 2222;
}

// Slices the stage loop (3 row shifts x a3_kp / 16 k-steps) of an underfilled launch is cut into: as many as fill two
// workgroups per CU, at least 12 stages each; 1 = no split.
static int conv_splits(int tile, int M, int N, int B, int a3_kp) {
  if (!g_x3_conv_split) return 1;
  const int bm = (tile / 1000) * (tile / 10 % 10) * 32, bn = (tile / 100 % 10) * (tile % 10) * 32;
  const long wgs = pir_cdiv(M, bm) * pir_cdiv(N, bn) * B;
  const int stages = 3 * (a3_kp / 16);
  long s = 2L * PIR_NUM_CU / wgs;
  if (s > stages / 12) s = stages / 12;
  if (s > 8) s = 8;
  // (two slices of an about-filled launch measured worse: up3_2 at 16 images 106 -> 137 us)
  return s < 3 ? 1 : (int)s;
}

// workspace floats pir_conv3x3_x3_ws wants for this call (0: it would not split; any buffer, or none, will do)
extern "C" size_t pir_conv3x3_x3_ws_floats(int B, int M, int K, int H, int W) {
  if (B <= 0 || M <= 0 || K <= 0 || H <= 0 || W <= 0) return 0;
  const int tile = conv_tile(M, H * W, B);
  const int sp = conv_splits(tile, M, H * W, B, (int)(pir_cdiv(K, 16) * 16));
  return sp > 1 ? (size_t)sp * B * M * H * W : 0;
}

int pir_reduce_partials_now(const float* parts, long stride, int S, float alpha, int accumulate, float* out, long count,
                            pir_stream_t stream);   // reduce_batch.hip: launched at once, never queued

static int conv3x3_x3_impl(const void* A3, int a3_kp, const float* X, long x_bs, float* Y, long y_bs, const float* R,
                           long r_bs, int B, int M, int K, int H, int W, float* ws, size_t ws_floats, pir_stream_t stream) {
  PIR_CHECK_ARG(A3 && X && Y && B > 0 && M > 0 && K > 0 && H > 0 && W > 0 && B <= 65535);
  PIR_CHECK_ARG(a3_kp == (int)(pir_cdiv(K, 16) * 16));
  PIR_CHECK_ARG((long)(M > K ? M : K) * H * W < (1L << 28) && 54L * M * a3_kp < (1L << 31));
  // pixel -> (row, column) uses pir_fastdiv(n, magic_w), exact only while n * W < 2^32 (n < H*W + one tile of slack)
  PIR_CHECK_ARG(((long)H * W + 256) * W < (1L << 32));
  pir_gemm_nn_t g;
  g.A = nullptr; g.a_s1 = g.a_s2 = 0; g.a_sm = 0; g.a_sk = 0;
  g.X = X; g.x_s1 = x_bs; g.x_s2 = 0; g.ldx = (long)H * W;
  g.Y = Y; g.y_s1 = y_bs; g.y_s2 = 0; g.ldy = (long)H * W;
  g.R = R; g.r_s1 = r_bs; g.r_s2 = 0; g.ldr = (long)H * W;
  g.rowscale = nullptr; g.rs_s1 = g.rs_s2 = 0;
  g.M = M; g.K = K; g.N = H * W; g.O1 = B; g.O2 = 1;
  g.A3 = A3; g.a3_kp = a3_kp;
  X3Conv cv;
  cv.ksplit = 1; cv.part_stride = 0;
  cv.H = H; cv.W = W; cv.ksteps = a3_kp / 16;
  cv.magic_ks = pir_magic((unsigned)cv.ksteps); cv.magic_w = pir_magic((unsigned)W);
  hipStream_t s = (hipStream_t)stream;
  const int tile = conv_tile(M, g.N, B);
  // Underfilled launch behind a long stage loop (up4_3: 48 - 96 workgroups walking 72 - 432 stages, 92 - 175 us at any batch
  // size): the stages are cut into slices that run side by side, each writes its partial sums (the residual rides in slice
  // 0) and the deterministic second stage adds them in order.  Needs a contiguous output and the whole-row kernel.
  const int sp = conv_splits(tile, M, g.N, B, a3_kp);
  const long out_floats = (long)B * M * g.N;
  if (sp > 1 && ws && (size_t)sp * out_floats <= ws_floats && y_bs == (long)M * g.N && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 &&
      pir_conv_rows_serves(&g, W, tile)) {
    pir_gemm_nn_t gs = g;
    gs.Y = ws; gs.y_s1 = (long)M * g.N;
    int st = pir_conv_rows_launch(&gs, H, W, tile, s, sp, out_floats);
    if (st) return st;
    return pir_reduce_partials_now(ws, out_floats, sp, 1.f, 0, Y, out_floats, stream);
  }
  {   // whole image rows per tile, activations loaded once per row shift (conv_rows.hip) where the shape allows
    const int st = pir_conv_rows_launch(&g, H, W, tile, s);
    if (st != 1000) return st;
  }
  switch (tile) {
    case 1214: return launch_cfg<1, 2, 1, 4>(g, s, &cv);
    case 2214: return launch_cfg<2, 2, 1, 4>(g, s, &cv);
    case 3214: return launch_cfg<3, 2, 1, 4>(g, s, &cv);
    case 3114: return launch_cfg<3, 1, 1, 4>(g, s, &cv);
    case 1222: return launch_cfg<1, 2, 2, 2>(g, s, &cv);
    default: return launch_cfg<2, 2, 2, 2>(g, s, &cv);
  }
}

extern "C" int pir_conv3x3_x3(const void* A3, int a3_kp, const float* X, long x_bs, float* Y, long y_bs, const float* R,
                              long r_bs, int B, int M, int K, int H, int W, pir_stream_t stream) {
  return conv3x3_x3_impl(A3, a3_kp, X, x_bs, Y, y_bs, R, r_bs, B, M, K, H, W, nullptr, 0, stream);
}

// The same convolution with a scratch buffer (pir_conv3x3_x3_ws_floats): launches that would leave most of the chip idle
// behind a long stage loop are split over their stages (results agree with the unsplit launch to fp32 rounding: another
// grouping of the same sum; deterministic).
extern "C" int pir_conv3x3_x3_ws(const void* A3, int a3_kp, const float* X, long x_bs, float* Y, long y_bs, const float* R,
                                 long r_bs, int B, int M, int K, int H, int W, float* ws, size_t ws_floats, pir_stream_t stream) {
  return conv3x3_x3_impl(A3, a3_kp, X, x_bs, Y, y_bs, R, r_bs, B, M, K, H, W, ws, ws_floats, stream);
}

extern "C" size_t pir_split_bf16x3_bytes(int M, int K) {
  if (M <= 0 || K <= 0) return 0;
  return (size_t)3 * M * (pir_cdiv(K, 16) * 16) * 2;
}

extern "C" int pir_split_bf16x3(const float* W, int M, int K, long sm, long sk, void* out, pir_stream_t stream) {
  PIR_CHECK_ARG(W && out && M > 0 && K > 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const int kp = (int)(pir_cdiv(K, 16) * 16);
  const long total = (long)M * kp;
  const int blocks = (int)(pir_cdiv(total, 256) < 2048 ? pir_cdiv(total, 256) : 2048);
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, M, K, sm, sk,
                     reinterpret_cast<__bf16*>(out), kp);
  return pir_launch_status();
}

int g_x3_narrow96 = 4;   // pir_tune_set knob 5: 96 x 128 instead of 96 x 256 below this many workgroups per CU

int pir_nn_x3_tune(int knob, int value) {
  if (knob == 5) { g_x3_narrow96 = value; return PIR_OK; }
  if (knob == 18) { g_x3_breg = value; return PIR_OK; }
  if (knob == 43) { g_x3_fill32 = value; return PIR_OK; }
  if (knob == 29) { g_x3_conv_fill = value; return PIR_OK; }
  if (knob == 44) { g_x3_conv_split = value; return PIR_OK; }
  if (knob == 45) { g_x3_ksplit = value; return PIR_OK; }
  return PIR_EINVAL;
}

// knob: -1 automatic, 0 never, 1 always
bool pir_nn_x3_wanted(const pir_gemm_nn_t* a, int knob) {
  // measured (bench.py, batch 32): the bf16x3 path is at least as fast as fp32 MFMA for every gemm_nn of the
  // train step, with or without pre-split weights, so it is the default; knob 0 forces the fp32 kernel.
  // The kernel addresses one image of X through a buffer descriptor of ((K-1)*ldx + N)*4 bytes and signed 32-bit
  // byte offsets (k_padded * ldx * 4): larger operands take the fp32 kernel (element offsets, checked < 2^31).
  const long kp = pir_cdiv(a->K, 16) * 16;
  if ((kp * a->ldx + a->N) * 4 >= (1L << 31)) return false;
  return knob != 0;
}

// Tile plan of the bf16x3 kernel for one call: TM*1000 + TN*100 + WM*10 + WN (a workgroup of WM x WN waves, each
// wave TM x TN MFMA tiles of 32 x 32).  Host-only, also exported (pir_gemm_nn_plan) so tests can pin which
// instantiation a shape selects.
int pir_nn_x3_plan(const pir_gemm_nn_t* a, int cfg) {
  const pir_gemm_nn_t& g = *a;
  const int M = g.M;
  const long batch = (long)g.O1 * g.O2;
  if (cfg >= 0) {
    switch (cfg) {
      case 0: return 1214;
      case 1: return 2214;
      case 2: return 3214;
      case 3: return 2222;
      case 7: return 3114;   // 96 x 128
      case 8: return 1114;   // 32 x 128
      default: return 1222;
    }
  }
  // tile choice from the sweep in tools/ktune.py (PIR_X3=1): the 96 x 256 tile (1 KB contiguous per
  // row and stage) wins at the high-resolution levels even with up to ~13 % more padded rows.
  if (M <= 32) return 1214;
  if (M <= 64) return g.K <= 192 ? 1222 : 2214;   // 64 x 128 up to k = 192 (sweep of round 2: K = 127, 144 at M = 48)
  const long pad96 = pir_cdiv(M, 96) * 96, pad128 = pir_cdiv(M, 128) * 128;
  // Low-resolution levels (32^2, 16^2: long k loops, few columns): the three-workgroups-per-CU tiles win by 6-12 %
  // over 96 x 256 / 64 x 128 (tools/cfg_ab.py over every level, profiles/r02_gemm_nn_tile_sweep.txt): 96 x 128 unless
  // 128-row tiles waste fewer rows (M = 510, 1020, 1021, 2042); also the 64^2 level's qkv (M = 288)
  // Underfilled launches (fewer 96 x 128 workgroups than g_x3_fill32 percent of the CUs: the 16^2 / 32^2 levels at part batches
  // of 1 - 4 images): 32-row tiles - three times the workgroups, a third of the MFMAs in every k-step's serial chain
  if (g.A3 && g_x3_fill32 > 0 && M >= 64 && g.N <= 4096 &&
      pir_cdiv(M, 96) * pir_cdiv(g.N, 128) * batch * 100 < (long)g_x3_fill32 * PIR_NUM_CU) return 1114;
  if (g.A3 && M >= 192 && g.N <= 1024) return pad128 < pad96 ? 2222 : 3114;
  if (g.A3 && M >= 192 && g.N <= 4096 && g.K <= 128 && pad96 < pad128) return 3114;
  // Short k loops on long pixel rows (forward / input gradient of the 1x1 convolutions at the 128^2 and 64^2 levels,
  // HBM-bound): when 128-row tiles waste no more rows than 96-row tiles (M = 254, 255, 510), the 128 x 128 tile
  // (three workgroups per CU) is 5-11 % faster than 96 x 128 / 96 x 256 (tools/cfg_ab.py, round 2)
  if (g.A3 && g.K <= 128 && g.N >= 4096 && M >= 192 && pad128 <= pad96) return 2222;
  // 96 x 128 (three workgroups per CU) for the project_in input gradient at the 96-channel full-resolution levels
  if (g.A3 && g.K >= 384 && M == 96 && g.N >= 16384) return 3114;
  const bool use96 = g.N >= 1024 ? pad96 * 100 <= pad128 * 113 : pad96 < pad128;
  const long blocks = use96 ? pir_cdiv(M, 96) * pir_cdiv(g.N, 256) * batch : pir_cdiv(M, 128) * pir_cdiv(g.N, 128) * batch;
  if (blocks < 2L * PIR_NUM_CU && pir_cdiv(M, 64) * 64 <= pad128) return 1222;
  // few 96 x 256 workgroups (one part-batch stream of the 64^2 level): 96 x 128 doubles them at three per CU
  if (use96 && blocks < (long)g_x3_narrow96 * PIR_NUM_CU) return 3114;
  if (use96) return 3214;
  return 2222;
}


// Slices the k loop of this product is cut into when a scratch buffer is at hand (1: no split): pre-split weights, one
// contiguous output per image, fewer workgroups than half of the CUs (the 16^2 level's 384-row products at 1 - 4 images per part:
// 8 - 32 workgroups walking 64 - 128 k-steps, 28 - 50 us each), at least 16 k-steps per slice.
int pir_nn_x3_ksplit(const pir_gemm_nn_t* a, int cfg) {
  const pir_gemm_nn_t& g = *a;
  if (!g_x3_ksplit || !g.A3 || g.O2 != 1 || g.y_s1 != (long)g.M * g.N || g.ldy != g.N) return 1;
  const int plan = pir_nn_x3_plan(a, cfg);
  const int bm = (plan / 1000) * (plan / 10 % 10) * 32, bn = (plan / 100 % 10) * (plan % 10) * 32;
  const long wgs = pir_cdiv(g.M, bm) * pir_cdiv(g.N, bn) * g.O1;
  const int iters = (int)pir_cdiv(g.K, 16);
  if (wgs * 2 >= PIR_NUM_CU) return 1;   // (at 128 workgroups - the 384-row products at 16 images - the split is neutral in the step)
  long s = 2L * PIR_NUM_CU / wgs;
  if (s > iters / 16) s = iters / 16;
  if (s > 8) s = 8;
  return s < 2 ? 1 : (int)s;
}

int pir_nn_x3_launch(const pir_gemm_nn_t* a, int cfg, hipStream_t s, int ksplit, long part_stride) {
  const pir_gemm_nn_t& g = *a;
  switch (pir_nn_x3_plan(a, cfg)) {
    case 1214: return launch_cfg<1, 2, 1, 4>(g, s, nullptr, ksplit, part_stride);
    case 2214: return launch_cfg<2, 2, 1, 4>(g, s, nullptr, ksplit, part_stride);
    case 3214: return launch_cfg<3, 2, 1, 4>(g, s, nullptr, ksplit, part_stride);
    case 2222: return launch_cfg<2, 2, 2, 2>(g, s, nullptr, ksplit, part_stride);
    case 3114: return launch_cfg<3, 1, 1, 4>(g, s, nullptr, ksplit, part_stride);
    case 1114: return launch_cfg<1, 1, 1, 4>(g, s, nullptr, ksplit, part_stride);
    default: return launch_cfg<1, 2, 2, 2>(g, s, nullptr, ksplit, part_stride);
  }
}
