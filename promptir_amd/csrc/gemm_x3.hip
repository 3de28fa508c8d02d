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
#ifndef PP_STAGE_PRIO
#define PP_STAGE_PRIO 0   // wave priority of the ping-pong kernel's staging phase (experiment knob)
#endif
#ifndef X3_MFMA_PRIO
#define X3_MFMA_PRIO 0   // experiment knob: > 0 raises the wave priority around the MFMA cluster, < 0 raises it everywhere else
#endif
#ifndef X3_VMEM_FILL
#define X3_VMEM_FILL 1   // issue the prefetch loads one per MFMA inside the main loop
#endif
#ifndef X3_FILL
#define X3_FILL 0   // N > 0: place N conversion VALU ops behind each MFMA of the main loop (measured: no gain)
#endif

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
struct X3Conv { int H, W, ksteps; unsigned magic_ks, magic_w; };

// A_PRE: A comes pre-split (pir_split_bf16x3): a fragment is three 16-byte loads, no conversion work.
// CONV (needs A_PRE, weights from pir_split_bf16x3_taps): the k loop also runs over the nine taps.
// waves_per_eu: the 128-column tiles (<= 50 KB of LDS) fit three workgroups per CU once the compiler is told to
// stay within 168 registers (it then also keeps the accumulators in VGPRs); the 256-column tiles run two.
template <int TM, int TN, int WM, int WN, bool A_MFAST, bool A_PRE, bool CONV = false>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(((WM == 2 && WN == 2 && (A_PRE || TM * TN <= 2)) || (TM == 3 && TN == 1 && A_PRE)) ? 3 : 2)))
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
  float* __restrict__ Y = g.Y + o1 * g.y_s1 + o2 * g.y_s2;

  constexpr int AF = 2 * BM, NA = (AF + T - 1) / T;   // 8-deep k fragments per stage
  constexpr int BF = 2 * BN, NB = (BF + T - 1) / T;
  struct Stage { float a[A_PRE ? 1 : NA][8]; bf16x8 a3[A_PRE ? NA : 1][3]; float b[NB][8]; };
  const int iters = CONV ? 9 * cv.ksteps : (g.K + XK - 1) / XK;

  // fragment -> (row, k-group).  k-fast A (forward weights): the two k-groups of a row sit on adjacent lanes
  auto a_map = [&](int f, int& mm, int& kg) { if (A_MFAST) { mm = f % BM; kg = f / BM; } else { kg = f & 1; mm = f >> 1; } };

  // Buffer descriptors (wave-uniform): B rows beyond K and everything past the last valid element read as 0
  // through the hardware range check, so the k tail needs no masks; per-lane offsets are computed once and
  // each load adds a scalar row offset (no vector address arithmetic inside the k loop).
  const __amdgpu_buffer_rsrc_t xrs = pir_make_rsrc(X, (unsigned)((((long)g.K - 1) * g.ldx + g.N) * 4));
  const __amdgpu_buffer_rsrc_t ars = pir_make_rsrc(g.A3, A_PRE ? (unsigned)((CONV ? 54L : 6L) * g.M * g.a3_kp) : 0u);
  static_assert(BF % T == 0, "B fragments must tile the workgroup");
  int b_voff[NB], b_kg[NB], a_voff[NA], b_taps[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int f = tid + i * T;
    const int n = n0 + f % BN;
    b_voff[i] = n * 4;
    b_kg[i] = __builtin_amdgcn_readfirstlane(f / BN);   // BN is a multiple of 64: uniform per wave
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
    const int it = it_raw < iters ? it_raw : iters - 1;
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
  auto stash_b = [&](int buf, const Stage& st) {
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
      if (X3_FILL > 0) __builtin_amdgcn_sched_group_barrier(0x002, X3_FILL, 0);
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
    for (int j = 0; j < TN; ++j) { bh[j] = bp[j * 32]; bm[j] = bp[PART + j * 32]; bl[j] = bp[2 * PART + j * 32]; }
    // term-major order: consecutive MFMAs go to DIFFERENT accumulators (the per-accumulator order of the six
    // terms, hence the result, is unchanged).  Left accumulator-major, the compiler emits six back-to-back
    // dependent MFMAs per accumulator.
#define PIR_X3_TERM(A_, B_)                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_[j], acc[i][j], 0, 0, 0);
    if (X3_MFMA_PRIO > 0) __builtin_amdgcn_s_setprio(X3_MFMA_PRIO);
    if (X3_MFMA_PRIO < 0) __builtin_amdgcn_s_setprio(0);
    PIR_X3_TERM(al, bh)
    PIR_X3_TERM(ah, bl)
    PIR_X3_TERM(am, bm)
    PIR_X3_TERM(am, bh)
    PIR_X3_TERM(ah, bm)
    PIR_X3_TERM(ah, bh)
    if (X3_MFMA_PRIO > 0) __builtin_amdgcn_s_setprio(0);
    if (X3_MFMA_PRIO < 0) __builtin_amdgcn_s_setprio(-X3_MFMA_PRIO);
#undef PIR_X3_TERM
  };

  // Two stages are loaded ahead into registers.  (Measured: making these loads unconditional so that the
  // compiler can keep exact vmcnt counts is slower - the two extra stage loads per tile cost more than the
  // deeper prefetch gains, K is often only 3-6 stages.)
  if (X3_MFMA_PRIO < 0) __builtin_amdgcn_s_setprio(-X3_MFMA_PRIO);
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
    if (X3_VMEM_FILL) interleave();
    stash_a(1, it + 1, s1);
    __syncthreads();
    load(it + 3, s1);
    compute(1);
    stash_b(0, s0);
    if (X3_VMEM_FILL) interleave();
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

  pir_nn_epilogue<TM, TN>(acc, g, Y, o1, o2, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------
// Ping-pong variant for pre-split weights (A3): one persistent 8-wave workgroup per CU, tile 128 x 256.
// The two 4-wave halves ("ping": rows 0-63, "pong": rows 64-127 of the tile; each wave 64 x 64) alternate
// between a matrix phase (24 MFMAs on the stage that is two steps old) and a staging phase (split the
// freshly arrived fp32 activations to bf16x3, write them to a 3-deep LDS ring, issue the global loads of
// four stages ahead, pre-read the next fragments).  The two workgroup barriers per k-step keep the halves
// in opposite phases, so on every SIMD one wave feeds the matrix pipe while its partner does the VALU /
// LDS / VMEM work.  The stage stream runs across tile boundaries (tiles b, b+G, ... of workgroup b), so the
// loads of the next tile are in flight while the current one finishes: no per-tile prologue bubble.
//
// Activations are read with 16-byte loads along the pixel axis (one wave = one 1 KB row segment; per-lane
// 4-byte gathers cost the same vector-memory issue slots for a quarter of the bytes and were the bottleneck),
// stored row-major [k][pixel] in LDS and transposed into the k-contiguous MFMA B operand by
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10).
constexpr int PP_BM = 128, PP_BN = 256;
constexpr int PP_AKS = PP_BM + 4;                        // 16-byte units between the two k-groups of A
constexpr int PP_BRS = PP_BN + 32;                       // B row stride (bf16): 576 B puts 4 rows on disjoint banks
constexpr int PP_A_BYTES = 2 * PP_AKS * 16, PP_B_BYTES = 16 * PP_BRS * 2;
constexpr int PP_PART_BYTES = PP_A_BYTES + PP_B_BYTES, PP_STAGE_BYTES = 3 * PP_PART_BYTES;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

struct PPRegs { f32x4 b[2]; bf16x8 a[2]; };
struct PPFrags { bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2]; };

__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x = v[j];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi[j] = h; mid[j] = m; lo[j] = (__bf16)r2;
  }
}

// transposed 8-byte LDS read: lane i of each 16-lane group receives column i of the 4 rows addressed by the group
__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* p, int row4_bytes) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + row4_bytes));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(512) void gemm_nn_pp_kernel(pir_gemm_nn_t g, int tiles_m, int tiles_n, int total) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * PP_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wq = wid & 3, tg = tid & 255;
  const int iters = (g.K + XK - 1) / XK;
  const int G = gridDim.x, bid = blockIdx.x;
  const int my_tiles = (total - bid + G - 1) / G;
  const int S = my_tiles * iters;
  const int ldx4 = (int)g.ldx * 4;
  const int a3_part_bytes = g.M * g.a3_kp * 2, a3_step_bytes = g.M * 32;
  const __amdgpu_buffer_rsrc_t ars = pir_make_rsrc(g.A3, (unsigned)(6L * g.M * g.a3_kp));
  const unsigned x_bytes = (unsigned)((((long)g.K - 1) * g.ldx + g.N) * 4);

  // staging roles inside a half (256 threads): B rows 8*grp + wq and + 4 (one wave = one 256-pixel row);
  // A: the half's own 64 rows, both k-groups (lanes 2r, 2r+1 = row r: a wave reads 1 KB contiguous);
  // threads 0-127 carry parts hi and lo, threads 128-255 part mid
  const int a_row = grp * 64 + ((tg & 127) >> 1), a_kg = tg & 1;
  const int a_part0 = __builtin_amdgcn_readfirstlane(tg >> 7);   // 0 -> parts {0, 2}, 1 -> part {1}
  const int b_row = 8 * grp + wq;

  auto tile_coords = [&](int ord, int& m0, int& n0, int& o) __attribute__((always_inline)) {
    const int t = pir_xcd_remap(ord * G + bid, total);
    m0 = (t % tiles_m) * PP_BM;
    const int rest = t / tiles_m;
    n0 = (rest % tiles_n) * PP_BN;
    o = rest / tiles_n;
  };

  // ---- load cursor
  int l_ord = 0, l_k = 0, b_voff = 0, a_voff = 0;
  __amdgpu_buffer_rsrc_t xrs = ars;
  auto set_load_tile = [&](int ord) __attribute__((always_inline)) {
    int m0, n0, o;
    tile_coords(ord, m0, n0, o);
    const int o1 = o / g.O2, o2 = o % g.O2;
    xrs = pir_make_rsrc(g.X + o1 * g.x_s1 + o2 * g.x_s2, x_bytes);
    b_voff = (n0 + lane * 4) * 4;
    const int m = m0 + a_row, mc = m < g.M ? m : g.M - 1;
    a_voff = (mc * 16 + 8 * a_kg) * 2;
  };
  set_load_tile(0);
  // Unconditional on purpose (see gemm_nn_x3_kernel): past the last stage the cursor stays on it and the
  // re-loaded data is never used, so the compiler can count the loads in flight exactly.
  auto issue = [&](PPRegs& R) __attribute__((always_inline)) {
    const int k0 = l_k * XK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      R.b[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, b_voff, (k0 + b_row + 4 * i) * ldx4, 0));
    R.a[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                            ars, a_voff, a_part0 * a3_part_bytes + l_k * a3_step_bytes, 0));
    R.a[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                            ars, a_voff, 2 * a3_part_bytes + l_k * a3_step_bytes, 0));
    if (l_k + 1 < iters) {
      ++l_k;
    } else if (l_ord + 1 < my_tiles) {
      l_k = 0;
      set_load_tile(++l_ord);
    }
  };
  auto stash = [&](int buf, const PPRegs& R) __attribute__((always_inline)) {
    unsigned char* base = smem + buf * PP_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bf16x4 hi, mid, lo;
      split4(R.b[i], hi, mid, lo);
      unsigned char* q = base + PP_A_BYTES + ((b_row + 4 * i) * PP_BRS + lane * 4) * 2;
      *reinterpret_cast<bf16x4*>(q) = hi;
      *reinterpret_cast<bf16x4*>(q + PP_PART_BYTES) = mid;
      *reinterpret_cast<bf16x4*>(q + 2 * PP_PART_BYTES) = lo;
    }
    unsigned char* qa = base + (a_kg * PP_AKS + a_row) * 16;
    if (a_part0 == 0) {
      *reinterpret_cast<bf16x8*>(qa) = R.a[0];
      *reinterpret_cast<bf16x8*>(qa + 2 * PP_PART_BYTES) = R.a[1];
    } else {
      *reinterpret_cast<bf16x8*>(qa + PP_PART_BYTES) = R.a[0];
    }
  };
  const int fr_h = lane >> 5, fr_r = lane & 31;
  const int tr_off = PP_A_BYTES + ((8 * fr_h + ((lane & 15) >> 2)) * PP_BRS + wq * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  auto read_frags = [&](int buf, PPFrags& F) __attribute__((always_inline)) {
    const unsigned char* base = smem + buf * PP_STAGE_BYTES;
    const unsigned char* ap = base + (fr_h * PP_AKS + grp * 64 + fr_r) * 16;
    const unsigned char* bp = base + tr_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      F.ah[i] = *reinterpret_cast<const bf16x8*>(ap + i * 512);
      F.am[i] = *reinterpret_cast<const bf16x8*>(ap + PP_PART_BYTES + i * 512);
      F.al[i] = *reinterpret_cast<const bf16x8*>(ap + 2 * PP_PART_BYTES + i * 512);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      F.bh[j] = tr_read8(bp + j * 64, 4 * PP_BRS * 2);
      F.bm[j] = tr_read8(bp + PP_PART_BYTES + j * 64, 4 * PP_BRS * 2);
      F.bl[j] = tr_read8(bp + 2 * PP_PART_BYTES + j * 64, 4 * PP_BRS * 2);
    }
  };

  f32x16 acc[2][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  int c_ord = 0, c_k = 0;
  auto matrix_phase = [&](const PPFrags& F) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.al[i], F.bh[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah[i], F.bl[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.am[i], F.bm[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.am[i], F.bh[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah[i], F.bm[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah[i], F.bh[j], c, 0, 0, 0);
        acc[i][j] = c;
      }
    if (++c_k == iters && c_ord < my_tiles) {   // tile finished: store this half's 64 rows
      int m0, n0, o;
      tile_coords(c_ord, m0, n0, o);
      const int o1 = o / g.O2, o2 = o % g.O2;
      float* __restrict__ Y = g.Y + o1 * g.y_s1 + o2 * g.y_s2;
      pir_nn_epilogue<2, 2>(acc, g, Y, o1, o2, m0 + grp * 64, n0, 0, wq, lane);
      zero_acc();
      c_k = 0;
      ++c_ord;
    }
  };

  PPRegs R0, R1, R2, R3;   // stage t travels in R[t & 3]: loaded 4 steps before it is staged, staged 2 before use
  PPFrags F;
  issue(R0); issue(R1);
  stash(0, R0); stash(1, R1);
  issue(R2); issue(R3); issue(R0); issue(R1);
  __syncthreads();
  int cbuf = 0, sbuf = 2;
  auto advance = [&]() __attribute__((always_inline)) {
    cbuf = cbuf == 2 ? 0 : cbuf + 1;
    sbuf = sbuf == 2 ? 0 : sbuf + 1;
  };
  // the compiler may move MFMAs (register-only) across s_barrier, which would put both halves' matrix
  // phases side by side: pin the phase boundaries
  auto phase_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };
  // The two halves run separate, branch-free step sequences (two barriers per step each, so the barrier
  // counts match); steps go in fours so that the register sets are compile-time names; a stage count that
  // is not a multiple of four just runs up to three dummy steps on stale data.
  if (grp == 0) {
    read_frags(0, F);
    auto ping = [&](PPRegs& R) __attribute__((always_inline)) {
      matrix_phase(F);
      phase_barrier();
      __builtin_amdgcn_s_setprio(PP_STAGE_PRIO);
      stash(sbuf, R);
      issue(R);
      read_frags(cbuf == 2 ? 0 : cbuf + 1, F);
      __builtin_amdgcn_s_setprio(0);
      phase_barrier();
      advance();
    };
    for (int s = 0; s < S; s += 4) { ping(R2); ping(R3); ping(R0); ping(R1); }
  } else {
    auto pong = [&](PPRegs& R) __attribute__((always_inline)) {
      __builtin_amdgcn_s_setprio(PP_STAGE_PRIO);
      stash(sbuf, R);
      issue(R);
      read_frags(cbuf, F);
      __builtin_amdgcn_s_setprio(0);
      phase_barrier();
      matrix_phase(F);
      phase_barrier();
      advance();
    };
    for (int s = 0; s < S; s += 4) { pong(R2); pong(R3); pong(R0); pong(R1); }
  }
}

int launch_pp(const pir_gemm_nn_t& g, hipStream_t s) {
  const int tiles_m = (int)pir_cdiv(g.M, PP_BM), tiles_n = (int)pir_cdiv(g.N, PP_BN);
  const long total = (long)tiles_m * tiles_n * g.O1 * g.O2;
  const int grid = total < PIR_NUM_CU ? (int)total : PIR_NUM_CU;   // PIR_NUM_CU is a multiple of the XCD count
  hipLaunchKernelGGL(gemm_nn_pp_kernel, dim3(grid), dim3(512), 0, s, g, tiles_m, tiles_n, (int)total);
  return pir_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// "Convert at read" variant for pre-split weights: one persistent 8-wave workgroup per CU, tile 128 x 256,
// every wave 64 x 64.  Global -> LDS copies are LDS-DMA (buffer_load ... lds, 16 bytes per lane): the fp32
// activation rows and the bf16 weight pieces land in a 4-deep LDS ring without touching VGPRs, three stages
// ahead.  There is no staging phase: each wave reads the raw fp32 values of ITS OWN B fragments of the NEXT
// stage, splits them to bf16x3 in registers and reads its A fragments while the matrix pipe works on the
// current stage (the conversion VALU ops are fillers in the shadow of the 32-cycle MFMAs).  One barrier per
// k-step; the stage stream runs across tile boundaries as in the ping-pong kernel.
constexpr int CR_BM = 128, CR_BN = 256, CR_RING = 4;
constexpr int CR_A_BYTES = 3 * 2 * CR_BM * 16;     // [part][k-group][row] 16-byte units
constexpr int CR_B_BYTES = XK * CR_BN * 4;         // [k][column] fp32
constexpr int CR_STAGE = CR_A_BYTES + CR_B_BYTES;
typedef __attribute__((address_space(3))) void* cr_lds_ptr;

struct CRFrags { bf16x8 a[2][3], b[2][3]; };

__global__ __launch_bounds__(512) void gemm_nn_cr_kernel(pir_gemm_nn_t g, int tiles_m, int tiles_n, int total) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CR_RING * CR_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int iters = (g.K + XK - 1) / XK;
  const int G = gridDim.x, bid = blockIdx.x;
  const int my_tiles = (total - bid + G - 1) / G;
  const int S = my_tiles * iters;
  const int ldx4 = (int)g.ldx * 4;
  const int a3_part_bytes = g.M * g.a3_kp * 2, a3_step_bytes = g.M * 32;
  const __amdgpu_buffer_rsrc_t ars = pir_make_rsrc(g.A3, (unsigned)(6L * g.M * g.a3_kp));
  const unsigned x_bytes = (unsigned)((((long)g.K - 1) * g.ldx + g.N) * 4);

  // DMA roles of a wave, four 1 KB copies per stage: B rows 2*wid and 2*wid+1; A chunks q0 = wid and q1 = 8 + (wid & 3)
  // (chunk q: part q/4, k-group (q>>1)&1, rows (q&1)*64 .. +63; waves 4-7 repeat chunks 8-11 so that every wave
  // has exactly four loads per stage in flight and the vmcnt waits below are the same for all)
  const int q0 = wid, q1 = 8 + (wid & 3);
  auto chunk_lds = [&](int q) __attribute__((always_inline)) { return (((q >> 2) * 2 + ((q >> 1) & 1)) * CR_BM + (q & 1) * 64) * 16; };

  auto tile_coords = [&](int ord, int& m0, int& n0, int& o) __attribute__((always_inline)) {
    const int t = pir_xcd_remap(ord * G + bid, total);
    m0 = (t % tiles_m) * CR_BM;
    const int rest = t / tiles_m;
    n0 = (rest % tiles_n) * CR_BN;
    o = rest / tiles_n;
  };

  int l_ord = 0, l_k = 0, b_voff = 0, a_voff0 = 0, a_voff1 = 0;
  __amdgpu_buffer_rsrc_t xrs = ars;
  auto set_load_tile = [&](int ord) __attribute__((always_inline)) {
    int m0, n0, o;
    tile_coords(ord, m0, n0, o);
    const int o1 = o / g.O2, o2 = o % g.O2;
    xrs = pir_make_rsrc(g.X + o1 * g.x_s1 + o2 * g.x_s2, x_bytes);
    b_voff = (n0 + lane * 4) * 4;
    const int r0 = m0 + (q0 & 1) * 64 + lane, r1 = m0 + (q1 & 1) * 64 + lane;
    a_voff0 = ((r0 < g.M ? r0 : g.M - 1) * 16 + 8 * ((q0 >> 1) & 1)) * 2;
    a_voff1 = ((r1 < g.M ? r1 : g.M - 1) * 16 + 8 * ((q1 >> 1) & 1)) * 2;
  };
  set_load_tile(0);
  // always four copies (past the last stage the cursor stays put and the ring slot it refills is never read)
  auto issue = [&](int buf) __attribute__((always_inline)) {
    unsigned char* st = smem + buf * CR_STAGE;
    const int k0 = l_k * XK;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (cr_lds_ptr)(st + CR_A_BYTES + (2 * wid) * (CR_BN * 4)), 16, b_voff,
                                             (k0 + 2 * wid) * ldx4, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (cr_lds_ptr)(st + CR_A_BYTES + (2 * wid + 1) * (CR_BN * 4)), 16, b_voff,
                                             (k0 + 2 * wid + 1) * ldx4, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (cr_lds_ptr)(st + chunk_lds(q0)), 16, a_voff0,
                                             (q0 >> 2) * a3_part_bytes + l_k * a3_step_bytes, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (cr_lds_ptr)(st + chunk_lds(q1)), 16, a_voff1,
                                             (q1 >> 2) * a3_part_bytes + l_k * a3_step_bytes, 0, 0);
  };
  auto advance_cursor = [&]() __attribute__((always_inline)) {   // kept out of the MFMA block (it branches)
    if (l_k + 1 < iters) {
      ++l_k;
    } else if (l_ord + 1 < my_tiles) {
      l_k = 0;
      set_load_tile(++l_ord);
    }
  };

  const int fr_h = lane >> 5, fr_r = lane & 31;
  auto read_a = [&](int buf, CRFrags& F) __attribute__((always_inline)) {
    const unsigned char* ap = smem + buf * CR_STAGE + (fr_h * CR_BM + wm * 64 + fr_r) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int part = 0; part < 3; ++part)
        F.a[i][part] = *reinterpret_cast<const bf16x8*>(ap + (part * 2 * CR_BM + i * 32) * 16);
  };
  auto read_b_raw = [&](int buf, float (&raw)[2][8]) __attribute__((always_inline)) {
    const float* bp = reinterpret_cast<const float*>(smem + buf * CR_STAGE + CR_A_BYTES) + 8 * fr_h * CR_BN + wn * 64 + fr_r;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 8; ++t) raw[j][t] = bp[t * CR_BN + j * 32];
  };
  auto convert_b = [&](const float (&raw)[2][8], CRFrags& F) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const Frag3 fr = split8(raw[j], true);
      F.b[j][0] = fr.hi; F.b[j][1] = fr.mid; F.b[j][2] = fr.lo;
    }
  };

  f32x16 acc[2][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  int c_ord = 0, c_k = 0;
  auto matrix = [&](const CRFrags& F) __attribute__((always_inline)) {
    // term-major: consecutive MFMAs go to different accumulators; F.a/F.b parts: 0 hi, 1 mid, 2 lo
#define PIR_CR_TERM(PA, PB)                                                                  \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][PA], F.b[j][PB], acc[i][j], 0, 0, 0);
    PIR_CR_TERM(2, 0)
    PIR_CR_TERM(0, 2)
    PIR_CR_TERM(1, 1)
    PIR_CR_TERM(1, 0)
    PIR_CR_TERM(0, 1)
    PIR_CR_TERM(0, 0)
#undef PIR_CR_TERM
  };
  auto tile_end = [&]() __attribute__((always_inline)) {
    if (++c_k == iters && c_ord < my_tiles) {
      int m0, n0, o;
      tile_coords(c_ord, m0, n0, o);
      const int o1 = o / g.O2, o2 = o % g.O2;
      float* __restrict__ Y = g.Y + o1 * g.y_s1 + o2 * g.y_s2;
      pir_nn_epilogue<2, 2>(acc, g, Y, o1, o2, m0 + wm * 64, n0, 0, wn, lane);
      zero_acc();
      c_k = 0;
      ++c_ord;
    }
  };

  // prologue: stages 0..2 in flight, stage 0 landed and visible, its fragments in registers
  issue(0); advance_cursor(); issue(1); advance_cursor(); issue(2); advance_cursor();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  CRFrags F0, F1;
  {
    float raw[2][8];
    read_a(0, F0);
    read_b_raw(0, raw);
    convert_b(raw, F0);
  }
  int nbuf = 1, ibuf = 3;   // ring slots of stage s+1 (read in step s) and of stage s+3 (issued in step s)
  auto step = [&](const CRFrags& cur, CRFrags& nxt) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this wave's copies of stage s+1 have landed
    __syncthreads();                                      // ... and everybody's; slot of stage s-1 is free
    // one basic block: copies of stage s+3, LDS reads of stage s+1, the MFMAs of stage s, the conversion of s+1
    issue(ibuf);
    float raw[2][8];
    read_b_raw(nbuf, raw);
    read_a(nbuf, nxt);
    matrix(cur);
    convert_b(raw, nxt);
    // keep the conversion in THIS block (the compiler otherwise sinks it behind the tile-end branch, out of the
    // MFMA shadow): an empty asm that "uses" the converted fragments
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int part = 0; part < 3; ++part) asm volatile("" : "+v"(nxt.b[j][part]));
    // schedule: LDS reads first, then behind every MFMA one copy instruction (while there are any) and a few
    // conversion VALU ops, all in the MFMA's 32-cycle shadow
    __builtin_amdgcn_sched_group_barrier(0x100, 32, 0);
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
    tile_end();
    advance_cursor();
    nbuf = nbuf == CR_RING - 1 ? 0 : nbuf + 1;
    ibuf = ibuf == CR_RING - 1 ? 0 : ibuf + 1;
  };
  for (int s = 0; s < S; s += 2) {
    step(F0, F1);
    step(F1, F0);
  }
}

int launch_cr(const pir_gemm_nn_t& g, hipStream_t s) {
  const int tiles_m = (int)pir_cdiv(g.M, CR_BM), tiles_n = (int)pir_cdiv(g.N, CR_BN);
  const long total = (long)tiles_m * tiles_n * g.O1 * g.O2;
  const int grid = total < PIR_NUM_CU ? (int)total : PIR_NUM_CU;
  hipLaunchKernelGGL(gemm_nn_cr_kernel, dim3(grid), dim3(512), 0, s, g, tiles_m, tiles_n, (int)total);
  return pir_launch_status();
}

template <int TM, int TN, int WM, int WN>
int launch_cfg(const pir_gemm_nn_t& g, hipStream_t s, const X3Conv* conv = nullptr) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const long tiles = pir_cdiv(g.M, BM) * pir_cdiv(g.N, BN);
  dim3 grid((unsigned)tiles, (unsigned)(g.O1 * g.O2)), block(WM * WN * 64);
  X3Conv cv = {0, 0, 0, 0u, 0u};
  if (conv) hipLaunchKernelGGL((gemm_nn_x3_kernel<TM, TN, WM, WN, true, true, true>), grid, block, 0, s, g, *conv);
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

// All weights of a model in one launch: block b handles elements [begin, begin + 4096) of descriptor blk[b].x
// (blk[b].y = begin / 4096).  Element order and layout per descriptor are those of the two kernels above.
__global__ __launch_bounds__(256) void split_bf16x3_batch_kernel(const pir_split_desc_t* __restrict__ descs,
                                                                 const int2* __restrict__ blk) {
  const int2 bd = blk[blockIdx.x];
  const pir_split_desc_t d = descs[bd.x];
  const int kp = (d.K + 15) / 16 * 16, ntap = d.taps ? 9 : 1;
  const long per_tap = (long)d.M * kp, total = ntap * per_tap;
  __bf16* __restrict__ out = reinterpret_cast<__bf16*>(d.out);
  const long base = (long)bd.y * 4096;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const long e = base + i * 256 + threadIdx.x;
    if (e >= total) break;
    const int tap = (int)(e / per_tap);
    const long r = e - tap * per_tap;
    const int m = (int)(r / kp), k = (int)(r % kp);
    const float x = k < d.K ? d.W[(d.flip ? 8 - tap : tap) * d.st + m * d.sm + k * d.sk] : 0.f;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    const long o = tap * per_tap + ((long)(k >> 4) * d.M + m) * 16 + (k & 15);
    out[o] = h; out[total + o] = mid; out[2 * total + o] = (__bf16)r2;
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

extern "C" int pir_conv3x3_x3(const void* A3, int a3_kp, const float* X, long x_bs, float* Y, long y_bs, const float* R,
                              long r_bs, int B, int M, int K, int H, int W, pir_stream_t stream) {
  PIR_CHECK_ARG(A3 && X && Y && B > 0 && M > 0 && K > 0 && H > 0 && W > 0 && B <= 65535);
  PIR_CHECK_ARG(a3_kp == (int)(pir_cdiv(K, 16) * 16));
  PIR_CHECK_ARG((long)(M > K ? M : K) * H * W < (1L << 28) && 54L * M * a3_kp < (1L << 31));
  pir_gemm_nn_t g;
  g.A = nullptr; g.a_s1 = g.a_s2 = 0; g.a_sm = 0; g.a_sk = 0;
  g.X = X; g.x_s1 = x_bs; g.x_s2 = 0; g.ldx = (long)H * W;
  g.Y = Y; g.y_s1 = y_bs; g.y_s2 = 0; g.ldy = (long)H * W;
  g.R = R; g.r_s1 = r_bs; g.r_s2 = 0; g.ldr = (long)H * W;
  g.rowscale = nullptr; g.rs_s1 = g.rs_s2 = 0;
  g.M = M; g.K = K; g.N = H * W; g.O1 = B; g.O2 = 1;
  g.A3 = A3; g.a3_kp = a3_kp;
  X3Conv cv;
  cv.H = H; cv.W = W; cv.ksteps = a3_kp / 16;
  cv.magic_ks = pir_magic((unsigned)cv.ksteps); cv.magic_w = pir_magic((unsigned)W);
  hipStream_t s = (hipStream_t)stream;
  // tile choice as for gemm_nn (rows = output channels, often few)
  if (M <= 32) return launch_cfg<1, 2, 1, 4>(g, s, &cv);
  if (M <= 64) return launch_cfg<2, 2, 1, 4>(g, s, &cv);
  const long pad96 = pir_cdiv(M, 96) * 96, pad128 = pir_cdiv(M, 128) * 128;
  const long blocks128 = pir_cdiv(M, 128) * pir_cdiv(g.N, 128) * B;
  if (pad96 <= pad128 && g.N >= 256) return launch_cfg<3, 2, 1, 4>(g, s, &cv);
  if (blocks128 < 2L * PIR_NUM_CU && pir_cdiv(M, 64) * 64 <= pad128) return launch_cfg<1, 2, 2, 2>(g, s, &cv);
  return launch_cfg<2, 2, 2, 2>(g, s, &cv);
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

// knob: -1 automatic, 0 never, 1 always
bool pir_nn_x3_wanted(const pir_gemm_nn_t* a, int knob) {
  // measured (bench.py, batch 32): the bf16x3 path is at least as fast as fp32 MFMA for every gemm_nn of the
  // train step, with or without pre-split weights, so it is the default; knob 0 forces the fp32 kernel.
  (void)a;
  return knob != 0;
}

int pir_nn_x3_launch(const pir_gemm_nn_t* a, int cfg, hipStream_t s) {
  const pir_gemm_nn_t& g = *a;
  const int M = g.M;
  const long batch = (long)g.O1 * g.O2;
  if (cfg == 5 && g.A3) return launch_pp(g, s);
  if (cfg == 6 && g.A3) return launch_cr(g, s);
  if (cfg >= 0) {
    switch (cfg) {
      case 0: return launch_cfg<1, 2, 1, 4>(g, s);
      case 1: return launch_cfg<2, 2, 1, 4>(g, s);
      case 2: return launch_cfg<3, 2, 1, 4>(g, s);
      case 3: return launch_cfg<2, 2, 2, 2>(g, s);
      case 7: return launch_cfg<3, 1, 1, 4>(g, s);   // 96 x 128
      default: return launch_cfg<1, 2, 2, 2>(g, s);
    }
  }
  // tile choice from the sweep in tools/ktune.py (PIR_X3=1): the 96 x 256 tile (1 KB contiguous per
  // row and stage) wins at the high-resolution levels even with up to ~13 % more padded rows.
  if (M <= 32) return launch_cfg<1, 2, 1, 4>(g, s);
  if (M <= 64) return g.K <= 64 ? launch_cfg<1, 2, 2, 2>(g, s) : launch_cfg<2, 2, 1, 4>(g, s);
  // 96 x 128 (three workgroups per CU) for the GDFN project_in pair at the 96-channel levels: -6 ... -11 % there,
  // neutral or worse for the other full-resolution shapes (tools/cfg7.py)
  if (g.A3 && ((M >= 384 && g.K <= 128 && g.N >= 4096) || (g.K >= 384 && M == 96 && g.N >= 16384)) &&
      pir_cdiv(M, 96) * 96 * 100 <= pir_cdiv(M, 128) * 128 * 113)
    return launch_cfg<3, 1, 1, 4>(g, s);
  const long pad96 = pir_cdiv(M, 96) * 96, pad128 = pir_cdiv(M, 128) * 128;
  const bool use96 = g.N >= 1024 ? pad96 * 100 <= pad128 * 113 : pad96 < pad128;
  const long blocks = use96 ? pir_cdiv(M, 96) * pir_cdiv(g.N, 256) * batch : pir_cdiv(M, 128) * pir_cdiv(g.N, 128) * batch;
  if (blocks < 2L * PIR_NUM_CU && pir_cdiv(M, 64) * 64 <= pad128) return launch_cfg<1, 2, 2, 2>(g, s);
  if (use96) return launch_cfg<3, 2, 1, 4>(g, s);
  return launch_cfg<2, 2, 2, 2>(g, s);
}
