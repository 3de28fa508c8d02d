// gemm_nn with the weight panel RESIDENT in LDS and wave-autonomous column streaming (bf16x3 split MFMA, gfx950).
//
//   Y[o][m][n] = sum_k A(m,k) X[o][k][n] (+ rowscale[o][m] * R[o][m][n])          (same contract as pir_gemm_nn)
//
// The 1x1 convolutions of the PromptIR path at the 128^2 / 64^2 levels are short-k GEMMs over very long pixel rows
// (K = 48..255 against 0.5 M columns): HBM-bound, with the matrix cores busy for about half of the stream time.  The
// tiled kernel (gemm_x3.hip) runs a workgroup per output tile: prologue loads, k loop with two barriers per 16-deep
// step, store tail - phases that are serial inside a workgroup and overlap only across the 2-3 workgroups of a CU
// (round-2 ablation: 0.5-0.64 of the HBM roofline with neither HBM nor MFMA saturated).  Here instead
//   * a PERSISTENT workgroup owns one row tile (BM = TM x 32 output channels) and a contiguous range of 32-pixel
//     column blocks; it loads its whole pre-split weight panel [3 parts][K/8][BM][8 bf16] into LDS ONCE;
//   * after that single barrier every WAVE works alone: it loads its column block's activations straight into the
//     MFMA B-operand layout (lane = pixel, 8 consecutive k per lane and k-group: eight 4-byte loads per 16-deep step,
//     each a pair of 128-byte row segments), splits them to bf16x3 in registers, reads the A fragments from the
//     resident panel and issues 6 x TM MFMAs per step; activations never pass through LDS, there are no LDS writes
//     and no barriers in steady state, so the waves of a CU drift apart and one wave's loads / store tail overlap the
//     others' MFMAs;
//   * the loads of the NEXT column block are issued while the current one is computed (a ring of PF k-steps of raw
//     fp32 fragments in registers), so a wave never sees a cold prologue again after its first block.
// Per (m, n) the k order and the order of the six bf16 terms are those of gemm_nn_x3_kernel: results are bit-identical.
#include "gemm_common.h"
#include "wide_tiles.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Frag3 { bf16x8 hi, mid, lo; };

__device__ __forceinline__ Frag3 res_split8(const float (&v)[8]) {
  Frag3 f;
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

struct ResArgs {
  pir_gemm_nn_t g;
  int row_tiles;        // ceil(M / BM)
  int nbpi;             // 32-column blocks per image (N / 32)
  unsigned magic_nbpi;
  int blocks_total;     // O1 * nbpi
  int per_slice;        // column blocks per workgroup
  int slices_per_image; // > 0: per-image weights (A not shared): a workgroup's range lies inside ONE image
};

// k-steps of raw activations a wave keeps in flight (ring slots; must divide KS).  Four row blocks of accumulators
// leave room for three or four slots only.
constexpr int res_pf(int ks, int tm) {
  return tm >= 4 ? (ks % 3 == 0 ? 3 : ks % 4 == 0 ? 4 : 2) : (ks % 6 == 0 ? 6 : ks % 4 == 0 ? 4 : ks % 3 == 0 ? 3 : ks % 2 == 0 ? 2 : 1);
}

// A_PRE: weights come pre-split (pir_split_bf16x3, shared by all images); otherwise A is fp32 with free strides and may
// differ per image (the folded MDTA products, W_eff[b]): split in the prologue.
template <int TM, int KS, int NW, bool A_PRE>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2)))
void gemm_nn_res_kernel(ResArgs p) {
  constexpr int BM = TM * 32, T = NW * 64, KG = 2 * KS, PF = res_pf(KS, TM);
  constexpr int PART = KG * BM;                       // 16-byte units per part
  __shared__ bf16x8 smem[3 * PART];
  const pir_gemm_nn_t& g = p.g;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = pir_xcd_remap(blockIdx.x, gridDim.x);
  const int rt = wg % p.row_tiles, slice = wg / p.row_tiles;
  const int m0 = rt * BM;

  // column-block range of this workgroup
  int begin, end;
  if (p.slices_per_image > 0) {
    const int o = slice / p.slices_per_image, s = slice - o * p.slices_per_image;
    begin = o * p.nbpi + s * p.per_slice;
    end = begin + p.per_slice < (o + 1) * p.nbpi ? begin + p.per_slice : (o + 1) * p.nbpi;
  } else {
    begin = slice * p.per_slice;
    end = begin + p.per_slice < p.blocks_total ? begin + p.per_slice : p.blocks_total;
  }

  // ---- prologue: the weight panel of this row tile, all of K, into LDS
  if constexpr (A_PRE) {
    const __amdgpu_buffer_rsrc_t ars = pir_make_rsrc(g.A3, (unsigned)(6L * g.M * g.a3_kp));
    const int part_bytes = g.M * g.a3_kp * 2;
    for (int u = tid; u < 3 * PART; u += T) {
      const int kg = u & 1, rest = u >> 1;
      const int row = rest % BM, rest2 = rest / BM;
      const int ks = rest2 % KS, part = rest2 / KS;
      const int m = m0 + row, mc = m < g.M ? m : g.M - 1;   // rows beyond M only feed masked outputs
      const bf16x8 v = __builtin_bit_cast(
          bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, part * part_bytes + (ks * g.M + mc) * 32 + kg * 16, 0, 0));
      smem[(part * KS + ks) * 2 * BM + kg * BM + row] = v;
    }
  } else {
    const int o = begin / p.nbpi;
    const float* __restrict__ A = g.A + (long)o * g.a_s1;
    for (int f = tid; f < KG * BM; f += T) {
      const int row = f % BM, kgi = f / BM;               // kgi = 2 * ks + kg
      const int m = m0 + row, mc = m < g.M ? m : g.M - 1;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = kgi * 8 + j;
        v[j] = k < g.K ? A[(long)mc * g.a_sm + (long)k * g.a_sk] : 0.f;
      }
      const Frag3 fr = res_split8(v);
      smem[kgi * BM + row] = fr.hi; smem[PART + kgi * BM + row] = fr.mid; smem[2 * PART + kgi * BM + row] = fr.lo;
    }
  }
  __syncthreads();

  int cb = begin + wid;
  if (cb >= end) return;

  const int h = lane >> 5, r = lane & 31;
  const int ldx4 = (int)g.ldx * 4;
  const unsigned xbytes = (unsigned)((((long)g.K - 1) * g.ldx + g.N) * 4);   // rows beyond K read as 0 (range check)
  const int ldy = (int)g.ldy, ldr = (int)g.ldr;
  const unsigned ybytes = (unsigned)((((long)g.M - 1) * g.ldy + g.N) * 4);
  const unsigned rbytes = (unsigned)((((long)g.M - 1) * g.ldr + g.N) * 4);
  const bool has_r = g.R != nullptr;

  // Activations arrive by 16-byte loads: instruction t (0, 1) of a k-step gives lane (h, l4, l3 l2, l1 l0) the pixels
  // nb + 4m .. 4m + 3, m = (l4 l1 l0), of row 16 ks + 8 h + 4 t + (l3 l2) - eight full 128-byte lines per wave
  // instruction, two instructions per k-step instead of eight 4-byte ones (the vector-memory path is bound by
  // instructions, not bytes: round-3 counters).  res_transpose4 then swaps the register index (pixel in the
  // quad) with lane bits 3, 2 (row): the lane holds rows 4t .. 4t + 3 of ONE pixel, the MFMA B-operand layout, with
  // lane l standing for column 4m + (l3 l2) of the block - a fixed permutation of the 32 columns that the accumulators
  // inherit and the store tail undoes the same way.  The row term sits in the per-lane offset, so rows beyond K (and
  // everything past the image) are zeroed by the range check.
  const int qk = ((r >> 4) << 2) | (r & 3), qj = (r >> 2) & 3;
  f32x4 raw[PF][2];
  struct Cols { __amdgpu_buffer_rsrc_t rs; int vo; };
  auto cols = [&](int b) {
    Cols c;
    const int o = pir_fastdiv(b, p.magic_nbpi), nb = (b - o * p.nbpi) * 32;
    c.rs = pir_make_rsrc(g.X + (long)o * g.x_s1, xbytes);
    c.vo = ((8 * h + qj) * (int)g.ldx + nb + 4 * qk) * 4;
    return c;
  };
  auto load = [&](const Cols& c, int ks, f32x4 (&dst)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      dst[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c.rs, c.vo + (ks * 16 + 4 * t) * ldx4, 0, 0));
  };
  auto split = [&](f32x4 (&src)[2]) {
    float v[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a0 = src[t][0], a1 = src[t][1], a2 = src[t][2], a3 = src[t][3];
      res_transpose4(a0, a1, a2, a3);
      v[4 * t] = a0; v[4 * t + 1] = a1; v[4 * t + 2] = a2; v[4 * t + 3] = a3;
    }
    return res_split8(v);
  };
  Cols cur = cols(cb);
  // issue order pinned (slot 0 first): the waits at the top of the persistent loop are computed for the worse of the
  // two ways into it, and a scrambled prologue order would make them drain everything on every block
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    load(cur, s, raw[s]);
    __builtin_amdgcn_sched_barrier(0);
  }

  const bf16x8* ap0 = smem + h * BM + r;
  // Software pipeline over the flattened (block, k-step) sequence: while the 6 x TM MFMAs of step ks run, the wave
  // reads the weight fragments of step ks + 1 from the panel, splits the activations of step ks + 1 (VALU in the
  // shadow of the MFMAs) and issues the loads that refill the ring slot the previous split emptied.  The scheduler
  // is told the interleave per region (one sched_barrier per k-step): left alone it sinks all loads of a block behind
  // the last MFMA, where their order no longer matches the order of use and the first wait of the next block drains
  // everything.
  bf16x8 ah[TM], am[TM], al[TM];
  auto read_a = [&](const bf16x8* ap, int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = ap[ks * 2 * BM + i * 32];
      am[i] = ap[PART + ks * 2 * BM + i * 32];
      al[i] = ap[2 * PART + ks * 2 * BM + i * 32];
    }
  };
  read_a(ap0, 0);
  Frag3 b = split(raw[0]);
  for (; cb < end; cb += NW) {
    // The weight panel never changes after the prologue, so the compiler would hoist all 3 x TM x KS fragment reads out
    // of the persistent loop (hundreds of registers).  The panel base therefore carries a term it cannot prove zero
    // (block indices stay far below 2^30).  (An empty asm with a memory clobber does the same but makes the waitcnt
    // pass drain vmcnt to 0 behind it.)
    const bf16x8* ap = ap0 + (cb >> 30);
    const Cols nxt = cols(cb + NW < end ? cb + NW : cb);   // last block: re-reads its own (cached) columns, unused
    f32x16 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][0][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
      // refill the slot whose split ran in the previous region
      if (ks + PF < KS) load(cur, ks + PF, raw[ks % PF]);
      else load(nxt, ks + PF - KS, raw[ks % PF]);
      bf16x8 ch_[TM], cm_[TM], cl_[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) { ch_[i] = ah[i]; cm_[i] = am[i]; cl_[i] = al[i]; }
      const Frag3 c = b;
      read_a(ap, (ks + 1) % KS);                      // next step's weight fragments (step 0 again at the block's end)
      b = split(raw[(ks + 1) % PF]);                  // next step's activations (the next block's step 0 at the end)
      // term-major: consecutive MFMAs hit different accumulators; per accumulator the order of the six terms is
      // that of pir_mfma_x3 / gemm_nn_x3_kernel
#define PIR_RES_TERM(A_, B_) \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) \
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_, acc[i][0], 0, 0, 0);
      PIR_RES_TERM(cl_, c.hi)
      PIR_RES_TERM(ch_, c.lo)
      PIR_RES_TERM(cm_, c.mid)
      PIR_RES_TERM(cm_, c.hi)
      PIR_RES_TERM(ch_, c.mid)
      PIR_RES_TERM(ch_, c.hi)
#undef PIR_RES_TERM
      // interleave: weight-fragment reads first, then per MFMA a few transpose / conversion ops and (for the first two) one load
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * TM, 0);
#pragma unroll
      for (int q = 0; q < 6 * TM; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (q < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (68 + 6 * TM - 1) / (6 * TM), 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // Store tail.  An accumulator holds D[channel][pixel] with lane = pixel (under the column permutation above):
    // res_transpose4 of registers 4G .. 4G + 3 (rows 8G + 4h + 0..3 of the 32 x 32 block) gives lane (h, l4, l3 l2, l1 l0)
    // the pixels 4m .. 4m + 3 of row 8G + 4h + (l3 l2), so the block leaves in 4 x 16-byte stores per lane - eight full
    // 128-byte lines per wave instruction - instead of 16 x 4-byte ones; the residual is loaded in that same layout.  Rows sit in the per-lane offset: rows beyond M
    // fall outside the descriptor's range and are dropped (read as 0) by the hardware.
    {
      const int o = pir_fastdiv(cb, p.magic_nbpi), nb = (cb - o * p.nbpi) * 32;
      const __amdgpu_buffer_rsrc_t yrs = pir_make_rsrc(g.Y + (long)o * g.y_s1, ybytes);
      const int vy = ((m0 + 4 * h + qj) * ldy + nb + 4 * qk) * 4;
      const __amdgpu_buffer_rsrc_t rrs = pir_make_rsrc(has_r ? g.R + (long)o * g.r_s1 : g.Y, has_r ? rbytes : 0u);
      const int vr = ((m0 + 4 * h + qj) * ldr + nb + 4 * qk) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f32x4 res[4];
        if (has_r) {
#pragma unroll
          for (int G = 0; G < 4; ++G)
            res[G] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, vr, (i * 32 + 8 * G) * ldr * 4, 0));
        }
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          float a0 = acc[i][0][4 * G], a1 = acc[i][0][4 * G + 1], a2 = acc[i][0][4 * G + 2], a3 = acc[i][0][4 * G + 3];
          res_transpose4(a0, a1, a2, a3);
          f32x4 v = {a0, a1, a2, a3};
          if (has_r) v += res[G];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, pir_row_offset(vy, (i * 32 + 8 * G) * ldy * 4), 0, 0);
        }
      }
    }
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// B-STATIONARY variant for tall weight matrices over short k (project_in / qkv forward, project_out input gradient at
// the 128^2 and 64^2 levels: M = 255 .. 510 rows against K = 96).  In the row-tiled kernels every row tile splits the
// same activations again (4 - 6 times at M = 510) and the conversion VALU work rivals the MFMA time (round-3 counters:
// MFMA busy 0.44, VALU active 0.42).  Here a wave keeps the bf16x3 fragments of ITS 32-pixel column block for all of K
// in registers (12 registers per 16-deep k-step) - split exactly once - and sweeps ALL output rows against them: the
// pre-split weight rows stream through LDS in panels of 128 rows (double-buffered, one barrier per panel, loaded by the
// whole workgroup from L2 while the previous panel is multiplied), two 32 x 32 output tiles at a time, each stored as
// soon as its k loop ends.  Per column block: one split (~400 VALU operations) against 6 x KS x M/32 MFMAs, every
// activation read from HBM exactly once, the output written once in 16-byte stores.
struct BstArgs {
  pir_gemm_nn_t g;
  int nbpi; unsigned magic_nbpi; int blocks_total;
  int per_wg;     // column blocks per workgroup (a multiple of 8: one per wave and round)
  int panels;     // ceil(M / panel height)
  const float* ln_w; const float* ln_b;   // LN variant: gamma, beta [K] of the channel LayerNorm applied on load
  float* mean_out; float* rstd_out;       // LN variant: the statistics [O1][N] for the backward pass (or null)
};

// NT: 32-row output tiles multiplied at a time (2 where registers allow: K = 48); TP: tiles per weight panel (3 or 4:
// the panel height 32 TP is chosen for the fewest padded rows, 96 for M = 288)
// LN: the activations pass through the channel LayerNorm (WithBias: net/model.py:60-63) ON LOAD - a wave holds all K = 16 KS
// channels of its 32 pixels (one half of them per lane half), so mean and variance come from its own registers plus one
// exchange between the lane halves, and the normalised tensor is never written or read (no_grad forward: nothing saves it)
template <int KS, int NT, int TP, bool LN = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void gemm_nn_bst_kernel(BstArgs p) {
  constexpr int NW = 8, BP = 32 * TP, KG = 2 * KS, PART = KG * BP, PANEL = 3 * PART;   // 512 threads
  constexpr int NLD = (3 * KS + 1) / 2;       // passes (16-byte units per thread) per panel: two (part, k-step) slices each
  __shared__ bf16x8 smem[2 * PANEL];
  __shared__ f32x4 lnp[LN ? 2 * KS * 4 : 1];  // gamma then beta, K floats each
  const pir_gemm_nn_t& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int qk = ((r >> 4) << 2) | (r & 3), qj = (r >> 2) & 3;   // lane -> (pixel quad, row in quad group): res kernel
  const int begin = blockIdx.x * p.per_wg;
  const int end = begin + p.per_wg < p.blocks_total ? begin + p.per_wg : p.blocks_total;
  const int rounds = (end - begin + NW - 1) / NW;
  const int ldx4 = (int)g.ldx * 4, ldy = (int)g.ldy;
  const unsigned xbytes = (unsigned)((((long)g.K - 1) * g.ldx + g.N) * 4);
  const unsigned ybytes = (unsigned)((((long)g.M - 1) * g.ldy + g.N) * 4);
  const __amdgpu_buffer_rsrc_t ars = pir_make_rsrc(g.A3, (unsigned)(6L * g.M * g.a3_kp));
  const int part_bytes = g.M * g.a3_kp * 2;

  // ---- weight panel `pi` (rows BP pi ..): 16-byte unit (pass idx, thread) -> registers -> LDS buffer `buf`.
  // A pass covers two (part, k-step) slices c of the panel: thread t holds (kg = t & 1, row = (t >> 1) % BP) for all
  // passes and c = 2 idx + (t >> 1) / BP, which is uniform over a wave (BP = 128: waves 0-3 / 4-7; BP = 96: waves 0-2 /
  // 3-5, waves 6-7 stay out of the panel traffic) - scalar arithmetic and two VGPRs instead of per-pass address registers.
  const int p_kg = tid & 1, p_row = (tid >> 1) % BP;
  const int p_c0 = __builtin_amdgcn_readfirstlane((tid >> 1) / BP);
  const bool p_on = p_c0 < 2;
  auto panel_load = [&](int pi, int idx) {
    const int c0 = p_c0 + 2 * idx;
    const int c = (p_on && c0 < 3 * KS) ? c0 : 0;                    // (a last pass may be partly empty)
    const int ks = c % KS, part = c / KS;
    const int m = pi * BP + p_row, mc = m < g.M ? m : g.M - 1;       // rows beyond M only feed dropped outputs
    // (part, k-step) uniform: scalar offset (rows are clamped, nothing relies on the range check)
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
        ars, mc * 32 + p_kg * 16, part * part_bytes + ks * g.M * 32, 0));
  };
  auto panel_store = [&](int buf, int idx, const bf16x8& v) {
    const int c = p_c0 + 2 * idx;
    if (p_on && c < 3 * KS) smem[buf * PANEL + c * 2 * BP + p_kg * BP + p_row] = v;
  };

  // ---- activations of a column block: 2 x 16-byte loads per k-step (see gemm_nn_res_kernel)
  struct Cols { __amdgpu_buffer_rsrc_t rs; int vo; };
  auto cols = [&](int b) {
    Cols c;
    const int o = pir_fastdiv(b, p.magic_nbpi), nb = (b - o * p.nbpi) * 32;
    c.rs = pir_make_rsrc(g.X + (long)o * g.x_s1, xbytes);
    c.vo = ((8 * h + qj) * (int)g.ldx + nb + 4 * qk) * 4;
    return c;
  };
  f32x4 raw[KS][2];
  auto load_raw = [&](int b) {
    const Cols c = cols(b);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        raw[ks][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c.rs, c.vo, (ks * 16 + 4 * t) * ldx4, 0));   // row term scalar
  };

  // first panel and first block
  {
#pragma unroll
    for (int idx = 0; idx < NLD; ++idx) panel_store(0, idx, panel_load(0, idx));
  }
  int my = begin + wid;                       // this wave's column block in the current round
  load_raw(my < end ? my : end - 1);
  if constexpr (LN) {
    float* lf = reinterpret_cast<float*>(lnp);
    if (tid < 16 * KS) { lf[tid] = p.ln_w[tid]; lf[16 * KS + tid] = p.ln_b[tid]; }
  }
  __syncthreads();

  const bf16x8* ap0 = smem + h * BP + r;
  int gp = 0;                                 // panels consumed so far: panel gp sits in buffer gp & 1
  for (int round = 0; round < rounds; ++round, my += NW) {
    const bool active = my < end;
    // ---- this round's fragments: transpose (+ LayerNorm) + split, once
    Frag3 bf[KS];
    if constexpr (LN) {
      float val[KS][8];
      float s1 = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float a0 = raw[ks][t][0], a1 = raw[ks][t][1], a2 = raw[ks][t][2], a3 = raw[ks][t][3];
          res_transpose4(a0, a1, a2, a3);
          val[ks][4 * t] = a0; val[ks][4 * t + 1] = a1; val[ks][4 * t + 2] = a2; val[ks][4 * t + 3] = a3;
          s1 += (a0 + a1) + (a2 + a3);
        }
      s1 += __shfl_xor(s1, 32, 64);                       // the other lane half holds the other 8 KS channels of this pixel
      const float mu = s1 / (float)(16 * KS);
      float s2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { val[ks][e] -= mu; s2 += val[ks][e] * val[ks][e]; }
      s2 += __shfl_xor(s2, 32, 64);
      const float rstd = 1.f / sqrtf(s2 / (float)(16 * KS) + 1e-5f);   // biased variance, eps inside the root (:62-63)
      if (p.mean_out && active && h == 0) {                            // lane r stands for pixel 4 qk + qj of the block
        const int so = pir_fastdiv(my, p.magic_nbpi);
        const long at = (long)so * g.N + (my - so * p.nbpi) * 32 + 4 * qk + qj;
        p.mean_out[at] = mu; p.rstd_out[at] = rstd;
      }
      const f32x4* lp = lnp + 2 * h + (round >> 30);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // one base register + immediate offsets (separately computed addresses were hoisted out of the round loop as
          // 24 registers and spilled; the reloads in front of every read serialised the round start)
          const f32x4 gm = lp[4 * ks + t], bt = lp[4 * KS + 4 * ks + t];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * t + e] = val[ks][4 * t + e] * rstd * gm[e] + bt[e];
        }
        bf[ks] = res_split8(v);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float a0 = raw[ks][t][0], a1 = raw[ks][t][1], a2 = raw[ks][t][2], a3 = raw[ks][t][3];
          res_transpose4(a0, a1, a2, a3);
          v[4 * t] = a0; v[4 * t + 1] = a1; v[4 * t + 2] = a2; v[4 * t + 3] = a3;
        }
        bf[ks] = res_split8(v);
      }
    }
    const int o = pir_fastdiv(active ? my : end - 1, p.magic_nbpi), nb = ((active ? my : end - 1) - o * p.nbpi) * 32;
    const __amdgpu_buffer_rsrc_t yrs = pir_make_rsrc(g.Y + (long)o * g.y_s1, active ? ybytes : 0u);   // idle wave: every store dropped
    const int vy = ((4 * h + qj) * ldy + nb + 4 * qk) * 4;

    for (int pi = 0; pi < p.panels; ++pi, ++gp) {
      const int buf = gp & 1;
      // (the panel after the very last one is loaded too - panel 0 once more, unused: keeps the pipeline free of branches,
      // whose merges would make every wait a full drain)
      const int nxt = pi + 1 < p.panels ? pi + 1 : 0;
      if (pi + 1 == p.panels && round + 1 < rounds) {
        __builtin_amdgcn_sched_barrier(0);
        load_raw(my + NW < end ? my + NW : end - 1);                  // next round's activations fly during the last panel
        __builtin_amdgcn_sched_barrier(0);
      }
      const bf16x8* ap = ap0 + buf * PANEL + (gp >> 30);
      // (4 / NT) x KS iterations (TP / NT groups of NT 32-row tiles x k-steps), software-pipelined: iteration j multiplies with the
      // weight fragments read during j - 1, reads those of j + 1, issues the load of one 16-byte unit of the NEXT panel
      // and writes the unit issued PD iterations earlier into the other LDS buffer.
      constexpr int NIT = (TP / NT) * KS;
      // iterations between the load of a panel unit and its LDS write (an L2 round trip under load); the LayerNorm
      // variant has fewer registers to park units in
      constexpr int PD = LN ? (NIT / 6 > 2 ? NIT / 6 : 2) : NIT / 2;
      bf16x8 ah[NT], am[NT], al[NT];
      auto read_a = [&](int j) {
        const int pair = j / KS, ks = j % KS;
#pragma unroll
        for (int e = 0; e < NT; ++e) {
          const int off = ks * 2 * BP + (NT * pair + e) * 32;
          ah[e] = ap[off]; am[e] = ap[PART + off]; al[e] = ap[2 * PART + off];
        }
      };
      read_a(0);
      bf16x8 stage[NLD];
      f32x16 acc[NT];
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        const int ks = j % KS;
        if (ks == 0) {
#pragma unroll
          for (int e = 0; e < NT; ++e)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[e][q] = 0.f;
        }
        if (j < NLD) stage[j] = panel_load(nxt, j);
        if (j >= PD && j - PD < NLD) panel_store(buf ^ 1, j - PD, stage[j - PD]);
        bf16x8 ch[NT], cm[NT], cl[NT];
#pragma unroll
        for (int e = 0; e < NT; ++e) { ch[e] = ah[e]; cm[e] = am[e]; cl[e] = al[e]; }
        if (j + 1 < NIT) read_a(j + 1);
#define PIR_BST_TERM(A_, B_) \
        _Pragma("unroll") for (int e = 0; e < NT; ++e) \
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[e], B_, acc[e], 0, 0, 0);
        PIR_BST_TERM(cl, bf[ks].hi)
        PIR_BST_TERM(ch, bf[ks].lo)
        PIR_BST_TERM(cm, bf[ks].mid)
        PIR_BST_TERM(cm, bf[ks].hi)
        PIR_BST_TERM(ch, bf[ks].mid)
        PIR_BST_TERM(ch, bf[ks].hi)
#undef PIR_BST_TERM
        if (ks == KS - 1) {
          // store the finished tiles (rows 128 pi + 32 (NT pair + e) ..): quad-transposed 16-byte stores, see
          // gemm_nn_res_kernel
          const int pair = j / KS;
#pragma unroll
          for (int e = 0; e < NT; ++e) {
            const int mrow = pi * BP + (NT * pair + e) * 32;
#pragma unroll
            for (int G = 0; G < 4; ++G) {
              float a0 = acc[e][4 * G], a1 = acc[e][4 * G + 1], a2 = acc[e][4 * G + 2], a3 = acc[e][4 * G + 3];
              res_transpose4(a0, a1, a2, a3);
              f32x4 v = {a0, a1, a2, a3};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, pir_row_offset(vy, (mrow + 8 * G) * ldy * 4), 0, 0);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = (NIT - PD > 0 ? NIT - PD : 0); q < NLD; ++q) panel_store(buf ^ 1, q, stage[q]);   // write slots behind the last iteration
      __syncthreads();   // the next panel is complete in the other buffer; everyone is done reading this one
    }
  }
}

int g_res_tm = 0, g_res_wgs = 0;   // development overrides (knobs 21, 23): 0 = automatic
int g_bst_mode = -1;   // knob 24 (B-stationary kernel): -1 automatic, 0 never, 1 whenever the shape is served
int g_res_mode = -1;   // knob 20: -1 automatic, 0 never, 1 whenever the shape is served

template <int TM, int KS, int NW, bool A_PRE>
int res_launch(const ResArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((gemm_nn_res_kernel<TM, KS, NW, A_PRE>), dim3((unsigned)grid), dim3(NW * 64), 0, s, a);
  return pir_launch_status();
}

// waves per workgroup by LDS need: panels above 80 KB allow one workgroup per CU (eight waves), smaller ones two or
// three workgroups of four
constexpr int res_nw(int tm, int ks) { return (long)tm * ks * 2 * 32 * 16 * 3 > 80 * 1024 ? 8 : 4; }

template <int KS, bool A_PRE>
int res_dispatch_tm(int tm, const ResArgs& a, int grid, hipStream_t s) {
  constexpr int UNIT = KS * 2 * 32 * 16 * 3;   // LDS bytes per 32 rows
  if constexpr (4 * UNIT <= 160 * 1024) { if (tm == 4) return res_launch<4, KS, res_nw(4, KS), A_PRE>(a, grid, s); }
  if constexpr (3 * UNIT <= 160 * 1024) { if (tm == 3) return res_launch<3, KS, res_nw(3, KS), A_PRE>(a, grid, s); }
  if constexpr (2 * UNIT <= 160 * 1024) { if (tm == 2) return res_launch<2, KS, res_nw(2, KS), A_PRE>(a, grid, s); }
  return 1000;
}

}  // namespace

int pir_nn_res_tune(int knob, int value) {
  if (knob == 20) { g_res_mode = value; return PIR_OK; }
  return PIR_EINVAL;
}

int pir_nn_res_tune2(int knob, int value) {
  if (knob == 21) g_res_tm = value; else if (knob == 23) g_res_wgs = value; else if (knob == 24) g_bst_mode = value; else return PIR_EINVAL;
  return PIR_OK;
}

// Plan: TM (row tile = TM x 32), NW (waves per workgroup) or 0 when the shape is not served by the resident kernel.
struct ResPlan { int tm, nw, ks, grid, per_slice, row_tiles, spi; bool pre; };

static bool res_plan(const pir_gemm_nn_t& g, ResPlan& pl) {
  if (g_res_mode == 0) return false;
  if (g.O2 != 1 || g.N % 32 != 0 || g.N < 32 || g.rowscale != nullptr) return false;
  // 32-bit byte offsets inside one image of Y / R (rows padded to the row tile)
  if ((long)(g.M + 128) * g.ldy * 4 >= (1L << 31) || (g.R && (long)(g.M + 128) * g.ldr * 4 >= (1L << 31))) return false;
  // 16-byte stores / residual loads: image bases and row strides of Y and R in multiples of four floats
  if ((reinterpret_cast<uintptr_t>(g.Y) & 15) || g.ldy % 4 || g.y_s1 % 4) return false;
  if (g.R && ((reinterpret_cast<uintptr_t>(g.R) & 15) || g.ldr % 4 || g.r_s1 % 4)) return false;
  if ((reinterpret_cast<uintptr_t>(g.X) & 15) || g.ldx % 4 || g.x_s1 % 4) return false;
  const bool shared = g.a_s1 == 0 && g.a_s2 == 0;
  pl.pre = g.A3 != nullptr && shared;
  if (!pl.pre && g.A3 != nullptr) return false;
  if (!pl.pre && g.K != 48 && g.K != 96 && g.K != 192) return false;
  const int kp = (int)(pir_cdiv(g.K, 16) * 16);
  if (pl.pre && g.a3_kp != kp) return false;
  pl.ks = kp / 16;
  if (pl.ks != 3 && pl.ks != 6 && pl.ks != 8 && pl.ks != 9 && pl.ks != 12 && pl.ks != 16) return false;
  if ((kp * g.ldx + g.N) * 4 >= (1L << 31)) return false;                 // signed 32-bit byte offsets inside one image
  if (6L * g.M * kp >= (1L << 31)) return false;
  const long unit = (long)pl.ks * 2 * 32 * 16 * 3;                        // LDS bytes per 32 rows of the panel
  // row tile: the largest that fits LDS (at least two workgroups of four waves per CU when it can be had) with the
  // fewest padded rows
  int best = 0; long best_pad = 1L << 60;
  for (int tm = 4; tm >= 2; --tm) {
    if (tm * unit > 160 * 1024) continue;
    const long pad = pir_cdiv(g.M, tm * 32) * tm * 32;
    if (pad < best_pad) { best_pad = pad; best = tm; }
  }
  if (g_res_tm) best = g_res_tm;
  if (best < 2 || best * unit > 160 * 1024) return false;
  pl.tm = best;
  const long lds = best * unit;
  pl.nw = lds > 80 * 1024 ? 8 : 4;                                        // = res_nw(tm, ks)
  const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);
  pl.row_tiles = (int)pir_cdiv(g.M, best * 32);
  const int nbpi = g.N / 32;
  const long blocks_total = (long)g.O1 * nbpi;
  long wgs = g_res_wgs ? g_res_wgs : (long)PIR_NUM_CU * per_cu;           // resident workgroups aimed at
  long slices = wgs / pl.row_tiles;
  if (slices < 1) slices = 1;
  if (!shared) {   // per-image weights: slices inside one image
    long spi = slices / g.O1;
    if (spi < 1) spi = 1;
    if (spi > nbpi) spi = nbpi;
    pl.spi = (int)spi;
    pl.per_slice = (int)pir_cdiv(nbpi, spi);
    pl.spi = (int)pir_cdiv(nbpi, pl.per_slice);
    slices = (long)pl.spi * g.O1;
  } else {
    pl.spi = 0;
    if (slices > blocks_total) slices = blocks_total;
    pl.per_slice = (int)pir_cdiv(blocks_total, slices);
    slices = pir_cdiv(blocks_total, pl.per_slice);
  }
  // every wave should get a few blocks, or the one-off panel load is not amortised: automatic mode only takes long streams
  if (g_res_mode < 0 && pl.per_slice < 2 * pl.nw) return false;
  // ... and only the shape class where the A/B shows a gain over the tiled kernel (tools/resident_ab.py): three row
  // blocks of 96 rows against K = 96 on full-resolution planes (qkv forward of the 96-channel 128^2 levels, M = 288)
  if (g_res_mode < 0 && !(pl.pre && pl.ks == 6 && g.M > 256 && g.M <= 384 && g.N >= 16384 && g.R == nullptr)) return false;
  pl.grid = (int)(slices * pl.row_tiles);
  return true;
}

static bool bst_plan(const pir_gemm_nn_t& g, BstArgs& a, int& grid, bool ln = false) {
  if (g_bst_mode == 0) return false;
  if (g.O2 != 1 || g.N % 32 != 0 || g.N < 32 || g.rowscale != nullptr || g.R != nullptr) return false;
  if (g.A3 == nullptr || g.a_s1 != 0 || g.a_s2 != 0) return false;
  const int kp = (int)(pir_cdiv(g.K, 16) * 16);
  if (g.a3_kp != kp || (kp != 48 && kp != 96)) return false;
  // automatic use where the A/B over the step's shapes shows a gain (tools/resident_ab.py, profiles/r03_resident_ab.txt):
  // project_in forward (M = 254 / 510) at the 128^2 and 64^2 levels, project_out input gradient (M = 255) at 64^2
  if (!ln && g_bst_mode < 0 && !(g.M >= 384 || (g.M >= 250 && g.M <= 256 && (kp == 48 || g.N <= 4096)))) return false;
  if (!ln && g_bst_mode < 0 && g.N < 4096) return false;
  if (ln && (g.K != kp || g.M < 96)) return false;           // the fused LayerNorm needs all of K in the wave: no k tail
  if ((reinterpret_cast<uintptr_t>(g.Y) & 15) || g.ldy % 4 || g.y_s1 % 4) return false;
  if (g.R && ((reinterpret_cast<uintptr_t>(g.R) & 15) || g.ldr % 4 || g.r_s1 % 4)) return false;
  if ((reinterpret_cast<uintptr_t>(g.X) & 15) || g.ldx % 4 || g.x_s1 % 4) return false;
  if ((long)(g.M + 128) * g.ldy * 4 >= (1L << 31) || (g.R && (long)(g.M + 128) * g.ldr * 4 >= (1L << 31))) return false;
  if ((kp * g.ldx + g.N) * 4 >= (1L << 31) || 6L * g.M * kp >= (1L << 31)) return false;
  a.g = g;
  a.nbpi = g.N / 32;
  a.magic_nbpi = pir_magic((unsigned)a.nbpi);
  a.blocks_total = g.O1 * a.nbpi;
  if ((long)a.blocks_total * (a.nbpi > 1 ? a.nbpi : 2) >= (1L << 32)) return false;
  // panel height 96 or 128 rows, whichever pads M less (K = 48 keeps 128: its two-tile groups need an even count)
  const int bp = (kp == 96 && pir_cdiv(g.M, 96) * 96 < pir_cdiv(g.M, 128) * 128) ? 96 : 128;
  a.panels = (int)pir_cdiv(g.M, bp);
  long wgs = g_res_wgs ? g_res_wgs : PIR_NUM_CU;            // one eight-wave workgroup per CU
  long per = pir_cdiv(pir_cdiv(a.blocks_total, wgs), 8) * 8;
  if (g_bst_mode < 0 && per < (ln ? 8 : 16)) return false;  // at least two rounds per workgroup (one where a LayerNorm launch is replaced too)
  a.per_wg = (int)per;
  grid = (int)pir_cdiv(a.blocks_total, per);
  return true;
}

int pir_nn_bst_launch(const pir_gemm_nn_t* g, hipStream_t s, const float* ln_w = nullptr, const float* ln_b = nullptr,
                      float* mean_out = nullptr, float* rstd_out = nullptr) {
  BstArgs a;
  int grid = 0;
  const bool ln = ln_w != nullptr;
  if (!bst_plan(*g, a, grid, ln)) return 1000;
  a.ln_w = ln_w; a.ln_b = ln_b; a.mean_out = mean_out; a.rstd_out = rstd_out;
  if (ln) {
    const bool q96 = (long)a.panels * 96 >= g->M && (long)a.panels * 96 < pir_cdiv(g->M, 128) * 128 && g->a3_kp == 96;
    if (g->a3_kp == 96 && q96) hipLaunchKernelGGL((gemm_nn_bst_kernel<6, 1, 3, true>), dim3((unsigned)grid), dim3(512), 0, s, a);
    else if (g->a3_kp == 96) hipLaunchKernelGGL((gemm_nn_bst_kernel<6, 1, 4, true>), dim3((unsigned)grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm_nn_bst_kernel<3, 2, 4, true>), dim3((unsigned)grid), dim3(512), 0, s, a);
    return pir_launch_status();
  }
  const bool p96 = (long)a.panels * 96 >= g->M && (long)a.panels * 96 < pir_cdiv(g->M, 128) * 128 && g->a3_kp == 96;
  if (g->a3_kp == 96 && p96) hipLaunchKernelGGL((gemm_nn_bst_kernel<6, 1, 3>), dim3((unsigned)grid), dim3(512), 0, s, a);
  else if (g->a3_kp == 96) hipLaunchKernelGGL((gemm_nn_bst_kernel<6, 1, 4>), dim3((unsigned)grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((gemm_nn_bst_kernel<3, 2, 4>), dim3((unsigned)grid), dim3(512), 0, s, a);
  return pir_launch_status();
}

// 0: not served, 1: resident-panel kernel, 2: B-stationary kernel
int pir_nn_res_kind(const pir_gemm_nn_t* a) {
  BstArgs b;
  int grid;
  if (bst_plan(*a, b, grid)) return 2;
  ResPlan pl;
  return res_plan(*a, pl) ? 1 : 0;
}

// y = W LayerNorm(x) for the no_grad forward: 1000 = shape not served (nothing launched)
extern "C" int pir_ln_conv1x1_fwd(const float* x, long x_bs, const float* ln_w, const float* ln_b, const void* A3, int a3_kp,
                                  float* y, long y_bs, float* mean_out, float* rstd_out, int B, int M, int K, int HW,
                                  pir_stream_t stream) {
  PIR_CHECK_ARG(x && ln_w && ln_b && A3 && y && B > 0 && M > 0 && K > 0 && HW > 0 && (mean_out == nullptr) == (rstd_out == nullptr));
  pir_gemm_nn_t g;
  g.A = nullptr; g.a_s1 = g.a_s2 = 0; g.a_sm = K; g.a_sk = 1;
  g.X = x; g.x_s1 = x_bs; g.x_s2 = 0; g.ldx = HW;
  g.Y = y; g.y_s1 = y_bs; g.y_s2 = 0; g.ldy = HW;
  g.R = nullptr; g.r_s1 = g.r_s2 = 0; g.ldr = 0;
  g.rowscale = nullptr; g.rs_s1 = g.rs_s2 = 0;
  g.M = M; g.K = K; g.N = HW; g.O1 = B; g.O2 = 1; g.A3 = A3; g.a3_kp = a3_kp;
  return pir_nn_bst_launch(&g, (hipStream_t)stream, ln_w, ln_b, mean_out, rstd_out);
}

int pir_nn_res_launch(const pir_gemm_nn_t* a, hipStream_t s) {
  const pir_gemm_nn_t& g = *a;
  {
    const int st = pir_nn_bst_launch(a, s);
    if (st != 1000) return st;
  }
  ResPlan pl;
  if (!res_plan(g, pl)) return 1000;
  ResArgs ra;
  ra.g = g;
  ra.row_tiles = pl.row_tiles;
  ra.nbpi = g.N / 32;
  ra.magic_nbpi = pir_magic((unsigned)ra.nbpi);
  ra.blocks_total = g.O1 * ra.nbpi;
  ra.per_slice = pl.per_slice;
  ra.slices_per_image = pl.spi;
  if ((long)ra.blocks_total * (ra.nbpi > 1 ? ra.nbpi : 2) >= (1L << 32)) return 1000;   // pir_fastdiv bound
  // per-image fp32 weights (the folded MDTA products) only occur with K = C in {48, 96, 192}
  if (!pl.pre) {
    switch (pl.ks) {
      case 3: return res_dispatch_tm<3, false>(pl.tm, ra, pl.grid, s);
      case 6: return res_dispatch_tm<6, false>(pl.tm, ra, pl.grid, s);
      case 12: return res_dispatch_tm<12, false>(pl.tm, ra, pl.grid, s);
      default: return 1000;
    }
  }
  switch (pl.ks) {
    case 3: return res_dispatch_tm<3, true>(pl.tm, ra, pl.grid, s);
    case 6: return res_dispatch_tm<6, true>(pl.tm, ra, pl.grid, s);
    case 8: return res_dispatch_tm<8, true>(pl.tm, ra, pl.grid, s);
    case 9: return res_dispatch_tm<9, true>(pl.tm, ra, pl.grid, s);
    case 12: return res_dispatch_tm<12, true>(pl.tm, ra, pl.grid, s);
    case 16: return res_dispatch_tm<16, true>(pl.tm, ra, pl.grid, s);
    default: return 1000;
  }
}
