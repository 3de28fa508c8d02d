// gemm_nn, C-STATIONARY: few output rows (M = 48, 96 or 192) against a long k (bf16x3 split MFMA, gfx950).
//
//   Y[o][m][n] = sum_k A(m,k) X[o][k][n] (+ R[o][m][n])                       (same contract as pir_gemm_nn)
//
// The input gradients of project_in / qkv and the forward project_out of the GDFN at the 96-channel levels are GEMMs with
// M = 96 output rows, K = 255 .. 510 and very long pixel rows.  In the tiled kernel (gemm_x3.hip, 96 x 128 tile) a
// workgroup lives for one tile: prologue loads, K/16 barrier-separated steps, store tail.  Here
//   * a wave owns ALL M rows of its 32-pixel column block: TM accumulators stay in registers for the whole k loop
//     (C-stationary) and every activation is loaded (16-byte loads + DPP transpose, wide_tiles.h), split to bf16x3 and
//     multiplied by exactly one wave - no activation passes through LDS, each is converted once;
//   * the pre-split weights stream through LDS in k-panels of PK 16-deep steps x M rows (all three parts), double
//     buffered, loaded by the whole workgroup (registers -> LDS) while the previous panel is multiplied: one barrier per
//     PK x 6 x TM MFMAs per wave;
//   * workgroups are persistent (one per CU, eight waves): the raw activations of the next k-steps - across panel and
//     column-block boundaries - are always PF steps ahead in a register ring, so only a wave's very first block sees a
//     cold start.
// Per (m, n) the k order and the order of the six bf16 terms are those of gemm_nn_x3_kernel: results are bit-identical.
#include "gemm_common.h"
#include "wide_tiles.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct CstArgs {
  pir_gemm_nn_t g;
  int nbpi; unsigned magic_nbpi; int blocks_total;
  int per_wg;     // column blocks per workgroup (a multiple of the wave count: one per wave and round)
  int panels;     // k-panels: a3_kp / 16 / PK
  // LNB variant (input gradient of the convolution behind a WithBias LayerNorm, fused with that LayerNorm's backward):
  const float* lx; long lx_bs;            // the LayerNorm's input x [B][M][N]
  const float* mean; const float* rstd;   // [B][N]
  const float* gamma;                     // [M]
  const float* dres; long dres_bs;        // gradient arriving over the residual connection (or null)
  float* ws;                              // [grid][2][M] partial sums of dgamma, dbeta
};

// sum over the four lanes that differ in lane bits 3, 2 (two DPP row rotations), then over the two lane halves
__device__ __forceinline__ float cst_sum_rows(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  return v + __shfl_xor(v, 32, 64);
}

// LNB: the product is the gradient dy of a WithBias channel LayerNorm's output (net/model.py:60-63); a wave holds all M
// channels of its 32 pixels, so the LayerNorm backward runs in its registers - with d = dy, xn = (x - mean) rstd,
// g = d gamma: dx = rstd (g - mean_c(g) - xn mean_c(g xn)) + dres, dgamma += d xn, dbeta += d - and dy is never written.
// MR: the real row count (48: two 32-row MFMA tiles whose rows 48 .. 63 are zero in LDS and dropped at the store)
// SP = 2 (LNB, 192 rows): the rows of a column block are split between TWO waves (96 each, both stream the same dy block):
// twice the waves for the few column blocks of a part batch at the 32^2 level (512 blocks against 1024 SIMDs), half the MFMA
// chain per wave; the two channel sums of the LayerNorm backward are completed through LDS
template <int TM, int PK, int NW, bool LNB = false, int MR = 32 * TM, int SP = 1>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4)))
void gemm_nn_cst_kernel(CstArgs p) {
  constexpr int BM = 32 * TM, PRW = BM * SP, T = NW * 64, PF = PK % 4 == 0 ? 4 : 3, NWB = NW / SP;   // NWB: column blocks per round
  static_assert(SP == 1 || (LNB && MR == PRW), "row split: fused LayerNorm backward without padding rows");
  constexpr int PUG = PK * MR * 2;            // 16-byte units per part of a panel in global memory (MR rows)
  constexpr int PU = PK * PRW * 2;            // 16-byte units per part of a panel
  constexpr int PANEL = 3 * PU;
  constexpr int NLD = (PANEL + T - 1) / T;    // units per thread and panel
  constexpr int NIT = PK * TM;                // (k-step, row tile) iterations per panel
  constexpr int PD = LNB ? (NIT / 4 > 0 ? NIT / 4 : 1) : NIT / 2; // iterations between the load of a panel unit and its LDS write (LNB: fewer registers to park units in)
  constexpr int LPI = (NLD + NIT - 2) / (NIT - 1);   // panel units loaded per (k-step, row tile) iteration
  static_assert(PK % PF == 0 && NLD <= LPI * (NIT - 1), "panel staging");
  __shared__ bf16x8 smem[2 * PANEL];
  const pir_gemm_nn_t& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int qk = ((r >> 4) << 2) | (r & 3), qj = (r >> 2) & 3;   // lane -> (pixel quad, row in the quad group)
  const int begin = blockIdx.x * p.per_wg;
  const int end = begin + p.per_wg < p.blocks_total ? begin + p.per_wg : p.blocks_total;
  const int rounds = (end - begin + NWB - 1) / NWB;
  const int half = SP == 1 ? 0 : wid % SP;    // which BM rows of the block this wave owns
  const int ldx4 = (int)g.ldx * 4, ldy = (int)g.ldy, ldr = (int)g.ldr;
  const unsigned xbytes = (unsigned)((((long)g.K - 1) * g.ldx + g.N) * 4);   // rows beyond K read as 0 (range check)
  const unsigned ybytes = (unsigned)((((long)g.M - 1) * g.ldy + g.N) * 4);
  const unsigned rbytes = (unsigned)((((long)g.M - 1) * g.ldr + g.N) * 4);
  const bool has_r = g.R != nullptr;
  const __amdgpu_buffer_rsrc_t ars = pir_make_rsrc(g.A3, (unsigned)(6L * MR * g.a3_kp));
  const int part_bytes = MR * g.a3_kp * 2;

  // ---- weight panel `pi` (k-steps PK pi ..): pir_split_bf16x3 stores [part][k-step][row][k-group][8 bf16], so the three
  // parts of a panel are three contiguous runs of PU units; unit (idx, thread) -> registers -> LDS [part][k-step][k-group][row]
  auto panel_load = [&](int pi, int idx) {
    const int u0 = tid + idx * T, u = ((3 * PUG) % T == 0 || u0 < 3 * PUG) ? u0 : 0;
    const int part = (u >= PUG) + (u >= 2 * PUG), w = u - part * PUG;
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, part * part_bytes + w * 16, pi * (PK * MR * 32), 0));
  };
  auto panel_store = [&](int buf, int idx, const bf16x8& v) {
    const int u = tid + idx * T;
    const int part = (u >= PUG) + (u >= 2 * PUG), w = u - part * PUG;
    const int kg = w & 1, rw = (w >> 1) % MR, ksl = (w >> 1) / MR;
    if ((3 * PUG) % T == 0 || u < 3 * PUG) smem[buf * PANEL + part * PU + (ksl * 2 + kg) * PRW + rw] = v;
  };

  // ---- activations of a column block: 2 x 16-byte loads per k-step (layout: gemm_nn_res_kernel)
  struct Cols { __amdgpu_buffer_rsrc_t rs; int vo; };
  auto cols = [&](int b0) {
    const int b = b0 < end ? b0 : end - 1;
    Cols c;
    const int o = pir_fastdiv(b, p.magic_nbpi), nb = (b - o * p.nbpi) * 32;
    c.rs = pir_make_rsrc(g.X + (long)o * g.x_s1, xbytes);
    c.vo = ((8 * h + qj) * (int)g.ldx + nb + 4 * qk) * 4;
    return c;
  };
  f32x4 raw[PF][2];
  auto load = [&](const __amdgpu_buffer_rsrc_t& rs, int vo, int k, f32x4 (&dst)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      dst[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 4 * t * ldx4, k * 16 * ldx4, 0));
  };
  auto split = [&](f32x4 (&src)[2]) {
    float v[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a0 = src[t][0], a1 = src[t][1], a2 = src[t][2], a3 = src[t][3];
      res_transpose4(a0, a1, a2, a3);
      v[4 * t] = a0; v[4 * t + 1] = a1; v[4 * t + 2] = a2; v[4 * t + 3] = a3;
    }
    return pir_split8(v);
  };

  // first panel, first k-steps of the first block
  if constexpr (MR < PRW) {  // the padding rows of both buffers stay zero for the whole kernel
    const bf16x8 z = {};
    for (int u = tid; u < 2 * PANEL; u += T) smem[u] = z;
    __syncthreads();
  }
#pragma unroll
  for (int idx = 0; idx < NLD; ++idx) panel_store(0, idx, panel_load(0, idx));
  int my = begin + wid / SP;                  // this wave's column block in the current round (clamped by cols())
  Cols cur = cols(my);
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    load(cur.rs, cur.vo, s, raw[s]);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  pir_frag3 b = split(raw[0]);

  // LNB: gamma of the 4 TM rows this lane holds in the store layout, its partial sums of dgamma / dbeta
  __shared__ float gsm[LNB ? PRW : 1];
  __shared__ f32x4 xch[SP == 2 ? NW * 64 * 2 : 1];   // SP = 2: the channel sums of the partner wave
  float pw[TM][4], pb[TM][4];
  if constexpr (LNB) {
    if (tid < PRW) gsm[tid] = tid < MR ? p.gamma[tid] : 0.f;  // (visible behind the prologue's barrier)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int G = 0; G < 4; ++G) { pw[i][G] = 0.f; pb[i][G] = 0.f; }
  }

  const bf16x8* ap0 = smem + h * PRW + half * BM + r;
  int gp = 0;                                 // panels consumed so far: panel gp sits in buffer gp & 1
  for (int round = 0; round < rounds; ++round, my += NWB) {
    const Cols nxt = cols(my + NWB);          // (last round: its own block again, unused)
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    for (int pi = 0; pi < p.panels; ++pi, ++gp) {
      const int buf = gp & 1;
      const bool last = pi + 1 == p.panels;
      const int npi = last ? 0 : pi + 1;      // (the panel behind the very last one is loaded too, unused: no branches)
      // the panel base carries a term the compiler cannot prove zero, or it hoists every fragment read out of the loops
      const bf16x8* ap = ap0 + buf * PANEL + (gp >> 30);
      bf16x8 an[3];
      auto read_a = [&](int it) {
        const int ksl = it / TM, i = it % TM;
#pragma unroll
        for (int q = 0; q < 3; ++q) an[q] = ap[q * PU + ksl * 2 * PRW + i * 32];
      };
      read_a(0);
      bf16x8 stage[NLD];
#pragma unroll
      for (int ksl = 0; ksl < PK; ++ksl) {
        __builtin_amdgcn_sched_barrier(0);
        // refill the ring slot whose split ran in the previous step: k-step PK pi + ksl + PF of this block or, behind
        // the end of k, of the next block
        if (ksl + PF >= PK) {
          const __amdgpu_buffer_rsrc_t rs = last ? nxt.rs : cur.rs;
          const int vo = last ? nxt.vo : cur.vo;
          load(rs, vo, last ? ksl + PF - PK : pi * PK + ksl + PF, raw[ksl % PF]);
        } else {
          load(cur.rs, cur.vo, pi * PK + ksl + PF, raw[ksl % PF]);
        }
        const pir_frag3 c = b;
        b = split(raw[(ksl + 1) % PF]);        // next step's activations (the next panel's / block's first at the end)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int it = ksl * TM + i;
          const bf16x8 ah = an[0], am = an[1], al = an[2];
          if (it + 1 < NIT) read_a(it + 1);
#pragma unroll
          for (int q = it * LPI; q < (it + 1) * LPI; ++q)
            if (q < NLD) stage[q] = panel_load(npi, q);
          if (it >= PD) {
#pragma unroll
            for (int q = (it - PD) * LPI; q < (it - PD + 1) * LPI; ++q)
              if (q < NLD) panel_store(buf ^ 1, q, stage[q]);
          }
          acc[i] = pir_mfma_x3(ah, am, al, c.hi, c.mid, c.lo, acc[i]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = (NIT - PD > 0 ? NIT - PD : 0) * LPI; q < NLD; ++q) panel_store(buf ^ 1, q, stage[q]);
      __syncthreads();   // the next panel is complete in the other buffer; everyone is done reading this one
    }

    // ---- store tail (see gemm_nn_res_kernel): registers 4G .. 4G + 3 of an accumulator transposed against lane bits
    // 3, 2 leave as one 16-byte store per lane; an idle wave's stores are dropped by an empty descriptor
    if constexpr (LNB) {
      const bool active = my < end;
      const float actf = active ? 1.f : 0.f;
      const int bb = active ? my : end - 1;
      const int o = pir_fastdiv(bb, p.magic_nbpi), nb = (bb - o * p.nbpi) * 32;
      const __amdgpu_buffer_rsrc_t yrs = pir_make_rsrc(g.Y + (long)o * g.y_s1, active ? ybytes : 0u);
      const int vy = ((half * BM + 4 * h + qj) * ldy + nb + 4 * qk) * 4;
      const int n4 = g.N * 4;
      const unsigned pbytes = (unsigned)((long)MR * n4);
      const __amdgpu_buffer_rsrc_t xrs = pir_make_rsrc(p.lx + (long)o * p.lx_bs, pbytes);
      const __amdgpu_buffer_rsrc_t drs = pir_make_rsrc(p.dres ? p.dres + (long)o * p.dres_bs : p.lx, p.dres ? pbytes : 0u);
      const __amdgpu_buffer_rsrc_t mrs = pir_make_rsrc(p.mean + (long)o * g.N, (unsigned)n4);
      const __amdgpu_buffer_rsrc_t srs = pir_make_rsrc(p.rstd + (long)o * g.N, (unsigned)n4);
      const int vx = ((half * BM + 4 * h + qj) * g.N + nb + 4 * qk) * 4;
      const f32x4 mu = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, (nb + 4 * qk) * 4, 0, 0));
      const f32x4 rs = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (nb + 4 * qk) * 4, 0, 0));
      f32x4 xv[TM][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int G = 0; G < 4; ++G)
          xv[i][G] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vx, (i * 32 + 8 * G) * n4, 0));
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          float a0 = acc[i][4 * G], a1 = acc[i][4 * G + 1], a2 = acc[i][4 * G + 2], a3 = acc[i][4 * G + 3];
          res_transpose4(a0, a1, a2, a3);          // pixels 4 qk .. + 3 of channel 32 i + 8 G + 4 h + qj
          const f32x4 d = {a0, a1, a2, a3};
          const f32x4 xn = (xv[i][G] - mu) * rs;
          const f32x4 t = d * xn;
          pw[i][G] += actf * ((t[0] + t[1]) + (t[2] + t[3]));
          pb[i][G] += actf * ((d[0] + d[1]) + (d[2] + d[3]));
          const f32x4 gg = d * gsm[half * BM + 32 * i + 8 * G + 4 * h + qj];
          s1 += gg; s2 += gg * xn;
          xv[i][G] = xn;
          acc[i][4 * G] = gg[0]; acc[i][4 * G + 1] = gg[1]; acc[i][4 * G + 2] = gg[2]; acc[i][4 * G + 3] = gg[3];
        }
      f32x4 m1, m2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { m1[e] = cst_sum_rows(s1[e]); m2[e] = cst_sum_rows(s2[e]); }
      if constexpr (SP == 2) {   // the other half of the channels: the partner wave's sums (every wave of the workgroup is here)
        xch[(wid * 64 + lane) * 2] = m1; xch[(wid * 64 + lane) * 2 + 1] = m2;
        __syncthreads();
        m1 += xch[((wid ^ 1) * 64 + lane) * 2]; m2 += xch[((wid ^ 1) * 64 + lane) * 2 + 1];
      }
      m1 *= 1.f / (float)MR; m2 *= 1.f / (float)MR;
      // the residual gradient arrives one row tile ahead of its use (all of it at once would not fit the registers)
      f32x4 rv[2][4];
#pragma unroll
      for (int G = 0; G < 4; ++G)
        rv[0][G] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, vx, (8 * G) * n4, 0));
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < TM) {
#pragma unroll
          for (int G = 0; G < 4; ++G)
            rv[(i + 1) & 1][G] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, vx, ((i + 1) * 32 + 8 * G) * n4, 0));
        }
#pragma unroll
        for (int G = 0; G < 4; ++G) {
          const f32x4 gg = {acc[i][4 * G], acc[i][4 * G + 1], acc[i][4 * G + 2], acc[i][4 * G + 3]};
          const f32x4 v = rs * (gg - m1 - xv[i][G] * m2) + rv[i & 1][G];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, pir_row_offset(vy, (i * 32 + 8 * G) * ldy * 4), 0, 0);
        }
      }
    } else {
      const bool active = my < end;
      const int bb = active ? my : end - 1;
      const int o = pir_fastdiv(bb, p.magic_nbpi), nb = (bb - o * p.nbpi) * 32;
      const __amdgpu_buffer_rsrc_t yrs = pir_make_rsrc(g.Y + (long)o * g.y_s1, active ? ybytes : 0u);
      const int vy = ((4 * h + qj) * ldy + nb + 4 * qk) * 4;
      const __amdgpu_buffer_rsrc_t rrs = pir_make_rsrc(has_r ? g.R + (long)o * g.r_s1 : g.Y, has_r ? rbytes : 0u);
      const int vr = ((4 * h + qj) * ldr + nb + 4 * qk) * 4;
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
          float a0 = acc[i][4 * G], a1 = acc[i][4 * G + 1], a2 = acc[i][4 * G + 2], a3 = acc[i][4 * G + 3];
          res_transpose4(a0, a1, a2, a3);
          f32x4 v = {a0, a1, a2, a3};
          if (has_r) v += res[G];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, pir_row_offset(vy, (i * 32 + 8 * G) * ldy * 4), 0, 0);
        }
      }
    }
    cur = nxt;
  }
  if constexpr (LNB) {
    // this workgroup's row of the dgamma / dbeta partials: lanes that differ in the pixel quad (lane bits 4, 1, 0) hold
    // the same channels; then the eight waves through LDS (the panels are dead behind the last barrier)
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        float w = pw[i][G], bsum = pb[i][G];
        w += __shfl_xor(w, 1, 64); bsum += __shfl_xor(bsum, 1, 64);
        w += __shfl_xor(w, 2, 64); bsum += __shfl_xor(bsum, 2, 64);
        w += __shfl_xor(w, 16, 64); bsum += __shfl_xor(bsum, 16, 64);
        if (qk == 0) {
          const int row = half * BM + 32 * i + 8 * G + 4 * h + qj;
          red[(wid * 2) * PRW + row] = w; red[(wid * 2 + 1) * PRW + row] = bsum;
        }
      }
    __syncthreads();
    for (int t = tid; t < 2 * MR; t += T) {
      const int which = t / MR, c = t - which * MR;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (SP == 1 || w % SP == c / BM) sum += red[(w * 2 + which) * PRW + c];   // (SP = 2: the waves that own this half)
      p.ws[(long)blockIdx.x * 2 * MR + t] = sum;
    }
  }
}

int g_cst_mode = -1;   // knob 26: -1 automatic, 0 never, 1 whenever the shape is served
int g_cst_split = -1;     // knob 32 (-1: automatic - where the launch has at most one column block per CU: batch-8 step -0.45 ms, round 4): 192-channel fused LayerNorm backward with the rows of a block split between two waves
                          // (isolated 0.78 of the one-wave variant's time, no effect inside the two-stream step: off)
int g_cst_ln_maxc = 192;   // knob 27: most channels the fused LayerNorm backward serves (A/B of the 192-channel variant)

inline bool cst_split_on(const pir_gemm_nn_t& g) {
  return g_cst_split < 0 ? (long)g.O1 * (g.N / 32) <= PIR_NUM_CU : g_cst_split != 0;
}

bool cst_plan(const pir_gemm_nn_t& g, CstArgs& a, int& grid, int& pk, bool lnb = false) {
  if (g_cst_mode == 0 && !lnb) return false;
  if (g.O2 != 1 || g.N % 32 != 0 || g.N < 32 || g.rowscale != nullptr) return false;
  if (g.A3 == nullptr || g.a_s1 != 0 || g.a_s2 != 0) return false;
  if (g.M != 96 && g.M != 192 && g.M != 48) return false;
  const int kp = (int)(pir_cdiv(g.K, 16) * 16), ks = kp / 16;
  if (g.a3_kp != kp || ks < (g.M == 48 ? 8 : 12)) return false;
  // 96 rows: eight waves, panels of 8 or 6 k-steps; 192 rows: four waves (all registers of a SIMD to one wave), panels of 4;
  // fused LayerNorm backward at 192 rows: the rows of a block split between two waves, two blocks per round and workgroup
  const int nw = g.M == 96 || g.M == 48 ? 8 : (lnb && cst_split_on(g) ? 2 : 4);
  pk = g.M == 192 ? (ks % 4 == 0 ? 4 : 0) : g.M == 48 ? (ks % 8 == 0 ? 8 : ks == 9 ? 9 : 0) : ks % 8 == 0 ? 8 : ks % 6 == 0 ? 6 : 0;
  if (!pk) return false;
  // plain 192-row products (32^2 level) gain nothing over the tiled kernel (tools/cst_ab.py: 0.99-1.05 at batch 32, half
  // the chip idle at batch 16): automatic only inside the fused LayerNorm backward
  if (g.M == 192 && !lnb && g_cst_mode < 0) return false;
  if ((reinterpret_cast<uintptr_t>(g.Y) & 15) || g.ldy % 4 || g.y_s1 % 4) return false;
  if (g.R && ((reinterpret_cast<uintptr_t>(g.R) & 15) || g.ldr % 4 || g.r_s1 % 4)) return false;
  if ((reinterpret_cast<uintptr_t>(g.X) & 15) || g.ldx % 4 || g.x_s1 % 4) return false;
  if ((long)(g.M + 128) * g.ldy * 4 >= (1L << 31) || (g.R && (long)(g.M + 128) * g.ldr * 4 >= (1L << 31))) return false;
  if (((long)kp * g.ldx + g.N) * 4 >= (1L << 31) || 6L * g.M * kp >= (1L << 31)) return false;
  a = CstArgs{};
  a.g = g;
  a.nbpi = g.N / 32;
  a.magic_nbpi = pir_magic((unsigned)a.nbpi);
  a.blocks_total = g.O1 * a.nbpi;
  if ((long)a.blocks_total * (a.nbpi > 1 ? a.nbpi : 2) >= (1L << 32)) return false;
  a.panels = ks / pk;
  const long per = pir_cdiv(pir_cdiv(a.blocks_total, (long)PIR_NUM_CU), nw) * nw;   // one workgroup per CU
  a.per_wg = (int)per;
  grid = (int)pir_cdiv(a.blocks_total, per);
  return true;
}

}  // namespace

int pir_nn_cst_tune(int knob, int value) {
  if (knob == 26) g_cst_mode = value; else if (knob == 27) g_cst_ln_maxc = value; else if (knob == 32) g_cst_split = value; else return PIR_EINVAL;
  return PIR_OK;
}

bool pir_nn_cst_serves(const pir_gemm_nn_t* g) {
  CstArgs a; int grid, pk;
  return cst_plan(*g, a, grid, pk);
}

int pir_nn_cst_launch(const pir_gemm_nn_t* g, hipStream_t s) {
  CstArgs a; int grid = 0, pk = 0;
  if (!cst_plan(*g, a, grid, pk)) return 1000;
  if (g->M == 48 && pk == 9) hipLaunchKernelGGL((gemm_nn_cst_kernel<2, 9, 8, false, 48>), dim3((unsigned)grid), dim3(512), 0, s, a);
  else if (g->M == 48 && pk == 8) hipLaunchKernelGGL((gemm_nn_cst_kernel<2, 8, 8, false, 48>), dim3((unsigned)grid), dim3(512), 0, s, a);
  else if (g->M == 48) return 1000;
  else if (g->M == 192) hipLaunchKernelGGL((gemm_nn_cst_kernel<6, 4, 4>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else if (pk == 8) hipLaunchKernelGGL((gemm_nn_cst_kernel<3, 8, 8>), dim3((unsigned)grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((gemm_nn_cst_kernel<3, 6, 8>), dim3((unsigned)grid), dim3(512), 0, s, a);
  return pir_launch_status();
}

int pir_reduce_partials_to2(const float* parts, long stride, int S, float alpha, int accumulate, float* out, float* out2,
                            long split, long count, pir_stream_t stream);   // misc.hip

// dx = LayerNormBackward(W^T dy | x, mean, rstd, gamma) + dres, dweight, dbias in one pass (WithBias LayerNorm in front of a
// 1x1 convolution with weight W [K][C], pre-split as the input-gradient operand): the gradient of the normalised tensor
// is never written.  1000 = shape not served (nothing launched; the caller runs pir_gemm_nn + pir_layernorm_bwd).
extern "C" int pir_conv1x1_dgrad_ln_bwd(const float* dy, long dy_bs, const void* A3, int a3_kp, int K,
                                        const float* x, long x_bs, const float* ln_w, const float* mean, const float* rstd,
                                        const float* dres, long dres_bs, float* dx, long dx_bs, float* dweight, float* dbias,
                                        float* ws, size_t ws_floats, int B, int C, int HW, pir_stream_t stream) {
  PIR_CHECK_ARG(dy && A3 && x && ln_w && mean && rstd && dx && dweight && dbias && ws && B > 0 && C > 0 && K > 0 && HW > 0);
  pir_gemm_nn_t g;
  g.A = nullptr; g.a_s1 = g.a_s2 = 0; g.a_sm = K; g.a_sk = 1;
  g.X = dy; g.x_s1 = dy_bs; g.x_s2 = 0; g.ldx = HW;
  g.Y = dx; g.y_s1 = dx_bs; g.y_s2 = 0; g.ldy = HW;
  g.R = nullptr; g.r_s1 = g.r_s2 = 0; g.ldr = 0;
  g.rowscale = nullptr; g.rs_s1 = g.rs_s2 = 0;
  g.M = C; g.K = K; g.N = HW; g.O1 = B; g.O2 = 1; g.A3 = A3; g.a3_kp = a3_kp;
  CstArgs a; int grid = 0, pk = 0;
  if (C > g_cst_ln_maxc) return 1000;
  if (!cst_plan(g, a, grid, pk, true)) return 1000;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || x_bs % 4 || (reinterpret_cast<uintptr_t>(mean) & 15) || (reinterpret_cast<uintptr_t>(rstd) & 15)) return 1000;
  if (dres && ((reinterpret_cast<uintptr_t>(dres) & 15) || dres_bs % 4)) return 1000;
  if ((size_t)grid * 2 * C > ws_floats) return PIR_ENOMEM;
  a.lx = x; a.lx_bs = x_bs; a.mean = mean; a.rstd = rstd; a.gamma = ln_w; a.dres = dres; a.dres_bs = dres_bs; a.ws = ws;
  hipStream_t s = (hipStream_t)stream;
  if (C == 48 && pk == 9) hipLaunchKernelGGL((gemm_nn_cst_kernel<2, 9, 8, true, 48>), dim3((unsigned)grid), dim3(512), 0, s, a);
  else if (C == 48 && pk == 8) hipLaunchKernelGGL((gemm_nn_cst_kernel<2, 8, 8, true, 48>), dim3((unsigned)grid), dim3(512), 0, s, a);
  else if (C == 48) return 1000;
  else if (C == 192 && cst_split_on(g)) hipLaunchKernelGGL((gemm_nn_cst_kernel<3, 4, 4, true, 192, 2>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else if (C == 192) hipLaunchKernelGGL((gemm_nn_cst_kernel<6, 4, 4, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else if (pk == 8) hipLaunchKernelGGL((gemm_nn_cst_kernel<3, 8, 8, true>), dim3((unsigned)grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((gemm_nn_cst_kernel<3, 6, 8, true>), dim3((unsigned)grid), dim3(512), 0, s, a);
  const int st = pir_launch_status();
  if (st) return st;
  return pir_reduce_partials_to2(ws, 2L * C, grid, 1.f, 0, dweight, dbias, C, 2L * C, stream);
}
