// MDTA backward: dq and dk from ONE pass over q and k (gfx950, bf16x3 split MFMA).
//
//   dq[b,h] = dG[b,h] k[b,h] + alpha_q * q[b,h]          dk[b,h] = dG[b,h]^T q[b,h] + alpha_k * k[b,h]
//   (backward of attn = softmax((q/|q|)(k/|k|)^T t): net/model.py:127-131; dG and the row scales come from
//   pir_mdta_softmax_bwd; q, k, dq, dk are [c][HW] slices of the qkv / dqkv buffers)
//
// As two pir_gemm_nn calls each product reads q AND k (one as operand, one as the scaled residual) - 6 planes per
// channel and block where 4 suffice, and these GEMMs run at their HBM bound (c = 48 rows against 48 k: MFMA busy 0.07).
// Here a wave loads the 48 x 32-pixel blocks of q and k ONCE (16-byte loads), keeps the raw registers - the raw load
// layout IS the store layout, so the scaled residual needs no second load - builds the bf16x3 fragments of k through the
// DPP transpose (wide_tiles.h), multiplies them with dG (split once per workgroup into LDS), stores dq, then does the same
// with q against dG^T for dk.  Row r of the raw load (k-step ks, instruction t) is 16 ks + 8 t + 4 h + j: that is the row
// 32 i + 8 G + 4 h + j an accumulator holds after its transpose, with (i, G) = (ks / 2, 2 (ks % 2) + t); the MFMA's k
// index within a 16-deep step is permuted accordingly (lane half h holds k = 4h .. 4h+3 and 8 + 4h .. 8 + 4h + 3), and the
// dG panels in LDS are written with the same permutation.  c = 48 (every level of the network except the noise blocks) and,
// round 4, c = 96 (the one-head blocks of decoder level 1 and the refinement stage, the 128^2 level: four waves per
// workgroup, one per SIMD with the whole register file - 110 KB of dG panels per workgroup); the noise blocks keep the two GEMMs.
#include "gemm_common.h"
#include "wide_tiles.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct DFrag3 { bf16x8 hi, mid, lo; };

__device__ __forceinline__ DFrag3 dqk_split8(const float (&v)[8]) {
  DFrag3 f;
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

struct DqkArgs {
  const float* dgram;    // [B][heads][48][48]
  const float* q; long q_bs;     // q of (b, h) at q + b * q_bs + h * 48 * HW, rows at stride HW; k at + k_off
  long k_off;
  const float* alpha_q; const float* alpha_k;   // [B][heads * 48]
  float* dq; long dq_bs; long dk_off;
  int heads, HW;
  int nblocks;           // HW / 32
  int per_wg;            // 32-pixel blocks per workgroup (a multiple of 8)
};

// C48: rows per head (48 or 96), KS = C48 / 16 k-steps, PR: padded panel rows (a multiple of 32), NW: waves per workgroup,
// PF: prefetch the next block's planes under this block's MFMAs (needs 2 x KS x 8 more registers)
template <int C48, int KS, int PR, int NW, bool PF>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 2 : 1)))
void mdta_dqk_kernel(DqkArgs p) {
  constexpr int PANEL = 3 * 2 * KS * PR;              // 16-byte units of one split dG panel
  constexpr int NG = C48 / 8, NTI = PR / 32;          // 8-row groups a lane stores, 32-row output tiles
  __shared__ bf16x8 smem[2 * PANEL];                // dG (rows i, k = j) and dG^T (rows j, k = i)
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int qk = ((r >> 4) << 2) | (r & 3), qj = (r >> 2) & 3;   // lane -> (pixel quad, row in the quad group)
  const int bh = blockIdx.y, b = bh / p.heads, hd = bh - b * p.heads;
  const int HW = p.HW;

  // ---- prologue: both orientations of dG as bf16x3 panels.  Unit (kgi = 2 ks + h', row) holds the eight k values
  // 16 ks + 4 h' + (0..3) and 16 ks + 8 + 4 h' + (0..3) - the permutation the activation fragments use.
  const float* __restrict__ dG = p.dgram + (long)bh * C48 * C48;
  for (int u = tid; u < 2 * 2 * KS * PR; u += NW * 64) {
    const int pan = u / (2 * KS * PR), rest = u - pan * (2 * KS * PR);
    const int kgi = rest / PR, row = rest - kgi * PR;
    const int ks = kgi >> 1, hh = kgi & 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * ks + 4 * hh + (e & 3) + 8 * (e >> 2);
      v[e] = row < C48 ? (pan == 0 ? dG[row * C48 + k] : dG[k * C48 + row]) : 0.f;
    }
    const DFrag3 f = dqk_split8(v);
    bf16x8* dst = smem + pan * PANEL + kgi * PR + row;
    dst[0] = f.hi; dst[2 * KS * PR] = f.mid; dst[2 * 2 * KS * PR] = f.lo;
  }
  __syncthreads();

  const int begin = (int)blockIdx.x * p.per_wg;
  const int end = begin + p.per_wg < p.nblocks ? begin + p.per_wg : p.nblocks;
  int blk = begin + wid;
  if (blk >= end) return;

  const float* qb = p.q + (long)b * p.q_bs + (long)hd * C48 * HW;
  const unsigned bytes = (unsigned)((long)C48 * HW * 4);
  const __amdgpu_buffer_rsrc_t qrs = pir_make_rsrc(qb, bytes), krs = pir_make_rsrc(qb + p.k_off, bytes);
  float* dqb = p.dq + (long)b * p.dq_bs + (long)hd * C48 * HW;
  const __amdgpu_buffer_rsrc_t dqrs = pir_make_rsrc(dqb, bytes), dkrs = pir_make_rsrc(dqb + p.dk_off, bytes);
  const int lrow = (4 * h + qj) * HW * 4;            // lane's row offset inside a group of 8 rows, bytes
  // row scales of the rows this lane stores: row(i, G) = 32 i + 8 G + 4 h + qj for the six groups below 48
  float aq[NG], ak[NG];
#pragma unroll
  for (int e = 0; e < NG; ++e) {
    const int row = 8 * e + 4 * h + qj;
    aq[e] = p.alpha_q[(long)b * p.heads * C48 + hd * C48 + row];
    ak[e] = p.alpha_k[(long)b * p.heads * C48 + hd * C48 + row];
  }

  f32x4 qr[KS][2], kr[KS][2];
  auto load = [&](const __amdgpu_buffer_rsrc_t& rs, int bk, f32x4 (&dst)[KS][2]) {
    const int vo = lrow + (bk * 32 + 4 * qk) * 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t)   // rows 16 ks + 8 t + (4 h + qj)
        dst[ks][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + (16 * ks + 8 * t) * HW * 4, 0, 0));
  };
  auto frags = [&](const f32x4 (&raw)[KS][2], DFrag3 (&f)[KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float a0 = raw[ks][t][0], a1 = raw[ks][t][1], a2 = raw[ks][t][2], a3 = raw[ks][t][3];
        res_transpose4(a0, a1, a2, a3);
        v[4 * t] = a0; v[4 * t + 1] = a1; v[4 * t + 2] = a2; v[4 * t + 3] = a3;
      }
      f[ks] = dqk_split8(v);
    }
  };
  // out = panel x fragments + scale * raw, rows < 48: tile i = 0 (groups 0..3), i = 1 (groups 4, 5)
  auto product = [&](int pan, const DFrag3 (&f)[KS], const f32x4 (&raw)[KS][2], const float (&al)[NG],
                     const __amdgpu_buffer_rsrc_t& ors, int bk) {
    const bf16x8* ap = smem + pan * PANEL + h * PR + r + (bk >> 30);   // opaque zero: keeps the reads inside the loop
    f32x16 acc[NTI];
#pragma unroll
    for (int i = 0; i < NTI; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 ah[NTI], am[NTI], al2[NTI];
#pragma unroll
      for (int i = 0; i < NTI; ++i) {
        const int off = ks * 2 * PR + i * 32;
        ah[i] = ap[off]; am[i] = ap[2 * KS * PR + off]; al2[i] = ap[2 * 2 * KS * PR + off];
      }
#pragma unroll
      for (int i = 0; i < NTI; ++i) acc[i] = pir_mfma_x3(ah[i], am[i], al2[i], f[ks].hi, f[ks].mid, f[ks].lo, acc[i]);
    }
    const int vo = lrow + (bk * 32 + 4 * qk) * 4;
#pragma unroll
    for (int e = 0; e < NG; ++e) {
      const int i = e >> 2, G = e & 3;
      float a0 = acc[i][4 * G], a1 = acc[i][4 * G + 1], a2 = acc[i][4 * G + 2], a3 = acc[i][4 * G + 3];
      res_transpose4(a0, a1, a2, a3);
      const f32x4 res = raw[e >> 1][e & 1];          // rows 16 (e / 2) + 8 (e % 2) + 4 h + qj = 8 e + 4 h + qj
      const f32x4 v = {a0 + al[e] * res[0], a1 + al[e] * res[1], a2 + al[e] * res[2], a3 + al[e] * res[3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, vo + 8 * e * HW * 4, 0, 0);
    }
  };

  load(qrs, blk, qr);
  load(krs, blk, kr);
  __builtin_amdgcn_sched_barrier(0);
  for (; blk < end; blk += NW) {
    if constexpr (PF) {
      // the next block's planes fly while this one is multiplied (a wave's 12 KB stay in flight through its MFMAs)
      f32x4 qn[KS][2], kn[KS][2];
      const int nb = blk + NW < end ? blk + NW : blk;
      load(qrs, nb, qn);
      load(krs, nb, kn);
      __builtin_amdgcn_sched_barrier(0);
      DFrag3 f[KS];
      frags(kr, f);
      product(0, f, qr, aq, dqrs, blk);        // dq = dG k + alpha_q q
      frags(qr, f);
      product(1, f, kr, ak, dkrs, blk);        // dk = dG^T q + alpha_k k
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) { qr[ks][t] = qn[ks][t]; kr[ks][t] = kn[ks][t]; }
    } else {
      DFrag3 f[KS];
      frags(kr, f);
      product(0, f, qr, aq, dqrs, blk);
      frags(qr, f);
      product(1, f, kr, ak, dkrs, blk);
      if (blk + NW < end) { load(qrs, blk + NW, qr); load(krs, blk + NW, kr); }
    }
  }
}

int g_dqk96 = 1;   // knob 36: 0 = the 96-row variant off (two GEMMs)

}  // namespace

int pir_mdta_dqk_tune(int knob, int value) {
  if (knob == 36) { g_dqk96 = value; return PIR_OK; }
  return PIR_EINVAL;
}

extern "C" int pir_mdta_dqk(const float* dgram, const float* q, long q_bs, long k_off, const float* alpha_q,
                            const float* alpha_k, float* dq, long dq_bs, long dk_off, int B, int heads, int c, int HW,
                            pir_stream_t stream) {
  PIR_CHECK_ARG(dgram && q && alpha_q && alpha_k && dq && B > 0 && heads > 0 && HW > 0);
  // served: 48 or 96 rows per head, whole 32-pixel blocks, 16-byte aligned planes; the caller keeps the two-GEMM path otherwise
  if ((c != 48 && c != 96) || HW % 32 != 0) return 1000;
  if (c == 96 && !g_dqk96) return 1000;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(dq)) & 15) return 1000;
  if (q_bs % 4 || k_off % 4 || dq_bs % 4 || dk_off % 4) return 1000;
  if ((long)c * HW * 4 >= (1L << 31) || (long)B * heads > 65535) return 1000;
  DqkArgs a;
  a.dgram = dgram; a.q = q; a.q_bs = q_bs; a.k_off = k_off; a.alpha_q = alpha_q; a.alpha_k = alpha_k;
  a.dq = dq; a.dq_bs = dq_bs; a.dk_off = dk_off; a.heads = heads; a.HW = HW;
  a.nblocks = HW / 32;
  const long pairs = (long)B * heads;
  long nch = 2L * PIR_NUM_CU / pairs;                        // about two workgroups per CU in flight
  if (nch < 1) nch = 1;
  const int nw = c == 96 ? 4 : 8;
  if (c == 96) nch = (long)PIR_NUM_CU / pairs > 0 ? (long)PIR_NUM_CU / pairs : 1;   // 110 KB of panels: one workgroup per CU
  const long max_ch = pir_cdiv(a.nblocks, nw);
  if (nch > max_ch) nch = max_ch;
  a.per_wg = (int)(pir_cdiv(pir_cdiv(a.nblocks, nch), nw) * nw);
  const unsigned gx = (unsigned)pir_cdiv(a.nblocks, a.per_wg);
  if (c == 96) hipLaunchKernelGGL((mdta_dqk_kernel<96, 6, 96, 4, true>), dim3(gx, (unsigned)pairs), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((mdta_dqk_kernel<48, 3, 64, 8, true>), dim3(gx, (unsigned)pairs), dim3(512), 0, (hipStream_t)stream, a);
  return pir_launch_status();
}
