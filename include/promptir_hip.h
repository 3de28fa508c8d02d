/*
 * promptir_hip.h — C ABI of libpromptir_hip.so (gfx950 / MI355X).
 *
 * The reference (kongwanbianjinyu/PromptIR) has no FFI: its operator surface is
 * torch.nn.Module (net/model.py) and every op is an ATen dispatch.  This header
 * is the boundary we place UNDER that module surface: one entry point per
 * ATen op (or fused group of ops) the reference's hot path issues.  Each
 * declaration cites the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *  - all tensors are fp32, device pointers, NCHW; a "plane" is H*W contiguous floats
 *  - `*_bs` arguments are batch strides in ELEMENTS, so producers can write into /
 *    consumers can read from channel slices of a larger (concat / qkv) buffer
 *  - no allocation, no synchronisation inside; work is enqueued on `stream`
 *    (a hipStream_t passed as void*); workspaces are passed in by the caller
 *  - return value: 0 on success, negative on bad arguments (PIR_E*), a positive
 *    hipError_t if a launch failed.  Never aborts.
 */
#ifndef PROMPTIR_HIP_H
#define PROMPTIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIR_OK 0
#define PIR_EINVAL (-22)
#define PIR_ENOMEM (-12) /* workspace too small */

typedef void* pir_stream_t;

/* library identity: returns the ABI version (bumped on any signature change) */
int pir_abi_version(void);
/* name of the code-object architecture the kernels were built for ("gfx950") */
const char* pir_arch(void);

/* tuning / A-B knob used by tools/ktune.py: knob 0 = gemm_nn tile config, 1 = gemm_nt tile config,
 * 2 = gemm_nt split count, 3 / 4 = force (1) or forbid (0) the bf16x3 matrix-core path of gemm_nn / gemm_nt;
 * value -1 (0 for knob 2) restores the built-in policy.  6 = rows per band of the register-only GDFN backward
 * (0 = automatic), 7 = 1 disables that kernel (A/B against the LDS-tiled one).  Process-wide development switches:
 * set them before the first launch, never while another thread is launching. */
int pir_tune_set(int knob, int value);

/* Self-description of the LOADED library, derived from compile-time state: bit 0 = a diagnostic macro was defined
 * (kernels with pipeline components removed, results garbage; those macros no longer exist in the sources and
 * pir_common.h refuses them, so a product build has the bit clear), bits 8..23 = the ABI version the object was
 * compiled with, bit 24 = built with --offload-arch=gfx950.  tests/test_cabi.py checks all three against the binding.
 * Every remaining pir_tune_set knob selects between kernels / plans with identical results (tile plan, split count,
 * band height); none skips work. */
int pir_build_flags(void);

/* ------------------------------------------------------------------ GEMM core
 * Batched  Y[o][m][n] = sum_k A[o](m,k) * X[o][k][n]  (+ rowscale[o][m] * R[o][m][n])
 *   o = o1*O2 + o2 (two-level batch so that (batch, head) slices of a qkv buffer
 *   can be addressed without copies); A(m,k) = A[m*a_sm + k*a_sk].
 * Replaces every 1x1 nn.Conv2d of the path — Attention.qkv / project_out
 * (net/model.py:111,113,120,137), FeedForward.project_in / project_out (:88,92,95,98),
 * the reduce_* channel mixers (:294,296,303,305,313) — their input gradients
 * (A = W^T), `attn @ v` (:133) and the dq/dk/dv products of its backward.
 * rowscale==NULL means 1; R==NULL means no residual (TransformerBlock adds, :193-194).
 */
typedef struct {
  const float* A; long a_s1, a_s2; long a_sm, a_sk;
  const float* X; long x_s1, x_s2; long ldx;
  float* Y; long y_s1, y_s2; long ldy;
  const float* R; long r_s1, r_s2; long ldr;
  const float* rowscale; long rs_s1, rs_s2;
  int M, K, N;
  int O1, O2;
  /* optional: A pre-split into three bf16 pieces by pir_split_bf16x3 (layout [3][a3_kp/16][M][16], unbatched).
   * When set, the bf16x3 matrix-core path reads it instead of splitting A on the fly. NULL otherwise. */
  const void* A3; int a3_kp;
} pir_gemm_nn_t;
int pir_gemm_nn(const pir_gemm_nn_t* args, pir_stream_t stream);
/* The same product with a scratch buffer of pir_gemm_nn_ws_floats(args) floats (0: none needed).  Where the launch would leave
 * most CUs idle behind a long k loop (the 384-row 1x1 convolutions of the 16^2 level, net/model.py:88,92,111,113 at
 * k = 1021 ... 2042) the k loop is cut into slices that run side by side and a deterministic second stage adds the partial
 * sums in order: same result to fp32 rounding.  Needs pre-split weights (A3) and one contiguous output per image. */
size_t pir_gemm_nn_ws_floats(const pir_gemm_nn_t* args);
int pir_gemm_nn_ws(const pir_gemm_nn_t* args, float* ws, size_t ws_floats, pir_stream_t stream);
/* y[b] = W LayerNorm_c(x[b]) for the no_grad forward (round 3): the channel LayerNorm (WithBias, net/model.py:60-63) is
 * applied as the activations are loaded by the persistent B-stationary kernel, so the normalised tensor of
 * `self.attn(self.norm1(x))` / `self.ffn(self.norm2(x))` (:192-196) is never written or read.  A3 = pir_split_bf16x3 of
 * W [M][K].  Served for K = 48 or 96 channels, HW % 32 == 0, 16-byte aligned planes and enough pixels to fill the
 * chip; returns 1000 (nothing launched) otherwise and the caller runs pir_layernorm_fwd + pir_gemm_nn.  Used by the
 * no_grad forward and, with the statistics written out, by the training forward of the 128^2 levels. */
int pir_ln_conv1x1_fwd(const float* x, long x_bs, const float* ln_w, const float* ln_b, const void* A3, int a3_kp,
                       float* y, long y_bs, float* mean_out, float* rstd_out, int B, int M, int K, int HW, pir_stream_t stream);
/* Weight gradient of that convolution without the normalised tensor in memory (training): dw[co][ci] = sum_{b,p}
 * dy[b][co][p] * LayerNorm(x[b])[ci][p], the LayerNorm applied from (mean, rstd, ln_w, ln_b) as gemm_nt_xp_kernel stages
 * its shared operand.  mean_out / rstd_out of pir_ln_conv1x1_fwd (both or neither; [B][HW]) are the statistics this and
 * pir_conv1x1_dgrad_ln_bwd read.  Returns 1000 (nothing launched) where the tall-operand-private kernel does not serve the
 * shape: the caller then materialises LayerNorm(x) (pir_layernorm_fwd) and calls pir_gemm_nt. */
int pir_conv1x1_wgrad_ln(const float* dy, long dy_bs, const float* x, long x_bs, const float* mean, const float* rstd,
                         const float* ln_w, const float* ln_b, float* dw, float* ws, size_t ws_floats,
                         int B, int Cout, int Cin, int HW, pir_stream_t stream);

/* Input gradient of a 1x1 convolution that follows a WithBias channel LayerNorm, fused with that LayerNorm's backward
 * (round 3): dx[b] = LN'(W^T dy[b] | x[b], mean, rstd, ln_w) + dres[b], dweight, dbias of the LayerNorm
 * (net/model.py:60-63 behind :192-196: `self.attn(self.norm1(x))`, `self.ffn(self.norm2(x))`).  The persistent
 * C-stationary kernel (gemm_cst.hip) holds all C channels of a pixel block in one wave, so the gradient of the normalised
 * tensor never leaves its registers (2 of the 4 + 2 C-planes of the unfused pair are not moved).  A3 = pir_split_bf16x3
 * of W [K][C] as the input-gradient operand.  ws: at least 512 C floats.  Served for C = 48, 96 or 192 with K (padded to 16) a
 * multiple of the kernel's panel depth, HW % 32 == 0, 16-byte aligned planes; returns 1000 (nothing launched) otherwise and the caller
 * runs pir_gemm_nn + pir_layernorm_bwd. */
int pir_conv1x1_dgrad_ln_bwd(const float* dy, long dy_bs, const void* A3, int a3_kp, int K,
                             const float* x, long x_bs, const float* ln_w, const float* mean, const float* rstd,
                             const float* dres, long dres_bs, float* dx, long dx_bs, float* dweight, float* dbias,
                             float* ws, size_t ws_floats, int B, int C, int HW, pir_stream_t stream);

/* Host-only query: which kernel instantiation pir_gemm_nn would launch for `args` (pointers are not dereferenced
 * except A3 != NULL).  0 = plain fp32-MFMA kernel; otherwise the bf16x3 tile plan TM*1000 + TN*100 + WM*10 + WN
 * (workgroup of WM x WN waves, each TM x TN 32x32 MFMA tiles: 3114 = 96 x 128, 3214 = 96 x 256, 2222 = 128 x 128,
 * ...); 9000 = the persistent resident-weight-panel kernel, 9100 = the persistent B-stationary kernel (gemm_res.hip),
 * 9200 = the persistent C-stationary kernel (gemm_cst.hip).
 * Negative on bad sizes.  Lets tests pin that a benchmarked shape reaches the instantiation tuned for it. */
int pir_gemm_nn_plan(const pir_gemm_nn_t* args);
/* out[part][k/16][m][k%16] (bf16, k padded with zeros to kp = multiple of 16) = part-th piece of the exact
 * split W(m,k) = hi + mid + lo with W(m,k) = W[m*sm + k*sk]; out holds 3*M*kp bf16.  The 16 k-values of one
 * matrix-core k-step are contiguous per row and rows are contiguous per k-step. Used once per weight
 * tensor and step for the forward (sm=K, sk=1) and input-gradient (sm=1, sk=Cin) orientations. */
size_t pir_split_bf16x3_bytes(int M, int K);
int pir_split_bf16x3(const float* W, int M, int K, long sm, long sk, void* out, pir_stream_t stream);

/* Dense 3x3 convolution, stride 1, zero pad 1, no bias, as 9 shifted GEMMs:
 *   Y[b][m][h][w] = sum_{k,dh,dw} Wt[(dh+1)*3+(dw+1)](m,k) * X[b][k][h+dh][w+dw]
 * A(tap,m,k) = A[tap*a_st + m*a_sm + k*a_sk].  With the natural weight layout
 * [M][K][3][3]: a_st=1, a_sm=K*9, a_sk=9.  Input gradient: swap m/k strides and pass
 * flip=1 (tap -> 8-tap).  Replaces OverlapPatchEmbed.proj (net/model.py:206), Downsample /
 * Upsample body[0] (:164,174), PromptGenBlock.conv3x3 (:223), PromptIR.output (:320).
 * R (optional) is added to the result (the global residual `+ inp_img`, :377).
 */
int pir_conv3x3(const float* A, long a_st, long a_sm, long a_sk, int flip,
                const float* X, long x_bs, float* Y, long y_bs,
                const float* R, long r_bs,
                int B, int M, int K, int H, int W, pir_stream_t stream);

/* G[o][i][j] = sum_{r<BR} sum_n X[o,r][i][n] * Y[o,r][j][n]   (contraction over pixels)
 *   operand offset = o1*s1 + o2*s2 + r*sr, rows at stride ld.
 * Split-K over n and r: partial sums go to `ws` and are reduced deterministically by a
 * second kernel (no atomics).  Replaces `q @ k.transpose(-2,-1)` (net/model.py:130) and
 * every weight gradient of a 1x1 conv (dW = dY X^T, summed over batch and pixels) and
 * the dA = dOut V^T product of the attention backward.
 * Optional 3x3 tap shift on the Y operand (dh,dw with image H,W; n = h*W+w) gives the
 * dense-3x3 weight gradient: G written with element stride g_sj / row stride g_si.
 * ws must hold pir_gemm_nt_ws_floats(...) floats.
 */
typedef struct {
  const float* X; long x_s1, x_s2, x_sr; long ldx;
  const float* Y; long y_s1, y_s2, y_sr; long ldy;
  float* G; long g_so, g_si, g_sj;
  int M1, M2, N;
  int O1, O2, BR;
  int shift_dh, shift_dw, H, W; /* H==0: no shift */
  float* ws; size_t ws_floats;
  float alpha; /* result scale */
  int accumulate; /* G += result instead of G = result */
} pir_gemm_nt_t;
size_t pir_gemm_nt_ws_floats(int M1, int M2, int N, int O, int BR);
int pir_gemm_nt(const pir_gemm_nt_t* args, pir_stream_t stream);
/* The same split-K product without its second stage: args->ws[s][o][i][j], s < *splits (host int, written at launch), keeps
 * the partial sums and G is not written.  For the consumers that add the slices themselves in the reduction's order -
 * pir_mdta_softmax_fwd_parts (q k^T, net/model.py:129) and pir_mdta_softmax_bwd_parts (dattn = dout v^T, adjoint of :133)
 * - so that no reduction launch stands between the product and the softmax.  M1 >= M2, no shift. */
int pir_gemm_nt_partials(const pir_gemm_nt_t* args, int* splits, pir_stream_t stream);
/* n <= 4 products in ONE launch: the 1x1 weight gradients of one TransformerBlock (net/model.py:88,92,111,113) at the
 * 32^2 / 16^2 levels, where each alone has too few output tiles for the chip and splits its pixel axis 40 - 160 ways.
 * Every problem is a pir_gemm_nt_t of its own (O1 = O2 = 1, own ws); problems the grouped kernel does not serve run one
 * by one through pir_gemm_nt.  Same results as pir_gemm_nt up to the order of the split-K sum (deterministic). */
int pir_gemm_nt_group(const pir_gemm_nt_t* probs, int n, pir_stream_t stream);
/* Host-only: workspace floats a call really needs (its actual split count; pir_gemm_nt_ws_floats is the worst case over
 * every plan and far larger) - for callers that give every call a workspace piece of its own (deferred reductions). */
size_t pir_gemm_nt_ws_needed(const pir_gemm_nt_t* args);
size_t pir_gemm_nt_group_ws_needed(const pir_gemm_nt_t* probs, int n, int k);

/* bf16x3 matrix-core form of pir_conv3x3: A3 = pir_split_bf16x3_taps() of the weights, layout
 * [3 parts][9 taps][a3_kp/16][M][16] bf16 with W(tap, m, k) = W[(flip ? 8-tap : tap)*st + m*sm + k*sk].
 * Forward: st=1, sm=K*9, sk=9, flip=0 on w[M][K][3][3]; input gradient: st=1, sm=9, sk=M*9, flip=1 with the
 * roles of M and K swapped.  Same results as pir_conv3x3 to fp32 rounding (the six-term product drops <= 2^-27).
 * Round 3: where W is a power of two in [16, 256] and the planes are 16-byte aligned, a tile is 128 / 256 pixels of whole
 * image rows and the activations are loaded once per row shift (conv_rows.hip: the horizontal taps are neighbouring LDS
 * columns); other widths run the nine-pass kernel.  The taps are then summed in the order (dy, k, dx): same results to
 * fp32 rounding. */
int pir_split_bf16x3_taps(const float* W, int M, int K, long st, long sm, long sk, int flip, void* out,
                          pir_stream_t stream);   /* out holds 9 * pir_split_bf16x3_bytes(M, K) bytes */
int pir_conv3x3_x3(const void* A3, int a3_kp, const float* X, long x_bs, float* Y, long y_bs,
                   const float* R, long r_bs, int B, int M, int K, int H, int W, pir_stream_t stream);
/* The same convolution with a scratch buffer of pir_conv3x3_x3_ws_floats(...) floats (0: none needed): a launch that would leave
 * most of the chip idle behind a long stage loop (the up / down-sampling and prompt convolutions at the 16^2 / 32^2 levels,
 * net/model.py:164,174,223: 48 - 96 workgroups walking 72 - 432 stages) is cut into slices of its stages that run side by
 * side; a deterministic second stage adds the partial sums in order.  Same results to fp32 rounding. */
size_t pir_conv3x3_x3_ws_floats(int B, int M, int K, int H, int W);
int pir_conv3x3_x3_ws(const void* A3, int a3_kp, const float* X, long x_bs, float* Y, long y_bs,
                      const float* R, long r_bs, int B, int M, int K, int H, int W,
                      float* ws, size_t ws_floats, pir_stream_t stream);

/* Every pre-split weight of a model refreshed by ONE launch (after an optimiser step): descs (device memory)
 * lists the tensors, blocks (device memory, nblocks x {descriptor index, 4096-element chunk index}) the work.
 * taps=0: as pir_split_bf16x3(W, M, K, sm, sk); taps=1: as pir_split_bf16x3_taps(W, M, K, st, sm, sk, flip). */
typedef struct {
  const float* W; void* out;
  long st, sm, sk;
  int M, K, taps, flip;
} pir_split_desc_t;
int pir_split_bf16x3_batch(const pir_split_desc_t* descs, const int* blocks, int nblocks, pir_stream_t stream);

/* Weight gradient of the dense 3x3 convolutions (pir_conv3x3) in one call:
 *   dw[co][ci][dh+1][dw+1] (+)= sum_{b,h,w} dy[b][co][h][w] * x[b][ci][h+dh][w+dw]   (zero padding)
 * = what autograd derives for OverlapPatchEmbed.proj, Down/Upsample.body[0], PromptGenBlock.conv3x3 and
 * PromptIR.output (net/model.py:206,164,174,223,320).  One bf16x3 matrix-core launch with nine virtual
 * shifted rows per channel when W % 8 == 0 and the planes are 16-byte aligned; nine shifted pir_gemm_nt
 * launches otherwise.  ws holds pir_conv3x3_wgrad_ws_floats() floats. */
size_t pir_conv3x3_wgrad_ws_floats(int Cout, int Cin, int H, int W, int B);
int pir_conv3x3_wgrad(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, int B, int Cout, int Cin,
                      int H, int W, float* ws, size_t ws_floats, int accumulate, pir_stream_t stream);

/* ------------------------------------------------------------------ LayerNorm over channels
 * net/model.py:47-76 (WithBias, :60-63) and :27-41 (BiasFree, :39-41), applied per pixel over
 * the channel axis of an NCHW tensor (to_3d / to_4d, :21-25).  eps = 1e-5 inside the sqrt,
 * biased variance.  bias==NULL selects BiasFree (no mean subtraction in the numerator).
 * mean / rstd ([B][HW]) are saved for the backward.
 */
int pir_layernorm_fwd(const float* x, long x_bs, const float* weight, const float* bias,
                      float* y, long y_bs, float* mean, float* rstd,
                      int B, int C, int HW, pir_stream_t stream);
/* dx (+ dres, the gradient arriving over the residual connection net/model.py:193-194, if not NULL);
 * dweight/dbias partial sums are written to ws ([nblk][2][C]) and reduced into dweight/dbias. */
size_t pir_layernorm_bwd_ws_floats(int B, int C, int HW);
int pir_layernorm_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* weight,
                      int with_bias, const float* mean, const float* rstd,
                      float* dx, long dx_bs, const float* dres, long dres_bs, float* dweight, float* dbias,
                      float* ws, size_t ws_floats, int B, int C, int HW, pir_stream_t stream);

/* ------------------------------------------------------------------ depthwise 3x3 stencil
 * nn.Conv2d(C, C, 3, padding=1, groups=C, bias=False): Attention.qkv_dwconv
 * (net/model.py:112,120) and FeedForward.dwconv (:90,96).  w is [C][3][3].
 * flip=1 correlates with the 180-degree rotated taps (= input gradient).
 */
int pir_dwconv3x3(const float* x, long x_bs, const float* w, int flip, float* y, long y_bs,
                  int B, int C, int H, int W, pir_stream_t stream);
/* The same stencil (flip=0) that also leaves the squared L2 norms of its first nsq output channels, which
 * F.normalize needs for q and k (net/model.py:127-128: nsq = 2*dim of the 3*dim qkv channels), so q and k are not
 * read a second time: sq_parts[b][p][ch] (ch < nsq, p < *nparts) are partial sums of y[b][ch]^2 over the row bands of
 * the plane; the consumer (pir_mdta_softmax_fwd / _bwd) adds the *nparts slices in order.  *nparts (host int, written
 * before returning) depends only on the shape and alignment.  sq_parts holds pir_dwconv3x3_sumsq_floats() floats. */
size_t pir_dwconv3x3_sumsq_floats(int B, int nsq, int H);
int pir_dwconv3x3_sumsq(const float* x, long x_bs, const float* w, float* y, long y_bs, float* sq_parts,
                        size_t sq_floats, int nsq, int* nparts, int B, int C, int H, int W, pir_stream_t stream);
/* Fused GDFN tail: t = dwconv(x) on 2*hid channels; g = gelu_erf(t[:hid]) * t[hid:]
 * (net/model.py:96-97).  Writes g ([B][hid][H][W]). */
int pir_dwconv3x3_gate(const float* x, long x_bs, const float* w, float* g, long g_bs,
                       int B, int hid, int H, int W, pir_stream_t stream);
/* Backward of the fused tail: recomputes t from x, forms dt from dg, returns
 * dx = dwconv^T(dt) is NOT fused here: writes dt ([B][2hid][H][W]) for pir_dwconv3x3(flip=1)
 * and pir_dwconv3x3_wgrad. */
int pir_dwconv3x3_gate_bwd(const float* x, long x_bs, const float* w, const float* dg, long dg_bs,
                           float* dt, long dt_bs, int B, int hid, int H, int W, pir_stream_t stream);
/* dw[c][tap] = sum_{b,h,w} dy[b][c][h][w] * x[b][c][h+dh][w+dw]; partials in ws. */
size_t pir_dwconv3x3_wgrad_ws_floats(int B, int C, int H, int W);
int pir_dwconv3x3_wgrad(const float* dy, long dy_bs, const float* x, long x_bs, float* dw,
                        float* ws, size_t ws_floats, int B, int C, int H, int W, pir_stream_t stream);

/* Fused backward of one depthwise conv: dx = dwconv^T(dy) and dw in a single pass over dy and x. */
size_t pir_dwconv3x3_bwd_ws_floats(int B, int C, int H, int W);
int pir_dwconv3x3_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* w,
                      float* dx, long dx_bs, float* dw, float* ws, size_t ws_floats,
                      int B, int C, int H, int W, pir_stream_t stream);
/* Fused backward of the GDFN tail (net/model.py:96-97): from x = project_in output ([B][2hid]) and dg
 * ([B][hid]) to dx ([B][2hid]) and the depthwise weight gradient dw ([2hid][9]) in ONE pass:
 * the pre-gate tensor t = dwconv(x) and dt are recomputed, never stored.  Power-of-two widths <= 128 run a
 * register-only sliding-window kernel (3-row windows per lane, DPP halos, no LDS); other shapes an LDS-tiled one. */
size_t pir_gdfn_dwconv_bwd_ws_floats(int B, int hid, int H, int W);
int pir_gdfn_dwconv_bwd(const float* x, long x_bs, const float* w, const float* dg, long dg_bs,
                        float* dx, long dx_bs, float* dw, float* ws, size_t ws_floats,
                        int B, int hid, int H, int W, pir_stream_t stream);

/* ------------------------------------------------------------------ MDTA small-matrix stages
 * Row sums of squares over the pixel axis: out[b][c] = sum_n x[b][c][n]^2 (the squared
 * L2 norms F.normalize needs, net/model.py:127-128). */
int pir_row_sumsq(const float* x, long x_bs, float* out, int B, int C, int HW, pir_stream_t stream);
/* attn[b][h] = softmax_j( G[b][h][i][j] / (max(|q_i|,1e-12) max(|k_j|,1e-12)) * temperature[h] )
 * net/model.py:127-131.  sumsq is [B][nparts][2C] (q rows then k rows; nparts partial sums per row, added in
 * order: pir_row_sumsq gives nparts = 1, pir_dwconv3x3_sumsq one slice per row band). */
int pir_mdta_softmax_fwd(const float* gram, const float* sumsq, int nparts, const float* temperature,
                         float* attn, int B, int heads, int c, pir_stream_t stream);
/* ... reading the gram matrix as `splits` split-K partial slices (pir_gemm_nt_partials; [splits][B][h][c][c]), summed in
 * the order of the stand-alone reduction (bit-identical); also writes the gram matrix ([B][h][c][c]) the backward reads */
int pir_mdta_softmax_fwd_parts(const float* gram_parts, int splits, const float* sumsq, int nparts,
                               const float* temperature, float* gram, float* attn, int B, int heads, int c,
                               pir_stream_t stream);
/* Backward through softmax, temperature and the two normalisations.
 * In: dattn, attn, gram, sumsq, temperature.  Out: dgram ([B][h][c][c], already divided by the
 * norms), alpha_q / alpha_k ([B][C]) such that
 *   dq = dgram k + alpha_q * q,  dk = dgram^T q + alpha_k * k,
 * and dtemp_partial[b][h] (summed over b by pir_reduce_partials).  c <= 256. */
int pir_mdta_softmax_bwd(const float* dattn, const float* attn, const float* gram, const float* sumsq,
                         int nparts, const float* temperature, float* dgram,
                         float* alpha_q, float* alpha_k, float* dtemp_partial,
                         int B, int heads, int c, pir_stream_t stream);
/* ... reading dattn as `splits` split-K partial slices (pir_gemm_nt_partials), never materialised */
int pir_mdta_softmax_bwd_parts(const float* dattn_parts, int splits, const float* attn, const float* gram,
                               const float* sumsq, int nparts, const float* temperature, float* dgram,
                               float* alpha_q, float* alpha_k, float* dtemp_partial,
                               int B, int heads, int c, pir_stream_t stream);

/* dq and dk of the MDTA backward from ONE pass over q and k (round 3):
 *   dq[b,h] = dgram[b,h] k[b,h] + alpha_q * q[b,h],   dk[b,h] = dgram[b,h]^T q[b,h] + alpha_k * k[b,h]
 * q of (b, h) starts at q + b*q_bs + h*c*HW (rows at stride HW), k at + k_off; dq / dk likewise at dq + b*dq_bs + h*c*HW
 * (+ dk_off).  Replaces the two pir_gemm_nn calls (each of which reads q AND k) of net/model.py:127-131's backward.
 * Served for c == 48 and HW % 32 == 0 with 16-byte aligned planes; returns 1000 (nothing launched) otherwise and the
 * caller keeps the two-GEMM path. */
int pir_mdta_dqk(const float* dgram, const float* q, long q_bs, long k_off, const float* alpha_q,
                 const float* alpha_k, float* dq, long dq_bs, long dk_off, int B, int heads, int c, int HW,
                 pir_stream_t stream);

/* ------------------------------------------------------------------ pixel (un)shuffle
 * nn.PixelUnshuffle(2) / nn.PixelShuffle(2) (net/model.py:165,175). Each is the other's adjoint.
 * unshuffle: y[b][c*4+i*2+j][h][w] = x[b][c][2h+i][2w+j]   (x is [B][C][2H][2W], y [B][4C][H][W])
 * shuffle  : y[b][c][2h+i][2w+j]   = x[b][c*4+i*2+j][h][w] (x is [B][4C][H][W], y [B][C][2H][2W])
 * C, H, W always describe the LOW-resolution, 4C-channel side. */
int pir_pixel_unshuffle2(const float* x, long x_bs, float* y, long y_bs, int B, int C, int H, int W,
                         pir_stream_t stream);
int pir_pixel_shuffle2(const float* x, long x_bs, float* y, long y_bs, int B, int C, int H, int W,
                       pir_stream_t stream);

/* ------------------------------------------------------------------ GDFN forward without h0 (gdfn_fused.hip)
 * g = gelu_erf(dw3x3(W_in LN(x))[:hid]) * dw3x3(W_in LN(x))[hid:]   (FeedForward behind norm2, net/model.py:94-97,195)
 * in two launches that never write the 2 hid-channel tensor between project_in and the depthwise convolution: the
 * channel LayerNorm + bf16x3 split of x into MFMA fragments (6 bytes per element, into `ws`), then per (image, 32 gate
 * pairs) a walk down the image rows that multiplies, filters (register-resident pending rows), gates and stores.
 * w3: pir_split_bf16x3 pieces of W_in [2 hid][C] (kp = C); wd: depthwise weights [2 hid][9]; mean / rstd: optional
 * [B][HW] outputs.  Served: C = 48 or 96, W = 64 or 128, WithBias LayerNorm, no convolution bias; 1000 otherwise
 * (nothing launched: the caller runs pir_ln_conv1x1_fwd / pir_layernorm_fwd + pir_gemm_nn, then pir_dwconv3x3_gate). */
size_t pir_gdfn_fused_ws_bytes(int B, int C, int H, int W);
int pir_gdfn_fused_fwd(const float* x, long x_bs, const float* ln_w, const float* ln_b, const void* w3, int kp,
                       const float* wd, float* g, long g_bs, void* ws, size_t ws_bytes, float* mean, float* rstd,
                       int B, int C, int hid, int H, int W, pir_stream_t stream);

/* ------------------------------------------------------------------ PromptGenBlock (net/model.py:226-235)
 * emb = mean over pixels (:228) */
int pir_spatial_mean(const float* x, long x_bs, float* out, int B, int C, int HW, pir_stream_t stream);
/* mix = softmax(emb @ Wl^T + bl) over the L prompts (:229). Wl [L][C]. */
int pir_prompt_mix_fwd(const float* emb, const float* Wl, const float* bl, float* mix,
                       int B, int C, int L, pir_stream_t stream);
/* out[b][d][y][x] = bilinear_{align_corners=False}( sum_l mix[b][l] * P[l][d] )(y,x) (:230-232) */
int pir_prompt_resize_fwd(const float* mix, const float* P, float* out, long out_bs,
                          int B, int L, int D, int S, int H, int W, pir_stream_t stream);
/* adjoint: dP[l][d][s][t] = sum_b mix[b][l] * resize^T(dout[b][d])(s,t);
 *          dmix[b][l] = <dout[b], resize(P[l])> .  ws: B*D*S*S floats (+ partials). */
size_t pir_prompt_resize_bwd_ws_floats(int B, int L, int D, int S, int H, int W);
int pir_prompt_resize_bwd(const float* dout, long dout_bs, const float* mix, const float* P,
                          float* dP, float* dmix, float* ws, size_t ws_floats,
                          int B, int L, int D, int S, int H, int W, pir_stream_t stream);
/* backward of softmax+linear+mean: given dmix -> dWl [L][C], dbl [L], and writes (or, with
 * accumulate, adds) demb[b][c]/HW to every pixel of dx[b][c]. */
int pir_prompt_mix_bwd(const float* dmix, const float* mix, const float* emb, const float* Wl,
                       float* dWl, float* dbl, float* dx, long dx_bs, int accumulate,
                       int B, int C, int L, int HW, pir_stream_t stream);

/* ------------------------------------------------------------------ tiled inference (demo.py:17-48, test.py:100-104)
 * Gather nth*ntw overlapping tile_h x tile_w tiles (start_i = min(i*stride, Hp - tile), demo.py:31-33) of an
 * image logically padded at the bottom/right from HxW to HpxWp (pad_mode 0: F.pad 'reflect', demo.py:22;
 * 1: flipped copy, test.py:102-103) into out[B*nth*ntw][C][tile_h][tile_w]. */
int pir_tiles_gather(const float* img, long img_bs, float* out, int B, int C, int H, int W, int Hp, int Wp,
                     int tile_h, int tile_w, int stride_h, int stride_w, int nth, int ntw, int pad_mode,
                     pir_stream_t stream);
/* out[b][c][y][x] = clamp01( sum of covering tiles / count ) for y<Hout, x<Wout (demo.py:37-47 + crop :126) */
int pir_tiles_blend(const float* tiles, float* out, long out_bs, int B, int C, int Hp, int Wp,
                    int tile_h, int tile_w, int stride_h, int stride_w, int nth, int ntw,
                    int Hout, int Wout, int clamp01, pir_stream_t stream);

/* ------------------------------------------------------------------ loss, copies, optimiser
 * nn.L1Loss() (train.py:32,43): loss = lscale * mean|a-b|; optional fused grad = sign(a-b)/count * gscale
 * (grad may be NULL). ws: 1024 floats.  lscale / scale: a part batch's share of the whole batch, passed BY VALUE
 * (the data-parallel trainer cuts a batch into parts on their own streams; with 1 the result is the plain mean). */
int pir_l1_loss(const float* restored, const float* clean, float* loss, float* grad, float gscale, float lscale,
                float* ws, long count, pir_stream_t stream);
/* backward of the same loss with the upstream scalar gradient read from device memory:
 * grad = sign(restored-clean) * dloss[0] * scale / count */
int pir_l1_loss_grad(const float* restored, const float* clean, const float* dloss, float scale, float* grad,
                     long count, pir_stream_t stream);
/* GPU-side Gaussian degradation in the uint8 domain (utils/degradation_utils.py:21-27):
 * out = uint8(clip(floor(clean*255) + N(0,1)*sigma[b], 0, 255)) / 255 with the counter-based generator of
 * promptir_amd/weights.py (keys[2b], keys[2b+1] = the two Box-Muller stream keys of image b). */
int pir_degrade_gaussian(const float* clean, float* out, const float* sigma, const unsigned long long* keys,
                         long per_image, int B, pir_stream_t stream);
/* Training patches cut from whole decoded uint8 HWC images on the device: random crop window + augmentation mode chosen
 * by the host (utils/dataset_utils.py:102-111,140-168, utils/image_utils.py:133-182), applied here as an index map, plus
 * ToTensor (k / 255) and, for unpaired samples, the sigma noise of utils/degradation_utils.py:21-27 in the uint8 domain.
 * meta: int64 [B][8] = {byte offset of the clean image in `images`, byte offset of the paired degraded image or -1,
 * H, W, top, left, mode 0..7, 0}; sigma [B] (ignored for paired samples); keys [2B] as for pir_degrade_gaussian.
 * Outputs: degraded, clean [B][3][P][P]. */
int pir_crop_augment_u8(const unsigned char* images, const long* meta, const float* sigma, const unsigned long long* keys,
                        float* degraded, float* clean, int B, int P, pir_stream_t stream);
/* y[b][c][n] = x[b][c][n] (+ y if accumulate) for channel-slice copies (torch.cat, net/model.py:341-370) */
int pir_copy_planes(const float* x, long x_bs, float* y, long y_bs, int accumulate,
                    int B, long plane_floats, pir_stream_t stream);
/* y (contiguous [B][C][H][W]) = x read through four free element strides: layout repair for a permuted / channels-last
 * view handed to the module surface (the reference's callers pass contiguous NCHW) */
int pir_copy_strided4(const float* x, long s0, long s1, long s2, long s3, float* y, int B, int C, int H, int W,
                      pir_stream_t stream);
/* out[i] = a[i] + b[i] */
int pir_add(const float* a, const float* b, float* out, long count, pir_stream_t stream);

/* Convolution bias, `bias=True` (net/model.py:88-92,111-113,206,294-320; every reference caller passes False):
 *   pir_bias_add   y[b][c][:] += bias[c] in place (after the bias-free convolution kernel)
 *   pir_bias_grad  db[c] = sum_{b,p} dy[b][c][p] (the bias gradient autograd derives for nn.Conv2d)
 * and the GELU gate of FeedForward (:96-97) as a stand-alone pair, for the biased GDFN where the depthwise bias
 * sits between the stencil and the gate (the bias-free path uses the fused pir_dwconv3x3_gate instead):
 *   pir_gelu_gate      g = gelu_erf(t[:, :hid]) * t[:, hid:]
 *   pir_gelu_gate_bwd  dt[:, :hid] = dg * t[:, hid:] * gelu'(t[:, :hid]);  dt[:, hid:] = dg * gelu(t[:, :hid]) */
int pir_bias_add(float* y, long y_bs, const float* bias, int B, int C, int HW, pir_stream_t stream);
int pir_bias_grad(const float* dy, long dy_bs, float* db, int B, int C, int HW, pir_stream_t stream);
int pir_gelu_gate(const float* t, long t_bs, float* g, long g_bs, int B, int hid, int HW, pir_stream_t stream);
int pir_gelu_gate_bwd(const float* t, long t_bs, const float* dg, long dg_bs, float* dt, long dt_bs,
                      int B, int hid, int HW, pir_stream_t stream);
/* out[j] = alpha * sum_{s<S} parts[s*stride + j] (+ out[j] if accumulate) */
int pir_reduce_partials(const float* parts, long stride, int S, float alpha, int accumulate,
                        float* out, long count, pir_stream_t stream);
/* Deferred, batched second stages (reduce_batch.hip).  Every split reduction of the path that feeds a PARAMETER gradient
 * (split-K weight gradients, LayerNorm dweight / dbias, depthwise weight gradients, dtemperature rows) is a ~5 us launch
 * nothing waits for before the optimiser.  pir_reduce_defer(stream, 1): from now on the library QUEUES such a reduction
 * requested on `stream` instead of launching it; pir_reduce_defer(stream, 0) ends the scope (queued work stays queued);
 * pir_reduce_flush(stream) runs everything queued, up to 16 reductions per launch, with results bit-identical to the
 * immediate form.  The caller keeps the partial buffers (the `ws` it passed) alive and untouched until the flush.
 * pir_reduce_pending: number of queued reductions (host-only query). */
int pir_reduce_defer(pir_stream_t stream, int on);
int pir_reduce_flush(pir_stream_t stream);
int pir_reduce_pending(pir_stream_t stream);
/* Partial sets of more than `bytes` bytes (splits x elements x 4) are reduced at once even inside a deferral scope: right
 * behind its producer a reduction reads the partial sums from cache, a deferred one from HBM.  Default 4 MiB; returns the
 * limit in effect (bytes < 0: query only).  A caller that sizes its workspace pieces by the same rule may hand the
 * reused buffer to the large ones. */
long pir_reduce_defer_limit(long bytes);
/* torch.optim.AdamW step (train.py:53: lr 2e-4, betas .9/.999, eps 1e-8, weight_decay 1e-2),
 * over a flat parameter / gradient / moment buffer.  `step` is the 1-based step count. */
int pir_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count,
                   float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                   float grad_scale, pir_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PROMPTIR_HIP_H */
