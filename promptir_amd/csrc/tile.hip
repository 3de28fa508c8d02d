// Tiled inference support (reference demo.py:17-48, test.py:100-104), gfx950: gather overlapping tiles of a
// (logically padded) image into one batch, and blend the restored tiles back (sum / hit count, clamp to
// [0,1], crop).  The model then runs ONCE on all tiles as a batch — numerically identical to the reference's
// sequential loop because no op of the network mixes batch entries.  Pure data movement, HBM-bound.
#include "pir_common.h"

namespace {

// bottom/right padding index: mode 0 = 'reflect' (F.pad, demo.py:22: H+k -> H-2-k), 1 = flipped copy
// (test.py:102-103: H+k -> H-1-k)
__device__ __forceinline__ int pad_index(int i, int n, int mode) {
  if (i < n) return i;
  const int k = i - n;
  int r = mode == 0 ? n - 2 - k : n - 1 - k;
  return r < 0 ? 0 : r;
}
__device__ __forceinline__ int tile_start(int i, int stride, int last) { const int s = i * stride; return s < last ? s : last; }

__global__ __launch_bounds__(256) void tiles_gather_kernel(const float* __restrict__ img, long img_bs, float* __restrict__ out,
                                                           int B, int C, int H, int W, int Hp, int Wp, int th, int tw,
                                                           int sh, int sw, int nth, int ntw, int mode) {
  const long total = (long)B * nth * ntw * C * th * tw;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % tw);
    long t = e / tw;
    const int y = (int)(t % th); t /= th;
    const int c = (int)(t % C); t /= C;
    const int j = (int)(t % ntw); t /= ntw;
    const int i = (int)(t % nth);
    const int b = (int)(t / nth);
    const int hy = pad_index(tile_start(i, sh, Hp - th) + y, H, mode);
    const int wx = pad_index(tile_start(j, sw, Wp - tw) + x, W, mode);
    out[e] = img[b * img_bs + ((long)c * H + hy) * W + wx];
  }
}

// out[b][c][y][x] = clamp( sum_{tiles covering (y,x)} tile value / count, 0, 1 ), summed in the reference's
// tile order (rows outer, columns inner, demo.py:37-44); y < Hout, x < Wout crops the padding away.
__global__ __launch_bounds__(256) void tiles_blend_kernel(const float* __restrict__ tiles, float* __restrict__ out, long out_bs,
                                                          int B, int C, int Hp, int Wp, int th, int tw, int sh, int sw,
                                                          int nth, int ntw, int Hout, int Wout, int clamp01) {
  const long total = (long)B * C * Hout * Wout;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % Wout);
    long t = e / Wout;
    const int y = (int)(t % Hout); t /= Hout;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    float acc = 0.f, cnt = 0.f;
    for (int i = 0; i < nth; ++i) {
      const int hs = tile_start(i, sh, Hp - th);
      if (y < hs || y >= hs + th) continue;
      for (int j = 0; j < ntw; ++j) {
        const int ws = tile_start(j, sw, Wp - tw);
        if (x < ws || x >= ws + tw) continue;
        acc += tiles[((((long)b * nth + i) * ntw + j) * C + c) * th * tw + (long)(y - hs) * tw + (x - ws)];
        cnt += 1.f;
      }
    }
    float v = acc / cnt;
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    out[b * out_bs + ((long)c * Hout + y) * Wout + x] = v;
  }
}

inline int grid_for(long total) { long g = pir_cdiv(total, 256); return (int)(g < 8192 ? (g < 1 ? 1 : g) : 8192); }

}  // namespace

extern "C" int pir_tiles_gather(const float* img, long img_bs, float* out, int B, int C, int H, int W, int Hp, int Wp,
                                int tile_h, int tile_w, int stride_h, int stride_w, int nth, int ntw, int pad_mode,
                                pir_stream_t stream) {
  PIR_CHECK_ARG(img && out && B > 0 && C > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W);
  PIR_CHECK_ARG(tile_h > 0 && tile_w > 0 && tile_h <= Hp && tile_w <= Wp && stride_h > 0 && stride_w > 0 && nth > 0 && ntw > 0);
  PIR_CHECK_ARG(pad_mode == 0 || pad_mode == 1);
  // reflect (H + k -> H - 2 - k) reaches row 0 at k = H - 2; the flipped copy of test.py:102-103 (H + k -> H - 1 - k) at k = H - 1:
  // an image of exactly 64 rows is extended to 128 by its whole mirror image
  PIR_CHECK_ARG(Hp - H <= H - 1 + pad_mode && Wp - W <= W - 1 + pad_mode);
  const long total = (long)B * nth * ntw * C * tile_h * tile_w;
  hipLaunchKernelGGL(tiles_gather_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, img_bs, out,
                     B, C, H, W, Hp, Wp, tile_h, tile_w, stride_h, stride_w, nth, ntw, pad_mode);
  return pir_launch_status();
}

extern "C" int pir_tiles_blend(const float* tiles, float* out, long out_bs, int B, int C, int Hp, int Wp,
                               int tile_h, int tile_w, int stride_h, int stride_w, int nth, int ntw,
                               int Hout, int Wout, int clamp01, pir_stream_t stream) {
  PIR_CHECK_ARG(tiles && out && B > 0 && C > 0 && Hout > 0 && Wout > 0 && Hout <= Hp && Wout <= Wp);
  PIR_CHECK_ARG(tile_h > 0 && tile_w > 0 && tile_h <= Hp && tile_w <= Wp && stride_h > 0 && stride_w > 0 && nth > 0 && ntw > 0);
  const long total = (long)B * C * Hout * Wout;
  hipLaunchKernelGGL(tiles_blend_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tiles, out, out_bs,
                     B, C, Hp, Wp, tile_h, tile_w, stride_h, stride_w, nth, ntw, Hout, Wout, clamp01);
  return pir_launch_status();
}
