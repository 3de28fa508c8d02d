// Second stages of every split reduction of the path - split-K weight gradients (gemm.hip, gemm_ntx.hip), LayerNorm
// dgamma / dbeta, depthwise weight gradients, dtemperature partial rows - and their DEFERRED, BATCHED form (gfx950).
//
// A train step launched ~1070 of these 5 us kernels (30 % of its launches, 5.5 % of its kernel time inside the
// two-stream graph: profiles/r04_step_launches_start.json, r03_bench_b32_graph_kernel_stats.csv).  Every one of them feeds
// a PARAMETER gradient, which nothing reads before the optimiser: inside a `pir_reduce_defer(stream, 1)` scope the
// library queues the reduction (a 120-byte descriptor) instead of launching it, and `pir_reduce_flush(stream)` runs up
// to 16 queued reductions as ONE launch whose workgroups look their descriptor up in the kernel arguments.  The caller
// keeps the partial buffers alive and untouched until the flush (promptir_amd/ops.py: bump-allocated workspace arena).
// The batched kernel executes the very same device functions, with the same grouping of the splits, as the stand-alone
// kernels: results are bit-identical whether a reduction is deferred or not (tests/test_kernels_gpu.py).
#include "pir_common.h"
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

enum { RB_PLAIN4 = 0, RB_PLAIN16 = 1, RB_NARROW = 2, RB_NT4 = 3, RB_NT16 = 4 };

struct ReduceDesc {
  const float* parts; long stride; long count;   // out[e] = alpha * sum_s parts[s * stride + e], e < count
  float* out; float* out2; long split;           // plain: columns >= split go to out2 (if given)
  long g_so, g_si, g_sj, g_st;                   // nt: e -> (o, i, j) -> out + o g_so + i g_si + j g_sj (g_st: (channel, tap) columns)
  int S, M1, M2, kind, accumulate; float alpha;
};

__device__ __forceinline__ void store_plain(const ReduceDesc& d, long j, float s) {
  s *= d.alpha;
  float* dst = (d.out2 && j >= d.split) ? d.out2 + (j - d.split) : d.out + j;
  *dst = d.accumulate ? *dst + s : s;
}

__device__ __forceinline__ void store_nt(const ReduceDesc& d, long e, float s) {
  const long mm = (long)d.M1 * d.M2;
  const long o = e / mm, ij = e % mm;
  const long i = ij / d.M2, j = ij % d.M2;
  // g_st != 0: j enumerates (channel, tap) pairs, channel stride g_sj and tap stride g_st
  float* dst = d.g_st ? d.out + o * d.g_so + i * d.g_si + (j / 9) * d.g_sj + (j % 9) * d.g_st : d.out + o * d.g_so + i * d.g_si + j * d.g_sj;
  const float v = d.alpha * s;
  *dst = d.accumulate ? *dst + v : v;
}

// 64 consecutive elements x GR split groups per workgroup; every thread sums its group's splits with four independent
// accumulators (loads in flight), the groups are combined through LDS in a fixed order (deterministic).  Threads beyond
// 64 * GR (the batched kernel always launches 1024) only take part in the barrier.
template <int GR, bool NT>
__device__ __forceinline__ void reduce_wide_body(const ReduceDesc& d, long block, int tid, float* red /* [GR][64] */) {
  const int lane = tid & 63, grp = tid >> 6;
  const long e = block * 64L + lane;
  const bool act = grp < GR;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (act && e < d.count) {
    const float* __restrict__ p = d.parts;
    int k = grp;
    for (; k + 3 * GR < d.S; k += 4 * GR) {
      s0 += p[(long)k * d.stride + e];
      s1 += p[(long)(k + GR) * d.stride + e];
      s2 += p[(long)(k + 2 * GR) * d.stride + e];
      s3 += p[(long)(k + 3 * GR) * d.stride + e];
    }
    for (; k < d.S; k += GR) s0 += p[(long)k * d.stride + e];
  }
  if (act) red[grp * 64 + lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && e < d.count) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < GR; ++q) s += red[q * 64 + lane];
    if (NT) store_nt(d, e, s); else store_plain(d, e, s);
  }
}

// Few outputs, many rows (LayerNorm dweight / dbias, depthwise weight gradients: 48..1872 outputs from up to 2048 partial
// rows): 16 outputs x 64 row groups per workgroup, so that each thread walks S / 64 rows instead of S / 16 and 4x as many
// workgroups share the work.  Fixed summation order.
__device__ __forceinline__ void reduce_narrow_body(const ReduceDesc& d, long block, int tid, float* red /* [64][17] */) {
  const int col = tid & 15, grp = tid >> 4;
  const long j = block * 16L + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < d.count) {
    const float* __restrict__ p = d.parts;
    int k = grp;
    for (; k + 192 < d.S; k += 256) {
      s0 += p[(long)k * d.stride + j];
      s1 += p[(long)(k + 64) * d.stride + j];
      s2 += p[(long)(k + 128) * d.stride + j];
      s3 += p[(long)(k + 192) * d.stride + j];
    }
    for (; k < d.S; k += 64) s0 += p[(long)k * d.stride + j];
  }
  red[grp * 17 + col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && j < d.count) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) s += red[q * 17 + col];
    store_plain(d, j, s);
  }
}

__device__ __forceinline__ void reduce_dispatch(const ReduceDesc& d, long block, int tid, float* red) {
  switch (d.kind) {
    case RB_PLAIN4: reduce_wide_body<4, false>(d, block, tid, red); break;
    case RB_PLAIN16: reduce_wide_body<16, false>(d, block, tid, red); break;
    case RB_NARROW: reduce_narrow_body(d, block, tid, red); break;
    case RB_NT4: reduce_wide_body<4, true>(d, block, tid, red); break;
    default: reduce_wide_body<16, true>(d, block, tid, red); break;
  }
}

// stand-alone form: one reduction per launch, block size as small as its kind allows
template <int KIND>
__global__ __launch_bounds__((KIND == RB_PLAIN4 || KIND == RB_NT4) ? 256 : 1024) void reduce_one_kernel(ReduceDesc d) {
  __shared__ float red[64 * 17];
  if (KIND == RB_PLAIN4) reduce_wide_body<4, false>(d, blockIdx.x, threadIdx.x, red);
  else if (KIND == RB_PLAIN16) reduce_wide_body<16, false>(d, blockIdx.x, threadIdx.x, red);
  else if (KIND == RB_NARROW) reduce_narrow_body(d, blockIdx.x, threadIdx.x, red);
  else if (KIND == RB_NT4) reduce_wide_body<4, true>(d, blockIdx.x, threadIdx.x, red);
  else reduce_wide_body<16, true>(d, blockIdx.x, threadIdx.x, red);
}

constexpr int RB_MAX = 16;
struct ReduceBatch {
  ReduceDesc d[RB_MAX];
  int first[RB_MAX + 1];   // first workgroup of every descriptor, first[n] = grid size
  int n;
};

// batched form: the workgroup finds its descriptor among the kernel arguments
__global__ __launch_bounds__(1024) void reduce_batch_kernel(ReduceBatch b) {
  __shared__ float red[64 * 17];
  int k = 0;
  while (k + 1 < b.n && (int)blockIdx.x >= b.first[k + 1]) ++k;
  reduce_dispatch(b.d[k], (long)blockIdx.x - b.first[k], threadIdx.x, red);
}

inline long blocks_of(const ReduceDesc& d) { return pir_cdiv(d.count, d.kind == RB_NARROW ? 16 : 64); }

int launch_one(const ReduceDesc& d, hipStream_t s) {
  const dim3 grid((unsigned)blocks_of(d));
  switch (d.kind) {
    case RB_PLAIN4: hipLaunchKernelGGL((reduce_one_kernel<RB_PLAIN4>), grid, dim3(256), 0, s, d); break;
    case RB_PLAIN16: hipLaunchKernelGGL((reduce_one_kernel<RB_PLAIN16>), grid, dim3(1024), 0, s, d); break;
    case RB_NARROW: hipLaunchKernelGGL((reduce_one_kernel<RB_NARROW>), grid, dim3(1024), 0, s, d); break;
    case RB_NT4: hipLaunchKernelGGL((reduce_one_kernel<RB_NT4>), grid, dim3(256), 0, s, d); break;
    default: hipLaunchKernelGGL((reduce_one_kernel<RB_NT16>), grid, dim3(1024), 0, s, d); break;
  }
  return pir_launch_status();
}

struct StreamQueue { bool defer = false; std::vector<ReduceDesc> q; };
std::mutex g_mu;
// Only SMALL partial sets are worth deferring: a reduction launched right behind its producer reads the partial sums from
// cache, a deferred one reads them from HBM after a block's worth of other traffic.  Measured (round 4, same box): with
// every split-K reduction deferred (~2 GB of partial tiles per part batch parked until the flush) the batch-32 step lost
// 1 ms and the batch-8 step 2.7 ms; LayerNorm / depthwise / temperature partials are a few hundred KB.
long g_defer_limit_bytes = 4L << 20;
std::unordered_map<hipStream_t, StreamQueue> g_queues;

int flush_locked(StreamQueue& sq, hipStream_t s) {
  size_t i = 0;
  while (i < sq.q.size()) {
    ReduceBatch b;
    b.n = 0;
    long blocks = 0;
    while (i < sq.q.size() && b.n < RB_MAX) {
      b.d[b.n] = sq.q[i];
      b.first[b.n] = (int)blocks;
      blocks += blocks_of(sq.q[i]);
      ++b.n; ++i;
    }
    b.first[b.n] = (int)blocks;
    if (b.n == 1) {
      const int st = launch_one(b.d[0], s);
      if (st) { sq.q.clear(); return st; }
      continue;
    }
    for (int k = b.n + 1; k <= RB_MAX; ++k) b.first[k] = (int)blocks;
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, b);
    const int st = pir_launch_status();
    if (st) { sq.q.clear(); return st; }
  }
  sq.q.clear();
  return PIR_OK;
}

// launch now, or queue when the stream is inside a deferral scope.  Two queued reductions must not write the same
// destination (the batched launch runs them concurrently): a descriptor that would is preceded by a flush.
int submit(const ReduceDesc& d, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_queues.find(s);
  if (it == g_queues.end() || !it->second.defer) return launch_one(d, s);
  if ((long)d.S * d.stride * 4 > g_defer_limit_bytes) return launch_one(d, s);
  StreamQueue& sq = it->second;
  for (const ReduceDesc& q : sq.q)
    if (q.out == d.out || (d.out2 && (q.out == d.out2 || q.out2 == d.out2)) || (q.out2 && q.out2 == d.out)) {
      const int st = flush_locked(sq, s);
      if (st) return st;
      break;
    }
  sq.q.push_back(d);
  return PIR_OK;
}

}  // namespace

// columns [0, split) of the partial rows go to `out`, columns [split, count) to `out2` (one launch for LayerNorm's
// dweight and dbias, whose partial rows are [2][C]); out2 == nullptr: everything to `out`
int pir_reduce_partials_to2(const float* parts, long stride, int S, float alpha, int accumulate, float* out, float* out2,
                            long split, long count, pir_stream_t stream) {
  PIR_CHECK_ARG(parts && out && S > 0 && count > 0);
  ReduceDesc d{};
  d.parts = parts; d.stride = stride; d.count = count; d.out = out; d.out2 = out2; d.split = split;
  d.S = S; d.alpha = alpha; d.accumulate = accumulate;
  d.kind = (S >= 256 && count <= 4096) ? RB_NARROW : (S >= 64 ? RB_PLAIN16 : RB_PLAIN4);
  return submit(d, (hipStream_t)stream);
}

// never queued: the next kernel on the stream reads `out` (split dense convolutions, gemm_x3.hip)
int pir_reduce_partials_now(const float* parts, long stride, int S, float alpha, int accumulate, float* out, long count,
                            pir_stream_t stream) {
  PIR_CHECK_ARG(parts && out && S > 0 && count > 0);
  ReduceDesc d{};
  d.parts = parts; d.stride = stride; d.count = count; d.out = out; d.out2 = nullptr; d.split = 0;
  d.S = S; d.alpha = alpha; d.accumulate = accumulate;
  d.kind = (S >= 256 && count <= 4096) ? RB_NARROW : (S >= 64 ? RB_PLAIN16 : RB_PLAIN4);
  return launch_one(d, (hipStream_t)stream);
}

extern "C" int pir_reduce_partials(const float* parts, long stride, int S, float alpha, int accumulate,
                                   float* out, long count, pir_stream_t stream) {
  return pir_reduce_partials_to2(parts, stride, S, alpha, accumulate, out, nullptr, 0, count, stream);
}

// second stage of a split-K product: G[o][i][j] (strided) = alpha * sum_s ws[s][o][i][j] (+ G)
int pir_nt_reduce_submit(const float* ws, int splits, long O, int M1, int M2, float* G, long g_so, long g_si, long g_sj, long g_st,
                         float alpha, int accumulate, hipStream_t s) {
  ReduceDesc d{};
  d.parts = ws; d.stride = O * M1 * M2; d.count = d.stride; d.out = G;
  d.g_so = g_so; d.g_si = g_si; d.g_sj = g_sj; d.g_st = g_st;
  d.S = splits; d.M1 = M1; d.M2 = M2; d.alpha = alpha; d.accumulate = accumulate;
  d.kind = splits >= 64 ? RB_NT16 : RB_NT4;
  return submit(d, s);
}

extern "C" int pir_reduce_defer(pir_stream_t stream, int on) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_queues[(hipStream_t)stream].defer = on != 0;
  return PIR_OK;
}

// partial sets larger than this many bytes are reduced at once even inside a deferral scope (default 4 MiB); returns the
// limit in effect (bytes < 0: query only)
extern "C" long pir_reduce_defer_limit(long bytes) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (bytes >= 0) g_defer_limit_bytes = bytes;
  return g_defer_limit_bytes;
}

extern "C" int pir_reduce_pending(pir_stream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_queues.find((hipStream_t)stream);
  return it == g_queues.end() ? 0 : (int)it->second.q.size();
}

extern "C" int pir_reduce_flush(pir_stream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_queues.find((hipStream_t)stream);
  if (it == g_queues.end() || it->second.q.empty()) return PIR_OK;
  return flush_locked(it->second, (hipStream_t)stream);
}
