// Probe: HBM write / read bandwidth of the access patterns a GEMM epilogue and its B-operand loads produce.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
// Matrix Y[B][M][N] fp32 (N = 16384 pixels per row = 64 KB).  Each wave owns a [ROWS x COLS] block and writes
// (or reads) it row by row, COLS*4 bytes contiguous per row; successive rows are 64 KB apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int COLS, bool READ>
__global__ __launch_bounds__(256) void blk_kernel(float* __restrict__ y, int M, int N, int rows, long total_blocks, float* sink) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= total_blocks) return;
  const int nb = N / COLS, mb = (M + rows - 1) / rows;
  const long img = wave / ((long)nb * mb);
  const long rem = wave % ((long)nb * mb);
  const int m0 = (int)(rem % mb) * rows, n0 = (int)(rem / mb) * COLS;
  float* base = y + (img * M + m0) * (long)N + n0;
  constexpr int LPR = COLS / 4;            // lanes per row (16 B per lane)
  constexpr int RPI = 64 / LPR;            // rows per instruction
  const int rr = lane / LPR, cc = (lane % LPR) * 4;
  float acc = 0.f;
  for (int r = rr; r < rows && m0 + r < M; r += RPI) {
    float4* p = reinterpret_cast<float4*>(base + (long)r * N + cc);
    if (READ) { float4 v = *p; acc += v.x + v.y + v.z + v.w; }
    else *p = make_float4(1.f, 2.f, 3.f, (float)r);
  }
  if (READ && acc == 123.456f) sink[0] = acc;
}

// the gemm epilogue's pattern: a wave owns [32*TM rows x 32 cols]; one 4-byte store per lane and instruction,
// lane & 31 = column, lane >> 5 selects row +4 (v_mfma_f32_32x32 C layout): two 128-byte row segments per instruction
template <int TM, bool WIDE>
__global__ __launch_bounds__(256) void mfma_store_kernel(float* __restrict__ y, int M, int N, long total_blocks) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= total_blocks) return;
  const int rows = 32 * TM, nb = N / 32, mb = (M + rows - 1) / rows;
  const long img = wave / ((long)nb * mb);
  const long rem = wave % ((long)nb * mb);
  const int m0 = (int)(rem % mb) * rows, n0 = (int)(rem / mb) * 32;
  float* base = y + (img * M + m0) * (long)N + n0;
  if (!WIDE) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m0 + m < M) base[(long)m * N + (lane & 31)] = (float)r;
      }
  } else {   // the same tile, 16 bytes per lane: 8 lanes per row, 8 rows per instruction
#pragma unroll
    for (int r = 0; r < 4 * TM; ++r) {
      const int m = r * 8 + (lane >> 3);
      if (m0 + m < M) *reinterpret_cast<float4*>(base + (long)m * N + (lane & 7) * 4) = make_float4(1.f, 2.f, 3.f, (float)r);
    }
  }
}

template <int TM, bool WIDE>
float run_mfma(float* y, int B, int M, int N) {
  const long blocks = (long)B * (N / 32) * ((M + 32 * TM - 1) / (32 * TM));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mfma_store_kernel<TM, WIDE>), dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, 0, y, M, N, blocks);
  hipEventRecord(a);
  const int it = 5;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL((mfma_store_kernel<TM, WIDE>), dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, 0, y, M, N, blocks);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / it;
}

template <int COLS, bool READ>
float run(float* y, int B, int M, int N, int rows, float* sink) {
  const long blocks = (long)B * (N / COLS) * ((M + rows - 1) / rows);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((blk_kernel<COLS, READ>), dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, 0, y, M, N, rows, blocks, sink);
  hipEventRecord(a);
  const int it = 5;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL((blk_kernel<COLS, READ>), dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, 0, y, M, N, rows, blocks, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / it;
}

int main() {
  const int B = 32, M = 510, N = 16384;
  float *y, *sink;
  hipMalloc(&y, (size_t)B * M * N * 4); hipMalloc(&sink, 4);
  hipMemset(y, 0, (size_t)B * M * N * 4);
  const double gb = (double)B * M * N * 4 / 1e9;
  struct { const char* name; int rows; } cases[] = {{"96 rows", 96}, {"32 rows", 32}, {"510 rows (whole column block)", 510}, {"8 rows", 8}, {"1 row", 1}};
  for (auto& c : cases) {
    printf("%-32s write: 32 cols %6.0f | 64 cols %6.0f | 128 cols %6.0f | 256 cols %6.0f GB/s   read: 32 cols %6.0f | 128 cols %6.0f | 256 cols %6.0f GB/s\n", c.name,
           gb / run<32, false>(y, B, M, N, c.rows, sink) * 1e3, gb / run<64, false>(y, B, M, N, c.rows, sink) * 1e3,
           gb / run<128, false>(y, B, M, N, c.rows, sink) * 1e3, gb / run<256, false>(y, B, M, N, c.rows, sink) * 1e3,
           gb / run<32, true>(y, B, M, N, c.rows, sink) * 1e3, gb / run<128, true>(y, B, M, N, c.rows, sink) * 1e3,
           gb / run<256, true>(y, B, M, N, c.rows, sink) * 1e3);
  }
  printf("MFMA C-layout stores (4 B per lane, 2 x 128 B per instruction): 96x32 tile %6.0f GB/s, 32x32 tile %6.0f GB/s; same tiles with 16 B per lane: %6.0f / %6.0f GB/s\n",
         gb / run_mfma<3, false>(y, B, M, N) * 1e3, gb / run_mfma<1, false>(y, B, M, N) * 1e3,
         gb / run_mfma<3, true>(y, B, M, N) * 1e3, gb / run_mfma<1, true>(y, B, M, N) * 1e3);
  return 0;
}
