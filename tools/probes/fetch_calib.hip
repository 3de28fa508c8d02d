// Probe: what rocprofv3's FETCH_SIZE reports per byte actually streamed from HBM, by access width and row pattern
// (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc -O3 --offload-arch=gfx950 tools/probes/fetch_calib.hip -o tools/probes/fetch_calib.bin
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/probes/fetch_calib.bin
// then tools/fetch_calib_summary.py out.  Every kernel reads the same 1 GiB buffer exactly once (4x the Infinity
// Cache, so nothing is served on-die) and each pattern runs once; the kernel name carries the pattern.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr long N_FLOATS = 1L << 28;   // 1 GiB

// contiguous stream: consecutive lanes read consecutive VEC-float pieces
template <int VEC>
__global__ __launch_bounds__(256) void stream_read(const float* __restrict__ x, long n, float* sink) {
  float acc = 0.f;
  const long pieces = n / VEC;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < pieces; i += (long)gridDim.x * 256) {
    if (VEC == 1) acc += x[i];
    else if (VEC == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(x + 2 * i); acc += v[0] + v[1]; }
    else { const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i); acc += v[0] + v[1] + v[2] + v[3]; }
  }
  if (acc == 123.456f) sink[0] = acc;
}
// row pattern of the split-K weight-gradient stage loads (gemm.hip QUAD): a matrix [rows][ld]; per step 16 consecutive
// floats (64 B) of every row are read, LPR lanes x (64 / LPR) bytes per row; a wave covers 64 / LPR rows per instruction.
template <int LPR>
__global__ __launch_bounds__(256) void rows64_read(const float* __restrict__ x, int rows, int ld, float* sink) {
  constexpr int FPL = 16 / LPR;                 // floats per lane
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  const int rpw = 64 / LPR;                     // rows per wave-instruction
  float acc = 0.f;
  const long steps = ld / 16;
  // a wave owns a block of rpw rows and walks along them (the kernels' k loop)
  for (long rb = wave; rb < rows / rpw; rb += nwaves) {
    const float* row = x + ((long)rb * rpw + lane / LPR) * ld + (lane % LPR) * FPL;
    for (long s = 0; s < steps; ++s) {
      if (FPL == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(row + 16 * s); acc += v[0] + v[1] + v[2] + v[3]; }
      else if (FPL == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(row + 16 * s); acc += v[0] + v[1]; }
      else acc += row[16 * s];
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

// a lane reads 32 B (two 16-byte loads) of a row, 2 lanes per 64-B row piece: the non-QUAD fragment loads
__global__ __launch_bounds__(256) void rows64_read_2x16(const float* __restrict__ x, int rows, int ld, float* sink) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  float acc = 0.f;
  const long steps = ld / 16;
  for (long rb = wave; rb < rows / 32; rb += nwaves) {
    const float* row = x + ((long)rb * 32 + lane / 2) * ld + (lane % 2) * 8;
    for (long s = 0; s < steps; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(row + 16 * s), b = *reinterpret_cast<const f32x4*>(row + 16 * s + 4);
      acc += a[0] + a[1] + a[2] + a[3] + b[0] + b[1] + b[2] + b[3];
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int VEC>
__global__ __launch_bounds__(256) void stream_write(float* __restrict__ x, long n) {
  const long pieces = n / VEC;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < pieces; i += (long)gridDim.x * 256) {
    if (VEC == 1) x[i] = 1.f;
    else { const f32x4 v = {1.f, 2.f, 3.f, 4.f}; *reinterpret_cast<f32x4*>(x + 4 * i) = v; }
  }
}

int main() {
  float *x, *sink;
  if (hipMalloc(&x, N_FLOATS * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 2; }
  hipLaunchKernelGGL((stream_write<4>), dim3(4096), dim3(256), 0, 0, x, N_FLOATS);
  hipLaunchKernelGGL((stream_write<1>), dim3(4096), dim3(256), 0, 0, x, N_FLOATS);
  hipLaunchKernelGGL((stream_read<1>), dim3(4096), dim3(256), 0, 0, x, N_FLOATS, sink);
  hipLaunchKernelGGL((stream_read<2>), dim3(4096), dim3(256), 0, 0, x, N_FLOATS, sink);
  hipLaunchKernelGGL((stream_read<4>), dim3(4096), dim3(256), 0, 0, x, N_FLOATS, sink);
  const int ld = 16384, rows = (int)(N_FLOATS / ld);      // 16384 rows of 64 KB: an activation matrix [C*B][HW]
  hipLaunchKernelGGL((rows64_read<4>), dim3(1024), dim3(256), 0, 0, x, rows, ld, sink);   // 4 lanes x 16 B per row (QUAD)
  hipLaunchKernelGGL((rows64_read<8>), dim3(1024), dim3(256), 0, 0, x, rows, ld, sink);   // 8 lanes x 8 B per row
  hipLaunchKernelGGL((rows64_read<16>), dim3(1024), dim3(256), 0, 0, x, rows, ld, sink);  // 16 lanes x 4 B per row
  hipLaunchKernelGGL(rows64_read_2x16, dim3(1024), dim3(256), 0, 0, x, rows, ld, sink);   // 2 lanes x 2 x 16 B per row
  if (hipDeviceSynchronize() != hipSuccess) { printf("run failed\n"); return 1; }
  printf("{\"probe\": \"fetch_calib\", \"bytes_per_kernel\": %ld}\n", N_FLOATS * 4);
  return 0;
}
