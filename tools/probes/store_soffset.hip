// Probe: what a SCALAR offset operand does on gfx950 raw-buffer STORES (and loads).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_soffset.hip -o tools/probes/store_soffset.bin && tools/probes/store_soffset.bin
//
// Background: the persistent GEMM kernels (gemm_res.hip, gemm_cst.hip) store 32 x 32 output blocks
// with raw_buffer_store_b128 through a descriptor whose num_records ends at the last valid row, so that rows beyond M
// are dropped by the hardware range check.  Putting the wave-uniform row term of the address into the instruction's
// scalar-offset operand (as the loads do) produced WRONG results; the stores therefore add the row term to the per-lane
// offset with one v_add (wide_tiles.h: pir_row_offset).  This probe pins both halves of that finding:
//   (A) voffset = lane part + row part (what pir_row_offset produces): in-range stores land, out-of-range stores are
//       dropped - the behaviour the library relies on;
//   (B) voffset = lane part, soffset = row part: reports whether stores whose TOTAL offset lies beyond num_records are
//       still written (the scalar offset is not part of the range check) and whether in-range stores land where (A) puts them.
// Output: one JSON line.  tests/test_probes_gpu.py asserts (A) and records (B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}

// the library's own idiom, copied in spirit from wide_tiles.h (kept in step by tests/test_probes_gpu.py, which greps it)
__device__ __forceinline__ int row_offset(int lane_off, int row_off) {
  int r;
  asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "s"(row_off), "v"(lane_off));
  return r;
}

// every wave stores ROWS rows of 64 x 16 bytes; row r starts at byte r * pitch.  `records` bytes are visible through the
// descriptor; the allocation behind it is larger, so that a store that escapes the range check shows up as data, not a fault.
template <bool SOFF>
__global__ __launch_bounds__(64) void store_kernel(float* buf, unsigned records, int rows, int pitch_bytes) {
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(buf, records);
  for (int r = 0; r < rows; ++r) {
    const int row_off = __builtin_amdgcn_readfirstlane(r * pitch_bytes);
    const float v = (float)(r * 64 + lane + 1);
    const u32x4 data = __builtin_bit_cast(u32x4, (float __attribute__((ext_vector_type(4)))){v, v + 0.25f, v + 0.5f, v + 0.75f});
    if (SOFF) __builtin_amdgcn_raw_buffer_store_b128(data, rs, lane * 16, row_off, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(data, rs, row_offset(lane * 16, row_off), 0, 0);
  }
}

template <bool SOFF>
__global__ __launch_bounds__(64) void load_kernel(const float* buf, unsigned records, int rows, int pitch_bytes, float* out) {
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(const_cast<float*>(buf), records);
  for (int r = 0; r < rows; ++r) {
    const int row_off = __builtin_amdgcn_readfirstlane(r * pitch_bytes);
    u32x4 d;
    if (SOFF) d = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, row_off, 0);
    else d = __builtin_amdgcn_raw_buffer_load_b128(rs, row_offset(lane * 16, row_off), 0, 0);
    *reinterpret_cast<u32x4*>(out + ((long)r * 64 + lane) * 4) = d;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s\"}\n", hipGetErrorString(e)); return 2; } } while (0)

int main() {
  const int rows = 8, pitch = 64 * 16;               // 8 rows of 1 KB
  const int valid_rows = 5;                          // the descriptor ends behind row 4
  const unsigned records = valid_rows * pitch;
  const size_t floats = (size_t)rows * pitch / 4;
  float* d; float* dout;
  CK(hipMalloc(&d, floats * 4)); CK(hipMalloc(&dout, floats * 4));
  std::vector<float> h(floats), ha(floats), hb(floats);
  const float SENT = -7.f;
  auto fill = [&]() { for (auto& v : h) v = SENT; return hipMemcpy(d, h.data(), floats * 4, hipMemcpyHostToDevice); };
  // (A) per-lane sum
  CK(fill());
  hipLaunchKernelGGL((store_kernel<false>), dim3(1), dim3(64), 0, 0, d, records, rows, pitch);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ha.data(), d, floats * 4, hipMemcpyDeviceToHost));
  // (B) scalar offset operand
  CK(fill());
  hipLaunchKernelGGL((store_kernel<true>), dim3(1), dim3(64), 0, 0, d, records, rows, pitch);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(hb.data(), d, floats * 4, hipMemcpyDeviceToHost));
  auto expect = [&](size_t i) { const size_t lane4 = i % 256, r = i / 256; return (float)(r * 64 + lane4 / 4 + 1) + 0.25f * (lane4 % 4); };
  int a_in_ok = 1, a_clip_ok = 1, b_in_ok = 1, b_escaped = 0;
  for (size_t i = 0; i < floats; ++i) {
    const bool inside = i * 4 < records;
    if (inside) { if (ha[i] != expect(i)) a_in_ok = 0; if (hb[i] != expect(i)) b_in_ok = 0; }
    else { if (ha[i] != SENT) a_clip_ok = 0; if (hb[i] != SENT) ++b_escaped; }
  }
  // loads: data = index pattern; out-of-range loads return 0
  for (size_t i = 0; i < floats; ++i) h[i] = (float)i;
  CK(hipMemcpy(d, h.data(), floats * 4, hipMemcpyHostToDevice));
  int la_ok = 1, lb_in_ok = 1, lb_escaped = 0;
  hipLaunchKernelGGL((load_kernel<false>), dim3(1), dim3(64), 0, 0, d, records, rows, pitch, dout);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ha.data(), dout, floats * 4, hipMemcpyDeviceToHost));
  hipLaunchKernelGGL((load_kernel<true>), dim3(1), dim3(64), 0, 0, d, records, rows, pitch, dout);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(hb.data(), dout, floats * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < floats; ++i) {
    const bool inside = i * 4 < records;
    if (ha[i] != (inside ? (float)i : 0.f)) la_ok = 0;
    if (inside) { if (hb[i] != (float)i) lb_in_ok = 0; } else if (hb[i] != 0.f) ++lb_escaped;
  }
  printf("{\"probe\": \"store_soffset\", \"vgpr_sum_store_in_range_ok\": %d, \"vgpr_sum_store_clipped_ok\": %d, "
         "\"soffset_store_in_range_ok\": %d, \"soffset_store_floats_written_beyond_records\": %d, "
         "\"vgpr_sum_load_ok\": %d, \"soffset_load_in_range_ok\": %d, \"soffset_load_floats_read_beyond_records\": %d, "
         "\"floats_beyond_records\": %zu}\n",
         a_in_ok, a_clip_ok, b_in_ok, b_escaped, la_ok, lb_in_ok, lb_escaped, floats - records / 4);
  hipFree(d); hipFree(dout);
  return (a_in_ok && a_clip_ok && la_ok) ? 0 : 1;
}
