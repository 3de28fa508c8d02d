// 16-byte activation loads / stores for MFMA tiles through a DPP transpose (gemm_res.hip, mdta_dqk.hip).
#pragma once
#include "pir_common.h"

// Exchange of a register-index bit with a lane-index bit (the 2 x 2 block transpose that a 4 x 4 transpose is made of)
// for lane bits 2 and 3: a DPP row shift by 4 (8) lanes whose bank mask writes only the lanes with that bit set
// (cleared) IS the conditional move - one VALU operation per register, no select.  Banks are the four lane quads of a
// 16-lane row: lane bit 2 set = banks 1, 3 (0xA), clear = 0x5; lane bit 3 set = banks 2, 3 (0xC), clear = 0x3.
template <int BIT>
__device__ __forceinline__ void res_exchange(float& lo, float& hi) {
  constexpr int SH = BIT == 2 ? 4 : 8, SET = BIT == 2 ? 0xA : 0xC, CLR = BIT == 2 ? 0x5 : 0x3;
  const int l = __builtin_bit_cast(int, lo), h = __builtin_bit_cast(int, hi);
  // lanes with the bit set: lo <- hi of the lane SH below;  lanes with it clear: hi <- lo of the lane SH above
  const int nl = __builtin_amdgcn_update_dpp(l, h, 0x110 + SH, 0xf, SET, false);
  const int nh = __builtin_amdgcn_update_dpp(h, l, 0x100 + SH, 0xf, CLR, false);
  lo = __builtin_bit_cast(float, nl); hi = __builtin_bit_cast(float, nh);
}
// v[e] (e = 2 e1 + e0) at lane bits (l3, l2) = (a1, a0)  ->  v[2 a1 + a0] at lane bits (e1, e0): the 4 x 4 transpose
// between four registers and the lane-index bits 3, 2 in eight VALU operations.
__device__ __forceinline__ void res_transpose4(float& v0, float& v1, float& v2, float& v3) {
  res_exchange<2>(v0, v1); res_exchange<2>(v2, v3);
  res_exchange<3>(v0, v2); res_exchange<3>(v1, v3);
}


// per-lane byte offset + wave-uniform row term for a buffer STORE, added at the point of use.  (Written as plain
// arithmetic the compiler precomputes one offset register per store of a tile - 12 to 16 loop-invariant registers that
// spill in the persistent kernels; a scalar offset operand, which does this for loads, gave wrong results on stores.)
__device__ __forceinline__ int pir_row_offset(int lane_off, int row_off) {
  int r;
  asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "s"(row_off), "v"(lane_off));
  return r;
}
