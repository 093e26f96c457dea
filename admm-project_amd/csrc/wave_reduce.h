// wave_reduce.h -- cross-lane sums of a 64-lane wave WITHOUT the LDS crossbar: gfx950 v_permlane{32,16}_swap exchange
// steps and a DPP butterfly inside each 16-lane row (measured in the symmetric x-solve: 25 us cheaper per 400 MB pass
// than the ds_bpermute / __shfl_xor form).  Shared by symv.hip and unwrapped.hip.
#pragma once
#include "common.h"

namespace admm {

constexpr int kSyPanel = 4;   // values reduced together: columns per panel of the symmetric kernel

// a.upper32 <-> b.lower32: afterwards a + b holds, in the lower 32 lanes, the 2-way sum of the old
// a and, in the upper 32 lanes, the 2-way sum of the old b.
__device__ __forceinline__ void swap_half(double& a, double& b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  a = __hiloint2double(r1[0], r0[0]);
  b = __hiloint2double(r1[1], r0[1]);
}
// odd 16-lane rows of a <-> even rows of b: same idea one level down
__device__ __forceinline__ void swap_row(double& a, double& b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  a = __hiloint2double(r1[0], r0[0]);
  b = __hiloint2double(r1[1], r0[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// reduce-scatter of 4 per-lane values over the 64 lanes: every lane returns the wave-wide sum of
// element (lane >> 4)  [lane bit 5 -> +2, bit 4 -> +1].
__device__ __forceinline__ double reduce_scatter4(double (&t)[kSyPanel]) {
  swap_half(t[0], t[2]);
  swap_half(t[1], t[3]);
  t[0] += t[2];
  t[1] += t[3];
  swap_row(t[0], t[1]);
  double r = t[0] + t[1];
  r += dpp_mov<0x128>(r);  // row_ror:8
  r += dpp_mov<0x124>(r);  // row_ror:4
  r += dpp_mov<0x4E>(r);   // quad_perm [2,3,0,1]
  r += dpp_mov<0xB1>(r);   // quad_perm [1,0,3,2]
  return r;
}

}  // namespace admm
