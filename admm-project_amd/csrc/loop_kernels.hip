// loop_kernels.hip -- the per-iteration vector work of the ADMM loop as fused kernels.
//
//   prox kernel     : admm.m:515-560 (relaxation, z-prox, u-update), 563-569 (fast ADMM
//                     extrapolation), 608-610 (history), and every partial sum needed by
//                     621-624 / 648-654 / 305-306 / 572-573, plus the NEXT x-update's rhs
//                     (getProxOps.m:1195, 1455, 1031, 1514, 1067) -- one pass over the vectors.
//   finalize kernel : one workgroup; sums the block partials in fixed order (bitwise
//                     reproducible), forms pnorm/dnorm/perr/derr/Hnormsq/objective, runs the
//                     convergence test (686-702) and the stop logic (706-722) on the device.
// All kernels no-op once ctrl->stop is set, so the host can enqueue ahead of the stop test.
#include "loop_kernels.h"
#include "finalize_device.h"
#include "prox_device.h"

namespace admm {

__device__ __forceinline__ void block_reduce_slots(double (&acc)[S_COUNT], double* part) {
  __shared__ double sred[4][S_COUNT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) {
    const double w = wave_sum(acc[s]);
    if (lane == 0) sred[wid][s] = w;
  }
  __syncthreads();
  if (threadIdx.x < S_COUNT) {
    const int s = threadIdx.x;
    part[s * kMaxPartBlocks + blockIdx.x] = ((sred[0][s] + sred[1][s]) + sred[2][s]) + sred[3][s];
  }
}

__global__ __launch_bounds__(kBlock) void prox_kernel(ProxArgs a, const Ctrl* __restrict__ ctrl) {
  // the three control words in one scalar round trip, before the branch
  const int32_t stop = ctrl->stop;
  const int64_t it = ctrl->iter;
  const double aprev = ctrl->acurr;
  if (stop) return;
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  double kcoef = 0.0;
  if (a.alg == 1) {  // admm.m:504, 567: aprev = acurr; acurr = (1+sqrt(1+4 aprev^2))/2
    const double acn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * aprev * aprev));
    kcoef = (aprev - 1.0) / acn;
  }
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const ProxIn in = prox_load(a, i);
    const double ax = gather_chunks(a.axsrc, a.naxpart, a.axld, i);
    prox_apply(a, i, ax, it, kcoef, in, acc);
  }
  block_reduce_slots(acc, a.part);
}

// (A variant of this kernel that gathered x_i straight from symv_lower_kernel's partial sums -- 16 elements x 16
// slots per workgroup, saving the symv_reduce launch -- was measured SLOWER on the headline loop: 10.29k vs
// 11.55k it/s.  625 workgroups instead of 40 make the 12-slot block reductions and the finalize kernel's
// partial sums cost more than the launch they save.  Re-measured after the finalize kernel's partial loads were
// unrolled and the prox loads batched: still 10.24k against 10.32k it/s on the same box -- the two-kernel form stays.)

__global__ __launch_bounds__(kBlock) void prez_kernel(PreZArgs a, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double ax = gather_chunks(a.axsrc, a.naxpart, a.axld, i);
    const double zp = a.z[i];
    const double ci = a.c ? a.c[i] : 0.0;
    // same expression as prox_kernel, so the split update is bit-identical to the fused one
    const double axh = (a.relax != 1.0) ? a.relax * ax - (1.0 - a.relax) * ((-zp) - ci) : ax;
    a.xh[i] = axh;
    if (a.rz) a.rz[i] = (a.add ? a.add[i] : 0.0) + a.rho * ((axh + a.uo[i]) - ci);
  }
}

void launch_prez(const PreZArgs& a, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(a.len, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(prez_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a, ctrl);
}

__global__ __launch_bounds__(kBlock) void negate_kernel(const double* __restrict__ z, double* __restrict__ bz,
                                                        int64_t len, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < len;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    bz[i] = -z[i];  // B = -1 (a general B: the loop's z buffer holds w = -B*z, so this is B*z as well)
}

void launch_negate(const double* z, double* bz, int64_t len, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(len, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(negate_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, z, bz, len, ctrl);
}

// grid = the prox kernel's block count: slots S_U2 / S_DU2 of every block it wrote are rewritten
__global__ __launch_bounds__(kBlock) void ufix_kernel(UFixArgs a, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ double scratch[4];
  const int64_t it = ctrl->iter;
  double su = 0.0, sdu = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double un = a.unew[i], uo = a.uold[i], zn = a.z[i];
    const double ci = a.c ? a.c[i] : 0.0;
    const double add = a.rhs_add ? a.rhs_add[i] : 0.0;
    a.u[i] = un;
    if (a.uhist) a.uhist[it * a.len + i] = un;
    su += un * un;
    const double du = un - uo;
    sdu += du * du;
    if (a.rhs) {  // as prox_apply's epilogue (plain ADMM: zx = z, ux = u)
      switch (a.rhs_kind) {
        case RHS_RHO_DTS: a.rhs[i] = a.rho * (zn - un) + add; break;
        case RHS_RHO_MINUS_Q: a.rhs[i] = a.rho * (zn - un) - add; break;
        case RHS_DIFF: a.rhs[i] = zn - un; break;
        case RHS_T1: a.rhs[i] = (ci + zn) - un; break;
        default: break;
      }
    }
  }
  const double tu = block_sum(su, scratch);
  const double tdu = block_sum(sdu, scratch);
  if (threadIdx.x == 0) {
    a.part[S_U2 * kMaxPartBlocks + blockIdx.x] = tu;
    a.part[S_DU2 * kMaxPartBlocks + blockIdx.x] = tdu;
  }
}

void launch_ufix(const UFixArgs& a, const Ctrl* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(ufix_kernel, dim3(static_cast<unsigned>(a.nblk < 1 ? 1 : a.nblk)), dim3(kBlock), 0, stream, a, ctrl);
}

void launch_prox(const ProxArgs& args, const Ctrl* ctrl, int* nblk_out, hipStream_t stream) {
  ProxArgs a = args;  // operands this variant does not read -> null (the kernel loads every non-null one up front)
  const bool need_ell = a.prox == PROX_HINGE || a.prox == PROX_01 || a.objx == OBJX_HINGE ||
                        a.objx == OBJX_ZEROONE || a.objx == OBJX_DOT;
  if (!need_ell) a.ell = nullptr;
  if (a.prox != PROX_GIVEN) a.zgiven = nullptr;
  if (a.prox != PROX_BOX) a.lb = a.ub = nullptr;
  const bool need_add = (a.alg != 2 && a.rhs && (a.rhs_kind == RHS_RHO_DTS || a.rhs_kind == RHS_RHO_MINUS_Q)) ||
                        a.objx == OBJX_SOLVE || a.objx == OBJX_SOLVE_QP;
  if (!need_add) a.rhs_add = nullptr;
  int64_t blocks = ceil_div(a.len, kBlock);
  if (blocks > kMaxPartBlocks) blocks = kMaxPartBlocks;
  if (blocks < 1) blocks = 1;
  *nblk_out = static_cast<int>(blocks);
  hipLaunchKernelGGL(prox_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a, ctrl);
}

// ---------------------------------------------------------------- alg 2: decide + extrapolate
__device__ __forceinline__ double sum_slot(const double* part, int slot, int nblk, double* scratch) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) s += part[slot * kMaxPartBlocks + b];
  return block_sum(s, scratch);
}

__global__ __launch_bounds__(kBlock) void fast_decide_kernel(FinArgs a) {
  Ctrl* ctrl = a.ctrl;
  if (ctrl->stop) return;
  __shared__ double scratch[4];
  const double s_duh = a.slots_reduced ? a.slots_reduced[S_DUH2] : sum_slot(a.part, S_DUH2, a.nblk, scratch);
  const double s_dzv = a.slots_reduced ? a.slots_reduced[S_DZV2] : sum_slot(a.part, S_DZV2, a.nblk, scratch);
  if (threadIdx.x == 0) {
    const double aprev = ctrl->acurr;  // admm.m:504
    const double dprev = ctrl->d;      // admm.m:509
    double d = 1.0 / a.rho * s_duh + a.rho * s_dzv;  // admm.m:572-573 (B = -I)
    double acn, coef, rst;
    if (d < a.restart * dprev) {  // admm.m:576-582
      acn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * aprev * aprev));
      coef = (aprev - 1.0) / acn;
      rst = 0.0;
    } else {  // admm.m:583-591
      acn = 1.0;
      coef = 0.0;
      rst = 1.0;
      d = dprev / a.restart;
    }
    ctrl->aprev = aprev;
    ctrl->acurr = acn;
    ctrl->dprev = dprev;
    ctrl->d = d;
    ctrl->coef = coef;
    ctrl->restart_flag = rst;
  }
}

void launch_fast_decide(const FinArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(fast_decide_kernel, dim3(1), dim3(kBlock), 0, stream, a);
}

__device__ __forceinline__ double rhs_value(int kind, double rho, double zx, double ux, double ci, double add) {
  switch (kind) {
    case RHS_RHO_DTS:
      return rho * (zx - ux) + add;
    case RHS_RHO_MINUS_Q:
      return rho * (zx - ux) - add;
    case RHS_DIFF:
      return zx - ux;
    case RHS_T1:
      return (ci + zx) - ux;
    default:
      return 0.0;
  }
}

__global__ __launch_bounds__(kBlock) void extrapolate_kernel(ExtrapArgs a, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t it = ctrl->iter;
  const double coef = ctrl->coef;
  const bool rst = ctrl->restart_flag != 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double z = a.z[i], u = a.u[i], zp = a.zprev[i], up = a.uprev[i];
    const double vn = rst ? zp : z + coef * (z - zp);
    const double uh = rst ? up : u + coef * (u - up);
    a.v[i] = vn;
    a.uhat[i] = uh;
    if (a.vhist) a.vhist[it * a.len + i] = vn;
    if (a.uhathist) a.uhathist[it * a.len + i] = uh;
    if (a.rhs)
      a.rhs[i] = rhs_value(a.rhs_kind, a.rho, vn, uh, a.c ? a.c[i] : 0.0, a.rhs_add ? a.rhs_add[i] : 0.0);
  }
}

void launch_extrapolate(const ExtrapArgs& a, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(a.len, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(extrapolate_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a, ctrl);
}

__global__ __launch_bounds__(kBlock) void zstate_kernel(ZStateArgs a, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t it = ctrl->iter;
  double coef = 0.0;
  bool rst = false;
  if (a.phase == 1) {
    coef = ctrl->coef;
    rst = ctrl->restart_flag != 0.0;
  } else if (a.alg == 1) {  // admm.m:504, 567-568, as prox_kernel takes it: finalize has not advanced acurr yet
    const double aprev = ctrl->acurr;
    coef = (aprev - 1.0) / (0.5 * (1.0 + sqrt(1.0 + 4.0 * aprev * aprev)));
  }
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    double zn, zp;
    if (a.phase == 0) {
      zn = a.znew[i];
      zp = a.z[i];
      a.zprev[i] = zp;
      a.z[i] = zn;
      if (a.zhist) a.zhist[it * a.len + i] = zn;
      if (a.alg != 1) continue;
    } else {
      zn = a.z[i];
      zp = a.zprev[i];
    }
    const double vn = rst ? zp : zn + coef * (zn - zp);
    a.v[i] = vn;
    if (a.vhist) a.vhist[it * a.len + i] = vn;
  }
}

void launch_zstate(const ZStateArgs& a, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(a.len, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zstate_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a, ctrl);
}

__global__ __launch_bounds__(kBlock) void initial_rhs_kernel(int64_t len, int kind, double rho,
                                                             const double* __restrict__ zx,
                                                             const double* __restrict__ ux,
                                                             const double* __restrict__ c,
                                                             const double* __restrict__ add,
                                                             double* __restrict__ rhs) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < len;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    rhs[i] = rhs_value(kind, rho, zx[i], ux[i], c ? c[i] : 0.0, add ? add[i] : 0.0);
}

void launch_initial_rhs(int64_t len, int rhs_kind, double rho, const double* zx, const double* ux, const double* c,
                        const double* rhs_add, double* rhs, hipStream_t stream) {
  int64_t blocks = ceil_div(len, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(initial_rhs_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, len, rhs_kind,
                     rho, zx, ux, c, rhs_add, rhs);
}

// ---------------------------------------------------------------- finalize
// COHERENT: called by the last workgroup of the prox kernel to arrive (prox_fin_kernel): the block partials were
// published write-through by other compute units during this launch and are read past the L1 (sc1 loads).
__global__ __launch_bounds__(kBlock) void finalize_kernel(FinArgs a) {
  if (a.ctrl->stop) return;
  finalize_body<false>(a);
}

// ---------------------------------------------------------------- one-launch tail of an A = I iteration
// prox + finalize (and, after the lower-triangle x-solve, the sum of its partial rows) in ONE launch.  Workgroup =
// one 128-element tile x 4 slots: every thread sums a quarter of the x-solve's partial rows of its element (all
// loads of a thread in one round), the quarters meet in LDS, the first 128 threads run the fused element update, the
// block partials are published write-through (sc1) and the workgroup arrives on ctrl->arrive; the LAST workgroup to
// arrive runs the finalize logic (norms, tolerances, H-norm, stop) on the device in the same launch.  Replaces
// three launches (symv_reduce, prox, finalize) and two kernel boundaries: 21.7 -> ~10 us on the headline loop.
// Sums are taken in a fixed order whatever the arrival order: bitwise reproducible.
constexpr int kTailTile = 128;
constexpr int kTailSlots = 4;                       // threads per element: each sums a quarter of the partial rows
constexpr int kTailBlock = kTailTile * kTailSlots;  // 512 threads
constexpr int kTailRows = 24;                       // partial rows a thread loads in one round

__device__ __forceinline__ double tail_gather(const ProxArgs& a, int64_t i, int32_t p0, int32_t p1) {
  // unified partial row p of element i (diagonal tile d = i / 128): p <= d -> axsrc[p] (N-part), else ax_t[p - 1]
  // (one-block triangular solves, ax_tri: the caller's p counts from the diagonal tile, rows d + p of ax_t)
  const int32_t d = static_cast<int32_t>(i / kTailTile);
  double s = 0.0;
  for (int32_t p = p0; p < p1; p += kTailRows) {
    double v[kTailRows];
#pragma unroll
    for (int k = 0; k < kTailRows; ++k) {
      const int32_t q = (p + k < p1) ? p + k : p1 - 1;
      const double* src = a.ax_tri ? a.ax_t + static_cast<int64_t>(d + q) * a.axld
                          : (!a.ax_t || q <= d) ? a.axsrc + static_cast<int64_t>(q) * a.axld
                                                : a.ax_t + static_cast<int64_t>(q - 1) * a.axld;
      v[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < kTailRows; ++k)
      if (p + k < p1) s += v[k];
  }
  return s;
}

// defer = 1: the finalize logic is NOT run here; the block partials are stored plainly and the next launch on the
// stream (the x-solve of the next iteration, or a stand-alone finalize) takes them after the kernel boundary.
__global__ __launch_bounds__(kTailBlock) void prox_fin_kernel(ProxArgs a, FinArgs f, Ctrl* __restrict__ ctrl,
                                                              int32_t defer) {
  const int32_t stop = ctrl->stop;
  const int64_t it = ctrl->iter;
  const double aprev = ctrl->acurr;
  // (the stop test sits below the loads: they do not depend on it, and in front of them it costs this 9 us kernel one
  // more memory round trip)
  __shared__ double quarter[kTailSlots - 1][kTailTile];
  __shared__ int32_t last;
  const int e = threadIdx.x & (kTailTile - 1), slot = threadIdx.x >> 7;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kTailTile + e;
  const int64_t ic = i < a.len ? i : a.len - 1;
  ProxIn in{};
  if (slot == 0) in = prox_load(a, ic);  // in flight together with the partial rows
  double ax;
  {  // partial rows of this element, split over the four slots: naxpart + 1 rows after the lower-triangle x-solve
     // (N-part rows up to the diagonal tile, T-part rows beyond), naxpart chunk rows of a column-chunked GEMV otherwise
    const int32_t dblk = static_cast<int32_t>(blockIdx.x);  // (kTailTile = the x-solve's tile: one diagonal tile per workgroup)
    const int32_t P = a.ax_tri ? a.naxpart - dblk : (a.ax_t ? a.naxpart + 1 : a.naxpart);
    const int32_t q = (P + kTailSlots - 1) / kTailSlots;
    const int32_t p0 = slot * q, p1 = (p0 + q < P) ? p0 + q : P;
    ax = p0 < P ? tail_gather(a, ic, p0, p1) : 0.0;
  }
  if (stop) return;  // (uniform; nothing has been stored yet)
  if (slot > 0) quarter[slot - 1][e] = ax;
  __syncthreads();
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  if (slot == 0 && i < a.len) {
    double kcoef = 0.0;
    if (a.alg == 1) {
      const double acn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * aprev * aprev));
      kcoef = (aprev - 1.0) / acn;
    }
    prox_apply(a, i, ((ax + quarter[0][e]) + quarter[1][e]) + quarter[2][e], it, kcoef, in, acc);
  }
  // block partials of the two waves that hold elements, published write-through
  __shared__ double sred[2][S_COUNT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (wid < 2) {
#pragma unroll
    for (int s = 0; s < S_COUNT; ++s) {
      const double w = wave_sum(acc[s]);
      if (lane == 0) sred[wid][s] = w;
    }
  }
  __syncthreads();
  if (defer) {
    if (threadIdx.x < S_COUNT) a.part[threadIdx.x * kMaxPartBlocks + blockIdx.x] = sred[0][threadIdx.x] + sred[1][threadIdx.x];
    return;
  }
  if (threadIdx.x < S_COUNT) {
    const int s = threadIdx.x;
    __hip_atomic_store(a.part + s * kMaxPartBlocks + blockIdx.x, sred[0][s] + sred[1][s], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains: the element stores as well
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t old = __hip_atomic_fetch_add(&ctrl->arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (old == static_cast<int32_t>(gridDim.x) - 1) ? 1 : 0;
    if (last) __hip_atomic_store(&ctrl->arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last || threadIdx.x >= kBlock) return;  // the finalize logic is written for one 256-thread workgroup
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  finalize_body<true>(f);
}

FinArgs prox_fin_args(const ProxArgs& a, const FinArgs& f) {
  FinArgs ff = f;
  ff.nblk = static_cast<int32_t>(ceil_div(a.len, kTailTile));
  ff.g = nullptr;  // nothing but the block partials (and, for A = D, the x of this iteration) feeds the finalize logic
  ff.objpart = nullptr;
  ff.nobjpart = 0;
  ff.slots_reduced = nullptr;
  ff.objp_reduced = nullptr;
  return ff;
}

void launch_prox_fin(const ProxArgs& args, const FinArgs& f, Ctrl* ctrl, int* nblk_out, hipStream_t stream,
                     bool defer) {
  ProxArgs a = args;
  const bool need_ell = a.prox == PROX_HINGE || a.prox == PROX_01 || a.objx == OBJX_HINGE ||
                        a.objx == OBJX_ZEROONE || a.objx == OBJX_DOT;
  if (!need_ell) a.ell = nullptr;
  if (a.prox != PROX_GIVEN) a.zgiven = nullptr;
  if (a.prox != PROX_BOX) a.lb = a.ub = nullptr;
  const bool need_add = (a.alg != 2 && a.rhs && (a.rhs_kind == RHS_RHO_DTS || a.rhs_kind == RHS_RHO_MINUS_Q)) ||
                        a.objx == OBJX_SOLVE || a.objx == OBJX_SOLVE_QP;
  if (!need_add) a.rhs_add = nullptr;
  const int64_t blocks = ceil_div(a.len, kTailTile);
  *nblk_out = static_cast<int>(blocks);
  const FinArgs ff = prox_fin_args(a, f);
  hipLaunchKernelGGL(prox_fin_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kTailBlock), 0, stream, a, ff, ctrl,
                     defer ? 1 : 0);
}

void launch_finalize(const FinArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kBlock), 0, stream, a);
}

// ---------------------------------------------------------------- all-reduce payload packing
__global__ __launch_bounds__(kBlock) void pack_slots_kernel(const double* __restrict__ part, int32_t nblk,
                                                            double* __restrict__ out16,
                                                            const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  const int slot = threadIdx.x >> 4, sub = threadIdx.x & 15;
  double v = 0.0;
  if (slot < S_COUNT)
    for (int b = sub; b < nblk; b += 16) v += part[slot * kMaxPartBlocks + b];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (sub == 0) out16[slot] = v;
}

void launch_pack_slots(const double* part, int32_t nblk, double* out16, const Ctrl* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(pack_slots_kernel, dim3(1), dim3(kBlock), 0, stream, part, nblk, out16, ctrl);
}

__global__ __launch_bounds__(kBlock) void pack_sum_kernel(const double* __restrict__ v, int32_t n,
                                                          double* __restrict__ out1, const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double scratch[4];
  double s = 0.0;
  for (int b = threadIdx.x; b < n; b += blockDim.x) s += v[b];
  const double t = block_sum(s, scratch);
  if (threadIdx.x == 0) out1[0] = t;
}

void launch_pack_sum(const double* v, int32_t n, double* out1, const Ctrl* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(pack_sum_kernel, dim3(1), dim3(kBlock), 0, stream, v, n, out1, ctrl);
}

// ---------------------------------------------------------------- small helpers
__global__ __launch_bounds__(kBlock) void residual_sq_kernel(const double* __restrict__ part, int32_t nchunk,
                                                             int64_t ld, const double* __restrict__ s, int64_t len,
                                                             double* __restrict__ objpart,
                                                             const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double scratch[4];
  double acc = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double y = gather_chunks(part, nchunk, ld, i);
    const double r = y - s[i];
    acc += r * r;
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) objpart[blockIdx.x] = t;
}

void launch_residual_sq(const double* part, int32_t nchunk, int64_t ld, const double* s, int64_t len, double* objpart,
                        int* nblk_out, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(len, kBlock);
  if (blocks > kMaxPartBlocks) blocks = kMaxPartBlocks;
  if (blocks < 1) blocks = 1;
  *nblk_out = static_cast<int>(blocks);
  hipLaunchKernelGGL(residual_sq_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, part, nchunk,
                     ld, s, len, objpart, ctrl);
}

__global__ __launch_bounds__(kBlock) void qp_objective_kernel(const double* __restrict__ part, int32_t nchunk,
                                                              int64_t ld, const double* __restrict__ x,
                                                              const double* __restrict__ q, int64_t len,
                                                              double* __restrict__ objpart,
                                                              const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double scratch[4];
  double acc = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double y = gather_chunks(part, nchunk, ld, i);
    acc += x[i] * (0.5 * y + q[i]);
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) objpart[blockIdx.x] = t;
}

void launch_qp_objective(const double* part, int32_t nchunk, int64_t ld, const double* x, const double* q,
                         int64_t len, double* objpart, int* nblk_out, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(len, kBlock);
  if (blocks > kMaxPartBlocks) blocks = kMaxPartBlocks;
  if (blocks < 1) blocks = 1;
  *nblk_out = static_cast<int>(blocks);
  hipLaunchKernelGGL(qp_objective_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, part, nchunk,
                     ld, x, q, len, objpart, ctrl);
}

__global__ __launch_bounds__(kBlock) void run_init_kernel(RunInitArgs a) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x, T = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = t; i < a.nA; i += T) a.x[i] = 0.0;
  for (int64_t i = t; i < a.len; i += T) {
    a.z[i] = 0.0;
    a.u[i] = 0.0;
    a.v[i] = 0.0;
    a.uhat[i] = 0.0;
  }
  for (int64_t i = t; i < a.N; i += T) {
#pragma unroll
    for (int k = 0; k < 9; ++k) a.scal[k][i] = 0.0;
  }
  if (t == 0) *a.ctrl = a.c0;
}

void launch_run_init(const RunInitArgs& a, hipStream_t stream) {
  int64_t most = a.len > a.nA ? a.len : a.nA;
  if (a.N > most) most = a.N;
  int64_t blocks = ceil_div(most, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(run_init_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a);
}

__global__ __launch_bounds__(kBlock) void obj_compare_kernel(const double* __restrict__ pa, int na, double sa, double ca,
                                                             const double* __restrict__ pb, int nb, double sb, double cb,
                                                             double* __restrict__ disc, const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double scratch[4];
  double va = 0.0, vb = 0.0;
  for (int i = threadIdx.x; i < na; i += kBlock) va += pa[i];
  for (int i = threadIdx.x; i < nb; i += kBlock) vb += pb[i];
  const double ta = block_sum(va, scratch);
  __syncthreads();
  const double tb = block_sum(vb, scratch);
  if (threadIdx.x == 0) {
    const double a = sa * ta + ca, b = sb * tb + cb;
    const double rel = fabs(a - b) / fmax(fabs(a), 1e-300);
    if (!(rel <= disc[0])) disc[0] = rel;  // NaN-safe maximum
  }
}

void launch_obj_compare(const double* pa, int na, double sa, double ca, const double* pb, int nb, double sb, double cb,
                        double* disc, const Ctrl* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(obj_compare_kernel, dim3(1), dim3(kBlock), 0, stream, pa, na, sa, ca, pb, nb, sb, cb, disc, ctrl);
}

__global__ __launch_bounds__(kBlock) void combine_kernel(const double* __restrict__ part, int32_t nchunk, int64_t ld,
                                                         double alpha, const double* __restrict__ y, double beta,
                                                         const double* __restrict__ add, double* __restrict__ x,
                                                         int64_t len, const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    double s = gather_chunks(part, nchunk, ld, i);
    double v = alpha * s;
    if (y) v += beta * y[i];
    if (add) v += add[i];
    x[i] = v;
  }
}

void launch_combine(const double* part, int32_t nchunk, int64_t ld, double alpha, const double* y, double beta,
                    const double* add, double* x, int64_t len, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(len, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(combine_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, part, nchunk, ld,
                     alpha, y, beta, add, x, len, ctrl);
}

}  // namespace admm
