// consensus.h -- consensus (global-variable) lasso over row slices, getProxOps.m:383-442 + 1217-1343.
#pragma once
#include <vector>

#include "kernels.h"

namespace admm {

// cached factor of one slice and how it is applied
struct SliceFactor {
  double* F = nullptr;      // lower Cholesky factor of D_k'D_k + rho*I   (getProxOps.m:424-435)
  int64_t n = 0, ld = 0;
  double* dinv = nullptr;   // inverted 64x64 diagonal blocks
  double* Minv = nullptr;   // explicit inverse (xsolve = inverse), tile-padded like the engine's own (symv.hip)
  int64_t ldM = 0;
  SymvPlan planSy{};        // lower-triangle application for n >= 1536, one wave per column below
  TrsvPlan trsv{};
  double* work = nullptr;   // device storage of the blocked-substitution plan (xsolve = trsv)
  GemvTPlan plan{};
  double* part = nullptr;
};

struct ConsSlice {
  double* D = nullptr;      // m x n rows of this slice
  int64_t m = 0, ld = 0;
  double* s = nullptr;      // its rows of the signal
  double* Dts = nullptr;    // D_k' * s_k   (getProxOps.m:407-408)
  GemvNPlan planN{};
  GemvTPlan planT{};
  SliceFactor fac;
};

struct ConsArgs {
  int64_t n, ldn;
  int32_t K;                // local slices
  int32_t Ntot;             // slices over all ranks (slicenum)
  double rho, lambda;
  const double* sums;       // [2][ldn]: sum_k x_k ; sum_k u_k (already all-reduced when sharded)
  double* X;                // [K][ldn] x_k
  double* U;                // [K][ldn] u_k
  double* zc;               // consensus z (the closure variable; the z handed to admm is 0, q9)
  double* xave;
  double* xaveprev;
  double* ubar;             // mean_k u_k = what altu returns (getProxOps.m:1312-1326)
  double* xhist;            // results.xvals column = xave
  double* zhist;            // zeros (q9)
  double* uhist;            // ubar
  double* part;             // [S_COUNT][kMaxPartBlocks]
};

void launch_cons_rhs(int64_t n, double rho, const double* z, const double* u, const double* Dts, double* y,
                     const Ctrl* ctrl, hipStream_t stream);
void launch_cons_sum(int64_t n, int64_t ldn, int32_t K, const double* X, const double* U, double* sums,
                     const Ctrl* ctrl, hipStream_t stream);
void launch_cons_update(const ConsArgs& a, const Ctrl* ctrl, int* nblk_out, hipStream_t stream);

}  // namespace admm
