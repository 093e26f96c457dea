// consensus.h -- consensus (global-variable) lasso over row slices, getProxOps.m:383-442 + 1217-1343.
#pragma once
#include <vector>

#include "kernels.h"

namespace admm {

// explicit inverses below this order are applied one wave per column (cache-resident, latency-bound: a 128 x 128
// tile of the lower-triangle kernel is 32 dependent panel steps of one wave)
constexpr int64_t kSymvHalfMin = 1536;

// a cached Cholesky factor and how inv(L L') is applied
struct SliceFactor {
  double* F = nullptr;      // lower Cholesky factor (getProxOps.m:424-435, lasso.m:168, lad.m:134)
  int64_t n = 0, ld = 0;
  double* dinv = nullptr;   // inverted 64x64 diagonal blocks
  int mode = ADMM_XSOLVE_TRSV;  // ADMM_XSOLVE_TRSV or ADMM_XSOLVE_INVERSE: the form in use
  double* Minv = nullptr;   // explicit inverse (mode = inverse), tile-padded (symv.hip)
  int64_t ldM = 0;
  SymvPlan planSy{};        // lower-triangle application for n >= kSymvHalfMin, one wave per column below
  TrsvPlan trsv{};          // blocked triangular solves (mode = trsv)
  double* work = nullptr;   // device storage of that plan
  // diagnostics (admm_engine_info)
  double diag_min = 0.0, diag_max = 0.0, cond_diag = 0.0;  // (max L_ii / min L_ii)^2 <= cond(L L')
  bool probed = false;      // both forms were built and compared on a system with a known solution
  double err_inv = 0.0, err_trsv = 0.0, probe_diff = 0.0;
  double err_trsv_one = __builtin_nan("");  // the one-block form's error on the same system (NaN: not built)
  int32_t chol_info = 0;
  bool pinv = false;        // Minv is the pseudo-inverse of a rank-deficient D'D (linear SVM)
  int64_t rank = 0;
  int jacobi_sweeps = 0;
};

struct ConsSlice {
  double* D = nullptr;      // m x n rows of this slice
  int64_t m = 0, ld = 0;
  double* s = nullptr;      // its rows of the signal
  double* Dts = nullptr;    // D_k' * s_k   (getProxOps.m:407-408)
  GemvNPlan planN{};
  GemvTPlan planT{};
  SliceFactor fac;
  bool fat = false;         // rows < columns: fac factors D_k D_k'/rho + I (order m), applied through the Woodbury form
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
  const double* Dts;        // [K][ldn] D_k'*s_k of the local slices
  double* Y;                // [K][ldn] right-hand sides of the NEXT x-update, y_k = rho*(z - u_k) + D_k's_k
                            // (getProxOps.m:1240), written by the update kernel: no per-slice rhs launches
};

// the same from the x-solves' partial rows (slices with an explicit inverse of order >= kSymvHalfMin): also writes X
int launch_cons_gather_sum(int64_t n, int64_t ldn, int32_t K, const double* npart, const double* tpart, int64_t pstride,
                           int64_t ldp, int32_t ntile, double* X, const double* U, double* sums, const double* center,
                           double* qpart, const Ctrl* ctrl, hipStream_t stream);
int launch_cons_sum(int64_t n, int64_t ldn, int32_t K, const double* X, const double* U, double* sums,
                    const double* center, double* qpart, const Ctrl* ctrl, hipStream_t stream);
void launch_cons_update(const ConsArgs& a, const Ctrl* ctrl, int* nblk_out, hipStream_t stream);
// gather + update in one launch (unsharded runs whose slices all left partial rows; a.K <= 16)
bool cons_gather_update_ok(const ConsArgs& a);
void launch_cons_gather_update(const ConsArgs& a, const double* npart, const double* tpart, int64_t pstride, int64_t ldp,
                               int32_t ntile, const Ctrl* ctrl, int* nblk_out, hipStream_t stream);

}  // namespace admm
