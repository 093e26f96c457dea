// cg.h -- device-resident conjugate gradients for the matrix-free x-update (cg.hip).
#pragma once
#include "common.h"

namespace admm {

struct CgState {     // device scalars of one solve
  double rs;         // r.r
  double ynorm;      // ||y||
  int32_t iters;     // inner iterations of the current solve
  int32_t done;      // converged (or hit maxit): the remaining launches of the chunk are no-ops
  int64_t total;     // inner iterations since the run started
  int64_t capped;    // x-updates of this run that ended on the iteration cap with the residual still above tol
};

struct CgArgs {
  int64_t n;
  double shift;      // rho for lasso (D'D + rho I), 0 for LAD/Huber/SVM (D'D)
  double tol;        // relative residual ||y - Mx|| <= tol*||y||
  int32_t maxit;
  const double* y;
  double* x;
  double* r;
  double* p;
  double* q;
  double* part;      // [2][kMaxPartBlocks]
  CgState* st;
  const Ctrl* ctrl;
  int32_t* skip;     // nullable: mirrors (done || ctrl->stop) into the first word of a Ctrl-shaped block, so that the
                     // operator's own kernels (gemv_n / gemv_t, which only know `ctrl->stop`) become no-ops too
};

// q = sum of the gemv_t chunk partials + shift*p (and the block partials of p.q when with_dot)
void launch_cg_q(const CgArgs& a, const double* qin, int32_t nchunk, int64_t ldq, bool with_dot, hipStream_t stream);
// r = y - q, p = r, rs, ||y||, done-if-already-converged
void launch_cg_init(const CgArgs& a, hipStream_t stream);
// alpha, x/r update, beta, p update, scalar state advance (after launch_cg_q(..., with_dot = true))
void launch_cg_step_tail(const CgArgs& a, hipStream_t stream);
// the pieces of the tail, for operators that fuse the direction update into their own apply kernel (tv2d.hip):
// x += alpha p, r -= alpha q, partial r.r  |  rs <- r.r, iteration count, convergence flag
void launch_cg_update(const CgArgs& a, hipStream_t stream);
void launch_cg_advance(const CgArgs& a, hipStream_t stream);
// number of workgroups (= partial sums) every CG kernel of a length-n solve uses
int cg_num_blocks(int64_t n);

}  // namespace admm
