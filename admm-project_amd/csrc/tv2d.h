// tv2d.h -- 2-D anisotropic total variation (BASELINE config 5 as literally written: an H x W image,
// matrix-free x-update).  Engine-side extension: the reference's totalvariation.m is 1-D only.
#pragma once
#include "cg.h"
#include "common.h"

namespace admm {

// Image x is H x W, column-major (row index i fastest), N = H*W.  D = [Dv; Dh] is 2N x N:
//   (Dv x)[i,j] = x[i,j] - x[i+1,j]   (i < H-1, else 0)      first  N rows
//   (Dh x)[i,j] = x[i,j] - x[i,j+1]   (j < W-1, else 0)      second N rows
// so z, u have 2N elements and D'D is the 5-point Neumann Laplacian.
struct Tv2Args {
  int64_t H, W;
  double rho, thresh;      // thresh = lambda/rho
  const double* s;         // image (N)
  const double* x;         // current x (N)
  const double* z;         // current z, u (2N, read); the fused kernel with state_in reads v = z + u through z
  const double* u;
  double* zo;              // next compact state v = z + u (2N, written by the fused kernel; ping-pong)
  int32_t objevals;
  double* xhist;
  double* zhist;
  double* uhist;
  double* part;            // [S_COUNT][kMaxPartBlocks]
};

// w = rho * D'D p   (the CG operator is I + rho*D'D: cg_q_kernel adds the identity part)
void launch_tv2d_laplace(int64_t H, int64_t W, double rho, const double* p, double* w, const Ctrl* ctrl,
                         hipStream_t stream);
// One fused CG kernel per inner iteration: beta = (r.r)_new / (r.r)_old (0 when `first`), p_new = r + beta*p_old,
// q = (I + rho*D'D) p_new, block partials of p_new.q -- the direction update and the operator in one pass
// (p is ping-ponged because the stencil needs the neighbours' OLD p).  a.p = p_old, a.q = q; uses a.part as
// launch_cg_update / launch_cg_advance do.
void launch_tv2d_cg_pq(int64_t H, int64_t W, double rho, const CgArgs& a, double* p_new, bool first,
                       hipStream_t stream);
// b = s + rho * D'(z - u)      right-hand side of the x-update
void launch_tv2d_rhs(const Tv2Args& a, double* b, const Ctrl* ctrl, hipStream_t stream);
// prox + dual stencils + the NEXT x-update's right-hand side in one pass (launch_tv2d_rhs is still needed once, for the
// first iteration).  The loop state is v = z + u (z = soft(v), u = v - z): state_in = false reads a.z / a.u (a run's
// first iteration), true reads v from a.z; a.zo receives the new v either way.
void launch_tv2d_fused(const Tv2Args& a, bool state_in, double* bnext, const Ctrl* ctrl, int* nblk_out,
                       hipStream_t stream);
// z = soft(v, thresh), u = v - z   (len = 2N): the iterates behind a compact state
void launch_tv2d_expand(const double* v, double thresh, int64_t len, double* z, double* u, hipStream_t stream);

// Unfused building blocks for the fast / accelerated ADMM variants: ax = D*x (2N) with the objective in block partials
// and the x history column; the D' stencils of the dual residual / tolerance from dz = z - zprev and u
void launch_tv2d_dx(int64_t H, int64_t W, double lambda, int objevals, const double* x, const double* s, double* ax,
                    double* objpart, int* nobj_out, double* xhist, const Ctrl* ctrl, hipStream_t stream);
void launch_tv2d_dual_vec(int64_t H, int64_t W, const double* dz, const double* u, double* part, int nblk,
                          const Ctrl* ctrl, hipStream_t stream);

}  // namespace admm
