// cg.hip -- matrix-free x-update: conjugate gradients on (D'D + shift*I) x = y, warm-started from
// the previous x.  No reference counterpart (the reference always factors: lasso.m:168, lad.m:134);
// this is the A-streaming form north_star asks for ("LSQR when tall"): every inner iteration is
// exactly one A'(A p) unit = one gemv_n + one gemv_t pass over D (16mn bytes), nothing is factored
// and nothing n x n is stored.  In exact arithmetic CG on the normal equations generates the same
// iterates as LSQR on [D; sqrt(shift) I]; with unit-norm columns and m = 10n the system has
// condition number < 3, so plain CG is accurate to 1e-12 in ~15 iterations.
// All scalars (alpha, beta, residual norms, the convergence flag) stay on the device; the host only
// polls the 4-byte `done` flag every few inner iterations.
#include "cg.h"

namespace admm {

// q = sum_c gpart[c] (+ shift*p), block partial of p.q
__global__ __launch_bounds__(kBlock) void cg_q_kernel(CgArgs a, const double* __restrict__ qin, int32_t nchunk,
                                                      int64_t ldq, int with_dot) {
  if (a.ctrl->stop || a.st->done) return;
  __shared__ double scratch[4];
  double acc = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    double s = gather_chunks(qin, nchunk, ldq, i);
    const double pv = a.p[i];
    const double qv = s + a.shift * pv;
    a.q[i] = qv;
    acc += pv * qv;
  }
  if (with_dot) {
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) a.part[blockIdx.x] = t;
  }
}

// first residual: r = y - q, p = r, rs = r.r (partials)
__global__ __launch_bounds__(kBlock) void cg_init_kernel(CgArgs a) {
  if (a.ctrl->stop) return;
  __shared__ double scratch[4];
  double acc = 0.0, accy = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double yv = a.y[i];
    const double rv = yv - a.q[i];
    a.r[i] = rv;
    a.p[i] = rv;
    acc += rv * rv;
    accy += yv * yv;
  }
  const double t = block_sum(acc, scratch);
  const double ty = block_sum(accy, scratch);
  if (threadIdx.x == 0) {
    a.part[blockIdx.x] = t;
    a.part[kMaxPartBlocks + blockIdx.x] = ty;
  }
}

__device__ __forceinline__ double sum_parts(const double* part, int nblk, double* scratch) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) s += part[b];
  const double t = block_sum(s, scratch);
  __shared__ double bc;
  if (threadIdx.x == 0) bc = t;
  __syncthreads();
  return bc;
}

// one workgroup: rs = sum(parts), ||y||, decide whether x is already good enough
__global__ __launch_bounds__(kBlock) void cg_start_kernel(CgArgs a, int nblk) {
  if (a.ctrl->stop) return;
  __shared__ double scratch[4];
  const double rs = sum_parts(a.part, nblk, scratch);
  const double yy = sum_parts(a.part + kMaxPartBlocks, nblk, scratch);
  if (threadIdx.x == 0) {
    a.st->rs = rs;
    a.st->ynorm = sqrt(yy);
    a.st->iters = 0;
    a.st->done = (sqrt(rs) <= a.tol * sqrt(yy)) ? 1 : 0;
    if (a.skip) *a.skip = a.st->done;
  }
}

// alpha = rs / (p.q); x += alpha p; r -= alpha q; partial r.r
__global__ __launch_bounds__(kBlock) void cg_update_kernel(CgArgs a, int nblk) {
  if (a.ctrl->stop || a.st->done) return;
  __shared__ double scratch[4];
  const double pq = sum_parts(a.part, nblk, scratch);
  const double alpha = a.st->rs / pq;
  double acc = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    a.x[i] = a.x[i] + alpha * a.p[i];
    const double rv = a.r[i] - alpha * a.q[i];
    a.r[i] = rv;
    acc += rv * rv;
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) a.part[kMaxPartBlocks + blockIdx.x] = t;
}

// beta = rs_new / rs; p = r + beta p; the first block also advances the scalar state
__global__ __launch_bounds__(kBlock) void cg_direction_kernel(CgArgs a, int nblk) {
  if (a.ctrl->stop || a.st->done) return;
  __shared__ double scratch[4];
  const double rsn = sum_parts(a.part + kMaxPartBlocks, nblk, scratch);
  const double beta = rsn / a.st->rs;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    a.p[i] = a.r[i] + beta * a.p[i];
  // every block must have read st->rs before it changes: the scalar update runs in its own launch
}

__global__ void cg_advance_kernel(CgArgs a, int nblk) {
  if (a.ctrl->stop || a.st->done) return;
  __shared__ double scratch[4];
  const double rsn = sum_parts(a.part + kMaxPartBlocks, nblk, scratch);
  if (threadIdx.x == 0) {
    a.st->rs = rsn;
    a.st->iters += 1;
    a.st->total += 1;
    const bool converged = sqrt(rsn) <= a.tol * a.st->ynorm;
    if (converged || a.st->iters >= a.maxit) {
      if (!converged) a.st->capped += 1;
      a.st->done = 1;
      if (a.skip) *a.skip = 1;
    }
  }
}

static int cg_blocks(int64_t n) {
  int64_t b = ceil_div(n, kBlock);
  if (b > kMaxPartBlocks) b = kMaxPartBlocks;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

void launch_cg_q(const CgArgs& a, const double* qin, int32_t nchunk, int64_t ldq, bool with_dot, hipStream_t stream) {
  hipLaunchKernelGGL(cg_q_kernel, dim3(cg_blocks(a.n)), dim3(kBlock), 0, stream, a, qin, nchunk, ldq, with_dot ? 1 : 0);
}

void launch_cg_init(const CgArgs& a, hipStream_t stream) {
  const int nb = cg_blocks(a.n);
  hipLaunchKernelGGL(cg_init_kernel, dim3(nb), dim3(kBlock), 0, stream, a);
  hipLaunchKernelGGL(cg_start_kernel, dim3(1), dim3(kBlock), 0, stream, a, nb);
}

void launch_cg_step_tail(const CgArgs& a, hipStream_t stream) {
  const int nb = cg_blocks(a.n);
  hipLaunchKernelGGL(cg_update_kernel, dim3(nb), dim3(kBlock), 0, stream, a, nb);
  hipLaunchKernelGGL(cg_direction_kernel, dim3(nb), dim3(kBlock), 0, stream, a, nb);
  hipLaunchKernelGGL(cg_advance_kernel, dim3(1), dim3(kBlock), 0, stream, a, nb);
}

void launch_cg_update(const CgArgs& a, hipStream_t stream) {
  const int nb = cg_blocks(a.n);
  hipLaunchKernelGGL(cg_update_kernel, dim3(nb), dim3(kBlock), 0, stream, a, nb);
}

void launch_cg_advance(const CgArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(cg_advance_kernel, dim3(1), dim3(kBlock), 0, stream, a, cg_blocks(a.n));
}

int cg_num_blocks(int64_t n) { return cg_blocks(n); }

}  // namespace admm
