// jacobi.hip -- symmetric eigen-decomposition W = V diag(lambda) V' of a positive SEMI-definite n x n matrix by
// one-sided (Hestenes) Jacobi, used to form the pseudo-inverse W^+ when D'D is rank deficient:
//   linearsvm.m:185 / unwrappedadmm.m:76  Dplus = pinv(D);  x = Dplus*(z-u) = (D'D)^+ D'(z-u)   (getProxOps.m:1067)
// (cropped MNIST has pixels that are zero in every sample and duplicated columns: chol(D'D) breaks down there,
// pinv does not).
//
// One-sided Jacobi works on the columns of B = W*V (V starts as I): a rotation of the column pair (p, q) that
// makes b_p and b_q orthogonal is applied to B and to V; when all pairs are orthogonal the column norms of B are
// the eigenvalues (W is PSD: singular values == eigenvalues) and V holds the eigenvectors.  A sweep is n-1
// rounds of n/2 disjoint pairs (round-robin tournament order); every round is one launch, one workgroup per
// pair: coalesced streaming of two columns of B and V, wave64 shuffle reductions for the three dot products.
// Setup only (once per engine); O(n^3) bandwidth-bound work per sweep, ~6-10 sweeps.
#include <algorithm>
#include <cmath>
#include <vector>

#include "kernels.h"

namespace admm {

typedef double double2_t __attribute__((ext_vector_type(2)));

// pair (p, q) number `k` of round `r` of the round-robin tournament over ne (even) players
__device__ __forceinline__ void jacobi_pair(int32_t ne, int32_t r, int32_t k, int32_t& p, int32_t& q) {
  const int32_t m1 = ne - 1;
  if (k == 0) {
    p = m1;
    q = r % m1;
  } else {
    p = (r + k) % m1;
    q = (r - k + m1) % m1;
  }
  if (p > q) {
    const int32_t t = p;
    p = q;
    q = t;
  }
}

__global__ __launch_bounds__(kBlock) void jacobi_round_kernel(double* __restrict__ B, int64_t ldb,
                                                              double* __restrict__ V, int64_t ldv, int32_t n,
                                                              int32_t ne, int32_t round, double tol, double null2,
                                                              int32_t* __restrict__ rotations) {
  __shared__ double sred[3][kBlock / kWave];
  __shared__ double scs[2];
  int32_t p, q;
  jacobi_pair(ne, round, static_cast<int32_t>(blockIdx.x), p, q);
  if (q >= n) return;  // padding player
  double* __restrict__ bp = B + static_cast<int64_t>(p) * ldb;
  double* __restrict__ bq = B + static_cast<int64_t>(q) * ldb;
  double a = 0.0, b = 0.0, g = 0.0;
  for (int32_t i = threadIdx.x; i < n; i += kBlock) {
    const double x = bp[i], y = bq[i];
    a = __builtin_fma(x, x, a);
    b = __builtin_fma(y, y, b);
    g = __builtin_fma(x, y, g);
  }
  a = wave_sum(a);
  b = wave_sum(b);
  g = wave_sum(g);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    sred[0][wid] = a;
    sred[1][wid] = b;
    sred[2][wid] = g;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double aa = 0.0, bb = 0.0, gg = 0.0;
    for (int w = 0; w < kBlock / kWave; ++w) {
      aa += sred[0][w];
      bb += sred[1][w];
      gg += sred[2][w];
    }
    double c = 1.0, s = 0.0;
    // columns at the rounding-noise level (squared norm <= null2) are null directions: never rotated again
    if (aa > null2 && bb > null2 && fabs(gg) > tol * sqrt(aa * bb)) {
      const double zeta = (bb - aa) / (2.0 * gg);
      const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      c = 1.0 / sqrt(1.0 + t * t);
      s = c * t;
      atomicAdd(rotations, 1);
    }
    scs[0] = c;
    scs[1] = s;
  }
  __syncthreads();
  const double c = scs[0], s = scs[1];
  if (s == 0.0) return;
  double* __restrict__ vp = V + static_cast<int64_t>(p) * ldv;
  double* __restrict__ vq = V + static_cast<int64_t>(q) * ldv;
  for (int32_t i = threadIdx.x; i < n; i += kBlock) {
    const double x = bp[i], y = bq[i];
    bp[i] = c * x - s * y;
    bq[i] = s * x + c * y;
    const double vx = vp[i], vy = vq[i];
    vp[i] = c * vx - s * vy;
    vq[i] = s * vx + c * vy;
  }
}

// lam[j] = ||B(:, j)||
__global__ __launch_bounds__(kBlock) void jacobi_norms_kernel(const double* __restrict__ B, int64_t ldb, int32_t n,
                                                              double* __restrict__ lam) {
  __shared__ double scratch[kBlock / kWave];
  const double* __restrict__ bj = B + static_cast<int64_t>(blockIdx.x) * ldb;
  double a = 0.0;
  for (int32_t i = threadIdx.x; i < n; i += kBlock) a = __builtin_fma(bj[i], bj[i], a);
  a = block_sum(a, scratch);
  if (threadIdx.x == 0) lam[blockIdx.x] = sqrt(a);
}

__global__ __launch_bounds__(kBlock) void jacobi_identity_kernel(double* __restrict__ V, int64_t ldv, int32_t n) {
  const int64_t j = blockIdx.x;
  for (int32_t i = threadIdx.x; i < n; i += kBlock) V[i + j * ldv] = (i == j) ? 1.0 : 0.0;
}

// V(:, j) *= scale[j]
__global__ __launch_bounds__(kBlock) void jacobi_scale_cols_kernel(double* __restrict__ V, int64_t ldv, int32_t n,
                                                                   const double* __restrict__ scale) {
  const int64_t j = blockIdx.x;
  const double s = scale[j];
  for (int32_t i = threadIdx.x; i < n; i += kBlock) V[i + j * ldv] *= s;
}

// W: n x n symmetric PSD, FULL storage (both triangles), overwritten by B = W*V.  V: n x n (ldv), receives the
// eigenvectors; lam_host[n]: eigenvalues (unsorted, paired with the columns of V).  `rot` is a device int32.
int jacobi_eig_psd(double* W, int64_t n, int64_t ldw, double* V, int64_t ldv, double* lam_dev, int32_t* rot,
                   std::vector<double>* lam_host, int* sweeps_out, hipStream_t stream) {
  const int32_t nn = static_cast<int32_t>(n);
  const int32_t ne = (nn + 1) & ~1;
  hipLaunchKernelGGL(jacobi_identity_kernel, dim3(static_cast<unsigned>(nn)), dim3(kBlock), 0, stream, V, ldv, nn);
  // noise floor of a column of B: n * eps * (largest column norm of W)
  hipLaunchKernelGGL(jacobi_norms_kernel, dim3(static_cast<unsigned>(nn)), dim3(kBlock), 0, stream, W, ldw, nn, lam_dev);
  lam_host->resize(static_cast<size_t>(n));
  ADMM_HIP_TRY(hipMemcpyAsync(lam_host->data(), lam_dev, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
  ADMM_HIP_TRY(hipStreamSynchronize(stream));
  double cmax = 0.0;
  for (double v : *lam_host) cmax = v > cmax ? v : cmax;
  const double nulltol = static_cast<double>(n) * 2.220446049250313e-16 * cmax;
  const double null2 = nulltol * nulltol;
  // orthogonality threshold on |b_p.b_q| / (|b_p||b_q|): the rounding level of an n-term dot product
  const double tol = std::max(1e-15, std::sqrt(static_cast<double>(n)) * 2.220446049250313e-16);
  int sweeps = 0;
  if (nn > 1) {
    for (; sweeps < 40; ++sweeps) {
      ADMM_HIP_TRY(hipMemsetAsync(rot, 0, sizeof(int32_t), stream));
      for (int32_t r = 0; r < ne - 1; ++r)
        hipLaunchKernelGGL(jacobi_round_kernel, dim3(static_cast<unsigned>(ne / 2)), dim3(kBlock), 0, stream, W, ldw, V,
                           ldv, nn, ne, r, tol, null2, rot);
      int32_t nrot = 0;
      ADMM_HIP_TRY(hipMemcpyAsync(&nrot, rot, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
      ADMM_HIP_TRY(hipStreamSynchronize(stream));
      if (nrot == 0) {
        ++sweeps;
        break;
      }
    }
  }
  hipLaunchKernelGGL(jacobi_norms_kernel, dim3(static_cast<unsigned>(nn)), dim3(kBlock), 0, stream, W, ldw, nn, lam_dev);
  ADMM_HIP_TRY(hipMemcpyAsync(lam_host->data(), lam_dev, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
  ADMM_HIP_TRY(hipStreamSynchronize(stream));
  if (sweeps_out) *sweeps_out = sweeps;
  return ADMM_OK;
}

void launch_scale_cols(double* V, int64_t ldv, int64_t n, const double* scale, hipStream_t stream) {
  hipLaunchKernelGGL(jacobi_scale_cols_kernel, dim3(static_cast<unsigned>(n)), dim3(kBlock), 0, stream, V, ldv,
                     static_cast<int32_t>(n), scale);
}

}  // namespace admm
