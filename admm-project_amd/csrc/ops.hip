// ops.hip -- stand-alone operator entry points of the C ABI (HOST pointers in, HOST out).
// Thin wrappers: upload -> the same kernels the loop uses -> download.  They exist so the
// parity tests can pin every kernel against the oracle on its own, and for bindings that
// keep a user prox on the host but want the heavy linear algebra on the device.
#include <vector>

#include "kernels.h"
#include "loop_kernels.h"

using namespace admm;

namespace {

struct Scratch {  // frees everything on scope exit
  std::vector<void*> ptrs;
  ~Scratch() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  int alloc(double** out, size_t elems) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (elems ? elems : 1) * sizeof(double));
    if (e != hipSuccess) return fail(ADMM_E_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    ptrs.push_back(p);
    *out = static_cast<double*>(p);
    return ADMM_OK;
  }
};

int need_device() {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
    return fail(ADMM_E_DEVICE, "no HIP device visible: the ADMM engine has no CPU fallback");
  return ADMM_OK;
}

int put_matrix(Scratch& sc, double** dst, int64_t* ld, const double* src, int64_t rows, int64_t cols, int64_t ld_src) {
  *ld = round_up(rows, 16);
  ADMM_TRY(sc.alloc(dst, static_cast<size_t>(*ld) * cols));
  ADMM_HIP_TRY(hipMemset(*dst, 0, sizeof(double) * (*ld) * cols));
  ADMM_HIP_TRY(hipMemcpy2D(*dst, (*ld) * sizeof(double), src, ld_src * sizeof(double), rows * sizeof(double), cols,
                           hipMemcpyHostToDevice));
  return ADMM_OK;
}

}  // namespace

extern "C" {

int admm_op_gemv_n(const double* D, int64_t m, int64_t n, int64_t ldD, const double* x, double* y) {
  if (!D || !x || !y || m <= 0 || n <= 0 || ldD < m) return fail(ADMM_E_INVALID, "gemv_n: bad argument");
  ADMM_TRY(need_device());
  Scratch sc;
  double *dD, *dx, *dpart, *dy;
  int64_t ld;
  ADMM_TRY(put_matrix(sc, &dD, &ld, D, m, n, ldD));
  ADMM_TRY(sc.alloc(&dx, n));
  ADMM_HIP_TRY(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
  GemvNPlan p = gemv_n_plan(m, n, ld);
  ADMM_TRY(sc.alloc(&dpart, p.part_elems()));
  ADMM_TRY(sc.alloc(&dy, round_up(m, 2)));
  launch_gemv_n(p, dD, dx, dpart, nullptr, nullptr);
  launch_sum_partials(dpart, p.nchunk, p.ldy, m, dy, nullptr, nullptr);
  ADMM_HIP_TRY(hipDeviceSynchronize());
  ADMM_HIP_TRY(hipMemcpy(y, dy, sizeof(double) * m, hipMemcpyDeviceToHost));
  return ADMM_OK;
}

int admm_op_gemv_t(const double* D, int64_t m, int64_t n, int64_t ldD, const double* V, int64_t ldV, int32_t nrhs,
                   double* G, int64_t ldG) {
  if (!D || !V || !G || m <= 0 || n <= 0 || ldD < m || ldV < m || ldG < n || nrhs < 1 || nrhs > 3)
    return fail(ADMM_E_INVALID, "gemv_t: bad argument (1 <= nrhs <= 3)");
  ADMM_TRY(need_device());
  Scratch sc;
  double *dD, *dV, *dpart, *dG;
  int64_t ld, ldv;
  ADMM_TRY(put_matrix(sc, &dD, &ld, D, m, n, ldD));
  ADMM_TRY(put_matrix(sc, &dV, &ldv, V, m, nrhs, ldV));
  GemvTPlan p = gemv_t_plan(m, n, ld);
  ADMM_TRY(sc.alloc(&dpart, p.part_elems(nrhs)));
  ADMM_TRY(sc.alloc(&dG, static_cast<size_t>(p.ldg) * nrhs));
  launch_gemv_t(p, dD, dV, nrhs > 1 ? dV + ldv : nullptr, nrhs > 2 ? dV + 2 * ldv : nullptr, nrhs, dpart, nullptr,
                nullptr);
  launch_sum_partials_t(p, dpart, nrhs, dG, p.ldg, nullptr, nullptr);
  ADMM_HIP_TRY(hipDeviceSynchronize());
  ADMM_HIP_TRY(hipMemcpy2D(G, ldG * sizeof(double), dG, p.ldg * sizeof(double), n * sizeof(double), nrhs,
                           hipMemcpyDeviceToHost));
  return ADMM_OK;
}

int admm_op_gram(const double* D, int64_t m, int64_t n, int64_t ldD, double shift, double* W) {
  if (!D || !W || m <= 0 || n <= 0 || ldD < m) return fail(ADMM_E_INVALID, "gram: bad argument");
  ADMM_TRY(need_device());
  Scratch sc;
  double *dD, *dW;
  int64_t ld;
  ADMM_TRY(put_matrix(sc, &dD, &ld, D, m, n, ldD));
  const int64_t ldw = round_up(n, 16);
  ADMM_TRY(sc.alloc(&dW, static_cast<size_t>(ldw) * n));
  ADMM_HIP_TRY(hipMemset(dW, 0, sizeof(double) * ldw * n));
  launch_gemm(1, 0, n, n, m, 1.0, dD, ld, dD, ld, 0.0, dW, ldw, true, nullptr);
  if (shift != 0.0) launch_add_diag(dW, n, ldw, shift, nullptr);
  launch_symmetrize_lower(dW, n, ldw, nullptr);
  ADMM_HIP_TRY(hipDeviceSynchronize());
  ADMM_HIP_TRY(hipMemcpy2D(W, n * sizeof(double), dW, ldw * sizeof(double), n * sizeof(double), n,
                           hipMemcpyDeviceToHost));
  return ADMM_OK;
}

int admm_op_cholesky(double* A, int64_t n, int64_t ldA) {
  if (!A || n <= 0 || ldA < n) return fail(ADMM_E_INVALID, "cholesky: bad argument");
  ADMM_TRY(need_device());
  Scratch sc;
  double *dA, *dinfo;
  int64_t ld;
  ADMM_TRY(put_matrix(sc, &dA, &ld, A, n, n, ldA));
  ADMM_TRY(sc.alloc(&dinfo, 1));
  ADMM_TRY(cholesky_lower(dA, n, ld, reinterpret_cast<int32_t*>(dinfo), nullptr, nullptr));
  ADMM_HIP_TRY(hipDeviceSynchronize());
  int32_t info = 0;
  ADMM_HIP_TRY(hipMemcpy(&info, dinfo, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (info != 0)
    return fail(ADMM_E_NUMERIC, "Cholesky failed: matrix must be positive definite (pivot " + std::to_string(info) + ")");
  ADMM_HIP_TRY(hipMemcpy2D(A, ldA * sizeof(double), dA, ld * sizeof(double), n * sizeof(double), n,
                           hipMemcpyDeviceToHost));
  for (int64_t j = 1; j < n; ++j)
    for (int64_t i = 0; i < j; ++i) A[i + j * ldA] = 0.0;  // chol(.,'lower') returns a lower-triangular matrix
  return ADMM_OK;
}

int admm_op_trsv_pair(const double* L, int64_t n, int64_t ldL, const double* y, double* x) {
  if (!L || !y || !x || n <= 0 || ldL < n) return fail(ADMM_E_INVALID, "trsv_pair: bad argument");
  ADMM_TRY(need_device());
  Scratch sc;
  double *dL, *dy, *dx, *dinv, *buf;
  int64_t ld;
  ADMM_TRY(put_matrix(sc, &dL, &ld, L, n, n, ldL));
  ADMM_TRY(sc.alloc(&dy, round_up(n, 2)));
  ADMM_TRY(sc.alloc(&dx, round_up(n, 2)));
  ADMM_HIP_TRY(hipMemcpy(dy, y, sizeof(double) * n, hipMemcpyHostToDevice));
  TrsvPlan plan{};
  ADMM_TRY(sc.alloc(&dinv, static_cast<size_t>(ceil_div(n, 64)) * 64 * 64));
  launch_trtri_diag(dL, n, ld, dinv, nullptr);
  const int form = trsv_resolve_form(n, kTrsvBlocked);
  ADMM_TRY(sc.alloc(&buf, trsv_plan_elems(n, form)));
  ADMM_TRY(trsv_build(dL, n, ld, dinv, buf, &plan, nullptr, form));
  launch_trsv_pair(plan, dy, dx, nullptr, nullptr);
  ADMM_TRY(trsv_check_error(plan, nullptr));
  ADMM_HIP_TRY(hipDeviceSynchronize());
  ADMM_HIP_TRY(hipMemcpy(x, dx, sizeof(double) * n, hipMemcpyDeviceToHost));
  return ADMM_OK;
}

__global__ void soft_threshold_kernel(const double* __restrict__ v, int64_t n, double t, double* __restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const double a = fabs(v[i]) - t;
    const double p = a > 0.0 ? a : 0.0;
    out[i] = (v[i] > 0.0) ? p : ((v[i] < 0.0) ? -p : 0.0);
  }
}

int admm_op_soft_threshold(const double* v, int64_t n, double t, double* out) {
  if (!v || !out || n <= 0) return fail(ADMM_E_INVALID, "soft_threshold: bad argument");
  ADMM_TRY(need_device());
  Scratch sc;
  double *dv, *dout;
  ADMM_TRY(sc.alloc(&dv, n));
  ADMM_TRY(sc.alloc(&dout, n));
  ADMM_HIP_TRY(hipMemcpy(dv, v, sizeof(double) * n, hipMemcpyHostToDevice));
  int64_t blocks = ceil_div(n, kBlock);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(soft_threshold_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, nullptr, dv, n, t,
                     dout);
  ADMM_HIP_TRY(hipDeviceSynchronize());
  ADMM_HIP_TRY(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
  return ADMM_OK;
}

}  // extern "C"
