// gemv_real.hip -- developer microbenchmark: the PRODUCT gemv kernels (gemv.hip included verbatim) under
// different plans, against the pure-read ceiling measured by gemv_pattern.hip.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemv_real gemv_real.hip && ./gemv_real [m] [n]
#include "../gemv.hip"

#include <cstdlib>
#include <vector>

namespace admm {
void set_error(const std::string&) {}
int fail(int code, const std::string&) { return code; }
}  // namespace admm

using namespace admm;

#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t _e = (e);                                                        \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

static hipEvent_t e0, e1;

template <typename F>
void timeit(const char* name, F launch, double bytes) {
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-64s %8.3f ms  %6.3f TB/s\n", name, ms / reps, bytes / (ms / reps) * 1e-9);
  fflush(stdout);
}

template <int U>
void run_n(const double* D, int64_t m, int64_t n, int64_t ld, const double* x, double* ypart, int chunks) {
  GemvNPlan p = gemv_n_plan(m, n, ld);
  p.cols_per_chunk = ceil_div(n, chunks);
  p.nchunk = static_cast<int32_t>(ceil_div(n, p.cols_per_chunk));
  dim3 grid(static_cast<unsigned>(ceil_div(ceil_div(p.m, 2), kBlock)), static_cast<unsigned>(p.nchunk));
  char name[128];
  snprintf(name, sizeof name, "gemv_n U=%2d chunks=%3d cols/chunk=%4lld grid=%u", U, p.nchunk, (long long)p.cols_per_chunk,
           grid.x * grid.y);
  timeit(name,
         [&] {
           hipLaunchKernelGGL((gemv_n_kernel<U, true>), grid, dim3(kBlock), 0, 0, D, p.m, p.n, p.ld, x, ypart, p.ldy,
                              p.cols_per_chunk, nullptr);
         },
         8.0 * m * n);
}

// NOTE: gemv_n_exp has NO tail loop: it streams floor(cols_per_chunk/U)*U columns of every chunk, so its
// TB/s must be scaled by that fraction (a chunk of 63 columns at U=16 reads only 48: the "7.9 TB/s" such a
// run prints is 6.0).  Findings (MI355X): x loads, accumulator count, unroll depth 8..24 and rows per
// thread all land within 2 % of the product kernel; the pure-read pattern ceiling is ~6 % above it.
// experimental variants of gemv_n: MODE 0 = product structure, 1 = no x loads (constant), 2 = x chunk staged
// in LDS, 3 = product structure with RPT double2 per thread per column (512*RPT rows per block)
template <int U, int RPT, int MODE, int NACC = U, int MAP = 0>
__global__ __launch_bounds__(kBlock) void gemv_n_exp(const double* __restrict__ D, int64_t m, int64_t n, int64_t ld,
                                                     const double* __restrict__ x, double* __restrict__ ypart,
                                                     int64_t ldy, int64_t cols_per_chunk) {
  extern __shared__ double sx[];
  int rb = blockIdx.x, cc = blockIdx.y;
  if (MAP == 1) {  // the 8 XCDs take workgroups round-robin: give each XCD a contiguous range of row blocks
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, k = lin >> 3;
    const int per = (gridDim.x + 7) / 8;
    rb = xcd * per + (k % per);
    cc = k / per;
    if (rb >= static_cast<int>(gridDim.x) || cc >= static_cast<int>(gridDim.y)) return;
  }
  const int64_t row = static_cast<int64_t>(rb) * (512 * RPT) + 2 * threadIdx.x;
  const int64_t j0 = static_cast<int64_t>(cc) * cols_per_chunk;
  const int64_t j1 = (j0 + cols_per_chunk < n) ? j0 + cols_per_chunk : n;
  if (MODE == 2) {
    for (int64_t j = j0 + threadIdx.x; j < j1; j += kBlock) sx[j - j0] = x[j];
    __syncthreads();
  }
  if (row + 512 * (RPT - 1) + 1 >= m) return;  // benchmark: full blocks only
  const double* p = D + row + j0 * ld;
  double2_t acc[NACC][RPT];
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int q = 0; q < RPT; ++q) acc[k][q] = double2_t{0.0, 0.0};
  int64_t j = j0;
  for (; j + U <= j1; j += U) {
    double2_t d[U][RPT];
#pragma unroll
    for (int k = 0; k < U; ++k)
#pragma unroll
      for (int q = 0; q < RPT; ++q) d[k][q] = load2<true>(p + k * ld + q * 512);
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const double xj = (MODE == 1) ? 1.0001 : ((MODE == 2) ? sx[j - j0 + k] : x[j + k]);
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        acc[k % NACC][q].x = __builtin_fma(d[k][q].x, xj, acc[k % NACC][q].x);
        acc[k % NACC][q].y = __builtin_fma(d[k][q].y, xj, acc[k % NACC][q].y);
      }
    }
    p += U * ld;
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    double2_t s = acc[0][q];
#pragma unroll
    for (int k = 1; k < NACC; ++k) {
      s.x += acc[k][q].x;
      s.y += acc[k][q].y;
    }
    *reinterpret_cast<double2_t*>(ypart + static_cast<int64_t>(cc) * ldy + row + q * 512) = s;
  }
}

template <int U, int RPT, int MODE, int NACC = U, int MAP = 0>
void run_exp(const double* D, int64_t m, int64_t n, int64_t ld, const double* x, double* ypart, int chunks) {
  const int64_t cpc = ceil_div(n, chunks);
  const int nch = static_cast<int>(ceil_div(n, cpc));
  dim3 grid(static_cast<unsigned>(ceil_div(m, 512 * RPT)), static_cast<unsigned>(nch));
  char name[128];
  snprintf(name, sizeof name, "gemv_n EXP U=%d RPT=%d mode=%d nacc=%d map=%d chunks=%3d grid=%u", U, RPT, MODE, NACC, MAP, nch, grid.x * grid.y);
  const size_t lds = MODE == 2 ? static_cast<size_t>(cpc) * 8 : 0;
  timeit(name,
         [&] {
           hipLaunchKernelGGL((gemv_n_exp<U, RPT, MODE, NACC, MAP>), grid, dim3(kBlock), lds, 0, D, m, n, ld, x, ypart,
                              round_up(m, 2), cpc);
         },
         8.0 * m * n);
}

template <int NR>
void run_t(const double* D, int64_t m, int64_t n, int64_t ld, const double* v, double* gpart, int rc) {
  GemvTPlan p = gemv_t_plan(m, n, ld);
  p.rows_per_chunk = rc;
  p.nchunk = static_cast<int32_t>(ceil_div(m, rc));
  dim3 grid(static_cast<unsigned>(ceil_div(p.n, kTCols)), static_cast<unsigned>(p.nchunk));
  const size_t lds = static_cast<size_t>(NR) * p.rows_per_chunk * sizeof(double);
  char name[128];
  snprintf(name, sizeof name, "gemv_t nrhs=%d rows/chunk=%5d chunks=%3d grid=%u lds=%zu", NR, rc, p.nchunk, grid.x * grid.y, lds);
  timeit(name,
         [&] {
           hipLaunchKernelGGL((gemv_t_kernel<NR, true>), grid, dim3(kBlock), lds, 0, D, p.m, p.n, p.ld, v, v + m, v + 2 * m,
                              gpart, p.ldg, p.rows_per_chunk, nullptr);
         },
         8.0 * m * n);
}

int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 100000;
  const int64_t n = argc > 2 ? atoll(argv[2]) : 10000;
  const int64_t ld = argc > 3 ? atoll(argv[3]) : round_up(m, 16);
  double *D, *x, *ypart, *v, *gpart;
  CK(hipMalloc(&D, sizeof(double) * ld * n));
  CK(hipMalloc(&x, sizeof(double) * n));
  CK(hipMalloc(&v, sizeof(double) * 3 * m));
  CK(hipMalloc(&ypart, sizeof(double) * round_up(m, 2) * 256));
  CK(hipMalloc(&gpart, sizeof(double) * round_up(n, 2) * 3 * 512));
  CK(hipMemset(D, 0, sizeof(double) * ld * n));
  CK(hipMemset(x, 0, sizeof(double) * n));
  CK(hipMemset(v, 0, sizeof(double) * 3 * m));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("m=%lld n=%lld ld=%lld  %.2f GB\n", (long long)m, (long long)n, (long long)ld, 8.0 * m * n * 1e-9);
  {
    GemvNPlan p = gemv_n_plan(m, n, ld);
    printf("default gemv_n plan: chunks=%d cols/chunk=%lld\n", p.nchunk, (long long)p.cols_per_chunk);
    GemvTPlan q = gemv_t_plan(m, n, ld);
    printf("default gemv_t plan: rows/chunk=%d chunks=%d\n", q.rows_per_chunk, q.nchunk);
  }
  for (int chunks : {21, 32}) run_n<8>(D, m, n, ld, x, ypart, chunks);
  // cols_per_chunk multiples of 8 so that the tail-less EXP kernel streams everything: 10000/25 = 400, /50 = 200, /125 = 80
  for (int chunks : {25, 50, 125}) run_exp<8, 1, 0, 8, 0>(D, m, n, ld, x, ypart, chunks);
  for (int chunks : {25, 50, 125}) run_exp<8, 1, 0, 8, 1>(D, m, n, ld, x, ypart, chunks);
  for (int chunks : {25, 50, 125}) run_exp<8, 1, 0, 2, 1>(D, m, n, ld, x, ypart, chunks);
  for (int rc : {2048, 4096}) run_t<1>(D, m, n, ld, v, gpart, rc);
  for (int rc : {1024, 2048}) run_t<3>(D, m, n, ld, v, gpart, rc);
  return 0;
}
