// gemm_bench.hip -- developer microbenchmark of the fp64 MFMA GEMM in dense.hip (included verbatim):
// the Gram build D'D (TN, lower tiles only), the Cholesky trailing update (NT) and the inverse product (TT).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemm_bench gemm_bench.hip && ./gemm_bench [m] [n]
#include "../dense.hip"

#include <cstdlib>

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

__global__ void fill(double* p, size_t n) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    p[i] = static_cast<double>((i * 2654435761u) % 2001) / 1000.0 - 1.0;
}

int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 100000, n = argc > 2 ? atoll(argv[2]) : 10000;
  const int64_t ld = round_up(m, 512), ldw = round_up(n, 16);
  double *D, *W;
  CK(hipMalloc(&D, sizeof(double) * ld * n));
  CK(hipMalloc(&W, sizeof(double) * ldw * n));
  fill<<<4096, 256>>>(D, static_cast<size_t>(ld) * n);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double flops, auto fn) {
    fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    fn();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %9.3f ms  %6.2f TFLOP/s\n", name, ms, flops / ms * 1e-9);
    fflush(stdout);
  };
  const double nt = static_cast<double>((n + 127) / 128);
  const double tiles_lower = nt * (nt + 1) / 2;
  timeit("gram TN lower  D'D  (n x n, K = m)", tiles_lower * 128.0 * 128.0 * 2.0 * m,
         [&] { launch_gemm(1, 0, n, n, m, 1.0, D, ld, D, ld, 0.0, W, ldw, true, 0); });
  timeit("TN full  n x n, K = n", 2.0 * n * n * n,
         [&] { launch_gemm(1, 0, n, n, n, 1.0, D, ld, D, ld, 0.0, W, ldw, false, 0); });
  timeit("NT lower (trailing update) n x n, K = 64", tiles_lower * 128.0 * 128.0 * 2.0 * 64,
         [&] { launch_gemm(0, 1, n, n, 64, -1.0, D, ld, D, ld, 1.0, W, ldw, true, 0); });
  timeit("NN full  n x n, K = n", 2.0 * n * n * n,
         [&] { launch_gemm(0, 0, n, n, n, 1.0, D, ld, D, ld, 0.0, W, ldw, false, 0); });
  return 0;
}
