// dct_bench.hip -- developer microbenchmark (NOT part of libadmm_hip.so): where does the time of the LDS-resident
// DCT kernels (../dct.hip) go?  Variants on a 4096 x 4096 image, two columns per workgroup:
//   copy    load_pair + store_pair only (the Makhoul permutation through LDS): the memory floor of the structure
//   fft2    + forward and inverse FFT network, no spectral step
//   rows    the product kernel dct_rows_solve_kernel
//   fwd/inv the product column kernels
//   hipcc -O3 --offload-arch=gfx950 -I.. -I../../../include -o dct_bench dct_bench.hip && ./dct_bench
// MI355X, 4096 x 4096 (image MALL-resident): copy 35 us, fft2 75 us (20 us per FFT: ~50 % of the LDS peak with both
// resident workgroups of a CU in their transform phase), rows 99, fwd 54, inv 60, transpose 42 us.  The phases of a
// workgroup ADD UP (35 + 2*20 + 24 spectral): with 64 KB of LDS per column pair only two workgroups fit a CU and they
// run in lockstep.  Tried, no gain: twiddles preloaded into registers (cols_forward 65 -> 78 us); a persistent kernel
// prefetching the next pair into registers during the transforms (103 us) -- hipcc drains the prefetch at once (the
// loop-carried register arrays are copied after a vmcnt chain) and every twiddle / c4 / lam global load inside the
// transform phase is a vmcnt(0) that would drain it anyway; a real overlap needs all tables register- or LDS-resident
// and a two-register-set unrolled loop (~230 VGPRs).

#include "../dct.hip"

#include <cstdio>
#include <vector>

using namespace admm;

namespace admm {
int fail(int code, const std::string&) { return code; }
void set_error(const std::string&) {}
}  // namespace admm

#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t _e = (e);                                                        \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <int MODE>
__global__ __launch_bounds__(kBlock) void variant_kernel(double* __restrict__ tm, int64_t W, DctTables t) {
  extern __shared__ c64 zs[];
  const int n = t.n, p = t.log2n;
  double* a = tm + static_cast<int64_t>(2 * blockIdx.x) * W;
  double* b = a + W;
  load_pair(zs, a, b, n);
  __syncthreads();
  if (MODE >= 1) {
    fft_network<false>(zs, n, p, t.tw);
    fft_network<true>(zs, n, p, t.tw);
  }
  store_pair(zs, a, b, n, 1.0 / n);
}

int main() {
  const int64_t H = 4096, W = 4096, N = H * W;
  double* img = nullptr;
  CK(hipMalloc(&img, sizeof(double) * N));
  CK(hipMemset(img, 0, sizeof(double) * N));
  std::vector<admm_double2> tw(W / 2), c4(W / 2 + 1);
  std::vector<double> lam(W);
  dct_fill_tables(static_cast<int32_t>(W), tw.data(), c4.data(), lam.data());
  admm_double2 *dtw, *dc4;
  double* dlam;
  CK(hipMalloc(&dtw, sizeof(admm_double2) * tw.size()));
  CK(hipMalloc(&dc4, sizeof(admm_double2) * c4.size()));
  CK(hipMalloc(&dlam, sizeof(double) * W));
  CK(hipMemcpy(dtw, tw.data(), sizeof(admm_double2) * tw.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dc4, c4.data(), sizeof(admm_double2) * c4.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dlam, lam.data(), sizeof(double) * W, hipMemcpyHostToDevice));
  Ctrl* ctrl;
  CK(hipMalloc(&ctrl, sizeof(Ctrl)));
  CK(hipMemset(ctrl, 0, sizeof(Ctrl)));
  DctTables t{static_cast<int32_t>(W), 12, dtw, dc4, dlam};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t lds = sizeof(c64) * W;
  auto timeit = [&](const char* name, auto launch) {
    for (int k = 0; k < 3; ++k) launch();
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int k = 0; k < reps; ++k) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-8s %.1f us  (%.2f TB/s of 2 x 134 MB)\n", name, 1e3 * ms / reps, 2.0 * 8.0 * N / (ms / reps * 1e-3) / 1e12);
  };
  timeit("copy", [&] { hipLaunchKernelGGL(variant_kernel<0>, dim3(H / 2), dim3(kBlock), lds, 0, img, W, t); });
  timeit("fft2", [&] { hipLaunchKernelGGL(variant_kernel<1>, dim3(H / 2), dim3(kBlock), lds, 0, img, W, t); });
  timeit("rows", [&] { launch_dct_rows_solve(img, H, W, 1.0, t, t, ctrl, 0); });
  timeit("fwd", [&] { launch_dct_cols_forward(img, H, W, t, ctrl, 0); });
  timeit("inv", [&] { launch_dct_cols_inverse(img, img, H, W, t, ctrl, 0); });
  timeit("transp", [&] { launch_transpose(img, img + 0, H, W, ctrl, 0); });
  CK(hipDeviceSynchronize());
  return 0;
}
