// gemv_pattern.hip -- developer microbenchmark: read bandwidth of the column-major GEMV access
// patterns (no arithmetic beyond one add per element), to choose block shapes for gemv.hip.
//   hipcc -O3 --offload-arch=gfx950 -o gemv_pattern gemv_pattern.hip && ./gemv_pattern [m] [n]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef double double2_t __attribute__((ext_vector_type(2)));

#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t _e = (e);                                                        \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// N-pattern: block = 256 threads owning RPT*512 rows (thread: RPT double2 at stride 512 rows), loops
// over the columns of its chunk, U columns in flight.  MAP: 0 = blockIdx.x -> row block (fastest),
// 1 = XCD-contiguous (the 8 XCDs take blockIdx round-robin: give each XCD its own row range).
template <int RPT, int U, int MAP>
__global__ __launch_bounds__(256) void pat_n(const double* __restrict__ D, int64_t m, int64_t n, int64_t ld,
                                             int64_t cols_per_chunk, int nrb, double* out) {
  int rb = blockIdx.x, cc = blockIdx.y;
  if (MAP == 1) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, k = lin >> 3;
    const int per = nrb / 8;  // nrb is a multiple of 8 in this benchmark
    rb = xcd * per + (k % per);
    cc = k / per;
  }
  const int64_t row = static_cast<int64_t>(rb) * (512 * RPT) + 2 * threadIdx.x;
  const int64_t j0 = static_cast<int64_t>(cc) * cols_per_chunk;
  const int64_t j1 = (j0 + cols_per_chunk < n) ? j0 + cols_per_chunk : n;
  if (row >= m) return;
  const double* p = D + row + j0 * ld;
  double acc = 0.0;
  for (int64_t j = j0; j + U <= j1; j += U) {
    double2_t d[U][RPT];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < RPT; ++q)
        d[u][q] = __builtin_nontemporal_load(reinterpret_cast<const double2_t*>(p + u * ld + q * 512));
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < RPT; ++q) acc += d[u][q].x + d[u][q].y;
    p += U * ld;
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

// T-pattern: block = 256 threads (4 waves) on CT columns x R rows; each wave streams CW columns at a
// time, S row steps (1 KiB each) in flight per column.
template <int CW, int S>
__global__ __launch_bounds__(256) void pat_t(const double* __restrict__ D, int64_t m, int64_t n, int64_t ld, int CT,
                                             int R, double* out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * R;
  const int64_t jt = static_cast<int64_t>(blockIdx.x) * CT;
  double acc = 0.0;
  for (int pass = 0; pass < CT / (4 * CW); ++pass) {
    const int64_t jb = jt + (pass * 4 + wid) * CW;
    if (jb >= n) break;
    const double* col = D + r0 + jb * ld + 2 * lane;
    for (int p = 0; p + 128 * S <= R; p += 128 * S) {
      double2_t d[S][CW];
#pragma unroll
      for (int k = 0; k < S; ++k)
#pragma unroll
        for (int c = 0; c < CW; ++c)
          d[k][c] = __builtin_nontemporal_load(reinterpret_cast<const double2_t*>(col + c * ld + p + 128 * k));
#pragma unroll
      for (int k = 0; k < S; ++k)
#pragma unroll
        for (int c = 0; c < CW; ++c) acc += d[k][c].x + d[k][c].y;
    }
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

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
  printf("%-56s %8.3f ms  %6.3f TB/s\n", name, ms / reps, bytes / (ms / reps) * 1e-9);
  fflush(stdout);
}

template <int RPT, int U, int MAP>
void run_n(const double* D, int64_t m, int64_t n, int64_t ld, int chunks, double* out) {
  const int nrb = static_cast<int>(m / (512 * RPT));
  const int64_t cpc = (n + chunks - 1) / chunks;
  char name[128];
  snprintf(name, sizeof name, "N rows/blk=%5d U=%2d map=%d chunks=%3d (grid %d)", 512 * RPT, U, MAP, chunks, nrb * chunks);
  timeit(name, [&] { hipLaunchKernelGGL((pat_n<RPT, U, MAP>), dim3(nrb, chunks), dim3(256), 0, 0, D, m, n, ld, cpc, nrb, out); },
         8.0 * nrb * 512 * RPT * n);
}

template <int CW, int S>
void run_t(const double* D, int64_t m, int64_t n, int64_t ld, int CT, int R, double* out) {
  char name[128];
  const int gx = static_cast<int>(n / CT), gy = static_cast<int>(m / R);
  snprintf(name, sizeof name, "T cols/blk=%3d rows/blk=%5d CW=%d S=%d (grid %d)", CT, R, CW, S, gx * gy);
  timeit(name, [&] { hipLaunchKernelGGL((pat_t<CW, S>), dim3(gx, gy), dim3(256), 0, 0, D, m, n, ld, CT, R, out); },
         8.0 * gy * R * gx * CT);
}

int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 98304;  // multiple of 8*4096
  const int64_t n = argc > 2 ? atoll(argv[2]) : 10240;
  const int64_t ld = m;
  double *D, *out;
  CK(hipMalloc(&D, sizeof(double) * ld * n));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(D, 0, sizeof(double) * ld * n));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("m=%lld n=%lld  %.2f GB\n", (long long)m, (long long)n, 8.0 * m * n * 1e-9);
  run_n<1, 8, 0>(D, m, n, ld, 22, out);
  run_n<1, 8, 0>(D, m, n, ld, 40, out);
  run_n<1, 8, 0>(D, m, n, ld, 80, out);
  run_n<1, 16, 0>(D, m, n, ld, 22, out);
  run_n<1, 8, 1>(D, m, n, ld, 22, out);
  run_n<1, 8, 1>(D, m, n, ld, 80, out);
  run_n<2, 4, 0>(D, m, n, ld, 40, out);
  run_n<2, 8, 0>(D, m, n, ld, 40, out);
  run_n<2, 4, 1>(D, m, n, ld, 40, out);
  run_n<4, 2, 0>(D, m, n, ld, 80, out);
  run_n<4, 4, 0>(D, m, n, ld, 80, out);
  run_n<4, 4, 1>(D, m, n, ld, 80, out);
  run_n<8, 2, 0>(D, m, n, ld, 160, out);
  run_n<8, 2, 1>(D, m, n, ld, 160, out);
  run_t<4, 4>(D, m, n, ld, 32, 2048, out);
  run_t<4, 4>(D, m, n, ld, 32, 4096, out);
  run_t<4, 4>(D, m, n, ld, 16, 4096, out);
  run_t<4, 4>(D, m, n, ld, 16, 8192, out);
  run_t<2, 8>(D, m, n, ld, 32, 2048, out);
  run_t<2, 8>(D, m, n, ld, 16, 4096, out);
  run_t<2, 8>(D, m, n, ld, 8, 8192, out);
  run_t<1, 16>(D, m, n, ld, 16, 4096, out);
  run_t<1, 16>(D, m, n, ld, 4, 16384, out);
  run_t<1, 8>(D, m, n, ld, 4, 16384, out);
  run_t<8, 2>(D, m, n, ld, 32, 2048, out);
  return 0;
}
