// read_bw.hip -- developer microbenchmark: what read bandwidth does a plain streaming kernel reach on this
// part?  Gives the practical ceiling the GEMV / SYMV kernels are judged against (peak is 8 TB/s nominal).
//   hipcc -O3 --offload-arch=gfx950 -o read_bw read_bw.hip && ./read_bw [GiB]
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

// INTERLEAVED: consecutive blocks read consecutive 16*blockDim-byte slabs, grid-stride.
// U loads in flight per lane.  NT: nontemporal loads.
template <int U, bool NT>
__global__ void stream_interleaved(const double2_t* __restrict__ p, size_t n2, double* out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  double acc = 0.0;
  for (; i + (U - 1) * stride < n2; i += U * stride) {
    double2_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  for (; i < n2; i += stride) acc += p[i].x + p[i].y;
  if (acc == 1.2345e-300) out[0] = acc;
}

// CHUNKED: each block owns one contiguous chunk (like a GEMV block owning columns).
template <int U, bool NT>
__global__ void stream_chunked(const double2_t* __restrict__ p, size_t n2, double* out) {
  const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = (lo + per < n2) ? lo + per : n2;
  size_t i = lo + threadIdx.x;
  const size_t stride = blockDim.x;
  double acc = 0.0;
  for (; i + (U - 1) * stride < hi; i += U * stride) {
    double2_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  for (; i < hi; i += stride) acc += p[i].x + p[i].y;
  if (acc == 1.2345e-300) out[0] = acc;
}

template <typename K>
void run(const char* name, K kern, int blocks, int threads, const double2_t* p, size_t n2, double* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, p, n2, out);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, p, n2, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s blocks=%6d threads=%4d  %8.3f ms  %6.3f TB/s\n", name, blocks, threads, ms / reps,
         16.0 * n2 / (ms / reps) * 1e-9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  const size_t n2 = static_cast<size_t>(gib * (1ull << 30)) / 16;
  double2_t* p;
  double* out;
  CK(hipMalloc(&p, n2 * 16));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(p, 0, n2 * 16));
  printf("buffer %.2f GiB\n", gib);
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
    run("interleaved U=4", stream_interleaved<4, false>, blocks, 256, p, n2, out);
    run("interleaved U=8", stream_interleaved<8, false>, blocks, 256, p, n2, out);
    run("interleaved U=8 nt", stream_interleaved<8, true>, blocks, 256, p, n2, out);
    run("interleaved U=16", stream_interleaved<16, false>, blocks, 256, p, n2, out);
    run("chunked U=8", stream_chunked<8, false>, blocks, 256, p, n2, out);
    run("chunked U=8 nt", stream_chunked<8, true>, blocks, 256, p, n2, out);
  }
  run("interleaved U=8 t512", stream_interleaved<8, false>, 4096, 512, p, n2, out);
  run("interleaved U=8 t1024", stream_interleaved<8, false>, 2048, 1024, p, n2, out);
  run("interleaved U=4 t1024", stream_interleaved<4, false>, 2048, 1024, p, n2, out);
  run("interleaved U=8 t64", stream_interleaved<8, false>, 16384, 64, p, n2, out);
  return 0;
}
