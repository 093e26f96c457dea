// symv_bench.hip -- developer microbenchmark (NOT part of libadmm_hip.so): sweeps tile shapes, pipeline
// depths and tile orders of the lower-triangle SYMV to find what limits symv.hip.  Self-contained.
//   hipcc -O3 --offload-arch=gfx950 -o symv_bench symv_bench.hip && ./symv_bench [n]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double2_t __attribute__((ext_vector_type(2)));

#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t _e = (e);                                                        \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// reduce-scatter of P per-lane values over 64 lanes; returns the wave sum of element *col.
template <int P>
__device__ __forceinline__ double reduce_scatter(double (&t)[P], int lane, int* col) {
  int c = 0;
  int mask = 32;
#pragma unroll
  for (int half = P / 2; half >= 1; half /= 2) {
    const bool b = lane & mask;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      const double keep = b ? t[k + half] : t[k];
      const double send = b ? t[k] : t[k + half];
      t[k] = keep + __shfl_xor(send, mask, 64);
    }
    c += b ? half : 0;
    mask >>= 1;
  }
  double r = t[0];
#pragma unroll
  for (; mask >= 1; mask >>= 1) r += __shfl_xor(r, mask, 64);
  *col = c;
  return r;
}


// ---- gfx950 cross-lane primitives: v_permlane{32,16}_swap + DPP (no LDS crossbar) -------------
__device__ __forceinline__ void swap32(double& a, double& b) {  // a.hi <-> b.lo (32-lane halves)
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  a = __hiloint2double(r1[0], r0[0]);
  b = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b) {  // odd rows of a <-> even rows of b
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  a = __hiloint2double(r1[0], r0[0]);
  b = __hiloint2double(r1[1], r0[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// P in {4, 8}: same contract as reduce_scatter (result valid in lanes with (lane & (64/P-1)) == 0)
template <int P>
__device__ __forceinline__ double reduce_scatter_fast(double (&t)[P], int lane, int* col) {
#pragma unroll
  for (int k = 0; k < P / 2; ++k) {
    swap32(t[k], t[k + P / 2]);
    t[k] += t[k + P / 2];
  }
#pragma unroll
  for (int k = 0; k < P / 4; ++k) {
    swap16(t[k], t[k + P / 4]);
    t[k] += t[k + P / 4];
  }
  int c = ((lane & 32) ? P / 2 : 0) + ((lane & 16) ? P / 4 : 0);
  double r;
  if (P == 8) {
    const bool b3 = lane & 8;
    const double keep = b3 ? t[1] : t[0];
    const double send = b3 ? t[0] : t[1];
    r = keep + dpp_mov<0x128>(send);  // row_ror:8
    c += b3 ? 1 : 0;
  } else {
    r = t[0];
    r += dpp_mov<0x128>(r);
  }
  r += dpp_mov<0x124>(r);  // row_ror:4
  r += dpp_mov<0x4E>(r);   // quad_perm [2,3,0,1]
  r += dpp_mov<0xB1>(r);   // quad_perm [1,0,3,2]
  *col = c;
  return r;
}

// Q: double2 per column per lane (rows per wave = 128*Q), P: columns per panel, NB: panel buffers,
// C: columns per tile, MODE 0 full, 1 no cross-lane reduction, 2 loads + one add only.
template <int Q, int P, int NB, int C, int MODE, bool DIAG, bool NT>
__device__ __forceinline__ void tile_body(const double* __restrict__ M, int64_t ld, const double* __restrict__ x,
                                          int64_t w0, int64_t c0, int64_t limit, double* __restrict__ nout,
                                          double* __restrict__ tout, int lane) {
  double xr[Q][2], nacc[Q][2];
  int roff[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    roff[q] = static_cast<int>(w0) + q * 128 + 2 * lane;
    xr[q][0] = x[roff[q]];
    xr[q][1] = x[roff[q] + 1];
    nacc[q][0] = nacc[q][1] = 0.0;
  }
  double2_t buf[NB][P][Q];
  const int npan = static_cast<int>((limit - c0) / P);
  double junk = 0.0;

  auto load = [&](double2_t(&b)[P][Q], int pan) {
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const double* cb = M + (c0 + static_cast<int64_t>(pan) * P + k) * ld;
#pragma unroll
      for (int q = 0; q < Q; ++q)
        b[k][q] = NT ? __builtin_nontemporal_load(reinterpret_cast<const double2_t*>(cb + roff[q]))
                     : *reinterpret_cast<const double2_t*>(cb + roff[q]);
    }
  };
  auto compute = [&](double2_t(&b)[P][Q], int pan) {
    const int64_t cp = c0 + static_cast<int64_t>(pan) * P;
    double tacc[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int64_t j = cp + k;
      const double xj = x[j];
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        double a0 = b[k][q].x, a1 = b[k][q].y;
        if (MODE == 2) {
          junk += a0 + a1;
          continue;
        }
        double t0 = a0, t1 = a1;
        if (DIAG) {
          const int64_t r = roff[q];
          t0 = (r > j) ? a0 : 0.0;
          t1 = (r + 1 > j) ? a1 : 0.0;
          a0 = (j <= r) ? a0 : 0.0;
          a1 = (j <= r + 1) ? a1 : 0.0;
        }
        t = __builtin_fma(t0, xr[q][0], t);
        t = __builtin_fma(t1, xr[q][1], t);
        nacc[q][0] = __builtin_fma(a0, xj, nacc[q][0]);
        nacc[q][1] = __builtin_fma(a1, xj, nacc[q][1]);
      }
      tacc[k] = t;
    }
    if (MODE == 0) {
      int col;
      const double sum = reduce_scatter<P>(tacc, lane, &col);
      if ((lane & (64 / P - 1)) == 0) tout[cp + col] = sum;
    } else if (MODE == 3) {
      int col;
      const double sum = reduce_scatter_fast<P>(tacc, lane, &col);
      if ((lane & (64 / P - 1)) == 0) tout[cp + col] = sum;
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < P; ++k) junk += tacc[k];
    }
  };

#pragma unroll
  for (int b = 0; b < NB - 1; ++b)
    if (b < npan) load(buf[b], b);
#pragma unroll 1
  for (int p = 0; p < npan; p += NB) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int pp = p + b;
      if (pp < npan) {
        const int nxt = pp + NB - 1;
        if (nxt < npan) load(buf[(b + NB - 1) % NB], nxt);
        compute(buf[b], pp);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (MODE == 1 || MODE == 2) nacc[q][0] += junk;
    *reinterpret_cast<double2_t*>(nout + roff[q]) = double2_t{nacc[q][0], nacc[q][1]};
  }
}

template <int Q, int P, int NB, int C, int MODE, bool NT>
__global__ __launch_bounds__(64) void symv_var(const double* __restrict__ M, int64_t ld, const double* __restrict__ x,
                                               double* __restrict__ npart, double* __restrict__ tpart, int64_t ldp,
                                               const int2* __restrict__ tiles) {
  constexpr int R = 128 * Q;
  const int2 tl = tiles[blockIdx.x];
  const int64_t w0 = static_cast<int64_t>(tl.x) * R, c0 = static_cast<int64_t>(tl.y) * C;
  const int lane = threadIdx.x;
  const int64_t limit = (c0 + C < w0 + R) ? c0 + C : w0 + R;
  const bool diag = w0 < c0 + C;
  double* nout = npart + static_cast<int64_t>(tl.y) * ldp;
  double* tout = tpart + static_cast<int64_t>(tl.x) * ldp;
  if (diag) tile_body<Q, P, NB, C, MODE, true, NT>(M, ld, x, w0, c0, limit, nout, tout, lane);
  else tile_body<Q, P, NB, C, MODE, false, NT>(M, ld, x, w0, c0, limit, nout, tout, lane);
}

struct Result {
  double us;
  double err;
};

enum Order { COLMAJOR = 0, ROWMAJOR = 1, ALTERNATE = 2 };

template <int Q, int P, int NB, int C, int MODE, bool NT = true>
Result run_variant(const char* name, int64_t n, const double* dM, const double* dx, const std::vector<double>& yref,
                   int order, int reps) {
  constexpr int R = 128 * Q;
  const int nrow = static_cast<int>(n / R), ncol = static_cast<int>(n / C);
  std::vector<int2> tiles;
  if (order == ROWMAJOR) {
    for (int r = 0; r < nrow; ++r)
      for (int c = 0; c < ncol; ++c)
        if (static_cast<int64_t>(r) * R + R - 1 >= static_cast<int64_t>(c) * C) tiles.push_back(int2{r, c});
  } else {
    for (int c = 0; c < ncol; ++c)
      for (int r = 0; r < nrow; ++r)
        if (static_cast<int64_t>(r) * R + R - 1 >= static_cast<int64_t>(c) * C) tiles.push_back(int2{r, c});
  }
  std::vector<int2> rev(tiles.rbegin(), tiles.rend());
  int2 *dt, *dtr;
  CK(hipMalloc(&dt, tiles.size() * sizeof(int2)));
  CK(hipMalloc(&dtr, tiles.size() * sizeof(int2)));
  CK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
  CK(hipMemcpy(dtr, rev.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
  const int64_t ldp = n;
  double *np, *tp;
  CK(hipMalloc(&np, sizeof(double) * ldp * ncol));
  CK(hipMalloc(&tp, sizeof(double) * ldp * nrow));
  CK(hipMemset(np, 0, sizeof(double) * ldp * ncol));
  CK(hipMemset(tp, 0, sizeof(double) * ldp * nrow));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&](int it) {
    const int2* t = (order == ALTERNATE && (it & 1)) ? dtr : dt;
    hipLaunchKernelGGL((symv_var<Q, P, NB, C, MODE, NT>), dim3(static_cast<unsigned>(tiles.size())), dim3(64), 0, 0, dM, n,
                       dx, np, tp, ldp, t);
  };
  for (int i = 0; i < 4; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  Result res{ms * 1000.0 / reps, 0.0};
  if (MODE == 0 || MODE == 3) {
    std::vector<double> hn(ldp * ncol), ht(ldp * nrow);
    CK(hipMemcpy(hn.data(), np, sizeof(double) * hn.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht.data(), tp, sizeof(double) * ht.size(), hipMemcpyDeviceToHost));
    double emax = 0, ymax = 0;
    for (int64_t i = 0; i < n; ++i) {
      double s = 0;
      for (int c = 0; c < ncol; ++c) s += hn[c * ldp + i];
      for (int r = 0; r < nrow; ++r) s += ht[r * ldp + i];
      emax = std::max(emax, std::fabs(s - yref[i]));
      ymax = std::max(ymax, std::fabs(yref[i]));
    }
    res.err = emax / ymax;
  }
  const double bytes = 8.0 * n * (n + 1) / 2;
  int nregs = 0;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(symv_var<Q, P, NB, C, MODE, NT>)) == hipSuccess) nregs = fa.numRegs;
  printf("%-34s Q=%d P=%2d NB=%d C=%4d mode=%d order=%d tiles=%6zu vgpr=%3d  %8.2f us  %6.3f TB/s  err=%.2e\n", name, Q,
         P, NB, C, MODE, order, tiles.size(), nregs, res.us, bytes / res.us * 1e-6, res.err);
  fflush(stdout);
  CK(hipFree(dt));
  CK(hipFree(dtr));
  CK(hipFree(np));
  CK(hipFree(tp));
  return res;
}

__global__ void fill_sym(double* M, int64_t n, int64_t ld) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (i >= n) return;
  const int64_t a = i > j ? i : j, b = i > j ? j : i;
  uint64_t h = static_cast<uint64_t>(a) * 1000003ull + static_cast<uint64_t>(b) * 7919ull + 12345ull;
  h ^= h >> 13;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  M[j * ld + i] = static_cast<double>(h % 2001) / 1000.0 - 1.0;
}

__global__ void ref_gemv(const double* M, int64_t n, int64_t ld, const double* x, double* y) {
  const int64_t j = blockIdx.x;  // y[j] = column j dot x (M symmetric)
  double s = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += M[j * ld + i] * x[i];
  __shared__ double sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) y[j] = sh[0];
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10240;  // multiple of 512
  const int reps = argc > 2 ? atoi(argv[2]) : 40;
  const int64_t ld = n;
  double *dM, *dx, *dy;
  CK(hipMalloc(&dM, sizeof(double) * ld * n));
  CK(hipMalloc(&dx, sizeof(double) * n));
  CK(hipMalloc(&dy, sizeof(double) * n));
  fill_sym<<<dim3(static_cast<unsigned>((n + 255) / 256), static_cast<unsigned>(n)), 256>>>(dM, n, ld);
  std::vector<double> hx(n), yref(n);
  for (int64_t i = 0; i < n; ++i) hx[i] = std::sin(0.37 * i) + 0.1;
  CK(hipMemcpy(dx, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  ref_gemv<<<static_cast<unsigned>(n), 256>>>(dM, n, ld, dx, dy);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(yref.data(), dy, sizeof(double) * n, hipMemcpyDeviceToHost));
  printf("n=%lld  lower triangle %.1f MB\n", static_cast<long long>(n), 8.0 * n * (n + 1) / 2 * 1e-6);

  run_variant<1, 4, 2, 128, 3, false>("128x128 p4 nb2 fast, plain loads", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 4, 2, 128, 3>("128x128 p4 nb2 fast NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 4, 2, 128, 2>("128x128 p4 nb2 loads only NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 4, 2, 128, 1>("128x128 p4 nb2 no reduce NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 4, 4, 128, 3>("128x128 p4 nb4 fast NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 4, 4, 128, 2>("128x128 p4 nb4 loads only NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 8, 2, 128, 2>("128x128 p8 nb2 loads only NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<2, 4, 2, 128, 3>("256x128 p4 nb2 fast NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<2, 4, 2, 128, 2>("256x128 p4 nb2 loads only NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<2, 4, 3, 128, 3>("256x128 p4 nb3 fast NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<4, 4, 2, 128, 3>("512x128 p4 nb2 fast NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<4, 2, 2, 128, 2>("512x128 p2 nb2 loads only NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 4, 2, 256, 3>("128x256 p4 nb2 fast NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 4, 2, 64, 3>("128x64 p4 nb2 fast NT", n, dM, dx, yref, COLMAJOR, reps);
  run_variant<1, 4, 2, 128, 3>("128x128 p4 nb2 fast NT rowmajor", n, dM, dx, yref, ROWMAJOR, reps);
  return 0;
}
