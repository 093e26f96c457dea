// tv1d_bench.hip -- developer microbenchmark (NOT part of libadmm_hip.so): the 1-D total-variation iteration of
// tv_direct2.h (thread-owned 8-position runs, hierarchical exponential sums) in its variants, checked against a
// host Thomas solve of the same iteration and timed at n = 4096^2.
//   hipcc -O3 --offload-arch=gfx950 -I.. -o tv1d_bench tv1d_bench.hip && ./tv1d_bench [n] [rho]
#include "../tv_direct2.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace admm;

#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t _e = (e);                                                        \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <int NW, int OCC, bool EXTRA>
__global__ __launch_bounds__(NW * 64, OCC) void tv2_bench_kernel(TvArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  asm volatile("" ::"s"(a.n), "s"(a.z), "s"(a.s), "s"(a.zo), "s"(a.part), "s"(a.ftile), "s"(a.margin),
               "s"(a.part_stride), "s"(a.thresh), "s"(a.rho), "s"(a.bstar), "s"(a.green), "s"(a.objevals),
               "s"(a.xhist));
  tv2_tile<NW, true, true, EXTRA>(a, blockIdx.x, 0, lds, [] { return false; });
}

static double clampd(double v, double t) { return std::fmin(std::fmax(v, -t), t); }

template <int NW, int OCC>
static void run(const char* name, TvArgs a, double* v0, double* v1, const std::vector<double>& vref,
                const std::vector<double>& xref, const double* sums_ref, int iters) {
  constexpr int WIN = NW * 64 * 8;
  a.ftile = WIN - 2 * a.margin;
  const int64_t ntiles = (a.n + a.ftile - 1) / a.ftile;
  a.part_stride = ntiles;
  const size_t lds = sizeof(double) * tv2_lds_doubles<NW>();
  std::vector<double> xh(a.n), zh(a.n), uh(a.n);
  double *dx, *dz, *du;
  CK(hipMalloc(&dx, 8 * a.n));
  CK(hipMalloc(&dz, 8 * a.n));
  CK(hipMalloc(&du, 8 * a.n));
  // correctness: one iteration with histories
  a.z = v0;
  a.zo = v1;
  a.xhist = dx;
  a.zhist = dz;
  a.uhist = du;
  CK(hipMemset(v1, 0, 8 * a.n));
  hipLaunchKernelGGL((tv2_bench_kernel<NW, OCC, true>), dim3(ntiles), dim3(NW * 64), lds, 0, a);
  CK(hipDeviceSynchronize());
  std::vector<double> got(a.n), part(S_COUNT * ntiles);
  CK(hipMemcpy(got.data(), v1, 8 * a.n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(xh.data(), dx, 8 * a.n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(part.data(), a.part, 8 * S_COUNT * ntiles, hipMemcpyDeviceToHost));
  double ev = 0, ex = 0, mv = 0, mx = 0;
  for (int64_t i = 0; i < a.n; ++i) {
    ev = std::fmax(ev, std::fabs(got[i] - vref[i]));
    mv = std::fmax(mv, std::fabs(vref[i]));
    ex = std::fmax(ex, std::fabs(xh[i] - xref[i]));
    mx = std::fmax(mx, std::fabs(xref[i]));
  }
  double es = 0;
  for (int s = 0; s < S_COUNT; ++s) {
    double t = 0;
    for (int64_t b = 0; b < ntiles; ++b) t += part[s * ntiles + b];
    if (sums_ref[s] != 0) es = std::fmax(es, std::fabs(t - sums_ref[s]) / std::fabs(sums_ref[s]));
    else if (t != 0) es = 1;
  }
  CK(hipFree(dx));
  CK(hipFree(dz));
  CK(hipFree(du));
  a.xhist = a.zhist = a.uhist = nullptr;
  // timing: ping-pong
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int k = 0; k < 10; ++k) {
    a.z = (k & 1) ? v1 : v0;
    a.zo = (k & 1) ? v0 : v1;
    hipLaunchKernelGGL((tv2_bench_kernel<NW, OCC, false>), dim3(ntiles), dim3(NW * 64), lds, 0, a);
  }
  CK(hipEventRecord(e0));
  for (int k = 0; k < iters; ++k) {
    a.z = (k & 1) ? v1 : v0;
    a.zo = (k & 1) ? v0 : v1;
    hipLaunchKernelGGL((tv2_bench_kernel<NW, OCC, false>), dim3(ntiles), dim3(NW * 64), lds, 0, a);
  }
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / iters;
  printf("%-28s tiles %6lld  lds %5zu B  rel.err v+ %.2e x %.2e sums %.2e   %8.2f us/iter  %6.0f GB/s (3 passes) frac %.3f\n",
         name, static_cast<long long>(ntiles), lds, ev / mv, ex / mx, es, us, 24.0 * a.n / us * 1e-3,
         24.0 * a.n / us * 1e-3 / 8000.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 4096LL * 4096LL;
  const double rho = argc > 2 ? atof(argv[2]) : 1.0;
  const double lam = 1.0, th = lam / rho;
  // stationary pivot b* and halo as tv_plan computes them
  double cur = 1.0 + rho;
  for (int i = 0; i < 100000; ++i) {
    const double nxt = (1.0 + 2.0 * rho) - rho * rho / cur;
    if (nxt == cur) break;
    cur = nxt;
  }
  const double bstar = cur, r = rho / bstar;
  int H = 0;
  for (double prod = 1.0; prod > 1e-18; ++H) prod *= (H == 0) ? rho / (1.0 + rho) : r;
  if (H & 1) ++H;
  const int margin = (H + 8 + 7) / 8 * 8;
  printf("n %lld rho %g  b* %.6f r %.6f halo %d margin %d\n", static_cast<long long>(n), rho, bstar, r, H, margin);
  // data: piecewise constant + noise (signal), a state v with entries on both sides of the threshold
  std::vector<double> s(n), v(n);
  uint64_t st = 88172645463325252ULL;
  auto rnd = [&]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return (st >> 11) * (1.0 / 9007199254740992.0);
  };
  for (int64_t i = 0; i < n; ++i) {
    s[i] = ((i / 100000) % 7) + 2.0 * (rnd() - 0.5);
    v[i] = 3.0 * (rnd() - 0.5);
  }
  // host reference: one iteration (Thomas solve of the exact matrix)
  std::vector<double> u(n), z(n), bb(n), xr(n), vr(n), cp(n);
  for (int64_t i = 0; i < n; ++i) {
    u[i] = clampd(v[i], th);
    z[i] = v[i] - u[i];
  }
  for (int64_t i = 0; i < n; ++i) {
    const double t = z[i] - u[i], tm = i > 0 ? z[i - 1] - u[i - 1] : 0.0;
    bb[i] = s[i] + rho * (i > 0 ? t - tm : t);
  }
  {
    double den = 1.0 + rho;
    cp[0] = -rho / den;
    xr[0] = bb[0] / den;
    for (int64_t i = 1; i < n; ++i) {
      den = (1.0 + 2.0 * rho) + rho * cp[i - 1];
      cp[i] = -rho / den;
      xr[i] = (bb[i] + rho * xr[i - 1]) / den;
    }
    for (int64_t i = n - 2; i >= 0; --i) xr[i] -= cp[i] * xr[i + 1];
  }
  double sums[S_COUNT] = {0};
  {
    double unm = 0, dzm = 0;
    for (int64_t i = 0; i < n; ++i) {
      const double ax = i + 1 < n ? xr[i] - xr[i + 1] : xr[i];
      const double v1 = u[i] + ax, un = clampd(v1, th), zn = v1 - un, dz = zn - z[i];
      vr[i] = v1;
      const double rr = ax - zn, du = un - u[i], g2 = i > 0 ? dz - dzm : dz, g3 = i > 0 ? un - unm : un;
      sums[S_R2] += rr * rr;
      sums[S_AX2] += ax * ax;
      sums[S_Z2] += zn * zn;
      sums[S_DZ2] += dz * dz;
      sums[S_U2] += un * un;
      sums[S_DU2] += du * du;
      sums[S_G2] += g2 * g2;
      sums[S_G3] += g3 * g3;
      unm = un;
      dzm = dz;
    }
  }
  double *ds, *v0, *v1, *dpart;
  CK(hipMalloc(&ds, 8 * n));
  CK(hipMalloc(&v0, 8 * n));
  CK(hipMalloc(&v1, 8 * n));
  CK(hipMalloc(&dpart, 8 * S_COUNT * (n / 1024 + 16)));
  CK(hipMemcpy(ds, s.data(), 8 * n, hipMemcpyHostToDevice));
  TvArgs a{};
  a.n = n;
  a.rho = rho;
  a.thresh = th;
  a.s = ds;
  a.bstar = bstar;
  a.halo = H;
  a.margin = margin;
  a.green = 1.0 / (bstar * (1.0 - r * r));
  a.rpow[0] = r;
  for (int k = 1; k < 8; ++k) a.rpow[k] = a.rpow[k - 1] * r;
  a.part = dpart;
  const int iters = n >= (1 << 22) ? 200 : 20;
#define RUN(NW, OCC)                                                                 \
  CK(hipMemcpy(v0, v.data(), 8 * n, hipMemcpyHostToDevice));                         \
  run<NW, OCC>("NW=" #NW " OCC=" #OCC, a, v0, v1, vr, xr, sums, iters)
  RUN(4, 4);
  RUN(4, 5);
  RUN(4, 6);
  RUN(4, 7);
  RUN(4, 8);
  RUN(8, 2);
  RUN(8, 3);
  RUN(8, 4);
  RUN(2, 8);
  RUN(2, 10);
  RUN(2, 12);
  RUN(1, 8);
  return 0;
}
