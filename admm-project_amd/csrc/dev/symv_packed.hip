// symv_packed.hip -- developer microbenchmark (NOT part of libadmm_hip.so): the lower-triangle SYMV of symv.hip on
// the column-major padded storage against the same inner loop on TILE-PACKED storage (every 128x128 tile of the
// lower triangle contiguous, 128 KB, tiles in row-major triangle order).
//   hipcc -O3 --offload-arch=gfx950 -I.. -o symv_packed symv_packed.hip && ./symv_packed [n]
#include "../symv.hip"

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

__global__ void fill_sym(double* M, int64_t n, int64_t ld) {
  const int64_t j = blockIdx.x;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const int64_t a = i > j ? i : j, b = i > j ? j : i;
    M[j * ld + i] = 1e-3 * static_cast<double>((a * 131 + b * 7) % 1009 - 504) + (i == j ? 3.0 : 0.0);
  }
}

// tile t of the lower triangle (row-major: (0,0), (1,0), (1,1), (2,0) ...) <- column-major padded storage
__global__ void pack_tiles(const double* __restrict__ M, int64_t ld, double* __restrict__ P) {
  const unsigned t = blockIdx.x;
  unsigned bi = static_cast<unsigned>((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (bi * (bi + 1) / 2 > t) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const unsigned bj = t - bi * (bi + 1) / 2;
  const double* src = M + static_cast<int64_t>(bj) * kSyTile * ld + static_cast<int64_t>(bi) * kSyTile;
  double* dst = P + static_cast<int64_t>(t) * kSyTile * kSyTile;
  for (int e = threadIdx.x; e < kSyTile * kSyTile; e += blockDim.x) dst[e] = src[(e / kSyTile) * ld + (e % kSyTile)];
}

template <bool NT, int ORDER>
__global__ __launch_bounds__(kWave) void symv_packed_kernel(const double* __restrict__ P, int64_t n,
                                                            const double* __restrict__ x, double* __restrict__ npart,
                                                            double* __restrict__ tpart, int64_t ldp, unsigned ntile) {
  unsigned t = blockIdx.x;
  if (ORDER == 1) t = gridDim.x - 1 - t;  // long rows first
  unsigned bi = static_cast<unsigned>((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (bi * (bi + 1) / 2 > t) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const unsigned bj = t - bi * (bi + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = static_cast<int64_t>(bi) * kSyTile, c0 = static_cast<int64_t>(bj) * kSyTile;
  SyLane s;
  s.M = P + static_cast<int64_t>(t) * kSyTile * kSyTile;
  s.x = x + c0;
  s.ld = kSyTile;
  s.n = n - c0;
  s.r = 2 * lane;
  const int64_t gr = w0 + 2 * lane;
  s.xr0 = x[gr < n ? gr : n - 1];
  s.xr1 = x[gr + 1 < n ? gr + 1 : n - 1];
  double* __restrict__ tout = tpart + static_cast<int64_t>(bi) * ldp + c0;
  double n0 = 0.0, n1 = 0.0;
  if (bi == bj) sy_tile<true, NT>(s, 0, n0, n1, tout, lane);
  else sy_tile<false, NT>(s, 0, n0, n1, tout, lane);
  *reinterpret_cast<double2_t*>(npart + static_cast<int64_t>(bj) * ldp + gr) = double2_t{n0, n1};
}

// ---- lane-blocked tiles: lane (a, b) = (lane >> 3, lane & 7) owns the 16x16 sub-block (rows 16a.., cols 16b..) of the
// 128x128 tile; load k of a lane brings rows (2rp, 2rp+1), column cc of its sub-block (k = cc*8 + rp) and the tile is
// stored in load order: element pair (k, lane) at ((k*64 + lane)*2) -- every wave load is 1 KB contiguous.  All 512
// FMAs of a lane are register-local; cross-lane sums happen once per tile.
constexpr int kRB = 16, kCB = 16;
__global__ void pack_blocked(const double* __restrict__ M, int64_t ld, double* __restrict__ P) {
  const unsigned t = blockIdx.x;
  unsigned bi = static_cast<unsigned>((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (bi * (bi + 1) / 2 > t) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const unsigned bj = t - bi * (bi + 1) / 2;
  const double* src = M + static_cast<int64_t>(bj) * kSyTile * ld + static_cast<int64_t>(bi) * kSyTile;
  double* dst = P + static_cast<int64_t>(t) * kSyTile * kSyTile;
  for (int e = threadIdx.x; e < kSyTile * kSyTile; e += blockDim.x) {
    const int k = e / 128, rem = e % 128, L = rem / 2, h = rem % 2, a = L >> 3, b = L & 7, cc = k / 8, rp = k % 8;
    dst[e] = src[static_cast<int64_t>(16 * b + cc) * ld + 16 * a + 2 * rp + h];
  }
}

template <bool DIAG, bool NT, int DEPTH>
__device__ __forceinline__ void blk_tile(const double* __restrict__ T, int lane, const double (&xc)[kCB],
                                         const double (&xr)[kRB], double (&nacc)[kRB], double (&tacc)[kCB]) {
  constexpr int NL = kCB * kRB / 2;
  const double* p = T + 2 * lane;
  const int a = lane >> 3, b = lane & 7;
  double2_t buf[DEPTH];
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) buf[k] = load2<NT>(p + k * 128);
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const double2_t d = buf[k % DEPTH];
    if (k + DEPTH < NL) buf[k % DEPTH] = load2<NT>(p + (k + DEPTH) * 128);
    const int cc = k / (kRB / 2), rp = k % (kRB / 2);
    double m0 = d.x, m1 = d.y, t0 = d.x, t1 = d.y;
    if (DIAG) {
      const int r = kRB * a + 2 * rp, c = kCB * b + cc;
      t0 = (r > c) ? m0 : 0.0;
      t1 = (r + 1 > c) ? m1 : 0.0;
      m0 = (c <= r) ? m0 : 0.0;
      m1 = (c <= r + 1) ? m1 : 0.0;
    }
    nacc[2 * rp] = __builtin_fma(m0, xc[cc], nacc[2 * rp]);
    nacc[2 * rp + 1] = __builtin_fma(m1, xc[cc], nacc[2 * rp + 1]);
    tacc[cc] = __builtin_fma(t0, xr[2 * rp], tacc[cc]);
    tacc[cc] = __builtin_fma(t1, xr[2 * rp + 1], tacc[cc]);
  }
}

template <bool NT, int DEPTH, int OCC>
__global__ __launch_bounds__(kWave, OCC) void symv_blocked_kernel(const double* __restrict__ P, int64_t n,
                                                                 const double* __restrict__ x,
                                                                 double* __restrict__ npart,
                                                                 double* __restrict__ tpart, int64_t ldp) {
  const unsigned t = blockIdx.x;
  unsigned bi = static_cast<unsigned>((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (bi * (bi + 1) / 2 > t) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const unsigned bj = t - bi * (bi + 1) / 2;
  const int lane = threadIdx.x & 63, a = lane >> 3, b = lane & 7;
  const int64_t w0 = static_cast<int64_t>(bi) * kSyTile, c0 = static_cast<int64_t>(bj) * kSyTile;
  double xc[kCB], xr[kRB], nacc[kRB], tacc[kCB];
#pragma unroll
  for (int k = 0; k < kCB; k += 2) {  // x is zero-padded to whole tiles
    const double2_t v = *reinterpret_cast<const double2_t*>(x + c0 + kCB * b + k);
    xc[k] = v.x;
    xc[k + 1] = v.y;
  }
#pragma unroll
  for (int k = 0; k < kRB; k += 2) {
    const double2_t v = *reinterpret_cast<const double2_t*>(x + w0 + kRB * a + k);
    xr[k] = v.x;
    xr[k + 1] = v.y;
  }
#pragma unroll
  for (int k = 0; k < kRB; ++k) nacc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < kCB; ++k) tacc[k] = 0.0;
  const double* T = P + static_cast<int64_t>(t) * kSyTile * kSyTile;
  if (bi == bj) blk_tile<true, NT, DEPTH>(T, lane, xc, xr, nacc, tacc);
  else blk_tile<false, NT, DEPTH>(T, lane, xc, xr, nacc, tacc);
#pragma unroll
  for (int k = 0; k < kRB; ++k) {  // rows: sum over the 8 lanes b = 0..7 of one a (lane bits 0..2)
    double v = nacc[k];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    nacc[k] = v;
  }
#pragma unroll
  for (int k = 0; k < kCB; ++k) {  // columns: sum over a = 0..7 (lane bits 3..5)
    double v = tacc[k];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    tacc[k] = v;
  }
  if (b == 0) {
    double* dst = npart + static_cast<int64_t>(bj) * ldp + w0 + kRB * a;
#pragma unroll
    for (int k = 0; k < kRB; k += 2) *reinterpret_cast<double2_t*>(dst + k) = double2_t{nacc[k], nacc[k + 1]};
  }
  if (a == 0) {
    double* dst = tpart + static_cast<int64_t>(bi) * ldp + c0 + kCB * b;
#pragma unroll
    for (int k = 0; k < kCB; k += 2) *reinterpret_cast<double2_t*>(dst + k) = double2_t{tacc[k], tacc[k + 1]};
  }
}

template <bool DIAG, bool NT, int DEPTH>
__device__ __forceinline__ void persist_body(double2_t (&buf)[DEPTH], const double* __restrict__ p,
                                             const double* __restrict__ pn, int a, int b, const double (&xc)[kCB],
                                             const double (&xr)[kRB], double (&nacc)[kRB], double (&tacc)[kCB]) {
  constexpr int NL = kCB * kRB / 2;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const double2_t d = buf[k % DEPTH];
    if (k + DEPTH < NL) buf[k % DEPTH] = load2<NT>(p + (k + DEPTH) * 128);
    else buf[k % DEPTH] = load2<NT>(pn + (k + DEPTH - NL) * 128);  // next tile (or a harmless re-read of this one)
    const int cc = k / (kRB / 2), rp = k % (kRB / 2);
    double m0 = d.x, m1 = d.y, t0 = d.x, t1 = d.y;
    if (DIAG) {
      const int r = kRB * a + 2 * rp, c = kCB * b + cc;
      t0 = (r > c) ? m0 : 0.0;
      t1 = (r + 1 > c) ? m1 : 0.0;
      m0 = (c <= r) ? m0 : 0.0;
      m1 = (c <= r + 1) ? m1 : 0.0;
    }
    nacc[2 * rp] = __builtin_fma(m0, xc[cc], nacc[2 * rp]);
    nacc[2 * rp + 1] = __builtin_fma(m1, xc[cc], nacc[2 * rp + 1]);
    tacc[cc] = __builtin_fma(t0, xr[2 * rp], tacc[cc]);
    tacc[cc] = __builtin_fma(t1, xr[2 * rp + 1], tacc[cc]);
  }
}

// ---- persistent form: W one-wave workgroups, wave w takes tiles w, w + W, ...; the load pipeline runs across tile
// boundaries (the first DEPTH loads of the next tile are issued during the last DEPTH steps of the current one)
template <bool NT, int DEPTH>
__global__ __launch_bounds__(kWave, 2) void symv_persist_kernel(const double* __restrict__ P, int64_t n,
                                                               const double* __restrict__ x,
                                                               double* __restrict__ npart,
                                                               double* __restrict__ tpart, int64_t ldp,
                                                               unsigned ntri) {
  constexpr int NL = kCB * kRB / 2;
  const int lane = threadIdx.x & 63, a = lane >> 3, b = lane & 7;
  unsigned t = blockIdx.x;
  if (t >= ntri) return;
  double2_t buf[DEPTH];
  const double* p = P + static_cast<int64_t>(t) * kSyTile * kSyTile + 2 * lane;
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) buf[k] = load2<NT>(p + k * 128);
  for (; t < ntri; t += gridDim.x) {
    unsigned bi = static_cast<unsigned>((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while (bi * (bi + 1) / 2 > t) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    const unsigned bj = t - bi * (bi + 1) / 2;
    const int64_t w0 = static_cast<int64_t>(bi) * kSyTile, c0 = static_cast<int64_t>(bj) * kSyTile;
    const bool has_next = t + gridDim.x < ntri;
    const double* pn = P + static_cast<int64_t>(has_next ? t + gridDim.x : t) * kSyTile * kSyTile + 2 * lane;
    double xc[kCB], xr[kRB], nacc[kRB], tacc[kCB];
#pragma unroll
    for (int k = 0; k < kCB; k += 2) {
      const double2_t v = *reinterpret_cast<const double2_t*>(x + c0 + kCB * b + k);
      xc[k] = v.x;
      xc[k + 1] = v.y;
    }
#pragma unroll
    for (int k = 0; k < kRB; k += 2) {
      const double2_t v = *reinterpret_cast<const double2_t*>(x + w0 + kRB * a + k);
      xr[k] = v.x;
      xr[k + 1] = v.y;
    }
#pragma unroll
    for (int k = 0; k < kRB; ++k) nacc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < kCB; ++k) tacc[k] = 0.0;
    if (bi == bj) persist_body<true, NT, DEPTH>(buf, p, pn, a, b, xc, xr, nacc, tacc);
    else persist_body<false, NT, DEPTH>(buf, p, pn, a, b, xc, xr, nacc, tacc);
    p = pn;
#pragma unroll
    for (int k = 0; k < kRB; ++k) {
      double v = nacc[k];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      nacc[k] = v;
    }
#pragma unroll
    for (int k = 0; k < kCB; ++k) {
      double v = tacc[k];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      tacc[k] = v;
    }
    if (b == 0) {
      double* dst = npart + static_cast<int64_t>(bj) * ldp + w0 + kRB * a;
#pragma unroll
      for (int k = 0; k < kRB; k += 2) *reinterpret_cast<double2_t*>(dst + k) = double2_t{nacc[k], nacc[k + 1]};
    }
    if (a == 0) {
      double* dst = tpart + static_cast<int64_t>(bi) * ldp + c0 + kCB * b;
#pragma unroll
      for (int k = 0; k < kCB; k += 2) *reinterpret_cast<double2_t*>(dst + k) = double2_t{tacc[k], tacc[k + 1]};
    }
  }
}

// hybrid cache policy: the first `ncached` tiles with default loads (candidates for Infinity-Cache residency between
// launches), the rest streamed non-temporally
__global__ __launch_bounds__(kWave) void symv_hybrid_kernel(const double* __restrict__ P, int64_t n,
                                                            const double* __restrict__ x, double* __restrict__ npart,
                                                            double* __restrict__ tpart, int64_t ldp, unsigned ncached) {
  const unsigned t = blockIdx.x;
  unsigned bi = static_cast<unsigned>((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (bi * (bi + 1) / 2 > t) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const unsigned bj = t - bi * (bi + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = static_cast<int64_t>(bi) * kSyTile, c0 = static_cast<int64_t>(bj) * kSyTile;
  SyLane s;
  s.M = P + static_cast<int64_t>(t) * kSyTile * kSyTile;
  s.x = x + c0;
  s.ld = kSyTile;
  s.n = n - c0;
  s.r = 2 * lane;
  const int64_t gr = w0 + 2 * lane;
  s.xr0 = x[gr < n ? gr : n - 1];
  s.xr1 = x[gr + 1 < n ? gr + 1 : n - 1];
  double* __restrict__ tout = tpart + static_cast<int64_t>(bi) * ldp + c0;
  double n0 = 0.0, n1 = 0.0;
  if (bi == bj) sy_tile<true, true>(s, 0, n0, n1, tout, lane);
  else if (t < ncached) sy_tile<false, false>(s, 0, n0, n1, tout, lane);
  else sy_tile<false, true>(s, 0, n0, n1, tout, lane);
  *reinterpret_cast<double2_t*>(npart + static_cast<int64_t>(bj) * ldp + gr) = double2_t{n0, n1};
}

// the production (column-major, padded) kernel with the hybrid policy; MODE 0: tiles with linear index < ncached are
// cacheable, MODE 1: interleaved, tile cacheable iff (linear index % 16) < ncached
template <int MODE>
__global__ __launch_bounds__(kWave) void symv_lower_hybrid(const double* __restrict__ M, int64_t n, int64_t ld,
                                                           const double* __restrict__ x, double* __restrict__ npart,
                                                           double* __restrict__ tpart, int64_t ldp, unsigned ncached) {
  if (blockIdx.x < blockIdx.y) return;
  const unsigned lin = blockIdx.x * (blockIdx.x + 1u) / 2u + blockIdx.y;
  const bool cached = MODE == 0 ? lin < ncached : (lin & 15u) < ncached;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = static_cast<int64_t>(blockIdx.x) * kSyTile;
  const int64_t c0 = static_cast<int64_t>(blockIdx.y) * kSyTile;
  SyLane s;
  s.M = M;
  s.x = x;
  s.ld = ld;
  s.n = n;
  s.r = static_cast<int>(w0) + 2 * lane;
  s.xr0 = x[s.r < n ? s.r : n - 1];
  s.xr1 = x[s.r + 1 < n ? s.r + 1 : n - 1];
  double* __restrict__ tout = tpart + static_cast<int64_t>(blockIdx.x) * ldp;
  double n0 = 0.0, n1 = 0.0;
  if (blockIdx.x == blockIdx.y) sy_tile<true, true>(s, c0, n0, n1, tout, lane);
  else if (cached) sy_tile<false, false>(s, c0, n0, n1, tout, lane);
  else sy_tile<false, true>(s, c0, n0, n1, tout, lane);
  *reinterpret_cast<double2_t*>(npart + static_cast<int64_t>(blockIdx.y) * ldp + s.r) = double2_t{n0, n1};
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10000;
  const SymvPlan p = symv_plan(n);
  const int64_t ld = p.npad;
  const unsigned nt = static_cast<unsigned>(p.ntile), ntri = nt * (nt + 1) / 2;
  double *M, *P, *x, *np_, *tp, *y, *y2;
  CK(hipMalloc(&M, sizeof(double) * ld * p.npad));
  CK(hipMemset(M, 0, sizeof(double) * ld * p.npad));
  CK(hipMalloc(&P, sizeof(double) * ntri * kSyTile * kSyTile));
  CK(hipMalloc(&x, sizeof(double) * p.npad));
  CK(hipMalloc(&np_, sizeof(double) * nt * p.ldp));
  CK(hipMalloc(&tp, sizeof(double) * nt * p.ldp));
  CK(hipMalloc(&y, sizeof(double) * p.npad));
  CK(hipMalloc(&y2, sizeof(double) * p.npad));
  std::vector<double> hx(p.npad, 0.0);
  for (int64_t i = 0; i < n; ++i) hx[i] = 0.25 + 0.001 * (i % 97);
  CK(hipMemcpy(x, hx.data(), sizeof(double) * p.npad, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fill_sym, dim3(n), dim3(256), 0, 0, M, n, ld);
  hipLaunchKernelGGL(pack_tiles, dim3(ntri), dim3(256), 0, 0, M, ld, P);
  CK(hipDeviceSynchronize());
  printf("n=%lld tiles=%u packed %.1f MB (padded square %.1f MB)\n", (long long)n, ntri, ntri * 131072e-6,
         8e-6 * ld * p.npad);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 60;
  auto time_it = [&](const char* name, auto&& launch, double* yout) {
    CK(hipMemset(np_, 0, sizeof(double) * nt * p.ldp));
    CK(hipMemset(tp, 0, sizeof(double) * nt * p.ldp));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipLaunchKernelGGL(symv_reduce_kernel, dim3(ceil_div(n, 16)), dim3(kBlock), 0, 0, np_, tp, p.ldp, n, p.ntile, yout,
                       nullptr);
    CK(hipDeviceSynchronize());
    const double us = ms * 1e3 / reps;
    printf("%-44s %8.2f us  %6.3f TB/s\n", name, us, ntri * 131072.0 / us * 1e-6);
  };
  time_it("column-major padded (production)", [&] {
    hipLaunchKernelGGL(symv_lower_kernel<false>, dim3(nt, nt), dim3(kWave), 0, 0, M, n, ld, x, np_, tp, p.ldp, 0, 1, 0u,
                       nullptr);
  }, y);
  time_it("tile-packed, NT", [&] {
    hipLaunchKernelGGL((symv_packed_kernel<true, 0>), dim3(ntri), dim3(kWave), 0, 0, P, n, x, np_, tp, p.ldp, nt);
  }, y2);
  std::vector<double> h1(n), h2(n);
  CK(hipMemcpy(h1.data(), y, sizeof(double) * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), y2, sizeof(double) * n, hipMemcpyDeviceToHost));
  double err = 0;
  for (int64_t i = 0; i < n; ++i) err = fmax(err, fabs(h1[i] - h2[i]));
  printf("max |y_packed - y_colmajor| = %.3e\n", err);
  time_it("tile-packed, NT, reversed order", [&] {
    hipLaunchKernelGGL((symv_packed_kernel<true, 1>), dim3(ntri), dim3(kWave), 0, 0, P, n, x, np_, tp, p.ldp, nt);
  }, y2);
  time_it("tile-packed, default loads", [&] {
    hipLaunchKernelGGL((symv_packed_kernel<false, 0>), dim3(ntri), dim3(kWave), 0, 0, P, n, x, np_, tp, p.ldp, nt);
  }, y2);
  for (unsigned nc : {0u, 400u, 800u, 1200u, 1500u, 1800u, 2200u, 3160u}) {
    char nm[96];
    snprintf(nm, sizeof nm, "tile-packed hybrid: %u tiles (%.0f MB) cacheable", nc, nc * 0.131072);
    time_it(nm, [&] {
      hipLaunchKernelGGL(symv_hybrid_kernel, dim3(ntri), dim3(kWave), 0, 0, P, n, x, np_, tp, p.ldp, nc);
    }, y2);
  }
  for (unsigned nc : {0u, 800u, 1200u, 1500u, 1800u}) {
    char nm[96];
    snprintf(nm, sizeof nm, "column-major hybrid: first %u tiles cacheable", nc);
    time_it(nm, [&] {
      hipLaunchKernelGGL(symv_lower_hybrid<0>, dim3(nt, nt), dim3(kWave), 0, 0, M, n, ld, x, np_, tp, p.ldp, nc);
    }, y2);
  }
  for (unsigned nc : {4u, 6u, 7u, 8u, 9u}) {
    char nm[96];
    snprintf(nm, sizeof nm, "column-major hybrid: %u of 16 tiles cacheable", nc);
    time_it(nm, [&] {
      hipLaunchKernelGGL(symv_lower_hybrid<1>, dim3(nt, nt), dim3(kWave), 0, 0, M, n, ld, x, np_, tp, p.ldp, nc);
    }, y2);
  }
  hipLaunchKernelGGL(pack_blocked, dim3(ntri), dim3(256), 0, 0, M, ld, P);
  CK(hipDeviceSynchronize());
#define BLK(DEPTH, OCC)                                                                                               \
  time_it("lane-blocked depth " #DEPTH " occ " #OCC, [&] {                                                            \
    hipLaunchKernelGGL((symv_blocked_kernel<true, DEPTH, OCC>), dim3(ntri), dim3(kWave), 0, 0, P, n, x, np_, tp,      \
                       p.ldp);                                                                                        \
  }, y2);                                                                                                             \
  CK(hipMemcpy(h2.data(), y2, sizeof(double) * n, hipMemcpyDeviceToHost));                                            \
  err = 0;                                                                                                            \
  for (int64_t i = 0; i < n; ++i) err = fmax(err, fabs(h1[i] - h2[i]));                                               \
  printf("   max |y - y_colmajor| = %.3e\n", err);
  BLK(8, 1)
  BLK(16, 1)
  BLK(32, 1)
  BLK(8, 2)
  BLK(16, 2)
  BLK(16, 3)
  BLK(8, 4)
#define PERS(DEPTH, W)                                                                                               \
  {                                                                                                                   \
    char nm[96];                                                                                                      \
    snprintf(nm, sizeof nm, "persistent depth %d waves %u", DEPTH, static_cast<unsigned>(W));                        \
    time_it(nm, [&] {                                                                                                 \
      hipLaunchKernelGGL((symv_persist_kernel<true, DEPTH>), dim3(W), dim3(kWave), 0, 0, P, n, x, np_, tp, p.ldp,    \
                         ntri);                                                                                       \
    }, y2);                                                                                                           \
    CK(hipMemcpy(h2.data(), y2, sizeof(double) * n, hipMemcpyDeviceToHost));                                          \
    err = 0;                                                                                                          \
    for (int64_t i = 0; i < n; ++i) err = fmax(err, fabs(h1[i] - h2[i]));                                             \
    printf("   max |y - y_colmajor| = %.3e\n", err);                                                                 \
  }
  PERS(8, ntri)
  PERS(8, (ntri + 1) / 2)
  PERS(16, (ntri + 1) / 2)
  PERS(8, (ntri + 2) / 3)
  PERS(16, (ntri + 2) / 3)
  PERS(16, (ntri + 3) / 4)
  PERS(8, 2048)
  PERS(16, 1024)
  {
    SymvPlan pp = p;
    pp.packed = true;
    hipLaunchKernelGGL(pack_tiles, dim3(ntri), dim3(256), 0, 0, M, ld, P);
    CK(hipDeviceSynchronize());
    for (int64_t mb : {0, 96, 128, 168, 200, 240}) {
      pp.ncached = symv_cached_tiles(pp, mb << 20);
      char nm[96];
      snprintf(nm, sizeof nm, "PRODUCTION packed kernel, %lld MB cacheable", static_cast<long long>(mb));
      time_it(nm, [&] { launch_symv_lower(pp, P, 0, x, np_, tp, y2, nullptr, 0, 0, 1, false); }, y2);
    }
    CK(hipMemcpy(h2.data(), y2, sizeof(double) * n, hipMemcpyDeviceToHost));
    err = 0;
    for (int64_t i = 0; i < n; ++i) err = fmax(err, fabs(h1[i] - h2[i]));
    printf("   max |y - y_colmajor| = %.3e\n", err);
  }
  time_it("column-major padded again", [&] {
    hipLaunchKernelGGL(symv_lower_kernel<false>, dim3(nt, nt), dim3(kWave), 0, 0, M, n, ld, x, np_, tp, p.ldp, 0, 1, 0u,
                       nullptr);
  }, y);
  return 0;
}
