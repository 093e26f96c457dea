// symv.hip -- y = M*x for a SYMMETRIC M reading only its lower triangle (half the bytes of a GEMV).
// Used for the explicit-inverse x-update  x = inv(D'D + rho I) * y  (getProxOps.m:1200 semantics).
//
// Every element M[r,j] (r >= j) is loaded once and used twice:
//   N-part  yN[r] += M[r,j]*x[j]        register accumulation per row (like gemv_n)
//   T-part  yT[j] += M[r,j]*x[r], r > j per-column sums over the rows (like gemv_t)
// A WAVE is the unit of work (one-wave workgroups: 128 rows x 128 columns, 16-byte loads, lane
// owns a row pair), so the triangular tile set load-balances finely and nothing synchronises.
// The T-part column sums of an 8-column panel are combined inside the wave by a 10-shuffle
// reduce-scatter; the loads of the NEXT panel are issued before that reduction (two register
// buffers, hand-pipelined), so the memory pipe never drains while a wave shuffles.
// Partials (npart per 128-column group, tpart per 128-row wave chunk) are added by a second small
// kernel in a fixed order: bitwise reproducible, no float atomics.
#include <cstdlib>

#include "kernels.h"

namespace admm {

typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int kSyWaveRows = 128;  // rows per wave = per workgroup (64 lanes x 2)
constexpr int kSyCols = 128;      // columns per workgroup
constexpr int kSyPanel = 8;       // columns per panel = loads per buffer = T-part accumulators

// reduce-scatter of 8 per-lane values over the 64 lanes of a wave: on return every lane holds the
// wave-wide sum of element *col (its lane bits 5..3 select the column); 7 + 3 shuffles.
__device__ __forceinline__ double reduce_scatter8(const double (&t)[kSyPanel], int lane, int* col) {
  double a4[4], a2[2];
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double keep = b5 ? t[k + 4] : t[k];
    const double send = b5 ? t[k] : t[k + 4];
    a4[k] = keep + __shfl_xor(send, 32, 64);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double keep = b4 ? a4[k + 2] : a4[k];
    const double send = b4 ? a4[k] : a4[k + 2];
    a2[k] = keep + __shfl_xor(send, 16, 64);
  }
  const double keep = b3 ? a2[1] : a2[0];
  const double send = b3 ? a2[0] : a2[1];
  double r = keep + __shfl_xor(send, 8, 64);
  r += __shfl_xor(r, 4, 64);
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  *col = (b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0);
  return r;
}

struct SyLane {  // per-lane constants of a tile
  const double* M;
  const double* x;
  int64_t ld, r, cend;
  int r32;  // r as a 32-bit element offset from the wave-uniform column base
  bool live0, live1;
  double xr0, xr1;
};

// FAST: every row of the wave and every column of the panel is inside the matrix -> no guards at
// all on the hot path (straight-line loads).  DIAG: the tile intersects the diagonal -> mask.
template <bool FAST>
__device__ __forceinline__ void sy_load(const SyLane& s, int64_t cp, double2_t (&d)[kSyPanel]) {
#pragma unroll
  for (int k = 0; k < kSyPanel; ++k) {
    const int64_t j = cp + k;
    const double* cb = s.M + j * s.ld;  // wave-uniform column base
    if (FAST) {
      d[k] = *reinterpret_cast<const double2_t*>(cb + s.r32);
    } else {
      d[k] = double2_t{0.0, 0.0};
      if (j < s.cend) {
        if (s.live1) d[k] = *reinterpret_cast<const double2_t*>(cb + s.r32);
        else if (s.live0) d[k].x = cb[s.r32];
      }
    }
  }
}

template <bool FAST, bool DIAG>
__device__ __forceinline__ void sy_compute(const SyLane& s, int64_t cp, const double2_t (&d)[kSyPanel], double& n0,
                                           double& n1, double* __restrict__ tout, int lane) {
  double tacc[kSyPanel];
#pragma unroll
  for (int k = 0; k < kSyPanel; ++k) {
    const int64_t j = cp + k;
    const double xj = (FAST || j < s.cend) ? s.x[j] : 0.0;  // wave-uniform -> scalar load
    double a0 = d[k].x, a1 = d[k].y;
    double t0 = a0, t1 = a1;
    if (DIAG) {  // lower triangle only: N-part j <= row, T-part row > j
      t0 = (s.r > j) ? a0 : 0.0;
      t1 = (s.r + 1 > j) ? a1 : 0.0;
      a0 = (j <= s.r) ? a0 : 0.0;
      a1 = (j <= s.r + 1) ? a1 : 0.0;
    }
    tacc[k] = __builtin_fma(t0, s.xr0, t1 * s.xr1);
    n0 = __builtin_fma(a0, xj, n0);
    n1 = __builtin_fma(a1, xj, n1);
  }
  int col;
  const double sum = reduce_scatter8(tacc, lane, &col);
  if ((lane & 7) == 0 && (FAST || cp + col < s.cend)) tout[cp + col] = sum;
}

// the panels [c0, limit) of one tile, next panel's loads issued before this panel's reduction
template <bool FAST, bool DIAG>
__device__ __forceinline__ void sy_tile(const SyLane& s, int64_t c0, int64_t limit, double& n0, double& n1,
                                        double* __restrict__ tout, int lane) {
  double2_t bufA[kSyPanel], bufB[kSyPanel];
  int64_t cp = c0;
  sy_load<FAST>(s, cp, bufA);
#pragma unroll 1
  for (;;) {
    const bool haveB = cp + kSyPanel < limit;
    if (haveB) sy_load<FAST>(s, cp + kSyPanel, bufB);
    sy_compute<FAST, DIAG>(s, cp, bufA, n0, n1, tout, lane);
    if (!haveB) break;
    const bool haveA = cp + 2 * kSyPanel < limit;
    if (haveA) sy_load<FAST>(s, cp + 2 * kSyPanel, bufA);
    sy_compute<FAST, DIAG>(s, cp + kSyPanel, bufB, n0, n1, tout, lane);
    if (!haveA) break;
    cp += 2 * kSyPanel;
  }
}

__global__ __launch_bounds__(kWave) void symv_lower_kernel(const double* __restrict__ M, int64_t n, int64_t ld,
                                                           const double* __restrict__ x, double* __restrict__ npart,
                                                           double* __restrict__ tpart, int64_t ldp,
                                                           const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = static_cast<int64_t>(blockIdx.x) * kSyWaveRows;  // wave's first row
  const int64_t c0 = static_cast<int64_t>(blockIdx.y) * kSyCols;
  if (w0 >= n || w0 + kSyWaveRows - 1 < c0) return;  // nothing of the lower triangle in this tile
  SyLane s;
  s.M = M;
  s.x = x;
  s.ld = ld;
  s.r = w0 + 2 * lane;  // this lane's row pair (r, r+1)
  s.r32 = static_cast<int>(s.r);
  s.live0 = s.r < n;
  s.live1 = s.r + 1 < n;
  s.xr0 = s.live0 ? x[s.r] : 0.0;
  s.xr1 = s.live1 ? x[s.r + 1] : 0.0;
  s.cend = (c0 + kSyCols < n) ? c0 + kSyCols : n;
  const bool diag = w0 < c0 + kSyCols;  // tile intersects the diagonal: mask element-wise
  // panels at or beyond this column lie strictly above the diagonal for every row of the wave
  const int64_t limit = (s.cend < w0 + kSyWaveRows) ? s.cend : w0 + kSyWaveRows;
  const bool fast = (w0 + kSyWaveRows <= n) && ((limit - c0) % kSyPanel == 0);
  double* __restrict__ tout = tpart + static_cast<int64_t>(blockIdx.x) * ldp;
  double n0 = 0.0, n1 = 0.0;
  if (fast) {
    if (diag) sy_tile<true, true>(s, c0, limit, n0, n1, tout, lane);
    else sy_tile<true, false>(s, c0, limit, n0, n1, tout, lane);
  } else {
    // ragged edge tiles (last row chunk / last column group): one buffer, no pipelining -- keeps the
    // register budget of the kernel set by the fast path.  Masking is harmless off the diagonal.
    double2_t buf[kSyPanel];
#pragma unroll 1
    for (int64_t cp = c0; cp < limit; cp += kSyPanel) {
      sy_load<false>(s, cp, buf);
      sy_compute<false, true>(s, cp, buf, n0, n1, tout, lane);
    }
  }
  if (s.live1) *reinterpret_cast<double2_t*>(npart + static_cast<int64_t>(blockIdx.y) * ldp + s.r) = double2_t{n0, n1};
  else if (s.live0) npart[static_cast<int64_t>(blockIdx.y) * ldp + s.r] = n0;
}

// y[i] = sum_g npart[g][i] + sum_w tpart[w][i] over the partials that exist for element i.
// Workgroup = 64 consecutive elements; lanes run along i (coalesced 512-byte rows), the 4 waves
// split the partial rows and are combined through LDS in a fixed order.
__global__ __launch_bounds__(kBlock) void symv_reduce_kernel(const double* __restrict__ npart,
                                                             const double* __restrict__ tpart, int64_t ldp,
                                                             int64_t n, int32_t nwave, double* __restrict__ y,
                                                             const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double sacc[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 64 + lane;
  double s = 0.0;
  if (i < n) {
    // N partials: column groups g whose tile exists for i's wave chunk: g*kSyCols <= w0 + 127
    const int64_t w0 = (i / kSyWaveRows) * kSyWaveRows;
    const int32_t gmax = static_cast<int32_t>((w0 + kSyWaveRows - 1) / kSyCols);
    const int32_t glast = static_cast<int32_t>((n - 1) / kSyCols);
    const int32_t gend = gmax < glast ? gmax : glast;
    for (int32_t g = wid; g <= gend; g += 4) s += npart[static_cast<int64_t>(g) * ldp + i];
    // T partials: wave chunks w that processed the panel containing column i: w*128 + 127 >= panel start
    const int32_t wmin = static_cast<int32_t>(((i / kSyPanel) * kSyPanel) / kSyWaveRows);
    for (int32_t w = wmin + wid; w < nwave; w += 4) s += tpart[static_cast<int64_t>(w) * ldp + i];
  }
  sacc[wid][lane] = s;
  __syncthreads();
  if (wid == 0 && i < n) y[i] = ((sacc[0][lane] + sacc[1][lane]) + sacc[2][lane]) + sacc[3][lane];
}

SymvPlan symv_plan(int64_t n, int64_t ld) {
  SymvPlan p{};
  p.n = n;
  p.ld = ld;
  p.ldp = round_up(n, 2);
  p.nrow = static_cast<int32_t>(ceil_div(n, kSyWaveRows));  // tpart rows (wave chunks)
  p.ncol = static_cast<int32_t>(ceil_div(n, kSyCols));      // npart rows (column groups)
  return p;
}

void launch_symv_lower(const SymvPlan& p, const double* M, const double* x, double* npart, double* tpart, double* y,
                       const Ctrl* ctrl, hipStream_t stream) {
  dim3 grid(static_cast<unsigned>(p.nrow), static_cast<unsigned>(p.ncol));
  // Measured at n = 10^4 (MI355X): 104-114 us for the 400 MB lower triangle at 8-12 resident waves per
  // CU, flat in occupancy (an LDS-padding sweep moved it by < 4 %) -> ~3.7 TB/s; with the 8 us reduce
  // the x-solve takes 122 us against 141 us for the full 800 MB column-dot GEMV.
  hipLaunchKernelGGL(symv_lower_kernel, grid, dim3(kWave), 0, stream, M, p.n, p.ld, x, npart, tpart, p.ldp, ctrl);
  const int64_t blocks = ceil_div(p.n, 64);
  hipLaunchKernelGGL(symv_reduce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, npart, tpart,
                     p.ldp, p.n, p.nrow, y, ctrl);
}

}  // namespace admm
