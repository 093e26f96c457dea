// symv.hip -- y = M*x for a SYMMETRIC M reading only its lower triangle (half the bytes of a GEMV).
// Used for the explicit-inverse x-update  x = inv(D'D + rho I) * y  (getProxOps.m:1200 semantics).
//
// Every element M[r,j] (r >= j) is loaded once and used twice:
//   N-part  yN[r] += M[r,j]*x[j]        register accumulation per row (like gemv_n)
//   T-part  yT[j] += M[r,j]*x[r], r > j per-column sums over the rows (like gemv_t)
// A WAVE is the unit of work (128 rows x 256 columns, 16-byte loads, lane owns a row pair) and
// never synchronises with the other waves of its workgroup: the T-part column sums of an
// 8-column panel are combined inside the wave by a 10-shuffle reduce-scatter; with 4+ waves per
// SIMD the loads of other waves cover the reduction.
// Partials (npart per 256-column group, tpart per 128-row wave chunk) are added by a second small
// kernel in a fixed order: bitwise reproducible, no float atomics.
#include "kernels.h"

namespace admm {

typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int kSyWaveRows = 128;  // rows per wave (64 lanes x 2)
constexpr int kSyBlkRows = 128;   // rows per workgroup: ONE wave, so the triangular tile set load-balances finely
constexpr int kSyCols = 128;      // columns per workgroup
constexpr int kSyPanel = 16;      // columns per panel = loads in flight per lane = T-part accumulators

// reduce-scatter of 16 per-lane values over the 64 lanes of a wave: on return every lane holds the
// wave-wide sum of element *col (its lane bits 5..2 select the column); 15 + 2 shuffles.
__device__ __forceinline__ double reduce_scatter16(const double (&t)[kSyPanel], int lane, int* col) {
  double a8[8], a4[4], a2[2];
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double keep = b5 ? t[k + 8] : t[k];
    const double send = b5 ? t[k] : t[k + 8];
    a8[k] = keep + __shfl_xor(send, 32, 64);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double keep = b4 ? a8[k + 4] : a8[k];
    const double send = b4 ? a8[k] : a8[k + 4];
    a4[k] = keep + __shfl_xor(send, 16, 64);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double keep = b3 ? a4[k + 2] : a4[k];
    const double send = b3 ? a4[k] : a4[k + 2];
    a2[k] = keep + __shfl_xor(send, 8, 64);
  }
  const double keep = b2 ? a2[1] : a2[0];
  const double send = b2 ? a2[0] : a2[1];
  double r = keep + __shfl_xor(send, 4, 64);
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  *col = (b5 ? 8 : 0) + (b4 ? 4 : 0) + (b3 ? 2 : 0) + (b2 ? 1 : 0);
  return r;
}

__global__ __launch_bounds__(kWave) void symv_lower_kernel(const double* __restrict__ M, int64_t n, int64_t ld,
                                                            const double* __restrict__ x, double* __restrict__ npart,
                                                            double* __restrict__ tpart, int64_t ldp,
                                                            const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t w0 = static_cast<int64_t>(blockIdx.x) * kSyBlkRows + wid * kSyWaveRows;  // wave's first row
  const int64_t c0 = static_cast<int64_t>(blockIdx.y) * kSyCols;
  if (w0 >= n || w0 + kSyWaveRows - 1 < c0) return;  // nothing of the lower triangle in this wave's tile
  const int64_t r = w0 + 2 * lane;  // this lane's row pair (r, r+1)
  const bool live0 = r < n, live1 = r + 1 < n;
  const double xr0 = live0 ? x[r] : 0.0, xr1 = live1 ? x[r + 1] : 0.0;
  const bool diag = w0 < c0 + kSyCols;  // tile intersects the diagonal: mask element-wise
  const int64_t cend = (c0 + kSyCols < n) ? c0 + kSyCols : n;
  const int64_t wave_chunk = w0 / kSyWaveRows;
  double n0 = 0.0, n1 = 0.0;
#pragma unroll 1
  for (int64_t cp = c0; cp < cend; cp += kSyPanel) {
    if (w0 + kSyWaveRows - 1 < cp) break;  // the remaining panels lie strictly above the diagonal
    double tacc[kSyPanel];
    double2_t d[kSyPanel];
#pragma unroll
    for (int k = 0; k < kSyPanel; ++k) {
      const int64_t j = cp + k;
      d[k] = double2_t{0.0, 0.0};
      if (j < cend) {
        if (live1) d[k] = *reinterpret_cast<const double2_t*>(M + r + j * ld);
        else if (live0) d[k].x = M[r + j * ld];
      }
    }
#pragma unroll
    for (int k = 0; k < kSyPanel; ++k) {
      const int64_t j = cp + k;
      const double xj = (j < cend) ? x[j] : 0.0;  // wave-uniform -> scalar load
      double a0 = d[k].x, a1 = d[k].y;
      double t0 = a0, t1 = a1;
      if (diag) {  // lower triangle only: N-part j <= row, T-part row > j
        t0 = (r > j) ? a0 : 0.0;
        t1 = (r + 1 > j) ? a1 : 0.0;
        a0 = (j <= r) ? a0 : 0.0;
        a1 = (j <= r + 1) ? a1 : 0.0;
      }
      tacc[k] = __builtin_fma(t0, xr0, t1 * xr1);
      n0 = __builtin_fma(a0, xj, n0);
      n1 = __builtin_fma(a1, xj, n1);
    }
    int col;
    const double s = reduce_scatter16(tacc, lane, &col);
    if ((lane & 3) == 0 && cp + col < cend) tpart[wave_chunk * ldp + cp + col] = s;
  }
  if (live1) *reinterpret_cast<double2_t*>(npart + static_cast<int64_t>(blockIdx.y) * ldp + r) = double2_t{n0, n1};
  else if (live0) npart[static_cast<int64_t>(blockIdx.y) * ldp + r] = n0;
}

// y[i] = sum_g npart[g][i] + sum_w tpart[w][i] over the partials that exist for element i.
// Four lanes per element split the partial rows, then combine in a fixed order.
__global__ __launch_bounds__(kBlock) void symv_reduce_kernel(const double* __restrict__ npart,
                                                             const double* __restrict__ tpart, int64_t ldp,
                                                             int64_t n, int32_t nwave, double* __restrict__ y,
                                                             const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  const int sub = threadIdx.x & 3;
  for (int64_t i = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 2; i < n;
       i += (static_cast<int64_t>(gridDim.x) * kBlock) >> 2) {
    double s = 0.0;
    // N partials: column groups g with g*256 <= (first row of i's wave chunk) + 127
    const int64_t w0 = (i / kSyWaveRows) * kSyWaveRows;
    const int32_t gmax = static_cast<int32_t>((w0 + kSyWaveRows - 1) / kSyCols);
    const int32_t glast = static_cast<int32_t>((n - 1) / kSyCols);
    for (int32_t g = sub; g <= gmax && g <= glast; g += 4) s += npart[static_cast<int64_t>(g) * ldp + i];
    // T partials: wave chunks w >= 2*(i/256) whose rows reach column i's group, and whose panel
    // containing i was not skipped: w*128 + 127 >= (i/32)*32
    const int32_t wmin_group = static_cast<int32_t>((i / kSyCols) * kSyCols / kSyWaveRows);
    const int32_t wmin_panel = static_cast<int32_t>(((i / kSyPanel) * kSyPanel) / kSyWaveRows);
    const int32_t wmin = wmin_group > wmin_panel ? wmin_group : wmin_panel;
    for (int32_t w = wmin + sub; w < nwave; w += 4) s += tpart[static_cast<int64_t>(w) * ldp + i];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (sub == 0) y[i] = s;
  }
}

SymvPlan symv_plan(int64_t n, int64_t ld) {
  SymvPlan p{};
  p.n = n;
  p.ld = ld;
  p.ldp = round_up(n, 2);
  p.nrow = static_cast<int32_t>(ceil_div(n, kSyWaveRows));  // tpart rows
  p.ncol = static_cast<int32_t>(ceil_div(n, kSyCols));      // npart rows
  return p;
}

void launch_symv_lower(const SymvPlan& p, const double* M, const double* x, double* npart, double* tpart, double* y,
                       const Ctrl* ctrl, hipStream_t stream) {
  dim3 grid(static_cast<unsigned>(ceil_div(p.n, kSyBlkRows)), static_cast<unsigned>(p.ncol));
  hipLaunchKernelGGL(symv_lower_kernel, grid, dim3(kWave), 0, stream, M, p.n, p.ld, x, npart, tpart, p.ldp, ctrl);
  int64_t blocks = ceil_div(4 * p.n, kBlock);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(symv_reduce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, npart, tpart,
                     p.ldp, p.n, p.nrow, y, ctrl);
}

}  // namespace admm
