// consensus.hip -- vector kernels of consensus lasso (getProxOps.m:1217-1343).  Each slice k keeps
// its own x_k, u_k and cached factor; the consensus variable z couples them through the means
// of x_k and u_k (one all-reduce of 2n doubles when the slices live on several GPUs, X1 in SURVEY).
#include <cstdlib>

#include "consensus.h"
#include "loop_kernels.h"

namespace admm {

// sums[0][i] = sum_k x_k[i], sums[1][i] = sum_k u_k[i]   (getProxOps.m:1281-1284, slice order)
// qpart (sharded runs, else null): block partials of q = sum_k ||x_k - c||^2 about the PREVIOUS mean c = xaveprev,
// which every rank knows before the exchange.  With it lassonorms' first value (getProxOps.m:1338-1340)
//     sum_k ||x_k - xave||^2 = q - N*||xave - c||^2
// needs no second collective: q travels in the tail of the vector all-reduce, and ||xave - c||^2 is the dual
// residual sum every rank computes anyway.  Both terms shrink together as the iteration converges, so the
// subtraction loses about a digit (the naive sum ||x_k||^2 - N||xave||^2 would lose all of them).
__global__ __launch_bounds__(kBlock) void cons_sum_kernel(int64_t n, int64_t ldn, int32_t K,
                                                          const double* __restrict__ X,
                                                          const double* __restrict__ U, double* __restrict__ sums,
                                                          const double* __restrict__ center,
                                                          double* __restrict__ qpart,
                                                          const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ double scratch[4];
  double q = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    double sx = 0.0, su = 0.0;
    const double c = qpart ? center[i] : 0.0;
    for (int32_t k = 0; k < K; ++k) {
      const double xk = X[k * ldn + i];
      su = su + U[k * ldn + i];
      sx = sx + xk;
      const double d = xk - c;
      q = __builtin_fma(d, d, q);
    }
    sums[i] = sx;
    sums[ldn + i] = su;
  }
  if (qpart) {
    const double t = block_sum(q, scratch);
    if (threadIdx.x == 0) qpart[blockIdx.x] = t;
  }
}

// The same sums straight from the partial rows the lower-triangle x-solves left behind (symv.hip): x_k[i] is
// assembled here -- ntile + 1 rows per slice, split over the four slots of a 512-thread workgroup, all loads of a
// round issued together -- and stored for the update kernel, so the K symv_reduce launches (and their boundaries)
// disappear.  npart / tpart: [K][pstride] with rows of ldp elements.
constexpr int kCgTile = 128, kCgSlots = 4, kCgRows = 20;
__global__ __launch_bounds__(kCgTile* kCgSlots) void cons_gather_sum_kernel(
    int64_t n, int64_t ldn, int32_t K, const double* __restrict__ npart, const double* __restrict__ tpart,
    int64_t pstride, int64_t ldp, int32_t ntile, double* __restrict__ X, const double* __restrict__ U,
    double* __restrict__ sums, const double* __restrict__ center, double* __restrict__ qpart,
    const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ double part[kCgSlots - 1][kCgTile];
  __shared__ double scratch[kCgTile * kCgSlots / 64];
  const int e = threadIdx.x & (kCgTile - 1), slot = threadIdx.x >> 7;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kCgTile + e;
  const int64_t ic = i < n ? i : n - 1;
  const int32_t d = static_cast<int32_t>(ic / kCgTile);
  const int32_t P = ntile + 1, per = (P + kCgSlots - 1) / kCgSlots;
  const int32_t p0 = slot * per, p1 = (p0 + per < P) ? p0 + per : P;
  double sx = 0.0, su = 0.0, q = 0.0;
  const double c = (qpart && slot == 0) ? center[ic] : 0.0;
  for (int32_t k = 0; k < K; ++k) {
    const double* __restrict__ nk = npart + static_cast<int64_t>(k) * pstride;
    const double* __restrict__ tk = tpart + static_cast<int64_t>(k) * pstride;
    double s = 0.0;
    for (int32_t p = p0; p < p1; p += kCgRows) {
      double v[kCgRows];
#pragma unroll
      for (int r = 0; r < kCgRows; ++r) {
        const int32_t pq = (p + r < p1) ? p + r : p1 - 1;
        const double* src = (pq <= d) ? nk + static_cast<int64_t>(pq) * ldp : tk + static_cast<int64_t>(pq - 1) * ldp;
        v[r] = src[ic];
      }
#pragma unroll
      for (int r = 0; r < kCgRows; ++r)
        if (p + r < p1) s += v[r];
    }
    if (p0 >= P) s = 0.0;
    __syncthreads();  // part[] of the previous slice has been consumed
    if (slot > 0) part[slot - 1][e] = s;
    __syncthreads();
    if (slot == 0 && i < n) {
      const double xk = ((s + part[0][e]) + part[1][e]) + part[2][e];
      X[k * ldn + i] = xk;
      su = su + U[k * ldn + i];
      sx = sx + xk;
      const double dd = xk - c;
      q = __builtin_fma(dd, dd, q);
    }
  }
  if (slot == 0 && i < n) {
    sums[i] = sx;
    sums[ldn + i] = su;
  }
  if (qpart) {
    const double t = block_sum(q, scratch);
    if (threadIdx.x == 0) qpart[blockIdx.x] = t;
  }
}

int launch_cons_gather_sum(int64_t n, int64_t ldn, int32_t K, const double* npart, const double* tpart, int64_t pstride,
                           int64_t ldp, int32_t ntile, double* X, const double* U, double* sums, const double* center,
                           double* qpart, const Ctrl* ctrl, hipStream_t stream) {
  const int64_t blocks = ceil_div(n, kCgTile);
  hipLaunchKernelGGL(cons_gather_sum_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kCgTile * kCgSlots), 0, stream, n,
                     ldn, K, npart, tpart, pstride, ldp, ntile, X, U, sums, center, qpart, ctrl);
  return qpart ? static_cast<int>(blocks) : 0;
}

// returns the number of qpart blocks written (0 when qpart is null)
int launch_cons_sum(int64_t n, int64_t ldn, int32_t K, const double* X, const double* U, double* sums,
                    const double* center, double* qpart, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(n, kBlock);
  if (blocks > kMaxPartBlocks) blocks = kMaxPartBlocks;
  hipLaunchKernelGGL(cons_sum_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, n, ldn, K, X, U,
                     sums, center, qpart, ctrl);
  return qpart ? static_cast<int>(blocks) : 0;
}

// z-update + per-slice u-update + every partial sum admm / lassonorms need:
//   getProxOps.m:1286-1298 (means, soft threshold with lambda/(rho*N), u_k += x_k - z),
//   1312-1326 (altu: mean of the updated u_k = uave + xave - z), 1335-1343 (squared norms).
__global__ __launch_bounds__(kBlock) void cons_update_kernel(ConsArgs a, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t it = ctrl->iter;
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  const double Nd = static_cast<double>(a.Ntot);
  const double t = a.lambda / (a.rho * Nd);  // q11
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double xave = a.sums[i] / Nd;
    const double uave = a.sums[a.ldn + i] / Nd;
    const double v = uave + xave;
    const double q = fabs(v) - t;
    const double p = q > 0.0 ? q : 0.0;
    const double z = (v > 0.0) ? p : ((v < 0.0) ? -p : 0.0 * p);
    double dev = 0.0;
    for (int32_t k = 0; k < a.K; ++k) {
      const double xk = a.X[k * a.ldn + i];
      const double uk = a.U[k * a.ldn + i] + (xk - z);
      a.U[k * a.ldn + i] = uk;
      a.Y[k * a.ldn + i] = a.rho * (z - uk) + a.Dts[k * a.ldn + i];  // the next x-update's y_k (getProxOps.m:1240)
      const double d = xk - xave;
      dev += d * d;
    }
    const double xprev = a.xave[i];
    const double ub_old = a.ubar[i];
    const double ub = (uave + xave) - z;  // mean over ALL slices of u_k + (x_k - z)
    acc[S_R2] += dev;                     // lassonorms v(1), this rank's slices
    acc[S_AX2] += xave * xave;            // ||Ax||, A = 1, x = mean x_k (getProxOps.m:1259)
    const double dx = xave - xprev;
    acc[S_G2] += dx * dx;                 // lassonorms v(2) = N*rho^2*||xave - xaveprev||^2
    acc[S_U2] += ub * ub;
    const double du = ub - ub_old;
    acc[S_DU2] += du * du;
    a.zc[i] = z;
    a.xaveprev[i] = xprev;
    a.xave[i] = xave;
    a.ubar[i] = ub;
    if (a.xhist) a.xhist[it * a.n + i] = xave;
    if (a.zhist) a.zhist[it * a.n + i] = 0.0;  // q9: zminParallelLASSO hands zeros back to admm
    if (a.uhist) a.uhist[it * a.n + i] = ub;
  }
  __shared__ double sred[4][S_COUNT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) {
    const double w = wave_sum(acc[s]);
    if (lane == 0) sred[wid][s] = w;
  }
  __syncthreads();
  if (threadIdx.x < S_COUNT) {
    const int s = threadIdx.x;
    a.part[s * kMaxPartBlocks + blockIdx.x] = ((sred[0][s] + sred[1][s]) + sred[2][s]) + sred[3][s];
  }
}

// Unsharded consensus iteration tail as ONE launch: the slices' x_k are assembled from the partial rows the batched
// lower-triangle x-solve left behind (as cons_gather_sum_kernel does), and the means, z, the u_k, the next right-hand
// sides y_k and every partial sum follow in the same workgroup (as cons_update_kernel does) -- the K x n sums never
// travel through memory and one launch boundary disappears.  Round 2 ran the gather as 79 workgroups (128 elements x 4
// row slots at n = 10^4: a third of the CUs, 22.9 us for 52 MB); here a workgroup takes 64 elements (round 3, first
// form: 32) and its 16 slots are dealt to (slice, row range) pairs: 157 workgroups of 1024 threads at n = 10^4, 20 loads
// in flight per thread -- 21 us for gather + update (rocprof; 16 elements x 32 slots, one round of loads per thread on 625 workgroups: 25.7 us -- the
// rows' 128-byte segments are too short for the memory system), and 579 -> 570 us per iteration with the launch gone.
template <int kCuTile, int kCuSlots>
__global__ __launch_bounds__(kCuTile* kCuSlots) void cons_gather_update_kernel(
    ConsArgs a, const double* __restrict__ npart, const double* __restrict__ tpart, int64_t pstride, int64_t ldp,
    int32_t ntile, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ double part[kCuSlots][kCuTile];
  __shared__ double xs[kCuSlots][kCuTile], us[kCuSlots][kCuTile];
  __shared__ double sred[kCuTile * kCuSlots / 64][S_COUNT];
  const int64_t it = ctrl->iter;
  const int e = threadIdx.x & (kCuTile - 1), slot = threadIdx.x / kCuTile;
  const int32_t K = a.K;                  // <= kCuSlots (the launcher checks)
  const int32_t SUB = kCuSlots / K;       // row ranges per slice
  const int32_t kk = slot / SUB, sb = slot - kk * SUB;
  const bool active = kk < K;
  const int32_t P = ntile + 1, per = (P + SUB - 1) / SUB;
  const int32_t p0 = sb * per, p1 = (p0 + per < P) ? p0 + per : P;
  const double Nd = static_cast<double>(a.Ntot);
  const double t = a.lambda / (a.rho * Nd);  // q11
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  const int64_t ntiles = (a.n + kCuTile - 1) / kCuTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t i = tile * kCuTile + e;
    const int64_t ic = i < a.n ? i : a.n - 1;
    const int32_t d = static_cast<int32_t>(ic / kCgTile);  // the x-solve's tile row of this element
    // 1. this slot's share of the partial rows of its slice
    double s = 0.0;
    if (active) {
      const double* __restrict__ nk = npart + static_cast<int64_t>(kk) * pstride;
      const double* __restrict__ tk = tpart + static_cast<int64_t>(kk) * pstride;
      for (int32_t p = p0; p < p1; p += kCgRows) {
        double v[kCgRows];
#pragma unroll
        for (int r = 0; r < kCgRows; ++r) {
          const int32_t pq = (p + r < p1) ? p + r : p1 - 1;
          const double* src = (pq <= d) ? nk + static_cast<int64_t>(pq) * ldp : tk + static_cast<int64_t>(pq - 1) * ldp;
          v[r] = src[ic];
        }
#pragma unroll
        for (int r = 0; r < kCgRows; ++r)
          if (p + r < p1) s += v[r];
      }
    }
    const bool lead = active && sb == 0;  // the thread that owns (element, slice)
    const double uold = lead ? a.U[kk * a.ldn + ic] : 0.0;
    const double dts = lead ? a.Dts[kk * a.ldn + ic] : 0.0;
    part[slot][e] = s;
    __syncthreads();
    // 2. x_k of every slice
    double xk = 0.0;
    if (lead) {
      xk = part[slot][e];
      for (int32_t j = 1; j < SUB; ++j) xk += part[slot + j][e];
      xs[kk][e] = xk;
      us[kk][e] = uold;
    }
    __syncthreads();
    // 3. means and z (every thread, slice order: getProxOps.m:1281-1292)
    double sx = 0.0, su = 0.0;
    for (int32_t k = 0; k < K; ++k) {
      su = su + us[k][e];
      sx = sx + xs[k][e];
    }
    const double xave = sx / Nd, uave = su / Nd;
    const double v = uave + xave;
    const double q = fabs(v) - t;
    const double pz = q > 0.0 ? q : 0.0;
    const double z = (v > 0.0) ? pz : ((v < 0.0) ? -pz : 0.0 * pz);
    // 4. u_k += x_k - z, the next y_k   (getProxOps.m:1296-1298, 1240)
    if (lead && i < a.n) {
      const double uk = uold + (xk - z);
      a.X[kk * a.ldn + i] = xk;
      a.U[kk * a.ldn + i] = uk;
      a.Y[kk * a.ldn + i] = a.rho * (z - uk) + dts;
      const double dd = xk - xave;
      acc[S_R2] += dd * dd;  // lassonorms v(1)
    }
    // 5. the element's own state (getProxOps.m:1312-1326, 1335-1343)
    if (slot == 0 && i < a.n) {
      const double xprev = a.xave[i];
      const double ub_old = a.ubar[i];
      const double ub = (uave + xave) - z;  // mean over all slices of u_k + (x_k - z)
      acc[S_AX2] += xave * xave;
      const double dx = xave - xprev;
      acc[S_G2] += dx * dx;
      acc[S_U2] += ub * ub;
      const double du = ub - ub_old;
      acc[S_DU2] += du * du;
      a.zc[i] = z;
      a.xaveprev[i] = xprev;
      a.xave[i] = xave;
      a.ubar[i] = ub;
      if (a.xhist) a.xhist[it * a.n + i] = xave;
      if (a.zhist) a.zhist[it * a.n + i] = 0.0;  // q9
      if (a.uhist) a.uhist[it * a.n + i] = ub;
    }
    __syncthreads();  // part / xs / us are reused by the next tile
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) {
    const double w = wave_sum(acc[s]);
    if (lane == 0) sred[wid][s] = w;
  }
  __syncthreads();
  if (threadIdx.x < S_COUNT) {
    const int s = threadIdx.x;
    double w = sred[0][s];
#pragma unroll
    for (int q = 1; q < kCuTile * kCuSlots / 64; ++q) w += sred[q][s];
    a.part[s * kMaxPartBlocks + blockIdx.x] = w;
  }
}

bool cons_gather_update_ok(const ConsArgs& a) { return a.K >= 1 && a.K <= 16; }

void launch_cons_gather_update(const ConsArgs& a, const double* npart, const double* tpart, int64_t pstride, int64_t ldp,
                               int32_t ntile, const Ctrl* ctrl, int* nblk_out, hipStream_t stream) {
  // elements x (slice, row-range) slots per workgroup, for the 52 MB of partial rows of 8 slices at n = 10^4:
  // 32 x 16: 25.1 us, 64 x 16: 20.9 us (default), 128 x 8 (one slot per slice, 80 rows per thread): 25.4 us
  static const int tile = [] {
    const char* ev = std::getenv("ADMM_CONS_TILE");
    return (ev && std::atoi(ev) == 32) ? 32 : 64;
  }();
  int64_t blocks = ceil_div(a.n, int64_t{tile});
  if (blocks > kMaxPartBlocks) blocks = kMaxPartBlocks;
  *nblk_out = static_cast<int>(blocks);
  const dim3 grid(static_cast<unsigned>(blocks));
  if (tile == 64)
    hipLaunchKernelGGL((cons_gather_update_kernel<64, 16>), grid, dim3(1024), 0, stream, a, npart, tpart, pstride, ldp, ntile,
                       ctrl);
  else
    hipLaunchKernelGGL((cons_gather_update_kernel<32, 16>), grid, dim3(512), 0, stream, a, npart, tpart, pstride, ldp, ntile,
                       ctrl);
}

void launch_cons_update(const ConsArgs& a, const Ctrl* ctrl, int* nblk_out, hipStream_t stream) {
  int64_t blocks = ceil_div(a.n, kBlock);
  if (blocks > kMaxPartBlocks) blocks = kMaxPartBlocks;
  if (blocks < 1) blocks = 1;
  *nblk_out = static_cast<int>(blocks);
  hipLaunchKernelGGL(cons_update_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a, ctrl);
}

}  // namespace admm
