// trsv.hip -- x = L' \ (L \ y): the cached-factor x-update (getProxOps.m:1200 `U \ (L \ y)`,
// 1514 `Rt \ (R \ .)`, 1455, 1247) on a dense lower Cholesky factor, as blocked substitution.
//
// A triangular solve is a dependent chain; what bounds it on this part is the number of dependent steps, not
// bytes (round 1: 2*n/64 launches of 6 us = 0.05 of the HBM roofline).  So the chain is cut into K COARSE blocks
// (16 tiles of 128 = 2048 rows; K = 5 at n = 10^4) and each step is one bandwidth-bound launch:
//
//   forward  step k:  w_k = inv(L_kk) y_k ;   y_below -= L_below,k w_k
//   backward step k:  x_k = inv(L_kk)' w_k ;  w_above -= L_k,above' x_k
//
// With the diagonal-block inverse folded into its panel at build time (a block Gauss transform),
//   Fm[:, block k] = [ inv(L_kk) ; -L_below,k inv(L_kk) ]      (lower, block column k)
//   Um[:, block k] = [ -L_k,above' inv(L_kk)' ; inv(L_kk)' ]   (upper, block column k)
// a step is ONE column-panel GEMV  out = M[:, block k] * in_k  whose result rows inside the block are the
// solution block and whose other rows are added to the running right-hand side.  Both sweeps use the same
// kernel: lanes along rows (column-major => 16-byte coalesced non-temporal loads), register accumulation per
// row, one workgroup per 128 x 128 tile with 1..16 waves splitting its columns (chosen per step so that the grid
// keeps >= ~1500 waves), the input block broadcast inside the wave by v_readlane.  Each tile writes ONE partial
// row (its 128 columns); nothing synchronises inside a launch: the partial sums of step k are folded by step
// k+1 itself -- every tile sums the <= 16 partial rows of its own input columns while its matrix panel is already
// in flight, the first column tile of a row tile also carries the running right-hand side of its rows, and a few
// extra one-wave workgroups write out the previous solution block -- in fixed order: deterministic, no float
// atomics, no reduce launches.  2K + 1 launches per solve pair (K = 5 at n = 10^4); the explicit inverses are
// only of the K diagonal blocks (forward error eps*cond(L_kk), like any blocked TRSV with pre-inverted diagonal
// blocks), so this is the numerically safe fallback of the `inverse` form.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace admm {

typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int kTsTile = 128;   // rows per wave = tile granularity of the blocks
constexpr int kTsPanel = 8;    // columns per load group
constexpr int kTsBlockTiles = 16;  // 2048-row blocks: K = 5 at n = 10^4 (per-launch fixed cost ~6 us: 10 -> 217 us, 16 -> 182 us, 27 -> 177 us per pair)

struct TriStepArgs {
  const double* M;         // Fm or Um, tile-packed: the 128 x 128 tiles of the lower (Fm) / upper (Um) triangle back to
                           // back, tile (R, C) at index R(R+1)/2 + C resp. C(C+1)/2 + R, column-major inside (ld 128)
  uint32_t ncached;        // tiles with index < ncached are read with default loads (they stay in the Infinity Cache from
                           // one solve to the next), the others non-temporally (symv.hip has the measurements)
  int64_t n;               // valid length of the vectors (caller vectors are not padded)
  const double* base_in;   // running right-hand side: rows outside the diagonal block carry base + partial sums
  double* base_out;
  double* diag_out;        // where the PREVIOUS step's solution block is written (w forward, x backward)
  const double* Pprev;     // [tiles of the previous block][ldp] column-tile partials of the previous step
  double* Pcur;            // same for this step
  int64_t ldp;
  int32_t dt0, dt1;        // this step's diagonal block = its columns (tile range)
  int32_t rt0, rt1;        // this step's panel rows (tile range)
  int32_t upper;           // 0 = forward sweep on Fm (lower), 1 = backward sweep on Um (upper)
  int32_t pd0, pd1;        // previous step: diagonal block,
  int32_t pr0, pr1;        //                panel rows (tiles that have partials; empty range = none),
  int32_t pupper;          //                orientation
  int32_t x_use_base;      // the input block is base_in + partial sums (0: partial sums only)
  int32_t ext0, ext1;      // row tiles of the previous solution block to write to diag_out (grid rows beyond the panel)
  const Ctrl* ctrl;
};

// partial-sum range of row tile T in the previous step
__device__ __forceinline__ void ts_prev_range(const TriStepArgs& a, int32_t T, int32_t& q0, int32_t& q1) {
  q0 = 0;
  q1 = 0;
  if (T < a.pr0 || T >= a.pr1) return;
  q1 = a.pd1 - a.pd0;
  if (T >= a.pd0 && T < a.pd1) {  // triangular diagonal block: only the tiles that exist
    if (!a.pupper) q1 = T - a.pd0 + 1;
    else q0 = T - a.pd0;
  }
}

// sum over the previous step's partials of the element pair at index e (fixed order; the loads of a group of
// sixteen are issued together: one memory round trip for a whole block of partials)
__device__ __forceinline__ double2_t ts_sum_prev(const TriStepArgs& a, int32_t q0, int32_t q1, int64_t e) {
  constexpr int G = 16;
  double2_t t{0.0, 0.0};
  const double* __restrict__ p = a.Pprev + e;
  for (int32_t q = q0; q < q1; q += G) {
    double2_t v[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const int32_t qq = (q + k < q1) ? q + k : q1 - 1;  // clamped: loads stay unconditional
      v[k] = *reinterpret_cast<const double2_t*>(p + static_cast<int64_t>(qq) * a.ldp);
    }
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (q + k < q1) t += v[k];
  }
  return t;
}

__device__ __forceinline__ double2_t ts_load_vec(const double* __restrict__ b, int64_t e, int64_t n) {
  double2_t v{0.0, 0.0};
  if (e + 1 < n) v = double2_t{b[e], b[e + 1]};  // caller vectors are only 8-byte aligned
  else if (e < n) v.x = b[e];
  return v;
}

__device__ __forceinline__ void ts_store_vec(double* __restrict__ b, int64_t e, int64_t n, double2_t v) {
  if (e < n) b[e] = v.x;
  if (e + 1 < n) b[e + 1] = v.y;
}

__device__ __forceinline__ double ts_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <bool NT>
__device__ __forceinline__ void ts_load(const double* __restrict__ Mr, int64_t ld, int64_t c, double2_t (&d)[kTsPanel]) {
#pragma unroll
  for (int k = 0; k < kTsPanel; ++k) d[k] = load2<NT>(Mr + (c + k) * ld);
}

// xp: lane l holds the input pair of this wave's columns (2l, 2l+1); co = first column of the panel within the wave
__device__ __forceinline__ void ts_fma(double2_t xp, int co, const double2_t (&d)[kTsPanel], double& a0, double& a1) {
#pragma unroll
  for (int k = 0; k < kTsPanel; ++k) {
    const double xj = ts_readlane((k & 1) ? xp.y : xp.x, (co + k) >> 1);
    a0 = __builtin_fma(d[k].x, xj, a0);
    a1 = __builtin_fma(d[k].y, xj, a1);
  }
}

// the tile itself: lc0 = this wave's first column inside the tile, cw0 = the same as a global column
template <int WAVES, bool NT>
__device__ __forceinline__ void tri_tile(const TriStepArgs& a, const double* __restrict__ Mr, int lc0, int64_t cw0,
                                         int64_t r, int32_t R, int32_t c, bool diag, int lane, int wave,
                                         double2_t (*red)[kWave]) {
  constexpr int WC = kTsTile / WAVES;
  double2_t bufA[kTsPanel], bufB[kTsPanel];
  ts_load<NT>(Mr, kTsTile, lc0, bufA);  // the matrix does not depend on the vectors: in flight during the prologue
  // input block of this wave's columns: (base +) the previous step's partial sums, pair (2l, 2l+1) in lane l
  double2_t xp{0.0, 0.0};
  {
    int32_t q0, q1;
    ts_prev_range(a, a.dt0 + c, q0, q1);
    const int64_t e = cw0 + 2 * (lane < WC / 2 ? lane : 0);
    xp = ts_sum_prev(a, q0, q1, e);
    if (a.x_use_base) xp += ts_load_vec(a.base_in, e, a.n);
  }
  // rows outside the diagonal block carry on: base_out = base_in + previous partial sums (first column tile only)
  if (!diag && c == 0 && wave == 0) {
    int32_t q0, q1;
    ts_prev_range(a, R, q0, q1);
    if (q1 > q0 || a.base_out != a.base_in)
      ts_store_vec(a.base_out, r, a.n, ts_load_vec(a.base_in, r, a.n) + ts_sum_prev(a, q0, q1, r));
  }
  double a0 = 0.0, a1 = 0.0;
  if (WC > kTsPanel) {
#pragma unroll 1
    for (int co = 0; co < WC; co += 2 * kTsPanel) {
      ts_load<NT>(Mr, kTsTile, lc0 + co + kTsPanel, bufB);
      ts_fma(xp, co, bufA, a0, a1);
      if (co + 2 * kTsPanel < WC) ts_load<NT>(Mr, kTsTile, lc0 + co + 2 * kTsPanel, bufA);
      ts_fma(xp, co + kTsPanel, bufB, a0, a1);
    }
  } else {
    ts_fma(xp, 0, bufA, a0, a1);
  }
  double2_t acc{a0, a1};
  if (WAVES > 1) {
    red[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    acc = red[0][lane];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) acc += red[w][lane];
  }
  *reinterpret_cast<double2_t*>(a.Pcur + static_cast<int64_t>(c) * a.ldp + r) = acc;
}

// One workgroup = one 128 x 128 tile, WAVES waves of 128 x (128 / WAVES) columns each.
template <int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * kWave) void tri_step_kernel(TriStepArgs a) {
  // (the control block's address together with the tile path's arguments: one scalar round trip less in front of the
  // loads of a workgroup that lives for one tile)
  asm volatile("" ::"s"(a.M), "s"(a.ncached), "s"(a.n), "s"(a.Pprev), "s"(a.Pcur), "s"(a.ldp), "s"(a.dt0), "s"(a.dt1),
               "s"(a.rt0), "s"(a.rt1), "s"(a.upper), "s"(a.ctrl));
  if (a.ctrl && a.ctrl->stop) return;
  constexpr int WC = kTsTile / WAVES;  // columns per wave
  __shared__ double2_t red[WAVES > 1 ? WAVES : 1][kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t nrows = a.rt1 - a.rt0;
  const int32_t c = static_cast<int32_t>(blockIdx.y);
  if (static_cast<int32_t>(blockIdx.x) >= nrows) {  // write out the previous step's solution block
    if (c != 0 || wave != 0) return;
    const int32_t T = a.ext0 + static_cast<int32_t>(blockIdx.x) - nrows;
    int32_t q0, q1;
    ts_prev_range(a, T, q0, q1);
    const int64_t e = static_cast<int64_t>(T) * kTsTile + 2 * lane;
    ts_store_vec(a.diag_out, e, a.n, ts_sum_prev(a, q0, q1, e));
    return;
  }
  const int32_t R = a.rt0 + static_cast<int32_t>(blockIdx.x);  // row tile
  const bool diag = R >= a.dt0 && R < a.dt1;
  if (diag && (a.upper ? c < R - a.dt0 : c > R - a.dt0)) return;  // all-zero tile of the triangular block
  const int64_t r = static_cast<int64_t>(R) * kTsTile + 2 * lane;  // this lane's row pair
  const int64_t cw0 = static_cast<int64_t>(a.dt0 + c) * kTsTile + wave * WC;  // this wave's first column
  const uint32_t Ct = static_cast<uint32_t>(a.dt0 + c), Rt = static_cast<uint32_t>(R);
  const uint32_t lin = a.upper ? Ct * (Ct + 1u) / 2u + Rt : Rt * (Rt + 1u) / 2u + Ct;
  const double* __restrict__ Mr = a.M + static_cast<int64_t>(lin) * (kTsTile * kTsTile) + 2 * lane;
  if (NT && lin >= a.ncached) tri_tile<WAVES, true>(a, Mr, wave * WC, cw0, r, R, c, diag, lane, wave, red);
  else tri_tile<WAVES, false>(a, Mr, wave * WC, cw0, r, R, c, diag, lane, wave, red);
}

// x block of the last backward step: sum of its partials
__global__ __launch_bounds__(kWave) void tri_fold_kernel(TriStepArgs a) {
  if (a.ctrl && a.ctrl->stop) return;
  const int32_t T = a.ext0 + static_cast<int32_t>(blockIdx.x);
  int32_t q0, q1;
  ts_prev_range(a, T, q0, q1);
  const int64_t e = static_cast<int64_t>(T) * kTsTile + 2 * static_cast<int>(threadIdx.x);
  ts_store_vec(a.diag_out, e, a.n, ts_sum_prev(a, q0, q1, e));
}

// Um(j, i) = X(i, j) for the nb x nb lower-triangular block X (transposed copy of a diagonal block)
__global__ __launch_bounds__(kBlock) void ts_transpose_block_kernel(const double* __restrict__ X, int64_t ldx,
                                                                    double* __restrict__ U, int64_t ldu, int64_t nb) {
  __shared__ double T[32][33];
  const int64_t bi = blockIdx.x, bj = blockIdx.y;
  if (bi < bj) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int64_t i = bi * 32 + tx, j = bj * 32 + rr;
    T[rr][tx] = (i < nb && j < nb && i >= j) ? X[i + j * ldx] : 0.0;
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int64_t i = bj * 32 + tx, j = bi * 32 + rr;  // U(i, j) = X(j, i)
    if (i < nb && j < nb) U[i + j * ldu] = T[tx][rr];
  }
}

// padded column-major square -> tile-packed triangle; tile t = (bi, bj), bi >= bj, holds M(bi, bj) of the lower triangle
// (upper = 0) or M(bj, bi) of the upper one (upper = 1)
__global__ __launch_bounds__(kBlock) void ts_pack_kernel(const double* __restrict__ M, int64_t ld, double* __restrict__ P,
                                                         int upper) {
  const unsigned t = blockIdx.x;
  unsigned bi = static_cast<unsigned>((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (bi * (bi + 1u) / 2u > t) --bi;
  while ((bi + 1u) * (bi + 2u) / 2u <= t) ++bi;
  const unsigned bj = t - bi * (bi + 1u) / 2u;
  const int64_t tr = upper ? bj : bi, tc = upper ? bi : bj;  // tile row / column in M
  const double* src = M + tc * kTsTile * ld + tr * kTsTile;
  double* dst = P + static_cast<int64_t>(t) * (kTsTile * kTsTile);
  for (int e = 2 * threadIdx.x; e < kTsTile * kTsTile; e += 2 * kBlock)
    *reinterpret_cast<double2_t*>(dst + e) =
        *reinterpret_cast<const double2_t*>(src + static_cast<int64_t>(e / kTsTile) * ld + (e % kTsTile));
}

static int ts_block_tiles() {
  if (const char* f = std::getenv("ADMM_TRSV_BLOCK_TILES")) {
    const int v = std::atoi(f);
    if (v >= 1 && v <= 64) return v;
  }
  return kTsBlockTiles;
}

size_t trsv_plan_elems(int64_t n) {
  const int64_t npad = round_up(n, kTsTile);
  const int64_t ntile = npad / kTsTile;
  const int64_t bt = std::min<int64_t>(ts_block_tiles(), ntile);
  // Fm + Um (tile-packed triangles) + two partial buffers + v + w
  return static_cast<size_t>(ntile * (ntile + 1) * kTsTile * kTsTile + 2 * bt * npad + 2 * npad);
}

// L: n x n lower factor (upper part ignored); dinv64: its inverted 64x64 diagonal blocks.
// `buf` must hold trsv_plan_elems(n) doubles and is owned by the caller.
int trsv_build(const double* L, int64_t n, int64_t ldl, const double* dinv64, double* buf, TrsvPlan* plan,
               hipStream_t stream) {
  TrsvPlan& p = *plan;
  p.n = n;
  p.npad = round_up(n, kTsTile);
  p.ntile = static_cast<int32_t>(p.npad / kTsTile);
  p.bt = static_cast<int32_t>(std::min<int64_t>(ts_block_tiles(), p.ntile));
  p.nblk = static_cast<int32_t>(ceil_div(p.ntile, p.bt));
  p.bt = static_cast<int32_t>(ceil_div(p.ntile, p.nblk));  // balanced blocks
  p.ldm = p.npad;
  p.ldp = p.npad;
  const size_t msz = static_cast<size_t>(p.npad) * p.npad;                                // the dense squares they are built in
  const size_t psz = static_cast<size_t>(p.ntile) * (p.ntile + 1) / 2 * kTsTile * kTsTile;  // a packed triangle
  double* const packedF = buf;
  double* const packedU = buf + psz;
  p.P[0] = packedU + psz;
  p.P[1] = p.P[0] + static_cast<int64_t>(p.bt) * p.npad;
  p.v = p.P[1] + static_cast<int64_t>(p.bt) * p.npad;
  p.w = p.v + p.npad;
  p.streaming = stream_hint(static_cast<int64_t>(psz) * 2 * 8);
  // the split cache policy of symv.hip: both triangles are read once per solve pair and share the budget
  p.ncached = p.streaming ? static_cast<int64_t>(kSymvCacheBytes / 2 / (8 * kTsTile * kTsTile)) : INT64_MAX;
  ADMM_HIP_TRY(hipMemsetAsync(buf, 0, trsv_plan_elems(n) * sizeof(double), stream));
  double* dense = nullptr;  // Fm | Um as padded column-major squares: GEMM outputs, packed below
  ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dense), 2 * msz * sizeof(double)));
  ADMM_HIP_TRY(hipMemsetAsync(dense, 0, 2 * msz * sizeof(double), stream));
  p.Fm = dense;
  p.Um = dense + msz;
  for (int32_t k = 0; k < p.nblk; ++k) {
    const int64_t k0 = static_cast<int64_t>(k) * p.bt * kTsTile;
    const int64_t nb = std::min<int64_t>(static_cast<int64_t>(p.bt) * kTsTile, n - k0);
    if (nb <= 0) break;
    double* Xk = p.Fm + k0 + k0 * p.ldm;
    // X_k = inv(L_kk) into Fm's (already zero) diagonal block
    ADMM_TRY(trtri_lower_from_diag(L + k0 + k0 * ldl, nb, ldl, dinv64 + (k0 / 64) * 64 * 64, Xk, p.ldm, stream, false));
    const int64_t below = n - k0 - nb;
    if (below > 0)  // Fm_below = -L_below,k * X_k
      launch_gemm(0, 0, below, nb, nb, -1.0, L + (k0 + nb) + k0 * ldl, ldl, Xk, p.ldm, 0.0, p.Fm + (k0 + nb) + k0 * p.ldm,
                  p.ldm, false, stream);
    const unsigned tb = static_cast<unsigned>(ceil_div(nb, 32));
    hipLaunchKernelGGL(ts_transpose_block_kernel, dim3(tb, tb), dim3(kBlock), 0, stream, Xk, p.ldm,
                       p.Um + k0 + k0 * p.ldm, p.ldm, nb);
    if (k0 > 0)  // Um_above = -(L_k,above)' * X_k'
      launch_gemm(1, 1, k0, nb, nb, -1.0, L + k0, ldl, Xk, p.ldm, 0.0, p.Um + k0 * p.ldm, p.ldm, false, stream);
  }
  const unsigned ntri = static_cast<unsigned>(p.ntile) * static_cast<unsigned>(p.ntile + 1) / 2u;
  hipLaunchKernelGGL(ts_pack_kernel, dim3(ntri), dim3(kBlock), 0, stream, p.Fm, p.ldm, packedF, 0);
  hipLaunchKernelGGL(ts_pack_kernel, dim3(ntri), dim3(kBlock), 0, stream, p.Um, p.ldm, packedU, 1);
  const hipError_t se = hipStreamSynchronize(stream);
  (void)hipFree(dense);
  ADMM_HIP_TRY(se);
  p.Fm = packedF;
  p.Um = packedU;
  return ADMM_OK;
}

template <int WAVES>
static void ts_launch(const TriStepArgs& a, dim3 grid, bool nt, hipStream_t stream) {
  if (nt) hipLaunchKernelGGL((tri_step_kernel<WAVES, true>), grid, dim3(WAVES * kWave), 0, stream, a);
  else hipLaunchKernelGGL((tri_step_kernel<WAVES, false>), grid, dim3(WAVES * kWave), 0, stream, a);
}

static void ts_step(const TriStepArgs& a, bool nt, hipStream_t stream) {
  const int32_t rows = a.rt1 - a.rt0, cols = a.dt1 - a.dt0;
  // waves per 128 x 128 tile: as few as still give the grid ~1500 waves (HBM needs tens of KB in flight per CU)
  int waves = 1;
  while (waves < 16 && static_cast<int64_t>(rows) * cols * waves < 1536) waves <<= 1;
  const dim3 grid(static_cast<unsigned>(rows + (a.ext1 - a.ext0)), static_cast<unsigned>(cols));
  switch (waves) {
    case 1: ts_launch<1>(a, grid, nt, stream); break;
    case 2: ts_launch<2>(a, grid, nt, stream); break;
    case 4: ts_launch<4>(a, grid, nt, stream); break;
    case 8: ts_launch<8>(a, grid, nt, stream); break;
    default: ts_launch<16>(a, grid, nt, stream); break;
  }
}

// y, x: n elements (not padded); x may alias y.
void launch_trsv_pair(const TrsvPlan& p, const double* y, double* x, const Ctrl* ctrl, hipStream_t stream) {
  TriStepArgs a{};
  a.ncached = static_cast<uint32_t>(p.ncached > 0xffffffffLL ? 0xffffffffLL : p.ncached);
  a.ldp = p.ldp;
  a.n = p.n;
  a.ctrl = ctrl;
  int cur = 0;
  auto block = [&](int32_t k, int32_t& t0, int32_t& t1) {
    t0 = k * p.bt;
    t1 = std::min(p.ntile, t0 + p.bt);
  };
  // forward sweep: w = inv(L) y
  a.M = p.Fm;
  a.upper = 0;
  a.x_use_base = 1;
  a.diag_out = p.w;
  for (int32_t k = 0; k < p.nblk; ++k, cur ^= 1) {
    block(k, a.dt0, a.dt1);
    a.rt0 = a.dt0;
    a.rt1 = p.ntile;
    a.base_in = (k == 0) ? y : p.v;
    a.base_out = p.v;
    a.Pcur = p.P[cur];
    a.Pprev = p.P[cur ^ 1];
    if (k == 0) {
      a.pd0 = a.pd1 = a.pr0 = a.pr1 = a.ext0 = a.ext1 = 0;
    } else {
      block(k - 1, a.pd0, a.pd1);
      a.pr0 = a.pd0;
      a.pr1 = p.ntile;
      a.ext0 = a.pd0;  // w_{k-1}
      a.ext1 = a.pd1;
    }
    a.pupper = 0;
    ts_step(a, p.streaming, stream);
  }
  // backward sweep: x = inv(L)' w
  a.M = p.Um;
  a.upper = 1;
  a.diag_out = x;
  a.base_in = p.w;
  a.base_out = p.w;
  for (int32_t k = p.nblk - 1; k >= 0; --k, cur ^= 1) {
    block(k, a.dt0, a.dt1);
    a.rt0 = 0;
    a.rt1 = a.dt1;
    a.Pcur = p.P[cur];
    a.Pprev = p.P[cur ^ 1];
    if (k == p.nblk - 1) {  // the last forward step's partials ARE w of the last block
      block(k, a.pd0, a.pd1);
      a.pr0 = a.pd0;
      a.pr1 = p.ntile;
      a.pupper = 0;
      a.x_use_base = 0;
      a.ext0 = a.ext1 = 0;
    } else {
      block(k + 1, a.pd0, a.pd1);
      a.pr0 = 0;
      a.pr1 = a.pd1;
      a.pupper = 1;
      a.x_use_base = 1;
      a.ext0 = a.pd0;  // x_{k+1}
      a.ext1 = a.pd1;
    }
    ts_step(a, p.streaming, stream);
  }
  // x of block 0 from the last step's partials
  block(0, a.pd0, a.pd1);
  a.pr0 = 0;
  a.pr1 = a.pd1;
  a.pupper = 1;
  a.Pprev = p.P[cur ^ 1];
  a.ext0 = a.pd0;
  a.ext1 = a.pd1;
  hipLaunchKernelGGL(tri_fold_kernel, dim3(static_cast<unsigned>(a.ext1 - a.ext0)), dim3(kWave), 0, stream, a);
}

}  // namespace admm
