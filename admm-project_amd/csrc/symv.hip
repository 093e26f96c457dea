// symv.hip -- y = M*x for a SYMMETRIC M reading only its lower triangle (half the bytes of a GEMV).
// Used for the explicit-inverse x-update  x = inv(D'D + rho I) * y  (getProxOps.m:1200 semantics).
//
// Every element M[r,j] (r >= j) is loaded once and used twice:
//   N-part  yN[r] += M[r,j]*x[j]        register accumulation per row (like gemv_n)
//   T-part  yT[j] += M[r,j]*x[r], r > j per-column sums over the rows (like gemv_t)
// A WAVE is the unit of work (one-wave workgroups on 128x128 tiles, 16-byte loads, lane owns a row
// pair), so the triangular tile set load-balances finely and nothing synchronises.  The storage is
// padded to whole tiles with zeros (SymvPlan::npad), so the hot loop has no edge handling at all.
// The T-part column sums of a 4-column panel are combined inside the wave WITHOUT the LDS crossbar:
// two gfx950 v_permlane{32,16}_swap exchange steps (a reduce-scatter over lane bits 5 and 4) and a
// DPP butterfly inside each 16-lane row -- measured 25 us cheaper per 400 MB pass than the
// ds_bpermute (__shfl_xor) form, which was the limiter (dev/symv_bench.hip).  The next panel's
// loads are issued before the reduction of the current one.
// Partials (npart per 128-column group, tpart per 128-row wave chunk) are added by a second small
// kernel in a fixed order: bitwise reproducible, no float atomics.
// Storage (r2c): the explicit inverses are kept TILE-PACKED -- only the lower-triangle tiles, each 128x128 tile
// contiguous (column-major inside, 128 KB), tiles in row-major triangle order (SymvPlan::packed) -- half the memory of
// the padded square, one workgroup per stored tile (no empty workgroups above the diagonal) and every wave streaming
// one contiguous 128 KB run.  Cache policy: the first SymvPlan::ncached tiles are read with default loads, the rest
// non-temporally.  The matrix is re-read every iteration and only ~180 MB of it can stay in the 256 MB Infinity Cache:
// with default loads everywhere the LRU evicts every line before its reuse (5.4 TB/s), with non-temporal loads
// everywhere nothing is kept (5.9), with the split the cacheable part is served on-die from the second pass on:
// n = 10000 (414 MB): 73.7 us column-major / all NT -> 62.6 us packed / 160 MB cacheable (dev/symv_packed.hip).
// (r2: 8-column panels -- 16 loads in flight, a three-level reduce-scatter ending in row_half_mirror -- measured
// slower than this 4-column form: 78.4 vs 76.5 us event-timed per pass.  A pure N-part pass over the same tiles,
// tri_step_kernel in trsv.hip, streams at 6.0 TB/s, so the T-part's cross-lane work is what holds this kernel at 5.4.)
#include <cstdlib>

#include "finalize_device.h"
#include "kernels.h"
#include "wave_reduce.h"

namespace admm {

typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int kSyTile = 128;  // rows per wave (64 lanes x 2) = columns per tile
// (kSyPanel = 4 columns per panel = loads per buffer = T-part accumulators: wave_reduce.h)

struct SyLane {  // per-lane constants of a tile
  const double* M;
  const double* x;
  int64_t ld, n;
  int r;  // this lane's row pair (r, r+1), also its element offset from a column base
  double xr0, xr1;
};

template <bool NT>
__device__ __forceinline__ void sy_load(const SyLane& s, int64_t cp, double2_t (&d)[kSyPanel]) {
#pragma unroll
  for (int k = 0; k < kSyPanel; ++k) d[k] = load2<NT>(s.M + (cp + k) * s.ld + s.r);  // wave-uniform column base
}

// DIAG: the tile intersects the diagonal -> mask element-wise (N-part j <= row, T-part row > j)
// NP / TP: which of the two uses of an element are taken (both for the symmetric product; one each for the two
// passes over a triangular inverse further down, whose diagonal tiles store their zeros: DIAG = false there)
template <bool DIAG, bool NP = true, bool TP = true>
__device__ __forceinline__ void sy_compute(const SyLane& s, int64_t cp, const double2_t (&d)[kSyPanel], double& n0,
                                           double& n1, double* __restrict__ tout, int lane) {
  double tacc[kSyPanel];
#pragma unroll
  for (int k = 0; k < kSyPanel; ++k) {
    const int64_t j = cp + k;
    const double xj = NP ? s.x[j < s.n ? j : s.n - 1] : 0.0;  // wave-uniform -> scalar load; padding columns are 0
    double a0 = d[k].x, a1 = d[k].y;
    double t0 = a0, t1 = a1;
    if (DIAG) {
      t0 = (s.r > j) ? a0 : 0.0;
      t1 = (s.r + 1 > j) ? a1 : 0.0;
      a0 = (j <= s.r) ? a0 : 0.0;
      a1 = (j <= s.r + 1) ? a1 : 0.0;
    }
    if (TP) tacc[k] = __builtin_fma(t0, s.xr0, t1 * s.xr1);
    if (NP) {
      n0 = __builtin_fma(a0, xj, n0);
      n1 = __builtin_fma(a1, xj, n1);
    }
  }
  if (TP) {
    const double sum = reduce_scatter4(tacc);
    if ((lane & 15) == 0) tout[cp + (lane >> 4)] = sum;
  }
}

// the 32 panels of one tile, next panel's loads issued before this panel's reduction
// NBUF panel buffers: NBUF - 1 panels (4 KB each per wave) in flight while one is consumed.  The launch is ONE
// generation of one-tile waves (3160 tiles on 256 CUs at n = 10^4: 12.3 waves per CU, set by the tile count, not by
// registers), so the bytes in flight per CU are 12.3 x (NBUF - 1) x 4 KB (+ the panel being waited for).
template <bool DIAG, bool NT, bool NP = true, bool TP = true, int NBUF = 2>
__device__ __forceinline__ void sy_tile(const SyLane& s, int64_t c0, double& n0, double& n1,
                                        double* __restrict__ tout, int lane) {
  constexpr int kPanels = kSyTile / kSyPanel, kFull = (kPanels / NBUF) * NBUF;
  static_assert(NBUF >= 2 && NBUF <= 8, "ring of panel buffers");
  double2_t buf[NBUF][kSyPanel];
#pragma unroll
  for (int b = 0; b + 1 < NBUF; ++b) sy_load<NT>(s, c0 + b * kSyPanel, buf[b]);
#pragma unroll 1
  for (int g = 0; g < kFull; g += NBUF) {
#pragma unroll
    for (int b = 0; b < NBUF; ++b) {
      // panel g + b is consumed from buf[b]; the panel NBUF - 1 ahead goes into the buffer freed last
      const int nxt = g + b + NBUF - 1;
      if (kFull - NBUF + b + NBUF - 1 < kPanels || nxt < kPanels)  // (first test: compile time, true for every g)
        sy_load<NT>(s, c0 + nxt * kSyPanel, buf[(b + NBUF - 1) % NBUF]);
      sy_compute<DIAG, NP, TP>(s, c0 + (g + b) * kSyPanel, buf[b], n0, n1, tout, lane);
    }
  }
#pragma unroll
  for (int b = 0; b < kPanels - kFull; ++b) {  // the panels a ring that does not divide 32 leaves over
    if (kFull + b + NBUF - 1 < kPanels) sy_load<NT>(s, c0 + (kFull + b + NBUF - 1) * kSyPanel, buf[(b + NBUF - 1) % NBUF]);
    sy_compute<DIAG, NP, TP>(s, c0 + (kFull + b) * kSyPanel, buf[b], n0, n1, tout, lane);
  }
}

// tile t of the lower triangle in row-major order (0,0), (1,0), (1,1), (2,0), ... -> (bi, bj)
__device__ __forceinline__ void tri_decode(unsigned t, unsigned& bi, unsigned& bj) {
  unsigned r = static_cast<unsigned>((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while (r * (r + 1u) / 2u > t) --r;
  while ((r + 1u) * (r + 2u) / 2u <= t) ++r;
  bi = r;
  bj = t - r * (r + 1u) / 2u;
}

// PACKED = false: M is npad x npad (npad = round_up(n, 128)) column-major with ld >= npad, zero outside n x n; grid
// (ntile, ntile).  PACKED = true: M holds the lower-triangle tiles back to back (symv_pack_kernel); grid = their count.
// x: n elements.  Tiles with linear index < ncached use default loads, the others non-temporal ones.
template <bool PACKED, int NBUF = 2>
__device__ __forceinline__ void symv_lower_body(unsigned block_x, unsigned block_y, const double* __restrict__ M,
                                                int64_t n, int64_t ld, const double* __restrict__ x,
                                                double* __restrict__ npart, double* __restrict__ tpart, int64_t ldp,
                                                int32_t part_rank, int32_t part_count, uint32_t ncached) {
  unsigned bi, bj, lin;
  if (PACKED) {
    lin = block_x;
    tri_decode(lin, bi, bj);
  } else {
    bi = block_x;
    bj = block_y;
    if (bi < bj) return;  // tile strictly above the diagonal
    lin = bi * (bi + 1u) / 2u + bj;
  }
  // multi-GPU: the lower-triangle tiles are dealt round-robin to the ranks; the slots of the tiles a rank
  // does not own stay at their initial zero and the partial results are summed by one all-reduce of n doubles
  if (part_count > 1 && static_cast<int32_t>(lin % static_cast<unsigned>(part_count)) != part_rank) return;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = static_cast<int64_t>(bi) * kSyTile;  // wave's first row
  const int64_t c0 = static_cast<int64_t>(bj) * kSyTile;
  const int64_t gr = w0 + 2 * lane;  // this lane's row pair
  SyLane s;
  int64_t cbase;  // first column as sy_tile counts it
  double* __restrict__ tout = tpart + static_cast<int64_t>(bi) * ldp;
  if (PACKED) {  // the tile is its own 128 x 128 matrix: local row / column numbers, x and tout shifted to its columns
    s.M = M + static_cast<int64_t>(lin) * (kSyTile * kSyTile);
    s.x = x + c0;
    s.ld = kSyTile;
    s.n = n - c0;
    s.r = 2 * lane;
    cbase = 0;
    tout += c0;
  } else {
    s.M = M;
    s.x = x;
    s.ld = ld;
    s.n = n;
    s.r = static_cast<int>(gr);
    cbase = c0;
  }
  s.xr0 = x[gr < n ? gr : n - 1];  // rows >= n multiply zero padding
  s.xr1 = x[gr + 1 < n ? gr + 1 : n - 1];
  double n0 = 0.0, n1 = 0.0;
  if (lin < ncached) {
    if (bi == bj) sy_tile<true, false, true, true, NBUF>(s, cbase, n0, n1, tout, lane);
    else sy_tile<false, false, true, true, NBUF>(s, cbase, n0, n1, tout, lane);
  } else {
    if (bi == bj) sy_tile<true, true, true, true, NBUF>(s, cbase, n0, n1, tout, lane);
    else sy_tile<false, true, true, true, NBUF>(s, cbase, n0, n1, tout, lane);
  }
  *reinterpret_cast<double2_t*>(npart + static_cast<int64_t>(bj) * ldp + gr) = double2_t{n0, n1};
}

template <bool PACKED>
__global__ __launch_bounds__(kWave) void symv_lower_kernel(const double* __restrict__ M, int64_t n, int64_t ld,
                                                           const double* __restrict__ x, double* __restrict__ npart,
                                                           double* __restrict__ tpart, int64_t ldp,
                                                           int32_t part_rank, int32_t part_count, uint32_t ncached,
                                                           const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  symv_lower_body<PACKED>(blockIdx.x, blockIdx.y, M, n, ld, x, npart, tpart, ldp, part_rank, part_count, ncached);
}

// The packed x-solve with a passenger: workgroup 0 runs the finalize logic of the PREVIOUS iteration (norms, tolerances,
// stop decision: ~6 us of one workgroup's serial work) while the other workgroups stream the matrix.  Nothing in this
// launch depends on that decision, and the fused element update that follows starts after it and no-ops when it has
// raised ctrl->stop: the tail of an iteration shrinks to the element update itself (engine_run.hip, defer_fin).
__global__ __launch_bounds__(kWave) void symv_lower_fin_kernel(const double* __restrict__ M, int64_t n,
                                                               const double* __restrict__ x,
                                                               double* __restrict__ npart, double* __restrict__ tpart,
                                                               int64_t ldp, int32_t part_rank, int32_t part_count,
                                                               uint32_t ncached, FinArgs f, int32_t fin_pending,
                                                               const Ctrl* __restrict__ ctrl) {
  // Every argument of the tile path but x requested at once (fetched where first used they cost four dependent scalar
  // round trips behind the stop test, in a one-tile workgroup that lives ~15 us).  NOT x: with its pointer live this
  // early hipcc allocates 100 VGPRs instead of 84 -- five waves per SIMD instead of six, and 3 % slower, measured.
  asm volatile("" ::"s"(M), "s"(n), "s"(npart), "s"(tpart), "s"(ldp), "s"(part_rank), "s"(part_count), "s"(ncached),
               "s"(fin_pending), "s"(ctrl));
  if (ctrl->stop) return;
  if (blockIdx.x == 0) {
    if (fin_pending) finalize_body<false, kWave>(f);
    return;
  }
  symv_lower_body<true>(blockIdx.x - 1u, 0u, M, n, 0, x, npart, tpart, ldp, part_rank, part_count, ncached);
}

// K packed matrices of one size in ONE launch (consensus lasso: the K slice inverses of a rank; blockIdx.y = slice):
// x, npart, tpart of slice k at x0 + k*xstride, npart0 + k*pstride, tpart0 + k*pstride
// (workgroup (0, 0) is a passenger: the deferred finalize logic of the previous iteration, as in symv_lower_fin_kernel)
__global__ __launch_bounds__(kWave) void symv_lower_batch_kernel(const double* const* __restrict__ Ms, int64_t n,
                                                                 const double* __restrict__ x0, int64_t xstride,
                                                                 double* __restrict__ npart0,
                                                                 double* __restrict__ tpart0, int64_t pstride,
                                                                 int64_t ldp, uint32_t ncached, FinArgs f,
                                                                 int32_t fin_pending, const Ctrl* __restrict__ ctrl) {
  asm volatile("" ::"s"(Ms), "s"(n), "s"(xstride), "s"(npart0), "s"(tpart0), "s"(pstride), "s"(ldp), "s"(ncached),
               "s"(fin_pending), "s"(ctrl));  // (as in symv_lower_fin_kernel: everything but the vector pointer)
  if (ctrl && ctrl->stop) return;
  if (blockIdx.x == 0) {
    if (blockIdx.y == 0 && fin_pending) finalize_body<false, kWave>(f);
    return;
  }
  const int64_t k = blockIdx.y;
  symv_lower_body<true>(blockIdx.x - 1u, 0u, Ms[k], n, 0, x0 + k * xstride, npart0 + k * pstride, tpart0 + k * pstride,
                        ldp, 0, 1, ncached);
}

// ---------------------------------------------------------------- x = X' (X y),  X = inv(L) tile-packed: the two
// triangular solves of getProxOps.m:1200 `U \ (L \ y)` with the WHOLE factor as one pre-inverted block (trsv.hip has the
// blocked substitution this is the one-block case of, and the reason: dependent steps, not bytes, bound a triangular
// solve here).  Two passes over the same 8 n(n+1)/2 bytes -- the N-part alone (w = X y: rows in registers), then the
// T-part alone (x = X' w: column sums by the reduce-scatter above) -- so both sweeps share ONE array and its
// Infinity-Cache resident share.  Forward: tile (bi, bj) writes the partial row npart[bj][rows of bi]; w = their sum
// over bj <= bi, by tri1_fold_kernel (the backward pass needs w as a vector: two values per lane).  Tiles are dealt in
// DEscending order and the backward pass in ascending order: what one pass read last the next one reads first.
// Backward: tile (bi, bj) writes tpart[bi][columns of bj]; x = sum over bi >= bj, taken by the consumer
// (prox_fin_kernel's gather, or tri1_reduce_kernel).  Every sum in a fixed order.
// (Measured, n = 10^4: forward 60 us + fold + backward 62 us = 6.4 TB/s per pass, the ceiling of a read stream that
// does not fit L2 on this part (dev/gemv_pattern: 6.8 - 7.0).  A last-arriver fold inside the forward pass -- write-through
// partial rows, a drained counter add per tile, the row tile's last tile summing it -- was built first and is gone:
// 74 us for the pass instead of 60 + 4; 3 or 4 panel buffers instead of 2 change nothing, here and in symv_lower_*.)
constexpr int kT1Panel = 4;  // forward pass: columns per load group (8: 144 VGPRs = 3 waves per SIMD = 12 tile slots per CU
                             // for 12.3 tiles per CU at n = 10^4: a second generation for the last 88 tiles)

__device__ __forceinline__ double t1_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <bool NT>
__device__ __forceinline__ void t1_load(const double* __restrict__ Mr, int c, double2_t (&d)[kT1Panel]) {
#pragma unroll
  for (int k = 0; k < kT1Panel; ++k) d[k] = load2<NT>(Mr + (c + k) * kSyTile);
}
// yp: lane l holds the input pair of the tile's columns (2l, 2l + 1)
__device__ __forceinline__ void t1_fma(double2_t yp, int co, const double2_t (&d)[kT1Panel], double& a0, double& a1) {
#pragma unroll
  for (int k = 0; k < kT1Panel; ++k) {
    const double yj = t1_readlane((k & 1) ? yp.y : yp.x, (co + k) >> 1);
    a0 = __builtin_fma(d[k].x, yj, a0);
    a1 = __builtin_fma(d[k].y, yj, a1);
  }
}
template <bool NT, int NBUF>
__device__ __forceinline__ void t1_forward_tile(const double* __restrict__ Mr, double2_t yp, double& a0, double& a1) {
  constexpr int kPanels = kSyTile / kT1Panel, kFull = (kPanels / NBUF) * NBUF;
  double2_t buf[NBUF][kT1Panel];
#pragma unroll
  for (int b = 0; b + 1 < NBUF; ++b) t1_load<NT>(Mr, b * kT1Panel, buf[b]);
#pragma unroll 1
  for (int g = 0; g < kFull; g += NBUF) {
#pragma unroll
    for (int b = 0; b < NBUF; ++b) {  // (the ring of sy_tile)
      const int nxt = g + b + NBUF - 1;
      if (kFull + b - 1 < kPanels || nxt < kPanels) t1_load<NT>(Mr, nxt * kT1Panel, buf[(b + NBUF - 1) % NBUF]);
      t1_fma(yp, (g + b) * kT1Panel, buf[b], a0, a1);
    }
  }
#pragma unroll
  for (int b = 0; b < kPanels - kFull; ++b) {
    if (kFull + b + NBUF - 1 < kPanels) t1_load<NT>(Mr, (kFull + b + NBUF - 1) * kT1Panel, buf[(b + NBUF - 1) % NBUF]);
    t1_fma(yp, (kFull + b) * kT1Panel, buf[b], a0, a1);
  }
}

template <int NBUF>
__device__ __forceinline__ void tri1_forward_body(unsigned lin, const Tri1Args& a) {
  unsigned bi, bj;
  tri_decode(lin, bi, bj);
  const int lane = threadIdx.x & 63;
  const int64_t gr = static_cast<int64_t>(bi) * kSyTile + 2 * lane;  // this lane's row pair
  double* __restrict__ po = a.npart + static_cast<int64_t>(bj) * a.ldp + gr;
  const double* __restrict__ Mr = a.X + static_cast<int64_t>(lin) * (kSyTile * kSyTile) + 2 * lane;
  const int64_t e = static_cast<int64_t>(bj) * kSyTile + 2 * lane;  // the tile's input columns: pair (2l, 2l + 1)
  double2_t yp{0.0, 0.0};  // (the caller's vector is only 8-byte aligned and not padded)
  yp.x = a.y[e < a.n ? e : a.n - 1];
  yp.y = a.y[e + 1 < a.n ? e + 1 : a.n - 1];
  if (e >= a.n) yp.x = 0.0;
  if (e + 1 >= a.n) yp.y = 0.0;
  double a0 = 0.0, a1 = 0.0;
  if (lin < a.ncached) t1_forward_tile<false, NBUF>(Mr, yp, a0, a1);
  else t1_forward_tile<true, NBUF>(Mr, yp, a0, a1);
  *reinterpret_cast<double2_t*>(po) = double2_t{a0, a1};
}

// workgroup 0 is the passenger of symv_lower_fin_kernel: the deferred finalize logic of the previous iteration
__global__ __launch_bounds__(kWave) void tri1_forward_kernel(Tri1Args a, FinArgs f, int32_t fin_pending,
                                                             const Ctrl* __restrict__ ctrl) {
  asm volatile("" ::"s"(a.X), "s"(a.n), "s"(a.npart), "s"(a.ldp), "s"(a.ncached), "s"(a.ntri), "s"(fin_pending),
               "s"(ctrl));
  if (ctrl && ctrl->stop) return;
  if (blockIdx.x == 0) {
    if (fin_pending) finalize_body<false, kWave>(f);
    return;
  }
  tri1_forward_body<2>(a.ntri - blockIdx.x, a);  // blockIdx 1 .. ntri -> tiles ntri - 1 .. 0
}

__global__ __launch_bounds__(kWave) void tri1_backward_kernel(Tri1Args a, const Ctrl* __restrict__ ctrl) {
  asm volatile("" ::"s"(a.X), "s"(a.n), "s"(a.tpart), "s"(a.ldp), "s"(a.ncached), "s"(ctrl));
  if (ctrl && ctrl->stop) return;
  const unsigned lin = blockIdx.x;
  unsigned bi, bj;
  tri_decode(lin, bi, bj);
  const int lane = threadIdx.x & 63;
  const int64_t gr = static_cast<int64_t>(bi) * kSyTile + 2 * lane;
  SyLane s;
  s.M = a.X + static_cast<int64_t>(lin) * (kSyTile * kSyTile);
  s.x = nullptr;
  s.ld = kSyTile;
  s.n = 0;
  s.r = 2 * lane;
  const double2_t wr = *reinterpret_cast<const double2_t*>(a.w + gr);
  s.xr0 = wr.x;
  s.xr1 = wr.y;
  double* __restrict__ tout = a.tpart + static_cast<int64_t>(bi) * a.ldp + static_cast<int64_t>(bj) * kSyTile;
  double n0 = 0.0, n1 = 0.0;
  if (lin < a.ncached) sy_tile<false, false, false, true>(s, 0, n0, n1, tout, lane);
  else sy_tile<false, true, false, true>(s, 0, n0, n1, tout, lane);
}

// x[i] = sum_{p >= d} tpart[p][i],  d = i / 128  (the stand-alone form of what prox_fin_kernel's gather does)
__global__ __launch_bounds__(kBlock) void tri1_reduce_kernel(const double* __restrict__ tpart, int64_t ldp, int64_t n,
                                                             int32_t ntile, double* __restrict__ x,
                                                             const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double sacc[16][17];
  const int ii = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + ii;
  double s = 0.0;
  if (i < n) {
    const int32_t d = static_cast<int32_t>(i / kSyTile);
#pragma unroll 4
    for (int32_t p = d + slot; p < ntile; p += 16) s += tpart[static_cast<int64_t>(p) * ldp + i];
  }
  sacc[slot][ii] = s;
  __syncthreads();
  if (slot == 0 && i < n) {
    double t = sacc[0][ii];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sacc[k][ii];
    x[i] = t;
  }
}

// w[i] = sum_{p <= d} npart[p][i], d = i / 128
__global__ __launch_bounds__(kBlock) void tri1_fold_kernel(const double* __restrict__ npart, int64_t ldp, int64_t npad,
                                                           double* __restrict__ w, const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double sacc[16][17];
  const int ii = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + ii;
  double s = 0.0;
  if (i < npad) {
    const int32_t d = static_cast<int32_t>(i / kSyTile);
#pragma unroll 4
    for (int32_t p = slot; p <= d; p += 16) s += npart[static_cast<int64_t>(p) * ldp + i];
  }
  sacc[slot][ii] = s;
  __syncthreads();
  if (slot == 0 && i < npad) {
    double t = sacc[0][ii];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sacc[k][ii];
    w[i] = t;
  }
}

void launch_tri1_forward(const Tri1Args& a, const FinArgs* fin, bool fin_pending, const Ctrl* ctrl, hipStream_t stream) {
  const FinArgs f = fin ? *fin : FinArgs{};
  hipLaunchKernelGGL(tri1_forward_kernel, dim3(a.ntri + 1u), dim3(kWave), 0, stream, a, f, (fin && fin_pending) ? 1 : 0,
                     ctrl);
  const int64_t npad = static_cast<int64_t>(a.ntile) * kSyTile;
  hipLaunchKernelGGL(tri1_fold_kernel, dim3(static_cast<unsigned>(ceil_div(npad, int64_t{16}))), dim3(kBlock), 0, stream,
                     a.npart, a.ldp, npad, a.w, ctrl);
}
void launch_tri1_backward(const Tri1Args& a, const Ctrl* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(tri1_backward_kernel, dim3(a.ntri), dim3(kWave), 0, stream, a, ctrl);
}
void launch_tri1_reduce(const Tri1Args& a, double* x, const Ctrl* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(tri1_reduce_kernel, dim3(static_cast<unsigned>(ceil_div(a.n, int64_t{16}))), dim3(kBlock), 0,
                     stream, a.tpart, a.ldp, a.n, a.ntile, x, ctrl);
}

// column-major padded storage -> tile-packed storage (one workgroup per lower-triangle tile)
__global__ __launch_bounds__(kBlock) void symv_pack_kernel(const double* __restrict__ M, int64_t ld,
                                                           double* __restrict__ P) {
  unsigned bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  const double* src = M + static_cast<int64_t>(bj) * kSyTile * ld + static_cast<int64_t>(bi) * kSyTile;
  double* dst = P + static_cast<int64_t>(blockIdx.x) * (kSyTile * kSyTile);
  for (int e = 2 * threadIdx.x; e < kSyTile * kSyTile; e += 2 * kBlock)
    *reinterpret_cast<double2_t*>(dst + e) =
        *reinterpret_cast<const double2_t*>(src + static_cast<int64_t>(e / kSyTile) * ld + (e % kSyTile));
}

// y[i] = sum_{g <= d} npart[g][i] + sum_{w >= d} tpart[w][i],  d = i / 128 (i's diagonal tile).
// Workgroup = 16 consecutive elements x 16 slots that split the T+1 partial rows; the slots are
// combined through LDS in a fixed order.
__global__ __launch_bounds__(kBlock) void symv_reduce_kernel(const double* __restrict__ npart,
                                                             const double* __restrict__ tpart, int64_t ldp,
                                                             int64_t n, int32_t ntile, double* __restrict__ y,
                                                             const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double sacc[16][17];
  const int ii = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + ii;
  double s = 0.0;
  if (i < n) {
    const int32_t d = static_cast<int32_t>(i / kSyTile);
    // unified partial index p in [0, ntile]: p <= d -> npart[p], else tpart[p - 1]
#pragma unroll 4
    for (int32_t p = slot; p <= ntile; p += 16) {
      const double* src = (p <= d) ? npart + static_cast<int64_t>(p) * ldp : tpart + static_cast<int64_t>(p - 1) * ldp;
      s += src[i];
    }
  }
  sacc[slot][ii] = s;
  __syncthreads();
  if (slot == 0 && i < n) {
    double t = sacc[0][ii];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sacc[k][ii];
    y[i] = t;
  }
}

// Small symmetric M (n < ~1500: cache-resident, latency-bound): y_j = column j of M dot x, one wave per
// column, result written directly (no partials, no second kernel).  n = 400: 3 us against 12 + 4 us for the
// chunked column-dot GEMV + its reduction.
__global__ __launch_bounds__(kBlock) void symv_small_kernel(const double* __restrict__ M, int64_t n, int64_t ld,
                                                            const double* __restrict__ x, double* __restrict__ y,
                                                            const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 4 + wid;
  if (j >= n) return;
  const double* __restrict__ col = M + j * ld;  // ld is even and M 16-byte aligned: pairs are aligned
  const int64_t npairs = n >> 1;
  double a0 = 0.0, a1 = 0.0;
  for (int64_t p = lane; p < npairs; p += 64) {
    const double2_t m2 = *reinterpret_cast<const double2_t*>(col + 2 * p);
    const double2_t x2 = *reinterpret_cast<const double2_t*>(x + 2 * p);
    a0 = __builtin_fma(m2.x, x2.x, a0);
    a1 = __builtin_fma(m2.y, x2.y, a1);
  }
  double s = a0 + a1;
  if ((n & 1) && lane == 0) s = __builtin_fma(col[n - 1], x[n - 1], s);
  s = wave_sum(s);
  if (lane == 0) y[j] = s;
}

void launch_symv_small(const double* M, int64_t n, int64_t ld, const double* x, double* y, const Ctrl* ctrl,
                       hipStream_t stream) {
  hipLaunchKernelGGL(symv_small_kernel, dim3(static_cast<unsigned>(ceil_div(n, 4))), dim3(kBlock), 0, stream, M, n,
                     ld, x, y, ctrl);
}

SymvPlan symv_plan(int64_t n) {
  SymvPlan p{};
  p.n = n;
  p.npad = round_up(n, kSyTile);
  p.ldp = p.npad;
  p.ntile = static_cast<int32_t>(p.npad / kSyTile);
  p.packed = false;
  int64_t budget = kSymvCacheBytes;
  if (const char* env = getenv("ADMM_HIP_SYMV_CACHE_MB")) budget = static_cast<int64_t>(atoll(env)) << 20;  // tuning knob
  p.ncached = symv_cached_tiles(p, budget);
  return p;
}

int64_t symv_tiles(const SymvPlan& p) { return static_cast<int64_t>(p.ntile) * (p.ntile + 1) / 2; }
size_t symv_packed_elems(const SymvPlan& p) { return static_cast<size_t>(symv_tiles(p)) * kSyTile * kSyTile; }

int64_t symv_cached_tiles(const SymvPlan& p, int64_t budget_bytes) {
  const int64_t tiles = symv_tiles(p), tile_bytes = int64_t{8} * kSyTile * kSyTile;
  if (!stream_hint(tiles * tile_bytes)) return tiles;  // small enough to stay cache-resident as a whole
  const int64_t fit = budget_bytes / tile_bytes;
  return fit < tiles ? fit : tiles;
}

void launch_symv_pack(const SymvPlan& p, const double* M, int64_t ld, double* P, hipStream_t stream) {
  hipLaunchKernelGGL(symv_pack_kernel, dim3(static_cast<unsigned>(symv_tiles(p))), dim3(kBlock), 0, stream, M, ld, P);
}

void launch_symv_lower(const SymvPlan& p, const double* M, int64_t ld, const double* x, double* npart, double* tpart,
                       double* y, const Ctrl* ctrl, hipStream_t stream, int part_rank, int part_count, bool reduce) {
  const uint32_t ncached = static_cast<uint32_t>(p.ncached < 0 ? 0 : p.ncached);
  if (p.packed)
    hipLaunchKernelGGL(symv_lower_kernel<true>, dim3(static_cast<unsigned>(symv_tiles(p))), dim3(kWave), 0, stream, M,
                       p.n, ld, x, npart, tpart, p.ldp, part_rank, part_count, ncached, ctrl);
  else
    hipLaunchKernelGGL(symv_lower_kernel<false>, dim3(static_cast<unsigned>(p.ntile), static_cast<unsigned>(p.ntile)),
                       dim3(kWave), 0, stream, M, p.n, ld, x, npart, tpart, p.ldp, part_rank, part_count, ncached,
                       ctrl);
  if (!reduce) return;  // the consumer sums the partial rows itself (prox_fin_kernel)
  const int64_t blocks = ceil_div(p.n, 16);
  hipLaunchKernelGGL(symv_reduce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, npart, tpart,
                     p.ldp, p.n, p.ntile, y, ctrl);
}

void launch_symv_lower_batch(const SymvPlan& p, const double* const* Ms_dev, int32_t K, const double* x0, int64_t xstride,
                             double* npart0, double* tpart0, int64_t pstride, const Ctrl* ctrl, hipStream_t stream,
                             const FinArgs* fin) {
  const uint32_t ncached = static_cast<uint32_t>(p.ncached < 0 ? 0 : p.ncached);
  const FinArgs f = fin ? *fin : FinArgs{};
  const dim3 grid(static_cast<unsigned>(symv_tiles(p)) + 1u, static_cast<unsigned>(K));
  hipLaunchKernelGGL(symv_lower_batch_kernel, grid, dim3(kWave), 0, stream, Ms_dev, p.n, x0, xstride, npart0, tpart0,
                     pstride, p.ldp, ncached, f, fin ? 1 : 0, ctrl);
}

void launch_symv_lower_fin(const SymvPlan& p, const double* M, const double* x, double* npart, double* tpart,
                           const FinArgs& f, bool fin_pending, const Ctrl* ctrl, hipStream_t stream, int part_rank,
                           int part_count, double* y) {
  const uint32_t ncached = static_cast<uint32_t>(p.ncached < 0 ? 0 : p.ncached);
  const dim3 grid(static_cast<unsigned>(symv_tiles(p)) + 1u);
  hipLaunchKernelGGL(symv_lower_fin_kernel, grid, dim3(kWave), 0, stream, M, p.n, x, npart, tpart, p.ldp, part_rank,
                     part_count, ncached, f, fin_pending ? 1 : 0, ctrl);
  if (!y) return;  // the consumer sums the partial rows itself
  const int64_t blocks = ceil_div(p.n, 16);
  hipLaunchKernelGGL(symv_reduce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, npart, tpart,
                     p.ldp, p.n, p.ntile, y, ctrl);
}

}  // namespace admm
