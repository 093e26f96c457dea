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
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

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

// ---------------------------------------------------------------- the same solve pair as ONE launch
// The 2K + 1 launches above cost ~6 us each beyond their streaming (boundary, ramp-up, the fold of the previous step's
// partials in front of every tile, the tail): 55 of 187 us at n = 10^4.  tri_persist_kernel runs the whole pair -- the
// same tiles, the same panels, the same fixed-order sums -- from an ordered work list:
//   * a workgroup (4 waves: 32 columns of a 128 x 128 tile each, summed through LDS) takes the next ticket, i.e. the
//     next tile of the list, as soon as it is free -- whatever workgroups the device has resident make progress, and a
//     tile's producers always hold earlier tickets, so nobody waits for work that has not been handed out;
//   * step s writes its column-tile partials and its running right-hand side into buffers of its OWN (write-once per
//     launch: no reuse hazards, nothing stale to find in a cache), published with write-through (sc1) stores; the
//     storing wave drains them (s_waitcnt vmcnt(0)) and one lane adds 1 to the counter of (step, row tile);
//   * a tile whose input block is row tile T of the previous step polls that counter (one lane, sc1 loads, s_sleep),
//     the workgroup meets at a barrier, and every load of the handed-over bytes is an sc1 load
//     (MI355X_MICROARCH.md, inter-workgroup visibility: the flag-table protocol);
//   * within a step the list puts the row tiles the NEXT step's inputs come from first (column-tile major), then the
//     diagonal block, then the rest: while the next block's input is completing, the rest of this step's panel streams.
// The last workgroup to finish zeroes counters and tickets for the next launch.  A poll that does not complete within
// ~2 s (a lost producer: should not happen) raises the plan's error word instead of hanging the device.
// MEASURED (round 3, n = 10^4, K = 5): 294 us per pair, against 171 us for the stepwise launches -- opt-in only, see
// launch_trsv_pair for the accounting.
struct TpItem {
  int16_t step, R, c, kind;  // kind 0: tile (R, c) of step; 1: x of row tile R from the partials of (backward) step
};

struct TpArgs {
  const double* Fm;
  const double* Um;
  uint32_t ncached;
  int32_t streaming;
  int64_t n;
  const double* y;
  double* x;
  double* PP;
  double* BB;
  int32_t* sync;  // [0] ticket, [1] done, [2] error, [3] pad, counters from [4]
  const TpItem* items;
  const int32_t* chunks;
  int32_t nchunks;
  int64_t ldp;
  int32_t ntile, bt, nblk;
  const Ctrl* ctrl;
};

constexpr int kTpWaves = 4;
constexpr int kTpSpinMax = 1 << 20;

__device__ __forceinline__ void tp_block(const TpArgs& a, int s, int& dt0, int& dt1, int& upper) {
  upper = s >= a.nblk ? 1 : 0;
  const int k = upper ? 2 * a.nblk - 1 - s : s;
  dt0 = k * a.bt;
  dt1 = (dt0 + a.bt < a.ntile) ? dt0 + a.bt : a.ntile;
}
// partials row tile T has in step p (T inside p's panel)
__device__ __forceinline__ void tp_range(const TpArgs& a, int p, int T, int& q0, int& q1) {
  int dt0, dt1, upper;
  tp_block(a, p, dt0, dt1, upper);
  q0 = 0;
  q1 = dt1 - dt0;
  if (T >= dt0 && T < dt1) {
    if (!upper) q1 = T - dt0 + 1;
    else q0 = T - dt0;
  }
}
__device__ __forceinline__ double tp_ld(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: past this CU's L1
}
__device__ __forceinline__ void tp_st(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through
}
// sum over the partials q0 .. q1-1 of step p at element pair e (fixed order, sixteen loads of each half in flight)
__device__ __forceinline__ double2_t tp_sum(const TpArgs& a, int p, int q0, int q1, int64_t e) {
  constexpr int G = 16;
  double2_t t{0.0, 0.0};
  const double* __restrict__ base = a.PP + static_cast<int64_t>(p) * a.bt * a.ldp + e;
  for (int q = q0; q < q1; q += G) {
    double v0[G], v1[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const int qq = (q + k < q1) ? q + k : q1 - 1;
      v0[k] = tp_ld(base + static_cast<int64_t>(qq) * a.ldp);
      v1[k] = tp_ld(base + static_cast<int64_t>(qq) * a.ldp + 1);
    }
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (q + k < q1) t += double2_t{v0[k], v1[k]};
  }
  return t;
}
// one lane waits until row tile T of step p has all its partials
__device__ __forceinline__ void tp_wait(const TpArgs& a, int p, int T) {
  int q0, q1;
  tp_range(a, p, T, q0, q1);
  const int32_t target = q1 - q0;
  const int32_t* cnt = a.sync + 4 + p * a.ntile + T;
  for (int spin = 0;; ++spin) {
    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
    if (spin > kTpSpinMax || ((spin & 255) == 255 &&
                              __hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      __hip_atomic_store(a.sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(16);
  }
}

template <bool NT>
__device__ __forceinline__ void tp_stream(const double* __restrict__ Mr, int lc0, double2_t xp, double& a0, double& a1) {
  constexpr int WC = kTsTile / kTpWaves;  // 32 columns per wave
  double2_t bufA[kTsPanel], bufB[kTsPanel];
  ts_load<NT>(Mr, kTsTile, lc0, bufA);
#pragma unroll 1
  for (int co = 0; co < WC; co += 2 * kTsPanel) {
    ts_load<NT>(Mr, kTsTile, lc0 + co + kTsPanel, bufB);
    ts_fma(xp, co, bufA, a0, a1);
    if (co + 2 * kTsPanel < WC) ts_load<NT>(Mr, kTsTile, lc0 + co + 2 * kTsPanel, bufA);
    ts_fma(xp, co + kTsPanel, bufB, a0, a1);
  }
}

__global__ __launch_bounds__(kTpWaves* kWave) void tri_persist_kernel(TpArgs a) {
  if (a.ctrl && a.ctrl->stop) return;  // (the same answer in every workgroup: written by an earlier launch)
  constexpr int WC = kTsTile / kTpWaves;
  __shared__ double2_t red[kTpWaves][kWave];
  __shared__ int32_t sh_t, sh_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.nblk;
  int32_t* const ticket = a.sync;
  if (tid == 0) sh_t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int32_t t = sh_t;
  while (t < a.nchunks) {
    int32_t next_t = 0;  // the next ticket is asked for now and looked at when this one is done
    if (tid == 0) next_t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int32_t i0 = a.chunks[t], i1 = a.chunks[t + 1];
    for (int32_t it = i0; it < i1; ++it) {
      const TpItem item = a.items[it];
      const int s = item.step, R = item.R, c = item.c;
      int dt0, dt1, upper;
      tp_block(a, s, dt0, dt1, upper);
      const int64_t r = static_cast<int64_t>(R) * kTsTile + 2 * lane;  // this lane's row pair
      if (item.kind == 1) {  // x of row tile R: the sum of its partials in (backward) step s
        if (tid == 0) tp_wait(a, s, R);
        __syncthreads();
        if (wave == 0) {
          int q0, q1;
          tp_range(a, s, R, q0, q1);
          double2_t v = tp_sum(a, s, q0, q1, r);
          if (__hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            v = double2_t{__builtin_nan(""), __builtin_nan("")};  // a poll gave up: make the failure loud
          ts_store_vec(a.x, r, a.n, v);
        }
        continue;
      }
      const bool diag = R >= dt0 && R < dt1;
      const int T = dt0 + c;                       // the input block of this tile: row tile T of the previous step
      const bool fold_base = c == 0 && !diag;      // this tile also carries the running right-hand side of row tile R
      const int pin = (s == 0) ? -1 : s - 1;       // step whose partials feed the input (s == K: the last forward step)
      const int pbase = (s == 0) ? -1 : (s == K ? R / a.bt : s - 1);  // ... and the base of row tile R
      if (tid == 0) {
        if (pin >= 0) tp_wait(a, pin, T);
        if (fold_base && pbase >= 0) tp_wait(a, pbase, R);
      }
      __syncthreads();
      // the input pair of this wave's columns: lane l < 16 holds columns (2l, 2l + 1) of the wave's 32
      const int64_t cw0 = static_cast<int64_t>(T) * kTsTile + wave * WC;
      double2_t xp{0.0, 0.0};
      {
        const int64_t e = cw0 + 2 * (lane < WC / 2 ? lane : 0);
        if (s == 0) {
          xp = ts_load_vec(a.y, e, a.n);
        } else {
          int q0, q1;
          tp_range(a, pin, T, q0, q1);
          xp = tp_sum(a, pin, q0, q1, e);
          if (s != K) {
            const double* __restrict__ b = a.BB + static_cast<int64_t>(pin) * a.ldp + e;
            xp += double2_t{tp_ld(b), tp_ld(b + 1)};
          }
        }
      }
      if (fold_base && wave == 0) {  // B[s][R] = what the rows of R carry into the later steps
        double2_t bv;
        if (s == 0) {
          bv = ts_load_vec(a.y, r, a.n);
        } else {
          int q0, q1;
          tp_range(a, pbase, R, q0, q1);
          bv = tp_sum(a, pbase, q0, q1, r);
          if (s != K) {
            const double* __restrict__ b = a.BB + static_cast<int64_t>(pbase) * a.ldp + r;
            bv += double2_t{tp_ld(b), tp_ld(b + 1)};
          }
        }
        double* __restrict__ bo = a.BB + static_cast<int64_t>(s) * a.ldp + r;
        tp_st(bo, bv.x);
        tp_st(bo + 1, bv.y);
      }
      const uint32_t Ct = static_cast<uint32_t>(T), Rt = static_cast<uint32_t>(R);
      const uint32_t lin = upper ? Ct * (Ct + 1u) / 2u + Rt : Rt * (Rt + 1u) / 2u + Ct;
      const double* __restrict__ Mr = (upper ? a.Um : a.Fm) + static_cast<int64_t>(lin) * (kTsTile * kTsTile) + 2 * lane;
      double a0 = 0.0, a1 = 0.0;
      if (a.streaming && lin >= a.ncached) tp_stream<true>(Mr, wave * WC, xp, a0, a1);
      else tp_stream<false>(Mr, wave * WC, xp, a0, a1);
      red[wave][lane] = double2_t{a0, a1};
      __syncthreads();
      if (wave == 0) {
        double2_t acc = red[0][lane];
#pragma unroll
        for (int w = 1; w < kTpWaves; ++w) acc += red[w][lane];
        double* __restrict__ po = a.PP + (static_cast<int64_t>(s) * a.bt + c) * a.ldp + r;
        tp_st(po, acc.x);
        tp_st(po + 1, acc.y);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial row (and the base) have left this wave ...
        if (lane == 0)                                    // ... before the row tile's counter says so
          __hip_atomic_fetch_add(a.sync + 4 + s * a.ntile + R, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    if (tid == 0) sh_t = next_t;
    __syncthreads();
    t = sh_t;
  }
  // the last workgroup out resets the list for the next launch (everybody else has stopped reading the counters)
  if (tid == 0) {
    const int32_t old = __hip_atomic_fetch_add(a.sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = (old == static_cast<int32_t>(gridDim.x) - 1) ? 1 : 0;
  }
  __syncthreads();
  if (sh_last) {
    const int32_t ncnt = 2 * K * a.ntile;
    for (int32_t i = tid; i < ncnt; i += kTpWaves * kWave) a.sync[4 + i] = 0;
    if (tid == 0) {
      a.sync[0] = 0;
      a.sync[1] = 0;
    }
  }
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

// geometry shared by trsv_plan_elems and trsv_build
struct TsGeom {
  int64_t npad, ntile, bt, nblk;
  int64_t nitems;      // tiles of both sweeps + one x item per row tile
  size_t base_elems;   // Fm + Um (tile-packed triangles) + two partial buffers + v + w
  size_t pp_elems, bb_elems, sync_elems, item_elems, chunk_elems;  // one-launch form, in doubles
};
static TsGeom ts_geom(int64_t n) {
  TsGeom g{};
  g.npad = round_up(n, kTsTile);
  g.ntile = g.npad / kTsTile;
  g.bt = std::min<int64_t>(ts_block_tiles(), g.ntile);
  g.nblk = ceil_div(g.ntile, g.bt);
  g.bt = ceil_div(g.ntile, g.nblk);  // balanced blocks (as trsv_build)
  g.base_elems = static_cast<size_t>(g.ntile * (g.ntile + 1) * kTsTile * kTsTile + 2 * g.bt * g.npad + 2 * g.npad);
  g.nitems = g.ntile * (g.ntile + 1) + g.ntile;
  g.pp_elems = static_cast<size_t>(2 * g.nblk * g.bt * g.npad);
  g.bb_elems = static_cast<size_t>(2 * g.nblk * g.npad);
  g.sync_elems = static_cast<size_t>((4 + 2 * g.nblk * g.ntile + 1) / 2 + 1);
  g.item_elems = static_cast<size_t>(g.nitems + 1);            // 8-byte items
  g.chunk_elems = static_cast<size_t>((g.nitems + 2) / 2 + 64);  // 4-byte indices (at most one ticket per item + one per run)
  return g;
}

// ---------------------------------------------------------------- the one-block form (kernels: symv.hip, tri1_*)
// With the whole factor as ONE pre-inverted block the chain has no steps left: w = X y and x = X' w, X = inv(L), are
// two bandwidth-bound passes over the same tile-packed triangle (8 n(n+1) bytes per pair, what the substitution reads)
// and two launches.  The price is numerical: X is an explicit triangular inverse, forward error ~ eps * cond(L) per
// pass where the blocked form has eps * cond(L_kk) -- still the SQUARE ROOT of what the explicit inverse of L L' costs
// (`inverse` form).  Which of the two runs is decided by measurement at create (engine.hip: choose_trsv_form).
int trsv_resolve_form(int64_t n, int form) {
  if (const char* ev = std::getenv("ADMM_TRSV_FORM")) {
    if (ev[0] == 'b') form = kTrsvBlocked;
    else if (ev[0] == 'o') form = kTrsvOne;
  }
  if (n < 256) form = kTrsvBlocked;  // one or two tiles: nothing to gain
  return form;
}

struct T1Geom {
  int64_t npad, ntile, ntri;
  size_t x_elems, part_elems, w_elems;
};
static T1Geom t1_geom(int64_t n) {
  T1Geom g{};
  g.npad = round_up(n, kTsTile);
  g.ntile = g.npad / kTsTile;
  g.ntri = g.ntile * (g.ntile + 1) / 2;
  g.x_elems = static_cast<size_t>(g.ntri) * kTsTile * kTsTile;
  g.part_elems = static_cast<size_t>(g.ntile * g.npad);
  g.w_elems = static_cast<size_t>(g.npad);
  return g;
}

size_t trsv_plan_elems(int64_t n, int form) {
  if (form == kTrsvOne && n >= 256) {
    const T1Geom g = t1_geom(n);
    return g.x_elems + 2 * g.part_elems + g.w_elems;
  }
  const TsGeom g = ts_geom(n);
  return g.base_elems + g.pp_elems + g.bb_elems + g.sync_elems + g.item_elems + g.chunk_elems;
}

static int tri1_build(const double* L, int64_t n, int64_t ldl, const double* dinv64, double* buf, TrsvPlan* plan,
                      hipStream_t stream) {
  TrsvPlan& p = *plan;
  p = TrsvPlan{};
  const T1Geom g = t1_geom(n);
  p.n = n;
  p.npad = g.npad;
  p.ntile = static_cast<int32_t>(g.ntile);
  p.nblk = 1;
  p.bt = p.ntile;
  p.ldm = p.npad;
  p.ldp = p.npad;
  p.one = true;
  p.X1 = buf;
  p.np1 = p.X1 + g.x_elems;
  p.tp1 = p.np1 + g.part_elems;
  p.w1 = p.tp1 + g.part_elems;
  p.streaming = stream_hint(static_cast<int64_t>(g.x_elems) * 8);
  // ONE array serves both passes: the whole Infinity-Cache budget of symv.hip goes to its first tiles
  p.ncached = p.streaming ? static_cast<int64_t>(kSymvCacheBytes / (8 * kTsTile * kTsTile)) : INT64_MAX;
  ADMM_HIP_TRY(hipMemsetAsync(buf, 0, trsv_plan_elems(n, kTrsvOne) * sizeof(double), stream));
  double* dense = nullptr;  // X = inv(L) as a padded column-major square (zero above the diagonal and outside n x n)
  const size_t msz = static_cast<size_t>(p.npad) * p.npad;
  ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dense), msz * sizeof(double)));
  hipError_t se = hipMemsetAsync(dense, 0, msz * sizeof(double), stream);
  int rc = (se == hipSuccess) ? trtri_lower_from_diag(L, n, ldl, dinv64, dense, p.ldm, stream, false)
                              : fail(ADMM_E_DEVICE, "hipMemsetAsync failed");
  if (rc == ADMM_OK)
    hipLaunchKernelGGL(ts_pack_kernel, dim3(static_cast<unsigned>(g.ntri)), dim3(kBlock), 0, stream, dense, p.ldm, p.X1, 0);
  se = hipStreamSynchronize(stream);
  (void)hipFree(dense);
  ADMM_TRY(rc);
  ADMM_HIP_TRY(se);
  return ADMM_OK;
}

Tri1Args tri1_args(const TrsvPlan& p, const double* y) {
  Tri1Args a{};
  a.X = p.X1;
  a.n = p.n;
  a.y = y;
  a.npart = p.np1;
  a.tpart = p.tp1;
  a.w = p.w1;
  a.ldp = p.ldp;
  a.ncached = static_cast<uint32_t>(p.ncached > 0xffffffffLL ? 0xffffffffLL : (p.ncached < 0 ? 0 : p.ncached));
  a.ntile = p.ntile;
  a.ntri = static_cast<uint32_t>(p.ntile) * static_cast<uint32_t>(p.ntile + 1) / 2u;
  return a;
}

void launch_tri1_pair(const TrsvPlan& p, const double* y, double* x, const FinArgs* fin, bool fin_pending,
                      const Ctrl* ctrl, hipStream_t stream) {
  const Tri1Args a = tri1_args(p, y);
  launch_tri1_forward(a, fin, fin_pending, ctrl, stream);
  launch_tri1_backward(a, ctrl, stream);
  if (x) launch_tri1_reduce(a, x, ctrl, stream);
}

// the ordered work list of the one-launch form (tri_persist_kernel has the ordering rule)
static void ts_work_list(const TsGeom& g, std::vector<TpItem>* items, std::vector<int32_t>* chunks) {
  const int K = static_cast<int>(g.nblk), bt = static_cast<int>(g.bt), ntile = static_cast<int>(g.ntile);
  auto blk = [&](int k, int& t0, int& t1) {
    t0 = k * bt;
    t1 = std::min(ntile, t0 + bt);
  };
  auto put = [&](int s, int R, int c, int kind) {
    items->push_back(TpItem{static_cast<int16_t>(s), static_cast<int16_t>(R), static_cast<int16_t>(c),
                            static_cast<int16_t>(kind)});
  };
  // A ticket (one agent-scope atomic on ONE address: ~45 ns each, serialised -- 6400 single-tile tickets cost more than
  // the whole solve) hands out a run of consecutive items: short runs where the next step waits for the result (the
  // chain must spread over many workgroups), long ones in the rest of a panel.  ADMM_TRSV_CHUNK="crit,rest".
  int chunk_crit = 2, chunk_rest = 8;
  if (const char* ev = std::getenv("ADMM_TRSV_CHUNK")) {
    int a1 = 0, a2 = 0;
    if (std::sscanf(ev, "%d,%d", &a1, &a2) == 2 && a1 >= 1 && a2 >= 1 && a1 <= 64 && a2 <= 64) {
      chunk_crit = a1;
      chunk_rest = a2;
    }
  }
  chunks->clear();
  auto close_runs = [&](size_t from, int len) {  // tickets over items [from, items->size())
    for (size_t i = from; i < items->size(); i += static_cast<size_t>(len)) chunks->push_back(static_cast<int32_t>(i));
  };
  std::vector<std::pair<int, int>> pending_x;  // (step, row tile): x items, emitted behind the next step's first tiles
  for (int s = 0; s < 2 * K; ++s) {
    size_t mark = items->size();
    const bool upper = s >= K;
    const int k = upper ? 2 * K - 1 - s : s;
    int dt0, dt1;
    blk(k, dt0, dt1);
    const int ncols = dt1 - dt0;
    int n0 = 0, n1 = 0;  // the row tiles the next step's inputs come from
    if (!upper && k + 1 < K) blk(k + 1, n0, n1);
    if (upper && k > 0) blk(k - 1, n0, n1);
    for (int c = 0; c < ncols; ++c)
      for (int R = n0; R < n1; ++R) put(s, R, c, 0);
    close_runs(mark, chunk_crit);
    mark = items->size();
    for (const auto& px : pending_x) put(px.first, px.second, 0, 1);
    pending_x.clear();
    close_runs(mark, 4);
    mark = items->size();
    for (int c = 0; c < ncols; ++c)  // the diagonal block: the tiles that exist
      for (int R = dt0; R < dt1; ++R)
        if (upper ? c >= R - dt0 : c <= R - dt0) put(s, R, c, 0);
    // (the last forward step's diagonal block IS what the first backward step waits for)
    close_runs(mark, (!upper && k + 1 == K) || (upper && k == 0) ? chunk_crit : chunk_rest);
    mark = items->size();
    const int r0 = upper ? 0 : dt1, r1 = upper ? dt0 : ntile;  // the rest of the panel
    for (int c = 0; c < ncols; ++c)
      for (int R = r0; R < r1; ++R)
        if (R < n0 || R >= n1) put(s, R, c, 0);
    close_runs(mark, chunk_rest);
    if (upper)
      for (int R = dt0; R < dt1; ++R) pending_x.emplace_back(s, R);
  }
  const size_t mark = items->size();
  for (const auto& px : pending_x) put(px.first, px.second, 0, 1);
  close_runs(mark, 4);
  chunks->push_back(static_cast<int32_t>(items->size()));
}

// L: n x n lower factor (upper part ignored); dinv64: its inverted 64x64 diagonal blocks.
// `buf` must hold trsv_plan_elems(n) doubles and is owned by the caller.
int trsv_build(const double* L, int64_t n, int64_t ldl, const double* dinv64, double* buf, TrsvPlan* plan,
               hipStream_t stream, int form) {
  if (form == kTrsvOne && n >= 256) return tri1_build(L, n, ldl, dinv64, buf, plan, stream);
  TrsvPlan& p = *plan;
  p = TrsvPlan{};
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
  ADMM_HIP_TRY(hipMemsetAsync(buf, 0, trsv_plan_elems(n, kTrsvBlocked) * sizeof(double), stream));
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
  // ---- the one-launch form: buffers behind the stepwise ones, the work list uploaded once
  const TsGeom g = ts_geom(n);
  p.persist = false;
  if (g.nblk == p.nblk && g.bt == p.bt && p.ntile < 32000) {
    double* q = buf + g.base_elems;
    p.PP = q;
    q += g.pp_elems;
    p.BB = q;
    q += g.bb_elems;
    p.sync = reinterpret_cast<int32_t*>(q);
    q += g.sync_elems;
    TpItem* ditems = reinterpret_cast<TpItem*>(q);
    q += g.item_elems;
    int32_t* dchunks = reinterpret_cast<int32_t*>(q);
    std::vector<TpItem> items;
    std::vector<int32_t> chunks;
    items.reserve(static_cast<size_t>(g.nitems));
    ts_work_list(g, &items, &chunks);
    if (static_cast<int64_t>(items.size()) == g.nitems) {
      ADMM_HIP_TRY(hipMemcpyAsync(ditems, items.data(), items.size() * sizeof(TpItem), hipMemcpyHostToDevice, stream));
      ADMM_HIP_TRY(hipMemcpyAsync(dchunks, chunks.data(), chunks.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
      ADMM_HIP_TRY(hipStreamSynchronize(stream));  // (the host vectors go out of scope)
      p.items = ditems;
      p.chunks = dchunks;
      p.nitems = static_cast<int32_t>(items.size());
      p.nchunks = static_cast<int32_t>(chunks.size()) - 1;
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      p.grid = std::min<int32_t>(p.nchunks, 4 * cus);  // any size is safe (tickets); more than the device holds gains nothing
      p.persist = p.nblk >= 2;  // one block: the stepwise form is two launches already
    }
  }
  return ADMM_OK;
}

// 0 = fine; the one-launch form raises the word when a poll gives up (its x is NaN then)
int trsv_check_error(const TrsvPlan& p, hipStream_t stream) {
  if (!p.persist || !p.sync) return ADMM_OK;
  int32_t err = 0;
  ADMM_HIP_TRY(hipMemcpyAsync(&err, p.sync + 2, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  ADMM_HIP_TRY(hipStreamSynchronize(stream));
  if (err != 0) {
    ADMM_HIP_TRY(hipMemsetAsync(p.sync, 0, sizeof(int32_t) * (4 + 2 * static_cast<size_t>(p.nblk) * p.ntile), stream));
    return fail(ADMM_E_DEVICE, "triangular solves (one-launch form): a workgroup waited ~2 s for a tile that never "
                               "completed; the solve was abandoned");
  }
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
  if (p.one) {
    launch_tri1_pair(p, y, x, nullptr, false, ctrl, stream);
    return;
  }
  // The one-launch form is correct (tests/test_gpu_ops.py::test_trsv_pair runs both) but NOT the default: at n = 10^4 it
  // measured 294 us per pair against 171 us for the 2K + 1 launches below (profiles/r3_trsv_persist.txt).  A tile costs a
  // resident workgroup ~10 dependent memory round trips (ticket, item, two counter polls, the fold of up to 16 partial
  // rows + base, two rounds of panel loads, the write-through drain in front of the counter) at 4 workgroups per CU,
  // where the stepwise launches resolve dependencies at kernel boundaries and hide the same folds behind 16-32 resident
  // one-wave workgroups per CU.  ADMM_TRSV_ONE_LAUNCH=1 selects it.
  if (p.persist && std::getenv("ADMM_TRSV_ONE_LAUNCH") != nullptr) {
    TpArgs t{};
    t.Fm = p.Fm;
    t.Um = p.Um;
    t.ncached = static_cast<uint32_t>(p.ncached > 0xffffffffLL ? 0xffffffffLL : p.ncached);
    t.streaming = p.streaming ? 1 : 0;
    t.n = p.n;
    t.y = y;
    t.x = x;
    t.PP = p.PP;
    t.BB = p.BB;
    t.sync = p.sync;
    t.items = static_cast<const TpItem*>(p.items);
    t.chunks = p.chunks;
    t.nchunks = p.nchunks;
    t.ldp = p.ldp;
    t.ntile = p.ntile;
    t.bt = p.bt;
    t.nblk = p.nblk;
    t.ctrl = ctrl;
    hipLaunchKernelGGL(tri_persist_kernel, dim3(static_cast<unsigned>(p.grid)), dim3(kTpWaves * kWave), 0, stream, t);
    return;
  }
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
