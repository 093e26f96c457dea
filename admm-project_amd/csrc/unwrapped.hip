// unwrapped.hip -- two launches per iteration of unwrapped ADMM with an explicit pseudo-inverse
// (solvers/unwrappedadmm.m:76-92, linearsvm.m:183-217, getProxOps.m:1062-1100):
//     x = Dplus*(z - u);  z = prox(D*x + u);  u = u + D*x - z;  norms, tolerances, stop.
// BASELINE config 3 (6000 x 400, 60000 x 400) is latency-bound: D fits the caches, and the five launches of the generic
// A = D iteration (D'-product, its partial sum, x, D*x, prox + finalize) cost ~30 us where the arithmetic needs ~5.
// An iteration has two global reductions -- over the rows for x, over the columns for D*x -- hence two launches, each
// carrying its reduction across the kernel boundary as partial rows that the NEXT launch sums while it starts:
//   uw_ax_kernel   grid (row blocks of 128) x (column chunks of 64): sums the partial rows of Dplus*(c + z - u) for
//                  its 64 entries of x (the first row block also stores them), multiplies its 128 x 64 tile of D
//                  with them and stores one partial of D*x per column chunk.  Small tiles: every CU takes part.
//   uw_prox_kernel one workgroup per block of RB rows: sums the column-chunk partials of D*x, runs the fused element
//                  update (prox_apply: the same code as every other loop), multiplies the new c + z - u of its rows
//                  into its OWN partial row of the next x (Dplus stored n x m: coalesced along x), publishes the
//                  block partials of the residual sums.
// The finalize logic of iteration i (norms, tolerances, stop decision: admm.m:612-722; ~6 us as one workgroup's serial
// work) rides along with iteration i + 1's uw_ax launch as one extra workgroup: nothing in that launch depends on it,
// and uw_prox of iteration i + 1 starts after it and no-ops when it has raised ctrl->stop -- x is double-buffered on
// the iteration parity, so the x of the stopping iteration survives the speculative uw_ax.  The iteration index is a
// kernel argument (speculative launches past a stop are no-ops, so the host's count is the device's).
// (A single launch per iteration -- one workgroup per row block doing all of the above -- was measured first: 31 us per
// iteration at 6000 x 400, no better than five launches: 47 workgroups each pull 1 MB through one CU's L1, so the
// kernel runs at the bandwidth of 47 CUs.  The tiles have to be small enough for all 256 CUs to share the bytes.)
// Partial rows are double-buffered on the parity of ctrl->iter; every sum runs in a fixed order (bitwise reproducible).
// All loads are unconditional (clamped addresses, zero weights): a branch per load would serialise the round trips.
#include <algorithm>
#include <cstdlib>

#include "finalize_device.h"
#include "kernels.h"
#include "loop_kernels.h"
#include "prox_device.h"
#include "wave_reduce.h"

namespace admm {

constexpr int kUwTileRows = 128;   // uw_ax_kernel: rows per tile (lane = rows l and 64 + l)
constexpr int kUwTileCols = 64;    // uw_ax_kernel: entries of x per tile
constexpr int kUwSub = 64;         // uw_prox_kernel: rows per sub-block
constexpr int kUwProxThreads = 512;   // thread = entry of x (1024 threads leave 128 registers each: the load batches spill)
constexpr int kUwHalf = 512;
constexpr int kUwMaxCols = 2;      // entries of x per thread of uw_prox_kernel: n <= 1024

__global__ __launch_bounds__(kBlock) void uw_ax_kernel(UwArgs a, FinArgs f, Ctrl* __restrict__ ctrl) {
  // all kernel arguments of the tile path requested at once (hipcc otherwise fetches them in dependent groups behind
  // the stop test: four scalar round trips in a 7 us kernel)
  asm volatile("" ::"s"(a.D), "s"(a.ldD), "s"(a.m), "s"(a.n), "s"(a.nblk), "s"(a.G), "s"(a.ldg), "s"(a.axpart),
               "s"(a.ldax), "s"(a.xbuf), "s"(a.ldx), "s"(a.iter), "s"(a.fin_pending), "s"(ctrl));
  if (ctrl->stop) return;
  if (blockIdx.x == gridDim.x - 1) {  // the extra workgroup(s): finalize of the previous iteration
    if (blockIdx.y == 0 && a.fin_pending) finalize_body<false>(f);
    return;
  }
  const int64_t it = a.iter;
  __shared__ double xq[4][kUwTileCols];
  __shared__ double xs[kUwTileCols];
  __shared__ double red[4][kUwTileRows];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t n = a.n, m = a.m;
  const int64_t c0 = static_cast<int64_t>(blockIdx.y) * kUwTileCols;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kUwTileRows;
  const int ncol = (n - c0 < kUwTileCols) ? static_cast<int>(n - c0) : kUwTileCols;
  const int nrow = (m - r0 < kUwTileRows) ? static_cast<int>(m - r0) : kUwTileRows;
  const double* __restrict__ G = a.G + (it & 1) * (static_cast<int64_t>(a.nblk) * a.ldg);
  {  // ---- x[c0 .. c0 + 63] = sum of the partial rows; wave q takes rows q, q + 4, ...
    const int64_t j = c0 + (lane < ncol ? lane : ncol - 1);
    double s = 0.0;
    for (int32_t b0 = wid; b0 < a.nblk; b0 += 64) {
      double v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int32_t b = (b0 + 4 * k < a.nblk) ? b0 + 4 * k : a.nblk - 1;
        v[k] = G[static_cast<int64_t>(b) * a.ldg + j];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) s += (b0 + 4 * k < a.nblk) ? v[k] : 0.0;
    }
    xq[wid][lane] = s;
  }
  __syncthreads();
  if (wid == 0) {
    const double xv = (lane < ncol) ? ((xq[0][lane] + xq[1][lane]) + xq[2][lane]) + xq[3][lane] : 0.0;
    xs[lane] = xv;
    if (blockIdx.x == 0 && lane < ncol) a.xbuf[(it & 1) * a.ldx + c0 + lane] = xv;  // xopt, ||x||^2, history column
  }
  __syncthreads();
  // ---- D(r0 .. r0 + 127, c0 .. c0 + 63) * x: wave w takes columns w, w + 4, ...
  const double* __restrict__ d0 = a.D + r0 + (lane < nrow ? lane : nrow - 1) + c0 * a.ldD;
  const double* __restrict__ d1 = a.D + r0 + (64 + lane < nrow ? 64 + lane : nrow - 1) + c0 * a.ldD;
  double da[16], db[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = (wid + 4 * k < ncol) ? wid + 4 * k : ncol - 1;
    da[k] = d0[c * a.ldD];
    db[k] = d1[c * a.ldD];
  }
  double p0 = 0.0, p1 = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double xv = xs[wid + 4 * k];  // zero beyond ncol
    p0 = __builtin_fma(da[k], xv, p0);
    p1 = __builtin_fma(db[k], xv, p1);
  }
  red[wid][lane] = p0;
  red[wid][64 + lane] = p1;
  __syncthreads();
  if (tid < nrow)
    a.axpart[static_cast<int64_t>(blockIdx.y) * a.ldax + r0 + tid] =
        ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

__global__ __launch_bounds__(kUwProxThreads) void uw_prox_kernel(UwArgs a, ProxArgs pa, const Ctrl* __restrict__ ctrl) {
  asm volatile("" ::"s"(a.Dp), "s"(a.ldP), "s"(a.m), "s"(a.n), "s"(a.R), "s"(a.nblk), "s"(a.G), "s"(a.ldg), "s"(a.axpart),
               "s"(a.ldax), "s"(a.nchunk), "s"(a.iter), "s"(a.init), "s"(pa.z), "s"(pa.u), "s"(pa.c), "s"(pa.ell),
               "s"(pa.len), "s"(pa.part), "s"(pa.t), "s"(pa.rho), "s"(ctrl));
  const int32_t stop = ctrl->stop;
  const int64_t it = a.iter;
  const double aprev = ctrl->acurr;
  if (stop && !a.init) return;
  __shared__ double ws[kUwSub];
  __shared__ double accl[S_COUNT][kUwSub];  // per-lane residual sums of wave 0, kept out of its registers
  const int tid = threadIdx.x, lane = tid & 63;
  const int tj = tid;
  if (tid < kUwSub) {
#pragma unroll
    for (int s = 0; s < S_COUNT; ++s) accl[s][tid] = 0.0;
  }
  const int64_t n = a.n, m = a.m;
  double* __restrict__ Gnext = a.G + (a.init ? (it & 1) : ((it + 1) & 1)) * (static_cast<int64_t>(a.nblk) * a.ldg) +
                               static_cast<int64_t>(blockIdx.x) * a.ldg;
  double g[kUwMaxCols];
#pragma unroll
  for (int jj = 0; jj < kUwMaxCols; ++jj) g[jj] = 0.0;
  double kcoef = 0.0;
  if (pa.alg == 1) kcoef = (aprev - 1.0) / (0.5 * (1.0 + sqrt(1.0 + 4.0 * aprev * aprev)));

  const int64_t rbase = static_cast<int64_t>(blockIdx.x) * a.R;
  for (int sb = 0; sb < a.R / kUwSub; ++sb) {
    const int64_t r0 = rbase + static_cast<int64_t>(sb) * kUwSub;
    if (r0 >= m) break;  // workgroup-uniform
    const int nrow = (m - r0 < kUwSub) ? static_cast<int>(m - r0) : kUwSub;
    if (tid < kUwSub) {  // ---- the fused element update of these rows (one wave)
      double acc[S_COUNT];
#pragma unroll
      for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
      const int64_t row = r0 + (tid < nrow ? tid : nrow - 1);
      double w = 0.0;
      if (!a.init) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = a.axpart[static_cast<int64_t>(k < a.nchunk ? k : a.nchunk - 1) * a.ldax + row];
        const ProxIn in = prox_load(pa, row);
        double ax = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) ax += (k < a.nchunk) ? v[k] : 0.0;
        if (tid < nrow) prox_apply(pa, row, ax, it, kcoef, in, acc, &w);
#pragma unroll
        for (int s = 0; s < S_COUNT; ++s) accl[s][tid] += acc[s];
      } else if (tid < nrow) {
        w = ((pa.c ? pa.c[row] : 0.0) + pa.z[row]) - pa.u[row];
      }
      ws[tid] = w;
    }
    __syncthreads();
    // ---- this sub-block's share of Dplus*(c + z - u): thread = entry of x, 64 rows in two rounds of 32 loads
    // (rows past the end of the matrix carry ws = 0 and re-read the last row)
    auto share = [&](int64_t j, double s) -> double {
      const int64_t jc = j < n ? j : n - 1;
      const double* __restrict__ pcol = a.Dp + jc + r0 * a.ldP;
#pragma unroll 1
      for (int r = 0; r < kUwSub; r += 32) {
        double d[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) d[k] = pcol[static_cast<int64_t>(r + k < nrow ? r + k : nrow - 1) * a.ldP];
#pragma unroll
        for (int k = 0; k < 32; ++k) s = __builtin_fma(d[k], ws[r + k], s);
      }
      return s;
    };
    g[0] = share(tj, g[0]);
    if (n > kUwHalf) g[1] = share(tj + kUwHalf, g[1]);  // workgroup-uniform
    __syncthreads();  // ws is rewritten by the next sub-block
  }
  if (tj < n) Gnext[tj] = g[0];
  if (tj + kUwHalf < n) Gnext[tj + kUwHalf] = g[1];
  if (a.init) return;

  // ---- block partials (wave 0 holds them): read by the finalize workgroup of the next launch
  if (tid < 64) {
#pragma unroll
    for (int s = 0; s < S_COUNT; ++s) {
      const double w = wave_sum(accl[s][tid]);
      if (lane == 0) pa.part[s * kMaxPartBlocks + blockIdx.x] = w;
    }
  }
}

int uw_rows_per_block(int64_t m) {
  // about 128 partial rows: each tile of uw_ax_kernel sums nblk x 64 of them next to its 128 x 64 entries of D
  int64_t R = round_up(ceil_div(m, int64_t{128}), kUwSub);
  if (R < kUwSub) R = kUwSub;
  while (ceil_div(m, R) > kMaxPartBlocks) R += kUwSub;
  return static_cast<int>(R);
}

// both matrices (2 x 8mn bytes) have to stay in the Infinity Cache: beyond that the iteration is bandwidth-bound and the
// generic kernels, which spread every pass over thousands of workgroups, are faster (60000 x 400: 85 against 125 us)
bool uw_supported(int64_t m, int64_t n) {
  return n >= 1 && n <= kUwMaxCols * kUwHalf && m >= 1 && m * n <= (int64_t{8} << 20);
}
int uw_chunks(int64_t n) { return static_cast<int>(ceil_div(n, int64_t{kUwTileCols})); }

FinArgs uw_fin_args(const UwArgs& a, const FinArgs& f) {
  FinArgs ff = f;
  ff.nblk = a.nblk;
  ff.g = nullptr;
  ff.objpart = nullptr;
  ff.nobjpart = 0;
  ff.slots_reduced = nullptr;
  ff.objp_reduced = nullptr;
  return ff;
}

// f: the finalize arguments of the PREVIOUS iteration (f.x = its x), used when a.fin_pending
void launch_uw_ax(const UwArgs& a, const FinArgs& f, Ctrl* ctrl, hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>(ceil_div(a.m, int64_t{kUwTileRows})) + 1u, static_cast<unsigned>(a.nchunk));
  hipLaunchKernelGGL(uw_ax_kernel, grid, dim3(kBlock), 0, stream, a, uw_fin_args(a, f), ctrl);
}

void launch_uw_prox(const UwArgs& args, const ProxArgs& pargs, const Ctrl* ctrl, hipStream_t stream) {
  ProxArgs pa = pargs;
  const bool need_ell = pa.prox == PROX_HINGE || pa.prox == PROX_01 || pa.objx == OBJX_HINGE ||
                        pa.objx == OBJX_ZEROONE || pa.objx == OBJX_DOT;
  if (!need_ell) pa.ell = nullptr;
  pa.zgiven = nullptr;
  pa.lb = pa.ub = nullptr;
  pa.rhs_add = nullptr;
  hipLaunchKernelGGL(uw_prox_kernel, dim3(static_cast<unsigned>(args.nblk)), dim3(kUwProxThreads), 0, stream, args, pa,
                     ctrl);
}

// ---------------------------------------------------------------- ONE pass over D per iteration (tall, narrow D)
// An A = D iteration reads D twice -- D*x for the element update, D'*(c + z - u) for the next x-update -- although both
// uses of a ROW of D belong to the same iteration: with x known, rows r0 .. r0+63 give (D x)_r, the new z_r, u_r, and
// their contribution D_r'*(c + z - u)_r to the next right-hand side.  Where a 64-row block of D fits the registers of a
// workgroup (n <= 448: 8 waves x 56 columns x 64 lanes), D is read ONCE: BASELINE config 3 at MNIST's full size
// (60000 x 400: 192 MB per pass) is two such passes in the generic path.  Lane = row, wave w holds the columns
// j = w (mod 8) of the block (every load of a lane issued before any is used: 205 KB in flight per workgroup);
//   1. (D x)_r: per-wave partial over its columns, the eight partials summed in wave order through LDS;
//   2. wave 0: the fused element update (prox_apply, the code every loop shares) -> t_r = (c + z - u)_r to LDS;
//   3. every wave: column sums of D_rj t_r over the 64 rows -- four at a time by the permlane / DPP reduce-scatter of
//      the symmetric x-solve (56 __shfl_down sums per wave and block made the kernel slower than the two passes it
//      replaces: 113 against 79 us per iteration at 60000 x 400) --, accumulated per workgroup in LDS.
// Persistent workgroups walk the row blocks; a workgroup's partial row of the right-hand side is written once, at the
// end, and summed (with the iteration's deferred finalize as passenger) by sum_partials_t_fin_kernel.  Requires a loop
// that records no dual residual (unwrappedadmm.m:92 / nodualerror: no D'*(z - zprev), no D'*u) and plain ADMM.
constexpr int kOpRows = 64, kOpWaves = 8, kOpCols = 56, kOpMaxN = kOpWaves * kOpCols;  // 448

__global__ __launch_bounds__(kOpWaves* kWave) void ad_onepass_kernel(OnePassArgs a, ProxArgs pa,
                                                                    const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t it = ctrl->iter;
  __shared__ double xs[kOpMaxN], gacc[kOpMaxN];
  __shared__ double axr[kOpWaves][kOpRows];
  __shared__ double tsh[kOpRows];
  __shared__ double sred[S_COUNT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t n = a.n, m = a.m;
  for (int j = tid; j < kOpMaxN; j += kOpWaves * kWave) {
    xs[j] = j < n ? a.x[j] : 0.0;
    gacc[j] = 0.0;
  }
  __syncthreads();
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  const int64_t nblocks = (m + kOpRows - 1) / kOpRows;
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int64_t r = blk * kOpRows + lane;
    const int64_t rc = r < m ? r : m - 1;
    const double* __restrict__ drow = a.D + rc;
    double d[kOpCols];
#pragma unroll
    for (int k = 0; k < kOpCols; ++k) {  // clamped columns: the loads stay unconditional, x is zero beyond n
      const int64_t j = w + kOpWaves * k;
      d[k] = drow[(j < n ? j : n - 1) * a.ldD];
    }
    ProxIn in{};
    if (w == 0) in = prox_load(pa, rc);  // in flight with the block
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < kOpCols; ++k) p = __builtin_fma(d[k], xs[w + kOpWaves * k], p);
    axr[w][lane] = p;
    __syncthreads();
    if (w == 0) {
      double ax = axr[0][lane];
#pragma unroll
      for (int q = 1; q < kOpWaves; ++q) ax += axr[q][lane];
      double t = 0.0;
      if (r < m) prox_apply(pa, r, ax, it, 0.0, in, acc, &t);
      tsh[lane] = t;  // rows beyond m contribute nothing
    }
    __syncthreads();
    const double t = tsh[lane];
    static_assert(kOpCols % kSyPanel == 0, "column sums are taken four at a time");
#pragma unroll
    for (int k = 0; k < kOpCols; k += kSyPanel) {  // four column sums per permlane / DPP reduce-scatter (wave_reduce.h)
      double q[kSyPanel];
#pragma unroll
      for (int c = 0; c < kSyPanel; ++c) q[c] = d[k + c] * t;
      const double sum = reduce_scatter4(q);  // lanes 16c .. 16c+15 hold the sum of column k + c
      const int j = w + kOpWaves * (k + (lane >> 4));
      if ((lane & 15) == 0 && j < n) gacc[j] += sum;  // (each wave owns its columns)
    }
  }
  __syncthreads();
  double* __restrict__ gout = a.gpart + static_cast<int64_t>(blockIdx.x) * a.ldg;
  for (int j = tid; j < n; j += kOpWaves * kWave) gout[j] = gacc[j];
  if (w == 0) {  // the block partials of the residual sums: wave 0 made all of them
#pragma unroll
    for (int s = 0; s < S_COUNT; ++s) {
      const double v = wave_sum(acc[s]);
      if (lane == 0) sred[s] = v;
    }
    if (lane < S_COUNT) pa.part[lane * kMaxPartBlocks + blockIdx.x] = sred[lane];
  }
}

bool onepass_supported(int64_t m, int64_t n) {
  // (worth it where the passes over D are what an iteration costs: below ~48 MB the launches are)
  return n >= 1 && n <= kOpMaxN && m * n * 8 >= (int64_t{48} << 20) && std::getenv("ADMM_HIP_NO_ONEPASS") == nullptr;
}

int onepass_workgroups(int64_t m) {
  const int64_t nblocks = ceil_div(m, int64_t{kOpRows});
  return static_cast<int>(std::min<int64_t>(nblocks, 256));  // one per CU: a block is 205 KB of registers' worth of loads
}

void launch_ad_onepass(const OnePassArgs& a, const ProxArgs& pargs, const Ctrl* ctrl, int* nblk_out, hipStream_t stream) {
  ProxArgs pa = pargs;
  const bool need_ell = pa.prox == PROX_HINGE || pa.prox == PROX_01 || pa.objx == OBJX_HINGE || pa.objx == OBJX_ZEROONE ||
                        pa.objx == OBJX_DOT;
  if (!need_ell) pa.ell = nullptr;
  if (pa.prox != PROX_GIVEN) pa.zgiven = nullptr;
  if (pa.prox != PROX_BOX) pa.lb = pa.ub = nullptr;
  pa.rhs_add = nullptr;
  pa.rhs = nullptr;  // t = c + z - u never leaves the kernel
  pa.dz = nullptr;
  const int nwg = onepass_workgroups(a.m);
  *nblk_out = nwg;
  hipLaunchKernelGGL(ad_onepass_kernel, dim3(static_cast<unsigned>(nwg)), dim3(kOpWaves * kWave), 0, stream, a, pa, ctrl);
}

}  // namespace admm
