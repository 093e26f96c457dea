// tv.h -- argument block and launchers of the 1-D total-variation kernels (tv.hip).
#pragma once
#include "loop_kernels.h"
#include <vector>

#include "common.h"

namespace admm {

struct TvArgs {
  int64_t n;
  double rho, thresh;      // thresh = lambda/rho   (getProxOps.m:199)
  const double* s;         // noisy signal
  const double* z;         // current z, u (read)
  const double* u;
  double* zo;              // next z, u (written by the prox kernel; ping-pong)
  double* uo;
  double* x;
  double* y;               // forward-sweep intermediate
  const double* bprefix;   // leading LDL' pivots until they become stationary
  int64_t nprefix;
  double bstar;            // stationary pivot
  int32_t halo, elems, tile;
  int32_t objevals;
  double* xhist;
  double* zhist;
  double* uhist;
  double* part;            // [S_COUNT][kMaxPartBlocks]
  // fused iteration kernel (tv_fused_kernel): y ping-pong and its own tile size
  const double* yin;       // forward-sweep result of THIS iteration (written by the previous launch)
  double* yout;            // forward-sweep result for the NEXT iteration
  int32_t ftile;           // owned positions per tile = 256*elems - 2*halo - 4
  int64_t part_stride;     // fused kernel: part is [S_COUNT][part_stride], one column per tile
  int32_t skip_x;          // fused kernel: do not store x (no history wanted): the engine materialises the final x
                           // with one backward sweep of the surviving y after the loop -- 7 vector passes instead of 8
  // one-launch iteration: the tile partials are summed and the finalize logic runs inside the fused kernel.  Groups
  // of kTvGroup consecutive tiles: the last tile of a group to arrive sums the group's partials into gpart
  // ([S_COUNT][kMaxPartBlocks], one column per group), the last GROUP to arrive runs finalize_body on them.
  int32_t* gcount;         // [ngroups + 1] arrival counters (zero between launches); null = two extra launches instead
  double* gpart;
  int32_t ngroups;
  // deferred tail (default): the tile partials of iteration i are summed and its finalize logic run by ONE extra
  // workgroup of iteration i + 1's launch (~40 us of serial work hidden behind 220 us of tiles); the tile partials are
  // double-buffered, the iteration index comes from the host (the passenger advances ctrl->iter during the launch)
  int32_t deferred;        // 1: grid = tiles + 1, `it` = iter_host
  int32_t fin_pending;     // the passenger has an iteration to finalize
  int64_t iter_host;
  const double* prev_part; // tile partials of the previous iteration, [S_COUNT][part_stride]
  int32_t prev_ntiles;
  double* slots16;         // where the passenger leaves the 16 sums for finalize_body (FinArgs::slots_reduced)
  // direct form (tv_direct_kernel): window margin (multiple of 8, >= halo + 8) and the Green's-function scale
  int32_t margin;
  double green;            // A = 1/(b*(1 - r^2)), r = rho/b*
  // direct form, compact dual state: z (read) and zo (written) hold v = z + u; state_in = 0 on a run's first iteration,
  // which reads z and u as given.  uo is not used.
  int32_t state_in;
  // tv_direct2.h: r^1 .. r^8 (r = rho/b*), computed on the host so that they reach the kernel as scalar registers
  double rpow[8];
};
constexpr int kTvGroup = 64;
constexpr int kTvDirectE = 8;  // positions per thread of tv_direct_kernel (tile = 256 * kTvDirectE window positions)

// pivots of I + rho*D'D and the launch geometry for a given rho
int tv_plan(double rho, int64_t n, std::vector<double>* prefix, double* bstar, int* halo, int* elems, int* tile);
void launch_tv_sweep(const TvArgs& a, bool backward, const Ctrl* ctrl, hipStream_t stream);
void launch_tv_prox(const TvArgs& a, const Ctrl* ctrl, int* nblk_out, hipStream_t stream);
// One launch per iteration: backward sweep (x) + z/u update + residual sums + the NEXT iteration's forward
// sweep, 8 vector passes instead of 11.  Available when tv_fused_ok(a) (small halo: elems == 8).
// Unfused building blocks for the fast / accelerated ADMM variants (the generic prox_kernel does the z/u/v/uhat
// work): ax = D*x as a vector plus the objective 1/2||x-s||^2 + lambda*sum|x_{i+1}-x_i| in block partials, and the
// D' stencils of the dual residual / tolerance from dz = z - zprev and u.
void launch_tv_dx(const double* x, const double* s, int64_t n, double lambda, int objevals, double* ax,
                  double* objpart, int* nobj_out, const Ctrl* ctrl, hipStream_t stream);
void launch_tv_dual(const double* dz, const double* u, int64_t n, double* part, int nblk, const Ctrl* ctrl,
                    hipStream_t stream);
// z of the relaxed iteration as the reference computes it (D applied to Axhat): zgiven = soft(u + D*Axhat, t)
void launch_tv_relax_z(const double* ax, const double* zp, const double* u, int64_t n, double relax, double t,
                       double* zgiven, const Ctrl* ctrl, hipStream_t stream);
bool tv_fused_ok(const TvArgs& a);
// slots16 receives the 16 reduction slots summed over all tiles (FinArgs::slots_reduced); with a.gcount set the
// launch ends the iteration itself (fin = the finalize arguments) and slots16 is unused
void launch_tv_fused(const TvArgs& a, const FinArgs& fin, double* slots16, const Ctrl* ctrl, hipStream_t stream);
// One launch per iteration without the y vector: 5 vector passes (tv.hip: tv_direct_kernel).  a.ftile = 2048 - 2*margin.
int tv_direct_margin(const TvArgs& a);
bool tv_direct_ok(const TvArgs& a);
void launch_tv_direct(const TvArgs& a, const FinArgs& fin, const Ctrl* ctrl, hipStream_t stream);
// slots16[s] = sum over the tiles of part[s][.] (the stand-alone form of the deferred tail's first half)
void launch_tv_pack(const double* part, int64_t stride, int32_t ntiles, double* slots16, const Ctrl* ctrl,
                    hipStream_t stream);

}  // namespace admm
