// tv.hip -- 1-D total-variation ADMM kernels (solvers/totalvariation.m, getProxOps.m:145-199).
//
//   x-update  (getProxOps.m:1047)  x = (I + rho*D'D) \ (s + rho*D'(z-u)),  D = spdiags([1 -1],0:1,n,n)
//   z-update  (getProxOps.m:199)   z = soft(u + D*x, lambda/rho),  then u += D*x - z  (admm.m:548)
//
// I + rho*D'D is the constant SPD tridiagonal [-rho, 1+rho*(1|2|..|2), -rho].  The reference
// re-assembles and re-factors it sparsely every iteration; here its LDL' pivots are computed
// once (host, O(n)) and each solve is two first-order recurrences
//     forward   y_i = r_i + (rho/b_{i-1}) * y_{i-1}
//     backward  x_i = y_i/b_i + (rho/b_i) * x_{i+1}
// evaluated as block-parallel affine-map scans.  The recurrence multipliers are < 1 in
// magnitude (strict diagonal dominance), so an element depends on anything further than H
// positions away by less than 1e-18 relative: each workgroup warms up on an H-element halo
// instead of waiting for its predecessor -- no inter-workgroup communication, pure streaming.
// The D / D' stencils and the rhs assembly are fused into the sweeps and the prox kernel.
//
// Three forms of an iteration, chosen by engine_run_tv.hip:
//   tv_direct_kernel   (default, small halo)  one launch, 5 vector passes: away from the two ends of the matrix the
//                      solve is the truncated two-sided exponential kernel of the Toeplitz operator, evaluated per
//                      thread from LDS without scans; only the two end tiles run the recurrences (as block scans)
//   tv_fused_kernel    one launch, 7 passes: this iteration's backward sweep + z/u update + the NEXT iteration's
//                      forward sweep, the intermediate y travelling between launches (ADMM_HIP_TV_SCAN=1)
//   tv_sweep x 2 + tv_prox   three launches (large halos: rho >~ 16), and the building blocks of the fast /
//                      accelerated / relaxed variants (tv_dx, tv_dual, tv_relax_z around the generic prox kernel)
#include <cstdlib>

#include "kernels.h"
#include "loop_kernels.h"
#include "tv.h"
#include "tv_direct2.h"
#include "finalize_device.h"

namespace admm {

typedef double double2_t __attribute__((ext_vector_type(2)));

// recurrence multiplier: forward a_i = rho/b_{i-1} (a_0 = 0); backward a_i = rho/b_i, 0 for the last row
template <bool BWD>
__device__ __forceinline__ double tv_coef(const TvArgs& a, int64_t i, int64_t n, double rho, double cstar) {
  if (!BWD) {
    if (i <= 0) return 0.0;
    return (i - 1 < a.nprefix) ? rho / a.bprefix[i - 1] : cstar;
  }
  if (i >= n - 1) return 0.0;
  return (i < a.nprefix) ? rho / a.bprefix[i] : cstar;
}

// One sweep.  Logical position q in [0, count) maps to global index i = p0 + q (forward) or
// p1 - 1 - q (backward); the scan always runs in increasing q with zero incoming carry.
template <int E, bool BWD>
__global__ __launch_bounds__(kBlock) void tv_sweep_kernel(TvArgs a, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // padded: pos(q) = q + (q >> 4)
  __shared__ double wA[4], wB[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t n = a.n;
  const int64_t o0 = static_cast<int64_t>(blockIdx.x) * a.tile;
  const int64_t o1 = (o0 + a.tile < n) ? o0 + a.tile : n;
  int64_t p0, p1;
  if (!BWD) {
    p0 = (o0 - a.halo > 0) ? o0 - a.halo : 0;
    p1 = o1;
  } else {
    p0 = o0;
    p1 = (o1 + a.halo < n) ? o1 + a.halo : n;
  }
  const int count = static_cast<int>(p1 - p0);
  const double rho = a.rho;
  const double cstar = rho / a.bstar;   // stationary multiplier (no per-element division past the prefix)
  const double ibstar = 1.0 / a.bstar;

  // ---- stage the per-element constant term r (global -> padded LDS).  Walks i ascending in
  // aligned pairs (16-byte loads; p0 is even by construction) whatever the sweep direction, and is
  // fully unrolled so that all of a thread's independent loads are in flight together.
  auto qpos = [&](int64_t i) -> int { return BWD ? static_cast<int>(p1 - 1 - i) : static_cast<int>(i - p0); };
#pragma unroll
  for (int k = 0; k < E / 2; ++k) {
    const int j = tid + k * kBlock;  // pair index
    const int64_t i0 = p0 + 2 * static_cast<int64_t>(j);
    const bool live0 = 2 * j < count, live1 = 2 * j + 1 < count;
    double r0 = 0.0, r1 = 0.0;
    if (!BWD) {  // r_i = s_i + rho*(D'(z-u))_i     getProxOps.m:1047
      double t0 = 0.0, t1 = 0.0, s0 = 0.0, s1 = 0.0;
      if (live1) {
        const double2_t zz = *reinterpret_cast<const double2_t*>(a.z + i0);
        const double2_t uu = *reinterpret_cast<const double2_t*>(a.u + i0);
        const double2_t ss = *reinterpret_cast<const double2_t*>(a.s + i0);
        t0 = zz.x - uu.x;
        t1 = zz.y - uu.y;
        s0 = ss.x;
        s1 = ss.y;
      } else if (live0) {
        t0 = a.z[i0] - a.u[i0];
        s0 = a.s[i0];
      }
      double tm = __shfl_up(t1, 1, 64);  // element i0-1 is the previous lane's second element
      if (lane == 0 && live0 && i0 > 0) tm = a.z[i0 - 1] - a.u[i0 - 1];
      r0 = s0 + rho * ((i0 > 0) ? t0 - tm : t0);
      r1 = s1 + rho * (t1 - t0);
    } else {
      double y0 = 0.0, y1 = 0.0;
      if (live1) {
        const double2_t yy = *reinterpret_cast<const double2_t*>(a.y + i0);
        y0 = yy.x;
        y1 = yy.y;
      } else if (live0) {
        y0 = a.y[i0];
      }
      r0 = y0 * ((i0 < a.nprefix) ? 1.0 / a.bprefix[i0] : ibstar);
      r1 = y1 * ((i0 + 1 < a.nprefix && live1) ? 1.0 / a.bprefix[i0 + 1] : ibstar);
    }
    if (live0) {
      const int q = qpos(i0);
      lds[q + (q >> 4)] = r0;
    }
    if (live1) {
      const int q = qpos(i0 + 1);
      lds[q + (q >> 4)] = r1;
    }
  }
  __syncthreads();

  // ---- thread-local composite over its E consecutive positions: y_out = A*y_in + B
  const int q0 = tid * E;
  double A = 1.0, B = 0.0;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int q = q0 + k;
    if (q < count) {
      const int64_t i = BWD ? (p1 - 1 - q) : (p0 + q);
      const double c = tv_coef<BWD>(a, i, n, rho, cstar);
      B = c * B + lds[q + (q >> 4)];
      A = c * A;
    }
  }
  // ---- exclusive scan of (A,B) across the block: wave shuffles, then the 4 wave totals
  double sA = A, sB = B;  // inclusive within the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double pA = __shfl_up(sA, off, 64), pB = __shfl_up(sB, off, 64);
    if (lane >= off) {
      sB = sA * pB + sB;
      sA = sA * pA;
    }
  }
  if (lane == 63) {
    wA[wid] = sA;
    wB[wid] = sB;
  }
  __syncthreads();
  // carry entering this wave = composition of the previous waves applied to 0
  double carry = 0.0;
  for (int w = 0; w < wid; ++w) carry = wA[w] * carry + wB[w];
  // exclusive value for this thread: apply the inclusive prefix of the previous lane
  double eA = __shfl_up(sA, 1, 64), eB = __shfl_up(sB, 1, 64);
  if (lane == 0) {
    eA = 1.0;
    eB = 0.0;
  }
  double yin = eA * carry + eB;

  // ---- recompute the thread's positions with the true incoming value, write results to LDS
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int q = q0 + k;
    if (q < count) {
      const int64_t i = BWD ? (p1 - 1 - q) : (p0 + q);
      const double c = tv_coef<BWD>(a, i, n, rho, cstar);
      yin = c * yin + lds[q + (q >> 4)];
      lds[q + (q >> 4)] = yin;
    }
  }
  __syncthreads();
  // ---- coalesced store of the owned range (ascending pairs, 16-byte stores)
  const int64_t it = ctrl->iter;
  double* __restrict__ out = BWD ? a.x : a.y;
  double* __restrict__ hist = (BWD && a.xhist) ? a.xhist + it * n : nullptr;
#pragma unroll
  for (int k = 0; k < E / 2; ++k) {
    const int j = tid + k * kBlock;
    const int64_t i0 = p0 + 2 * static_cast<int64_t>(j);
    const bool own0 = 2 * j < count && i0 >= o0 && i0 < o1;
    const bool own1 = 2 * j + 1 < count && i0 + 1 >= o0 && i0 + 1 < o1;
    double v0 = 0.0, v1 = 0.0;
    if (own0) {
      const int q = qpos(i0);
      v0 = lds[q + (q >> 4)];
    }
    if (own1) {
      const int q = qpos(i0 + 1);
      v1 = lds[q + (q >> 4)];
    }
    if (own0 && own1) {
      *reinterpret_cast<double2_t*>(out + i0) = double2_t{v0, v1};
      if (hist) {  // history columns start at it*n, which is odd-aligned for odd n: scalar stores
        hist[i0] = v0;
        hist[i0 + 1] = v1;
      }
    } else {
      if (own0) {
        out[i0] = v0;
        if (hist) hist[i0] = v0;
      }
      if (own1) {
        out[i0 + 1] = v1;
        if (hist) hist[i0 + 1] = v1;
      }
    }
  }
}

__device__ __forceinline__ double tv_soft(double v, double t) {
  const double q = fabs(v) - t;
  const double p = q > 0.0 ? q : 0.0;
  return (v > 0.0) ? p : ((v < 0.0) ? -p : 0.0 * p);
}

// z/u update with the D stencil, every residual partial sum, and the D' stencils of the dual
// residual (admm.m:624) and dual tolerance (admm.m:654).  Reads z,u (old) and writes zo,uo
// (ping-pong) because element i also needs the NEW values of element i-1.
__global__ __launch_bounds__(kBlock) void tv_prox_kernel(TvArgs a, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t it = ctrl->iter;
  const int64_t n = a.n;
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double xi = a.x[i];
    const double xn = (i + 1 < n) ? a.x[i + 1] : 0.0;
    const double ax = (i + 1 < n) ? xi - xn : xi;  // D*x, last row is x_n (totalvariation.m:127)
    const double zp = a.z[i], uo = a.u[i];
    const double zn = tv_soft(uo + ax, a.thresh);  // getProxOps.m:199
    const double Bz = -zn;
    const double un = uo + (ax + Bz);              // admm.m:548 (c = 0)
    const double r = ax + Bz;
    const double dz = zn - zp;
    // the same quantities for element i-1 (recomputed; feeds the D' stencils)
    double dzm = 0.0, unm = 0.0;
    if (i > 0) {
      const double axm = a.x[i - 1] - xi;
      const double zpm = a.z[i - 1], uom = a.u[i - 1];
      const double znm = tv_soft(uom + axm, a.thresh);
      unm = uom + (axm + (-znm));
      dzm = znm - zpm;
    }
    const double g2 = dz - dzm;  // (D'(z - zprev))_i
    const double g3 = un - unm;  // (D'u)_i
    acc[S_R2] += r * r;
    acc[S_AX2] += ax * ax;
    acc[S_Z2] += zn * zn;
    acc[S_DZ2] += dz * dz;
    acc[S_U2] += un * un;
    const double du = un - uo;
    acc[S_DU2] += du * du;
    acc[S_G2] += g2 * g2;
    acc[S_G3] += g3 * g3;
    if (a.objevals) {  // totalvariation.m:134-135
      if (i + 1 < n) acc[S_OBJZ] += fabs(xn - xi);
      const double e = xi - a.s[i];
      acc[S_OBJX] += e * e;
    }
    a.zo[i] = zn;
    a.uo[i] = un;
    if (a.zhist) a.zhist[it * n + i] = zn;
    if (a.uhist) a.uhist[it * n + i] = un;
  }
  // block partials
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

// ---------------------------------------------------------------------------------------------
// Fused iteration kernel.  Per tile [o0, o1) with window [w0, w1) = [o0-H-2, o1+H+2):
//   1. stage y/b over the window, backward scan             -> x on [w0, w1-H)
//   2. z/u update with the D stencil on [w0, o1)             -> z+, u+, t = z+ - u+, every residual sum
//   3. r = s + rho*D'(t) on [o0-H, o1), forward scan         -> y of the NEXT iteration on [o0, o1)
// so an iteration reads y, z, u, s and writes x, z, u, y: 8 vector passes instead of the 11 of the
// three-kernel form, in one launch.  The halo elements are recomputed by both neighbours (2H+4 of
// 2048 positions); their z+/u+ are bit-identical on both sides because they depend on x only through
// positions whose backward-scan halo error is below 1e-18 relative -- the same argument as for the
// sweeps -- and only the owner stores them.
// A block-wide scan of y_out = c*y_in + r with zero incoming carry over `count` LDS positions
// (pos(q) = q + (q >> 4)); COEF(q) gives the multiplier of position q.
template <int E, typename COEF>
__device__ __forceinline__ void tv_block_scan(double* __restrict__ lds, int count, COEF coef, double* wA, double* wB) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int q0 = tid * E;
  double A = 1.0, B = 0.0;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int q = q0 + k;
    if (q < count) {
      const double c = coef(q);
      B = c * B + lds[q + (q >> 4)];
      A = c * A;
    }
  }
  double sA = A, sB = B;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double pA = __shfl_up(sA, off, 64), pB = __shfl_up(sB, off, 64);
    if (lane >= off) {
      sB = sA * pB + sB;
      sA = sA * pA;
    }
  }
  __syncthreads();  // wA/wB may still be read by the previous scan's tail
  if (lane == 63) {
    wA[wid] = sA;
    wB[wid] = sB;
  }
  __syncthreads();
  double carry = 0.0;
  for (int w = 0; w < wid; ++w) carry = wA[w] * carry + wB[w];
  double eA = __shfl_up(sA, 1, 64), eB = __shfl_up(sB, 1, 64);
  if (lane == 0) {
    eA = 1.0;
    eB = 0.0;
  }
  double yin = eA * carry + eB;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int q = q0 + k;
    if (q < count) {
      yin = coef(q) * yin + lds[q + (q >> 4)];
      lds[q + (q >> 4)] = yin;
    }
  }
  __syncthreads();
}

// (r2: issuing every global load of a tile -- z, u, s pairs and the wave-boundary neighbours -- before the first scan
// costs 40 more VGPRs: 3 workgroups per CU instead of 4 and 0.2390 against 0.2331 ms per iteration on the same box,
// 2 workgroups 0.3165.  Residency, not the number of dependent load rounds inside a workgroup, carries this kernel.)
template <int E, int MODE>  // MODE 0: default stores; 1: streaming stores; 2: + s kept cacheable; 3: + y kept cacheable; 4: both
__global__ __launch_bounds__(kBlock, 4) void tv_fused_kernel(TvArgs a, FinArgs fin, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ int32_t tail_group, tail_all;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int kCap = E * kBlock;
  double* __restrict__ L1 = lds;                        // y/b -> x (backward positions)
  double* __restrict__ L2 = lds + kCap + (kCap >> 4) + 1;  // forward right-hand side -> next y
  __shared__ double wA[4], wB[4];
  __shared__ double sred[4][S_COUNT];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (a.deferred && blockIdx.x == 0) {  // the passenger (dispatched first): tail of the PREVIOUS iteration
    if (!a.fin_pending) return;
    {  // all slots at once: 16 lanes per slot stride over the tiles, 16 loads in flight per lane; fixed order
      const int slot = tid >> 4, sub = tid & 15;
      double v = 0.0;
      if (slot < S_COUNT) {
        const double* __restrict__ ps = a.prev_part + slot * a.part_stride;
        for (int32_t b0 = 0; b0 < a.prev_ntiles; b0 += 256) {
          double w[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int32_t b = b0 + sub + 16 * k;
            w[k] = ps[b < a.prev_ntiles ? b : a.prev_ntiles - 1];
          }
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (b0 + sub + 16 * k < a.prev_ntiles) v += w[k];
        }
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (slot < S_COUNT && sub == 0) a.slots16[slot] = v;
    }
    __threadfence_block();
    __syncthreads();
    finalize_body<false>(fin);  // fin.slots_reduced = a.slots16
    return;
  }
  const unsigned tile_id = a.deferred ? blockIdx.x - 1u : blockIdx.x;
  const int64_t n = a.n, it = a.deferred ? a.iter_host : ctrl->iter;
  const double rho = a.rho;
  const double cstar = rho / a.bstar, ibstar = 1.0 / a.bstar;
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;

  {
    const int64_t o0 = static_cast<int64_t>(tile_id) * a.ftile;
    const int64_t o1 = (o0 + a.ftile < n) ? o0 + a.ftile : n;
    const int64_t w0 = (o0 - a.halo - 2 > 0) ? o0 - a.halo - 2 : 0;          // even
    const int64_t w1 = (o1 + a.halo + 2 < n) ? o1 + a.halo + 2 : n;
    const int64_t f0 = (o0 - a.halo > 0) ? o0 - a.halo : 0;                  // forward scan start (even)
    const int count = static_cast<int>(w1 - w0);
    const int fcount = static_cast<int>(o1 - f0);
    auto qb = [&](int64_t i) -> int { const int q = static_cast<int>(w1 - 1 - i); return q + (q >> 4); };
    auto qf = [&](int64_t i) -> int { const int q = static_cast<int>(i - f0); return q + (q >> 4); };

    // ---- 1. y/b over the window into L1 (backward order)
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
      const int j = tid + k * kBlock;
      const int64_t i0 = w0 + 2 * static_cast<int64_t>(j);
      const bool live0 = 2 * j < count, live1 = 2 * j + 1 < count;
      double y0 = 0.0, y1 = 0.0;
      if (live1) {
        const admm_double2 yy = load2<!(MODE == 3 || MODE == 4)>(a.yin + i0);
        y0 = yy.x;
        y1 = yy.y;
      } else if (live0) {
        y0 = a.yin[i0];
      }
      if (live0) L1[qb(i0)] = y0 * ((i0 < a.nprefix) ? 1.0 / a.bprefix[i0] : ibstar);
      if (live1) L1[qb(i0 + 1)] = y1 * ((i0 + 1 < a.nprefix) ? 1.0 / a.bprefix[i0 + 1] : ibstar);
    }
    __syncthreads();
    tv_block_scan<E>(L1, count,
                     [&](int q) { return tv_coef<true>(a, w1 - 1 - q, n, rho, cstar); }, wA, wB);

    // ---- 2. z/u update (pairs, ascending), forward right-hand side into L2, owned outputs
#pragma unroll 1
    for (int k = 0; k < E / 2; ++k) {
      const int j = tid + k * kBlock;
      const int64_t i0 = w0 + 2 * static_cast<int64_t>(j);
      const bool live0 = 2 * j < count, live1 = 2 * j + 1 < count;
      double zn0 = 0.0, zn1 = 0.0, un0 = 0.0, un1 = 0.0, dz0 = 0.0, dz1 = 0.0, s0 = 0.0, s1 = 0.0;
      double x0 = 0.0, x1 = 0.0, x2 = 0.0, ax0 = 0.0, ax1 = 0.0, uo0 = 0.0, uo1 = 0.0;
      if (live0) {
        x0 = L1[qb(i0)];
        if (live1) x1 = L1[qb(i0 + 1)];
        if (i0 + 2 < w1) x2 = L1[qb(i0 + 2)];
        double zp0, zp1 = 0.0;
        if (live1) {
          const admm_double2 zz = load2<true>(a.z + i0), uu = load2<true>(a.u + i0),
                             ss = load2<!(MODE == 2 || MODE == 4)>(a.s + i0);
          zp0 = zz.x;
          zp1 = zz.y;
          uo0 = uu.x;
          uo1 = uu.y;
          s0 = ss.x;
          s1 = ss.y;
        } else {
          zp0 = a.z[i0];
          uo0 = a.u[i0];
          s0 = a.s[i0];
        }
        ax0 = (i0 + 1 < n) ? x0 - x1 : x0;  // D*x, last row is x_n (totalvariation.m:127)
        zn0 = tv_soft(uo0 + ax0, a.thresh);  // getProxOps.m:199
        un0 = uo0 + (ax0 + (-zn0));          // admm.m:548 (c = 0)
        dz0 = zn0 - zp0;
        if (live1) {
          ax1 = (i0 + 2 < n) ? x1 - x2 : x1;
          zn1 = tv_soft(uo1 + ax1, a.thresh);
          un1 = uo1 + (ax1 + (-zn1));
          dz1 = zn1 - zp1;
        }
      }
      // the left neighbour (element i0-1) is the previous lane's second element
      double znm = __shfl_up(zn1, 1, 64), unm = __shfl_up(un1, 1, 64), dzm = __shfl_up(dz1, 1, 64);
      if (lane == 0 && live0 && i0 > w0) {  // wave boundary: recompute element i0-1
        const double xm = L1[qb(i0 - 1)];
        const double zpm = a.z[i0 - 1], uom = a.u[i0 - 1];
        const double axm = xm - x0;
        znm = tv_soft(uom + axm, a.thresh);
        unm = uom + (axm + (-znm));
        dzm = znm - zpm;
      }
      const bool left = i0 > 0;  // i0 == w0 > 0 never feeds an owned or forward position (window margin of 2)
      if (live0) {
        const double t0 = zn0 - un0, tm = znm - unm, t1 = zn1 - un1;
        if (i0 >= f0 && i0 < o1) L2[qf(i0)] = s0 + rho * (left ? t0 - tm : t0);  // getProxOps.m:1047
        if (live1 && i0 + 1 >= f0 && i0 + 1 < o1) L2[qf(i0 + 1)] = s1 + rho * (t1 - t0);
        const bool own0 = i0 >= o0 && i0 < o1, own1 = live1 && i0 + 1 >= o0 && i0 + 1 < o1;
        if (own0) {
          const double r = ax0 + (-zn0), du = un0 - uo0;
          const double g2 = left ? dz0 - dzm : dz0, g3 = left ? un0 - unm : un0;
          acc[S_R2] += r * r;
          acc[S_AX2] += ax0 * ax0;
          acc[S_Z2] += zn0 * zn0;
          acc[S_DZ2] += dz0 * dz0;
          acc[S_U2] += un0 * un0;
          acc[S_DU2] += du * du;
          acc[S_G2] += g2 * g2;
          acc[S_G3] += g3 * g3;
          if (a.objevals) {  // totalvariation.m:134-135
            if (i0 + 1 < n) acc[S_OBJZ] += fabs(x1 - x0);
            const double e = x0 - s0;
            acc[S_OBJX] += e * e;
          }
        }
        if (own1) {
          const double r = ax1 + (-zn1), du = un1 - uo1;
          const double g2 = dz1 - dz0, g3 = un1 - un0;
          acc[S_R2] += r * r;
          acc[S_AX2] += ax1 * ax1;
          acc[S_Z2] += zn1 * zn1;
          acc[S_DZ2] += dz1 * dz1;
          acc[S_U2] += un1 * un1;
          acc[S_DU2] += du * du;
          acc[S_G2] += g2 * g2;
          acc[S_G3] += g3 * g3;
          if (a.objevals) {
            if (i0 + 2 < n) acc[S_OBJZ] += fabs(x2 - x1);
            const double e = x1 - s1;
            acc[S_OBJX] += e * e;
          }
        }
        if (own0 && own1) {
          constexpr bool NTS = MODE != 0;
          if (!a.skip_x) store2<NTS>(a.x + i0, admm_double2{x0, x1});
          store2<NTS>(a.zo + i0, admm_double2{zn0, zn1});
          store2<NTS>(a.uo + i0, admm_double2{un0, un1});
        } else {
          if (own0) {
            if (!a.skip_x) a.x[i0] = x0;
            a.zo[i0] = zn0;
            a.uo[i0] = un0;
          }
          if (own1) {
            if (!a.skip_x) a.x[i0 + 1] = x1;
            a.zo[i0 + 1] = zn1;
            a.uo[i0 + 1] = un1;
          }
        }
        if (a.xhist) {  // history columns start at it*n (odd-aligned for odd n): scalar stores
          if (own0) {
            a.xhist[it * n + i0] = x0;
            a.zhist[it * n + i0] = zn0;
            a.uhist[it * n + i0] = un0;
          }
          if (own1) {
            a.xhist[it * n + i0 + 1] = x1;
            a.zhist[it * n + i0 + 1] = zn1;
            a.uhist[it * n + i0 + 1] = un1;
          }
        }
      }
    }
    // block partials of the residual sums (before the second scan: keeps the accumulators short-lived)
#pragma unroll
    for (int s = 0; s < S_COUNT; ++s) {
      const double w = wave_sum(acc[s]);
      if (lane == 0) sred[wid][s] = w;
    }
    __syncthreads();
    if (threadIdx.x < S_COUNT) {
      const int s = threadIdx.x;
      const double t = ((sred[0][s] + sred[1][s]) + sred[2][s]) + sred[3][s];
      if (a.gcount) __hip_atomic_store(a.part + s * a.part_stride + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else a.part[s * a.part_stride + tile_id] = t;
    }
    // ---- 3. forward scan of the next iteration's right-hand side, owned store
    tv_block_scan<E>(L2, fcount, [&](int q) { return tv_coef<false>(a, f0 + q, n, rho, cstar); }, wA, wB);
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
      const int j = tid + k * kBlock;
      const int64_t i0 = f0 + 2 * static_cast<int64_t>(j);
      const bool own0 = 2 * j < fcount && i0 >= o0, own1 = 2 * j + 1 < fcount && i0 + 1 >= o0;
      if (own0 && own1) {
        store2<(MODE == 1 || MODE == 2)>(a.yout + i0, admm_double2{L2[qf(i0)], L2[qf(i0 + 1)]});
      } else {
        if (own0) a.yout[i0] = L2[qf(i0)];
        if (own1) a.yout[i0 + 1] = L2[qf(i0 + 1)];
      }
    }
  }
  if (!a.gcount) return;
  // ---- 4. tail of the one-launch iteration.  This tile's partials were published write-through in step 2; once the
  // wave that stored them has drained, the tile arrives at its group.  (Arriving mid-kernel, right after step 2, was
  // measured slower -- 0.2446 against 0.2349 ms per iteration with the two extra launches: the drain of the z/u
  // stores then stalls wave 0 in front of the barriers of step 3's scan, in every tile.)  The last tile to arrive
  // implies every tile has started and read ctrl at its top, so the finalize logic may advance ctrl.
  if (wid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      const int32_t g = static_cast<int32_t>(blockIdx.x) / kTvGroup;
      const int32_t ntiles = static_cast<int32_t>(gridDim.x);
      const int32_t gsize = (ntiles - g * kTvGroup < kTvGroup) ? ntiles - g * kTvGroup : kTvGroup;
      const int32_t old = __hip_atomic_fetch_add(a.gcount + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tail_group = (old == gsize - 1) ? 1 : 0;
      if (tail_group) __hip_atomic_store(a.gcount + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!tail_group) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  {  // this workgroup arrived last in its group: the group's partials, tile order, 16 lanes per slot
    const int32_t g = static_cast<int32_t>(blockIdx.x) / kTvGroup;
    const int32_t t0 = g * kTvGroup;
    const int32_t ntiles = static_cast<int32_t>(gridDim.x);
    const int32_t gsize = (ntiles - t0 < kTvGroup) ? ntiles - t0 : kTvGroup;
    const int slot = tid >> 4, sub = tid & 15;
    double v = 0.0;
    if (slot < S_COUNT) {
      const double* __restrict__ ps = a.part + slot * a.part_stride + t0;
      double w[kTvGroup / 16];
#pragma unroll
      for (int k = 0; k < kTvGroup / 16; ++k) {
        const int b = sub + 16 * k;
        w[k] = __hip_atomic_load(ps + (b < gsize ? b : gsize - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int k = 0; k < kTvGroup / 16; ++k)
        if (sub + 16 * k < gsize) v += w[k];
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (slot < S_COUNT && sub == 0)
      __hip_atomic_store(a.gpart + slot * kMaxPartBlocks + g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int32_t old = __hip_atomic_fetch_add(a.gcount + a.ngroups, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tail_all = (old == a.ngroups - 1) ? 1 : 0;
      if (tail_all) __hip_atomic_store(a.gcount + a.ngroups, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (!tail_all) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  finalize_body<true>(fin);  // fin.part = gpart, fin.nblk = ngroups
}

// ---------------------------------------------------------------- one launch per iteration WITHOUT the y vector
// tv_fused_kernel splits the tridiagonal solve over two launches (this iteration's backward sweep, the next one's
// forward sweep) and pays for it with the y vector's round trip (7 vector passes) and two block-wide scans per tile --
// a chain of barriers and shuffle steps that keeps a tile resident for ~26 us while it issues loads in two of its eight
// phases: 4.3 TB/s.  Here a tile solves (I + rho*D'D) x = s + rho*D'(z - u) directly from z, u, s (5 passes: reads
// s, z, u; writes z, u):
//   * away from the two ends the matrix is Toeplitz, tridiag(-rho, 1+2rho, -rho), and its inverse is the two-sided
//     exponential kernel  x_i = A * sum_k r^|k| b_(i+k),  r = rho/b* (the stationary multiplier of the sweeps),
//     A = 1/(b*(1 - r^2)); |k| <= K = the sweeps' halo leaves a truncation below 1e-18.  A thread owns 8 consecutive
//     positions: one K-term Horner sum on each side, then the two first-order recurrences across its 8 positions --
//     ~2K + 30 FMAs and as many LDS reads per thread, no barrier, no cross-lane traffic;
//   * the first and the last tile (where the boundary rows reflect) run the exact recurrences as two block scans
//     inside the tile, from the true boundary on one side and with the usual warm-up margin on the other.
// The z/u update, the residual sums, the histories and the deferred tail are those of tv_fused_kernel.  The final x
// (no y to rebuild it from) is recomputed after the loop from the z, u the last executed iteration read: z and u
// rotate through THREE buffers so that the speculative iteration behind a stop does not overwrite them.
//
// Compact dual state (VIN): the loop carries v = z + u, ONE vector, instead of z and u.  With u+ = clamp(v+, -t, t) and
// z+ = v+ - u+ (the soft threshold and admm.m:548 up to the rounding of one subtraction), v+ = u + D x, both old
// iterates are functions of the stored v, recomputed bitwise as the previous pass had them: 3 vector passes (reads v, s;
// writes v) instead of 5.  VIN = false is a run's first iteration (z, u as given: a warm start need not satisfy
// z = soft(z + u)); launch_tv2d_expand turns the last v back into z, u when the run ends.
__device__ __forceinline__ double tv_clamp(double v, double t) { return __builtin_fmin(__builtin_fmax(v, -t), t); }

template <bool NTS, bool VIN>
__global__ __launch_bounds__(kBlock, 4) void tv_direct_kernel(TvArgs a, FinArgs fin, const Ctrl* __restrict__ ctrl) {
  // every kernel argument the tile path reads, requested together with the control block's address: the compiler
  // otherwise fetches them in three dependent groups behind the stop test (four scalar round trips before the first
  // vector load of a 14 us tile)
  asm volatile("" ::"s"(a.n), "s"(a.z), "s"(a.u), "s"(a.s), "s"(a.zo), "s"(a.part), "s"(a.ftile), "s"(a.margin),
               "s"(a.halo), "s"(a.deferred), "s"(a.part_stride), "s"(a.thresh), "s"(a.rho), "s"(a.bstar), "s"(a.green),
               "s"(a.objevals), "s"(a.xhist), "s"(ctrl));
  // the stop flag is asked for here and tested behind the tile's loads (nothing is stored before that): in front of
  // them the test costs every tile one more memory round trip
  // (a scalar load issued by hand and waited for by hand: written as `ctrl->stop` the compiler fetches it through the
  // vector memory path and moves it to a scalar register at once, i.e. waits for it right here.  Between this statement
  // and the wait below the tile path issues only global loads and scalar arithmetic; any s_waitcnt lgkmcnt(0) the compiler
  // adds in between is merely early, and its counted lgkmcnt(n) waits belong to LDS traffic, which starts after the wait.)
  int32_t stop;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(stop) : "s"(ctrl));
  constexpr int E = kTvDirectE;
  constexpr int kCap = E * kBlock;
  constexpr int kSh = (E == 8) ? 3 : 2;            // pad one slot per E positions: E-strided thread positions hit
  constexpr int kArr = kCap + (kCap >> kSh) + 1;   // E + 1 apart, coprime with the 16 bank pairs
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* __restrict__ Bq = lds;         // right-hand side b (forward positions)
  double* __restrict__ Xq = lds + kArr;  // x (forward positions; scan tiles: backward positions)
  __shared__ double wA[4], wB[4];
  __shared__ double sred[4][S_COUNT];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (a.deferred && blockIdx.x == 0) {  // the passenger (dispatched first): tail of the PREVIOUS iteration
    if (ctrl->stop || !a.fin_pending) return;  // (its own read: `stop` must not be needed before the tiles' loads)
    {
      const int slot = tid >> 4, sub = tid & 15;
      double v = 0.0;
      if (slot < S_COUNT) {
        const double* __restrict__ ps = a.prev_part + slot * a.part_stride;
        for (int32_t b0 = 0; b0 < a.prev_ntiles; b0 += 256) {
          double w[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int32_t b = b0 + sub + 16 * k;
            w[k] = ps[b < a.prev_ntiles ? b : a.prev_ntiles - 1];
          }
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (b0 + sub + 16 * k < a.prev_ntiles) v += w[k];
        }
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (slot < S_COUNT && sub == 0) a.slots16[slot] = v;
    }
    __threadfence_block();
    __syncthreads();
    finalize_body<false>(fin);  // fin.slots_reduced = a.slots16
    return;
  }
  const unsigned tile_id = a.deferred ? blockIdx.x - 1u : blockIdx.x;
  const int64_t n = a.n, it = a.deferred ? a.iter_host : ctrl->iter;
  const double rho = a.rho;
  const double r = rho / a.bstar, ibstar = 1.0 / a.bstar;
  const int K = a.halo, M = a.margin;
  const int64_t o0 = static_cast<int64_t>(tile_id) * a.ftile;
  const int64_t o1 = (o0 + a.ftile < n) ? o0 + a.ftile : n;
  const int64_t w0 = (o0 - M > 0) ? o0 - M : 0;  // even (ftile and M are multiples of 8)
  const int64_t w1 = (o1 + M < n) ? o1 + M : n;
  const int count = static_cast<int>(w1 - w0);
  // the Toeplitz kernel is exact to 1e-18 where neither end of the matrix is within K of a needed x
  const bool direct = (o0 - 1 >= K) && (o1 + 1 + K <= n) && (o0 - M >= 0) && (o1 + M <= n);
  auto pos8 = [](int q) -> int { return q + (q >> kSh); };
  auto pos16 = [](int q) -> int { return q + (q >> 4); };

  // ---- 1. z, u, s of the window (kept in registers for step 3); b = s + rho*D'(z - u) into LDS
  // Every load of the tile is issued before any is used: the window pairs AND, in lane 0 of each wave, the element just
  // left of the wave's first pair -- loaded behind the shuffle it cost one more memory round trip per pair group, four
  // per tile, on the tile's critical path.  s is not kept (the objective re-reads it: a cache hit).
  admm_double2 zr[E / 2], ur[E / 2];  // VIN: zr holds v, ur is unused
  double zb[E / 2], ub[E / 2];        // z, u of element i0 - 1 (lane 0 only; steps 1 and 3); VIN: zb holds v
  const double th = a.thresh;
  {
    admm_double2 sr[E / 2];
    // Branch-free: every lane loads a whole pair (lanes past the window, and the single last element of an odd-length
    // signal, from the window's last full pair).  With the loads inside `if (live1) ... else if (live0) ...` hipcc merged
    // the two branches through register copies placed right behind each group of loads -- an s_waitcnt per group, i.e.
    // the four groups of a tile fetched one memory round trip after the other.
    const int64_t wlast = w0 + ((count - 2) & ~1);
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
      const int j = tid + k * kBlock;
      const int64_t i0 = w0 + 2 * static_cast<int64_t>(j);
      const bool live0 = 2 * j < count, live1 = 2 * j + 1 < count;
      const int64_t ip = live1 ? i0 : wlast;
      zr[k] = load2<true>(a.z + ip);
      ur[k] = VIN ? admm_double2{0.0, 0.0} : load2<true>(a.u + ip);
      sr[k] = load2<false>(a.s + ip);
      zb[k] = 0.0;
      ub[k] = 0.0;
      if (lane == 0) {
        const int64_t ib = (live0 && i0 > 0) ? i0 - 1 : 0;
        zb[k] = a.z[ib];
        if (!VIN) ub[k] = a.u[ib];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(stop));
    if (stop) return;  // (uniform)
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
      const int j = tid + k * kBlock;
      const int64_t i0 = w0 + 2 * static_cast<int64_t>(j);
      const bool live0 = 2 * j < count, live1 = 2 * j + 1 < count;
      if (live0 && !live1) {  // the last element of an odd-length signal (one lane of the last tile)
        zr[k] = admm_double2{a.z[i0], 0.0};
        if (!VIN) ur[k] = admm_double2{a.u[i0], 0.0};
        sr[k] = admm_double2{a.s[i0], 0.0};
      }
      const admm_double2 zz = zr[k], uu = ur[k], ss = sr[k];
      // z - u;  VIN: (v - c) - c with c = clamp(v), the two roundings of z = v - c and z - u
      const double t0 = VIN ? (zz.x - tv_clamp(zz.x, th)) - tv_clamp(zz.x, th) : zz.x - uu.x;
      const double t1 = VIN ? (zz.y - tv_clamp(zz.y, th)) - tv_clamp(zz.y, th) : zz.y - uu.y;
      double tm = __shfl_up(t1, 1, 64);  // element i0-1 is the previous lane's second element
      if (lane == 0) tm = VIN ? (zb[k] - tv_clamp(zb[k], th)) - tv_clamp(zb[k], th) : zb[k] - ub[k];
      const double b0 = ss.x + rho * ((i0 > 0) ? t0 - tm : t0);  // getProxOps.m:1047
      const double b1 = ss.y + rho * (t1 - t0);
      if (live0) Bq[direct ? pos8(2 * j) : pos16(2 * j)] = b0;
      if (live1) Bq[direct ? pos8(2 * j + 1) : pos16(2 * j + 1)] = b1;
    }
  }
  __syncthreads();

  if (direct) {
    // ---- 2a. x = A * (causal + anticausal exponential sums) for this thread's 8 consecutive positions
    const int p0 = tid * E;
    if (p0 >= M - E && p0 + E <= count - M + E) {  // covers every x the update below needs: [M - 1, count - M + 1)
      // Taps in whole groups of 8 (G*8 >= K of them; the margin M = 8*G + 8 holds them): the padded position of
      // element p0 + 8*g + j is 9*(tid + g) + j, so a group is eight reads at constant offsets from one address.
      static_assert(E == 8, "the grouped tap loops assume one pad slot per 8 positions");
      const int G = (K + 7) >> 3;
      const double* __restrict__ own = Bq + 9 * tid;
      double c = 0.0;  // sum_{k=1..8G} r^(k-1) b(p0 - k)
      {
        const double* __restrict__ g = own - 9 * G;
#pragma unroll 2
        for (int q = 0; q < G; ++q, g += 9) {
#pragma unroll
          for (int j = 0; j < 8; ++j) c = __builtin_fma(r, c, g[j]);
        }
      }
      double ac = 0.0;  // sum_{k=1..8G} r^k b(p0 + 7 + k)
      {
        const double* __restrict__ g = own + 9 * G;
#pragma unroll 2
        for (int q = 0; q < G; ++q, g -= 9) {
#pragma unroll
          for (int j = 7; j >= 0; --j) ac = __builtin_fma(r, ac, g[j]);
        }
      }
      ac *= r;
      double bv[E], cv[E], av[E];
#pragma unroll
      for (int j = 0; j < E; ++j) bv[j] = own[j];
      cv[0] = __builtin_fma(r, c, bv[0]);
#pragma unroll
      for (int j = 1; j < E; ++j) cv[j] = __builtin_fma(r, cv[j - 1], bv[j]);
      av[E - 1] = ac;
#pragma unroll
      for (int j = E - 1; j >= 1; --j) av[j - 1] = r * (av[j] + bv[j]);
      const double A = a.green;
#pragma unroll
      for (int j = 0; j < E; ++j) Xq[9 * tid + j] = A * (cv[j] + av[j]);
    }
    __syncthreads();
  } else {
    // ---- 2b. a tile at an end of the matrix: the exact recurrences as two block scans (forward in place, then
    // y/b in backward order into Xq, backward in place)
    const double cstar = r;
    tv_block_scan<E>(Bq, count, [&](int q) { return tv_coef<false>(a, w0 + q, n, rho, cstar); }, wA, wB);
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int q = tid + k * kBlock;  // forward position
      if (q < count) {
        const int64_t i = w0 + q;
        const int qq = count - 1 - q;
        Xq[pos16(qq)] = Bq[pos16(q)] * ((i < a.nprefix) ? 1.0 / a.bprefix[i] : ibstar);
      }
    }
    __syncthreads();
    tv_block_scan<E>(Xq, count, [&](int q) { return tv_coef<true>(a, w1 - 1 - q, n, rho, cstar); }, wA, wB);
  }
  auto xat = [&](int64_t i) -> double {
    const int q = static_cast<int>(i - w0);
    return direct ? Xq[pos8(q)] : Xq[pos16(count - 1 - q)];
  };

  // ---- 3. z/u update (pairs, ascending), owned outputs, residual sums
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
#pragma unroll
  for (int k = 0; k < E / 2; ++k) {
    const int j = tid + k * kBlock;
    const int64_t i0 = w0 + 2 * static_cast<int64_t>(j);
    const bool live0 = 2 * j < count, live1 = 2 * j + 1 < count;
    double zn0 = 0.0, zn1 = 0.0, un0 = 0.0, un1 = 0.0, dz0 = 0.0, dz1 = 0.0;
    double x0 = 0.0, x1 = 0.0, x2 = 0.0, ax0 = 0.0, ax1 = 0.0;
    double zp0 = zr[k].x, zp1 = zr[k].y, uo0 = ur[k].x, uo1 = ur[k].y;
    if (VIN) {  // z, u again from v -- behind a barrier for the optimiser: carried over from step 1 they would cost
                // 16 registers across the solve (and spill); three instructions per element here
      asm volatile("" : "+v"(zp0), "+v"(zp1));
      uo0 = tv_clamp(zp0, th);
      uo1 = tv_clamp(zp1, th);
      zp0 -= uo0;
      zp1 -= uo1;
    }
    double vn0 = 0.0, vn1 = 0.0;
    double s0 = 0.0, s1 = 0.0;
    if (a.objevals && live0) {  // totalvariation.m:134: 1/2*||x - s||^2
      s0 = a.s[i0];
      if (live1) s1 = a.s[i0 + 1];
    }
    if (live0) {
      x0 = xat(i0);
      if (live1) x1 = xat(i0 + 1);
      if (i0 + 2 < w1) x2 = xat(i0 + 2);
      ax0 = (i0 + 1 < n) ? x0 - x1 : x0;   // D*x, last row is x_n (totalvariation.m:127)
      vn0 = uo0 + ax0;
      un0 = tv_clamp(vn0, th);  // admm.m:548 (c = 0): u + (D x - z) = v+ - z
      zn0 = vn0 - un0;          // getProxOps.m:199: soft(u + D x, lambda/rho)
      dz0 = zn0 - zp0;
      if (live1) {
        ax1 = (i0 + 2 < n) ? x1 - x2 : x1;
        vn1 = uo1 + ax1;
        un1 = tv_clamp(vn1, th);
        zn1 = vn1 - un1;
        dz1 = zn1 - zp1;
      }
    }
    // the left neighbour (element i0-1) is the previous lane's second element
    double unm = __shfl_up(un1, 1, 64), dzm = __shfl_up(dz1, 1, 64);
    if (lane == 0 && live0 && i0 > w0) {  // wave boundary: recompute element i0-1 (its old z, u: loaded in step 1)
      const double xm = xat(i0 - 1);
      double zpm = zb[k], uom = ub[k];
      if (VIN) {
        asm volatile("" : "+v"(zpm));
        uom = tv_clamp(zpm, th);
        zpm -= uom;
      }
      const double vnm = uom + (xm - x0);
      unm = tv_clamp(vnm, th);
      dzm = (vnm - unm) - zpm;
    }
    const bool left = i0 > 0;  // i0 == w0 > 0 is never an owned position (window margin)
    if (live0) {
      const bool own0 = i0 >= o0 && i0 < o1, own1 = live1 && i0 + 1 >= o0 && i0 + 1 < o1;
      if (own0) {
        const double rr = ax0 + (-zn0), du = un0 - uo0;
        const double g2 = left ? dz0 - dzm : dz0, g3 = left ? un0 - unm : un0;
        acc[S_R2] += rr * rr;
        acc[S_AX2] += ax0 * ax0;
        acc[S_Z2] += zn0 * zn0;
        acc[S_DZ2] += dz0 * dz0;
        acc[S_U2] += un0 * un0;
        acc[S_DU2] += du * du;
        acc[S_G2] += g2 * g2;
        acc[S_G3] += g3 * g3;
        if (a.objevals) {  // totalvariation.m:134-135
          if (i0 + 1 < n) acc[S_OBJZ] += fabs(x1 - x0);
          const double e = x0 - s0;
          acc[S_OBJX] += e * e;
        }
      }
      if (own1) {
        const double rr = ax1 + (-zn1), du = un1 - uo1;
        const double g2 = dz1 - dz0, g3 = un1 - un0;
        acc[S_R2] += rr * rr;
        acc[S_AX2] += ax1 * ax1;
        acc[S_Z2] += zn1 * zn1;
        acc[S_DZ2] += dz1 * dz1;
        acc[S_U2] += un1 * un1;
        acc[S_DU2] += du * du;
        acc[S_G2] += g2 * g2;
        acc[S_G3] += g3 * g3;
        if (a.objevals) {
          if (i0 + 2 < n) acc[S_OBJZ] += fabs(x2 - x1);
          const double e = x1 - s1;
          acc[S_OBJX] += e * e;
        }
      }
      if (own0 && own1) {
        store2<NTS>(a.zo + i0, admm_double2{vn0, vn1});  // the compact state v+ = z+ + u+
      } else {
        if (own0) a.zo[i0] = vn0;
        if (own1) a.zo[i0 + 1] = vn1;
      }
      if (a.xhist) {  // history columns start at it*n (odd-aligned for odd n): scalar stores
        if (own0) {
          a.xhist[it * n + i0] = x0;
          a.zhist[it * n + i0] = zn0;
          a.uhist[it * n + i0] = un0;
        }
        if (own1) {
          a.xhist[it * n + i0 + 1] = x1;
          a.zhist[it * n + i0 + 1] = zn1;
          a.uhist[it * n + i0 + 1] = un1;
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) {
    const double w = wave_sum(acc[s]);
    if (lane == 0) sred[wid][s] = w;
  }
  __syncthreads();
  if (tid < S_COUNT)
    a.part[tid * a.part_stride + tile_id] = ((sred[0][tid] + sred[1][tid]) + sred[2][tid]) + sred[3][tid];
}

// ---------------------------------------------------------------- the same iteration, thread-owned 8-position runs
// tv_direct2.h: hierarchical exponential sums (2G + 4 LDS doubles per thread instead of ~130), 26 KB of LDS per tile,
// 73 VGPRs: six tiles per CU, 72 us per iteration at n = 4096^2 where tv_direct_kernel takes 113 (dev/tv1d_bench.hip).
// Both ends of the signal by mirror images -- no scan path.  Grid: [passenger] + tiles, the tiles that reach past an end
// of the signal (element-wise loads: slow) dispatched first.  EXTRA = objective and / or history columns compiled in.
template <bool NTS, bool VIN, bool EXTRA>
__global__ __launch_bounds__(kBlock, (EXTRA || !VIN) ? 4 : 6) void tv_direct2_kernel(TvArgs a, FinArgs fin,
                                                                            const Ctrl* __restrict__ ctrl) {
  // every kernel argument the tile path reads, requested together with the control block's address (see tv_direct_kernel)
  asm volatile("" ::"s"(a.n), "s"(a.z), "s"(a.u), "s"(a.s), "s"(a.zo), "s"(a.part), "s"(a.ftile), "s"(a.margin),
               "s"(a.deferred), "s"(a.part_stride), "s"(a.thresh), "s"(a.rho), "s"(a.green), "s"(a.rpow[0]),
               "s"(a.rpow[7]), "s"(a.objevals), "s"(a.xhist), "s"(ctrl));
  // The stop flag is asked for here and tested behind the tile's loads (nothing is stored before that): in front of them
  // the test costs every tile one more memory round trip.  A scalar load issued and waited for by hand; the register
  // holds garbage between the two statements, and nothing may read it there -- tests/test_abi_and_host.py checks the
  // disassembly of the shipped library for exactly that, tests/test_gpu_ops.py runs the kernel with the flag set.
  int32_t stop;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(stop) : "s"(ctrl));
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  if (a.deferred && blockIdx.x == 0) {  // the passenger (dispatched first): tail of the PREVIOUS iteration
    if (ctrl->stop || !a.fin_pending) return;  // (its own read: `stop` must not be needed before the tiles' loads)
    {
      const int slot = tid >> 4, sub = tid & 15;
      double v = 0.0;
      if (slot < S_COUNT) {
        const double* __restrict__ ps = a.prev_part + slot * a.part_stride;
        for (int32_t b0 = 0; b0 < a.prev_ntiles; b0 += 256) {
          double w[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int32_t b = b0 + sub + 16 * k;
            w[k] = ps[b < a.prev_ntiles ? b : a.prev_ntiles - 1];
          }
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (b0 + sub + 16 * k < a.prev_ntiles) v += w[k];
        }
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (slot < S_COUNT && sub == 0) a.slots16[slot] = v;
    }
    __threadfence_block();
    __syncthreads();
    finalize_body<false>(fin);  // fin.slots_reduced = a.slots16
    return;
  }
  const unsigned bid = a.deferred ? blockIdx.x - 1u : blockIdx.x;
  const unsigned ntiles = a.deferred ? gridDim.x - 1u : gridDim.x;
  // the last two tiles, then tile 0, then the rest in order
  const unsigned tile_id = bid < 2u && ntiles > 2u ? ntiles - 1u - bid : (ntiles > 2u ? bid - 2u : bid);
  const int64_t it = a.deferred ? a.iter_host : ctrl->iter;
  tv2_tile<4, VIN, NTS, EXTRA>(a, tile_id, it, lds, [&]() -> bool {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(stop));
    return stop != 0;  // (uniform)
  });
}

int tv_direct_margin(const TvArgs& a) { return static_cast<int>(round_up(static_cast<int64_t>(a.halo) + 8, 8)); }
bool tv_direct_ok(const TvArgs& a) { return a.elems == 8 && a.halo >= 2 && tv_direct_margin(a) <= 256 && a.n >= 2; }

// a.ftile = 2048 - 2*a.margin, a.green = 1/(b*(1 - r^2)); grid = tiles (+ 1 passenger when a.deferred)
void launch_tv_direct(const TvArgs& a, const FinArgs& fin, const Ctrl* ctrl, hipStream_t stream) {
  const int64_t ntiles = ceil_div(a.n, a.ftile);
  // thread-owned runs (tv_direct2.h) unless the signal is shorter than two margins (the mirror images of an end would
  // reach past the other end) or ADMM_HIP_TV_DIRECT1 asks for the first form
  const bool first_form = std::getenv("ADMM_HIP_TV_DIRECT1") != nullptr;
  if (!first_form && a.n >= 2 * static_cast<int64_t>(a.margin)) {
    const size_t lds2 = sizeof(double) * tv2_lds_doubles<4>();
    const dim3 grid2(static_cast<unsigned>(ntiles) + (a.deferred ? 1u : 0u)), block2(kBlock);
    const bool nts2 = stream_hint(8 * 8 * a.n), extra = a.objevals || a.xhist;
#define ADMM_TV2(NTS_, VIN_, EXTRA_) \
  hipLaunchKernelGGL((tv_direct2_kernel<NTS_, VIN_, EXTRA_>), grid2, block2, lds2, stream, a, fin, ctrl)
    if (a.state_in) {
      if (nts2) { if (extra) ADMM_TV2(true, true, true); else ADMM_TV2(true, true, false); }
      else { if (extra) ADMM_TV2(false, true, true); else ADMM_TV2(false, true, false); }
    } else {
      if (nts2) { if (extra) ADMM_TV2(true, false, true); else ADMM_TV2(true, false, false); }
      else { if (extra) ADMM_TV2(false, false, true); else ADMM_TV2(false, false, false); }
    }
#undef ADMM_TV2
    return;
  }
  constexpr int kCap = kTvDirectE * kBlock;
  const size_t lds = 2 * static_cast<size_t>(kCap + (kCap >> (kTvDirectE == 8 ? 3 : 2)) + 1) * sizeof(double);
  const dim3 grid(static_cast<unsigned>(ntiles) + (a.deferred ? 1u : 0u)), block(kBlock);
  const bool nts = stream_hint(8 * 8 * a.n);
  if (a.state_in) {
    if (nts) hipLaunchKernelGGL((tv_direct_kernel<true, true>), grid, block, lds, stream, a, fin, ctrl);
    else hipLaunchKernelGGL((tv_direct_kernel<false, true>), grid, block, lds, stream, a, fin, ctrl);
  } else {
    if (nts) hipLaunchKernelGGL((tv_direct_kernel<true, false>), grid, block, lds, stream, a, fin, ctrl);
    else hipLaunchKernelGGL((tv_direct_kernel<false, false>), grid, block, lds, stream, a, fin, ctrl);
  }
}

// out16[s] = sum over the tiles of part[s][.], one workgroup per slot, fixed order
__global__ __launch_bounds__(kBlock) void tv_pack_kernel(const double* __restrict__ part, int64_t stride, int32_t nblk,
                                                         double* __restrict__ out16, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ double scratch[4];
  const int s = blockIdx.x;
  double v = 0.0;
  for (int32_t b = threadIdx.x; b < nblk; b += kBlock) v += part[s * stride + b];
  const double t = block_sum(v, scratch);
  if (threadIdx.x == 0) out16[s] = t;
}

__global__ __launch_bounds__(kBlock) void tv_dx_kernel(const double* __restrict__ x, const double* __restrict__ s,
                                                       int64_t n, double lambda, int objevals,
                                                       double* __restrict__ ax, double* __restrict__ objpart,
                                                       const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ double scratch[4];
  double acc = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double xi = x[i];
    const bool inner = i + 1 < n;
    const double xn = inner ? x[i + 1] : 0.0;
    ax[i] = inner ? xi - xn : xi;  // D = spdiags([1 -1], 0:1, n, n): the last row is x_n (totalvariation.m:127)
    if (objevals) {                // totalvariation.m:134-135
      const double e = xi - s[i];
      acc += 0.5 * e * e + (inner ? lambda * fabs(xn - xi) : 0.0);
    }
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) objpart[blockIdx.x] = t;
}

__global__ __launch_bounds__(kBlock) void tv_dual_kernel(const double* __restrict__ dz, const double* __restrict__ u,
                                                         int64_t n, double* __restrict__ part,
                                                         const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ double scratch[4];
  double a2 = 0.0, a3 = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double g2 = (i > 0) ? dz[i] - dz[i - 1] : dz[i];  // (D'w)_i = w_i - w_{i-1}
    const double g3 = (i > 0) ? u[i] - u[i - 1] : u[i];
    a2 += g2 * g2;
    a3 += g3 * g3;
  }
  const double t2 = block_sum(a2, scratch);
  const double t3 = block_sum(a3, scratch);
  if (threadIdx.x == 0) {
    part[S_G2 * kMaxPartBlocks + blockIdx.x] = t2;
    part[S_G3 * kMaxPartBlocks + blockIdx.x] = t3;
  }
}

void launch_tv_dx(const double* x, const double* s, int64_t n, double lambda, int objevals, double* ax,
                  double* objpart, int* nobj_out, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(n, kBlock);
  if (blocks > kMaxPartBlocks) blocks = kMaxPartBlocks;
  if (blocks < 1) blocks = 1;
  *nobj_out = static_cast<int>(blocks);
  hipLaunchKernelGGL(tv_dx_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, x, s, n, lambda,
                     objevals, ax, objpart, ctrl);
}

// Over-relaxation with the reference's total-variation closures (admm.m:515-532 hands Axhat to zming in place of x,
// and getProxOps.m:199 applies D to whatever it is given): z = soft(u + D*Axhat, t),
// Axhat = relax*ax - (1-relax)*(-zprev) with ax = D*x -- D is applied twice, reproduced as is.
__global__ __launch_bounds__(kBlock) void tv_relax_z_kernel(const double* __restrict__ ax, const double* __restrict__ zp,
                                                            const double* __restrict__ u, int64_t n, double relax,
                                                            double t, double* __restrict__ zgiven,
                                                            const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    // the expression of prox_element / prez_kernel (c = 0), so that both see the same Axhat bit for bit
    const double h0 = relax * ax[i] - (1.0 - relax) * ((-zp[i]) - 0.0);
    double w = h0;
    if (i + 1 < n) w = h0 - (relax * ax[i + 1] - (1.0 - relax) * ((-zp[i + 1]) - 0.0));
    zgiven[i] = tv_soft(u[i] + w, t);
  }
}

void launch_tv_relax_z(const double* ax, const double* zp, const double* u, int64_t n, double relax, double t,
                       double* zgiven, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(n, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(tv_relax_z_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, ax, zp, u, n,
                     relax, t, zgiven, ctrl);
}

void launch_tv_dual(const double* dz, const double* u, int64_t n, double* part, int nblk, const Ctrl* ctrl,
                    hipStream_t stream) {
  hipLaunchKernelGGL(tv_dual_kernel, dim3(static_cast<unsigned>(nblk)), dim3(kBlock), 0, stream, dz, u, n, part, ctrl);
}

bool tv_fused_ok(const TvArgs& a) { return a.elems == 8 && a.ftile >= 512; }

void launch_tv_fused(const TvArgs& a, const FinArgs& fin, double* slots16, const Ctrl* ctrl, hipStream_t stream) {
  const int64_t ntiles = ceil_div(a.n, a.ftile);
  constexpr int kCap = 8 * kBlock;
  const size_t lds = 2 * static_cast<size_t>(kCap + (kCap >> 4) + 1) * sizeof(double);
  // streaming stores once the eight vectors of an iteration cannot stay cache-resident: +7 % at n = 4096^2
  const bool nts = stream_hint(8 * 8 * a.n);
  FinArgs f = fin;
  if (a.gcount) {  // one-launch iteration: the finalize logic reads the group partials
    f.part = a.gpart;
    f.nblk = a.ngroups;
    f.slots_reduced = nullptr;
  }
  int mode = nts ? 1 : 0;
  if (nts)
    if (const char* env = getenv("ADMM_HIP_TV_CACHE")) mode = atoi(env);
  const dim3 grid(static_cast<unsigned>(ntiles) + (a.deferred ? 1u : 0u)), block(kBlock);
  switch (mode) {
    case 0: hipLaunchKernelGGL((tv_fused_kernel<8, 0>), grid, block, lds, stream, a, f, ctrl); break;
    case 2: hipLaunchKernelGGL((tv_fused_kernel<8, 2>), grid, block, lds, stream, a, f, ctrl); break;
    case 3: hipLaunchKernelGGL((tv_fused_kernel<8, 3>), grid, block, lds, stream, a, f, ctrl); break;
    case 4: hipLaunchKernelGGL((tv_fused_kernel<8, 4>), grid, block, lds, stream, a, f, ctrl); break;
    default: hipLaunchKernelGGL((tv_fused_kernel<8, 1>), grid, block, lds, stream, a, f, ctrl); break;
  }
  if (a.gcount || a.deferred) return;
  hipLaunchKernelGGL(tv_pack_kernel, dim3(S_COUNT), dim3(kBlock), 0, stream, a.part, a.part_stride,
                     static_cast<int32_t>(ntiles), slots16, ctrl);
}

void launch_tv_pack(const double* part, int64_t stride, int32_t ntiles, double* slots16, const Ctrl* ctrl,
                    hipStream_t stream) {
  hipLaunchKernelGGL(tv_pack_kernel, dim3(S_COUNT), dim3(kBlock), 0, stream, part, stride, ntiles, slots16, ctrl);
}

// ---- host side -------------------------------------------------------------------------
int tv_plan(double rho, int64_t n, std::vector<double>* prefix, double* bstar, int* halo, int* elems, int* tile) {
  // LDL' pivots of I + rho*D'D:  b_1 = 1+rho, b_i = 1+2rho - rho^2/b_{i-1}
  std::vector<double> b;
  b.reserve(4096);
  double cur = 1.0 + rho;
  b.push_back(cur);
  const int64_t cap = n < (1 << 20) ? n : (1 << 20);
  int64_t stationary = -1;
  for (int64_t i = 1; i < cap; ++i) {
    const double nxt = (1.0 + 2.0 * rho) - rho * rho / cur;
    if (nxt == cur) {
      stationary = i;
      break;
    }
    b.push_back(nxt);
    cur = nxt;
  }
  if (stationary < 0 && cap < n)
    return fail(ADMM_E_UNSUPPORTED, "total variation: tridiagonal pivots did not become stationary (rho too large)");
  *bstar = cur;
  *prefix = b;
  // Halo length: a window that does not reach row 0 must damp an arbitrary incoming carry below
  // 1e-18.  The multipliers rho/b_i decrease monotonically towards rho/b*, so the weakest damping
  // is the window that starts right after row 0: multiply the actual multipliers from there.
  // (Both sweeps use the same multipliers, the backward one indexed from the other end where they
  // are stationary, so this bound covers it.)
  const double astar = rho / cur;
  if (!(astar < 1.0) || !(rho / b[0] < 1.0))
    return fail(ADMM_E_NUMERIC, "total variation: matrix is not diagonally dominant");
  double prod = 1.0;
  int H = 0;
  while (prod > 1e-18 && H < (1 << 20)) {
    const size_t idx = static_cast<size_t>(H);
    prod *= (idx < b.size()) ? rho / b[idx] : astar;
    ++H;
  }
  if (H < 2) H = 2;
  if (H & 1) H += 1;  // even: the sweeps walk the processing range in aligned pairs
  // a window never needs to be longer than the signal: clipped at both ends it starts and ends exactly (no incoming
  // carry to damp).  Without this a short signal with a large rho (n = 3, rho = 430: 5900 positions to damp a carry by
  // 1e-18) was refused below although one workgroup holds all of it.
  {
    const int64_t whole = (n + 1) & ~int64_t{1};
    if (static_cast<int64_t>(H) > whole) H = static_cast<int>(whole < 2 ? 2 : whole);
  }
  // workgroup capacity = 256*elems positions = owned tile + halo.  Small tiles keep the LDS
  // footprint low (17 KiB at elems = 8 -> 8 workgroups per CU), which is what hides the
  // load -> scan -> store phase structure of a workgroup behind its neighbours.
  if (H <= 256) {
    *elems = 8;
    *tile = 256 * 8 - H;
  } else if (H <= 1024) {
    *elems = 20;
    *tile = 256 * 20 - H;
  } else if (H <= 8192) {  // (beyond 4096 the owned part of a workgroup's 12288 positions shrinks to a third: slow, not wrong)
    *elems = 48;
    *tile = 256 * 48 - H;
  } else {
    return fail(ADMM_E_UNSUPPORTED, "total variation: rho too large for the windowed tridiagonal solve");
  }
  *halo = H;
  return ADMM_OK;
}

void launch_tv_sweep(const TvArgs& a, bool backward, const Ctrl* ctrl, hipStream_t stream) {
  const unsigned blocks = static_cast<unsigned>(ceil_div(a.n, a.tile));
  const int cap = a.elems * kBlock;
  const size_t lds = static_cast<size_t>(cap + (cap >> 4) + 1) * sizeof(double);
  if (a.elems == 8) {
    if (backward) hipLaunchKernelGGL((tv_sweep_kernel<8, true>), dim3(blocks), dim3(kBlock), lds, stream, a, ctrl);
    else hipLaunchKernelGGL((tv_sweep_kernel<8, false>), dim3(blocks), dim3(kBlock), lds, stream, a, ctrl);
  } else if (a.elems == 20) {
    if (backward) hipLaunchKernelGGL((tv_sweep_kernel<20, true>), dim3(blocks), dim3(kBlock), lds, stream, a, ctrl);
    else hipLaunchKernelGGL((tv_sweep_kernel<20, false>), dim3(blocks), dim3(kBlock), lds, stream, a, ctrl);
  } else {
    if (backward) hipLaunchKernelGGL((tv_sweep_kernel<48, true>), dim3(blocks), dim3(kBlock), lds, stream, a, ctrl);
    else hipLaunchKernelGGL((tv_sweep_kernel<48, false>), dim3(blocks), dim3(kBlock), lds, stream, a, ctrl);
  }
}

void launch_tv_prox(const TvArgs& a, const Ctrl* ctrl, int* nblk_out, hipStream_t stream) {
  int64_t blocks = ceil_div(a.n, kBlock);
  if (blocks > kMaxPartBlocks) blocks = kMaxPartBlocks;
  if (blocks < 1) blocks = 1;
  *nblk_out = static_cast<int>(blocks);
  hipLaunchKernelGGL(tv_prox_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a, ctrl);
}

}  // namespace admm
