// tv_direct2.h -- the 1-D total-variation iteration with a thread owning 8 CONSECUTIVE positions from the loads to the
// stores (device code shared by tv.hip and dev/tv1d_bench.hip).
//
// One iteration of totalvariation.m's loop on the compact dual state v = z + u (tv.hip: tv_direct_kernel):
//   x  = (I + rho*D'D) \ (s + rho*D'(z - u))                getProxOps.m:1044-1048
//   v+ = u + D*x,  u+ = clamp(v+, +-lambda/rho),  z+ = v+ - u+   getProxOps.m:199, admm.m:548
// Away from the ends the matrix is Toeplitz, tridiag(-rho, 1+2rho, -rho), and its inverse the two-sided exponential
// kernel x_i = A * sum_k r^|k| b_(i+k), r = rho/b*, A = 1/(b*(1-r^2)).  tv_direct_kernel evaluates the two one-sided
// sums of a thread's 8 positions as Horner sums over 2*8G taps read from LDS (~112 LDS reads per thread, b and x
// both staged through 37 KB of LDS: 4 tiles per CU, latency-bound at 0.44 of the HBM roofline).  Here the sums are
// hierarchical: with b_0..b_7 in registers a thread forms its own causal / anticausal sums and the two end values
//   S = sum_j r^(7-j) b_j            (what it contributes to everything on its right)
//   T = r * sum_j r^j b_j            (... on its left)
// publishes S and T (two LDS doubles), and receives the carries of its 8-position block as G-term Horner sums in
// r^8 over its G left / right neighbour threads:
//   c_in = sum_g r^(8(g-1)) S[t-g],   a_in = sum_g r^(8(g-1)) T[t+g],   r^(8G) < 1e-18
//   x_j = A * ( (local causal)_j + r^(j+1) c_in  +  (local anticausal)_j + r^(7-j) a_in ).
// 2G + 4 LDS doubles per thread instead of ~130, 4.5 KB of LDS per tile instead of 37 KB, x never goes through LDS
// (only the first and the last x of a thread, for the neighbours' D*x and dual-residual stencils), the residual sums
// leave the waves through DPP / permlane swaps instead of 144 ds_bpermute per thread.
//
// The two ends of the signal need no special solver: by the method of images the first row (diagonal 1 + rho: x_0 =
// x_1 ghost, even symmetry about -1/2) and the last row (x_n = 0 ghost, odd symmetry about n) are the infinite
// Toeplitz system on the mirrored right-hand side, b_(-1-k) = b_k and b_n = 0, b_(n+k) = -b_(n-k): a tile whose
// window reaches past an end fills its ghost positions with mirrored b (element-wise loads; two or three tiles).
#pragma once
#include "tv.h"

namespace admm {

constexpr int kTv2E = 8;  // positions per thread

__device__ __forceinline__ double tv2_clamp(double v, double t) { return __builtin_fmin(__builtin_fmax(v, -t), t); }

template <int CTRL>
__device__ __forceinline__ double tv2_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// a.upper32 <-> b.lower32 (v_permlane32_swap): afterwards a + b holds the 2-way sum of the old a in the lower 32 lanes
// and of the old b in the upper 32
__device__ __forceinline__ void tv2_swap_half(double& a, double& b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  a = __hiloint2double(r1[0], r0[0]);
  b = __hiloint2double(r1[1], r0[1]);
}
// odd 16-lane rows of a <-> even rows of b (v_permlane16_swap)
__device__ __forceinline__ void tv2_swap_row(double& a, double& b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  a = __hiloint2double(r1[0], r0[0]);
  b = __hiloint2double(r1[1], r0[1]);
}
// Wave-wide sums of 4 per-lane values: every lane of 16-lane row q returns the total of t[kOrder[q]] with
// kOrder = {0, 2, 1, 3} (no LDS traffic: permlane swaps + DPP)
__device__ __forceinline__ double tv2_reduce4(double t0, double t1, double t2, double t3) {
  tv2_swap_half(t0, t1);
  tv2_swap_half(t2, t3);
  t0 += t1;  // lower 32 lanes: t0, upper: t1
  t2 += t3;  // lower: t2, upper: t3
  tv2_swap_row(t0, t2);
  double r = t0 + t2;  // rows: t0 | t2 | t1 | t3
  r += tv2_dpp<0x128>(r);  // row_ror:8
  r += tv2_dpp<0x124>(r);  // row_ror:4
  r += tv2_dpp<0x4E>(r);   // quad_perm [2,3,0,1]
  r += tv2_dpp<0xB1>(r);   // quad_perm [1,0,3,2]
  return r;
}

// LDS doubles a workgroup of NW waves needs: S, T, first x, last x of every thread (the waves' partial sums reuse S once
// the carries are in); one staging buffer of 512 positions (+ 1 pad per 8) per wave.  26 KB at NW = 4: six tiles per CU.
template <int NW>
constexpr int tv2_lds_doubles() {
  return 4 * NW * 64 + NW * 576;
}

// One tile.  Global accesses are coalesced 16-byte pairs; a wave transposes its 512 positions through its own staging
// buffer into 8-position runs per lane (DS operations of one wave execute in order: no barrier), and the old v stays
// parked there -- lane l's slots 9l .. 9l+7 -- until the update reads it back and writes v+ over it for the transposed
// store.  VIN: a.z holds v = z + u (a run's first iteration reads z and u as given: VIN = false; u then stays in
// registers).  EXTRA: the objective (a.objevals) and the history columns (a.xhist) are compiled in -- a separate
// instantiation, because their live values would otherwise set the register allocation of the plain iteration.
// stop_after_loads(): called once every load of the tile has been issued and before anything is stored; a
// true result abandons the tile (uniform).
template <int NW, bool VIN, bool NTS, bool EXTRA, class StopFn>
__device__ __forceinline__ void tv2_tile(const TvArgs& a, unsigned tile_id, int64_t it, double* __restrict__ lds,
                                         StopFn stop_after_loads) {
  constexpr int E = kTv2E, NT = NW * 64, WIN = NT * E;
  double* __restrict__ Sq = lds;            // [NT] S of every thread
  double* __restrict__ Tq = lds + NT;       // [NT] T
  double* __restrict__ X0q = lds + 2 * NT;  // [NT] first x of every thread
  double* __restrict__ X7q = lds + 3 * NT;  // [NT] last x
  double* __restrict__ sred = lds;          // [NW][S_COUNT], over S: dead behind the second barrier
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double* __restrict__ stage = lds + 4 * NT + wid * 576;
  double* __restrict__ mine = stage + 9 * lane;  // this lane's run
  const int64_t n = a.n;
  const double rho = a.rho, th = a.thresh;
  const double r = a.rpow[0];
  const int M = a.margin, G = (M >> 3) - 1;
  const int64_t o0 = static_cast<int64_t>(tile_id) * a.ftile;  // a.ftile = WIN - 2*M
  const int64_t o1 = (o0 + a.ftile < n) ? o0 + a.ftile : n;
  const int64_t w0 = o0 - M;                                   // negative in the first tile (ghost positions)
  const int64_t p0 = w0 + static_cast<int64_t>(tid) * E;       // this thread's first position
  const int64_t c0 = w0 + 512 * wid;                           // the wave's first position
  const bool edge = (w0 < 0) || (w0 + WIN > n);                // uniform: the window reaches past an end

  // ---- 1. loads; b = s + rho*D'(z - u) in registers, old v parked in `mine`
  double ua[E], b[E];         // ua: u (only !VIN)
  double vl = 0.0, ul = 0.0;  // v (or z) and u of position p0 - 1
#pragma unroll
  for (int j = 0; j < E; ++j) ua[j] = 0.0;
  if (!edge) {
    admm_double2 qv[4], qu[4], qs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      qv[k] = load2<true>(a.z + c0 + 128 * k + 2 * lane);
      if (!VIN) qu[k] = load2<true>(a.u + c0 + 128 * k + 2 * lane);
      qs[k] = load2<false>(a.s + c0 + 128 * k + 2 * lane);
    }
    if (lane == 0) {  // p0 - 1 >= 0: an interior window starts at o0 - M > 0
      vl = a.z[p0 - 1];
      if (!VIN) ul = a.u[p0 - 1];
    }
    if (stop_after_loads()) return;
    auto put = [&](const admm_double2 (&q)[4]) {  // position q of the wave at q + (q >> 3)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pq = 128 * k + 2 * lane, ix = pq + (pq >> 3);
        stage[ix] = q[k].x;
        stage[ix + 1] = q[k].y;
      }
      __builtin_amdgcn_wave_barrier();
    };
    auto get = [&](double (&d)[E]) {
#pragma unroll
      for (int j = 0; j < E; ++j) d[j] = mine[j];
      __builtin_amdgcn_wave_barrier();
    };
    double sv[E], va[E];
    put(qs);
    get(sv);
    if (!VIN) {
      put(qu);
      get(ua);
    }
    put(qv);
    get(va);  // ... and v stays in the buffer
    // z - u;  VIN: (v - c) - c with c = clamp(v): the two roundings of z = v - c and z - u
    double t[E];
#pragma unroll
    for (int j = 0; j < E; ++j)
      t[j] = VIN ? (va[j] - tv2_clamp(va[j], th)) - tv2_clamp(va[j], th) : va[j] - ua[j];
    const double vlw = tv2_dpp<0x138>(va[E - 1]);  // wave_shr:1 -- position p0 - 1 is the previous lane's last
    const double ulw = VIN ? 0.0 : tv2_dpp<0x138>(ua[E - 1]);
    if (lane != 0) {
      vl = vlw;
      ul = ulw;
    }
    const double tl = VIN ? (vl - tv2_clamp(vl, th)) - tv2_clamp(vl, th) : vl - ul;
    b[0] = sv[0] + rho * (t[0] - tl);  // getProxOps.m:1047 (p0 > 0 here)
#pragma unroll
    for (int j = 1; j < E; ++j) b[j] = sv[j] + rho * (t[j] - t[j - 1]);
  } else {
    // a window that reaches past an end of the signal (two or three tiles of a launch): one element at a time, ghost
    // positions from their mirror images.  A rolled loop on purpose -- unrolled, its 8 x 7 live values would set the
    // register allocation of the whole kernel; b (and u) are filled by shifting.
    if (stop_after_loads()) return;
    {
      const int64_t ql = p0 - 1 < 0 ? 0 : (p0 - 1 > n - 1 ? n - 1 : p0 - 1);
      vl = a.z[ql];
      if (!VIN) ul = a.u[ql];
    }
#pragma unroll
    for (int j = 0; j < E; ++j) b[j] = 0.0;
#pragma unroll 1
    for (int j = 0; j < E; ++j) {
      const int64_t i = p0 + j;
      int64_t q = i;
      double sgn = 1.0;
      if (i < 0) {
        q = -1 - i;
      } else if (i == n) {
        q = n - 1;
        sgn = 0.0;
      } else if (i > n) {
        q = 2 * n - i;
        sgn = -1.0;
      }
      q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
      const int64_t qm = q > 0 ? q - 1 : 0;
      const double zc = a.z[q], zm = a.z[qm], sc = a.s[q];
      const double uc = VIN ? 0.0 : a.u[q], um = VIN ? 0.0 : a.u[qm];
      const double tc = VIN ? (zc - tv2_clamp(zc, th)) - tv2_clamp(zc, th) : zc - uc;
      const double tm = VIN ? (zm - tv2_clamp(zm, th)) - tv2_clamp(zm, th) : zm - um;
      const double bn = sgn * (sc + rho * (q > 0 ? tc - tm : tc));
#pragma unroll
      for (int k = 0; k < E - 1; ++k) {
        b[k] = b[k + 1];
        ua[k] = ua[k + 1];
      }
      b[E - 1] = bn;
      ua[E - 1] = uc;
      mine[j] = zc;
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- 2. local sums, end values out, carries in
  double al[E];  // anticausal: al[j] = sum_{k>=1, j+k<=7} r^k b[j+k]
  al[E - 1] = 0.0;
#pragma unroll
  for (int j = E - 1; j >= 1; --j) al[j - 1] = r * (al[j] + b[j]);
  const double Tv = r * (al[0] + b[0]);
#pragma unroll
  for (int j = 1; j < E; ++j) b[j] = __builtin_fma(r, b[j - 1], b[j]);  // b becomes the local causal sum
  Sq[tid] = b[E - 1];
  Tq[tid] = Tv;
  __syncthreads();
  double x[E];
  {
    const double r8 = a.rpow[7];
    // carries: Horner in r^8 over the neighbour threads' end values, far to near, four loads of each array in flight
    // at a time.  Terms beyond G (the group is rounded up to a multiple of 4) are real terms where the neighbour exists
    // and zero where it does not; threads outside [G, NT-G) compute an x nobody reads.
    double cin = 0.0, ain = 0.0;
    for (int g0 = (G + 3) & ~3; g0 > 0; g0 -= 4) {
      double sv4[4], tv4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int il = tid - (g0 - k), ir = tid + (g0 - k);
        sv4[k] = Sq[il < 0 ? 0 : il];
        tv4[k] = Tq[ir > NT - 1 ? NT - 1 : ir];
        if (il < 0) sv4[k] = 0.0;
        if (ir > NT - 1) tv4[k] = 0.0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cin = __builtin_fma(r8, cin, sv4[k]);
        ain = __builtin_fma(r8, ain, tv4[k]);
      }
    }
    const double A = a.green;
#pragma unroll
    for (int j = 0; j < E; ++j)  // a.rpow[k] = r^(k+1): scalar operands
      x[j] = A * (__builtin_fma(a.rpow[j], cin, b[j]) +
                  ((j < E - 1) ? __builtin_fma(a.rpow[(E - 2 - j) & (E - 1)], ain, al[j]) : ain));
  }
  X0q[tid] = x[0];
  X7q[tid] = x[E - 1];
  __syncthreads();

  // ---- 3. z/u update of the owned positions, residual sums; v+ = z+ + u+ replaces v in `mine`
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  const bool owner = tid > G && tid < NT - G - 1 && p0 < o1;
  if (owner) {
    // x of the next position; 0 when this run ends the signal: D*x's last row is x_n itself (totalvariation.m:127)
    const bool ends = p0 + E >= n;
    const double xnext = ends ? 0.0 : X0q[tid + 1];
    double unm = 0.0, dzm = 0.0;  // new u and z - z_prev of position p0 - 1 (recomputed from its old state); 0 at i = 0
    if (p0 > 0) {
      const double xm = X7q[tid - 1];
      const double uom = VIN ? tv2_clamp(vl, th) : ul;
      const double zpm = VIN ? vl - uom : vl;
      const double vnm = uom + (xm - x[0]);
      unm = tv2_clamp(vnm, th);
      dzm = (vnm - unm) - zpm;
    }
    // one position: xn = x of the next one (0 behind the last row)
    auto step = [&](double xj, double xn, double vaj, double uaj) -> double {
      const double uo = VIN ? tv2_clamp(vaj, th) : uaj;
      const double zp = VIN ? vaj - uo : vaj;
      const double ax = xj - xn;            // D*x
      const double v1 = uo + ax;
      const double un = tv2_clamp(v1, th);  // admm.m:548 (c = 0): u + (D x - z) = v+ - z+
      const double zn = v1 - un;            // getProxOps.m:199: soft(u + D x, lambda/rho)
      const double dz = zn - zp;
      const double rr = ax + (-zn), du = un - uo;
      const double g2 = dz - dzm, g3 = un - unm;
      acc[S_R2] += rr * rr;
      acc[S_AX2] += ax * ax;
      acc[S_Z2] += zn * zn;
      acc[S_DZ2] += dz * dz;
      acc[S_U2] += un * un;
      acc[S_DU2] += du * du;
      acc[S_G2] += g2 * g2;
      acc[S_G3] += g3 * g3;
      unm = un;
      dzm = dz;
      return v1;
    };
    if (p0 + E <= o1) {  // every thread but at most one of the launch: branch-free
#pragma unroll
      for (int j = 0; j < E; ++j) {
        mine[j] = step(x[j], (j + 1 < E) ? x[(j + 1) & (E - 1)] : xnext, mine[j], ua[j]);
      }
      if (EXTRA && a.objevals) {  // totalvariation.m:134-135
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const double xn = (j + 1 < E) ? x[(j + 1) & (E - 1)] : xnext;
          if (j + 1 < E || !ends) acc[S_OBJZ] += fabs(xn - x[j]);
          const double e = x[j] - a.s[p0 + j];
          acc[S_OBJX] += e * e;
        }
      }
      if (EXTRA && a.xhist) {  // history columns start at it*n (odd-aligned for odd n): scalar stores
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const double v1 = mine[j], un = tv2_clamp(v1, th);
          a.xhist[it * n + p0 + j] = x[j];
          a.zhist[it * n + p0 + j] = v1 - un;
          a.uhist[it * n + p0 + j] = un;
        }
      }
    } else {
      // the one run the end of the signal cuts (n not a multiple of 8): a rolled loop, x (and u) shifted along
      const int live = static_cast<int>(o1 - p0);
#pragma unroll 1
      for (int j = 0; j < live; ++j) {
        const int64_t i = p0 + j;
        const double xj = x[0], xn = (i + 1 < n) ? x[1] : 0.0;
        const double v1 = step(xj, xn, mine[j], ua[0]);
        const double un = tv2_clamp(v1, th);
        mine[j] = v1;
        if (EXTRA && a.objevals) {
          if (i + 1 < n) acc[S_OBJZ] += fabs(xn - xj);
          const double e = xj - a.s[i];
          acc[S_OBJX] += e * e;
        }
        if (EXTRA && a.xhist) {
          a.xhist[it * n + i] = xj;
          a.zhist[it * n + i] = v1 - un;
          a.uhist[it * n + i] = un;
        }
#pragma unroll
        for (int k = 0; k < E - 1; ++k) {
          x[k] = x[k + 1];
          ua[k] = ua[k + 1];
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  {  // the compact state: the wave's buffer back in coalesced pairs, owned positions only
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pq = 128 * k + 2 * lane, ix = pq + (pq >> 3);
      const int64_t i = c0 + pq;
      const admm_double2 q{stage[ix], stage[ix + 1]};
      if (i >= o0 && i + 1 < o1) store2<NTS>(a.zo + i, q);
      else {
        if (i >= o0 && i < o1) a.zo[i] = q.x;
        if (i + 1 >= o0 && i + 1 < o1) a.zo[i + 1] = q.y;
      }
    }
  }
  // ---- 4. the tile's partial sums (S_DUH2 / S_DZV2 are not produced by this iteration: zero)
  {
    const double q0 = tv2_reduce4(acc[S_R2], acc[S_AX2], acc[S_Z2], acc[S_DZ2]);      // rows: R2 | Z2 | AX2 | DZ2
    const double q1 = tv2_reduce4(acc[S_U2], acc[S_DU2], acc[S_OBJZ], acc[S_OBJX]);   // rows: U2 | OBJZ | DU2 | OBJX
    const double q2 = tv2_reduce4(acc[S_G2], acc[S_G3], 0.0, 0.0);                    // rows: G2 | 0 | G3 | 0
    if (lane == 0) {
#pragma unroll
      for (int s = 12; s < S_COUNT; ++s) sred[wid * S_COUNT + s] = 0.0;  // slots this iteration does not produce
    }
    if ((lane & 15) == 0) {
      const int row = lane >> 4;
      const int s0 = row == 0 ? S_R2 : row == 1 ? S_Z2 : row == 2 ? S_AX2 : S_DZ2;
      const int s1 = row == 0 ? S_U2 : row == 1 ? S_OBJZ : row == 2 ? S_DU2 : S_OBJX;
      const int s2 = row == 0 ? S_G2 : row == 1 ? S_DUH2 : row == 2 ? S_G3 : S_DZV2;
      sred[wid * S_COUNT + s0] = q0;
      sred[wid * S_COUNT + s1] = q1;
      sred[wid * S_COUNT + s2] = q2;
    }
  }
  __syncthreads();
  if (tid < S_COUNT) {
    double v = sred[tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += sred[w * S_COUNT + tid];
    a.part[tid * a.part_stride + tile_id] = v;
  }
}

}  // namespace admm
