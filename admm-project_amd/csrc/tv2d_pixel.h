// tv2d_pixel.h -- one pixel of the fused 2-D TV pass (tv2d.hip: tv2d_fused_kernel has the derivation and the same
// arithmetic, statement for statement): z/u update on the compact state v = z + u, the D' stencils of the dual
// residual / tolerance and the next right-hand side b = s + rho*D'(z+ - u+).  Shared with the kernel that sends b
// straight into the forward column transform (dct.hip: tv2d_fused_dct_kernel).
#pragma once
#include "kernels.h"
#include "loop_kernels.h"
#include "tv2d.h"

namespace admm {

__device__ __forceinline__ double tv2px_clamp(double v, double t) { return __builtin_fmin(__builtin_fmax(v, -t), t); }

// what pixel (i, j) reads: x at the 5-point stencil, the dual state of its own two differences, of the vertical
// difference above it and of the horizontal difference to its left (z as given, or v = z + u when VIN), the image
struct Tv2Px {
  double xi, x_dn, x_up, x_lf, x_rt;
  double in0, in1, in_up, in_lf;      // VIN: v; else z
  double u0, u1, u_up, u_lf;          // !VIN only
  double si;
};

// base = j*H + row0 (first row of the thread block's row chunk), r = the thread's row inside it; o_up / o_dn: row offsets
// of the neighbours relative to (base - 1) / base, clamped at the image border; lbase / rbase: the same chunk in the
// columns j -+ 1, clamped.  Every load unconditional.
template <bool VIN>
__device__ __forceinline__ void tv2px_load(const Tv2Args& a, int64_t N, int64_t base, int64_t lbase, int64_t rbase,
                                           uint32_t r, uint32_t o_up, uint32_t o_dn, Tv2Px& p) {
  const double* __restrict__ xc = a.x + base;
  const double* __restrict__ zc = a.z + base;
  p.xi = xc[r];
  p.x_dn = xc[o_dn];
  p.x_up = (xc - 1)[o_up];
  p.x_lf = (a.x + lbase)[r];
  p.x_rt = (a.x + rbase)[r];
  p.in0 = zc[r];
  p.in1 = (zc + N)[r];
  p.in_up = (zc - 1)[o_up];
  p.in_lf = (a.z + N + lbase)[r];
  if (!VIN) {
    const double* __restrict__ uc = a.u + base;
    p.u0 = uc[r];
    p.u1 = (uc + N)[r];
    p.u_up = (uc - 1)[o_up];
    p.u_lf = (a.u + N + lbase)[r];
  }
  p.si = (a.s + base)[r];
}

// returns b(i, j); stores the new compact state and the history columns, adds the pixel's terms to acc
template <bool VIN>
__device__ __forceinline__ double tv2px_apply(const Tv2Args& a, int64_t N, int64_t it, int64_t base, uint32_t r, bool hasv,
                                              bool up, bool hash, bool left, const Tv2Px& p, double (&acc)[S_COUNT]) {
  const double t = a.thresh;
  double z_own[2], u_own[2], z_up, u_up, z_lf, u_lf;
  if (VIN) {
    u_own[0] = tv2px_clamp(p.in0, t);
    u_own[1] = tv2px_clamp(p.in1, t);
    u_up = tv2px_clamp(p.in_up, t);
    u_lf = tv2px_clamp(p.in_lf, t);
    z_own[0] = p.in0 - u_own[0];
    z_own[1] = p.in1 - u_own[1];
    z_up = p.in_up - u_up;
    z_lf = p.in_lf - u_lf;
  } else {
    z_own[0] = p.in0;
    z_own[1] = p.in1;
    z_up = p.in_up;
    z_lf = p.in_lf;
    u_own[0] = p.u0;
    u_own[1] = p.u1;
    u_up = p.u_up;
    u_lf = p.u_lf;
  }
  const double xi = p.xi, si = p.si;
  const double d[2] = {hasv ? xi - p.x_dn : 0.0, hash ? xi - p.x_rt : 0.0};
  double zn[2], un[2];
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    const double ax = d[part];
    const double uo = u_own[part];
    const double vn = uo + ax;
    un[part] = tv2px_clamp(vn, t);
    zn[part] = vn - un[part];
    const double rr = ax - zn[part], dz = zn[part] - z_own[part], du = un[part] - uo;
    acc[S_R2] += rr * rr;
    acc[S_AX2] += ax * ax;
    acc[S_Z2] += zn[part] * zn[part];
    acc[S_DZ2] += dz * dz;
    acc[S_U2] += un[part] * un[part];
    acc[S_DU2] += du * du;
    if (a.objevals) acc[S_OBJZ] += fabs(ax);
    (a.zo + part * N + base)[r] = vn;
    if (a.zhist) {
      a.zhist[it * 2 * N + part * N + base + r] = zn[part];
      a.uhist[it * 2 * N + part * N + base + r] = un[part];
    }
  }
  if (a.objevals) {
    const double e = xi - si;
    acc[S_OBJX] += e * e;
  }
  if (a.xhist) a.xhist[it * N + base + r] = xi;
  // the rows above / to the left: new z, u of (i-1, j) in the vertical part and of (i, j-1) in the horizontal one
  const double vnu = u_up + (p.x_up - xi), vnl = u_lf + (p.x_lf - xi);
  const double unu = tv2px_clamp(vnu, t), unl = tv2px_clamp(vnl, t);
  const double znu = vnu - unu, znl = vnl - unl;
  // D'w at (i, j), in tv2_dt's order of operations
  double g2 = 0.0, g3 = 0.0, gb = 0.0;
  if (hasv) {
    g2 += zn[0] - z_own[0];
    g3 += un[0];
    gb += zn[0] - un[0];
  }
  if (up) {
    g2 -= znu - z_up;
    g3 -= unu;
    gb -= znu - unu;
  }
  if (hash) {
    g2 += zn[1] - z_own[1];
    g3 += un[1];
    gb += zn[1] - un[1];
  }
  if (left) {
    g2 -= znl - z_lf;
    g3 -= unl;
    gb -= znl - unl;
  }
  acc[S_G2] += g2 * g2;
  acc[S_G3] += g3 * g3;
  return si + a.rho * gb;
}

}  // namespace admm
