// prox_device.h -- the fused element update of one ADMM iteration (admm.m:515-569, 608-654) as inline device functions:
// shared by the generic prox kernels (loop_kernels.hip) and the one-launch unwrapped iteration (unwrapped.hip).
#pragma once
#include "loop_kernels.h"

namespace admm {

__device__ __forceinline__ double soft(double v, double t) {
  // sign(v).*subplus(abs(v) - t)   getProxOps.m:937
  const double a = fabs(v) - t;
  const double p = a > 0.0 ? a : 0.0;
  return (v > 0.0) ? p : ((v < 0.0) ? -p : 0.0 * p);
}

__device__ __forceinline__ double huber_cvx(double x) {
  const double a = fabs(x);
  return a <= 1.0 ? x * x : 2.0 * a - 1.0;
}

struct ProxIn {  // every input of one element (see prox_load)
  double zp, u_old, uhat_i, c_i, ell_i, zg_i, lb_i, ub_i, v_i, add_i, rhs_i;
};

__device__ __forceinline__ ProxIn prox_load(const ProxArgs& a, int64_t i) {
  // Every input this element can need, loaded up front and unconditionally (operands a variant does not use are
  // redirected to z: a cache hit): one memory round trip.  With the loads behind their (kernel-uniform) conditions
  // hipcc waits for each one separately -- eight dependent L2 round trips, 9 us for a kernel with 2 us of work.
  const double* puhat = (a.alg != 0) ? a.uhat : a.u;
  const double* pc = a.c ? a.c : a.z;
  // ell / lb / ub / zgiven / rhs_add are null unless this launch reads them (launch_prox clears the unused ones:
  // some of them are shorter than len otherwise).  One null test per pointer: a compound condition here made hipcc
  // (ROCm 7.2) drop one of its terms when it split the select across blocks.
  const double* pell = a.ell ? a.ell : a.z;
  const double* pzg = a.zgiven ? a.zgiven : a.z;
  const double* plb = a.lb ? a.lb : a.z;
  const double* pub = a.ub ? a.ub : a.z;
  const double* pv = (a.alg == 2) ? a.v : a.z;
  const double* padd = a.rhs_add ? a.rhs_add : a.z;
  const double* prhs = (a.objx == OBJX_SOLVE || a.objx == OBJX_SOLVE_QP) ? a.rhs : a.z;  // the right-hand side the x-update just solved with
  ProxIn in;
  in.zp = a.z[i];
  in.u_old = a.u[i];
  in.uhat_i = puhat[i];
  in.c_i = pc[i];
  in.ell_i = pell[i];
  in.zg_i = pzg[i];
  in.lb_i = plb[i];
  in.ub_i = pub[i];
  in.v_i = pv[i];
  in.add_i = padd[i];
  in.rhs_i = prhs[i];
  return in;
}

// Everything the loop does with element i once Ax_i is known (admm.m:515-569, 608-654): relaxation, z-prox,
// u-update, fast-ADMM extrapolation, histories, the residual / objective partial sums and the next rhs.
// rhs_out (optional) receives c + zx - ux, the element of the next unwrapped x-update's right-hand side (RHS_T1),
// whatever a.rhs_kind stores.
__device__ __forceinline__ void prox_apply(const ProxArgs& a, int64_t i, double ax, int64_t it, double kcoef,
                                           const ProxIn& in, double (&acc)[S_COUNT], double* rhs_out = nullptr) {
  const double zp = in.zp, u_old = in.u_old, uhat_i = in.uhat_i, c_i = in.c_i, ell_i = in.ell_i, zg_i = in.zg_i;
  const double lb_i = in.lb_i, ub_i = in.ub_i, v_i = in.v_i, add_i = in.add_i;
  if (a.a_identity) {
    if (a.x_out) a.x_out[i] = ax;
    if (a.xhist) a.xhist[it * a.len + i] = ax;
  }
  const double uo = (a.alg == 0) ? u_old : uhat_i;
  const double ci = a.c ? c_i : 0.0;
  // admm.m:517  Axhat = relax*A(x) - (1-relax)*(B(zprev) - c),  B = -1
  const double axh = (a.relax != 1.0) ? a.relax * ax - (1.0 - a.relax) * ((-zp) - ci) : ax;
  const double v = (axh + uo) - ci;
  double zn;
  switch (a.prox) {
    case PROX_SOFT:
      zn = soft(v, a.t);
      break;
    case PROX_HUBER:
      zn = 1.0 / (1.0 + a.rho) * (a.rho * v + soft(v, 1.0 + 1.0 / a.rho));
      break;
    case PROX_HINGE: {
      const double l = ell_i;
      const double lv = l * v;
      zn = v + l * fmax(fmin(1.0 - lv, a.t), 0.0);
      break;
    }
    case PROX_01: {
      const double l = ell_i;
      const double s = l * v;
      const double y = ((s >= 1.0) || (s < (1.0 - sqrt(2.0 / a.t)))) ? s : 1.0;
      zn = l * y;
      break;
    }
    case PROX_GIVEN:
      zn = zg_i;
      break;
    case PROX_POS:
      zn = fmax(v, 0.0);
      break;
    default:  // PROX_BOX
      zn = fmin(ub_i, fmax(lb_i, v));
      break;
  }
  const double Bz = -zn;
  const double un = uo + ((axh + Bz) - ci);  // admm.m:542-550
  const double r = (ax + Bz) - ci;           // admm.m:621 uses Ax, not Axhat
  const double dzv = zn - zp;
  acc[S_R2] += r * r;
  acc[S_AX2] += ax * ax;
  acc[S_Z2] += zn * zn;
  acc[S_DZ2] += dzv * dzv;
  acc[S_U2] += un * un;
  const double du = un - u_old;
  acc[S_DU2] += du * du;
  if (a.objz == OBJZ_ABS) acc[S_OBJZ] += fabs(zn);
  else if (a.objz == OBJZ_HUBER) acc[S_OBJZ] += huber_cvx(zn);
  if (a.objx == OBJX_HINGE) acc[S_OBJX] += fmax(1.0 - ell_i * ax, 0.0);
  else if (a.objx == OBJX_ZEROONE) {
    const double q = 1.0 - ell_i * ax;
    acc[S_OBJX] += (q > 0.0) ? 1.0 : 0.0;  // max(sign(q),0)
  } else if (a.objx == OBJX_ABS) acc[S_OBJX] += fabs(ax);
  else if (a.objx == OBJX_DOT) acc[S_OBJX] += ell_i * ax;
  else if (a.objx == OBJX_SOLVE) {
    const double term = ax * (0.5 * (in.rhs_i - a.rho_solve * ax) - add_i);
    acc[S_OBJX] += term;
    acc[S_OBJA] += fabs(ax * (0.5 * (in.rhs_i - a.rho_solve * ax))) + fabs(ax * add_i);  // what cancels in it
  } else if (a.objx == OBJX_SOLVE_QP) {
    const double term = ax * (0.5 * (in.rhs_i - a.rho_solve * ax) + add_i);
    acc[S_OBJX] += term;
    acc[S_OBJA] += fabs(ax * (0.5 * (in.rhs_i - a.rho_solve * ax))) + fabs(ax * add_i);
  }

  a.z[i] = zn;
  a.u[i] = un;
  if (a.dz) a.dz[i] = dzv;
  if (a.zhist) a.zhist[it * a.len + i] = zn;
  if (a.uhist) a.uhist[it * a.len + i] = un;

  double zx = zn, ux = un;
  if (a.alg == 1) {  // admm.m:568-569
    zx = zn + kcoef * (zn - zp);
    ux = un + kcoef * (un - u_old);
    a.v[i] = zx;
    a.uhat[i] = ux;
    if (a.vhist) a.vhist[it * a.len + i] = zx;
    if (a.uhathist) a.uhathist[it * a.len + i] = ux;
  } else if (a.alg == 2) {  // decision needs d first: keep what the extrapolation kernel needs
    const double vo = v_i;
    const double duh = un - uo, dzv2 = zn - vo;
    acc[S_DUH2] += duh * duh;
    acc[S_DZV2] += dzv2 * dzv2;
    a.zprev[i] = zp;
    a.uprev[i] = u_old;
  }
  if (rhs_out) *rhs_out = (ci + zx) - ux;
  if (a.alg != 2 && a.rhs) {
    switch (a.rhs_kind) {
      case RHS_RHO_DTS:
        a.rhs[i] = a.rho * (zx - ux) + add_i;
        break;
      case RHS_RHO_MINUS_Q:
        a.rhs[i] = a.rho * (zx - ux) - add_i;
        break;
      case RHS_DIFF:
        a.rhs[i] = zx - ux;
        break;
      case RHS_T1:
        a.rhs[i] = (ci + zx) - ux;
        break;
      default:
        break;
    }
  }
}

}  // namespace admm
