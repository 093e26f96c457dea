// finalize_device.h -- the device-side finalize logic of one ADMM iteration (admm.m:612-722: norms, tolerances, H-norm,
// convergence test, the three stop conditions, iteration counter), as an inline device function: run by the
// stand-alone finalize_kernel and, inside the one-launch tails (prox_fin_kernel, tv_fused_kernel), by the last
// workgroup to arrive.
#pragma once
#include "loop_kernels.h"

namespace admm {

// sum over the NT threads that run finalize_body (four waves, or one: whatever the launch's block size); result in
// thread 0
template <int NT>
__device__ __forceinline__ double fin_block_sum(double v, double* scratch) {
  v = wave_sum(v);
  if (NT == 64) return v;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  return threadIdx.x == 0 ? ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3] : 0.0;
}

// Exactly NT threads take part: kBlock (the first four waves of the calling workgroup), or 64 -- the one-wave
// workgroup that rides along with the lower-triangle x-solve (symv.hip).
template <bool COHERENT, int NT = kBlock>
__device__ __forceinline__ void finalize_body(const FinArgs& a) {
  Ctrl* ctrl = a.ctrl;
  __shared__ double scratch[4];
  __shared__ double S[16];
  const int it = ctrl->iter;
  if (a.slots_reduced) {  // row-sharded: already summed over blocks and ranks
    if (threadIdx.x < 16) S[threadIdx.x] = a.slots_reduced[threadIdx.x];
    __syncthreads();
  } else {
    // all slots at once: 16 lanes per slot stride over the block partials, then a 16-lane
    // shuffle tree (fixed order -> reproducible); one round of global loads instead of S_COUNT.
    static_assert(S_COUNT <= 16, "slot layout");
    const int sub = threadIdx.x & 15;
    for (int slot = threadIdx.x >> 4; slot < 16; slot += NT / 16) {
    double v = 0.0;
    if (slot < S_COUNT) {
      const double* __restrict__ ps = a.part + slot * kMaxPartBlocks;
      for (int b0 = 0; b0 < a.nblk; b0 += 128) {  // eight loads per lane issued together (clamped, unconditional)
        double w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int b = b0 + sub + 16 * k;
          const int bc = b < a.nblk ? b : a.nblk - 1;
          w[k] = COHERENT ? __hip_atomic_load(ps + bc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ps[bc];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (b0 + sub + 16 * k < a.nblk) v += w[k];
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (sub == 0) S[slot] = v;
    }
    __syncthreads();
  }
  double ng2 = 0.0, ng3 = 0.0, nx2 = 0.0, objp = 0.0;
  if (a.dual_from_slots) {
    ng2 = S[S_G2];
    ng3 = S[S_G3];
  } else if (a.g && !a.nodualerror) {
    double s2 = 0.0, s3 = 0.0;
    for (int64_t j = threadIdx.x; j < a.nA; j += NT) {
      const double g2 = a.g[a.ldg + j], g3 = a.g[2 * a.ldg + j];
      s2 += g2 * g2;
      s3 += g3 * g3;
    }
    ng2 = fin_block_sum<NT>(s2, scratch);
    ng3 = fin_block_sum<NT>(s3, scratch);
  }
  if (a.x) {
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < a.nA; j += NT) {
      const double xv = a.x[j];
      s += xv * xv;
      if (a.xhist) a.xhist[static_cast<int64_t>(it) * a.nA + j] = xv;
    }
    nx2 = fin_block_sum<NT>(s, scratch);
  }
  if (a.objp_reduced) {
    objp = a.objp_reduced[0];
  } else if (a.objpart) {
    double s = 0.0;
    for (int b = threadIdx.x; b < a.nobjpart; b += NT) s += a.objpart[b];
    objp = fin_block_sum<NT>(s, scratch);
  }
  if (threadIdx.x != 0) return;
  const double Mlen = static_cast<double>(a.len_global > 0 ? a.len_global : a.len);

  const int i1 = it + 1;  // 1-based iteration number (admm.m loop variable)
  const double NaN = __longlong_as_double(0x7ff8000000000000LL);
  double hn = 0.0;
  if (a.use_h) {  // admm.m:305-306 with w = [x; z; rho*u]: rho*||dz||^2 + rho*||rho*du||^2
    hn = a.rhoH * S[S_DZ2] + a.rhoH * (a.rho * a.rho) * S[S_DU2];
    a.hnorm[it] = hn;
  }
  if (a.objevals && a.objv) {
    const double ov = a.obj_scale_part * objp + a.obj_scale_z * S[S_OBJZ] + a.obj_scale_x * S[S_OBJX] +
                      a.obj_half_xnorm * nx2 + a.obj_const;
    a.objv[it] = ov;
    if (a.obj_track_bound) {
      // 1/2*||D*x - s||^2 = sum_i x_i*(1/2*(y_i - rho*x_i) - (D's)_i) + 1/2*s's: terms of the size of s's that cancel down
      // to the data misfit.  eps * (sum of their magnitudes) / |objective| bounds the relative rounding error of the
      // recorded value; the host reads the run's maximum after every batch and goes back to the literal D*x pass when
      // it leaves 1e-10 (engine_run.hip)
      const double bound = 2.220446049250313e-16 * (fabs(a.obj_scale_x) * S[S_OBJA] + fabs(a.obj_const)) /
                           fmax(fabs(ov), 1e-300);
      if (!(bound <= ctrl->obj_bound)) ctrl->obj_bound = bound;  // NaN-safe maximum
    }
  }
  bool stop = false;
  if (a.alg == 2) {
    a.avals[it] = ctrl->acurr;
    a.dvals[it] = ctrl->d;
    a.restarted[it] = ctrl->restart_flag;
    // admm.m:706: abs(d - dprev) <= DVALTOL*dprev
    if (i1 >= 2 && fabs(ctrl->d - ctrl->dprev) <= a.dvaltol * ctrl->dprev) stop = true;
  } else {
    double coef = 0.0;
    if (a.alg == 1) {
      const double aprev = ctrl->acurr;
      const double acn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * aprev * aprev));
      coef = (aprev - 1.0) / acn;
      ctrl->aprev = aprev;
      ctrl->acurr = acn;
      ctrl->coef = coef;
      a.avals[it] = acn;
    }
    // options.specialnorms (admm.m:612-616) = lassonorms (getProxOps.m:1335-1343): both values are
    // SQUARED sums (q10): sum_k ||x_k - xave||^2 and N*rho^2*||xave - xaveprev||^2
    double pn = a.specialnorms ? S[S_R2] : sqrt(S[S_R2]);
    if (a.specialnorms && a.cons_q) pn = fmax(a.cons_q[0] - static_cast<double>(a.nslices_total) * S[S_G2], 0.0);
    double dn, de;
    if (a.specialnorms) {
      dn = static_cast<double>(a.nslices_total) * (a.rho * a.rho) * S[S_G2];
      de = a.nodualerror ? NaN : sqrt(Mlen) * a.abstol + a.reltol * (a.rho * sqrt(S[S_U2]));
    } else if (a.nodualerror) {
      dn = NaN;
      de = NaN;
    } else {
      const double base = a.a_identity ? sqrt(S[S_DZ2]) : sqrt(ng2);
      // alg 0: ||rho*At(B(z - zprev))||; alg 1: rho*||At(B(z - v))|| with z - v = -coef*(z - zprev)
      dn = (a.alg == 0) ? a.rho * base : a.rho * (fabs(coef) * base);
      const double un = a.a_identity ? sqrt(S[S_U2]) : sqrt(ng3);
      de = sqrt(Mlen) * a.abstol + a.reltol * (a.rho * un);
    }
    if (a.norms_given) {  // options.specialnorms, the caller's handle (admm.m:612-616)
      pn = a.norms_given[0];
      dn = a.norms_given[1];
    }
    const double pe = sqrt(Mlen) * a.abstol +
                      a.reltol * fmax(fmax(sqrt(S[S_AX2]), sqrt(S[S_Z2])), a.cnorm);
    a.pnorm[it] = pn;
    a.dnorm[it] = dn;
    a.perr[it] = pe;
    a.derr[it] = de;
    // admm.m:710-713
    if ((a.stopcond == ADMM_STOP_STANDARD || a.stopcond == ADMM_STOP_BOTH) && !a.domaxiters && pn < pe &&
        (a.nodualerror || dn < de))
      stop = true;
  }
  if (a.use_h) {
    if (a.convtest && i1 >= 2) {  // admm.m:686-701
      const double H2 = a.hnorm[it], H1 = a.hnorm[it - 1];
      if (a.alg == 0 && H1 > 2.220446049250313e-16 && H2 > H1 && !((H2 - H1) <= H1 * a.convtol)) {
        ctrl->convfail = i1;
        ctrl->steps = i1;
        ctrl->iter = i1;
        ctrl->stop = 1;
        return;
      }
    }
    // admm.m:719-722
    if ((a.stopcond == ADMM_STOP_HNORM || a.stopcond == ADMM_STOP_BOTH) && !a.domaxiters && i1 > 2 &&
        hn <= a.Hnormtol)
      stop = true;
  }
  ctrl->iter = i1;
  ctrl->steps = i1;
  if (stop || i1 >= a.maxiters) ctrl->stop = 1;
}


}  // namespace admm
