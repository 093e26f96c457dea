// engine_run_consensus.hip -- the iteration sequence of consensus lasso (getProxOps.m:383-442, 1217-1343; one slice
// per rank when row-sharded), split out of admm_engine_run.
#include <cstdlib>

#include "engine_internal.h"

namespace admm {

int run_consensus_lasso(admm_engine* e, RunState& rs, admm_run_summary* summary) {
  const admm_options& o = rs.o;
  const int alg = rs.alg;
  const int32_t N = rs.N;
  const int64_t len = rs.len;
  ProxArgs& pa = rs.pa;
  FinArgs& fa = rs.fa;
  ExtrapArgs& xa = rs.xa;
  (void)alg; (void)len; (void)pa; (void)xa;

  if (alg != 0) return fail(ADMM_E_UNSUPPORTED, "fast/accelerated ADMM is not implemented for consensus lasso");
  // options.relax: admm.m:515-532 only changes what it hands to zming and to the u-update -- the consensus closures
  // ignore every argument (getProxOps.m:1217, 1272, 1312 take `~`) and options.altu replaces the u-update, so the
  // iterates, norms and histories are those of relax = 1: accepted, no effect.
  const bool shard = e->comm && comm_nranks(e->comm) > 1;
  const int32_t K = static_cast<int32_t>(e->cslices.size());
  const int64_t n = e->n, ldn = e->cldn;
  // closure state at getproxops time: x_k = u_k = 0, z = 0, xave = 0 (getProxOps.m:390-411); admm's own
  // u starts at options.u0 (admm.m:254) and only enters the first H-norm difference
  ADMM_HIP_TRY(hipMemsetAsync(e->cX, 0, sizeof(double) * K * ldn, e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(e->cU, 0, sizeof(double) * K * ldn, e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(e->czc, 0, sizeof(double) * ldn, e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(e->cxave, 0, sizeof(double) * ldn, e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(e->cxaveprev, 0, sizeof(double) * ldn, e->stream));
  ADMM_HIP_TRY(hipMemcpyAsync(e->cubar, e->u, sizeof(double) * n, hipMemcpyDeviceToDevice, e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(e->cobjpart, 0, sizeof(double) * K * kMaxPartBlocks, e->stream));
  // y_k of the first x-update: rho*(z - u_k) + D_k's_k with z = u_k = 0; every later one comes from the update kernel
  ADMM_HIP_TRY(hipMemcpyAsync(e->cY, e->cDts, sizeof(double) * K * ldn, hipMemcpyDeviceToDevice, e->stream));
  ConsArgs ca{};
  ca.n = n;
  ca.ldn = ldn;
  ca.K = K;
  ca.Ntot = e->cons_total;
  ca.rho = o.rho;
  ca.lambda = e->lambda;
  ca.sums = e->csums;
  ca.X = e->cX;
  ca.U = e->cU;
  ca.zc = e->czc;
  ca.xave = e->cxave;
  ca.xaveprev = e->cxaveprev;
  ca.ubar = e->cubar;
  ca.xhist = e->xhist;
  ca.zhist = e->zhist;
  ca.uhist = e->uhist;
  ca.part = e->part;
  ca.Dts = e->cDts;
  ca.Y = e->cY;
  fa.specialnorms = 1;
  fa.nslices_total = e->cons_total;
  fa.g = nullptr;
  fa.x = nullptr;
  fa.xhist = nullptr;
  fa.obj_scale_part = o.objevals ? 0.5 : 0.0;  // lasso.m:227 with the z admm holds (zeros): 0.5*||D*x - s||^2
  fa.obj_scale_z = 0.0;
  const int check_c = o.check_every > 0 ? o.check_every : (o.domaxiters ? 64 : 8);
  const auto t0 = std::chrono::steady_clock::now();
  int32_t done = 0;
  bool stop_seen = false;
  // with the batched x-solve in front, the finalize logic of iteration i is deferred into iteration i + 1's batched
  // launch (one passenger workgroup; engine_run.hip: defer_fin has the reasoning); a batch's last one runs stand-alone
  const bool batched = e->cMptr && std::getenv("ADMM_HIP_CONS_UNBATCHED") == nullptr;
  const bool defer_fin = batched && std::getenv("ADMM_HIP_NO_DEFERRED_FINALIZE") == nullptr;
  FinArgs fprev{};
  bool fin_pending = false;
  while (done < N && !stop_seen) {
    const int32_t batch = (N - done < check_c) ? N - done : check_c;
    for (int32_t b = 0; b < batch; ++b) {
      {
        TimerScope ts(e, ADMM_K_XSOLVE);
        if (batched)
          launch_symv_lower_batch(e->cslices[0].fac.planSy, e->cMptr, K, e->cY, ldn, e->csyN, e->csyT, e->cpstride,
                                  e->ctrl, e->stream, fin_pending ? &fprev : nullptr);
        else
        for (int32_t k = 0; k < K; ++k) {  // getProxOps.m:1228-1253
          ConsSlice& sl = e->cslices[k];
          const double* yk = e->cY + k * ldn;  // rho*(z - u_k) + D_k's_k, left by the previous update kernel
          if (e->csyN) {  // leave the partial rows: the exchange kernel assembles x_k
            launch_symv_lower(sl.fac.planSy, sl.fac.Minv, sl.fac.ldM, yk, e->csyN + k * e->cpstride,
                              e->csyT + k * e->cpstride, nullptr, e->ctrl, e->stream, 0, 1, false);
          } else if (!sl.fat) {
            apply_slice_factor(e, sl.fac, yk, e->cX + k * ldn);
          } else {  // x_k = y/rho - D_k'(U\(L\(D_k y)))/rho^2   (getProxOps.m:1204 form, q12)
            launch_gemv_n(sl.planN, sl.D, yk, e->partDN, e->ctrl, e->stream);
            launch_sum_partials(e->partDN, sl.planN.nchunk, sl.planN.ldy, sl.m, e->tmpA, e->ctrl, e->stream);
            apply_slice_factor(e, sl.fac, e->tmpA, e->tmpB);
            launch_gemv_t(sl.planT, sl.D, e->tmpB, nullptr, nullptr, 1, e->partDT, e->ctrl, e->stream);
            launch_combine(e->partDT, sl.planT.nchunk, sl.planT.ldg, -1.0 / (o.rho * o.rho), yk, 1.0 / o.rho, nullptr,
                           e->cX + k * ldn, n, e->ctrl, e->stream);
          }
        }
      }
      const SymvPlan* gp = e->csyN ? &e->cslices[0].fac.planSy : nullptr;
      int nblk = 1;
      const bool one_tail = !shard && gp && cons_gather_update_ok(ca) && std::getenv("ADMM_HIP_CONS_TWO_TAIL") == nullptr;
      if (one_tail) {
        TimerScope ts(e, ADMM_K_PROX);
        launch_cons_gather_update(ca, e->csyN, e->csyT, e->cpstride, gp->ldp, gp->ntile, e->ctrl, &nblk, e->stream);
      } else if (shard) {
        // X1 + X2 in ONE collective: [sum x_k; sum u_k; q] with q = sum_k ||x_k - xave_prev||^2 (consensus.hip)
        const int nq = gp ? launch_cons_gather_sum(n, ldn, K, e->csyN, e->csyT, e->cpstride, gp->ldp, gp->ntile, e->cX,
                                                   e->cU, e->csums, e->cxave, e->objpart, e->ctrl, e->stream)
                          : launch_cons_sum(n, ldn, K, e->cX, e->cU, e->csums, e->cxave, e->objpart, e->ctrl, e->stream);
        launch_pack_sum(e->objpart, nq, e->csums + 2 * ldn, e->ctrl, e->stream);
        ADMM_TRY(comm_allreduce_device(e->comm, e->csums, static_cast<size_t>(2 * ldn + 1), e->stream));
      } else if (gp) {
        launch_cons_gather_sum(n, ldn, K, e->csyN, e->csyT, e->cpstride, gp->ldp, gp->ntile, e->cX, e->cU, e->csums,
                               nullptr, nullptr, e->ctrl, e->stream);
      } else {
        launch_cons_sum(n, ldn, K, e->cX, e->cU, e->csums, nullptr, nullptr, e->ctrl, e->stream);
      }
      if (!one_tail) {
        TimerScope ts(e, ADMM_K_PROX);
        launch_cons_update(ca, e->ctrl, &nblk, e->stream);
      }
      fa.nblk = nblk;
      fa.objpart = nullptr;
      fa.nobjpart = 0;
      fa.slots_reduced = nullptr;
      fa.objp_reduced = nullptr;
      if (o.objevals) {
        TimerScope ts(e, ADMM_K_GEMV_N);
        for (int32_t k = 0; k < K; ++k) {
          ConsSlice& sl = e->cslices[k];
          int nob = 0;
          launch_gemv_n(sl.planN, sl.D, e->cxave, e->partDN, e->ctrl, e->stream);
          launch_residual_sq(e->partDN, sl.planN.nchunk, sl.planN.ldy, sl.s, sl.m, e->cobjpart + k * kMaxPartBlocks,
                             &nob, e->ctrl, e->stream);
        }
        fa.objpart = e->cobjpart;
        fa.nobjpart = K * kMaxPartBlocks;
      }
      if (shard) {
        // every slot the finalize logic reads is either identical on all ranks (means, dual sums) or comes from q;
        // only the objective 0.5*sum_k ||D_k*xave - s_k||^2 (objevals) is a rank-local sum that needs its own exchange
        fa.cons_q = e->csums + 2 * ldn;
        if (o.objevals) {
          launch_pack_sum(e->cobjpart, K * kMaxPartBlocks, e->red + 17, e->ctrl, e->stream);
          ADMM_TRY(comm_allreduce_device(e->comm, e->red + 17, 1, e->stream));
          fa.objp_reduced = e->red + 17;
        }
      }
      if (defer_fin) {
        fprev = fa;
        fin_pending = true;
      } else {
        TimerScope ts(e, ADMM_K_FINALIZE);
        launch_finalize(fa, e->stream);
      }
    }
    done += batch;
    if (fin_pending) {
      launch_finalize(fprev, e->stream);
      fin_pending = false;
    }
    {  // poll after every batch (see engine_run.hip)
      ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
      ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
      if (e->ctrl_host->stop) stop_seen = true;
    }
  }
  // what admm holds at exit: x = mean x_k, z = 0 (q9), u = mean u_k
  ADMM_HIP_TRY(hipMemcpyAsync(e->x, e->cxave, sizeof(double) * n, hipMemcpyDeviceToDevice, e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(e->z, 0, sizeof(double) * n, e->stream));
  ADMM_HIP_TRY(hipMemcpyAsync(e->u, e->cubar, sizeof(double) * n, hipMemcpyDeviceToDevice, e->stream));
  ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(ADMM_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
  }
  const double rt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (e->profiling) collect_timers(e);
  const int32_t steps = e->ctrl_host->steps;
  e->last = admm_run_summary{};
  e->last.steps = steps;
  e->last.stopped_early = (steps < N) ? 1 : 0;
  e->last.convtest_failed_at = e->ctrl_host->convfail;
  e->last.runtime_s = rt;
  e->last.objopt = NAN;
  if (o.objevals && steps > 0) {
    double v = NAN;
    ADMM_HIP_TRY(hipMemcpy(&v, e->objv + (steps - 1), sizeof(double), hipMemcpyDeviceToHost));
    e->last.objopt = v;
  }
  e->has_run = true;
  if (summary) *summary = e->last;
  return ADMM_OK;
}

}  // namespace admm
