// engine_run.hip -- C ABI (include/admm_engine.h), part 2: run() = admm.m:252-767, the host side of the
// device-resident ADMM loop (x-update dispatch, the per-problem iteration sequences, stop polling, epilogue).
#include "engine_internal.h"

extern "C" {

static int cg_apply(admm_engine* e, const double* v, const double** qin, int32_t* nchunk, int64_t* ldq) {
  if (e->problem == ADMM_PROB_TV2D) {  // operator I + rho*D'D: the stencil part here, the identity via shift = 1
    TimerScope ts(e, ADMM_K_GEMV_N);
    launch_tv2d_laplace(e->tv2_H, e->tv2_W, e->last_opts.rho, v, e->cg_tmp, e->ctrl, e->stream);
    *qin = e->cg_tmp;
    *nchunk = 1;
    *ldq = 0;
    return ADMM_OK;
  }
  const Ctrl* sk = e->cg_skip;  // no-ops once this solve has converged (or the run has stopped)
  {
    TimerScope ts(e, ADMM_K_GEMV_N);
    launch_gemv_n(e->planDN, e->D, v, e->partDN, sk, e->stream);
  }
  launch_sum_partials(e->partDN, e->planDN.nchunk, e->planDN.ldy, e->m, e->tmpA, sk, e->stream);
  {
    TimerScope ts(e, ADMM_K_GEMV_T);
    launch_gemv_t(e->planDT, e->D, e->tmpA, nullptr, nullptr, 1, e->partDT, sk, e->stream);
  }
  *qin = e->partDT;
  *nchunk = e->planDT.nchunk;
  *ldq = e->planDT.ldg;
  if (e->comm && comm_nranks(e->comm) > 1) {  // sum_g D_g'(D_g v): n doubles per inner iteration
    launch_sum_partials_t(e->planDT, e->partDT, 1, e->cg_tmp, round_up(e->n, 2), e->ctrl, e->stream);
    ADMM_TRY(comm_allreduce_device(e->comm, e->cg_tmp, static_cast<size_t>(e->n), e->stream));
    *qin = e->cg_tmp;
    *nchunk = 1;
    *ldq = 0;
  }
  return ADMM_OK;
}

}  // extern "C"

namespace admm {

// x <- solve of (D'D + shift I) x = y by warm-started CG (cg.hip); one poll of the device per solve
int cg_solve(admm_engine* e, const double* y) {
  if (e->problem == ADMM_PROB_TV2D) return cg_solve_tv2d(e, y);
  CgArgs a{};
  a.n = e->n;
  a.shift = (e->problem == ADMM_PROB_TV2D) ? 1.0 : (e->cg_shift_is_rho ? e->last_opts.rho : 0.0);
  a.tol = e->cg_tol;
  a.maxit = e->cg_maxit;
  a.y = y;
  a.x = e->x;
  a.r = e->cg_r;
  a.p = e->cg_p;
  a.q = e->cg_q;
  a.part = e->cg_part;
  a.st = e->cg_st;
  a.ctrl = e->ctrl;
  a.skip = &e->cg_skip->stop;
  // clear iters/done of the previous solve (total keeps counting) and re-arm the operator kernels
  ADMM_HIP_TRY(hipMemsetAsync(&e->cg_st->iters, 0, 2 * sizeof(int32_t), e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(&e->cg_skip->stop, 0, sizeof(int32_t), e->stream));
  const double* qin;
  int32_t nchunk;
  int64_t ldq;
  ADMM_TRY(cg_apply(e, e->x, &qin, &nchunk, &ldq));
  CgArgs a0 = a;
  a0.p = e->x;  // q = D'D x + shift*x
  launch_cg_q(a0, qin, nchunk, ldq, false, e->stream);
  launch_cg_init(a, e->stream);
  // everything enqueued after convergence is a no-op (operator kernels included), so a chunk can be as long as
  // the previous solve was: one or two polls of the device per solve instead of one per 4 iterations
  const int chunk = e->cg_chunk;
  for (int done_it = 0; done_it < e->cg_maxit;) {
    const int k = (e->cg_maxit - done_it < chunk) ? e->cg_maxit - done_it : chunk;
    for (int c = 0; c < k; ++c) {
      ADMM_TRY(cg_apply(e, e->cg_p, &qin, &nchunk, &ldq));
      launch_cg_q(a, qin, nchunk, ldq, true, e->stream);
      launch_cg_step_tail(a, e->stream);
    }
    done_it += k;
    ADMM_HIP_TRY(hipMemcpyAsync(e->cg_st_host, e->cg_st, sizeof(CgState), hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->cg_st_host->done || e->ctrl_host->stop) break;
  }
  e->cg_chunk = std::min(64, std::max(4, static_cast<int>(e->cg_st_host->iters) + 2));
  return ADMM_OK;
}

}  // namespace admm

extern "C" {


static int factor_x_update(admm_engine* e, const double** axsrc, int32_t* naxpart, int64_t* axld, const double** axt,
                           int32_t* axtri, bool leave_partials);

// the one-block triangular solves (symv.hip: tri1_*) leave the backward pass's partial rows to the one-launch tail
static bool xsolve_tri1_partials(const admm_engine* e) {
  return e->xfac.mode == ADMM_XSOLVE_TRSV && e->xfac.trsv.one && !e->xcb && e->xsolve != ADMM_XSOLVE_CG && !e->fat &&
         (e->problem == ADMM_PROB_LASSO || e->problem == ADMM_PROB_QP_BOUNDED);
}

// the lower-triangle x-solve may hand its partial rows to the one-launch tail instead of reducing them itself
static bool xsolve_has_partials(const admm_engine* e) {
  if (xsolve_tri1_partials(e)) return true;
  return e->xfac.mode == ADMM_XSOLVE_INVERSE && e->xfac.Minv && e->xfac.n >= kSymvHalfMin && !e->sy_split && !e->xcb &&
         e->xsolve == ADMM_XSOLVE_INVERSE && !e->fat &&
         (e->problem == ADMM_PROB_LASSO || e->problem == ADMM_PROB_QP_BOUNDED);
}

// one x-update (admm.m:501-511) from e->rhs into e->x, or into chunk partials for the fused consumer
static int x_update(admm_engine* e, const double** axsrc, int32_t* naxpart, int64_t* axld, const double** axt,
                    int32_t* axtri, bool leave_partials) {
  TimerScope ts(e, ADMM_K_XSOLVE);
  *axsrc = e->x;
  *naxpart = 1;
  *axld = 0;
  *axt = nullptr;
  *axtri = 0;
  if (e->xcb) {  // x = xminf(x, z, u, rho), fast ADMM: xminf(x, v, uhat, rho)   (admm.m:502, 506)
    const bool fastalg = e->last_opts.fast != ADMM_FAST_OFF;
    const double* zarg = e->bgen ? (fastalg ? e->vt : e->zt) : (fastalg ? e->v : e->z);
    if (e->xcb(e->xuser, e->x, zarg, fastalg ? e->uhat : e->u, e->last_opts.rho, e->xext, e->nA,
               static_cast<void*>(e->stream)) != 0)
      return fail(ADMM_E_INVALID, "the xminf callback reported a failure");
    if (e->a_identity) {
      *axsrc = e->xext;  // the fused kernel stores it into x (guarded by the device stop flag)
    } else {  // A = D: D*x follows; copy through a kernel that honours the stop flag
      launch_combine(e->xext, 1, 0, 1.0, nullptr, 0.0, nullptr, e->x, e->nA, e->ctrl, e->stream);
    }
    return ADMM_OK;
  }
  if (e->xsolve == ADMM_XSOLVE_CG) return cg_solve(e, e->a_identity ? e->rhs : e->g);
  switch (e->problem) {
    case ADMM_PROB_LASSO:
      if (!e->fat) {
        ADMM_TRY(factor_x_update(e, axsrc, naxpart, axld, axt, axtri, leave_partials));
      } else {
        // getProxOps.m:1204  x = y/rho - D'*(U\(L\(D*y)))/rho^2
        launch_gemv_n(e->planDN, e->D, e->rhs, e->partDN, e->ctrl, e->stream);
        launch_sum_partials(e->partDN, e->planDN.nchunk, e->planDN.ldy, e->m, e->tmpA, e->ctrl, e->stream);
        ADMM_TRY(solve_factor(e, e->tmpA, e->tmpB));
        launch_gemv_t(e->planDT, e->D, e->tmpB, nullptr, nullptr, 1, e->partDT, e->ctrl, e->stream);
        const double rho = e->last_opts.rho;
        launch_combine(e->partDT, e->planDT.nchunk, e->planDT.ldg, -1.0 / (rho * rho), e->rhs, 1.0 / rho, nullptr,
                       e->x, e->n, e->ctrl, e->stream);
      }
      break;
    case ADMM_PROB_QP_BOUNDED:  // planSq/partSq are shared with the objective GEMV; the x-update consumes them first
    case ADMM_PROB_MODEL:
      ADMM_TRY(factor_x_update(e, axsrc, naxpart, axld, axt, axtri, leave_partials));
      break;
    case ADMM_PROB_LINEARPROGRAM:
    case ADMM_PROB_QP_STANDARD:  // x = K*y + k0: the KKT solve of getProxOps.m:1363 / 1410, reduced once
      launch_gemv_t(e->planK, e->Kmat, e->rhs, nullptr, nullptr, 1, e->partK, e->ctrl, e->stream);
      launch_combine(e->partK, e->planK.nchunk, e->planK.ldg, 1.0, nullptr, 0.0, e->k0, e->x, e->n, e->ctrl,
                     e->stream);
      break;
    case ADMM_PROB_BASISPURSUIT:
      launch_gemv_t(e->planSq, e->Pmat, e->rhs, nullptr, nullptr, 1, e->partSq, e->ctrl, e->stream);
      launch_combine(e->partSq, e->planSq.nchunk, e->planSq.ldg, 1.0, nullptr, 0.0, e->q, e->x, e->n, e->ctrl,
                     e->stream);
      break;
    default:  // LAD / Huber / SVM: rhs already holds D'*(c + z - u) (row 0 of g)
      if (e->DplusT)  // row 0 of g is Dplus*(z - u) already (getProxOps.m:1067)
        launch_combine(e->g, 1, 0, 1.0, nullptr, 0.0, nullptr, e->x, e->nA, e->ctrl, e->stream);
      else
        ADMM_TRY(solve_factor(e, e->g, e->x));
      break;
  }
  return ADMM_OK;
}

// the cached-factor x-update shared by lasso (tall), bounded QP and the model problem
static int factor_x_update(admm_engine* e, const double** axsrc, int32_t* naxpart, int64_t* axld, const double** axt,
                           int32_t* axtri, bool leave_partials) {
  if (leave_partials && xsolve_tri1_partials(e)) {  // x = sum of the backward pass's rows from the diagonal tile on
    const TrsvPlan& t = e->xfac.trsv;
    launch_tri1_pair(t, e->rhs, nullptr, e->dfin, e->dfin && e->dfin_pending, e->ctrl, e->stream);
    *axsrc = t.tp1;
    *axt = t.tp1;
    *axtri = 1;
    *naxpart = t.ntile;
    *axld = t.ldp;
    return ADMM_OK;
  }
  if (leave_partials && xsolve_has_partials(e)) {  // x = sum of these rows, taken by prox_fin_kernel
    const SliceFactor& f = e->xfac;
    if (e->dfin && f.planSy.packed)
      launch_symv_lower_fin(f.planSy, f.Minv, e->rhs, e->syN, e->syT, *e->dfin, e->dfin_pending, e->ctrl, e->stream);
    else
      launch_symv_lower(f.planSy, f.Minv, f.ldM, e->rhs, e->syN, e->syT, e->x, e->ctrl, e->stream, 0, 1, false);
    *axsrc = e->syN;
    *axt = e->syT;
    *naxpart = f.planSy.ntile;
    *axld = f.planSy.ldp;
    return ADMM_OK;
  }
  return solve_factor(e, e->rhs, e->x);
}

int admm_engine_set_callbacks(admm_engine* e, admm_prox_callback xmin, void* xuser, admm_prox_callback zmin,
                              void* zuser, admm_obj_callback obj, void* objuser) {
  if (!e) return fail(ADMM_E_INVALID, "engine is NULL");
  const bool a1 = e->problem == ADMM_PROB_MODEL || e->problem == ADMM_PROB_QP_BOUNDED ||
                  e->problem == ADMM_PROB_BASISPURSUIT || e->problem == ADMM_PROB_LINEARPROGRAM ||
                  e->problem == ADMM_PROB_QP_STANDARD || (e->problem == ADMM_PROB_LASSO && !e->fat);
  const bool ad = e->problem == ADMM_PROB_LAD || e->problem == ADMM_PROB_HUBERFIT || e->problem == ADMM_PROB_LINEARSVM;
  if ((xmin || zmin || obj) && !((a1 || ad) && e->xsolve != ADMM_XSOLVE_CG))
    return fail(ADMM_E_UNSUPPORTED,
                "prox callbacks are supported for the A = 1 problems (model/generic, tall lasso, QP, LP, basis pursuit) "
                "and the A = D problems (LAD, Huber, linear SVM / unwrapped ADMM) with a cached-factor x-solve");
  if ((xmin || zmin || obj) && e->comm && comm_nranks(e->comm) > 1)
    return fail(ADMM_E_UNSUPPORTED, "prox callbacks are not supported on row-sharded engines");
  ADMM_HIP_TRY(hipSetDevice(e->device));
  const int64_t n2 = round_up(e->len, 2);
  if (!e->xext) ADMM_TRY(e->mem.alloc(&e->xext, round_up(e->nA, 2)));
  if (!e->zext) ADMM_TRY(e->mem.alloc(&e->zext, n2));
  if (!e->xh) ADMM_TRY(e->mem.alloc(&e->xh, n2));
  e->xcb = xmin;
  e->xuser = xuser;
  e->zcb = zmin;
  e->zuser = zuser;
  e->ocb = obj;
  e->ouser = objuser;
  return ADMM_OK;
}

}  // extern "C"
namespace admm {
admm_comm* engine_comm(admm_engine* e) { return e ? e->comm : nullptr; }
}  // namespace admm
extern "C" {

int admm_engine_set_hooks(admm_engine* e, admm_altu_callback altu, void* altu_user, admm_norms_callback norms,
                          void* norms_user) {
  if (!e) return fail(ADMM_E_INVALID, "engine is NULL");
  if (altu || norms) {
    if (e->problem == ADMM_PROB_LASSO_CONSENSUS || e->problem == ADMM_PROB_TOTALVARIATION || e->problem == ADMM_PROB_TV2D)
      return fail(ADMM_E_UNSUPPORTED, "caller-supplied options.altu / options.specialnorms: not for the consensus-lasso "
                                      "and total-variation loops (consensus lasso's own hooks are engine-native)");
    if (e->comm && comm_nranks(e->comm) > 1)
      return fail(ADMM_E_UNSUPPORTED, "caller-supplied options.altu / options.specialnorms are not supported on "
                                      "row-sharded engines");
    ADMM_HIP_TRY(hipSetDevice(e->device));
    const int64_t n2 = round_up(e->len, 2);
    if (!e->xh) ADMM_TRY(e->mem.alloc(&e->xh, n2));
    if (!e->hk_uold) {
      ADMM_TRY(e->mem.alloc(&e->hk_uold, n2));
      ADMM_TRY(e->mem.alloc(&e->hk_bz, n2));
      ADMM_TRY(e->mem.alloc(&e->hk_unew, n2));
      ADMM_TRY(e->mem.alloc(&e->hk_zero, n2));
      ADMM_TRY(e->mem.alloc(&e->hk_norms, 2));
      ADMM_HIP_TRY(hipMemsetAsync(e->hk_zero, 0, sizeof(double) * n2, e->stream));
    }
  }
  e->altucb = altu;
  e->altuuser = altu_user;
  e->normscb = norms;
  e->normsuser = norms_user;
  return ADMM_OK;
}

int admm_engine_set_operators(admm_engine* e, admm_operator_callback A, void* Auser, admm_operator_callback At,
                              void* Atuser) {
  if (!e) return fail(ADMM_E_INVALID, "engine is NULL");
  if (e->D || e->a_identity || e->xsolve != ADMM_XSOLVE_CALLBACK)
    return fail(ADMM_E_UNSUPPORTED, "operator callbacks belong to an engine created without a matrix "
                                    "(ADMM_PROB_LAD, ADMM_XSOLVE_CALLBACK, desc.D = NULL)");
  e->acb = A;
  e->auser = Auser;
  e->atcb = At;
  e->atuser = Atuser;
  return ADMM_OK;
}

int admm_engine_set_constraint_b(admm_engine* e, const double* B, int64_t ldB, int64_t nB, int32_t memkind,
                                 double scalar, admm_operator_callback Bop, void* Buser) {
  if (!e) return fail(ADMM_E_INVALID, "engine is NULL");
  const bool generic = (e->problem == ADMM_PROB_MODEL && !e->has_xfac && !e->has_zfac) ||
                       (e->problem == ADMM_PROB_LAD && e->xsolve == ADMM_XSOLVE_CALLBACK);
  if (!generic)
    return fail(ADMM_E_UNSUPPORTED, "a general B belongs to an engine whose two prox operators are both the caller's "
                                    "(the library's operators are written for B = -1)");
  if (e->comm && comm_nranks(e->comm) > 1) return fail(ADMM_E_UNSUPPORTED, "a general B is not supported on row-sharded engines");
  if (e->bgen && Bop && e->bcb && nB == e->nBz) {  // a fresh thunk for the same operator (host bindings re-create them per run)
    e->bcb = Bop;
    e->buser = Buser;
    return ADMM_OK;
  }
  if (e->bgen) return fail(ADMM_E_INVALID, "the constraint operator B of this engine is already set");
  const int64_t len = e->len;
  if (!Bop && !B) {  // a scalar: B = scalar*I, z has as many elements as the constraint
    if (nB != 0 && nB != len) return fail(ADMM_E_INVALID, "a scalar B needs nB equal to the constraint length m");
    nB = len;
  } else if (nB <= 0) {
    return fail(ADMM_E_INVALID, "nB (the length of z) must be positive");
  }
  if (B && !Bop && ldB < len) ldB = len;
  ADMM_HIP_TRY(hipSetDevice(e->device));
  const int64_t np = round_up(nB, 64) + 64;  // the streaming kernels read whole 16-byte pairs
  for (double** p : {&e->zt, &e->ztprev, &e->ztnew, &e->vt}) {
    ADMM_TRY(e->mem.alloc(p, np));
    ADMM_HIP_TRY(hipMemsetAsync(*p, 0, sizeof(double) * np, e->stream));
  }
  ADMM_TRY(e->mem.alloc(&e->btmp, round_up(len, 2)));
  if (Bop) {
    e->bcb = Bop;
    e->buser = Buser;
  } else if (B) {
    ADMM_TRY(upload_matrix(e->mem, &e->Bmat, &e->ldB, B, len, nB, ldB, memkind, e->stream));
    e->planBN = gemv_n_plan(len, nB, e->ldB);
    ADMM_TRY(e->mem.alloc(&e->partBN, e->planBN.part_elems()));
  }
  e->bscalar = scalar;
  e->nBz = nB;
  e->bgen = true;
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  return ADMM_OK;
}

// w[len] = -B*z[nBz]   (admm.m:536 Bz = B(z); the loop carries w so that B = -1 is the identity)
static int apply_b(admm_engine* e, const double* zin, double* wout) {
  if (e->bcb) {
    if (e->bcb(e->buser, zin, e->nBz, e->btmp, e->len, static_cast<void*>(e->stream)) != 0)
      return fail(ADMM_E_INVALID, "the B operator callback reported a failure");
    launch_combine(e->btmp, 1, 0, -1.0, nullptr, 0.0, nullptr, wout, e->len, e->ctrl, e->stream);
  } else if (e->Bmat) {
    launch_gemv_n(e->planBN, e->Bmat, zin, e->partBN, e->ctrl, e->stream);
    launch_combine(e->partBN, e->planBN.nchunk, e->planBN.ldy, -1.0, nullptr, 0.0, nullptr, wout, e->len, e->ctrl,
                   e->stream);
  } else {
    launch_combine(zin, 1, 0, -e->bscalar, nullptr, 0.0, nullptr, wout, e->len, e->ctrl, e->stream);
  }
  return ADMM_OK;
}

int admm_engine_run(admm_engine* e, const admm_options* opts, admm_run_summary* summary) {
  if (!e || !opts) return fail(ADMM_E_INVALID, "engine/options is NULL");
  if (opts->struct_size != static_cast<int32_t>(sizeof(admm_options)))
    return fail(ADMM_E_INVALID, "admm_options.struct_size mismatch (ABI version skew)");
  ADMM_HIP_TRY(hipSetDevice(e->device));
  admm_options o = *opts;
  if (!(o.rho > 0.0)) return fail(ADMM_E_INVALID, "options.rho must be positive");
  if (o.maxiters <= 0) o.maxiters = 1000;  // admm.m:334-339
  if (o.restart <= 0.0 || o.restart >= 1.0) o.restart = 0.999;  // admm.m:285-287
  if (o.fast != ADMM_FAST_OFF && o.fast != ADMM_FAST_WEAK && o.fast != ADMM_FAST_STRONG)
    return fail(ADMM_E_INVALID, "bad options.fast");
  if (o.rho != e->rho_factor && !o.stale_factor_ok && (e->F || e->has_zfac || e->Kmat) && e->problem != ADMM_PROB_LAD && e->problem != ADMM_PROB_HUBERFIT &&
      e->problem != ADMM_PROB_LINEARSVM)
    return fail(ADMM_E_INVALID, "options.rho differs from the rho the cached factor was built for");
  if (e->problem == ADMM_PROB_LASSO_CONSENSUS && o.rho != e->rho_factor)
    return fail(ADMM_E_INVALID, "options.rho differs from the rho the cached slice factors were built for");
  if (e->xsolve == ADMM_XSOLVE_CALLBACK && !e->xcb)
    return fail(ADMM_E_INVALID, "this engine was created with xsolve=callback: set the xminf callback before running");
  if (!e->a_identity && !e->D && e->axbuf && (!e->acb || !e->atcb))
    return fail(ADMM_E_INVALID, "this engine has no constraint matrix: set the A and At operator callbacks "
                                "(admm_engine_set_operators) before running");
  if (o.relax != 1.0 && (e->problem == ADMM_PROB_LINEARSVM))
    return fail(ADMM_E_INVALID,
                "relaxation with the linear SVM prox is a dimension error in the reference (getProxOps.m:1088)");
  if (o.relax != 1.0 && (e->problem == ADMM_PROB_LAD || e->problem == ADMM_PROB_HUBERFIT)) {
    // lad.m:124-126 switches to the userelax closures, which take Axhat directly: same fused formula
  }
  if (e->problem == ADMM_PROB_MODEL) {
    if (!e->has_xfac && !e->xcb)
      return fail(ADMM_E_INVALID, "no x-update: the model was created without PtP/Ptr and no xminf callback is set");
    if (!e->has_zfac && !e->zcb)
      return fail(ADMM_E_INVALID, "no z-update: the model was created without QtQ/Qts and no zming callback is set");
  }
  if (e->bgen && !(e->xcb && e->zcb))
    return fail(ADMM_E_INVALID, "an engine with a general B needs both the xminf and the zming callback");
  e->last_opts = o;
  const int alg = o.fast;  // 0, 1 (strong), 2 (weak)
  const bool use_h = o.convtest || o.stopcond == ADMM_STOP_HNORM || o.stopcond == ADMM_STOP_BOTH;
  const int64_t len = e->len, nA = e->nA;
  const int32_t N = o.maxiters;

  // ---- histories
  // (a run with the shape of the previous one keeps its buffers: hipFree synchronises the device and nine hipMallocs
  // cost ~0.1 ms -- 7 % of a 20-iteration run of the headline loop)
  const bool same_shape = e->hist_cap == N && e->hist_cap > 0 && e->hist_vectors == (o.record_history != 0) &&
                          e->hist_fast == (alg != 0) && e->hist_bgen == e->bgen && e->pnorm != nullptr;
  if (!same_shape) free_hist(e);
  e->hist_cap = N;
  e->hist_vectors = o.record_history != 0;
  e->hist_fast = alg != 0;
  e->hist_bgen = e->bgen;
  if (!same_shape && e->hist_vectors) {
    ADMM_TRY(hist_alloc(e, &e->xhist, static_cast<size_t>(nA) * N));
    ADMM_TRY(hist_alloc(e, &e->zhist, static_cast<size_t>(len) * N));
    ADMM_TRY(hist_alloc(e, &e->uhist, static_cast<size_t>(len) * N));
    if (alg != 0) {
      ADMM_TRY(hist_alloc(e, &e->vhist, static_cast<size_t>(len) * N));
      ADMM_TRY(hist_alloc(e, &e->uhathist, static_cast<size_t>(len) * N));
    }
    if (e->bgen) {  // results.zvals / vvals hold the caller's z (nB elements), not w = -B*z
      ADMM_TRY(hist_alloc(e, &e->zthist, static_cast<size_t>(e->nBz) * N));
      if (alg != 0) ADMM_TRY(hist_alloc(e, &e->vthist, static_cast<size_t>(e->nBz) * N));
    }
  }
  double** scal[] = {&e->pnorm, &e->dnorm, &e->perr, &e->derr, &e->objv, &e->hnorm, &e->avals, &e->dvals,
                     &e->restarted};
  for (double** p : scal)
    if (!same_shape) ADMM_TRY(hist_alloc(e, p, N));
  Ctrl c0{};
  c0.acurr = 1.0;
  c0.aprev = 1.0;
  c0.d = INFINITY;
  c0.dprev = INFINITY;
  *e->ctrl_host = c0;
  if (!o.x0 && !o.z0 && !o.u0 && !e->bgen) {
    // zero start (admm.m:252-254): the iterates, v = z, uhat = u (admm.m:269-270), the nine scalar histories and the
    // control block in ONE launch -- fifteen memset / memcpy calls cost ~60 us of host time per run, the price of
    // four iterations of a 20-iteration run of the headline loop
    RunInitArgs ia{};
    ia.x = e->x;
    ia.nA = nA;
    ia.z = e->z;
    ia.u = e->u;
    ia.v = e->v;
    ia.uhat = e->uhat;
    ia.len = len;
    int k = 0;
    for (double** p : scal) ia.scal[k++] = *p;
    ia.N = N;
    ia.ctrl = e->ctrl;
    ia.c0 = c0;
    launch_run_init(ia, e->stream);
  } else {
    for (double** p : scal) ADMM_HIP_TRY(hipMemsetAsync(*p, 0, sizeof(double) * N, e->stream));
    // ---- initial iterates (admm.m:252-254) and control block
    auto init_vec = [&](double* dst, const double* src, int64_t cnt) -> int {
      if (src) ADMM_HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * cnt, hipMemcpyHostToDevice, e->stream));
      else ADMM_HIP_TRY(hipMemsetAsync(dst, 0, sizeof(double) * cnt, e->stream));
      return ADMM_OK;
    };
    ADMM_TRY(init_vec(e->x, o.x0, nA));
    if (e->bgen) ADMM_TRY(init_vec(e->zt, o.z0, e->nBz));
    else ADMM_TRY(init_vec(e->z, o.z0, len));
    ADMM_TRY(init_vec(e->u, o.u0, len));
    ADMM_HIP_TRY(hipMemcpyAsync(e->v, e->z, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));     // admm.m:269
    ADMM_HIP_TRY(hipMemcpyAsync(e->uhat, e->u, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));  // admm.m:270
    ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl, e->ctrl_host, sizeof(Ctrl), hipMemcpyHostToDevice, e->stream));
  }
  if (e->bgen) {  // w0 = -B*z0; v starts as z (admm.m:269)
    ADMM_TRY(apply_b(e, e->zt, e->z));
    ADMM_HIP_TRY(hipMemcpyAsync(e->v, e->z, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));
    ADMM_HIP_TRY(hipMemcpyAsync(e->vt, e->zt, sizeof(double) * e->nBz, hipMemcpyDeviceToDevice, e->stream));
    ADMM_HIP_TRY(hipMemcpyAsync(e->ztprev, e->zt, sizeof(double) * e->nBz, hipMemcpyDeviceToDevice, e->stream));
  }
  if (o.x0 || o.z0 || o.u0 || e->bgen) ADMM_HIP_TRY(hipStreamSynchronize(e->stream));  // (host buffers were read)
  for (auto& t : e->timers) {
    t.used = 0;
    t.total_ms = 0.0;
    t.launches = 0;
  }
  if (e->cg_st) ADMM_HIP_TRY(hipMemsetAsync(e->cg_st, 0, sizeof(CgState), e->stream));

  // ---- objective wiring (solver-supplied handles: lasso.m:227, lad.m:148, huberfit.m:180,
  //      linearsvm.m:231-236, quadraticprogram.m:242, basispursuit.m:140)
  ProxArgs pa{};
  FinArgs fa{};
  fa.obj_scale_part = 0.0;
  pa.objz = OBJZ_NONE;
  pa.objx = OBJX_NONE;
  bool obj_lasso_gemv = false, obj_qp_gemv = false, obj_model_gemv = false;
  if (o.objevals && e->ocb) {  // options.obj is the caller's handle (admm.m:603-605)
    fa.obj_scale_part = 1.0;
  } else if (o.objevals) {
    switch (e->problem) {
      case ADMM_PROB_LASSO:
        if (!e->s)
          return fail(ADMM_E_INVALID, "objevals on a lasso engine created from args.Dts alone: the objective "
                                      "0.5*||D*x - s||^2 (lasso.m:227) needs s (or an objective callback)");
        obj_lasso_gemv = true;
        fa.obj_scale_part = 0.5;  // (the Gram form sets its own scale and constant per iteration, below)
        pa.objz = OBJZ_ABS;
        fa.obj_scale_z = e->lambda;
        break;
      case ADMM_PROB_LAD:
        pa.objz = OBJZ_ABS;
        fa.obj_scale_z = 1.0;
        break;
      case ADMM_PROB_HUBERFIT:
        pa.objz = OBJZ_HUBER;
        fa.obj_scale_z = 0.5;
        break;
      case ADMM_PROB_LINEARSVM:
        pa.objx = (e->loss == ADMM_LOSS_HINGE) ? OBJX_HINGE : OBJX_ZEROONE;  // linearsvm.m:231-237
        fa.obj_scale_x = e->C;
        fa.obj_half_xnorm = 0.5;
        break;
      case ADMM_PROB_QP_BOUNDED:
      case ADMM_PROB_QP_STANDARD:
        obj_qp_gemv = true;
        fa.obj_scale_part = 1.0;
        fa.obj_const = e->rconst;
        break;
      case ADMM_PROB_LINEARPROGRAM:  // b'*x   (linearprogram.m:178)
        pa.objx = OBJX_DOT;
        fa.obj_scale_x = 1.0;
        break;
      case ADMM_PROB_BASISPURSUIT:
        pa.objx = OBJX_ABS;
        fa.obj_scale_x = 1.0;
        break;
      case ADMM_PROB_MODEL:  // 1/2||P*x - r||^2 + 1/2||Q*z - s||^2   (model.m:133-134)
        if (!e->D || !e->D2)
          return fail(ADMM_E_INVALID, "objevals on the model problem needs the matrices P, Q and vectors r, s "
                                      "(or an objective callback)");
        obj_model_gemv = true;
        fa.obj_scale_part = 0.5;
        break;
      default:
        break;
    }
  }

  // ---- static parts of the kernel argument blocks
  pa.len = len;
  pa.c = e->c;
  pa.ell = e->ell;
  pa.lb = e->lb;
  pa.ub = e->ub;
  pa.z = e->z;
  pa.u = e->u;
  pa.uhat = e->uhat;
  pa.v = e->v;
  pa.zprev = e->zprev;
  pa.uprev = e->uprev;
  pa.dz = e->dz;
  pa.rhs = e->rhs;
  pa.rhs_add = e->rhs_add;
  pa.zhist = e->zhist;
  pa.uhist = e->uhist;
  pa.xhist = e->a_identity ? e->xhist : nullptr;
  pa.vhist = e->vhist;
  pa.uhathist = e->uhathist;
  pa.part = e->part;
  pa.rho = o.rho;
  pa.rho_solve = e->rho_factor;
  pa.relax = o.relax;
  const bool split_z = e->zcb != nullptr || e->problem == ADMM_PROB_MODEL;  // z is computed between two kernels
  pa.prox = split_z ? PROX_GIVEN : e->prox;
  pa.zgiven = e->zext;
  pa.rhs_kind = e->xcb ? RHS_NONE : e->rhs_kind;
  pa.alg = alg;
  pa.a_identity = e->a_identity ? 1 : 0;
  switch (e->prox) {
    case PROX_SOFT:
      pa.t = (e->problem == ADMM_PROB_LASSO) ? e->lambda / o.rho : 1.0 / o.rho;  // getProxOps.m:455 | 810, 142
      break;
    case PROX_HINGE:
      pa.t = e->C / o.rho;  // getProxOps.m:1096
      break;
    case PROX_01:
      pa.t = o.rho / e->C;  // getProxOps.m:1100
      break;
    default:
      pa.t = 0.0;
      break;
  }

  fa.len = len;
  fa.nA = nA;
  fa.part = e->part;
  fa.g = e->a_identity ? nullptr : e->g;
  fa.ldg = e->ldg;
  fa.x = e->a_identity ? nullptr : e->x;
  fa.xhist = e->a_identity ? nullptr : e->xhist;
  fa.cnorm = e->cnorm;
  fa.rho = o.rho;
  fa.rhoH = o.rho;
  fa.abstol = o.abstol;
  fa.reltol = o.reltol;
  fa.Hnormtol = o.Hnormtol;
  fa.convtol = o.convtol;
  fa.restart = o.restart;
  fa.dvaltol = o.dvaltol;
  fa.alg = alg;
  fa.a_identity = pa.a_identity;
  fa.nodualerror = o.nodualerror;
  fa.objevals = o.objevals;
  fa.use_h = use_h ? 1 : 0;
  fa.convtest = o.convtest;
  fa.stopcond = o.stopcond;
  fa.domaxiters = o.domaxiters;
  fa.maxiters = N;
  fa.pnorm = e->pnorm;
  fa.dnorm = e->dnorm;
  fa.perr = e->perr;
  fa.derr = e->derr;
  fa.objv = e->objv;
  fa.hnorm = e->hnorm;
  fa.avals = e->avals;
  fa.dvals = e->dvals;
  fa.restarted = e->restarted;
  fa.ctrl = e->ctrl;

  ExtrapArgs xa{};
  xa.len = len;
  xa.z = e->z;
  xa.u = e->u;
  xa.zprev = e->zprev;
  xa.uprev = e->uprev;
  xa.c = e->c;
  xa.v = e->v;
  xa.uhat = e->uhat;
  xa.rhs = e->rhs;
  xa.rhs_add = e->rhs_add;
  xa.vhist = e->vhist;
  xa.uhathist = e->uhathist;
  xa.rho = o.rho;
  xa.rhs_kind = e->xcb ? RHS_NONE : e->rhs_kind;

  RunState rs{o, alg, N, len, pa, fa, xa};
  auto trsv_ok = [&](int rc) -> int {  // the one-launch triangular solves report a lost tile through their plan
    if (rc != ADMM_OK) return rc;
    ADMM_TRY(trsv_check_error(e->xfac.trsv, e->stream));
    ADMM_TRY(trsv_check_error(e->zfac.trsv, e->stream));
    for (const ConsSlice& sl : e->cslices) ADMM_TRY(trsv_check_error(sl.fac.trsv, e->stream));
    return comm_check_error(e->comm, e->stream);
  };
  if (e->problem == ADMM_PROB_LASSO_CONSENSUS) return trsv_ok(run_consensus_lasso(e, rs, summary));
  if (e->problem == ADMM_PROB_TV2D) return run_total_variation_2d(e, rs, summary);
  if (e->problem == ADMM_PROB_TOTALVARIATION) return run_total_variation(e, rs, summary);

  const int nrhs_dual = o.nodualerror ? 1 : 3;
  const bool sharded = e->comm && comm_nranks(e->comm) > 1;
  const bool hooks = e->altucb != nullptr || e->normscb != nullptr;
  if (hooks && alg != 0)
    return fail(ADMM_E_UNSUPPORTED, "caller-supplied options.altu / options.specialnorms with fast ADMM: not supported "
                                    "(admm.m:614 overwrites fast ADMM's v with the norms: q5)");
  if (hooks && sharded) return fail(ADMM_E_UNSUPPORTED, "options.altu / options.specialnorms on a row-sharded engine");
  fa.norms_given = e->normscb ? e->hk_norms : nullptr;
  fa.len_global = e->len_global;
  int check_every = o.check_every > 0 ? o.check_every : (o.domaxiters ? 64 : 8);

  // ---- loop (admm.m:315 tic .. 756 toc)
  const auto tstart = std::chrono::steady_clock::now();
  // rhs of the first x-update from the initial iterates (zx = v = z0, ux = uhat = u0)
  launch_initial_rhs(len, e->rhs_kind, o.rho, e->z, e->u, e->c, e->rhs_add, e->rhs, e->stream);
  // D'*[t1, z - zprev, u] in one pass over D (getProxOps.m:1514; admm.m:624, 654).  With the caller's pseudo-inverse
  // (args.Dplus, linearsvm.m:185-186) row 0 is Dplus*t1 = the x-update itself (getProxOps.m:1067) and only the two
  // dual-residual products still stream D.
  int op_rc = ADMM_OK;
  const FinArgs* tp_fin = nullptr;  // set for ONE call: that call's partial-sum launch carries this finalize
  auto transposed_products = [&](int nrhs) {
    TimerScope ts(e, ADMM_K_GEMV_T);
    const FinArgs* fin = tp_fin;
    tp_fin = nullptr;
    auto sum_t = [&](int nr, double* gout) {
      if (fin) launch_sum_partials_t_fin(e->planDT, e->partDT, nr, gout, e->ldg, *fin, e->ctrl, e->stream);
      else launch_sum_partials_t(e->planDT, e->partDT, nr, gout, e->ldg, e->ctrl, e->stream);
      fin = nullptr;
    };
    if (e->atcb && !e->D) {  // options.At as a function handle (admm.m:165-167): one call per right-hand side
      const double* vecs[3] = {e->rhs, e->dz, e->u};
      for (int r = 0; r < nrhs && op_rc == ADMM_OK; ++r)
        if (e->atcb(e->atuser, vecs[r], len, e->g + r * e->ldg, nA, static_cast<void*>(e->stream)) != 0)
          op_rc = fail(ADMM_E_INVALID, "the At operator callback reported a failure");
    } else if (e->DplusT) {
      launch_gemv_t(e->planDT, e->DplusT, e->rhs, nullptr, nullptr, 1, e->partDT, e->ctrl, e->stream);
      sum_t(1, e->g);
      if (nrhs > 1) {
        launch_gemv_t(e->planDT, e->D, e->dz, e->u, nullptr, 2, e->partDT, e->ctrl, e->stream);
        sum_t(2, e->g + e->ldg);
      }
    } else {
      launch_gemv_t(e->planDT, e->D, e->rhs, e->dz, e->u, nrhs, e->partDT, e->ctrl, e->stream);
      sum_t(nrhs, e->g);
    }
  };
  // The unwrapped iteration with an explicit pseudo-inverse (linear SVM, unwrappedadmm.m:76-92) runs as TWO launches per
  // iteration (unwrapped.hip) when nothing outside the fused element update is asked for: plain ADMM, no recorded dual
  // residual (unwrappedadmm.m:92 sets nodualerror), library operators, library objective, one rank.
  const bool uw_fused = e->Dp && e->problem == ADMM_PROB_LINEARSVM && alg == 0 && o.relax == 1.0 && o.nodualerror &&
                        !sharded && !hooks && !e->xcb && !e->zcb && !e->ocb && std::getenv("ADMM_HIP_NO_UNWRAPPED_FUSED") == nullptr;
  UwArgs ua{};
  if (uw_fused) {
    ua.D = e->D;
    ua.ldD = e->ldD;
    ua.Dp = e->Dp;
    ua.ldP = e->ldDp;
    ua.m = e->m;
    ua.n = e->n;
    ua.R = e->uwR;
    ua.nblk = e->uwnblk;
    ua.G = e->uwG;
    ua.ldg = e->uwldg;
    ua.axpart = e->uwAx;
    ua.ldax = e->uwldax;
    ua.nchunk = e->uwnchunk;
    ua.xbuf = e->uwX;
    ua.ldx = e->uwldg;
    ua.iter = 0;
    ua.fin_pending = 0;
    ua.init = 1;  // partial rows of Dplus*(z0 - u0): what the first iteration sums into its x
    launch_uw_prox(ua, pa, e->ctrl, e->stream);
    ua.init = 0;
  } else if (!e->a_identity) {
    transposed_products(1);
    ADMM_TRY(op_rc);
    if (sharded) ADMM_TRY(comm_allreduce_device(e->comm, e->g, static_cast<size_t>(e->ldg), e->stream));
  }
  // One iteration = a fixed sequence of launches with iteration-independent arguments (the iteration
  // index lives in ctrl->iter), and iterations past a stop condition or past maxiters are no-ops on the
  // device, so a batch of iterations CAN be captured once into a hipGraph and replayed.  Measured on
  // MI355X / ROCm 7.x (profiles/svm_bench.py, profiles/trsv_graph_bench.py) replay is never faster than
  // eager launches on one stream -- SVM 6000x400: 23.0k vs 23.9k it/s -- so eager is the default and
  // ADMM_HIP_GRAPH=1 opts in.  Never used when collectives or host callbacks sit inside the iteration, for
  // the CG x-solve (which polls the device between inner iterations), or with event timing on
  // (hipEventElapsedTime rejects events recorded by graph nodes: "invalid resource handle").
  const int64_t heavy = std::max<int64_t>(e->m * e->n, e->nF * e->nF);
  const bool use_graph = std::getenv("ADMM_HIP_GRAPH") != nullptr && !sharded && e->profiling == 0 && !uw_fused &&
                         e->xsolve != ADMM_XSOLVE_CG && heavy <= (int64_t{32} << 20) && !e->xcb && !e->zcb && !e->ocb &&
                         !hooks;
  // A = I iterations whose finalize depends on nothing but the prox kernel's partial sums end in ONE launch
  // (prox_fin_kernel): no accelerated-ADMM decision, no split z-update, no objective kernels behind the prox, one rank
  // the lasso objective through the cached Gram matrix: always (obj_gram = 1), or once the calibration of the first
  // batch has shown it agrees with the literal D*x form to 1e-11 (obj_gram = 0; admm_engine.h)
  // (the form: 1/2*x'(y - rho*x) - x'D's + 1/2*s's with y the right-hand side x was solved from -- OBJX_SOLVE, summed by
  // the element update itself: no objective kernel at all, and the one-launch tail stays available with objevals = 1)
  const bool alt_qp = obj_qp_gemv && e->problem == ADMM_PROB_QP_BOUNDED && e->rhs_kind == RHS_RHO_MINUS_Q;
  const bool alt_ok = ((obj_lasso_gemv && e->rhs_kind == RHS_RHO_DTS) || alt_qp) && e->obj_alt && e->a_identity &&
                      !e->xcb && e->xsolve != ADMM_XSOLVE_CG;
  const double alt_const = alt_qp ? e->rconst : e->half_ssq;
  bool gram_now = alt_ok && (!e->obj_auto || e->obj_gram_ok);
  bool gram_calibrating = alt_ok && e->obj_auto && !e->obj_gram_ok && !e->obj_gram_bad;
  if (use_graph || sharded) gram_calibrating = false;  // (a captured batch cannot switch; shards would have to agree)
  if (gram_calibrating)
    ADMM_HIP_TRY(hipMemsetAsync(e->gobjpart + kMaxPartBlocks, 0, sizeof(double), e->stream));
  if (gram_now || gram_calibrating) pa.objx = alt_qp ? OBJX_SOLVE_QP : OBJX_SOLVE;
  if (gram_now) {
    fa.obj_scale_part = 0.0;
    fa.obj_scale_x = 1.0;
    fa.obj_const = alt_const;
  }
  // the cancellation bound of that form (finalize_device.h) is tracked whenever the engine chose it itself
  // (obj_gram = 0): past 1e-10 the run goes back to the literal pass at the next batch boundary, below
  const bool gram_guard = alt_ok && e->obj_auto && !use_graph && !sharded;
  fa.obj_track_bound = (gram_guard && (gram_now || gram_calibrating)) ? 1 : 0;
  bool obj_kernels =
      o.objevals && (((obj_lasso_gemv || obj_qp_gemv) && !gram_now) || obj_model_gemv || e->ocb);
  // ... and so do A = D iterations that record no dual residual (unwrappedadmm.m:92 sets nodualerror for the SVM):
  // without it the finalize logic needs none of the D' products that follow the prox kernel
  // (row-sharded A = I engines keep x, z, u replicated and exchange nothing per iteration unless the x-solve's tiles
  // are split over the ranks -- symv_apply's one all-reduce, before this tail: they run the same tail as one rank)
  bool fuse_tail = (e->a_identity || o.nodualerror) && alg != 2 && !split_z && (!sharded || e->a_identity) &&
                         !obj_kernels && !hooks &&
                         len <= int64_t{128} * kMaxPartBlocks && std::getenv("ADMM_HIP_NO_FUSED_TAIL") == nullptr;
  // With the packed lower-triangle x-solve in front of it, the finalize logic of an A = I iteration is deferred: the
  // element update stores its block partials and ends; the next iteration's x-solve carries the finalize in one extra
  // workgroup (symv_lower_fin_kernel), where its ~6 us of serial work overlap with 60 us of streaming, and the element
  // update after that starts with the decision in ctrl.  A batch's last iteration gets a stand-alone finalize.
  // (also with the x-solve's tiles split over the ranks: symv_apply hands the same passenger to its launch)
  const bool split_symv = sharded && e->sy_split && e->xfac.mode == ADMM_XSOLVE_INVERSE && e->xfac.Minv && !e->xcb &&
                          (e->problem == ADMM_PROB_LASSO || e->problem == ADMM_PROB_QP_BOUNDED) && !e->fat;
  bool defer_fin = fuse_tail && e->a_identity &&
                         (((xsolve_has_partials(e) || split_symv) && e->xfac.planSy.packed) || xsolve_tri1_partials(e)) && !use_graph &&
                         std::getenv("ADMM_HIP_NO_DEFERRED_FINALIZE") == nullptr;
  // A = D iterations without a dual residual (fuse_tail): the finalize logic leaves the element update's launch too and
  // runs as one extra workgroup of the partial-sum launch of D'*(c + z - u) that follows it (gemv.hip)
  bool defer_fin_ad = fuse_tail && !e->a_identity && e->D && !(e->atcb && !e->D) && !use_graph &&
                            std::getenv("ADMM_HIP_NO_DEFERRED_FINALIZE") == nullptr;
  FinArgs dff{};
  e->dfin = nullptr;
  e->dfin_pending = false;
  // A = D iterations without a dual residual on a tall, narrow D (config 3 at MNIST's full size): ONE pass over D per
  // iteration instead of two (unwrapped.hip: ad_onepass_kernel) -- x-solve, the pass (D*x, element update, the partial
  // rows of D'*(c + z - u)), their sum with the iteration's finalize as passenger.
  const bool onepass = fuse_tail && defer_fin_ad && !uw_fused && alg == 0 && o.relax == 1.0 && o.nodualerror && e->D &&
                       !e->DplusT && !sharded && !hooks && !e->xcb && !e->zcb && !e->ocb && !e->acb && !e->atcb && !e->bgen &&
                       !split_z && !use_graph && e->xsolve != ADMM_XSOLVE_CG && onepass_supported(e->m, e->n);
  OnePassArgs opa{};
  GemvTPlan op_plan = e->planDT;
  if (onepass) {
    const int nwg = onepass_workgroups(e->m);
    if (!e->opG) ADMM_TRY(e->mem.alloc(&e->opG, static_cast<size_t>(nwg) * static_cast<size_t>(e->ldg)));
    opa.D = e->D;
    opa.ldD = e->ldD;
    opa.m = e->m;
    opa.n = e->n;
    opa.x = e->x;
    opa.gpart = e->opG;
    opa.ldg = e->ldg;
    op_plan.nchunk = nwg;
    op_plan.ldg = e->ldg;
  }
  auto enqueue_iteration = [&]() -> int {
    if (onepass) {
      {
        TimerScope ts(e, ADMM_K_XSOLVE);
        ADMM_TRY(solve_factor(e, e->g, e->x));  // (row 0 of g: D'*(c + z - u) of the previous pass)
      }
      int nblk = 1;
      {
        TimerScope ts(e, ADMM_K_PROX);
        pa.axsrc = nullptr;
        pa.ax_t = nullptr;
        pa.ax_tri = 0;
        pa.x_out = nullptr;
        launch_ad_onepass(opa, pa, e->ctrl, &nblk, e->stream);
      }
      dff = prox_fin_args(pa, fa);
      dff.nblk = nblk;
      TimerScope ts(e, ADMM_K_GEMV_T);
      launch_sum_partials_t_fin(op_plan, e->opG, 1, e->g, e->ldg, dff, e->ctrl, e->stream);
      return ADMM_OK;
    }
    if (uw_fused) {
      TimerScope ts(e, ADMM_K_PROX);
      pa.axsrc = nullptr;
      pa.ax_t = nullptr;
      pa.x_out = nullptr;
      FinArgs fprev = fa;  // the previous iteration's finalize rides along with this iteration's first launch
      fprev.x = e->uwX + ((ua.iter + 1) & 1) * e->uwldg;
      launch_uw_ax(ua, fprev, e->ctrl, e->stream);
      launch_uw_prox(ua, pa, e->ctrl, e->stream);
      ua.iter += 1;
      ua.fin_pending = 1;
      return ADMM_OK;
    }
    {
      const double* axsrc;
      const double* axt;
      int32_t naxpart, axtri;
      int64_t axld;
      ADMM_TRY(x_update(e, &axsrc, &naxpart, &axld, &axt, &axtri, fuse_tail));
      if (!e->a_identity && !e->D) {  // Ax = A(x) with options.A a function handle (admm.m:117-120, 535)
        TimerScope ts(e, ADMM_K_GEMV_N);
        if (e->acb(e->auser, e->x, nA, e->axbuf, len, static_cast<void*>(e->stream)) != 0)
          return fail(ADMM_E_INVALID, "the A operator callback reported a failure");
        axsrc = e->axbuf;
        naxpart = 1;
        axld = 0;
      } else if (!e->a_identity) {  // Ax = D*x (admm.m:535), summed inside the prox kernel
        TimerScope ts(e, ADMM_K_GEMV_N);
        launch_gemv_n(e->planDN, e->D, e->x, e->partDN, e->ctrl, e->stream);
        axsrc = e->partDN;
        naxpart = e->planDN.nchunk;
        axld = e->planDN.ldy;
      }
      int nblk = 1;
      if (split_z) {  // z = zming(x or Axhat, z, u or uhat, rho) between the two halves of the fused kernel
        TimerScope ts(e, ADMM_K_PROX);
        PreZArgs za{};
        za.len = len;
        za.axsrc = axsrc;
        za.naxpart = naxpart;
        za.axld = axld;
        za.c = e->c;
        za.z = e->z;
        za.uo = (alg == 0) ? e->u : e->uhat;
        za.add = e->qz;
        za.xh = e->xh;
        za.rz = e->zcb ? nullptr : e->rz;
        za.rho = o.rho;
        za.relax = o.relax;
        launch_prez(za, e->ctrl, e->stream);
        if (e->zcb) {
          // admm.m:521-530: zming is called with x itself, or with the relaxed Axhat when relax != 1; with A = 1
          // the two have the same length (xh), with A = D the un-relaxed call passes the n-vector x
          const double* zarg = (e->a_identity || o.relax != 1.0) ? e->xh : e->x;
          if (e->bgen) {  // z = zming(., z, u, rho) in the caller's own space, then w = -B*z for the fused kernel
            if (e->zcb(e->zuser, zarg, e->zt, za.uo, o.rho, e->ztnew, e->nBz, static_cast<void*>(e->stream)) != 0)
              return fail(ADMM_E_INVALID, "the zming callback reported a failure");
            ADMM_TRY(apply_b(e, e->ztnew, e->zext));
            ZStateArgs zs{e->nBz, e->zt, e->ztprev, e->vt, e->ztnew, e->zthist, e->vthist, alg, 0};
            launch_zstate(zs, e->ctrl, e->stream);
          } else if (e->zcb(e->zuser, zarg, e->z, za.uo, o.rho, e->zext, len, static_cast<void*>(e->stream)) != 0)
            return fail(ADMM_E_INVALID, "the zming callback reported a failure");
        } else {  // zminModel: (QtQ + rho I) \ (Qts + rho*(x + u))   getProxOps.m:1012
          apply_slice_factor(e, e->zfac, e->rz, e->zext);
        }
      }
      fa.slots_reduced = nullptr;
      fa.objp_reduced = nullptr;
      fa.objpart = nullptr;
      fa.nobjpart = 0;
      {
        TimerScope ts(e, ADMM_K_PROX);
        pa.axsrc = axsrc;
        pa.ax_t = axt;
        pa.ax_tri = axtri;
        pa.naxpart = naxpart;
        pa.axld = axld;
        pa.x_out = e->a_identity ? e->x : nullptr;
        if (defer_fin) {  // the element update alone; the finalize rides along with the next x-solve
          launch_prox_fin(pa, fa, e->ctrl, &nblk, e->stream, true);
          dff = prox_fin_args(pa, fa);
          e->dfin = &dff;
          e->dfin_pending = true;
          return ADMM_OK;
        }
        if (fuse_tail && defer_fin_ad) {  // A = D without a dual residual: the finalize rides along with the partial
          launch_prox_fin(pa, fa, e->ctrl, &nblk, e->stream, true);  // sums of the next right-hand side, just below
          dff = prox_fin_args(pa, fa);
          tp_fin = &dff;
        } else if (fuse_tail) {  // z/u update + finalize in one launch
          launch_prox_fin(pa, fa, e->ctrl, &nblk, e->stream);
          if (e->a_identity) return ADMM_OK;  // the iteration ends here
        } else {
          if (e->altucb) {  // options.altu needs the old u and Ax (or the relaxed Axhat) as vectors: admm.m:553-559
            ADMM_HIP_TRY(hipMemcpyAsync(e->hk_uold, e->u, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));
            if (!split_z) {
              PreZArgs za{};
              za.len = len;
              za.axsrc = axsrc;
              za.naxpart = naxpart;
              za.axld = axld;
              za.c = e->c;
              za.z = e->z;
              za.uo = e->u;
              za.add = nullptr;
              za.xh = e->xh;
              za.rz = nullptr;
              za.rho = o.rho;
              za.relax = o.relax;
              launch_prez(za, e->ctrl, e->stream);
            }
          }
          launch_prox(pa, e->ctrl, &nblk, e->stream);
          if (e->altucb) {
            launch_negate(e->z, e->hk_bz, len, e->ctrl, e->stream);
            if (e->altucb(e->altuuser, e->hk_uold, e->xh, e->hk_bz, e->c ? e->c : e->hk_zero, len, e->hk_unew,
                          static_cast<void*>(e->stream)) != 0)
              return fail(ADMM_E_INVALID, "the altu callback reported a failure");
            UFixArgs ux{};
            ux.len = len;
            ux.unew = e->hk_unew;
            ux.uold = e->hk_uold;
            ux.z = e->z;
            ux.c = e->c;
            ux.rhs_add = e->rhs_add;
            ux.u = e->u;
            ux.uhist = e->uhist;
            ux.rhs = pa.rhs;
            ux.part = e->part;
            ux.nblk = nblk;
            ux.rhs_kind = pa.rhs_kind;
            ux.rho = o.rho;
            launch_ufix(ux, e->ctrl, e->stream);
          }
          if (e->normscb) {  // v = options.specialnorms(x, z, u, rho)   admm.m:612-616
            if (e->normscb(e->normsuser, e->x, nA, e->bgen ? e->zt : e->z, e->bgen ? e->nBz : len, e->u, len, o.rho,
                           e->hk_norms, static_cast<void*>(e->stream)) != 0)
              return fail(ADMM_E_INVALID, "the specialnorms callback reported a failure");
          }
        }
      }
      if (fuse_tail) {  // A = D: only the next x-update's right-hand side D'*(c + z - u) is left to do
        transposed_products(1);
        ADMM_TRY(op_rc);
        return ADMM_OK;
      }
      fa.nblk = nblk;
      const bool shard_rows = sharded && !e->a_identity;  // z, u and the residual sums are row-local
      if (alg == 2) {
        if (shard_rows) {  // the restart decision needs the global ||u-uhat||^2, ||z-v||^2 (admm.m:572-573)
          launch_pack_slots(e->part, nblk, e->red, e->ctrl, e->stream);
          ADMM_TRY(comm_allreduce_device(e->comm, e->red, 16, e->stream));
          fa.slots_reduced = e->red;
        }
        launch_fast_decide(fa, e->stream);
        launch_extrapolate(xa, e->ctrl, e->stream);
        if (e->bgen) {
          ZStateArgs zs{e->nBz, e->zt, e->ztprev, e->vt, nullptr, nullptr, e->vthist, alg, 1};
          launch_zstate(zs, e->ctrl, e->stream);
        }
      }
      if (!e->a_identity) {  // D'*[c+zx-ux, z-zprev, u]  (getProxOps.m:1514; admm.m:624, 654) in ONE pass
        transposed_products(nrhs_dual);
        ADMM_TRY(op_rc);
        if (shard_rows) {
          // ONE all-reduce per iteration: d = sum_g D_g'(...) (unwrappedadmm.m:135-137) for up to
          // three right-hand sides plus the 16 residual/objective partial sums (X3 + X6)
          double* slots = e->g + 3 * e->ldg;
          if (alg == 2) ADMM_HIP_TRY(hipMemcpyAsync(slots, e->red, 16 * sizeof(double), hipMemcpyDeviceToDevice,
                                                     e->stream));
          else launch_pack_slots(e->part, nblk, slots, e->ctrl, e->stream);
          if (alg == 2) {  // slots were already reduced: only the vectors travel
            ADMM_TRY(comm_allreduce_device(e->comm, e->g, static_cast<size_t>(3 * e->ldg), e->stream));
          } else {
            ADMM_TRY(comm_allreduce_device(e->comm, e->g, static_cast<size_t>(3 * e->ldg + 16), e->stream));
          }
          fa.slots_reduced = slots;
        }
      }
      fa.objpart = nullptr;
      fa.nobjpart = 0;
      if ((obj_lasso_gemv || obj_qp_gemv) && gram_now) {  // the element update's S_OBJX slot holds the data term already
        fa.obj_scale_part = 0.0;
        fa.obj_scale_x = 1.0;
        fa.obj_const = alt_const;
      } else if (obj_lasso_gemv) {  // 0.5*||D*x - s||^2  (lasso.m:227)
        TimerScope ts(e, ADMM_K_GEMV_N);
        int nob = 0;
        launch_gemv_n(e->planDN, e->D, e->x, e->partDN, e->ctrl, e->stream);
        launch_residual_sq(e->partDN, e->planDN.nchunk, e->planDN.ldy, e->s, e->m, e->objpart, &nob, e->ctrl,
                           e->stream);
        fa.obj_scale_part = 0.5;
        fa.obj_const = 0.0;
        fa.objpart = e->objpart;
        fa.nobjpart = nob;
        if (gram_calibrating)  // the solve-identity form beside it (slot partials of the element update): how far apart?
          launch_obj_compare(e->objpart, nob, 0.5, 0.0, e->part + static_cast<size_t>(S_OBJX) * kMaxPartBlocks, nblk, 1.0,
                             e->half_ssq, e->gobjpart + kMaxPartBlocks, e->ctrl, e->stream);
        if (sharded) {  // sum over the row shards of ||D_g*x - s_g||^2
          launch_pack_sum(e->objpart, nob, e->red + 16, e->ctrl, e->stream);
          ADMM_TRY(comm_allreduce_device(e->comm, e->red + 16, 1, e->stream));
          fa.objp_reduced = e->red + 16;
        }
      } else if (o.objevals && e->ocb) {  // objevals(i) = obj(x, z) with the caller's handle (admm.m:604)
        if (e->ocb(e->ouser, e->x, nA, e->bgen ? e->zt : e->z, e->bgen ? e->nBz : len, e->objpart,
                   static_cast<void*>(e->stream)) != 0)
          return fail(ADMM_E_INVALID, "the objective callback reported a failure");
        fa.objpart = e->objpart;
        fa.nobjpart = 1;
      } else if (obj_model_gemv) {
        TimerScope ts(e, ADMM_K_GEMV_N);
        int nob1 = 0, nob2 = 0;
        launch_gemv_n(e->planDN, e->D, e->x, e->partDN, e->ctrl, e->stream);
        launch_residual_sq(e->partDN, e->planDN.nchunk, e->planDN.ldy, e->ell, e->m, e->objpart, &nob1, e->ctrl,
                           e->stream);
        launch_gemv_n(e->planD2N, e->D2, e->z, e->partD2N, e->ctrl, e->stream);
        launch_residual_sq(e->partD2N, e->planD2N.nchunk, e->planD2N.ldy, e->s2, e->m2, e->objpart + nob1, &nob2,
                           e->ctrl, e->stream);
        fa.objpart = e->objpart;
        fa.nobjpart = nob1 + nob2;
      } else if (obj_qp_gemv) {  // 1/2 x'Px + q'x + r  (quadraticprogram.m:242)
        int nob = 0;
        const GemvTPlan& p = e->planSq;  // P is symmetric
        launch_gemv_t(p, e->Pmat, e->x, nullptr, nullptr, 1, e->partSq, e->ctrl, e->stream);
        launch_qp_objective(e->partSq, p.nchunk, p.ldg, e->x, e->q, e->n, e->objpart, &nob, e->ctrl, e->stream);
        fa.objpart = e->objpart;
        fa.nobjpart = nob;
        if (gram_calibrating)  // the right-hand-side form beside it
          launch_obj_compare(e->objpart, nob, 1.0, e->rconst, e->part + static_cast<size_t>(S_OBJX) * kMaxPartBlocks, nblk,
                             1.0, e->rconst, e->gobjpart + kMaxPartBlocks, e->ctrl, e->stream);
      }
      {
        TimerScope ts(e, ADMM_K_FINALIZE);
        launch_finalize(fa, e->stream);
      }
    }
    return ADMM_OK;
  };

  int32_t enq = 0;
  bool stopped = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  int32_t gbatch = 0;
  if (use_graph) {
    gbatch = (N < check_every) ? N : check_every;
    hipError_t ge = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
    int rc_cap = ADMM_OK;
    if (ge == hipSuccess) {
      for (int32_t b = 0; b < gbatch && rc_cap == ADMM_OK; ++b) rc_cap = enqueue_iteration();
      ge = hipStreamEndCapture(e->stream, &graph);
    }
    if (ge == hipSuccess && rc_cap == ADMM_OK) ge = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
    if (ge != hipSuccess || rc_cap != ADMM_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      return fail(ADMM_E_DEVICE, std::string("hipGraph capture of the iteration failed: ") + hipGetErrorString(ge));
    }
  }
  int loop_rc = ADMM_OK;
  bool polled = false;  // ctrl_host is what the device holds (nothing was enqueued since the last poll)
  while (enq < N && !stopped && loop_rc == ADMM_OK) {
    polled = false;
    int32_t batch = (N - enq < check_every) ? N - enq : check_every;
    if (gexec) {
      batch = gbatch;  // a full batch; iterations beyond maxiters are device-side no-ops
      if (hipGraphLaunch(gexec, e->stream) != hipSuccess) loop_rc = fail(ADMM_E_DEVICE, "hipGraphLaunch failed");
    } else {
      for (int32_t b = 0; b < batch && loop_rc == ADMM_OK; ++b) loop_rc = enqueue_iteration();
    }
    enq += batch;
    if (defer_fin && e->dfin_pending && loop_rc == ADMM_OK) {  // the batch's last iteration
      launch_finalize(dff, e->stream);
      e->dfin_pending = false;
    }
    if (uw_fused && ua.fin_pending && loop_rc == ADMM_OK) {  // the last enqueued iteration's finalize, on its own
      FinArgs flast = uw_fin_args(ua, fa);
      flast.x = e->uwX + ((ua.iter + 1) & 1) * e->uwldg;
      launch_finalize(flast, e->stream);
      ua.fin_pending = 0;
    }
    // the host polls after EVERY batch, domaxiters runs included: an unbounded run of launches without a host
    // sync (6000 for a 1000-iteration SVM run) overruns a buffer inside rocprofv3's counter-collection mode
    // (SIGSEGV in the launch path of the profiler, r2 record in DESIGN section 6); one 64-byte read-back per 64
    // iterations costs < 1 %
    if (loop_rc == ADMM_OK) {
      if (hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
          hipStreamSynchronize(e->stream) != hipSuccess)
        loop_rc = fail(ADMM_E_DEVICE, "polling the device control block failed");
      else {
        polled = true;
        if (e->ctrl_host->stop) stopped = true;
      }
    }
    // The right-hand-side form of the objective cancels terms of the size of 1/2*s's down to the data misfit: once
    // eps * |terms| / |objective| (the device keeps the run's maximum in ctrl) leaves 1e-10 -- a near-interpolating fit,
    // small lambda, little noise -- the engine goes back to the literal D*x pass (lasso.m:227), for the rest of this
    // run and for every later one.  Nothing is pending at a batch boundary, so the launch sequence may change here.
    if (gram_guard && gram_now && polled && e->ctrl_host->obj_bound > 1e-10) {
      e->obj_gram_ok = false;
      e->obj_gram_bad = true;
      gram_now = false;
      obj_kernels = true;
      fuse_tail = false;
      defer_fin = false;
      defer_fin_ad = false;
      fa.obj_track_bound = 0;
      pa.objx = OBJX_NONE;
      fa.obj_scale_x = 0.0;
      fa.obj_const = 0.0;
      fa.obj_scale_part = alt_qp ? 1.0 : 0.5;
      if (alt_qp) fa.obj_const = e->rconst;
    }
    if (polled && e->ctrl_host->obj_bound > e->obj_bound_seen) e->obj_bound_seen = e->ctrl_host->obj_bound;
    if (gram_calibrating && loop_rc == ADMM_OK) {  // the batch evaluated both forms of the lasso objective
      double disc = 1.0;
      if (hipMemcpy(&disc, e->gobjpart + kMaxPartBlocks, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        loop_rc = fail(ADMM_E_DEVICE, "reading the objective calibration failed");
      gram_calibrating = false;
      if (disc <= 1e-11 && !(e->ctrl_host->obj_bound > 1e-10)) {
        e->obj_gram_ok = true;
        gram_now = true;
      } else {
        e->obj_gram_bad = true;
      }
    }
  }
  e->dfin = nullptr;
  e->dfin_pending = false;
  if (gexec) (void)hipGraphExecDestroy(gexec);
  if (graph) (void)hipGraphDestroy(graph);
  ADMM_TRY(loop_rc);
  if (!polled) {
    ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  }
  {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(ADMM_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
  }
  if (uw_fused && e->ctrl_host->steps > 0)  // the x of the last completed iteration (double-buffered on its parity)
  {
    ADMM_HIP_TRY(hipMemcpyAsync(e->x, e->uwX + ((e->ctrl_host->steps - 1) & 1) * e->uwldg, sizeof(double) * e->n,
                                hipMemcpyDeviceToDevice, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  }
  const double runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count();
  if (e->profiling) collect_timers(e);
  for (auto& t : e->timers) t.used = 0;

  e->last = admm_run_summary{};
  e->last.steps = e->ctrl_host->steps;
  if (e->cg_st) {
    ADMM_HIP_TRY(hipMemcpy(e->cg_st_host, e->cg_st, sizeof(CgState), hipMemcpyDeviceToHost));
    e->cg_total_last = e->cg_st_host->total;
    e->cg_capped_last = e->cg_st_host->capped;
  }
  e->last.stopped_early = (e->ctrl_host->steps < N) ? 1 : 0;
  e->last.convtest_failed_at = e->ctrl_host->convfail;
  e->last.obj_gram_used = (o.objevals && gram_now) ? 1 : 0;
  e->last.runtime_s = runtime;
  e->last.objopt = NAN;
  if (o.objevals && e->last.steps > 0) {  // admm.m:752-754: obj(x,z) at the final iterates == last objevals entry
    double v = NAN;
    ADMM_HIP_TRY(hipMemcpy(&v, e->objv + (e->last.steps - 1), sizeof(double), hipMemcpyDeviceToHost));
    e->last.objopt = v;
  }
  e->has_run = true;
  if (summary) *summary = e->last;
  return trsv_ok(ADMM_OK);
}

}  // extern "C"
