// engine_run_tv.hip -- the iteration sequences of total variation (totalvariation.m:122-164; fused kernel, the
// unfused fast / relaxed form) and of its 2-D extension (spectral or CG x-update), split out of admm_engine_run.
#include <cstdlib>

#include "engine_internal.h"

namespace admm {

// 2-D TV: CG with the direction update fused into the stencil apply (10 instead of 14 vector passes and 3 instead
// of 5 launches per inner iteration); p ping-pongs between cg_p and cg_tmp
int cg_solve_tv2d(admm_engine* e, const double* y) {
  CgArgs a{};
  a.n = e->n;
  a.shift = 1.0;
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
  const double rho = e->last_opts.rho;
  // the default cap follows the conditioning: I + rho*D'D has its spectrum in [1, 1 + 8 rho), and CG reaches a relative
  // residual tol within 1/2*sqrt(cond)*ln(2/tol) steps -- 500 steps stop short of 1e-11 from rho = 190 on, and the
  // x-update silently lost digits there (2e-4 at rho = 5000; found by the solver sweep on images without a spectral path)
  int32_t maxit = e->cg_maxit;
  if (e->cg_maxit_auto) {
    const double need = 0.5 * std::sqrt(1.0 + 8.0 * rho) * std::log(2.0 / e->cg_tol) + 20.0;
    if (need > maxit) maxit = need < 2.0e5 ? static_cast<int32_t>(need) : 200000;
  }
  a.maxit = maxit;
  ADMM_HIP_TRY(hipMemsetAsync(&e->cg_st->iters, 0, 2 * sizeof(int32_t), e->stream));
  // r = y - (I + rho*D'D) x, p = r, rs, ||y||
  launch_tv2d_laplace(e->tv2_H, e->tv2_W, rho, e->x, e->cg_tmp, e->ctrl, e->stream);
  CgArgs a0 = a;
  a0.p = e->x;
  launch_cg_q(a0, e->cg_tmp, 1, 0, false, e->stream);
  launch_cg_init(a, e->stream);
  double* pbuf[2] = {e->cg_p, e->cg_tmp};
  int cur = 0;
  launch_tv2d_cg_pq(e->tv2_H, e->tv2_W, rho, a, pbuf[1], true, e->stream);  // q = A p, p.q (beta = 0)
  cur = 1;
  const int chunk = e->cg_chunk;  // as long as the previous solve: launches after convergence are no-ops
  for (int done_it = 0; done_it < maxit;) {
    const int k = (maxit - done_it < chunk) ? maxit - done_it : chunk;
    for (int c = 0; c < k; ++c) {
      a.p = pbuf[cur];
      launch_cg_update(a, e->stream);   // alpha, x += alpha p, r -= alpha q, (r.r)_new partials
      launch_tv2d_cg_pq(e->tv2_H, e->tv2_W, rho, a, pbuf[cur ^ 1], false, e->stream);
      cur ^= 1;
      launch_cg_advance(a, e->stream);  // rs <- (r.r)_new, convergence flag
    }
    done_it += k;
    ADMM_HIP_TRY(hipMemcpyAsync(e->cg_st_host, e->cg_st, sizeof(CgState), hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->cg_st_host->done || e->ctrl_host->stop) break;
  }
  e->cg_chunk = std::min(64, std::max(4, static_cast<int>(e->cg_st_host->iters) + 2));
  return ADMM_OK;
}

// 2-D TV x-update, direct: x = C2' diag(1/(1 + rho*(lamH_i + lamW_j))) C2 y with the 2-D DCT-II C2 (dct.h).
// Three passes over the image (6 N doubles of traffic): column DCT in place, the row transform + spectral division +
// inverse row transform on row PAIRS read at stride H (dct_rows_solve_strided_kernel), inverse column DCT.  Round 1
// transposed the image around the row pass (five passes, 10 N): 0.645 -> 0.590 ms per iteration at 4096^2 on the
// same box (ADMM_HIP_TV2D_TRANSPOSED=1 brings that form back; e->cg_r is its scratch).
// the Toeplitz row stage (dct.hip) covers this rho on this width
static bool tv2d_rows_green_ok(const admm_engine* e, double rho) {
  const int taps = tv2d_rows_green_taps(rho);
  return taps <= 96 && e->tv2_W >= 4 * taps;
}

// ... and it is the form dct_solve_tv2d takes by default (no environment override asks for a row transform)
static bool tv2d_rows_green_default(const admm_engine* e, double rho) {
  return tv2d_rows_green_ok(e, rho) && (!e->tv2_rows_dct || e->tv2_H % 2 != 0 ||
                                        (std::getenv("ADMM_HIP_TV2D_ROWS_DCT") == nullptr &&
                                                            std::getenv("ADMM_HIP_TV2D_TRANSPOSED") == nullptr));
}

// the exact tridiagonal row stage (dct.hip) is what is left when neither of the two applies
static bool tv2d_rows_thomas_default(const admm_engine* e, double rho) {
  return !tv2d_rows_green_default(e, rho) && !(e->tv2_rows_dct && e->tv2_H % 2 == 0) &&
         std::getenv("ADMM_HIP_TV2D_NO_THOMAS") == nullptr;
}

// fin != nullptr: the forward transform's launch carries the finalize logic of the previous iteration (when pending)
static int dct_solve_tv2d(admm_engine* e, double* y, const FinArgs* fin = nullptr, bool fin_pending = false) {
  TimerScope ts(e, ADMM_K_XSOLVE);
  const int64_t H = e->tv2_H, W = e->tv2_W;
  if (fin) launch_dct_cols_forward_fin(y, H, W, e->dctH, *fin, fin_pending, e->ctrl, e->stream);
  else launch_dct_cols_forward(y, H, W, e->dctH, e->ctrl, e->stream);            // along i, in place
  if (tv2d_rows_green_default(e, e->last_opts.rho)) {
    // default: no row transform at all -- the exact Toeplitz kernel of the row operator on the mirrored row (dct.hip)
    launch_tv2d_rows_green(y, e->x, H, W, e->last_opts.rho, e->dctH, e->ctrl, e->stream);
    launch_dct_cols_inverse(e->x, e->x, H, W, e->dctH, e->ctrl, e->stream);
    return ADMM_OK;
  }
  if (tv2d_rows_thomas_default(e, e->last_opts.rho)) {
    // neither the Toeplitz form (width / rho) nor a row transform (width not a power of two, or an odd height):
    // the row systems solved as they stand (dct.hip: tv2d_rows_thomas_kernel; factors in e->cg_p / e->cg_q, set up by
    // run_total_variation_2d)
    launch_tv2d_rows_thomas(y, e->x, H, W, e->last_opts.rho, e->cg_p, e->cg_q, e->ctrl, e->stream);
    launch_dct_cols_inverse(e->x, e->x, H, W, e->dctH, e->ctrl, e->stream);
    return ADMM_OK;
  }
  if (std::getenv("ADMM_HIP_TV2D_TRANSPOSED") == nullptr) {  // row transform on the untransposed image
    launch_dct_rows_solve_strided(y, H, W, e->last_opts.rho, e->dctH, e->dctW, e->ctrl, e->stream);
    launch_dct_cols_inverse(y, e->x, H, W, e->dctH, e->ctrl, e->stream);
    return ADMM_OK;
  }
  launch_transpose(y, e->cg_r, H, W, e->ctrl, e->stream);                        // -> W x H
  launch_dct_rows_solve(e->cg_r, H, W, e->last_opts.rho, e->dctH, e->dctW, e->ctrl, e->stream);
  launch_transpose(e->cg_r, y, W, H, e->ctrl, e->stream);                        // -> H x W
  launch_dct_cols_inverse(y, e->x, H, W, e->dctH, e->ctrl, e->stream);
  return ADMM_OK;
}

int run_total_variation_2d(admm_engine* e, RunState& rs, admm_run_summary* summary) {
  const admm_options& o = rs.o;
  const int alg = rs.alg;
  const int32_t N = rs.N;
  const int64_t len = rs.len;
  ProxArgs& pa = rs.pa;
  FinArgs& fa = rs.fa;
  ExtrapArgs& xa = rs.xa;
  (void)alg; (void)len; (void)pa; (void)xa;

  // the z-closure of totalvariation.m applies D to what it is handed (getProxOps.m:199); with D of size 2N x N the
  // relaxed Axhat (2N elements, admm.m:517) does not fit: a dimension error, as for the linear SVM (getProxOps.m:1088)
  if (o.relax != 1.0)
    return fail(ADMM_E_INVALID, "relaxation with the 2-D total-variation prox is a dimension error (D is 2N x N)");
  const int64_t Npix = e->tv2_H * e->tv2_W;
  // spectral x-update: the column DCT, then along the rows the Toeplitz stage (small rho, wide image), the row DCT
  // (width a power of two, even height: it works on row PAIRS) or the exact tridiagonal solve (anything else)
  const bool spectral = e->tv2_dct && ((e->tv2_rows_dct && e->tv2_H % 2 == 0) || tv2d_rows_green_ok(e, o.rho) ||
                                       tv2d_rows_thomas_default(e, o.rho));
  if (spectral && tv2d_rows_thomas_default(e, o.rho))  // the elimination factors of this run's rho
    launch_tv2d_rows_thomas_setup(e->tv2_H, e->tv2_W, o.rho, e->dctH, e->cg_p, e->cg_q, e->stream);
  if (e->z != e->tv_zA) {  // the initial iterates were written to e->z / e->u; make buffer A the current one
    ADMM_HIP_TRY(hipMemcpyAsync(e->tv_zA, e->z, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));
    ADMM_HIP_TRY(hipMemcpyAsync(e->tv_uA, e->u, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));
  }
  Tv2Args ta{};
  ta.H = e->tv2_H;
  ta.W = e->tv2_W;
  ta.rho = o.rho;
  ta.thresh = e->lambda / o.rho;
  ta.s = e->s;
  ta.x = e->x;
  ta.objevals = o.objevals;
  ta.xhist = e->xhist;
  ta.zhist = e->zhist;
  ta.uhist = e->uhist;
  ta.part = e->part;
  fa.g = nullptr;
  fa.x = nullptr;
  fa.xhist = nullptr;
  fa.dual_from_slots = 1;
  if (o.objevals) {  // 1/2*||x - s||^2 + lambda*||D x||_1
    fa.obj_scale_x = 0.5;
    fa.obj_scale_z = e->lambda;
  }
  (void)Npix;
  const int check_tv2 = o.check_every > 0 ? o.check_every : (o.domaxiters ? 64 : 8);
  const auto t0 = std::chrono::steady_clock::now();
  int32_t done = 0;
  bool stop_seen = false;
  if (alg != 0) {
    // Fast / accelerated ADMM (admm.m:267-298, 563-600), as for the 1-D solver: the x-update takes (v, uhat), the
    // generic fused prox kernel does the z/u update, extrapolation, histories and partial sums on the vector D*x,
    // the D' stencils of the dual residual come from dz = z - zprev and u.  z, u live in buffer A, updated in place.
    if (!e->dz) ADMM_TRY(e->mem.alloc(&e->dz, round_up(len, 2)));
    if (!e->tmpA) ADMM_TRY(e->mem.alloc(&e->tmpA, round_up(len, 2)));
    pa.z = e->tv_zA;
    pa.u = e->tv_uA;
    xa.z = e->tv_zA;
    xa.u = e->tv_uA;
    xa.rhs = nullptr;
    pa.dz = e->dz;
    pa.prox = PROX_SOFT;
    pa.t = e->lambda / o.rho;  // getProxOps.m:199
    pa.objz = OBJZ_NONE;
    pa.objx = OBJX_NONE;
    pa.x_out = nullptr;
    pa.xhist = nullptr;
    pa.rhs = nullptr;
    pa.rhs_kind = RHS_NONE;
    pa.a_identity = 0;
    pa.c = nullptr;
    fa.obj_scale_x = 0.0;
    fa.obj_scale_z = 0.0;
    fa.obj_scale_part = o.objevals ? 1.0 : 0.0;
    while (done < N && !stop_seen) {
      ta.z = e->v;  // x = xminf(x, v, uhat, rho)   admm.m:506
      ta.u = e->uhat;
      {
        TimerScope ts(e, ADMM_K_XSOLVE);
        launch_tv2d_rhs(ta, e->rhs, e->ctrl, e->stream);
      }
      if (spectral) ADMM_TRY(dct_solve_tv2d(e, e->rhs));
      else ADMM_TRY(cg_solve(e, e->rhs));
      int nob = 0, nblk = 1;
      launch_tv2d_dx(ta.H, ta.W, e->lambda, o.objevals, e->x, e->s, e->tmpA, e->objpart, &nob, e->xhist, e->ctrl,
                     e->stream);
      {
        TimerScope ts(e, ADMM_K_PROX);
        pa.axsrc = e->tmpA;
        pa.ax_t = nullptr;
        pa.naxpart = 1;
        pa.axld = 0;
        launch_prox(pa, e->ctrl, &nblk, e->stream);
      }
      fa.nblk = nblk;
      fa.slots_reduced = nullptr;
      fa.objp_reduced = nullptr;
      if (alg == 2) {
        launch_fast_decide(fa, e->stream);
        launch_extrapolate(xa, e->ctrl, e->stream);
      }
      launch_tv2d_dual_vec(ta.H, ta.W, e->dz, e->tv_uA, e->part, nblk, e->ctrl, e->stream);
      fa.objpart = o.objevals ? e->objpart : nullptr;
      fa.nobjpart = o.objevals ? nob : 0;
      {
        TimerScope ts(e, ADMM_K_FINALIZE);
        launch_finalize(fa, e->stream);
      }
      done += 1;
      if (!spectral || done % check_tv2 == 0 || done == N) {
        ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
        ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->ctrl_host->stop) stop_seen = true;
      }
    }
  }
  // alg == 0 carries the compact state v = z + u between iterations (tv2d.hip): iteration 0 reads z, u from buffer A
  // and writes v into V0 = tv_zB, iteration k reads V((k-1)&1) and writes V(k&1), V1 = tv_uB; the last executed
  // iteration's v is expanded into buffer A after the loop.
  // Deferred tail (spectral x-update): the finalize logic of iteration i rides in the first launch of iteration i + 1
  // (dct_cols_forward_fin_kernel); a batch's last iteration gets the stand-alone launch.  A stop it raises turns the
  // rest of iteration i + 1 into no-ops -- only the in-place transform of the right-hand side has run by then.
  const bool tv2_defer = spectral && std::getenv("ADMM_HIP_NO_DEFERRED_FINALIZE") == nullptr;
  // Default spectral form ("glued"): the fused pass hands its right-hand side to the forward column transform inside
  // one kernel (dct.hip: tv2d_fused_dct_kernel), so e->rhs holds the TRANSFORMED right-hand side from one iteration to
  // the next and an iteration is three launches: row stage (+ the previous iteration's finalize as a passenger) into the
  // scratch image e->cg_r, inverse column transform into x, fused pass + forward transform.  (The row stage writes a
  // scratch image, not x: it is the launch that carries the passenger, so it still runs when the passenger raises stop.)
  const bool tv2_glued = spectral && tv2d_rows_green_default(e, o.rho) && e->dctH.bm == 0 &&  // (power-of-two heights)
                         std::getenv("ADMM_HIP_TV2D_NO_GLUE") == nullptr;
  bool tv2_pending = false;
  while (alg == 0 && done < N && !stop_seen) {
    double* const vbuf[2] = {e->tv_zB, e->tv_uB};
    ta.z = done == 0 ? e->tv_zA : vbuf[(done - 1) & 1];
    ta.u = done == 0 ? e->tv_uA : nullptr;
    ta.zo = vbuf[done & 1];
    if (done == 0) {  // later right-hand sides come out of the fused z/u pass of the previous iteration
      TimerScope ts(e, ADMM_K_XSOLVE);
      launch_tv2d_rhs(ta, e->rhs, e->ctrl, e->stream);
      if (tv2_glued) launch_dct_cols_forward(e->rhs, ta.H, ta.W, e->dctH, e->ctrl, e->stream);
    }
    int nblk = 1;
    if (tv2_glued) {
      {
        TimerScope ts(e, ADMM_K_XSOLVE);
        launch_tv2d_rows_green(e->rhs, e->cg_r, ta.H, ta.W, o.rho, e->dctH, e->ctrl, e->stream, tv2_defer ? &fa : nullptr,
                               tv2_pending);
        launch_dct_cols_inverse(e->cg_r, e->x, ta.H, ta.W, e->dctH, e->ctrl, e->stream);
      }
      tv2_pending = false;
      TimerScope ts(e, ADMM_K_PROX);
      launch_tv2d_fused_dct(ta, done > 0, e->rhs, e->dctH, e->ctrl, &nblk, e->stream);
    } else {
      // (I + rho*D'D) x = s + rho*D'(z - u): spectral, or warm-started CG (polls the device)
      if (spectral) ADMM_TRY(dct_solve_tv2d(e, e->rhs, tv2_defer ? &fa : nullptr, tv2_pending));
      else ADMM_TRY(cg_solve(e, e->rhs));
      tv2_pending = false;
      TimerScope ts(e, ADMM_K_PROX);
      launch_tv2d_fused(ta, done > 0, e->rhs, e->ctrl, &nblk, e->stream);
    }
    fa.nblk = nblk;
    if (tv2_defer && (done + 1) % check_tv2 != 0 && done + 1 != N) {
      tv2_pending = true;  // finalized by the next iteration's first launch
    } else {
      TimerScope ts(e, ADMM_K_FINALIZE);
      launch_finalize(fa, e->stream);
    }
    done += 1;
    // the CG path synchronises inside every solve anyway; the spectral path runs check_tv2 iterations ahead
    // (everything enqueued after the stop flag is a no-op)
    if (!spectral || done % check_tv2 == 0 || done == N) {
      ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
      ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
      if (e->ctrl_host->stop) stop_seen = true;
    }
  }
  {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(ADMM_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
  }
  const double rt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (e->profiling) collect_timers(e);
  const int32_t steps = e->ctrl_host->steps;
  if (alg == 0 && steps > 0)  // z, u of the last executed iteration (fast ADMM updates buffer A in place)
    launch_tv2d_expand((steps - 1) & 1 ? e->tv_uB : e->tv_zB, ta.thresh, len, e->tv_zA, e->tv_uA, e->stream);
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  e->z = e->tv_zA;
  e->u = e->tv_uA;
  ADMM_HIP_TRY(hipMemcpy(e->cg_st_host, e->cg_st, sizeof(CgState), hipMemcpyDeviceToHost));
  e->cg_total_last = e->cg_st_host->total;
  e->cg_capped_last = e->cg_st_host->capped;
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

int run_total_variation(admm_engine* e, RunState& rs, admm_run_summary* summary) {
  const admm_options& o = rs.o;
  const int alg = rs.alg;
  const int32_t N = rs.N;
  const int64_t len = rs.len;
  ProxArgs& pa = rs.pa;
  FinArgs& fa = rs.fa;
  ExtrapArgs& xa = rs.xa;
  (void)alg; (void)len; (void)pa; (void)xa;

  std::vector<double> prefix;
  double bstar = 0.0;
  int halo = 0, elems = 0, tile = 0;
  ADMM_TRY(tv_plan(o.rho, e->n, &prefix, &bstar, &halo, &elems, &tile));
  if (prefix.size() > e->tv_bprefix_cap) {
    ADMM_TRY(e->mem.alloc(&e->tv_bprefix, prefix.size()));
    e->tv_bprefix_cap = prefix.size();
  }
  ADMM_HIP_TRY(hipMemcpyAsync(e->tv_bprefix, prefix.data(), sizeof(double) * prefix.size(), hipMemcpyHostToDevice,
                              e->stream));
  // the initial iterates were written to e->z / e->u; make buffer A the current one
  if (e->z != e->tv_zA) {
    ADMM_HIP_TRY(hipMemcpyAsync(e->tv_zA, e->z, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));
    ADMM_HIP_TRY(hipMemcpyAsync(e->tv_uA, e->u, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));
  }
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  TvArgs ta{};
  ta.n = e->n;
  ta.rho = o.rho;
  ta.thresh = e->lambda / o.rho;
  ta.s = e->s;
  ta.x = e->x;
  ta.y = e->tv_y;
  ta.bprefix = e->tv_bprefix;
  ta.nprefix = static_cast<int64_t>(prefix.size());
  ta.bstar = bstar;
  ta.halo = halo;
  ta.elems = elems;
  ta.tile = tile;
  ta.ftile = 256 * elems - 2 * halo - 4;
  ta.objevals = o.objevals;
  ta.xhist = e->xhist;
  ta.zhist = e->zhist;
  ta.uhist = e->uhist;
  ta.part = e->part;
  // one fused launch per iteration when the halo is small; the three-kernel form otherwise.  Default fused form: the
  // direct kernel (no y vector, 5 vector passes, tv.hip: tv_direct_kernel); ADMM_HIP_TV_SCAN=1 keeps the 7-pass form
  // whose sweeps are split over two launches
  const bool tv_fused = tv_fused_ok(ta) && std::getenv("ADMM_HIP_TV_UNFUSED") == nullptr;
  const bool tv_direct = tv_fused && tv_direct_ok(ta) && std::getenv("ADMM_HIP_TV_SCAN") == nullptr &&
                         std::getenv("ADMM_HIP_TV_ONE_LAUNCH") == nullptr && alg == 0 && o.relax == 1.0;
  if (tv_direct) {
    ta.margin = tv_direct_margin(ta);
    ta.ftile = 256 * kTvDirectE - 2 * ta.margin;
    const double rr = o.rho / bstar;
    ta.green = 1.0 / (bstar * (1.0 - rr * rr));
    ta.rpow[0] = rr;
    for (int k = 1; k < 8; ++k) ta.rpow[k] = ta.rpow[k - 1] * rr;
  }
  double* tv_part = nullptr;  // per-tile partials of the fused kernel (one column per tile)
  bool tv_one_launch = false;
  if (tv_fused) {
    const int64_t ntiles = ceil_div(e->n, ta.ftile);
    ta.part_stride = round_up(ntiles, 2);
    const int64_t ngroups = ceil_div(ntiles, kTvGroup);
    // One launch per iteration (tile partials -> group partials -> finalize inside the fused kernel, tv.hip) is
    // implemented and tested but NOT the default: at n = 4096^2 it measured 0.2419-0.2421 ms per iteration against
    // 0.2405 with the two small launches (tv_pack + finalize) behind the fused kernel, on the same box -- the drain
    // and the two arrival hops at the end of 8600 tiles cost what the two launches cost.  ADMM_HIP_TV_ONE_LAUNCH=1.
    tv_one_launch = ngroups <= kMaxPartBlocks && std::getenv("ADMM_HIP_TV_ONE_LAUNCH") != nullptr;
    const size_t extra = tv_one_launch ? static_cast<size_t>(S_COUNT) * kMaxPartBlocks + (ngroups + 2) / 2 + 1 : 0;
    // (two sets of tile partials: the deferred tail of iteration i reads its set while iteration i + 1 writes the other)
    const size_t want = 2 * static_cast<size_t>(S_COUNT) * ta.part_stride + extra;
    if (want > e->tv_part_cap) {  // (a per-run hipMalloc / hipFree pair costs more than 100 iterations at n = 2^24)
      ADMM_TRY(e->mem.alloc(&e->tv_part, want));
      e->tv_part_cap = want;
    }
    tv_part = e->tv_part;
    ta.part = tv_part;
    if (tv_one_launch) {
      ta.gpart = tv_part + 2 * S_COUNT * ta.part_stride;
      ta.gcount = reinterpret_cast<int32_t*>(ta.gpart + static_cast<size_t>(S_COUNT) * kMaxPartBlocks);
      ta.ngroups = static_cast<int32_t>(ngroups);
      ADMM_HIP_TRY(hipMemsetAsync(ta.gcount, 0, sizeof(int32_t) * (ngroups + 1), e->stream));
    }
  }
  fa.g = nullptr;
  fa.x = nullptr;
  fa.xhist = nullptr;
  fa.dual_from_slots = 1;
  if (o.objevals) {  // totalvariation.m:134-135
    fa.obj_scale_x = 0.5;
    fa.obj_scale_z = e->lambda;
  }
  const int check_tv = o.check_every > 0 ? o.check_every : (o.domaxiters ? 64 : 8);
  const bool tv_relaxed = o.relax != 1.0;
  if (alg != 0 || tv_relaxed) {
    // Fast / accelerated ADMM (admm.m:267-298, 563-600): the x-update takes (v, uhat), the generic fused prox
    // kernel does the z/u update, extrapolation, histories and partial sums on the vector D*x, and the D'
    // stencils of the dual residual come from dz = z - zprev and u.  z, u are updated in place here.
    // Over-relaxation (admm.m:515-532) takes the same unfused route: the reference's z-closure applies D to the
    // relaxed Axhat it is handed (getProxOps.m:199), so z comes from launch_tv_relax_z and the generic kernel
    // does everything else with z given (PROX_GIVEN).
    if (tv_relaxed && !e->zext) ADMM_TRY(e->mem.alloc(&e->zext, round_up(len, 2)));
    if (!e->dz) ADMM_TRY(e->mem.alloc(&e->dz, round_up(len, 2)));
    if (!e->tmpA) ADMM_TRY(e->mem.alloc(&e->tmpA, round_up(len, 2)));
    pa.dz = e->dz;
    pa.t = e->lambda / o.rho;  // getProxOps.m:199
    pa.objz = OBJZ_NONE;
    pa.objx = OBJX_NONE;
    pa.x_out = nullptr;
    fa.obj_scale_x = 0.0;
    fa.obj_scale_z = 0.0;
    fa.obj_scale_part = o.objevals ? 1.0 : 0.0;
    ta.part = e->part;
    const auto t0f = std::chrono::steady_clock::now();
    int32_t donef = 0;
    bool stopf = false;
    while (donef < N && !stopf) {
      const int32_t batch = (N - donef < check_tv) ? N - donef : check_tv;
      for (int32_t b = 0; b < batch; ++b) {
        ta.z = alg ? e->v : e->z;  // x = xminf(x, v, uhat, rho)   admm.m:506 (plain ADMM: z, u)
        ta.u = alg ? e->uhat : e->u;
        ta.y = e->tv_y;
        {
          TimerScope ts(e, ADMM_K_XSOLVE);
          launch_tv_sweep(ta, false, e->ctrl, e->stream);
          launch_tv_sweep(ta, true, e->ctrl, e->stream);
        }
        int nob = 0, nblk = 1;
        launch_tv_dx(e->x, e->s, e->n, e->lambda, o.objevals, e->tmpA, e->objpart, &nob, e->ctrl, e->stream);
        {
          TimerScope ts(e, ADMM_K_PROX);
          pa.axsrc = e->tmpA;
          pa.naxpart = 1;
          pa.axld = 0;
          if (tv_relaxed) {
            // admm.m:517-523: Axhat from z_prev; zming(Axhat, z, u | uhat, rho) -- fast ADMM hands it uhat
            launch_tv_relax_z(e->tmpA, e->z, alg ? e->uhat : e->u, e->n, o.relax, pa.t, e->zext, e->ctrl, e->stream);
            pa.prox = PROX_GIVEN;
            pa.zgiven = e->zext;
          }
          launch_prox(pa, e->ctrl, &nblk, e->stream);
        }
        fa.nblk = nblk;
        fa.slots_reduced = nullptr;
        fa.objp_reduced = nullptr;
        if (alg == 2) {
          launch_fast_decide(fa, e->stream);
          launch_extrapolate(xa, e->ctrl, e->stream);
        }
        launch_tv_dual(e->dz, e->u, e->n, e->part, nblk, e->ctrl, e->stream);
        fa.objpart = o.objevals ? e->objpart : nullptr;
        fa.nobjpart = o.objevals ? nob : 0;
        {
          TimerScope ts(e, ADMM_K_FINALIZE);
          launch_finalize(fa, e->stream);
        }
      }
      donef += batch;
      {  // poll after every batch (see engine_run.hip)
        ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
        ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->ctrl_host->stop) stopf = true;
      }
    }
    ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    {
      hipError_t le = hipGetLastError();
      if (le != hipSuccess) return fail(ADMM_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
    }
    if (e->profiling) collect_timers(e);
    const int32_t stepsf = e->ctrl_host->steps;
    e->last = admm_run_summary{};
    e->last.steps = stepsf;
    e->last.stopped_early = (stepsf < N) ? 1 : 0;
    e->last.convtest_failed_at = e->ctrl_host->convfail;
    e->last.runtime_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0f).count();
    e->last.objopt = NAN;
    if (o.objevals && stepsf > 0) {
      double v = NAN;
      ADMM_HIP_TRY(hipMemcpy(&v, e->objv + (stepsf - 1), sizeof(double), hipMemcpyDeviceToHost));
      e->last.objopt = v;
    }
    e->has_run = true;
    if (summary) *summary = e->last;
    return ADMM_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  int32_t done = 0;
  bool stop_seen = false;
  // x only leaves the fused kernel when its history is recorded; otherwise one backward sweep after the loop
  // rebuilds the final x from the forward-sweep vector the last executed iteration read (still intact: every
  // launch after the stop flag is a no-op, and an iteration writes the OTHER y buffer)
  ta.skip_x = (tv_fused && !e->xhist) ? 1 : 0;
  if (tv_direct) ta.skip_x = 0;  // (x leaves the direct kernel only as a history column)
  if (tv_fused && !tv_direct) {  // the forward sweep of iteration 0; every later one is produced by the fused kernel
    TimerScope ts(e, ADMM_K_XSOLVE);
    ta.z = e->tv_zA;
    ta.u = e->tv_uA;
    ta.y = e->tv_y;
    launch_tv_sweep(ta, false, e->ctrl, e->stream);
  }
  // Deferred tail (default for the fused kernel): the tile-partial sums and the finalize logic of iteration i are done by
  // one extra workgroup of iteration i + 1's launch, hidden behind its tiles; a batch's last iteration gets the two
  // small launches.  A stop raised by that workgroup makes iteration i + 2 a no-op; iteration i + 1 has run
  // speculatively into the OTHER ping-pong buffers, and the final z, u, x are picked by the device's step count.
  const bool tv_deferred = tv_fused && !tv_one_launch && std::getenv("ADMM_HIP_NO_DEFERRED_FINALIZE") == nullptr;
  // ... and the forward-sweep vector iteration i read must survive iteration i + 1 (the final x is rebuilt from it when
  // no history holds x): three y buffers in rotation instead of two.  The direct kernel has no y and carries the
  // compact state v = z + u (tv.hip): iteration 0 reads z, u from buffer A, iteration k reads v from vbuf[(k-1) % 3]
  // and writes vbuf[k % 3] -- three buffers, so that the speculative iteration behind a stop overwrites neither the
  // last executed iteration's output nor its input (the final x is recomputed from the z, u it read).
  if (tv_deferred && !tv_direct && !e->tv_y3) ADMM_TRY(e->mem.alloc(&e->tv_y3, round_up(e->n, 2)));
  if (tv_direct && !e->tv_v3) ADMM_TRY(e->mem.alloc(&e->tv_v3, round_up(e->n, 2)));
  double* const tv_y3 = e->tv_y3;
  double* const tv_v3 = e->tv_v3;
  double* const ybuf[3] = {e->tv_y, e->tv_y2, tv_y3};
  double* const vbuf[3] = {e->tv_zB, e->tv_uB, tv_v3};
  const int64_t tv_ntiles = tv_fused ? ceil_div(e->n, ta.ftile) : 0;
  const int64_t tv_pset = static_cast<int64_t>(S_COUNT) * ta.part_stride;
  bool tv_pending = false;
  while (done < N && !stop_seen) {
    const int32_t batch = (N - done < check_tv) ? N - done : check_tv;
    for (int32_t b = 0; b < batch; ++b) {
      const bool a_cur = ((done + b) & 1) == 0;  // iteration k reads A when k is even
      ta.z = a_cur ? e->tv_zA : e->tv_zB;
      ta.u = a_cur ? e->tv_uA : e->tv_uB;
      ta.zo = a_cur ? e->tv_zB : e->tv_zA;
      ta.uo = a_cur ? e->tv_uB : e->tv_uA;
      int nblk = 1;
      if (tv_fused) {
        TimerScope ts(e, ADMM_K_XSOLVE);
        ta.yin = a_cur ? e->tv_y : e->tv_y2;
        ta.yout = a_cur ? e->tv_y2 : e->tv_y;
        fa.nblk = nblk;
        if (tv_direct) {
          const int64_t k = done + b;
          ta.state_in = k > 0 ? 1 : 0;
          ta.z = k > 0 ? vbuf[(k - 1) % 3] : e->tv_zA;
          ta.u = k > 0 ? nullptr : e->tv_uA;
          ta.zo = vbuf[k % 3];
          ta.uo = nullptr;
          ta.deferred = tv_deferred ? 1 : 0;
          ta.iter_host = k;
          ta.part = tv_part + (k & 1) * tv_pset;
          ta.prev_part = tv_part + ((k + 1) & 1) * tv_pset;
          ta.prev_ntiles = static_cast<int32_t>(tv_ntiles);
          ta.slots16 = e->red;
          ta.fin_pending = tv_pending ? 1 : 0;
          fa.slots_reduced = e->red;
          launch_tv_direct(ta, fa, e->ctrl, e->stream);
          if (tv_deferred) {
            tv_pending = true;
            continue;
          }
          launch_tv_pack(ta.part, ta.part_stride, static_cast<int32_t>(tv_ntiles), e->red, e->ctrl, e->stream);
        } else if (tv_deferred) {
          const int64_t k = done + b;
          ta.yin = ybuf[k % 3];
          ta.yout = ybuf[(k + 1) % 3];
          ta.deferred = 1;
          ta.iter_host = k;
          ta.part = tv_part + (k & 1) * tv_pset;
          ta.prev_part = tv_part + ((k + 1) & 1) * tv_pset;
          ta.prev_ntiles = static_cast<int32_t>(tv_ntiles);
          ta.slots16 = e->red;
          ta.fin_pending = tv_pending ? 1 : 0;
          fa.slots_reduced = e->red;
          launch_tv_fused(ta, fa, e->red, e->ctrl, e->stream);
          tv_pending = true;
          continue;
        }
        if (!tv_direct) launch_tv_fused(ta, fa, e->red, e->ctrl, e->stream);
        if (tv_one_launch) continue;  // the launch ended the iteration itself
        fa.slots_reduced = e->red;
      } else {
        {
          TimerScope ts(e, ADMM_K_XSOLVE);
          launch_tv_sweep(ta, false, e->ctrl, e->stream);
          launch_tv_sweep(ta, true, e->ctrl, e->stream);
        }
        TimerScope ts(e, ADMM_K_PROX);
        launch_tv_prox(ta, e->ctrl, &nblk, e->stream);
      }
      fa.nblk = nblk;
      {
        TimerScope ts(e, ADMM_K_FINALIZE);
        launch_finalize(fa, e->stream);
      }
    }
    done += batch;
    if (tv_deferred && tv_pending) {  // the batch's last iteration: its tail as two small launches
      launch_tv_pack(tv_part + ((done - 1) & 1) * tv_pset, ta.part_stride, static_cast<int32_t>(tv_ntiles), e->red, e->ctrl,
                     e->stream);
      fa.slots_reduced = e->red;
      fa.nblk = 1;
      launch_finalize(fa, e->stream);
      tv_pending = false;
    }
    {  // poll after every batch (see engine_run.hip)
      ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
      ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
      if (e->ctrl_host->stop) stop_seen = true;
    }
  }
  ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(ADMM_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
  }
  if (e->profiling) collect_timers(e);
  const int32_t steps = e->ctrl_host->steps;
  if (tv_deferred && !tv_direct && !ta.skip_x && e->xhist && steps > 0)  // x was overwritten by the speculative iteration after a stop
    ADMM_HIP_TRY(hipMemcpyAsync(e->x, e->xhist + static_cast<int64_t>(steps - 1) * e->n, sizeof(double) * e->n,
                                hipMemcpyDeviceToDevice, e->stream));
  // iterations executed on the device decide which ping-pong buffer holds the final z, u
  e->z = (steps & 1) ? e->tv_zB : e->tv_zA;
  e->u = (steps & 1) ? e->tv_uB : e->tv_uA;
  if (tv_direct) {
    if (!e->xhist && steps > 0) {  // x of the last executed iteration, from the z, u it read: two stand-alone sweeps
      if (steps > 1) launch_tv2d_expand(vbuf[(steps - 2) % 3], ta.thresh, e->n, e->tv_zA, e->tv_uA, e->stream);
      ta.z = e->tv_zA;
      ta.u = e->tv_uA;
      ta.y = e->tv_y;
      ta.x = e->x;
      ta.xhist = nullptr;
      launch_tv_sweep(ta, false, e->ctrl_idle, e->stream);  // the loop's own flag says "stopped" by now
      launch_tv_sweep(ta, true, e->ctrl_idle, e->stream);
    } else if (e->xhist && steps > 0) {
      ADMM_HIP_TRY(hipMemcpyAsync(e->x, e->xhist + static_cast<int64_t>(steps - 1) * e->n, sizeof(double) * e->n,
                                  hipMemcpyDeviceToDevice, e->stream));
    }
    // z, u of the last executed iteration out of its compact state, into buffer A
    if (steps > 0) launch_tv2d_expand(vbuf[(steps - 1) % 3], ta.thresh, e->n, e->tv_zA, e->tv_uA, e->stream);
    e->z = e->tv_zA;
    e->u = e->tv_uA;
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  }
  if (ta.skip_x && steps > 0) {
    ta.y = tv_deferred ? ybuf[(steps - 1) % 3] : (((steps - 1) & 1) ? e->tv_y2 : e->tv_y);
    launch_tv_sweep(ta, true, e->ctrl_idle, e->stream);  // the loop's own flag says "stopped" by now
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  }
  const double rt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
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
