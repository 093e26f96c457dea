// engine.hip -- C ABI (include/admm_engine.h), part 1: create() = the reference solver's one-time setup +
// getproxops (lasso.m:160-192, lad.m:129-137, huberfit.m:161-169, linearsvm.m:183-217 + unwrappedadmm.m:76-92,
// quadraticprogram.m:210-232, basispursuit.m:116-127), result fetch, timers, destroy.  The loop itself
// (admm.m:252-767) is engine_run.hip.
#include "engine_internal.h"
#include "loop_kernels.h"

namespace admm {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

int upload(DevMem& mem, double** dst, const double* src, size_t elems, int memkind, hipStream_t stream) {
  ADMM_TRY(mem.alloc(dst, elems));
  ADMM_HIP_TRY(hipMemcpyAsync(*dst, src, elems * sizeof(double),
                              memkind == ADMM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
  return ADMM_OK;
}

// copy a column-major m x n matrix into a zero-padded device buffer with leading dimension ld
int upload_matrix(DevMem& mem, double** dst, int64_t* ld_out, const double* src, int64_t rows, int64_t cols,
                  int64_t ld_src, int memkind, hipStream_t stream) {
  // 128-byte aligned columns always; 4 KiB aligned columns for tall matrices: measured +5 % on the
  // gemv_n stream (dev/gemv_real.hip, ld 100000 vs 98304) for 0.35 % more memory
  const int64_t ld = rows >= 8192 ? round_up(rows, 512) : round_up(rows, 16);
  ADMM_TRY(mem.alloc(dst, static_cast<size_t>(ld) * cols));
  if (ld != rows) ADMM_HIP_TRY(hipMemsetAsync(*dst, 0, sizeof(double) * ld * cols, stream));
  ADMM_HIP_TRY(hipMemcpy2DAsync(*dst, ld * sizeof(double), src, ld_src * sizeof(double), rows * sizeof(double), cols,
                                memkind == ADMM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                stream));
  *ld_out = ld;
  return ADMM_OK;
}

void free_hist(admm_engine* e) {
  for (void* p : e->hist_ptrs) (void)hipFree(p);
  e->hist_ptrs.clear();
  e->xhist = e->zhist = e->uhist = e->vhist = e->uhathist = e->zthist = e->vthist = nullptr;
  e->pnorm = e->dnorm = e->perr = e->derr = e->objv = e->hnorm = e->avals = e->dvals = e->restarted = nullptr;
  e->hist_cap = 0;
}

int hist_alloc(admm_engine* e, double** out, size_t elems) {
  void* p = nullptr;
  if (elems == 0) elems = 1;
  hipError_t err = hipMalloc(&p, elems * sizeof(double));
  if (err != hipSuccess)
    return fail(ADMM_E_DEVICE, std::string("hipMalloc(history ") + std::to_string(elems * sizeof(double)) +
                                   " B): " + hipGetErrorString(err));
  e->hist_ptrs.push_back(p);
  *out = static_cast<double*>(p);
  return ADMM_OK;
}

void collect_timers(admm_engine* e) {
  for (int w = 0; w < ADMM_K_COUNT; ++w) {
    KTimer& t = e->timers[w];
    t.total_ms = 0.0;
    t.launches = 0;
    for (size_t k = 0; k + 1 < t.used; k += 2) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, t.ev[k], t.ev[k + 1]) == hipSuccess) {
        t.total_ms += ms;
        t.launches += 1;
      }
    }
    t.used = 0;
  }
}

// ---- factor setup ------------------------------------------------------------------
static void mem_free_one(DevMem& mem, void* p) {
  if (!p) return;
  for (auto& q : mem.ptrs)
    if (q == p) {
      (void)hipFree(p);
      q = nullptr;
      return;
    }
}

// partial-sum buffers of the lower-triangle kernel, shared by every factor of the engine (same n)
static int ensure_sy_buffers(admm_engine* e, const SymvPlan& plan) {
  const size_t need = plan.npart_elems();
  if (e->syN && e->sy_elems >= need) return ADMM_OK;
  mem_free_one(e->mem, e->syN);
  mem_free_one(e->mem, e->syT);
  e->syN = e->syT = nullptr;
  ADMM_TRY(e->mem.alloc(&e->syN, need));
  ADMM_TRY(e->mem.alloc(&e->syT, plan.tpart_elems()));
  e->sy_elems = need;
  // zero once: with the tiles split over ranks, the slots of foreign tiles are never written
  ADMM_HIP_TRY(hipMemsetAsync(e->syN, 0, sizeof(double) * need, e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(e->syT, 0, sizeof(double) * plan.tpart_elems(), e->stream));
  return ADMM_OK;
}

// The padded column-major inverse of a factor -> tile-packed storage (symv.hip): half the memory, and the layout the
// lower-triangle kernel streams fastest.  Small factors (symv_small_kernel reads whole columns) stay as they are.
static int pack_inverse(admm_engine* e, SliceFactor& f) {
  if (f.n < kSymvHalfMin || f.planSy.packed) return ADMM_OK;
  double* P = nullptr;
  ADMM_TRY(e->mem.alloc(&P, symv_packed_elems(f.planSy)));
  launch_symv_pack(f.planSy, f.Minv, f.ldM, P, e->stream);
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  mem_free_one(e->mem, f.Minv);
  f.Minv = P;
  f.ldM = 0;
  f.planSy.packed = true;
  return ADMM_OK;
}

// f.Minv (tile-padded, zeros outside n x n) = X' X with X = inv(L): the explicit inverse of L L'; pack = false keeps
// the padded column-major square (full symmetric storage) for callers that multiply with it as a dense matrix
static int build_explicit_inverse(admm_engine* e, SliceFactor& f, bool pack = true) {
  const int64_t n = f.n, ld = f.ld;
  double* X = nullptr;
  ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&X), sizeof(double) * ld * n));
  int rc = trtri_lower_from_diag(f.F, n, ld, f.dinv, X, ld, e->stream);
  f.planSy = symv_plan(n);
  f.ldM = f.planSy.npad;  // (4 / 8 / 16 KiB-aligned columns were measured: no effect on the lower-triangle kernel)
  if (rc == ADMM_OK) rc = e->mem.alloc(&f.Minv, static_cast<size_t>(f.ldM) * f.planSy.npad);
  if (rc == ADMM_OK) {
    (void)hipMemsetAsync(f.Minv, 0, sizeof(double) * f.ldM * f.planSy.npad, e->stream);
    launch_gemm(1, 0, n, n, n, 1.0, X, ld, X, ld, 0.0, f.Minv, f.ldM, true, e->stream);
    launch_symmetrize_lower(f.Minv, n, f.ldM, e->stream);
  }
  (void)hipStreamSynchronize(e->stream);
  (void)hipFree(X);
  ADMM_TRY(rc);
  if (pack) ADMM_TRY(pack_inverse(e, f));
  if (n >= kSymvHalfMin) ADMM_TRY(ensure_sy_buffers(e, f.planSy));
  return ADMM_OK;
}

static void apply_inverse(admm_engine* e, const SliceFactor& f, const double* y, double* out, const Ctrl* ctrl) {
  if (f.n >= kSymvHalfMin) launch_symv_lower(f.planSy, f.Minv, f.ldM, y, e->syN, e->syT, out, ctrl, e->stream);
  else launch_symv_small(f.Minv, f.n, f.ldM, y, out, ctrl, e->stream);
}

// the probe itself: two ways of applying inv(L L') on (L L') x = L (L' x0) with a known x0 (max-norm errors relative to
// ||x0||_inf), then on a second, unstructured right-hand side where they are compared with each other
template <class ApplyA, class ApplyB>
static int probe_two_forms(admm_engine* e, const SliceFactor& f, ApplyA&& apply_a, ApplyB&& apply_b, double* err_a,
                           double* err_b, double* diff_ab) {
  const int64_t n = f.n, n2 = round_up(n, 2);
  std::vector<double> x0(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i) x0[i] = 1.0 + 0.5 * std::sin(1.0 + 0.7 * static_cast<double>(i));
  double* buf = nullptr;
  ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&buf), sizeof(double) * 5 * n2));
  double *dx0 = buf, *dt = buf + n2, *dy = buf + 2 * n2, *dxi = buf + 3 * n2, *dxt = buf + 4 * n2;
  int rc = ADMM_OK;
  std::vector<double> xi(static_cast<size_t>(n)), xt(static_cast<size_t>(n));
  auto run = [&]() -> int {
    ADMM_HIP_TRY(hipMemcpyAsync(dx0, x0.data(), sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
    launch_llt_apply(f.F, n, f.ld, dx0, dt, dy, e->stream);  // y = L (L' x0)
    apply_a(dy, dxi);
    apply_b(dy, dxt);
    ADMM_HIP_TRY(hipMemcpyAsync(xi.data(), dxi, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipMemcpyAsync(xt.data(), dxt, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    return trsv_check_error(f.trsv, e->stream);
  };
  rc = run();
  double ei = 0.0, et = 0.0;
  if (rc == ADMM_OK) {
    for (int64_t i = 0; i < n; ++i) {
      const double a = std::fabs(xi[i] - x0[i]), b = std::fabs(xt[i] - x0[i]);
      ei = (a > ei || a != a) ? a : ei;
      et = (b > et || b != b) ? b : et;
    }
    // second right-hand side: signs and sizes with no structure (its solution is dominated by the small singular
    // directions); no exact answer, so the two forms are compared with each other
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (int64_t i = 0; i < n; ++i) {
      lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
      const double u01 = static_cast<double>(lcg >> 11) * (1.0 / 9007199254740992.0);
      x0[i] = (u01 < 0.5 ? -1.0 : 1.0) * (0.5 + u01);
    }
    auto run2 = [&]() -> int {
      ADMM_HIP_TRY(hipMemcpyAsync(dy, x0.data(), sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
      apply_a(dy, dxi);
      apply_b(dy, dxt);
      ADMM_HIP_TRY(hipMemcpyAsync(xi.data(), dxi, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
      ADMM_HIP_TRY(hipMemcpyAsync(xt.data(), dxt, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
      ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
      return ADMM_OK;
    };
    rc = run2();
  }
  (void)hipFree(buf);
  ADMM_TRY(rc);
  double dmax = 0.0, xmax = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double d = std::fabs(xi[i] - xt[i]), a = std::fabs(xt[i]);
    dmax = (d > dmax || d != d) ? d : dmax;
    xmax = a > xmax ? a : xmax;
  }
  *err_a = ei / 1.5;  // relative to ||x0||_inf
  *err_b = et / 1.5;
  *diff_ab = xmax > 0.0 ? dmax / xmax : dmax;
  return ADMM_OK;
}

// Which form applies inv(L L')?  The explicit inverse is one bandwidth-bound pass but its forward error grows like
// cond(LL')^1.5 * eps; the blocked triangular solves stay at the cond(LL') * eps of a backward-stable solve.  The
// engine does not guess from a condition estimate: it builds both, solves  (L L') x = L (L' x0)  for a known x0 with
// each, compares them on a second, unstructured right-hand side, and keeps the explicit inverse only while it is
// as accurate as the triangular solves (within 2x / 4x) or below 1e-9 (three orders under the 1e-6 parity bar);
// otherwise the triangular solves run, also when `inverse` was requested explicitly.
static int probe_and_choose(admm_engine* e, SliceFactor& f) {
  ADMM_TRY(probe_two_forms(
      e, f, [&](const double* y, double* x) { apply_inverse(e, f, y, x, nullptr); },
      [&](const double* y, double* x) { launch_trsv_pair(f.trsv, y, x, nullptr, e->stream); }, &f.err_inv, &f.err_trsv,
      &f.probe_diff));
  f.probed = true;
  // keep the one-pass form while it is as good as the backward-stable solves (or simply good enough)
  const bool ok1 = f.err_inv <= std::max(1e-9, 2.0 * f.err_trsv);
  const bool ok2 = f.probe_diff <= std::max(1e-9, 4.0 * f.err_trsv);
  f.mode = (ok1 && ok2) ? ADMM_XSOLVE_INVERSE : ADMM_XSOLVE_TRSV;
  return ADMM_OK;
}

// The triangular solves themselves have two forms (trsv.hip): blocked substitution over K = ceil(n / 2048) coarse blocks
// with pre-inverted DIAGONAL blocks (2K + 1 dependent launches per pair), and the one-block form -- the whole factor
// pre-inverted, X = inv(L), two passes and two launches per pair, same 8 n(n+1) bytes.  X costs forward error
// eps * cond(L) per pass where a diagonal block costs eps * cond(L_kk).  Same rule as above: the blocked form (already
// built: f.trsv) is the yardstick, and the one-block form replaces it only while it is as accurate on both probe
// systems (within 4x / 8x) or below 1e-11 -- five orders under the parity bar.  ADMM_TRSV_FORM=one|blocked forces it.
static int choose_trsv_form(admm_engine* e, SliceFactor& f) {
  f.err_trsv_one = NAN;
  const char* forced = std::getenv("ADMM_TRSV_FORM");
  if (trsv_resolve_form(f.n, kTrsvOne) != kTrsvOne) return ADMM_OK;
  if (f.n < kSymvHalfMin && !(forced && forced[0] == 'o')) return ADMM_OK;  // a few tiles: the steps are cheap
  double* work1 = nullptr;
  ADMM_TRY(e->mem.alloc(&work1, trsv_plan_elems(f.n, kTrsvOne)));
  TrsvPlan one{};
  int rc = trsv_build(f.F, f.n, f.ld, f.dinv, work1, &one, e->stream, kTrsvOne);
  double e1 = NAN, eb = NAN, diff = NAN;
  if (rc == ADMM_OK)
    rc = probe_two_forms(
        e, f, [&](const double* y, double* x) { launch_trsv_pair(one, y, x, nullptr, e->stream); },
        [&](const double* y, double* x) { launch_trsv_pair(f.trsv, y, x, nullptr, e->stream); }, &e1, &eb, &diff);
  if (rc != ADMM_OK) {
    mem_free_one(e->mem, work1);
    return rc;
  }
  f.err_trsv_one = e1;
  if (!f.probed) f.err_trsv = eb;
  const bool ok = (forced && forced[0] == 'o') ||
                  (e1 <= std::max(1e-11, 4.0 * eb) && diff <= std::max(1e-11, 8.0 * eb));
  if (ok) {
    mem_free_one(e->mem, f.work);
    f.work = work1;
    f.trsv = one;
  } else {
    mem_free_one(e->mem, work1);
  }
  return ADMM_OK;
}

// W (n x n, ld) holds an SPD matrix in its lower triangle (or receives the caller's factor Lgiven) -> f: the
// Cholesky factor in place, its inverted 64 x 64 diagonal blocks, and ONE way of applying inv(L L'):
// want = TRSV: blocked triangular solves; INVERSE / AUTO: the explicit inverse if the probe allows it.
int build_slice_factor(admm_engine* e, SliceFactor& f, double* W, int64_t n, int64_t ld, int want, const double* Lgiven,
                       int memkind) {
  f.F = W;
  f.n = n;
  f.ld = ld;
  const int64_t nblk = ceil_div(n, 64);
  ADMM_TRY(e->mem.alloc(&f.dinv, static_cast<size_t>(nblk) * 64 * 64));
  if (Lgiven) {
    ADMM_HIP_TRY(hipMemsetAsync(W, 0, sizeof(double) * ld * n, e->stream));
    ADMM_HIP_TRY(hipMemcpy2DAsync(W, ld * sizeof(double), Lgiven, n * sizeof(double), n * sizeof(double), n,
                                  memkind == ADMM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                  e->stream));
    launch_trtri_diag(W, n, ld, f.dinv, e->stream);
  } else {
    double* infod = nullptr;
    ADMM_TRY(e->mem.alloc(&infod, 1));
    int32_t* info_dev = reinterpret_cast<int32_t*>(infod);
    ADMM_TRY(cholesky_lower(W, n, ld, info_dev, f.dinv, e->stream));
    int32_t info = 0;
    ADMM_HIP_TRY(hipMemcpyAsync(&info, info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    mem_free_one(e->mem, infod);
    if (info != 0) {
      f.chol_info = info;
      return fail(ADMM_E_NUMERIC, "Cholesky failed: matrix must be positive definite (pivot " + std::to_string(info) + ")");
    }
  }
  {  // cond(L L') >= (max L_ii / min L_ii)^2: a lower bound that costs one strided copy of the diagonal
    std::vector<double> dg(static_cast<size_t>(n));
    ADMM_HIP_TRY(hipMemcpy2DAsync(dg.data(), sizeof(double), W, (ld + 1) * sizeof(double), sizeof(double), n,
                                  hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    double lo = INFINITY, hi = 0.0;
    for (double v : dg) {
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
    }
    f.diag_min = lo;
    f.diag_max = hi;
    f.cond_diag = (lo > 0.0) ? (hi / lo) * (hi / lo) : INFINITY;
  }
  // AUTO: the explicit inverse at every size while the probe below allows it -- one launch per x-update where the
  // triangular solves take three or more, which is all a small problem pays for (256 x 64, the testers' default:
  // 20 -> 13 us per iteration)
  if (want == ADMM_XSOLVE_AUTO) want = ADMM_XSOLVE_INVERSE;
  // the blocked triangular solves are built in every case: they are the fallback and the probe's yardstick
  ADMM_TRY(e->mem.alloc(&f.work, trsv_plan_elems(n)));
  ADMM_TRY(trsv_build(W, n, ld, f.dinv, f.work, &f.trsv, e->stream));
  f.mode = ADMM_XSOLVE_TRSV;
  if (want == ADMM_XSOLVE_INVERSE) {
    ADMM_TRY(build_explicit_inverse(e, f));
    ADMM_TRY(probe_and_choose(e, f));
    if (f.mode == ADMM_XSOLVE_INVERSE) {  // the triangular-solve plan was only the yardstick
      mem_free_one(e->mem, f.work);
      f.work = nullptr;
      f.trsv = TrsvPlan{};
    } else {
      mem_free_one(e->mem, f.Minv);
      f.Minv = nullptr;
    }
  }
  if (f.mode == ADMM_XSOLVE_TRSV) ADMM_TRY(choose_trsv_form(e, f));
  return ADMM_OK;
}

static void release_slice_factor(admm_engine* e, SliceFactor& f) {
  mem_free_one(e->mem, f.Minv);
  mem_free_one(e->mem, f.work);
  mem_free_one(e->mem, f.dinv);
  f = SliceFactor{};
}

void apply_slice_factor(admm_engine* e, const SliceFactor& f, const double* y, double* out) {
  if (f.mode == ADMM_XSOLVE_INVERSE) apply_inverse(e, f, y, out, e->ctrl);
  else launch_trsv_pair(f.trsv, y, out, e->ctrl, e->stream);
}

// engine-level multi-GPU refinement of the explicit-inverse x-solve: split the tiles over the ranks?
static int decide_sy_split(admm_engine* e) {
  const SliceFactor& f = e->xfac;
  e->sy_split = false;
  const int nr = e->comm ? comm_nranks(e->comm) : 1;
  if (nr <= 1 || f.mode != ADMM_XSOLVE_INVERSE || f.n < kSymvHalfMin) return ADMM_OK;
  // Splitting the tiles over the ranks removes t*(1 - 1/N) of streaming time per x-solve and adds one all-reduce
  // of n doubles.  Decide with the latency this communicator actually has (measured here, the mean over the
  // ranks so that every rank takes the same decision): t = 4*npad^2 bytes at ~5.5 TB/s.
  double* probe = e->syN;  // any device buffer of >= n + 1 doubles
  for (int k = 0; k < 3; ++k) ADMM_TRY(comm_allreduce_device(e->comm, probe, static_cast<size_t>(f.n), e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  const auto c0 = std::chrono::steady_clock::now();
  const int reps = 10;
  for (int k = 0; k < reps; ++k) ADMM_TRY(comm_allreduce_device(e->comm, probe, static_cast<size_t>(f.n), e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  double lat_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count() * 1e6 / reps;
  ADMM_HIP_TRY(hipMemcpyAsync(probe, &lat_us, sizeof(double), hipMemcpyHostToDevice, e->stream));
  ADMM_TRY(comm_allreduce_device(e->comm, probe, 1, e->stream));
  ADMM_HIP_TRY(hipMemcpyAsync(&lat_us, probe, sizeof(double), hipMemcpyDeviceToHost, e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  lat_us /= nr;
  // (6.4e6 bytes/us: the packed kernel with the split cache policy; the unsplit path also keeps the deferred finalize
  // and needs no reduce launch for its partial rows: ~8 us the split has to win back on top of the collective)
  const double t_us = 4.0 * static_cast<double>(f.planSy.npad) * static_cast<double>(f.planSy.npad) / 6.4e6;
  e->sy_split = t_us * (1.0 - 1.0 / nr) > 1.15 * lat_us + 8.0;
  if (const char* fl = std::getenv("ADMM_HIP_XSPLIT")) e->sy_split = fl[0] == '1';  // tests force either form
  // the probe (and the latency measurement) left partial sums behind; with the tiles split over ranks the slots
  // of foreign tiles are never written again and must read as zero
  ADMM_HIP_TRY(hipMemsetAsync(e->syN, 0, sizeof(double) * f.planSy.npart_elems(), e->stream));
  ADMM_HIP_TRY(hipMemsetAsync(e->syT, 0, sizeof(double) * f.planSy.tpart_elems(), e->stream));
  return ADMM_OK;
}

// pinv(D) as an explicit n x m matrix for the two-launch unwrapped iteration (unwrapped.hip): the caller's args.Dplus as
// it is, or inv(D'D)*D' from the factor create() has just built -- its explicit inverse when that is the form in use
// (or the pseudo-inverse of a rank-deficient D'D), a temporary one for the small factors that default to triangular
// solves.  Not built when the accuracy probe rejected the explicit inverse, when the caller asked for another x-solve
// form, on sharded engines, or for sizes the kernel does not cover: the generic A = D iteration runs then.
static int build_unwrapped_pinv(admm_engine* e, const double* Dp_given) {
  const int64_t m = e->m, n = e->n;
  if (std::getenv("ADMM_HIP_NO_UNWRAPPED_FUSED")) return ADMM_OK;
  if (!uw_supported(m, n) || (e->comm && comm_nranks(e->comm) > 1)) return ADMM_OK;
  if (static_cast<double>(m) * static_cast<double>(n) * 8.0 > 1.5e9) return ADMM_OK;
  const int64_t ldp = round_up(n, 2);
  if (Dp_given) {
    ADMM_TRY(e->mem.alloc(&e->Dp, static_cast<size_t>(ldp) * m));
    ADMM_HIP_TRY(hipMemcpy2DAsync(e->Dp, ldp * sizeof(double), Dp_given, n * sizeof(double), n * sizeof(double), m,
                                  hipMemcpyDeviceToDevice, e->stream));
  } else {
    SliceFactor& f = e->xfac;
    if (e->xsolve_requested == ADMM_XSOLVE_TRSV || e->xsolve_requested == ADMM_XSOLVE_CG) return ADMM_OK;
    SliceFactor tmp{};
    const double* Minv = nullptr;
    int64_t ldM = 0;
    if (f.mode == ADMM_XSOLVE_INVERSE && f.Minv && !f.planSy.packed) {
      Minv = f.Minv;
      ldM = f.ldM;
    } else if (f.mode == ADMM_XSOLVE_TRSV && !f.probed && f.F) {
      tmp.F = f.F;
      tmp.n = f.n;
      tmp.ld = f.ld;
      tmp.dinv = f.dinv;
      ADMM_TRY(build_explicit_inverse(e, tmp, false));
      Minv = tmp.Minv;
      ldM = tmp.ldM;
    } else {
      return ADMM_OK;
    }
    ADMM_TRY(e->mem.alloc(&e->Dp, static_cast<size_t>(ldp) * m));
    ADMM_HIP_TRY(hipMemsetAsync(e->Dp, 0, sizeof(double) * ldp * m, e->stream));
    launch_gemm(0, 1, n, m, n, 1.0, Minv, ldM, e->D, e->ldD, 0.0, e->Dp, ldp, false, e->stream);
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    if (tmp.Minv) mem_free_one(e->mem, tmp.Minv);
  }
  e->ldDp = ldp;
  e->uwR = uw_rows_per_block(m);
  e->uwnblk = static_cast<int32_t>(ceil_div(m, static_cast<int64_t>(e->uwR)));
  e->uwldg = round_up(n, 2);
  ADMM_TRY(e->mem.alloc(&e->uwG, static_cast<size_t>(2) * e->uwnblk * e->uwldg));
  ADMM_HIP_TRY(hipMemsetAsync(e->uwG, 0, sizeof(double) * 2 * e->uwnblk * e->uwldg, e->stream));
  e->uwnchunk = uw_chunks(n);
  e->uwldax = round_up(m, 2);
  ADMM_TRY(e->mem.alloc(&e->uwAx, static_cast<size_t>(e->uwnchunk) * e->uwldax));
  ADMM_TRY(e->mem.alloc(&e->uwX, static_cast<size_t>(2) * e->uwldg));
  return ADMM_OK;
}

// ---- one-time reductions some solvers do before the loop, on the device ---------------------------------------------
// W (n x n, ld, lower triangle valid, SPD) -> its explicit inverse, full symmetric storage (tile-padded, ld *ldM).
// W is destroyed; the result is owned by e->mem (release with mem_free_one).
static int spd_inverse(admm_engine* e, double* W, int64_t n, int64_t ld, double** Minv, int64_t* ldM) {
  SliceFactor f{};
  f.F = W;
  f.n = n;
  f.ld = ld;
  ADMM_TRY(e->mem.alloc(&f.dinv, static_cast<size_t>(ceil_div(n, 64)) * 64 * 64));
  double* infod = nullptr;
  ADMM_TRY(e->mem.alloc(&infod, 1));
  ADMM_TRY(cholesky_lower(W, n, ld, reinterpret_cast<int32_t*>(infod), f.dinv, e->stream));
  int32_t info = 0;
  ADMM_HIP_TRY(hipMemcpyAsync(&info, infod, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  mem_free_one(e->mem, infod);
  if (info != 0)
    return fail(ADMM_E_NUMERIC, "Cholesky failed: matrix must be positive definite (pivot " + std::to_string(info) +
                                    "): the constraint matrix needs full row rank");
  ADMM_TRY(build_explicit_inverse(e, f, false));
  mem_free_one(e->mem, f.dinv);
  *Minv = f.Minv;
  *ldM = f.ldM;
  return ADMM_OK;
}

// out[n] = alpha * T' * v for T (m x n, ldt)
static int scaled_tdot(admm_engine* e, const double* T, int64_t m, int64_t n, int64_t ldt, const double* v, double alpha,
                       double* out) {
  const GemvTPlan pt = gemv_t_plan(m, n, ldt);
  double *part = nullptr, *tmp = nullptr;
  ADMM_TRY(e->mem.alloc(&part, pt.part_elems(1)));
  ADMM_TRY(e->mem.alloc(&tmp, round_up(n, 2)));
  launch_gemv_t(pt, T, v, nullptr, nullptr, 1, part, nullptr, e->stream);
  launch_sum_partials_t(pt, part, 1, tmp, round_up(n, 2), nullptr, e->stream);
  launch_combine(tmp, 1, 0, alpha, nullptr, 0.0, nullptr, out, n, nullptr, e->stream);
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  mem_free_one(e->mem, part);
  mem_free_one(e->mem, tmp);
  return ADMM_OK;
}

// The affine maps built below (x = K*y + k0 of the eliminated KKT system; x = P*v + q of basis pursuit) go through
// explicit inverses of D*inv(M)*D' resp. D*D', whose condition number is cond(D)^2 -- where the reference solves the
// KKT system by pivoted backslash in every iteration (getProxOps.m:1363, 1410) or forms P by backslash
// (basispursuit.m:116-120).  What must survive is the constraint: every x the map produces satisfies D*x = s.  Probe it
// once, for an unstructured argument: relative violation max|D*x - s| / (max|s| + max|D*x|) above 1e-9 refuses the
// problem (ADMM_E_NUMERIC) instead of iterating on iterates that silently leave the feasible set.
static int probe_affine_map(admm_engine* e, const double* D, int64_t m, int64_t n, int64_t ldD, const double* s_dev,
                            const double* Kmat, int64_t ldK, const double* k0, const char* what) {
  std::vector<double> yh(static_cast<size_t>(n)), xh(static_cast<size_t>(n)), rh(static_cast<size_t>(m)),
      sh(static_cast<size_t>(m));
  for (int64_t i = 0; i < n; ++i) yh[i] = std::sin(0.7 * static_cast<double>(i) + 0.3) + 0.25;
  double *y = nullptr, *x = nullptr, *r = nullptr, *partK = nullptr, *partD = nullptr;
  ADMM_TRY(e->mem.alloc(&y, round_up(n, 2)));
  ADMM_TRY(e->mem.alloc(&x, round_up(n, 2)));
  ADMM_TRY(e->mem.alloc(&r, round_up(m, 2)));
  ADMM_HIP_TRY(hipMemcpyAsync(y, yh.data(), sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
  const GemvTPlan pk = gemv_t_plan(n, n, ldK);  // K (P) is symmetric: K*y as column dots
  ADMM_TRY(e->mem.alloc(&partK, pk.part_elems(1)));
  launch_gemv_t(pk, Kmat, y, nullptr, nullptr, 1, partK, nullptr, e->stream);
  launch_sum_partials_t(pk, partK, 1, x, round_up(n, 2), nullptr, e->stream);
  launch_combine(x, 1, 0, 1.0, nullptr, 0.0, k0, x, n, nullptr, e->stream);  // x = K*y + k0
  const GemvNPlan pd = gemv_n_plan(m, n, ldD);
  ADMM_TRY(e->mem.alloc(&partD, pd.part_elems()));
  launch_gemv_n(pd, D, x, partD, nullptr, e->stream);
  launch_sum_partials(partD, pd.nchunk, pd.ldy, m, r, nullptr, e->stream);
  ADMM_HIP_TRY(hipMemcpyAsync(rh.data(), r, sizeof(double) * m, hipMemcpyDeviceToHost, e->stream));
  ADMM_HIP_TRY(hipMemcpyAsync(sh.data(), s_dev, sizeof(double) * m, hipMemcpyDeviceToHost, e->stream));
  // the size of D*(an unstructured vector): with s = 0 (or tiny) D*x and s are both rounding noise, and their own
  // magnitudes would make a relative violation of order 1 out of nothing (found by the solver sweep: 1 x 2, s = 0)
  std::vector<double> dyh(static_cast<size_t>(m));
  launch_gemv_n(pd, D, y, partD, nullptr, e->stream);
  launch_sum_partials(partD, pd.nchunk, pd.ldy, m, r, nullptr, e->stream);
  ADMM_HIP_TRY(hipMemcpyAsync(dyh.data(), r, sizeof(double) * m, hipMemcpyDeviceToHost, e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  for (double* p : {y, x, r, partK, partD}) mem_free_one(e->mem, p);
  double viol = 0.0, scale = 0.0;
  bool finite = true;
  for (int64_t i = 0; i < m; ++i) {
    finite = finite && std::isfinite(rh[i]);
    viol = std::max(viol, std::fabs(rh[i] - sh[i]));
    scale = std::max(scale, std::max(std::max(std::fabs(rh[i]), std::fabs(sh[i])), std::fabs(dyh[i])));
  }
  const double rel = viol / (scale > 0.0 ? scale : 1.0);
  if (!finite || !(rel <= 1e-9))
    return fail(ADMM_E_NUMERIC, std::string(what) + ": the constraint matrix D is too ill-conditioned for the eliminated "
                                    "solve (cond(D)^2 enters it): D*x = s is violated by " + std::to_string(rel) +
                                    " (relative) for a probe vector; the reference's pivoted KKT solve would be needed");
  return ADMM_OK;
}

// basispursuit.m:116-120: P = I - D'(DD')^-1 D, q = D'(DD')^-1 s for a fat D (m x n, m < n) -- MFMA GEMMs, the
// Cholesky factorisation and the explicit m x m inverse on the device (the reference computes them in MATLAB)
static int build_bp_projector(admm_engine* e, const double* D, int64_t m, int64_t n, int64_t ldD, const double* s_dev) {
  const int64_t ldw = round_up(m, 16);
  double *W = nullptr, *Ginv = nullptr, *X = nullptr;
  int64_t ldG = 0;
  ADMM_TRY(e->mem.alloc(&W, static_cast<size_t>(ldw) * m));
  ADMM_HIP_TRY(hipMemsetAsync(W, 0, sizeof(double) * ldw * m, e->stream));
  launch_gemm(0, 1, m, m, n, 1.0, D, ldD, D, ldD, 0.0, W, ldw, true, e->stream);  // D*D'
  ADMM_TRY(spd_inverse(e, W, m, ldw, &Ginv, &ldG));
  ADMM_TRY(e->mem.alloc(&X, static_cast<size_t>(ldw) * n));
  launch_gemm(0, 0, m, n, m, 1.0, Ginv, ldG, D, ldD, 0.0, X, ldw, false, e->stream);  // (DD')^-1 D
  e->ldP = round_up(n, 16);
  ADMM_TRY(e->mem.alloc(&e->Pmat, static_cast<size_t>(e->ldP) * n));
  ADMM_HIP_TRY(hipMemsetAsync(e->Pmat, 0, sizeof(double) * e->ldP * n, e->stream));
  launch_gemm(1, 0, n, n, m, -1.0, D, ldD, X, ldw, 0.0, e->Pmat, e->ldP, false, e->stream);  // -D'(DD')^-1 D
  launch_add_diag(e->Pmat, n, e->ldP, 1.0, e->stream);
  ADMM_TRY(e->mem.alloc(&e->q, round_up(n, 2)));
  ADMM_TRY(scaled_tdot(e, X, m, n, ldw, s_dev, 1.0, e->q));  // X' s
  mem_free_one(e->mem, W);
  mem_free_one(e->mem, Ginv);
  mem_free_one(e->mem, X);
  return probe_affine_map(e, D, m, n, ldD, s_dev, e->Pmat, e->ldP, e->q, "basis pursuit");
}

// getProxOps.m:1363 / 1410 solve [M D'; D 0] [x; nu] = [y; s] every iteration, M = rho*I (linear program) or P + rho*I
// (standard-form QP).  Eliminating nu ONCE gives x = K*y + k0 with
//   K = inv(M) - inv(M) D' inv(S) D inv(M),  k0 = inv(M) D' inv(S) s,  S = D inv(M) D'
// -- built here with MFMA GEMMs and two explicit SPD inverses; the loop then runs one symmetric GEMV per x-update.
static int build_kkt_map(admm_engine* e, const double* D, int64_t m, int64_t n, int64_t ldD, const double* s_dev,
                         const double* P /* null: LP */, int64_t ldPm, double rho) {
  const int64_t ldn = round_up(n, 16), ldm = round_up(m, 16);
  double *MD = nullptr, *S = nullptr, *Sinv = nullptr, *T2 = nullptr, *Mi = nullptr;
  int64_t ldS = 0, ldMi = 0;
  e->ldK = ldn;
  ADMM_TRY(e->mem.alloc(&e->Kmat, static_cast<size_t>(ldn) * n));
  ADMM_HIP_TRY(hipMemsetAsync(e->Kmat, 0, sizeof(double) * ldn * n, e->stream));
  ADMM_TRY(e->mem.alloc(&MD, static_cast<size_t>(ldn) * m));  // inv(M) D'  (n x m)
  if (P) {
    double* Mw = nullptr;
    ADMM_TRY(e->mem.alloc(&Mw, static_cast<size_t>(ldn) * n));
    ADMM_HIP_TRY(hipMemsetAsync(Mw, 0, sizeof(double) * ldn * n, e->stream));
    ADMM_HIP_TRY(hipMemcpy2DAsync(Mw, ldn * sizeof(double), P, ldPm * sizeof(double), n * sizeof(double), n,
                                  hipMemcpyDeviceToDevice, e->stream));
    launch_add_diag(Mw, n, ldn, rho, e->stream);
    ADMM_TRY(spd_inverse(e, Mw, n, ldn, &Mi, &ldMi));
    mem_free_one(e->mem, Mw);
    launch_gemm(0, 1, n, m, n, 1.0, Mi, ldMi, D, ldD, 0.0, MD, ldn, false, e->stream);
    ADMM_HIP_TRY(hipMemcpy2DAsync(e->Kmat, ldn * sizeof(double), Mi, ldMi * sizeof(double), n * sizeof(double), n,
                                  hipMemcpyDeviceToDevice, e->stream));  // K starts as inv(M)
  } else {
    // inv(M) = I / rho:  MD = D'/rho  (a transposed, scaled copy through the GEMM with the identity left out:
    // MD(:, j) = D(j, :)'/rho is just the GEMM D' * (I/rho) -- done as a transpose-free product below)
    double* Ieye = nullptr;
    ADMM_TRY(e->mem.alloc(&Ieye, static_cast<size_t>(ldm) * m));
    ADMM_HIP_TRY(hipMemsetAsync(Ieye, 0, sizeof(double) * ldm * m, e->stream));
    launch_add_diag(Ieye, m, ldm, 1.0 / rho, e->stream);
    launch_gemm(1, 0, n, m, m, 1.0, D, ldD, Ieye, ldm, 0.0, MD, ldn, false, e->stream);
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    mem_free_one(e->mem, Ieye);
    launch_add_diag(e->Kmat, n, ldn, 1.0 / rho, e->stream);  // K starts as I / rho
  }
  ADMM_TRY(e->mem.alloc(&S, static_cast<size_t>(ldm) * m));
  ADMM_HIP_TRY(hipMemsetAsync(S, 0, sizeof(double) * ldm * m, e->stream));
  launch_gemm(0, 0, m, m, n, 1.0, D, ldD, MD, ldn, 0.0, S, ldm, true, e->stream);  // S = D inv(M) D'
  ADMM_TRY(spd_inverse(e, S, m, ldm, &Sinv, &ldS));
  ADMM_TRY(e->mem.alloc(&T2, static_cast<size_t>(ldm) * n));  // inv(S) D inv(M)  (m x n)
  launch_gemm(0, 1, m, n, m, 1.0, Sinv, ldS, MD, ldn, 0.0, T2, ldm, false, e->stream);
  launch_gemm(0, 0, n, n, m, -1.0, MD, ldn, T2, ldm, 1.0, e->Kmat, ldn, false, e->stream);  // K -= MD * T2
  ADMM_TRY(e->mem.alloc(&e->k0, round_up(n, 2)));
  ADMM_TRY(scaled_tdot(e, T2, m, n, ldm, s_dev, 1.0, e->k0));  // k0 = T2' s = inv(M) D' inv(S) s
  for (double* p : {MD, S, Sinv, T2, Mi}) mem_free_one(e->mem, p);
  return probe_affine_map(e, D, m, n, ldD, s_dev, e->Kmat, ldn, e->k0, P ? "quadratic program" : "linear program");
}

// the engine's own x-update factor
int factorize(admm_engine* e, double* W, int64_t nF, int64_t ld, const double* Lgiven, int memkind) {
  SliceFactor& f = e->xfac;
  // sharded engines must take the same decision on every rank: the factor is replicated bit for bit (all-reduced
  // Gram matrix, same kernels), so the probe is too
  ADMM_TRY(build_slice_factor(e, f, W, nF, ld, e->xsolve_requested, Lgiven, memkind));
  e->F = f.F;
  e->nF = nF;
  e->ldF = ld;
  e->xsolve = f.mode;
  return decide_sy_split(e);
}

// D'D is rank deficient (or numerically so): x = pinv(D)*(z-u) = (D'D)^+ D'(z-u)  (linearsvm.m:185,
// unwrappedadmm.m:76-78, getProxOps.m:1067).  W: the Gram matrix, lower triangle valid.  Builds
// (D'D)^+ = V diag(1/lambda_i : lambda_i > tol) V' with the Jacobi eigen-solver and applies it like the explicit
// inverse.  tol = n * eps * lambda_max: MATLAB's pinv rule max(size)*eps(norm) applied to D'D (the rank decisions
// agree with pinv(D)'s whenever the small singular values of D are zero to rounding, e.g. zero or duplicated columns).
int factorize_pinv(admm_engine* e, double* W, int64_t n, int64_t ld) {
  SliceFactor& f = e->xfac;
  f = SliceFactor{};
  f.n = n;
  f.ld = ld;
  launch_symmetrize_lower(W, n, ld, e->stream);
  double *V = nullptr, *lam = nullptr, *rotd = nullptr;
  const int64_t ldv = round_up(n, 16);
  ADMM_TRY(e->mem.alloc(&V, static_cast<size_t>(ldv) * n));
  ADMM_TRY(e->mem.alloc(&lam, n));
  ADMM_TRY(e->mem.alloc(&rotd, 1));
  std::vector<double> lh;
  int sweeps = 0;
  ADMM_TRY(jacobi_eig_psd(W, n, ld, V, ldv, lam, reinterpret_cast<int32_t*>(rotd), &lh, &sweeps, e->stream));
  double lmax = 0.0;
  for (double v : lh) lmax = v > lmax ? v : lmax;
  const double tol = static_cast<double>(n) * 2.220446049250313e-16 * lmax;
  int64_t rank = 0;
  double lmin = INFINITY;
  for (double& v : lh) {
    if (v > tol) {
      ++rank;
      lmin = v < lmin ? v : lmin;
      v = 1.0 / std::sqrt(v);
    } else {
      v = 0.0;
    }
  }
  ADMM_HIP_TRY(hipMemcpyAsync(lam, lh.data(), sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
  launch_scale_cols(V, ldv, n, lam, e->stream);  // V <- V diag(lambda^-1/2)
  f.planSy = symv_plan(n);
  f.ldM = f.planSy.npad;  // (4 / 8 / 16 KiB-aligned columns were measured: no effect on the lower-triangle kernel)
  ADMM_TRY(e->mem.alloc(&f.Minv, static_cast<size_t>(f.ldM) * f.planSy.npad));
  ADMM_HIP_TRY(hipMemsetAsync(f.Minv, 0, sizeof(double) * f.ldM * f.planSy.npad, e->stream));
  launch_gemm(0, 1, n, n, n, 1.0, V, ldv, V, ldv, 0.0, f.Minv, f.ldM, true, e->stream);
  launch_symmetrize_lower(f.Minv, n, f.ldM, e->stream);
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  mem_free_one(e->mem, V);
  mem_free_one(e->mem, lam);
  mem_free_one(e->mem, rotd);
  ADMM_TRY(pack_inverse(e, f));
  if (n >= kSymvHalfMin) ADMM_TRY(ensure_sy_buffers(e, f.planSy));
  f.mode = ADMM_XSOLVE_INVERSE;
  f.pinv = true;
  f.rank = rank;
  f.jacobi_sweeps = sweeps;
  f.cond_diag = rank > 0 ? lmax / lmin : INFINITY;  // over the retained spectrum
  e->F = nullptr;  // no Cholesky factor exists
  e->nF = n;
  e->ldF = ld;
  e->xsolve = ADMM_XSOLVE_INVERSE;
  return decide_sy_split(e);
}

// x = inv(LL')*y from the lower triangle of the explicit inverse; on a sharded engine every rank may stream 1/N of
// the tiles and ONE all-reduce of n doubles assembles x
int symv_apply(admm_engine* e, const double* y, double* out) {
  const SliceFactor& f = e->xfac;
  const int nr = e->comm ? comm_nranks(e->comm) : 1;
  if (nr > 1 && e->sy_split) {
    if (e->dfin && f.planSy.packed)  // the previous iteration's deferred finalize rides along (engine_run.hip)
      launch_symv_lower_fin(f.planSy, f.Minv, y, e->syN, e->syT, *e->dfin, e->dfin_pending, e->ctrl, e->stream,
                            comm_rank(e->comm), nr, out);
    else
      launch_symv_lower(f.planSy, f.Minv, f.ldM, y, e->syN, e->syT, out, e->ctrl, e->stream, comm_rank(e->comm), nr);
    return comm_allreduce_device(e->comm, out, static_cast<size_t>(f.n), e->stream);
  }
  apply_inverse(e, f, y, out, e->ctrl);
  return ADMM_OK;
}

// out = inv(L L') y with the engine's own factor
int solve_factor(admm_engine* e, const double* y, double* out) {
  if (e->xfac.mode == ADMM_XSOLVE_INVERSE) return symv_apply(e, y, out);
  launch_trsv_pair(e->xfac.trsv, y, out, e->ctrl, e->stream);
  return ADMM_OK;
}

}  // namespace admm

extern "C" {

int admm_abi_version(void) { return ADMM_ABI_VERSION; }
const char* admm_last_error(void) { return g_last_error.c_str(); }

int admm_device_count(int* count) {
  if (!count) return fail(ADMM_E_INVALID, "count is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return fail(ADMM_E_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = c;
  return ADMM_OK;
}

int admm_device_info(int device, char* name, size_t cap, int64_t* hbm_bytes, int32_t* cus) {
  hipDeviceProp_t prop;
  ADMM_HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (name && cap > 0) {
    std::strncpy(name, prop.gcnArchName, cap - 1);
    name[cap - 1] = 0;
  }
  if (hbm_bytes) *hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
  if (cus) *cus = prop.multiProcessorCount;
  return ADMM_OK;
}

void admm_options_default(admm_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->struct_size = sizeof(admm_options);
  o->maxiters = 1000;   // admm.m:58
  o->rho = 1.0;         // admm.m:57
  o->relax = 1.0;       // admm.m:60
  o->abstol = 1e-5;     // admm.m:71
  o->reltol = 1e-3;     // admm.m:72
  o->Hnormtol = 1e-6;   // admm.m:73
  o->convtol = 1e-10;   // admm.m:68
  o->restart = 0.999;   // admm.m:282
  o->dvaltol = 1e-8;    // admm.m:290
  o->record_history = 1;
}

void admm_problem_desc_default(admm_problem_desc* d) {
  if (!d) return;
  std::memset(d, 0, sizeof(*d));
  d->struct_size = sizeof(admm_problem_desc);
  d->rho = 1.0;
  d->cg_tol = 1e-12;
  d->cg_maxit = 200;
  d->nslices = 1;
}

int admm_engine_create(const admm_problem_desc* desc, admm_engine** out) {
  if (!desc || !out) return fail(ADMM_E_INVALID, "desc/out is NULL");
  if (desc->struct_size != static_cast<int32_t>(sizeof(admm_problem_desc)))
    return fail(ADMM_E_INVALID, "admm_problem_desc.struct_size mismatch (ABI version skew)");
  *out = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  int ndev = 0;
  ADMM_TRY(admm_device_count(&ndev));
  if (ndev <= 0) return fail(ADMM_E_DEVICE, "no HIP device visible: the ADMM engine has no CPU fallback");
  if (desc->device < 0 || desc->device >= ndev) return fail(ADMM_E_INVALID, "bad device ordinal");
  ADMM_HIP_TRY(hipSetDevice(desc->device));

  admm_engine* e = new admm_engine();
  auto bail = [&](int rc) {
    admm_engine_destroy(e);
    return rc;
  };
#define E_TRY(expr)                  \
  do {                               \
    int _rc = (expr);                \
    if (_rc != ADMM_OK) return bail(_rc); \
  } while (0)
#define E_HIP(expr)                                                                                   \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess) return bail(fail(ADMM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)

  e->device = desc->device;
  e->comm = desc->comm;
  E_TRY(comm_stream_create(e->comm, &e->stream));
  e->problem = desc->problem;
  e->m = desc->m;
  e->n = desc->n;
  e->lambda = desc->lambda;
  e->C = desc->C;
  e->rconst = desc->r;
  e->loss = desc->loss;
  e->rho_factor = desc->rho;
  const int mk = desc->mem;
  const int64_t m = desc->m, n = desc->n;
  if (!(desc->rho > 0.0)) return bail(fail(ADMM_E_INVALID, "rho must be a positive real (lasso.m:138)"));

  int xs = desc->xsolve;
  if (xs < ADMM_XSOLVE_AUTO || xs > ADMM_XSOLVE_PINV) return bail(fail(ADMM_E_INVALID, "bad desc.xsolve"));
  if (xs == ADMM_XSOLVE_PINV && desc->problem != ADMM_PROB_LINEARSVM)
    return bail(fail(ADMM_E_UNSUPPORTED, "xsolve=pinv is the linear SVM / unwrapped ADMM x-update (unwrappedadmm.m:76-78)"));
  // AUTO stays AUTO here: build_slice_factor resolves it per factor (blocked triangular solves up to n = 256, beyond
  // that the explicit inverse if its probe holds, see probe_and_choose)
  e->xsolve_requested = (xs == ADMM_XSOLVE_TRSV || xs == ADMM_XSOLVE_INVERSE) ? xs : ADMM_XSOLVE_AUTO;
  if (xs == ADMM_XSOLVE_CALLBACK && desc->problem != ADMM_PROB_LAD)
    return bail(fail(ADMM_E_UNSUPPORTED, "xsolve=callback is the generic A = D engine: use ADMM_PROB_LAD with D = A, s = c"));
  if (xs == ADMM_XSOLVE_CALLBACK && e->comm && comm_nranks(e->comm) > 1)
    return bail(fail(ADMM_E_UNSUPPORTED, "prox callbacks are not supported on row-sharded engines"));
  // 2-D TV: AUTO = the direct spectral solve when both sides are powers of two, else (or on request) warm-started
  // CG; the CG vectors are allocated either way (one of them is the transposition scratch of the spectral solve)
  // (the row stage needs no transform -- dct.hip: tv2d_rows_green_kernel -- so only the HEIGHT has to be a power of two
  // as long as the row kernel's truncation fits the width; run() falls back to CG for a rho where it does not)
  // (a height that is not a power of two takes the chirp form of the column transform, dct.hip: any height up to 4096)
  const bool tv2_want_dct = desc->problem == ADMM_PROB_TV2D && desc->xsolve != ADMM_XSOLVE_CG &&
                            (dct_length_ok(desc->m) || dct_chirp_length_ok(desc->m));  // (any width: run() picks the row stage)
  if (desc->problem == ADMM_PROB_TV2D) xs = ADMM_XSOLVE_CG;
  if (xs == ADMM_XSOLVE_CG && desc->problem != ADMM_PROB_TV2D && desc->problem != ADMM_PROB_LASSO && desc->problem != ADMM_PROB_LAD &&
      desc->problem != ADMM_PROB_HUBERFIT && desc->problem != ADMM_PROB_LINEARSVM)
    return bail(fail(ADMM_E_UNSUPPORTED, "xsolve=cg applies to problems whose x-update solves with D'D (+ rho I)"));
  e->xsolve = xs;
  e->cg_tol = desc->cg_tol > 0 ? desc->cg_tol : 1e-12;
  e->cg_maxit = desc->cg_maxit > 0 ? desc->cg_maxit : 200;
  const bool sharded = e->comm && comm_nranks(e->comm) > 1;
  if (sharded && desc->problem != ADMM_PROB_LASSO && desc->problem != ADMM_PROB_LASSO_CONSENSUS &&
      desc->problem != ADMM_PROB_LAD &&
      desc->problem != ADMM_PROB_HUBERFIT && desc->problem != ADMM_PROB_LINEARSVM)
    return bail(fail(ADMM_E_UNSUPPORTED, "row sharding applies to problems with a data matrix D (lasso/LAD/Huber/SVM)"));
  // global number of rows of D (the local m when not sharded)
  int64_t m_global = m;
  if (sharded) {
    double* cnt = nullptr;
    E_TRY(e->mem.alloc(&cnt, 2));
    const double mine = static_cast<double>(m);
    E_HIP(hipMemcpyAsync(cnt, &mine, sizeof(double), hipMemcpyHostToDevice, e->stream));
    E_TRY(comm_allreduce_device(e->comm, cnt, 1, e->stream));
    double tot = 0.0;
    E_HIP(hipMemcpyAsync(&tot, cnt, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    E_HIP(hipStreamSynchronize(e->stream));
    m_global = static_cast<int64_t>(tot + 0.5);
  }
  e->len_global = 0;

  switch (desc->problem) {
    case ADMM_PROB_LASSO: {
      if (!desc->D || (!desc->s && !desc->Dts) || m <= 0 || n <= 0)
        return bail(fail(ADMM_E_INVALID, "lasso needs D (m x n) and s (or D'*s)"));
      if (desc->lambda < 0) return bail(fail(ADMM_E_INVALID, "lambda must be a nonnegative real (lasso.m:132)"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_SOFT;
      e->rhs_kind = RHS_RHO_DTS;
      e->fat = m_global < n;
      if (sharded && e->fat)
        return bail(fail(ADMM_E_UNSUPPORTED, "row-sharded lasso needs a tall matrix (global m >= n)"));
      // A big host matrix whose Gram matrix is needed anyway: upload it in row chunks and accumulate D_c'*D_c of the
      // chunk that has arrived on a second stream while the next chunk crosses PCIe (the copy from pageable memory
      // keeps the host busy, the GEMM does not): create() drops from upload + Gram to about the longer of the two
      double* Wpre = nullptr;  // D'*D, accumulated during the upload (lasso.m:168 before the rho shift)
      const int64_t ldW = round_up(e->fat ? m : n, 16);
      if (mk == ADMM_MEM_HOST && !e->fat && !desc->L && e->xsolve != ADMM_XSOLVE_CG && m >= 32768 &&
          static_cast<double>(m) * n >= 1e8 && std::getenv("ADMM_HIP_NO_UPLOAD_OVERLAP") == nullptr) {
        const int64_t ldsrc = desc->ldD ? desc->ldD : m;
        e->ldD = round_up(m, 512);
        E_TRY(e->mem.alloc(&e->D, static_cast<size_t>(e->ldD) * n));
        if (e->ldD != m) E_HIP(hipMemsetAsync(e->D, 0, sizeof(double) * e->ldD * n, e->stream));
        E_TRY(e->mem.alloc(&Wpre, static_cast<size_t>(ldW) * n));
        E_HIP(hipMemsetAsync(Wpre, 0, sizeof(double) * ldW * n, e->stream));
        hipStream_t gs = nullptr;
        hipEvent_t ev = nullptr;
        E_HIP(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
        E_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        const int64_t chunk = round_up(ceil_div(m, int64_t{8}), 512);
        hipError_t herr = hipSuccess;
        for (int64_t r0 = 0; r0 < m && herr == hipSuccess; r0 += chunk) {
          const int64_t rows = (m - r0 < chunk) ? m - r0 : chunk;
          herr = hipMemcpy2DAsync(e->D + r0, e->ldD * sizeof(double), desc->D + r0, ldsrc * sizeof(double),
                                  rows * sizeof(double), n, hipMemcpyHostToDevice, e->stream);
          if (herr == hipSuccess) herr = hipEventRecord(ev, e->stream);
          if (herr == hipSuccess) herr = hipStreamWaitEvent(gs, ev, 0);
          if (herr == hipSuccess)
            launch_gemm(1, 0, n, n, rows, 1.0, e->D + r0, e->ldD, e->D + r0, e->ldD, 1.0, Wpre, ldW, true, gs);
        }
        if (herr == hipSuccess) herr = hipStreamSynchronize(gs);
        (void)hipEventDestroy(ev);
        (void)hipStreamDestroy(gs);
        E_HIP(herr);
      } else {
        E_TRY(upload_matrix(e->mem, &e->D, &e->ldD, desc->D, m, n, desc->ldD ? desc->ldD : m, mk, e->stream));
      }
      if (desc->s) E_TRY(upload(e->mem, &e->s, desc->s, m, mk, e->stream));
      e->planDN = gemv_n_plan(m, n, e->ldD);
      e->planDT = gemv_t_plan(m, n, e->ldD);
      E_TRY(e->mem.alloc(&e->partDN, e->planDN.part_elems()));
      E_TRY(e->mem.alloc(&e->partDT, e->planDT.part_elems(3)));
      // Dts = D'*s   lasso.m:160 (or handed in: args.Dts, getProxOps.m:446)
      E_TRY(e->mem.alloc(&e->rhs_add, round_up(n, 2)));
      if (desc->Dts) {
        if (sharded) return bail(fail(ADMM_E_UNSUPPORTED, "args.Dts on a row-sharded engine: pass the local rows of s"));
        E_HIP(hipMemcpyAsync(e->rhs_add, desc->Dts, sizeof(double) * n,
                             mk == ADMM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
      } else {
        launch_gemv_t(e->planDT, e->D, e->s, nullptr, nullptr, 1, e->partDT, nullptr, e->stream);
        launch_sum_partials_t(e->planDT, e->partDT, 1, e->rhs_add, round_up(n, 2), nullptr, e->stream);
      }
      if (sharded) E_TRY(comm_allreduce_device(e->comm, e->rhs_add, n, e->stream));  // sum_g D_g'*s_g
      if (e->xsolve == ADMM_XSOLVE_CG) {  // matrix-free: nothing n x n is ever formed
        if (e->fat) return bail(fail(ADMM_E_UNSUPPORTED, "xsolve=cg needs a tall matrix (m >= n)"));
        e->cg_shift_is_rho = true;
        E_TRY(e->mem.alloc(&e->tmpA, round_up(m, 2)));
        break;
      }
      const int64_t nF = e->fat ? m : n;
      const int64_t ld = round_up(nF, 16);
      double* W = Wpre;
      if (!W) E_TRY(e->mem.alloc(&W, static_cast<size_t>(ld) * nF));
      if (!desc->L) {
        if (!Wpre) E_HIP(hipMemsetAsync(W, 0, sizeof(double) * ld * nF, e->stream));
        if (!e->fat) {  // lasso.m:168  chol(D'*D + rho*I)
          if (!Wpre) launch_gemm(1, 0, n, n, m, 1.0, e->D, e->ldD, e->D, e->ldD, 0.0, W, ld, true, e->stream);
          // W = sum_g D_g'*D_g  (unwrappedadmm.m:118-122); one-time, bandwidth-bound all-reduce
          if (sharded) E_TRY(comm_allreduce_device(e->comm, W, static_cast<size_t>(ld) * n, e->stream));
          launch_add_diag(W, n, ld, desc->rho, e->stream);
        } else {  // lasso.m:172  chol(1/rho*(D*D') + I)
          launch_gemm(0, 1, m, m, n, 1.0 / desc->rho, e->D, e->ldD, e->D, e->ldD, 0.0, W, ld, true, e->stream);
          launch_add_diag(W, m, ld, 1.0, e->stream);
        }
        const bool gram_auto = desc->obj_gram == 0;  // (any size: the form costs nothing, and is calibrated first)
        if ((desc->obj_gram > 0 || gram_auto) && e->s) {
          // The objective's data term without a pass over D: x solves (G + rho*I) x = y (directly, or through the
          // matrix-inversion lemma of the fat case, lasso.m:172), so G x = y - rho*x and
          // 1/2*||D x - s||^2 = 1/2*x'(y - rho*x) - x'D's + 1/2*s's comes out of the element update's own operands
          // (OBJX_SOLVE, prox_device.h).  Needs 1/2*s's, over all shards, once.  (Up to r2i this kept a packed
          // copy of G and spent one more symmetric-half pass per iteration on x'Gx.)
          e->obj_alt = true;
          e->obj_auto = desc->obj_gram == 0;
          E_TRY(e->mem.alloc(&e->gobjpart, kMaxPartBlocks + 2));
          std::vector<double> hs(static_cast<size_t>(m));
          E_HIP(hipMemcpyAsync(hs.data(), e->s, sizeof(double) * m, hipMemcpyDeviceToHost, e->stream));
          E_HIP(hipStreamSynchronize(e->stream));
          double ssq = 0.0;
          for (double v : hs) ssq += v * v;
          if (sharded) {
            E_HIP(hipMemcpyAsync(e->gobjpart, &ssq, sizeof(double), hipMemcpyHostToDevice, e->stream));
            E_TRY(comm_allreduce_device(e->comm, e->gobjpart, 1, e->stream));
            E_HIP(hipMemcpyAsync(&ssq, e->gobjpart, sizeof(double), hipMemcpyDeviceToHost, e->stream));
            E_HIP(hipStreamSynchronize(e->stream));
          }
          e->half_ssq = 0.5 * ssq;
        }
      }
      E_TRY(factorize(e, W, nF, ld, desc->L, mk));
      if (e->fat) {
        E_TRY(e->mem.alloc(&e->tmpA, round_up(m, 2)));
        E_TRY(e->mem.alloc(&e->tmpB, round_up(m, 2)));
      }
      break;
    }
    case ADMM_PROB_LAD:
    case ADMM_PROB_HUBERFIT:
    case ADMM_PROB_LINEARSVM: {
      const bool svm = desc->problem == ADMM_PROB_LINEARSVM;
      if (!desc->D && xs == ADMM_XSOLVE_CALLBACK && desc->problem == ADMM_PROB_LAD && m > 0 && n > 0) {
        // options.A / options.At are function handles (admm.m:117-158): no matrix, both operators are callbacks
        if (!desc->s) return bail(fail(ADMM_E_INVALID, "the operator form needs the constraint vector c (as s)"));
        if (sharded) return bail(fail(ADMM_E_UNSUPPORTED, "operator callbacks are not supported on row-sharded engines"));
        e->a_identity = false;
        e->nA = n;
        e->len = m;
        e->len_global = m;
        e->rhs_kind = RHS_T1;
        e->prox = PROX_SOFT;
        E_TRY(upload(e->mem, &e->s, desc->s, m, mk, e->stream));
        e->c = e->s;
        E_TRY(e->mem.alloc(&e->axbuf, round_up(m, 2)));
        break;
      }
      if (!desc->D || m <= 0 || n <= 0) return bail(fail(ADMM_E_INVALID, "problem needs D (m x n)"));
      if (!svm && !desc->s) return bail(fail(ADMM_E_INVALID, "LAD/Huber need the signal vector s"));
      if (svm && !desc->ell) return bail(fail(ADMM_E_INVALID, "linear SVM needs the label vector ell"));
      // (the linear SVM's x-update is pinv(D)*(z-u), linearsvm.m:185: it exists for a wide D too -- the rank-deficient
      // D'D falls through to the pseudo-inverse below; lad.m:134 / huberfit.m:166 call chol, which errors)
      if (m_global < n && xs != ADMM_XSOLVE_CALLBACK && !(svm && !sharded))
        return bail(fail(ADMM_E_INVALID, "D must have full column rank (m >= n) for chol(D'*D) (lad.m:134)"));
      e->a_identity = false;
      e->nA = n;
      e->len = m;
      e->len_global = m_global;
      e->rhs_kind = RHS_T1;
      if (desc->problem == ADMM_PROB_LAD) e->prox = PROX_SOFT;
      else if (desc->problem == ADMM_PROB_HUBERFIT) e->prox = PROX_HUBER;
      else e->prox = (desc->loss == ADMM_LOSS_01) ? PROX_01 : PROX_HINGE;
      E_TRY(upload_matrix(e->mem, &e->D, &e->ldD, desc->D, m, n, desc->ldD ? desc->ldD : m, mk, e->stream));
      if (!svm) {
        E_TRY(upload(e->mem, &e->s, desc->s, m, mk, e->stream));
        e->c = e->s;  // lad.m:142  options.c = s
      } else {
        E_TRY(upload(e->mem, &e->ell, desc->ell, m, mk, e->stream));
      }
      e->planDN = gemv_n_plan(m, n, e->ldD);
      e->planDT = gemv_t_plan(m, n, e->ldD);
      E_TRY(e->mem.alloc(&e->partDN, e->planDN.part_elems()));
      E_TRY(e->mem.alloc(&e->partDT, e->planDT.part_elems(3)));
      if (e->xsolve == ADMM_XSOLVE_CG) {  // matrix-free normal equations D'D x = D'(c + z - u)
        e->cg_shift_is_rho = false;
        E_TRY(e->mem.alloc(&e->tmpA, round_up(m, 2)));
        break;
      }
      if (e->xsolve == ADMM_XSOLVE_CALLBACK) break;  // the caller's xminf is the x-update: nothing to factor
      if (svm && desc->Dplus) {
        // args.Dplus = pinv(D) from the caller (linearsvm.m:185-186): x = Dplus*(z-u) is one pass over its
        // transpose with the column-dot kernel (the same bytes as D'*v); nothing is factored
        double* Dp = nullptr;
        E_TRY(upload(e->mem, &Dp, desc->Dplus, static_cast<size_t>(n) * m, mk, e->stream));
        double* Dpt = nullptr;
        E_TRY(e->mem.alloc(&Dpt, static_cast<size_t>(m) * n));
        launch_transpose(Dp, Dpt, n, m, nullptr, e->stream);  // (n x m) -> (m x n)
        E_TRY(e->mem.alloc(&e->DplusT, static_cast<size_t>(e->ldD) * n));
        E_HIP(hipMemsetAsync(e->DplusT, 0, sizeof(double) * e->ldD * n, e->stream));
        E_HIP(hipMemcpy2DAsync(e->DplusT, e->ldD * sizeof(double), Dpt, m * sizeof(double), m * sizeof(double), n,
                               hipMemcpyDeviceToDevice, e->stream));
        E_HIP(hipStreamSynchronize(e->stream));
        E_TRY(build_unwrapped_pinv(e, Dp));
        E_HIP(hipStreamSynchronize(e->stream));
        mem_free_one(e->mem, Dp);
        mem_free_one(e->mem, Dpt);
        e->xsolve = ADMM_XSOLVE_PINV;
        break;
      }
      const int64_t ld = round_up(n, 16);
      double* W = nullptr;
      E_TRY(e->mem.alloc(&W, static_cast<size_t>(ld) * n));
      if (!desc->L) {  // lad.m:134  chol(D'*D,'lower') (un-shifted; also D^+ = (D'D)^-1 D' for the SVM)
        E_HIP(hipMemsetAsync(W, 0, sizeof(double) * ld * n, e->stream));
        launch_gemm(1, 0, n, n, m, 1.0, e->D, e->ldD, e->D, e->ldD, 0.0, W, ld, true, e->stream);
        // W = sum_g D_g'*D_g  (unwrappedadmm.m:96-123)
        if (sharded) E_TRY(comm_allreduce_device(e->comm, W, static_cast<size_t>(ld) * n, e->stream));
      }
      if (!svm || desc->L) {  // lad.m:134 / huberfit.m:166: chol errors on a rank-deficient D, and so does the engine
        E_TRY(factorize(e, W, n, ld, desc->L, mk));
        // exactly dependent columns leave a pivot at the rounding level of D'D, of either sign: where MATLAB's chol
        // may or may not error, the engine always refuses (the iterates would be noise amplified by 1/pivot)
        if (!desc->L && !(e->xfac.cond_diag < 1.0 / (static_cast<double>(n) * 2.220446049250313e-16)))
          return bail(fail(ADMM_E_NUMERIC, "Cholesky failed: D'*D is numerically singular (pivot ratio " +
                                               std::to_string(e->xfac.cond_diag) + "): D must have full column rank (lad.m:134)"));
        break;
      }
      // linear SVM: the reference's x-update is pinv(D)*(z-u) (linearsvm.m:185, unwrappedadmm.m:76-78), which exists
      // for every D.  Full column rank: (D'D)^-1 D' through the Cholesky factor (same map, to rounding).  Rank
      // deficient -- Cholesky breaks down, or its pivots fall to the rounding level of D'D -- or on request
      // (xsolve = pinv): the pseudo-inverse of D'D from its eigen-decomposition.
      {
        double* Wkeep = nullptr;  // the Gram matrix survives the in-place factorisation attempt
        E_TRY(e->mem.alloc(&Wkeep, static_cast<size_t>(ld) * n));
        E_HIP(hipMemcpyAsync(Wkeep, W, sizeof(double) * ld * n, hipMemcpyDeviceToDevice, e->stream));
        bool need_pinv = xs == ADMM_XSOLVE_PINV;
        if (!need_pinv) {
          const int rc = factorize(e, W, n, ld, nullptr, mk);
          if (rc == ADMM_E_NUMERIC) need_pinv = true;
          else if (rc != ADMM_OK) return bail(rc);
          else if (!(e->xfac.cond_diag < 1.0 / (static_cast<double>(n) * 2.220446049250313e-16))) need_pinv = true;
          if (need_pinv) {
            release_slice_factor(e, e->xfac);
            mem_free_one(e->mem, W);
          }
        }
        if (need_pinv) E_TRY(factorize_pinv(e, Wkeep, n, ld));
        else mem_free_one(e->mem, Wkeep);
      }
      E_TRY(build_unwrapped_pinv(e, nullptr));
      break;
    }
    case ADMM_PROB_QP_BOUNDED: {
      if (!desc->P || !desc->q || !desc->lb || !desc->ub || n <= 0)
        return bail(fail(ADMM_E_INVALID, "bounded QP needs P (n x n), q, lb, ub"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_BOX;
      e->rhs_kind = RHS_RHO_MINUS_Q;
      E_TRY(upload_matrix(e->mem, &e->Pmat, &e->ldP, desc->P, n, n, n, mk, e->stream));
      E_TRY(upload(e->mem, &e->q, desc->q, n, mk, e->stream));
      E_TRY(upload(e->mem, &e->lb, desc->lb, n, mk, e->stream));
      E_TRY(upload(e->mem, &e->ub, desc->ub, n, mk, e->stream));
      e->rhs_add = e->q;
      const int64_t ld = e->ldP;
      double* W = nullptr;
      E_TRY(e->mem.alloc(&W, static_cast<size_t>(ld) * n));
      if (!desc->L) {  // getProxOps.m:640-641  chol(P + rho*I)
        E_HIP(hipMemcpyAsync(W, e->Pmat, sizeof(double) * ld * n, hipMemcpyDeviceToDevice, e->stream));
        launch_add_diag(W, n, ld, desc->rho, e->stream);
      }
      E_TRY(factorize(e, W, n, ld, desc->L, mk));
      // the objective 1/2 x'Px + q'x + r needs P*x
      e->planSq = gemv_t_plan(n, n, ld);
      if (!e->partSq) E_TRY(e->mem.alloc(&e->partSq, e->planSq.part_elems(1)));
      // ... or, with the engine's own factor, nothing: P x = y - rho*x from the right-hand side the x-update solved
      // with (OBJX_SOLVE_QP), calibrated against the P*x form in the first batch like the lasso objective
      if (!desc->L && desc->obj_gram >= 0) {
        e->obj_alt = true;
        e->obj_auto = desc->obj_gram == 0;
        E_TRY(e->mem.alloc(&e->gobjpart, kMaxPartBlocks + 2));
      }
      break;
    }
    case ADMM_PROB_BASISPURSUIT: {
      const bool from_data = !desc->P && desc->D && desc->s && m > 0 && n > m;
      if (!from_data && (!desc->P || !desc->q || n <= 0))
        return bail(fail(ADMM_E_INVALID, "basis pursuit needs the projector P (n x n) and q, or a fat D (m < n) and s"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_SOFT;
      e->rhs_kind = RHS_DIFF;
      if (from_data) {  // basispursuit.m:116-120 on the device
        double *Dd = nullptr, *sd = nullptr;
        int64_t ldd = 0;
        E_TRY(upload_matrix(e->mem, &Dd, &ldd, desc->D, m, n, desc->ldD ? desc->ldD : m, mk, e->stream));
        E_TRY(upload(e->mem, &sd, desc->s, m, mk, e->stream));
        E_TRY(build_bp_projector(e, Dd, m, n, ldd, sd));
        mem_free_one(e->mem, Dd);
        mem_free_one(e->mem, sd);
        e->m = n;
      } else {
        E_TRY(upload_matrix(e->mem, &e->Pmat, &e->ldP, desc->P, n, n, n, mk, e->stream));
        E_TRY(upload(e->mem, &e->q, desc->q, n, mk, e->stream));
      }
      e->planSq = gemv_t_plan(n, n, e->ldP);
      E_TRY(e->mem.alloc(&e->partSq, e->planSq.part_elems(1)));
      e->xsolve = ADMM_XSOLVE_INVERSE;  // x = P*(z-u) + q is a GEMV by construction
      break;
    }
    case ADMM_PROB_MODEL: {
      // model.m:111-128 + getProxOps.m:83-95: x - z = c with quadratic f and g given by their Gram data.
      // Either half may be left out (NULL): it must then be supplied by admm_engine_set_callbacks.
      if (n <= 0) return bail(fail(ADMM_E_INVALID, "the model / generic problem needs n (args.n, getProxOps.m:89)"));
      if ((desc->P != nullptr) != (desc->q != nullptr) || (desc->Q != nullptr) != (desc->qz != nullptr))
        return bail(fail(ADMM_E_INVALID, "model: PtP comes with Ptr and QtQ with Qts (getProxOps.m:83-88)"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_GIVEN;
      e->rhs_kind = RHS_RHO_DTS;  // y = rho*(z-u) + Ptr   (getProxOps.m:978)
      if (xs == ADMM_XSOLVE_CG) return bail(fail(ADMM_E_UNSUPPORTED, "xsolve=cg needs a data matrix"));
      if (desc->c) {
        E_TRY(upload(e->mem, &e->s, desc->c, n, mk, e->stream));
        e->c = e->s;
      }
      e->has_xfac = desc->P != nullptr;
      if (desc->P) {
        E_TRY(upload(e->mem, &e->q, desc->q, n, mk, e->stream));
        e->rhs_add = e->q;
        double* W = nullptr;
        int64_t ld = 0;
        E_TRY(upload_matrix(e->mem, &W, &ld, desc->P, n, n, n, mk, e->stream));
        launch_add_diag(W, n, ld, desc->rho, e->stream);  // getProxOps.m:972-975
        E_TRY(factorize(e, W, n, ld, nullptr, mk));
      } else {
        e->rhs_kind = RHS_NONE;
      }
      if (desc->Q) {
        E_TRY(upload(e->mem, &e->qz, desc->qz, n, mk, e->stream));
        double* W = nullptr;
        int64_t ld = 0;
        E_TRY(upload_matrix(e->mem, &W, &ld, desc->Q, n, n, n, mk, e->stream));
        launch_add_diag(W, n, ld, desc->rho, e->stream);  // getProxOps.m:1005-1008
        E_TRY(build_slice_factor(e, e->zfac, W, n, ld, e->xsolve_requested, nullptr, mk));
        e->has_zfac = true;
      }
      // optional: the matrices of the objective 1/2||P*x-r||^2 + 1/2||Q*z-s||^2 (model.m:133-134)
      if (desc->D && desc->s && desc->D2 && desc->s2 && m > 0 && desc->m2 > 0) {
        E_TRY(upload_matrix(e->mem, &e->D, &e->ldD, desc->D, m, n, desc->ldD ? desc->ldD : m, mk, e->stream));
        E_TRY(upload(e->mem, &e->ell, desc->s, m, mk, e->stream));  // r (kept apart from the constraint vector)
        e->planDN = gemv_n_plan(m, n, e->ldD);
        E_TRY(e->mem.alloc(&e->partDN, e->planDN.part_elems()));
        e->m2 = desc->m2;
        E_TRY(upload_matrix(e->mem, &e->D2, &e->ldD2, desc->D2, e->m2, n, desc->ldD2 ? desc->ldD2 : e->m2, mk,
                            e->stream));
        E_TRY(upload(e->mem, &e->s2, desc->s2, e->m2, mk, e->stream));
        e->planD2N = gemv_n_plan(e->m2, n, e->ldD2);
        E_TRY(e->mem.alloc(&e->partD2N, e->planD2N.part_elems()));
      }
      const int64_t n2 = round_up(n, 2);
      E_TRY(e->mem.alloc(&e->xext, n2));
      E_TRY(e->mem.alloc(&e->zext, n2));
      E_TRY(e->mem.alloc(&e->xh, n2));
      E_TRY(e->mem.alloc(&e->rz, n2));
      break;
    }
    case ADMM_PROB_LINEARPROGRAM:
    case ADMM_PROB_QP_STANDARD: {
      const bool qp = desc->problem == ADMM_PROB_QP_STANDARD;
      const bool from_data = !desc->K && desc->D && desc->s && m > 0 && m < n;
      if ((!from_data && (!desc->K || !desc->k0)) || !desc->q || n <= 0 || (qp && !desc->P))
        return bail(fail(ADMM_E_INVALID, qp ? "standard-form QP needs P, q and either the reduced KKT map K, k0 or D, s"
                                            : "linear program needs b (as q) and either the reduced KKT map K, k0 or D, s"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_POS;                 // getProxOps.m:1381, 1425
      e->rhs_kind = RHS_RHO_MINUS_Q;      // y = rho*(z-u) - b   (getProxOps.m:1363, 1410)
      if (qp) E_TRY(upload_matrix(e->mem, &e->Pmat, &e->ldP, desc->P, n, n, n, mk, e->stream));
      if (from_data) {  // the KKT elimination on the device, for desc.rho
        double *Dd = nullptr, *sd = nullptr;
        int64_t ldd = 0;
        E_TRY(upload_matrix(e->mem, &Dd, &ldd, desc->D, m, n, desc->ldD ? desc->ldD : m, mk, e->stream));
        E_TRY(upload(e->mem, &sd, desc->s, m, mk, e->stream));
        E_TRY(build_kkt_map(e, Dd, m, n, ldd, sd, qp ? e->Pmat : nullptr, e->ldP, desc->rho));
        mem_free_one(e->mem, Dd);
        mem_free_one(e->mem, sd);
        e->m = n;
      } else {
        E_TRY(upload_matrix(e->mem, &e->Kmat, &e->ldK, desc->K, n, n, n, mk, e->stream));
        E_TRY(upload(e->mem, &e->k0, desc->k0, n, mk, e->stream));
      }
      E_TRY(upload(e->mem, &e->q, desc->q, n, mk, e->stream));
      e->rhs_add = e->q;
      e->ell = e->q;                      // objective b'*x (linearprogram.m:178) reads it as the dot vector
      e->planK = gemv_t_plan(n, n, e->ldK);
      E_TRY(e->mem.alloc(&e->partK, e->planK.part_elems(1)));
      if (qp) {  // the objective 1/2 x'Px + q'x + r needs P*x
        e->planSq = gemv_t_plan(n, n, e->ldP);
        E_TRY(e->mem.alloc(&e->partSq, e->planSq.part_elems(1)));
      }
      e->xsolve = ADMM_XSOLVE_INVERSE;  // a GEMV by construction
      break;
    }
    case ADMM_PROB_TV2D: {
      // 2-D anisotropic TV of an m x n image (column-major): z, u have 2*m*n entries ([vertical; horizontal])
      if (!desc->s || m <= 0 || n <= 0) return bail(fail(ADMM_E_INVALID, "2-D total variation needs the m x n image in s"));
      if (desc->lambda < 0) return bail(fail(ADMM_E_INVALID, "Given lambda parameter is not a nonnegative number!"));
      const int64_t N = m * n;
      e->tv2_H = m;
      e->tv2_W = n;
      e->m = 2 * N;
      e->n = N;  // length of the CG vectors
      e->a_identity = false;
      e->nA = N;
      e->len = 2 * N;
      e->prox = PROX_SOFT;
      e->rhs_kind = RHS_NONE;
      if (desc->cg_tol <= 0 || desc->cg_tol == 1e-12) e->cg_tol = 1e-11;
      if (desc->cg_maxit <= 0 || desc->cg_maxit == 200) {
        e->cg_maxit = 500;
        e->cg_maxit_auto = true;  // (raised per run to what rho needs: engine_run_tv.hip)
      }
      E_TRY(upload(e->mem, &e->s, desc->s, N, mk, e->stream));
      E_TRY(e->mem.alloc(&e->tv_zB, round_up(2 * N, 2)));
      E_TRY(e->mem.alloc(&e->tv_uB, round_up(2 * N, 2)));
      if (tv2_want_dct) {
        auto tables = [&](int64_t len, DctTables* t) -> int {
          const int32_t L = static_cast<int32_t>(len);
          *t = DctTables{};
          t->odd_pair = -1;
          if (!dct_length_ok(L)) {  // chirp form: an FFT of length M >= 2L - 1 behind every column pair
            const int32_t M = dct_chirp_fft_length(L);
            std::vector<admm_double2> tw(static_cast<size_t>(M / 2)), c4(static_cast<size_t>(L)), ch(static_cast<size_t>(L)),
                hb(static_cast<size_t>(M));
            std::vector<double> lam(static_cast<size_t>(L));
            dct_fill_chirp_tables(L, tw.data(), c4.data(), lam.data(), ch.data(), hb.data());
            double *dtw = nullptr, *dc4 = nullptr, *dlam = nullptr, *dch = nullptr, *dhb = nullptr;
            ADMM_TRY(upload(e->mem, &dtw, reinterpret_cast<const double*>(tw.data()), M, ADMM_MEM_HOST, e->stream));
            ADMM_TRY(upload(e->mem, &dc4, reinterpret_cast<const double*>(c4.data()), 2 * L, ADMM_MEM_HOST, e->stream));
            ADMM_TRY(upload(e->mem, &dlam, lam.data(), L, ADMM_MEM_HOST, e->stream));
            ADMM_TRY(upload(e->mem, &dch, reinterpret_cast<const double*>(ch.data()), 2 * L, ADMM_MEM_HOST, e->stream));
            ADMM_TRY(upload(e->mem, &dhb, reinterpret_cast<const double*>(hb.data()), 2 * M, ADMM_MEM_HOST, e->stream));
            ADMM_HIP_TRY(hipStreamSynchronize(e->stream));  // the host vectors go out of scope
            t->n = L;
            t->bm = M;
            while ((1 << t->log2bm) < M) ++t->log2bm;
            t->tw = reinterpret_cast<const admm_double2*>(dtw);
            t->c4 = reinterpret_cast<const admm_double2*>(dc4);
            t->lam = dlam;
            t->chirp = reinterpret_cast<const admm_double2*>(dch);
            t->hbr = reinterpret_cast<const admm_double2*>(dhb);
            return ADMM_OK;
          }
          std::vector<admm_double2> tw(static_cast<size_t>(L / 2)), c4(static_cast<size_t>(L / 2 + 1));
          std::vector<double> lam(static_cast<size_t>(L));
          dct_fill_tables(L, tw.data(), c4.data(), lam.data());
          double *dtw = nullptr, *dc4 = nullptr, *dlam = nullptr;
          ADMM_TRY(upload(e->mem, &dtw, reinterpret_cast<const double*>(tw.data()), L, ADMM_MEM_HOST, e->stream));
          ADMM_TRY(upload(e->mem, &dc4, reinterpret_cast<const double*>(c4.data()), L + 2, ADMM_MEM_HOST, e->stream));
          ADMM_TRY(upload(e->mem, &dlam, lam.data(), L, ADMM_MEM_HOST, e->stream));
          ADMM_HIP_TRY(hipStreamSynchronize(e->stream));  // the host vectors go out of scope
          t->n = L;
          t->log2n = 0;
          while ((1 << t->log2n) < L) ++t->log2n;
          t->tw = reinterpret_cast<const admm_double2*>(dtw);
          t->c4 = reinterpret_cast<const admm_double2*>(dc4);
          t->lam = dlam;
          return ADMM_OK;
        };
        E_TRY(tables(m, &e->dctH));
        e->tv2_rows_dct = dct_length_ok(n);
        if (e->tv2_rows_dct) E_TRY(tables(n, &e->dctW));
        e->tv2_dct = true;
      }
      break;
    }
    case ADMM_PROB_TOTALVARIATION: {
      // totalvariation.m:122-157: s is the (column) signal, D = spdiags([1 -1],0:1,n,n) is implicit
      const int64_t nn = n > 0 ? n : m;
      if (!desc->s || nn <= 0) return bail(fail(ADMM_E_INVALID, "Argument s is not a vector! (totalvariation.m:197)"));
      if (desc->lambda < 0) return bail(fail(ADMM_E_INVALID, "Given lambda parameter is not a nonnegative number!"));
      e->m = e->n = nn;
      e->a_identity = false;
      e->nA = nn;
      e->len = nn;
      e->prox = PROX_SOFT;
      e->rhs_kind = RHS_NONE;
      E_TRY(upload(e->mem, &e->s, desc->s, nn, mk, e->stream));
      E_TRY(e->mem.alloc(&e->tv_y, round_up(nn, 2)));
      E_TRY(e->mem.alloc(&e->tv_y2, round_up(nn, 2)));
      {
        double* ci = nullptr;
        E_TRY(e->mem.alloc(&ci, (sizeof(Ctrl) + 7) / 8));
        e->ctrl_idle = reinterpret_cast<Ctrl*>(ci);
        E_HIP(hipMemsetAsync(e->ctrl_idle, 0, sizeof(Ctrl), e->stream));
      }
      E_TRY(e->mem.alloc(&e->tv_zB, round_up(nn, 2)));
      E_TRY(e->mem.alloc(&e->tv_uB, round_up(nn, 2)));
      break;
    }
    case ADMM_PROB_LASSO_CONSENSUS: {
      // lasso.m:193-224 + getProxOps.m:383-442: one (D_k, D_k's_k, chol(D_k'D_k + rho*I)) per row slice
      if (!desc->D || !desc->s || m <= 0 || n <= 0) return bail(fail(ADMM_E_INVALID, "lasso needs D (m x n) and s"));
      if (desc->lambda < 0) return bail(fail(ADMM_E_INVALID, "lambda must be a nonnegative real (lasso.m:132)"));
      if (desc->nslices < 1 || !desc->slices) return bail(fail(ADMM_E_INVALID, "consensus lasso needs args.slices"));
      int64_t tot = 0;
      for (int32_t k = 0; k < desc->nslices; ++k) {
        if (desc->slices[k] <= 0) return bail(fail(ADMM_E_INVALID, "empty slice"));
        tot += desc->slices[k];
      }
      if (tot != m)
        return bail(fail(ADMM_E_INVALID, "The number of parallel slices does not match length of x! (errorcheck.m:264)"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_SOFT;
      e->rhs_kind = RHS_NONE;
      e->cons_total = desc->nslices;
      if (e->comm && comm_nranks(e->comm) > 1) {
        double* cnt = nullptr;
        E_TRY(e->mem.alloc(&cnt, 2));
        const double mine = static_cast<double>(desc->nslices);
        E_HIP(hipMemcpyAsync(cnt, &mine, sizeof(double), hipMemcpyHostToDevice, e->stream));
        E_TRY(comm_allreduce_device(e->comm, cnt, 1, e->stream));
        double totd = 0.0;
        E_HIP(hipMemcpyAsync(&totd, cnt, sizeof(double), hipMemcpyDeviceToHost, e->stream));
        E_HIP(hipStreamSynchronize(e->stream));
        e->cons_total = static_cast<int32_t>(totd + 0.5);
      }
      const int64_t ldsrc = desc->ldD ? desc->ldD : m;
      const int64_t ld = round_up(n, 16);
      e->cslices.resize(desc->nslices);
      int64_t r0 = 0;
      for (int32_t k = 0; k < desc->nslices; ++k) {
        ConsSlice& sl = e->cslices[k];
        sl.m = desc->slices[k];
        E_TRY(upload_matrix(e->mem, &sl.D, &sl.ld, desc->D + r0, sl.m, n, ldsrc, mk, e->stream));
        E_TRY(upload(e->mem, &sl.s, desc->s + r0, sl.m, mk, e->stream));
        sl.planN = gemv_n_plan(sl.m, n, sl.ld);
        sl.planT = gemv_t_plan(sl.m, n, sl.ld);
        if (k == 0 || sl.planN.part_elems() > e->planDN.part_elems()) e->planDN = sl.planN;  // largest = buffer size
        if (k == 0 || sl.planT.part_elems(1) > e->planDT.part_elems(1)) e->planDT = sl.planT;
        r0 += sl.m;
      }
      E_TRY(e->mem.alloc(&e->partDN, e->planDN.part_elems()));
      E_TRY(e->mem.alloc(&e->partDT, e->planDT.part_elems(1)));
      int64_t fat_rows = 0;
      for (int32_t k = 0; k < desc->nslices; ++k) {
        ConsSlice& sl = e->cslices[k];
        E_TRY(e->mem.alloc(&sl.Dts, round_up(n, 2)));
        launch_gemv_t(sl.planT, sl.D, sl.s, nullptr, nullptr, 1, e->partDT, nullptr, e->stream);
        launch_sum_partials_t(sl.planT, e->partDT, 1, sl.Dts, round_up(n, 2), nullptr, e->stream);
        double* W = nullptr;
        if (sl.m >= n) {  // getProxOps.m:424, 429-435: chol(D_k'D_k + rho I)
          E_TRY(e->mem.alloc(&W, static_cast<size_t>(ld) * n));
          E_HIP(hipMemsetAsync(W, 0, sizeof(double) * ld * n, e->stream));
          launch_gemm(1, 0, n, n, sl.m, 1.0, sl.D, sl.ld, sl.D, sl.ld, 0.0, W, ld, true, e->stream);
          launch_add_diag(W, n, ld, desc->rho, e->stream);
          E_TRY(build_slice_factor(e, sl.fac, W, n, ld, e->xsolve_requested, nullptr, mk));
        } else {
          // fat slice (rows < columns).  q12, documented deviation: the reference's branch (getProxOps.m:426-430,
          // 1251) shifts the wrong entries of D_k D_k' and is exact for no rho; the engine applies the serial
          // solver's form, chol(D_k D_k'/rho + I) with x = y/rho - D_k'(U\(L\(D_k y)))/rho^2 (lasso.m:172,
          // getProxOps.m:1204) = the Woodbury identity of (D_k'D_k + rho I)^-1 y
          sl.fat = true;
          const int64_t ldf = round_up(sl.m, 16);
          E_TRY(e->mem.alloc(&W, static_cast<size_t>(ldf) * sl.m));
          E_HIP(hipMemsetAsync(W, 0, sizeof(double) * ldf * sl.m, e->stream));
          launch_gemm(0, 1, sl.m, sl.m, n, 1.0 / desc->rho, sl.D, sl.ld, sl.D, sl.ld, 0.0, W, ldf, true, e->stream);
          launch_add_diag(W, sl.m, ldf, 1.0, e->stream);
          E_TRY(build_slice_factor(e, sl.fac, W, sl.m, ldf, e->xsolve_requested, nullptr, mk));
          if (sl.m > fat_rows) fat_rows = sl.m;
        }
      }
      if (fat_rows > 0) {
        E_TRY(e->mem.alloc(&e->tmpA, round_up(fat_rows, 2)));
        E_TRY(e->mem.alloc(&e->tmpB, round_up(fat_rows, 2)));
      }
      e->cldn = round_up(n, 2);
      const size_t K = static_cast<size_t>(desc->nslices);
      E_TRY(e->mem.alloc(&e->cX, K * e->cldn));
      E_TRY(e->mem.alloc(&e->cU, K * e->cldn));
      E_TRY(e->mem.alloc(&e->csums, 2 * e->cldn + 2));  // + the packed scalar of the one-collective exchange
      E_TRY(e->mem.alloc(&e->czc, e->cldn));
      E_TRY(e->mem.alloc(&e->cxave, e->cldn));
      E_TRY(e->mem.alloc(&e->cxaveprev, e->cldn));
      E_TRY(e->mem.alloc(&e->cubar, e->cldn));
      {  // every slice applies an explicit inverse through the lower-triangle kernel: keep one set of partial rows per
         // slice, summed by the exchange kernel itself (launch_cons_gather_sum) instead of K symv_reduce launches
        bool all_half = n >= kSymvHalfMin;
        for (const ConsSlice& sl : e->cslices) all_half = all_half && !sl.fat && sl.fac.mode == ADMM_XSOLVE_INVERSE;
        if (all_half && K * kMaxPartBlocks >= K && ceil_div(n, 128) <= kMaxPartBlocks) {
          const SymvPlan& pl = e->cslices[0].fac.planSy;
          e->cpstride = static_cast<int64_t>(pl.npart_elems());
          E_TRY(e->mem.alloc(&e->csyN, K * static_cast<size_t>(e->cpstride)));
          E_TRY(e->mem.alloc(&e->csyT, K * static_cast<size_t>(e->cpstride)));
          E_HIP(hipMemsetAsync(e->csyN, 0, sizeof(double) * K * e->cpstride, e->stream));
          E_HIP(hipMemsetAsync(e->csyT, 0, sizeof(double) * K * e->cpstride, e->stream));
        }
      }
      // the K slice inverses are each read once per iteration: they share the Infinity-Cache budget of the split policy
      for (ConsSlice& sl : e->cslices)
        if (sl.fac.Minv) sl.fac.planSy.ncached = symv_cached_tiles(sl.fac.planSy, kSymvCacheBytes / (K > 0 ? K : 1));
      if (e->csyN) {  // all slices packed and of one size: their x-solves run as ONE launch
        bool packed = true;
        std::vector<const double*> hp;
        for (const ConsSlice& sl : e->cslices) {
          packed = packed && sl.fac.planSy.packed;
          hp.push_back(sl.fac.Minv);
        }
        if (packed) {
          double* raw = nullptr;
          E_TRY(e->mem.alloc(&raw, hp.size()));  // K pointers in K doubles
          E_HIP(hipMemcpyAsync(raw, hp.data(), sizeof(double*) * hp.size(), hipMemcpyHostToDevice, e->stream));
          E_HIP(hipStreamSynchronize(e->stream));
          e->cMptr = reinterpret_cast<const double**>(raw);
        }
      }
      E_TRY(e->mem.alloc(&e->cY, K * e->cldn));
      E_TRY(e->mem.alloc(&e->cDts, K * e->cldn));
      E_HIP(hipMemsetAsync(e->cDts, 0, sizeof(double) * K * e->cldn, e->stream));
      for (int32_t k = 0; k < desc->nslices; ++k)
        E_HIP(hipMemcpyAsync(e->cDts + k * e->cldn, e->cslices[k].Dts, sizeof(double) * n, hipMemcpyDeviceToDevice,
                             e->stream));
      E_TRY(e->mem.alloc(&e->cobjpart, K * kMaxPartBlocks));
      break;
    }
    default:
      return bail(fail(ADMM_E_INVALID, "Invalid input for problem - not a solver (getProxOps.m:916)"));
  }

  // iterates and scratch
  const int64_t L2 = round_up(e->len, 2), N2 = round_up(e->nA, 2);
  E_TRY(e->mem.alloc(&e->x, N2));
  E_TRY(e->mem.alloc(&e->z, L2));
  E_TRY(e->mem.alloc(&e->u, L2));
  E_TRY(e->mem.alloc(&e->rhs, L2 > N2 ? L2 : N2));
  E_TRY(e->mem.alloc(&e->v, L2));
  E_TRY(e->mem.alloc(&e->uhat, L2));
  E_TRY(e->mem.alloc(&e->zprev, L2));
  E_TRY(e->mem.alloc(&e->uprev, L2));
  if (!e->a_identity && e->problem != ADMM_PROB_TOTALVARIATION && e->problem != ADMM_PROB_TV2D) {
    E_TRY(e->mem.alloc(&e->dz, L2));
    e->ldg = N2;
    E_TRY(e->mem.alloc(&e->g, 3 * N2 + 16));  // + 16 reduction slots: one all-reduce payload
  }
  E_TRY(e->mem.alloc(&e->red, 32));  // packed scalar payloads of the sharded runs
  if (e->xsolve == ADMM_XSOLVE_CG) {
    E_TRY(e->mem.alloc(&e->cg_r, N2));
    E_TRY(e->mem.alloc(&e->cg_p, N2));
    E_TRY(e->mem.alloc(&e->cg_q, N2));
    E_TRY(e->mem.alloc(&e->cg_tmp, N2));
    E_TRY(e->mem.alloc(&e->cg_part, 2 * kMaxPartBlocks));
    double* st = nullptr;
    E_TRY(e->mem.alloc(&st, (sizeof(CgState) + 7) / 8));
    e->cg_st = reinterpret_cast<CgState*>(st);
    E_HIP(hipMemsetAsync(e->cg_st, 0, sizeof(CgState), e->stream));
    E_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->cg_st_host), sizeof(CgState), hipHostMallocDefault));
    double* sk = nullptr;
    E_TRY(e->mem.alloc(&sk, (sizeof(Ctrl) + 7) / 8));
    e->cg_skip = reinterpret_cast<Ctrl*>(sk);
    E_HIP(hipMemsetAsync(e->cg_skip, 0, sizeof(Ctrl), e->stream));
  }
  e->tv_zA = e->z;
  e->tv_uA = e->u;
  E_TRY(e->mem.alloc(&e->part, static_cast<size_t>(S_COUNT) * kMaxPartBlocks));
  E_TRY(e->mem.alloc(&e->objpart, 2 * kMaxPartBlocks));  // the model objective has two residual terms
  {
    double* cd = nullptr;
    E_TRY(e->mem.alloc(&cd, (sizeof(Ctrl) + 7) / 8));
    e->ctrl = reinterpret_cast<Ctrl*>(cd);
    E_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->ctrl_host), sizeof(Ctrl), hipHostMallocDefault));
  }
  // ||c||  (admm.m:650)
  if (e->c) {
    std::vector<double> hc(static_cast<size_t>(e->len));
    E_HIP(hipMemcpyAsync(hc.data(), e->c, sizeof(double) * e->len, hipMemcpyDeviceToHost, e->stream));
    E_HIP(hipStreamSynchronize(e->stream));
    double ss = 0.0;
    for (double vv : hc) ss += vv * vv;
    if (sharded) {  // ||c||^2 = sum over the row shards
      E_HIP(hipMemcpyAsync(e->red, &ss, sizeof(double), hipMemcpyHostToDevice, e->stream));
      E_TRY(comm_allreduce_device(e->comm, e->red, 1, e->stream));
      E_HIP(hipMemcpyAsync(&ss, e->red, sizeof(double), hipMemcpyDeviceToHost, e->stream));
      E_HIP(hipStreamSynchronize(e->stream));
    }
    e->cnorm = std::sqrt(ss);
  }
  E_HIP(hipStreamSynchronize(e->stream));
  e->setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *out = e;
  return ADMM_OK;
#undef E_TRY
#undef E_HIP
}


int admm_engine_fetch(admm_engine* e, int field, double* dst, size_t cap, size_t* written) {
  if (!e || !dst) return fail(ADMM_E_INVALID, "engine/dst is NULL");
  if (!e->has_run && field != ADMM_F_FACTOR) return fail(ADMM_E_INVALID, "no run to fetch results from");
  ADMM_HIP_TRY(hipSetDevice(e->device));
  const size_t steps = static_cast<size_t>(e->last.steps);
  const double* src = nullptr;
  size_t count = 0;
  bool need_vec_hist = false, need_fast = false;
  switch (field) {
    case ADMM_F_XOPT: src = e->x; count = e->nA; break;
    case ADMM_F_ZOPT: src = e->bgen ? e->zt : e->z; count = e->bgen ? e->nBz : e->len; break;
    case ADMM_F_UOPT: src = e->u; count = e->len; break;
    case ADMM_F_XVALS: src = e->xhist; count = e->nA * steps; need_vec_hist = true; break;
    case ADMM_F_ZVALS:
      src = e->bgen ? e->zthist : e->zhist;
      count = (e->bgen ? e->nBz : e->len) * steps;
      need_vec_hist = true;
      break;
    case ADMM_F_UVALS: src = e->uhist; count = e->len * steps; need_vec_hist = true; break;
    case ADMM_F_VVALS:
      src = e->bgen ? e->vthist : e->vhist;
      count = (e->bgen ? e->nBz : e->len) * steps;
      need_vec_hist = true;
      need_fast = true;
      break;
    case ADMM_F_UHATVALS: src = e->uhathist; count = e->len * steps; need_vec_hist = true; need_fast = true; break;
    case ADMM_F_PNORM: src = e->pnorm; count = steps; break;
    case ADMM_F_DNORM: src = e->dnorm; count = steps; break;
    case ADMM_F_PERR: src = e->perr; count = steps; break;
    case ADMM_F_DERR: src = e->derr; count = steps; break;
    case ADMM_F_OBJEVALS: src = e->objv; count = steps; break;
    case ADMM_F_HNORMSQ: src = e->hnorm; count = steps; break;
    case ADMM_F_AVALS: src = e->avals; count = steps; need_fast = true; break;
    case ADMM_F_DVALS: src = e->dvals; count = steps; need_fast = true; break;
    case ADMM_F_RESTARTED: src = e->restarted; count = steps; need_fast = true; break;
    case ADMM_F_WVALS: {  // w = [x; z; rho*u] per iteration (admm.m:678-681)
      if (!e->hist_vectors) return fail(ADMM_E_INVALID, "vector histories were not recorded (options.record_history = 0)");
      const size_t nx = static_cast<size_t>(e->nA), nz = static_cast<size_t>(e->bgen ? e->nBz : e->len),
                   nu = static_cast<size_t>(e->len), nw = nx + nz + nu;
      count = nw * steps;
      if (cap < count) return fail(ADMM_E_CAPACITY, "destination too small");
      if (steps > 0) {
        const size_t pitch = nw * sizeof(double);
        ADMM_HIP_TRY(hipMemcpy2D(dst, pitch, e->xhist, nx * sizeof(double), nx * sizeof(double), steps, hipMemcpyDeviceToHost));
        ADMM_HIP_TRY(hipMemcpy2D(dst + nx, pitch, e->bgen ? e->zthist : e->zhist, nz * sizeof(double), nz * sizeof(double),
                                 steps, hipMemcpyDeviceToHost));
        ADMM_HIP_TRY(hipMemcpy2D(dst + nx + nz, pitch, e->uhist, nu * sizeof(double), nu * sizeof(double), steps,
                                 hipMemcpyDeviceToHost));
        const double rho = e->last_opts.rho;
        for (size_t i = 0; i < steps; ++i)
          for (size_t j = 0; j < nu; ++j) dst[i * nw + nx + nz + j] *= rho;
      }
      if (written) *written = count;
      return ADMM_OK;
    }
    case ADMM_F_CG_ITERS: {
      if (e->xsolve != ADMM_XSOLVE_CG) return fail(ADMM_E_INVALID, "field exists only for xsolve = cg");
      if (cap < 1) return fail(ADMM_E_CAPACITY, "destination too small");
      dst[0] = static_cast<double>(e->cg_total_last);
      if (cap >= 2) dst[1] = static_cast<double>(e->cg_capped_last);  // x-updates that ended on cg_maxit above cg_tol
      if (written) *written = cap >= 2 ? 2 : 1;
      return ADMM_OK;
    }
    case ADMM_F_ZCONSENSUS:
      if (e->problem != ADMM_PROB_LASSO_CONSENSUS) return fail(ADMM_E_INVALID, "field exists only for consensus lasso");
      src = e->czc;
      count = e->nA;
      break;
    case ADMM_F_CONS_X:
    case ADMM_F_CONS_U: {
      if (e->problem != ADMM_PROB_LASSO_CONSENSUS) return fail(ADMM_E_INVALID, "field exists only for consensus lasso");
      const size_t K = e->cslices.size();
      count = static_cast<size_t>(e->nA) * K;
      if (cap < count) return fail(ADMM_E_CAPACITY, "destination too small");
      ADMM_HIP_TRY(hipMemcpy2D(dst, e->nA * sizeof(double), field == ADMM_F_CONS_X ? e->cX : e->cU,
                               e->cldn * sizeof(double), e->nA * sizeof(double), K, hipMemcpyDeviceToHost));
      if (written) *written = count;
      return ADMM_OK;
    }
    case ADMM_F_FACTOR: {
      if (!e->F) return fail(ADMM_E_INVALID, "problem has no cached factor");
      count = static_cast<size_t>(e->nF) * e->nF;
      if (cap < count) return fail(ADMM_E_CAPACITY, "destination too small");
      ADMM_HIP_TRY(hipMemcpy2D(dst, e->nF * sizeof(double), e->F, e->ldF * sizeof(double), e->nF * sizeof(double),
                               e->nF, hipMemcpyDeviceToHost));
      // strictly-upper part of the buffer is not part of the factor
      for (int64_t j = 1; j < e->nF; ++j)
        for (int64_t i = 0; i < j; ++i) dst[i + j * e->nF] = 0.0;
      if (written) *written = count;
      return ADMM_OK;
    }
    default:
      return fail(ADMM_E_INVALID, "unknown result field");
  }
  if (need_vec_hist && !e->hist_vectors)
    return fail(ADMM_E_INVALID, "vector histories were not recorded (options.record_history = 0)");
  if (need_fast && !e->hist_fast) return fail(ADMM_E_INVALID, "field exists only for fast/accelerated ADMM runs");
  if (!src) return fail(ADMM_E_INVALID, "field not available for this run");
  if (cap < count) return fail(ADMM_E_CAPACITY, "destination too small");
  if (count) ADMM_HIP_TRY(hipMemcpy(dst, src, count * sizeof(double), hipMemcpyDeviceToHost));
  if (written) *written = count;
  return ADMM_OK;
}

int admm_memcpy_d2h(void* host_dst, const void* device_src, size_t bytes, void* hip_stream) {
  if ((!host_dst || !device_src) && bytes) return fail(ADMM_E_INVALID, "NULL argument");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  ADMM_HIP_TRY(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, st));
  ADMM_HIP_TRY(hipStreamSynchronize(st));
  return ADMM_OK;
}

int admm_memcpy_h2d(void* device_dst, const void* host_src, size_t bytes, void* hip_stream) {
  if ((!device_dst || !host_src) && bytes) return fail(ADMM_E_INVALID, "NULL argument");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  ADMM_HIP_TRY(hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, st));
  ADMM_HIP_TRY(hipStreamSynchronize(st));
  return ADMM_OK;
}

int admm_engine_info(admm_engine* e, admm_engine_info_t* info) {
  if (!e || !info) return fail(ADMM_E_INVALID, "NULL argument");
  if (info->struct_size != static_cast<int32_t>(sizeof(admm_engine_info_t)))
    return fail(ADMM_E_INVALID, "admm_engine_info_t.struct_size mismatch (ABI version skew)");
  const SliceFactor* f = &e->xfac;
  if (e->problem == ADMM_PROB_LASSO_CONSENSUS && !e->cslices.empty()) f = &e->cslices[0].fac;
  const bool has = f->F != nullptr || f->Minv != nullptr;
  info->xsolve_requested = e->xsolve_requested;
  info->xsolve_used = has ? f->mode : e->xsolve;  // ADMM_XSOLVE_PINV: the caller's Dplus is applied
  info->pinv_used = f->pinv ? 1 : 0;
  info->probed = f->probed ? 1 : 0;
  info->factor_n = has ? f->n : 0;
  info->rank = f->pinv ? f->rank : (has ? f->n : 0);
  info->trsv_blocks = (has && f->mode == ADMM_XSOLVE_TRSV) ? f->trsv.nblk : 0;
  info->jacobi_sweeps = f->jacobi_sweeps;
  info->unwrapped_fused = e->Dp ? 1 : 0;
  info->cond_estimate = has ? f->cond_diag : NAN;
  info->probe_err_inverse = f->probed ? f->err_inv : NAN;
  info->probe_err_trsv = (f->probed || f->err_trsv_one == f->err_trsv_one) ? f->err_trsv : NAN;
  info->probe_err_trsv_one = f->err_trsv_one;
  info->probe_diff = f->probed ? f->probe_diff : NAN;
  // what one iteration's x-solve reads, by cache policy (symv.hip / trsv.hip: tiles below ncached use default loads)
  auto tally = [&](const SliceFactor& s) {
    const int64_t tile_bytes = int64_t{8} * 128 * 128;  // kSyTile = kTsTile = 128
    if (s.mode == ADMM_XSOLVE_INVERSE && s.Minv && s.n >= kSymvHalfMin) {
      const int64_t tiles = symv_tiles(s.planSy);
      const int64_t cached = std::min<int64_t>(tiles, std::max<int64_t>(0, s.planSy.ncached));
      info->xsolve_cacheable_bytes += cached * tile_bytes;
      info->xsolve_stream_bytes += (tiles - cached) * tile_bytes;
    } else if (s.mode == ADMM_XSOLVE_INVERSE && s.Minv) {
      info->xsolve_cacheable_bytes += int64_t{8} * s.n * s.n;  // one wave per column, cache-resident
    } else if (s.mode == ADMM_XSOLVE_TRSV && s.trsv.one) {  // ONE triangle, read by both passes
      const int64_t nt = s.trsv.ntile, tiles = nt * (nt + 1) / 2;
      const int64_t cached = s.trsv.streaming ? std::min<int64_t>(tiles, std::max<int64_t>(0, s.trsv.ncached)) : tiles;
      info->xsolve_cacheable_bytes += 2 * cached * tile_bytes;
      info->xsolve_stream_bytes += 2 * (tiles - cached) * tile_bytes;
    } else if (s.mode == ADMM_XSOLVE_TRSV && s.trsv.Fm) {
      const int64_t nt = s.trsv.ntile, tiles = nt * (nt + 1) / 2;  // per triangle
      const int64_t cached = s.trsv.streaming ? std::min<int64_t>(tiles, std::max<int64_t>(0, s.trsv.ncached)) : tiles;
      info->xsolve_cacheable_bytes += 2 * cached * tile_bytes;
      info->xsolve_stream_bytes += 2 * (tiles - cached) * tile_bytes;
    }
  };
  info->obj_bound_max = e->obj_bound_seen;
  info->obj_form_literal = (e->obj_alt && e->obj_auto && e->obj_gram_bad) ? 1 : 0;
  info->reserved0 = 0;
  info->xsolve_cacheable_bytes = info->xsolve_stream_bytes = 0;
  if (e->problem == ADMM_PROB_LASSO_CONSENSUS) {
    for (const ConsSlice& sl : e->cslices) tally(sl.fac);
  } else if (has) {
    tally(*f);
  }
  return ADMM_OK;
}

int admm_engine_setup_seconds(admm_engine* e, double* seconds) {
  if (!e || !seconds) return fail(ADMM_E_INVALID, "NULL argument");
  *seconds = e->setup_seconds;
  return ADMM_OK;
}

int admm_engine_set_profiling(admm_engine* e, int enabled) {
  if (!e) return fail(ADMM_E_INVALID, "engine is NULL");
  e->profiling = enabled < 0 ? 0xffffffffu : static_cast<uint32_t>(enabled);  // bitmask of (1 << ADMM_K_*)
  return ADMM_OK;
}

int admm_engine_set_profiling_stride(admm_engine* e, int stride) {
  if (!e || stride < 1) return fail(ADMM_E_INVALID, "bad argument");
  e->prof_stride = stride;
  return ADMM_OK;
}

int admm_engine_kernel_time(admm_engine* e, int which, double* total_ms, int64_t* launches) {
  if (!e || which < 0 || which >= ADMM_K_COUNT) return fail(ADMM_E_INVALID, "bad argument");
  if (total_ms) *total_ms = e->timers[which].total_ms;
  if (launches) *launches = e->timers[which].launches;
  return ADMM_OK;
}

void admm_engine_destroy(admm_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  free_hist(e);
  for (auto& t : e->timers)
    for (hipEvent_t ev : t.ev) (void)hipEventDestroy(ev);
  e->mem.release();
  if (e->ctrl_host) (void)hipHostFree(e->ctrl_host);
  if (e->cg_st_host) (void)hipHostFree(e->cg_st_host);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

}  // extern "C"
