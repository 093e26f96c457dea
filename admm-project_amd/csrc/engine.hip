// engine.hip -- C ABI (include/admm_engine.h) and the host side of the device-resident
// ADMM loop.  create() = the reference solver's one-time setup + getproxops (lasso.m:160-192,
// lad.m:129-137, huberfit.m:161-169, linearsvm.m:183-217 + unwrappedadmm.m:76-92,
// quadraticprogram.m:210-232, basispursuit.m:116-127); run() = admm.m:252-767.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "kernels.h"
#include "loop_kernels.h"
#include "tv.h"
#include "tv2d.h"
#include "dct.h"
#include "consensus.h"
#include "cg.h"

namespace admm {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

struct DevMem {  // owns every device allocation of an engine
  std::vector<void*> ptrs;
  int alloc(double** out, size_t elems) {
    void* p = nullptr;
    if (elems == 0) elems = 1;
    hipError_t e = hipMalloc(&p, elems * sizeof(double));
    if (e != hipSuccess)
      return fail(ADMM_E_DEVICE, std::string("hipMalloc(") + std::to_string(elems * sizeof(double)) +
                                     " B): " + hipGetErrorString(e));
    ptrs.push_back(p);
    *out = static_cast<double*>(p);
    return ADMM_OK;
  }
  void release() {
    for (void* p : ptrs) (void)hipFree(p);
    ptrs.clear();
  }
};

struct KTimer {  // per-kernel-class HIP event timing on the engine's stream (bench roofline leg)
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
  double total_ms = 0.0;
  int64_t launches = 0;
};

}  // namespace admm

using namespace admm;

struct admm_comm;  // comm.hip
namespace admm {
int comm_allreduce_device(admm_comm* comm, double* buf, size_t count, hipStream_t stream);
int comm_nranks(admm_comm* comm);
int comm_rank(admm_comm* comm);
}  // namespace admm

struct admm_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  DevMem mem;
  admm_comm* comm = nullptr;

  int problem = 0;
  int64_t m = 0, n = 0;  // D is m x n (local rows)
  int64_t nA = 0;        // length of x
  int64_t len = 0;       // nB = length of z, u, c (local rows when sharded)
  int64_t len_global = 0;  // rows over all ranks (0 = not sharded / A = I)
  double* red = nullptr;   // 32 doubles: packed scalar all-reduce payloads
  bool a_identity = true;
  int prox = PROX_SOFT;
  int rhs_kind = RHS_NONE;
  int xsolve = ADMM_XSOLVE_TRSV;
  bool fat = false;      // lasso with m < n (getProxOps.m:1201-1205)
  double lambda = 0, C = 0, rconst = 0, rho_factor = 1.0;
  int loss = 0;

  // data
  double* D = nullptr;
  int64_t ldD = 0;
  double *s = nullptr, *ell = nullptr, *q = nullptr, *lb = nullptr, *ub = nullptr;
  double* c = nullptr;      // constraint vector (alias of s) or null
  double cnorm = 0.0;
  double* rhs_add = nullptr;  // Dts (lasso) or q (QP)
  double* Pmat = nullptr;     // QP P (for the objective) or BP projector
  int64_t ldP = 0;
  double* Kmat = nullptr;     // LP / standard-form QP: x = K*y + k0 (Schur-reduced KKT solve)
  int64_t ldK = 0;
  double* k0 = nullptr;
  GemvTPlan planK{};
  double* partK = nullptr;

  // cached factor
  double* F = nullptr;  // lower Cholesky factor, nF x nF
  int64_t nF = 0, ldF = 0;
  double* dinv = nullptr;
  double* Minv = nullptr;  // explicit inverse (symmetric, full), ld = ldMinv (tile-padded, zeros outside)
  int64_t ldMinv = 0;
  TrsvPlan trsv{};
  double* trsv_work = nullptr;

  // GEMV plans + partial buffers
  GemvNPlan planDN{};   // D*x
  GemvTPlan planDT{};   // D'*v
  GemvTPlan planSq{};   // square symmetric nA x nA GEMV (Minv or P) run as column dots: M*v == M'*v
  SymvPlan planSy{};    // Minv applied from its lower triangle only (half the bytes)
  double *syN = nullptr, *syT = nullptr;
  bool sy_half = false;
  bool sy_split = false;  // multi-GPU: split the tiles of the x-solve over the ranks (decided by measurement at create)
  double *partDN = nullptr, *partDT = nullptr, *partSq = nullptr;

  // iterates
  double *x = nullptr, *z = nullptr, *u = nullptr, *rhs = nullptr, *dz = nullptr, *g = nullptr;
  int64_t ldg = 0;
  double *v = nullptr, *uhat = nullptr, *zprev = nullptr, *uprev = nullptr;
  double *tmpA = nullptr, *tmpB = nullptr;  // fat lasso scratch (m and n long)
  // total variation: forward-sweep intermediate, ping-pong partners of z/u, LDL' pivot prefix
  int64_t tv2_H = 0, tv2_W = 0;  // 2-D TV image shape
  Ctrl* ctrl_idle = nullptr;     // an all-zero control block for clean-up launches after the loop has stopped
  bool tv2_dct = false;          // spectral (DCT) x-update instead of CG: both sides a power of two (dct.h)
  DctTables dctH{}, dctW{};
  double* tv_y2 = nullptr;  // ping-pong partner of tv_y (fused iteration kernel)
  double *tv_y = nullptr, *tv_zA = nullptr, *tv_uA = nullptr, *tv_zB = nullptr, *tv_uB = nullptr;
  double* tv_bprefix = nullptr;
  size_t tv_bprefix_cap = 0;
  // matrix-free x-update (xsolve = cg)
  double cg_tol = 1e-12;
  int32_t cg_maxit = 200;
  bool cg_shift_is_rho = false;
  double *cg_r = nullptr, *cg_p = nullptr, *cg_q = nullptr, *cg_tmp = nullptr, *cg_part = nullptr;
  CgState* cg_st = nullptr;
  Ctrl* cg_skip = nullptr;  // Ctrl-shaped block whose .stop mirrors (CG converged || ctrl->stop): skips the operator kernels
  int cg_chunk = 8;         // inner iterations enqueued between polls: follows the last solve's count
  CgState* cg_st_host = nullptr;  // pinned
  int64_t cg_total_last = 0;
  // consensus lasso (getProxOps.m:383-442, 1217-1343)
  std::vector<ConsSlice> cslices;
  int32_t cons_total = 0;  // slicenum over all ranks
  double *cX = nullptr, *cU = nullptr, *csums = nullptr, *czc = nullptr, *cxave = nullptr, *cxaveprev = nullptr,
         *cubar = nullptr, *cy = nullptr, *cobjpart = nullptr;
  int64_t cldn = 0;
  // model problem (getProxOps.m:60-95) and caller-supplied prox operators: split z-update (PROX_GIVEN)
  SliceFactor zfac{};        // cached factor of QtQ + rho*I (zminModel, getProxOps.m:1005-1012)
  bool has_xfac = true;      // an engine-native x-update exists (false: model created without PtP)
  bool has_zfac = false;
  double* qz = nullptr;      // Qts
  double *xext = nullptr, *zext = nullptr, *xh = nullptr, *rz = nullptr;
  double* D2 = nullptr;      // the matrix Q of the model objective (model.m:134)
  int64_t ldD2 = 0, m2 = 0;
  double* s2 = nullptr;
  GemvNPlan planD2N{};
  double* partD2N = nullptr;
  admm_prox_callback xcb = nullptr, zcb = nullptr;
  admm_obj_callback ocb = nullptr;
  void *xuser = nullptr, *zuser = nullptr, *ouser = nullptr;
  double* part = nullptr;     // [S_COUNT][kMaxPartBlocks]
  double* objpart = nullptr;  // [kMaxPartBlocks]
  Ctrl* ctrl = nullptr;
  Ctrl* ctrl_host = nullptr;  // pinned

  // histories of the last run
  int32_t hist_cap = 0;
  bool hist_vectors = false, hist_fast = false;
  double *xhist = nullptr, *zhist = nullptr, *uhist = nullptr, *vhist = nullptr, *uhathist = nullptr;
  double *pnorm = nullptr, *dnorm = nullptr, *perr = nullptr, *derr = nullptr, *objv = nullptr, *hnorm = nullptr,
         *avals = nullptr, *dvals = nullptr, *restarted = nullptr;
  std::vector<void*> hist_ptrs;
  admm_options last_opts{};
  admm_run_summary last{};
  bool has_run = false;
  double setup_seconds = 0.0;

  uint32_t profiling = 0;  // bit k set: time kernel class k with HIP events
  KTimer timers[ADMM_K_COUNT];
};

namespace {

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
  e->xhist = e->zhist = e->uhist = e->vhist = e->uhathist = nullptr;
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

struct TimerScope {
  admm_engine* e;
  int which;
  bool on;
  size_t slot = 0;
  TimerScope(admm_engine* eng, int w) : e(eng), which(w), on((eng->profiling >> w) & 1u) {
    if (!on) return;
    KTimer& t = e->timers[which];
    if (t.used + 2 > t.ev.size()) {
      for (int k = 0; k < 2; ++k) {
        hipEvent_t ev;
        (void)hipEventCreate(&ev);
        t.ev.push_back(ev);
      }
    }
    slot = t.used;
    t.used += 2;
    (void)hipEventRecord(t.ev[slot], e->stream);
  }
  ~TimerScope() {
    if (!on) return;
    (void)hipEventRecord(e->timers[which].ev[slot + 1], e->stream);
  }
};

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
// W (nF x nF, ld) holds an SPD matrix in its lower triangle -> F (in place), dinv, optionally Minv.
int factorize(admm_engine* e, double* W, int64_t nF, int64_t ld, const double* Lgiven, int memkind) {
  e->nF = nF;
  e->ldF = ld;
  e->F = W;
  const int64_t nblk = ceil_div(nF, 64);
  ADMM_TRY(e->mem.alloc(&e->dinv, static_cast<size_t>(nblk) * 64 * 64));
  if (Lgiven) {
    ADMM_HIP_TRY(hipMemsetAsync(W, 0, sizeof(double) * ld * nF, e->stream));
    ADMM_HIP_TRY(hipMemcpy2DAsync(W, ld * sizeof(double), Lgiven, nF * sizeof(double), nF * sizeof(double), nF,
                                  memkind == ADMM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                  e->stream));
    launch_trtri_diag(W, nF, ld, e->dinv, e->stream);
  } else {
    double* infod = nullptr;
    ADMM_TRY(e->mem.alloc(&infod, 1));
    int32_t* info_dev = reinterpret_cast<int32_t*>(infod);
    ADMM_TRY(cholesky_lower(W, nF, ld, info_dev, e->dinv, e->stream));
    int32_t info = 0;
    ADMM_HIP_TRY(hipMemcpyAsync(&info, info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    if (info != 0)
      return fail(ADMM_E_NUMERIC, "Cholesky failed: matrix must be positive definite (pivot " +
                                      std::to_string(info) + ")");
  }
  if (e->xsolve == ADMM_XSOLVE_INVERSE) {
    double* X = nullptr;
    ADMM_TRY(e->mem.alloc(&X, static_cast<size_t>(ld) * nF));
    ADMM_TRY(trtri_lower_from_diag(W, nF, ld, e->dinv, X, ld, e->stream));
    // Minv is stored padded to whole 128x128 tiles (zeros outside nF x nF): symv.hip has no edge path
    e->planSy = symv_plan(nF);
    const int64_t ldM = e->planSy.npad;
    e->ldMinv = ldM;
    ADMM_TRY(e->mem.alloc(&e->Minv, static_cast<size_t>(ldM) * ldM));
    ADMM_HIP_TRY(hipMemsetAsync(e->Minv, 0, sizeof(double) * ldM * ldM, e->stream));
    // Minv = X' * X  (lower tiles, then mirrored)
    launch_gemm(1, 0, nF, nF, nF, 1.0, X, ld, X, ld, 0.0, e->Minv, ldM, true, e->stream);
    launch_symmetrize_lower(e->Minv, nF, ldM, e->stream);
    ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
    // X is no longer needed
    (void)hipFree(X);
    for (auto& p : e->mem.ptrs)
      if (p == X) p = nullptr;
    e->planSq = gemv_t_plan(nF, nF, ldM);
    ADMM_TRY(e->mem.alloc(&e->partSq, e->planSq.part_elems(1)));
    // lower-triangle form only where bandwidth matters: a 128x128 tile is 32 dependent panel steps of one
    // wave, so for small n the latency of that chain exceeds the whole column-dot GEMV (n = 400: 18k -> 2xk it/s)
    e->sy_half = nF >= 1536;
    if (e->sy_half) {
      ADMM_TRY(e->mem.alloc(&e->syN, e->planSy.npart_elems()));
      ADMM_TRY(e->mem.alloc(&e->syT, e->planSy.tpart_elems()));
      // zero once: with the tiles split over ranks, the slots of foreign tiles are never written
      ADMM_HIP_TRY(hipMemsetAsync(e->syN, 0, sizeof(double) * e->planSy.npart_elems(), e->stream));
      ADMM_HIP_TRY(hipMemsetAsync(e->syT, 0, sizeof(double) * e->planSy.tpart_elems(), e->stream));
      // Multi-GPU: splitting the tiles over the ranks removes t*(1 - 1/N) of streaming time per x-solve and adds
      // one all-reduce of n doubles.  Decide with the latency this communicator actually has (measured here, the
      // mean over the ranks so that every rank takes the same decision): t = 4*npad^2 bytes at ~5.5 TB/s.
      const int nr = e->comm ? comm_nranks(e->comm) : 1;
      if (nr > 1) {
        double* probe = e->syN;  // any device buffer of >= nF + 1 doubles
        for (int k = 0; k < 3; ++k) ADMM_TRY(comm_allreduce_device(e->comm, probe, static_cast<size_t>(nF), e->stream));
        ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
        const auto c0 = std::chrono::steady_clock::now();
        const int reps = 10;
        for (int k = 0; k < reps; ++k) ADMM_TRY(comm_allreduce_device(e->comm, probe, static_cast<size_t>(nF), e->stream));
        ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
        double lat_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count() * 1e6 / reps;
        ADMM_HIP_TRY(hipMemcpyAsync(probe, &lat_us, sizeof(double), hipMemcpyHostToDevice, e->stream));
        ADMM_TRY(comm_allreduce_device(e->comm, probe, 1, e->stream));
        ADMM_HIP_TRY(hipMemcpyAsync(&lat_us, probe, sizeof(double), hipMemcpyDeviceToHost, e->stream));
        ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
        lat_us /= nr;
        const double t_us = 4.0 * static_cast<double>(e->planSy.npad) * static_cast<double>(e->planSy.npad) / 5.5e6;
        e->sy_split = t_us * (1.0 - 1.0 / nr) > 1.15 * lat_us;
        if (const char* f = std::getenv("ADMM_HIP_XSPLIT")) e->sy_split = f[0] == '1';  // tests force either form
        ADMM_HIP_TRY(hipMemsetAsync(e->syN, 0, sizeof(double) * e->planSy.npart_elems(), e->stream));
      }
    }
  } else {
    double* dv = e->dinv;
    ADMM_TRY(trsv_build(W, nF, ld, &dv, &e->trsv, e->stream));
    ADMM_TRY(e->mem.alloc(&e->trsv_work, trsv_workspace_elems(e->trsv)));
  }
  return ADMM_OK;
}

// x = Minv*y from the lower triangle; on a sharded engine every rank streams 1/N of the tiles and ONE
// all-reduce of n doubles assembles x (the only per-iteration collective of the cached-factor lasso loop)
static int symv_apply(admm_engine* e, const double* y, double* out) {
  const int nr = e->comm ? comm_nranks(e->comm) : 1;
  if (nr > 1 && e->sy_split) {
    launch_symv_lower(e->planSy, e->Minv, e->ldMinv, y, e->syN, e->syT, out, e->ctrl, e->stream, comm_rank(e->comm), nr);
    return comm_allreduce_device(e->comm, out, static_cast<size_t>(e->nF), e->stream);
  }
  launch_symv_lower(e->planSy, e->Minv, e->ldMinv, y, e->syN, e->syT, out, e->ctrl, e->stream);
  return ADMM_OK;
}

// out = F^-T F^-1 y  (out has nF elements).  For the INVERSE path the result is left as
// chunk partials in partSq unless `materialize`.
int solve_factor(admm_engine* e, const double* y, double* out) {
  if (e->xsolve == ADMM_XSOLVE_INVERSE && e->sy_half) {
    return symv_apply(e, y, out);
  } else if (e->xsolve == ADMM_XSOLVE_INVERSE) {  // small n: one wave per column, direct result
    launch_symv_small(e->Minv, e->nF, e->ldMinv, y, out, e->ctrl, e->stream);
  } else {
    launch_trsv_pair(e->trsv, y, out, e->trsv_work, e->ctrl, e->stream);
  }
  return ADMM_OK;
}


// ---- consensus slices: one cached factor per slice ----------------------------------------
int build_slice_factor(admm_engine* e, SliceFactor& f, double* W, int64_t n, int64_t ld) {
  f.F = W;
  f.n = n;
  f.ld = ld;
  const int64_t nblk = ceil_div(n, 64);
  ADMM_TRY(e->mem.alloc(&f.dinv, static_cast<size_t>(nblk) * 64 * 64));
  double* infod = nullptr;
  ADMM_TRY(e->mem.alloc(&infod, 1));
  int32_t* info_dev = reinterpret_cast<int32_t*>(infod);
  ADMM_TRY(cholesky_lower(W, n, ld, info_dev, f.dinv, e->stream));
  int32_t info = 0;
  ADMM_HIP_TRY(hipMemcpyAsync(&info, info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  if (info != 0)
    return fail(ADMM_E_NUMERIC, "Cholesky failed: matrix must be positive definite (pivot " + std::to_string(info) + ")");
  if (e->xsolve == ADMM_XSOLVE_INVERSE) {
    double* X = nullptr;
    ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&X), sizeof(double) * ld * n));
    int rc = trtri_lower_from_diag(W, n, ld, f.dinv, X, ld, e->stream);
    f.planSy = symv_plan(n);
    f.ldM = f.planSy.npad;
    if (rc == ADMM_OK) rc = e->mem.alloc(&f.Minv, static_cast<size_t>(f.ldM) * f.ldM);
    if (rc == ADMM_OK) {
      (void)hipMemsetAsync(f.Minv, 0, sizeof(double) * f.ldM * f.ldM, e->stream);
      launch_gemm(1, 0, n, n, n, 1.0, X, ld, X, ld, 0.0, f.Minv, f.ldM, true, e->stream);
      launch_symmetrize_lower(f.Minv, n, f.ldM, e->stream);
    }
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(X);
    ADMM_TRY(rc);
    if (n >= 1536 && !e->syN) {  // partial-sum buffers of the lower-triangle kernel, shared by all slices (same n)
      ADMM_TRY(e->mem.alloc(&e->syN, f.planSy.npart_elems()));
      ADMM_TRY(e->mem.alloc(&e->syT, f.planSy.tpart_elems()));
    }
  } else {
    double* dv = f.dinv;
    ADMM_TRY(trsv_build(W, n, ld, &dv, &f.trsv, e->stream));
    ADMM_TRY(e->mem.alloc(&f.work, trsv_workspace_elems(f.trsv)));
  }
  return ADMM_OK;
}

void apply_slice_factor(admm_engine* e, const SliceFactor& f, const double* y, double* out) {
  if (e->xsolve == ADMM_XSOLVE_INVERSE) {
    if (f.n >= 1536) launch_symv_lower(f.planSy, f.Minv, f.ldM, y, e->syN, e->syT, out, e->ctrl, e->stream);
    else launch_symv_small(f.Minv, f.n, f.ldM, y, out, e->ctrl, e->stream);
  } else {
    launch_trsv_pair(f.trsv, y, out, f.work, e->ctrl, e->stream);
  }
}

}  // namespace

// ======================================================================================
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
  E_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
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
  // AUTO: the literal two triangular solves are 2*n/64 dependent launches (latency-bound, ~1 ms at n = 10^4);
  // beyond a few diagonal blocks the one-pass symmetric GEMV with the explicit inverse is the faster form
  if (xs == ADMM_XSOLVE_AUTO) xs = (n > 256) ? ADMM_XSOLVE_INVERSE : ADMM_XSOLVE_TRSV;
  if (xs == ADMM_XSOLVE_CALLBACK && desc->problem != ADMM_PROB_LAD)
    return bail(fail(ADMM_E_UNSUPPORTED, "xsolve=callback is the generic A = D engine: use ADMM_PROB_LAD with D = A, s = c"));
  if (xs == ADMM_XSOLVE_CALLBACK && e->comm && comm_nranks(e->comm) > 1)
    return bail(fail(ADMM_E_UNSUPPORTED, "prox callbacks are not supported on row-sharded engines"));
  // 2-D TV: AUTO = the direct spectral solve when both sides are powers of two, else (or on request) warm-started
  // CG; the CG vectors are allocated either way (one of them is the transposition scratch of the spectral solve)
  const bool tv2_want_dct = desc->problem == ADMM_PROB_TV2D && desc->xsolve != ADMM_XSOLVE_CG &&
                            dct_length_ok(desc->m) && dct_length_ok(desc->n);
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
      if (!desc->D || !desc->s || m <= 0 || n <= 0) return bail(fail(ADMM_E_INVALID, "lasso needs D (m x n) and s"));
      if (desc->lambda < 0) return bail(fail(ADMM_E_INVALID, "lambda must be a nonnegative real (lasso.m:132)"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_SOFT;
      e->rhs_kind = RHS_RHO_DTS;
      e->fat = m_global < n;
      if (sharded && e->fat)
        return bail(fail(ADMM_E_UNSUPPORTED, "row-sharded lasso needs a tall matrix (global m >= n)"));
      E_TRY(upload_matrix(e->mem, &e->D, &e->ldD, desc->D, m, n, desc->ldD ? desc->ldD : m, mk, e->stream));
      E_TRY(upload(e->mem, &e->s, desc->s, m, mk, e->stream));
      e->planDN = gemv_n_plan(m, n, e->ldD);
      e->planDT = gemv_t_plan(m, n, e->ldD);
      E_TRY(e->mem.alloc(&e->partDN, e->planDN.part_elems()));
      E_TRY(e->mem.alloc(&e->partDT, e->planDT.part_elems(3)));
      // Dts = D'*s   lasso.m:160
      E_TRY(e->mem.alloc(&e->rhs_add, round_up(n, 2)));
      launch_gemv_t(e->planDT, e->D, e->s, nullptr, nullptr, 1, e->partDT, nullptr, e->stream);
      launch_sum_partials_t(e->planDT, e->partDT, 1, e->rhs_add, round_up(n, 2), nullptr, e->stream);
      if (sharded) E_TRY(comm_allreduce_device(e->comm, e->rhs_add, n, e->stream));  // sum_g D_g'*s_g
      if (e->xsolve == ADMM_XSOLVE_CG) {  // matrix-free: nothing n x n is ever formed
        if (e->fat) return bail(fail(ADMM_E_UNSUPPORTED, "xsolve=cg needs a tall matrix (m >= n)"));
        e->cg_shift_is_rho = true;
        E_TRY(e->mem.alloc(&e->tmpA, round_up(m, 2)));
        break;
      }
      const int64_t nF = e->fat ? m : n;
      const int64_t ld = round_up(nF, 16);
      double* W = nullptr;
      E_TRY(e->mem.alloc(&W, static_cast<size_t>(ld) * nF));
      if (!desc->L) {
        E_HIP(hipMemsetAsync(W, 0, sizeof(double) * ld * nF, e->stream));
        if (!e->fat) {  // lasso.m:168  chol(D'*D + rho*I)
          launch_gemm(1, 0, n, n, m, 1.0, e->D, e->ldD, e->D, e->ldD, 0.0, W, ld, true, e->stream);
          // W = sum_g D_g'*D_g  (unwrappedadmm.m:118-122); one-time, bandwidth-bound all-reduce
          if (sharded) E_TRY(comm_allreduce_device(e->comm, W, static_cast<size_t>(ld) * n, e->stream));
          launch_add_diag(W, n, ld, desc->rho, e->stream);
        } else {  // lasso.m:172  chol(1/rho*(D*D') + I)
          launch_gemm(0, 1, m, m, n, 1.0 / desc->rho, e->D, e->ldD, e->D, e->ldD, 0.0, W, ld, true, e->stream);
          launch_add_diag(W, m, ld, 1.0, e->stream);
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
      if (!desc->D || m <= 0 || n <= 0) return bail(fail(ADMM_E_INVALID, "problem needs D (m x n)"));
      if (!svm && !desc->s) return bail(fail(ADMM_E_INVALID, "LAD/Huber need the signal vector s"));
      if (svm && !desc->ell) return bail(fail(ADMM_E_INVALID, "linear SVM needs the label vector ell"));
      if (m_global < n && xs != ADMM_XSOLVE_CALLBACK)
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
      const int64_t ld = round_up(n, 16);
      double* W = nullptr;
      E_TRY(e->mem.alloc(&W, static_cast<size_t>(ld) * n));
      if (!desc->L) {  // lad.m:134  chol(D'*D,'lower') (un-shifted; also D^+ = (D'D)^-1 D' for the SVM)
        E_HIP(hipMemsetAsync(W, 0, sizeof(double) * ld * n, e->stream));
        launch_gemm(1, 0, n, n, m, 1.0, e->D, e->ldD, e->D, e->ldD, 0.0, W, ld, true, e->stream);
        // W = sum_g D_g'*D_g  (unwrappedadmm.m:96-123)
        if (sharded) E_TRY(comm_allreduce_device(e->comm, W, static_cast<size_t>(ld) * n, e->stream));
      }
      E_TRY(factorize(e, W, n, ld, desc->L, mk));
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
      break;
    }
    case ADMM_PROB_BASISPURSUIT: {
      if (!desc->P || !desc->q || n <= 0) return bail(fail(ADMM_E_INVALID, "basis pursuit needs P (n x n) and q"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_SOFT;
      e->rhs_kind = RHS_DIFF;
      E_TRY(upload_matrix(e->mem, &e->Pmat, &e->ldP, desc->P, n, n, n, mk, e->stream));
      E_TRY(upload(e->mem, &e->q, desc->q, n, mk, e->stream));
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
        E_TRY(build_slice_factor(e, e->zfac, W, n, ld));
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
      if (!desc->K || !desc->k0 || !desc->q || n <= 0 || (qp && !desc->P))
        return bail(fail(ADMM_E_INVALID, qp ? "standard-form QP needs P, q and the reduced KKT map K, k0"
                                            : "linear program needs b (as q) and the reduced KKT map K, k0"));
      e->a_identity = true;
      e->nA = n;
      e->len = n;
      e->prox = PROX_POS;                 // getProxOps.m:1381, 1425
      e->rhs_kind = RHS_RHO_MINUS_Q;      // y = rho*(z-u) - b   (getProxOps.m:1363, 1410)
      E_TRY(upload_matrix(e->mem, &e->Kmat, &e->ldK, desc->K, n, n, n, mk, e->stream));
      E_TRY(upload(e->mem, &e->k0, desc->k0, n, mk, e->stream));
      E_TRY(upload(e->mem, &e->q, desc->q, n, mk, e->stream));
      e->rhs_add = e->q;
      e->ell = e->q;                      // objective b'*x (linearprogram.m:178) reads it as the dot vector
      e->planK = gemv_t_plan(n, n, e->ldK);
      E_TRY(e->mem.alloc(&e->partK, e->planK.part_elems(1)));
      if (qp) {  // the objective 1/2 x'Px + q'x + r needs P*x
        E_TRY(upload_matrix(e->mem, &e->Pmat, &e->ldP, desc->P, n, n, n, mk, e->stream));
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
      if (desc->cg_maxit <= 0 || desc->cg_maxit == 200) e->cg_maxit = 500;
      E_TRY(upload(e->mem, &e->s, desc->s, N, mk, e->stream));
      E_TRY(e->mem.alloc(&e->tv_zB, round_up(2 * N, 2)));
      E_TRY(e->mem.alloc(&e->tv_uB, round_up(2 * N, 2)));
      if (tv2_want_dct) {
        auto tables = [&](int64_t len, DctTables* t) -> int {
          const int32_t L = static_cast<int32_t>(len);
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
        E_TRY(tables(n, &e->dctW));
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
        if (sl.m < n)  // q12: the reference's fat-slice branch indexes the wrong diagonal; not reproduced
          return bail(fail(ADMM_E_UNSUPPORTED, "consensus lasso needs tall slices (rows per slice >= columns), see q12"));
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
      for (int32_t k = 0; k < desc->nslices; ++k) {
        ConsSlice& sl = e->cslices[k];
        E_TRY(e->mem.alloc(&sl.Dts, round_up(n, 2)));
        launch_gemv_t(sl.planT, sl.D, sl.s, nullptr, nullptr, 1, e->partDT, nullptr, e->stream);
        launch_sum_partials_t(sl.planT, e->partDT, 1, sl.Dts, round_up(n, 2), nullptr, e->stream);
        double* W = nullptr;
        E_TRY(e->mem.alloc(&W, static_cast<size_t>(ld) * n));
        E_HIP(hipMemsetAsync(W, 0, sizeof(double) * ld * n, e->stream));
        launch_gemm(1, 0, n, n, sl.m, 1.0, sl.D, sl.ld, sl.D, sl.ld, 0.0, W, ld, true, e->stream);
        launch_add_diag(W, n, ld, desc->rho, e->stream);
        E_TRY(build_slice_factor(e, sl.fac, W, n, ld));
      }
      e->cldn = round_up(n, 2);
      const size_t K = static_cast<size_t>(desc->nslices);
      E_TRY(e->mem.alloc(&e->cX, K * e->cldn));
      E_TRY(e->mem.alloc(&e->cU, K * e->cldn));
      E_TRY(e->mem.alloc(&e->csums, 2 * e->cldn));
      E_TRY(e->mem.alloc(&e->czc, e->cldn));
      E_TRY(e->mem.alloc(&e->cxave, e->cldn));
      E_TRY(e->mem.alloc(&e->cxaveprev, e->cldn));
      E_TRY(e->mem.alloc(&e->cubar, e->cldn));
      E_TRY(e->mem.alloc(&e->cy, e->cldn));
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

// q-source for CG: D'*(D*v) as gemv_t chunk partials (or, row-sharded, the all-reduced sum in cg_tmp)
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

// 2-D TV: the same CG with the direction update fused into the stencil apply (10 instead of 14 vector
// passes and 3 instead of 5 launches per inner iteration); p ping-pongs between cg_p and cg_tmp
static int cg_solve_tv2d(admm_engine* e, const double* y) {
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
  for (int done_it = 0; done_it < e->cg_maxit;) {
    const int k = (e->cg_maxit - done_it < chunk) ? e->cg_maxit - done_it : chunk;
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

// x <- argmin-free solve of (D'D + shift I) x = y by warm-started CG (cg.hip); polls the device flag
// 2-D TV x-update, direct: x = C2' diag(1/(1 + rho*(lamH_i + lamW_j))) C2 y with the 2-D DCT-II C2 (dct.h).
// Five streaming passes (10 N doubles of traffic); y is overwritten, e->cg_r is the transposition scratch.
static int dct_solve_tv2d(admm_engine* e, double* y) {
  TimerScope ts(e, ADMM_K_XSOLVE);
  const int64_t H = e->tv2_H, W = e->tv2_W;
  launch_dct_cols_forward(y, H, W, e->dctH, e->ctrl, e->stream);                 // along i, in place
  launch_transpose(y, e->cg_r, H, W, e->ctrl, e->stream);                        // -> W x H
  launch_dct_rows_solve(e->cg_r, H, W, e->last_opts.rho, e->dctH, e->dctW, e->ctrl, e->stream);
  launch_transpose(e->cg_r, y, W, H, e->ctrl, e->stream);                        // -> H x W
  launch_dct_cols_inverse(y, e->x, H, W, e->dctH, e->ctrl, e->stream);
  return ADMM_OK;
}

static int cg_solve(admm_engine* e, const double* y) {
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

static int factor_x_update(admm_engine* e, const double** axsrc, int32_t* naxpart, int64_t* axld);

// one x-update (admm.m:501-511) from e->rhs into e->x, or into chunk partials for the fused consumer
static int x_update(admm_engine* e, const double** axsrc, int32_t* naxpart, int64_t* axld) {
  TimerScope ts(e, ADMM_K_XSOLVE);
  *axsrc = e->x;
  *naxpart = 1;
  *axld = 0;
  if (e->xcb) {  // x = xminf(x, z, u, rho), fast ADMM: xminf(x, v, uhat, rho)   (admm.m:502, 506)
    const bool fastalg = e->last_opts.fast != ADMM_FAST_OFF;
    if (e->xcb(e->xuser, e->x, fastalg ? e->v : e->z, fastalg ? e->uhat : e->u, e->last_opts.rho, e->xext, e->nA,
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
        ADMM_TRY(factor_x_update(e, axsrc, naxpart, axld));
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
      ADMM_TRY(factor_x_update(e, axsrc, naxpart, axld));
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
      ADMM_TRY(solve_factor(e, e->g, e->x));
      break;
  }
  return ADMM_OK;
}

// the cached-factor x-update shared by lasso (tall), bounded QP and the model problem
static int factor_x_update(admm_engine* e, const double** axsrc, int32_t* naxpart, int64_t* axld) {
  if (e->xsolve == ADMM_XSOLVE_INVERSE && e->sy_half) {  // x = Minv*y from the lower triangle
    return symv_apply(e, e->rhs, e->x);
  } else if (e->xsolve == ADMM_XSOLVE_INVERSE) {  // small n: one wave per column, direct result
    launch_symv_small(e->Minv, e->nF, e->ldMinv, e->rhs, e->x, e->ctrl, e->stream);
  } else {
    launch_trsv_pair(e->trsv, e->rhs, e->x, e->trsv_work, e->ctrl, e->stream);
  }
  return ADMM_OK;
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
  e->last_opts = o;
  const int alg = o.fast;  // 0, 1 (strong), 2 (weak)
  const bool use_h = o.convtest || o.stopcond == ADMM_STOP_HNORM || o.stopcond == ADMM_STOP_BOTH;
  const int64_t len = e->len, nA = e->nA;
  const int32_t N = o.maxiters;

  // ---- histories
  free_hist(e);
  e->hist_cap = N;
  e->hist_vectors = o.record_history != 0;
  e->hist_fast = alg != 0;
  if (e->hist_vectors) {
    ADMM_TRY(hist_alloc(e, &e->xhist, static_cast<size_t>(nA) * N));
    ADMM_TRY(hist_alloc(e, &e->zhist, static_cast<size_t>(len) * N));
    ADMM_TRY(hist_alloc(e, &e->uhist, static_cast<size_t>(len) * N));
    if (alg != 0) {
      ADMM_TRY(hist_alloc(e, &e->vhist, static_cast<size_t>(len) * N));
      ADMM_TRY(hist_alloc(e, &e->uhathist, static_cast<size_t>(len) * N));
    }
  }
  double** scal[] = {&e->pnorm, &e->dnorm, &e->perr, &e->derr, &e->objv, &e->hnorm, &e->avals, &e->dvals,
                     &e->restarted};
  for (double** p : scal) {
    ADMM_TRY(hist_alloc(e, p, N));
    ADMM_HIP_TRY(hipMemsetAsync(*p, 0, sizeof(double) * N, e->stream));
  }

  // ---- initial iterates (admm.m:252-254) and control block
  auto init_vec = [&](double* dst, const double* src, int64_t cnt) -> int {
    if (src) ADMM_HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * cnt, hipMemcpyHostToDevice, e->stream));
    else ADMM_HIP_TRY(hipMemsetAsync(dst, 0, sizeof(double) * cnt, e->stream));
    return ADMM_OK;
  };
  ADMM_TRY(init_vec(e->x, o.x0, nA));
  ADMM_TRY(init_vec(e->z, o.z0, len));
  ADMM_TRY(init_vec(e->u, o.u0, len));
  ADMM_HIP_TRY(hipMemcpyAsync(e->v, e->z, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));     // admm.m:269
  ADMM_HIP_TRY(hipMemcpyAsync(e->uhat, e->u, sizeof(double) * len, hipMemcpyDeviceToDevice, e->stream));  // admm.m:270
  Ctrl c0{};
  c0.acurr = 1.0;
  c0.aprev = 1.0;
  c0.d = INFINITY;
  c0.dprev = INFINITY;
  *e->ctrl_host = c0;
  ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl, e->ctrl_host, sizeof(Ctrl), hipMemcpyHostToDevice, e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
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
        obj_lasso_gemv = true;
        fa.obj_scale_part = 0.5;
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

  if (e->problem == ADMM_PROB_LASSO_CONSENSUS) {
    if (alg != 0) return fail(ADMM_E_UNSUPPORTED, "fast/accelerated ADMM is not implemented for consensus lasso");
    if (o.relax != 1.0) return fail(ADMM_E_UNSUPPORTED, "relaxation is not implemented for consensus lasso");
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
    while (done < N && !stop_seen) {
      const int32_t batch = (N - done < check_c) ? N - done : check_c;
      for (int32_t b = 0; b < batch; ++b) {
        {
          TimerScope ts(e, ADMM_K_XSOLVE);
          for (int32_t k = 0; k < K; ++k) {  // getProxOps.m:1228-1253
            ConsSlice& sl = e->cslices[k];
            launch_cons_rhs(n, o.rho, e->czc, e->cU + k * ldn, sl.Dts, e->cy, e->ctrl, e->stream);
            apply_slice_factor(e, sl.fac, e->cy, e->cX + k * ldn);
          }
        }
        launch_cons_sum(n, ldn, K, e->cX, e->cU, e->csums, e->ctrl, e->stream);
        if (shard) ADMM_TRY(comm_allreduce_device(e->comm, e->csums, static_cast<size_t>(2 * ldn), e->stream));  // X1
        int nblk = 1;
        {
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
        if (shard) {  // X2: only sum_k ||x_k - xave||^2 and the objective are rank-local sums
          launch_pack_slots(e->part, nblk, e->red, e->ctrl, e->stream);
          ADMM_HIP_TRY(hipMemcpyAsync(e->red + 16, e->red + S_R2, sizeof(double), hipMemcpyDeviceToDevice, e->stream));
          if (o.objevals) launch_pack_sum(e->cobjpart, K * kMaxPartBlocks, e->red + 17, e->ctrl, e->stream);
          ADMM_TRY(comm_allreduce_device(e->comm, e->red + 16, 2, e->stream));
          ADMM_HIP_TRY(hipMemcpyAsync(e->red + S_R2, e->red + 16, sizeof(double), hipMemcpyDeviceToDevice, e->stream));
          fa.slots_reduced = e->red;
          if (o.objevals) fa.objp_reduced = e->red + 17;
        }
        {
          TimerScope ts(e, ADMM_K_FINALIZE);
          launch_finalize(fa, e->stream);
        }
      }
      done += batch;
      if (!o.domaxiters || done >= N) {
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

  if (e->problem == ADMM_PROB_TV2D) {
    if (alg != 0) return fail(ADMM_E_UNSUPPORTED, "fast/accelerated ADMM is not implemented for 2-D total variation");
    if (o.relax != 1.0) return fail(ADMM_E_UNSUPPORTED, "relaxation is not implemented for 2-D total variation");
    const int64_t Npix = e->tv2_H * e->tv2_W;
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
    while (done < N && !stop_seen) {
      const bool a_cur = (done & 1) == 0;  // iteration k reads buffer A when k is even
      ta.z = a_cur ? e->tv_zA : e->tv_zB;
      ta.u = a_cur ? e->tv_uA : e->tv_uB;
      ta.zo = a_cur ? e->tv_zB : e->tv_zA;
      ta.uo = a_cur ? e->tv_uB : e->tv_uA;
      if (done == 0) {  // later right-hand sides come out of the fused z/u pass of the previous iteration
        TimerScope ts(e, ADMM_K_XSOLVE);
        launch_tv2d_rhs(ta, e->rhs, e->ctrl, e->stream);
      }
      // (I + rho*D'D) x = s + rho*D'(z - u): spectral, or warm-started CG (polls the device)
      if (e->tv2_dct) ADMM_TRY(dct_solve_tv2d(e, e->rhs));
      else ADMM_TRY(cg_solve(e, e->rhs));
      int nblk = 1;
      {
        TimerScope ts(e, ADMM_K_PROX);
        launch_tv2d_fused(ta, e->rhs, e->ctrl, &nblk, e->stream);
      }
      fa.nblk = nblk;
      {
        TimerScope ts(e, ADMM_K_FINALIZE);
        launch_finalize(fa, e->stream);
      }
      done += 1;
      // the CG path synchronises inside every solve anyway; the spectral path runs check_tv2 iterations ahead
      // (everything enqueued after the stop flag is a no-op)
      if (!e->tv2_dct || done % check_tv2 == 0 || done == N) {
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
    e->z = (steps & 1) ? e->tv_zB : e->tv_zA;
    e->u = (steps & 1) ? e->tv_uB : e->tv_uA;
    ADMM_HIP_TRY(hipMemcpy(e->cg_st_host, e->cg_st, sizeof(CgState), hipMemcpyDeviceToHost));
    e->cg_total_last = e->cg_st_host->total;
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

  if (e->problem == ADMM_PROB_TOTALVARIATION) {
    if (o.relax != 1.0 && alg != 0)
      return fail(ADMM_E_UNSUPPORTED, "relaxation together with fast ADMM is not implemented for total variation");
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
    // one fused launch per iteration (8 vector passes) when the halo is small; the three-kernel form otherwise
    const bool tv_fused = tv_fused_ok(ta) && std::getenv("ADMM_HIP_TV_UNFUSED") == nullptr;
    double* tv_part = nullptr;  // per-tile partials of the fused kernel (one column per tile)
    if (tv_fused) {
      ta.part_stride = round_up(ceil_div(e->n, ta.ftile), 2);
      ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&tv_part), sizeof(double) * S_COUNT * ta.part_stride));
      ta.part = tv_part;
    }
    struct DevFree {
      void* p;
      ~DevFree() {
        if (p) (void)hipFree(p);
      }
    } tv_part_guard{tv_part};
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
              launch_tv_relax_z(e->tmpA, e->z, e->u, e->n, o.relax, pa.t, e->zext, e->ctrl, e->stream);
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
        if (!o.domaxiters || donef >= N) {
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
    if (tv_fused) {  // the forward sweep of iteration 0; every later one is produced by the fused kernel
      TimerScope ts(e, ADMM_K_XSOLVE);
      ta.z = e->tv_zA;
      ta.u = e->tv_uA;
      ta.y = e->tv_y;
      launch_tv_sweep(ta, false, e->ctrl, e->stream);
    }
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
          launch_tv_fused(ta, e->red, e->ctrl, e->stream);
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
      if (!o.domaxiters || done >= N) {
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
    // iterations executed on the device decide which ping-pong buffer holds the final z, u
    e->z = (steps & 1) ? e->tv_zB : e->tv_zA;
    e->u = (steps & 1) ? e->tv_uB : e->tv_uA;
    if (ta.skip_x && steps > 0) {
      ta.y = ((steps - 1) & 1) ? e->tv_y2 : e->tv_y;
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

  const int nrhs_dual = o.nodualerror ? 1 : 3;
  const bool sharded = e->comm && comm_nranks(e->comm) > 1;
  fa.len_global = e->len_global;
  int check_every = o.check_every > 0 ? o.check_every : (o.domaxiters ? 64 : 8);

  // ---- loop (admm.m:315 tic .. 756 toc)
  const auto tstart = std::chrono::steady_clock::now();
  // rhs of the first x-update from the initial iterates (zx = v = z0, ux = uhat = u0)
  launch_initial_rhs(len, e->rhs_kind, o.rho, e->z, e->u, e->c, e->rhs_add, e->rhs, e->stream);
  if (!e->a_identity) {
    TimerScope ts(e, ADMM_K_GEMV_T);
    launch_gemv_t(e->planDT, e->D, e->rhs, nullptr, nullptr, 1, e->partDT, e->ctrl, e->stream);
    launch_sum_partials_t(e->planDT, e->partDT, 1, e->g, e->ldg, e->ctrl, e->stream);
    if (sharded) ADMM_TRY(comm_allreduce_device(e->comm, e->g, static_cast<size_t>(e->ldg), e->stream));
  }
  // One iteration = a fixed sequence of launches with iteration-independent arguments (the iteration
  // index lives in ctrl->iter), and iterations past a stop condition or past maxiters are no-ops on the
  // device, so a batch of iterations CAN be captured once into a hipGraph and replayed.  Measured on
  // MI355X / ROCm 7.x (profiles/svm_bench.py, profiles/trsv_graph_bench.py) replay is never faster than
  // eager launches on one stream -- SVM 6000x400: 23.0k vs 23.9k it/s; lasso 2000x512 with the 16-launch
  // TRSV x-solve: 3.7k vs 4.9k -- so eager is the default and ADMM_HIP_GRAPH=1 opts in.  Never used when
  // collectives or host callbacks sit inside the iteration, for the CG x-solve (which polls the device
  // between inner iterations), with event timing on (hipEventElapsedTime rejects events recorded by graph
  // nodes: "invalid resource handle"), or under rocprofv3 (which segfaults inside hipGraphLaunch).
  const int64_t heavy = std::max<int64_t>(e->m * e->n, e->nF * e->nF);
  const char* preload = std::getenv("LD_PRELOAD");
  const bool profiler_attached = std::getenv("ROCP_TOOL_LIBRARIES") != nullptr ||
                                 (preload && std::strstr(preload, "rocprof") != nullptr);
  const bool use_graph = std::getenv("ADMM_HIP_GRAPH") != nullptr && !sharded && e->profiling == 0 &&
                         e->xsolve != ADMM_XSOLVE_CG && !profiler_attached && heavy <= (int64_t{32} << 20) &&
                         !e->xcb && !e->zcb && !e->ocb;
  auto enqueue_iteration = [&]() -> int {
    {
      const double* axsrc;
      int32_t naxpart;
      int64_t axld;
      ADMM_TRY(x_update(e, &axsrc, &naxpart, &axld));
      if (!e->a_identity) {  // Ax = D*x (admm.m:535), summed inside the prox kernel
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
          if (e->zcb(e->zuser, zarg, e->z, za.uo, o.rho, e->zext, len, static_cast<void*>(e->stream)) != 0)
            return fail(ADMM_E_INVALID, "the zming callback reported a failure");
        } else {  // zminModel: (QtQ + rho I) \ (Qts + rho*(x + u))   getProxOps.m:1012
          apply_slice_factor(e, e->zfac, e->rz, e->zext);
        }
      }
      {
        TimerScope ts(e, ADMM_K_PROX);
        pa.axsrc = axsrc;
        pa.naxpart = naxpart;
        pa.axld = axld;
        pa.x_out = e->a_identity ? e->x : nullptr;
        launch_prox(pa, e->ctrl, &nblk, e->stream);
      }
      fa.nblk = nblk;
      fa.slots_reduced = nullptr;
      fa.objp_reduced = nullptr;
      const bool shard_rows = sharded && !e->a_identity;  // z, u and the residual sums are row-local
      if (alg == 2) {
        if (shard_rows) {  // the restart decision needs the global ||u-uhat||^2, ||z-v||^2 (admm.m:572-573)
          launch_pack_slots(e->part, nblk, e->red, e->ctrl, e->stream);
          ADMM_TRY(comm_allreduce_device(e->comm, e->red, 16, e->stream));
          fa.slots_reduced = e->red;
        }
        launch_fast_decide(fa, e->stream);
        launch_extrapolate(xa, e->ctrl, e->stream);
      }
      if (!e->a_identity) {  // D'*[c+zx-ux, z-zprev, u]  (getProxOps.m:1514; admm.m:624, 654) in ONE pass
        TimerScope ts(e, ADMM_K_GEMV_T);
        launch_gemv_t(e->planDT, e->D, e->rhs, e->dz, e->u, nrhs_dual, e->partDT, e->ctrl, e->stream);
        launch_sum_partials_t(e->planDT, e->partDT, nrhs_dual, e->g, e->ldg, e->ctrl, e->stream);
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
      if (obj_lasso_gemv) {  // 0.5*||D*x - s||^2  (lasso.m:227)
        TimerScope ts(e, ADMM_K_GEMV_N);
        int nob = 0;
        launch_gemv_n(e->planDN, e->D, e->x, e->partDN, e->ctrl, e->stream);
        launch_residual_sq(e->partDN, e->planDN.nchunk, e->planDN.ldy, e->s, e->m, e->objpart, &nob, e->ctrl,
                           e->stream);
        fa.objpart = e->objpart;
        fa.nobjpart = nob;
        if (sharded) {  // sum over the row shards of ||D_g*x - s_g||^2
          launch_pack_sum(e->objpart, nob, e->red + 16, e->ctrl, e->stream);
          ADMM_TRY(comm_allreduce_device(e->comm, e->red + 16, 1, e->stream));
          fa.objp_reduced = e->red + 16;
        }
      } else if (o.objevals && e->ocb) {  // objevals(i) = obj(x, z) with the caller's handle (admm.m:604)
        if (e->ocb(e->ouser, e->x, nA, e->z, len, e->objpart, static_cast<void*>(e->stream)) != 0)
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
  while (enq < N && !stopped && loop_rc == ADMM_OK) {
    int32_t batch = (N - enq < check_every) ? N - enq : check_every;
    if (gexec) {
      batch = gbatch;  // a full batch; iterations beyond maxiters are device-side no-ops
      if (hipGraphLaunch(gexec, e->stream) != hipSuccess) loop_rc = fail(ADMM_E_DEVICE, "hipGraphLaunch failed");
    } else {
      for (int32_t b = 0; b < batch && loop_rc == ADMM_OK; ++b) loop_rc = enqueue_iteration();
    }
    enq += batch;
    if (loop_rc == ADMM_OK && (!o.domaxiters || enq >= N)) {
      if (hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
          hipStreamSynchronize(e->stream) != hipSuccess)
        loop_rc = fail(ADMM_E_DEVICE, "polling the device control block failed");
      else if (e->ctrl_host->stop) stopped = true;
    }
  }
  if (gexec) (void)hipGraphExecDestroy(gexec);
  if (graph) (void)hipGraphDestroy(graph);
  ADMM_TRY(loop_rc);
  ADMM_HIP_TRY(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
  ADMM_HIP_TRY(hipStreamSynchronize(e->stream));
  {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(ADMM_E_DEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
  }
  const double runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count();
  if (e->profiling) collect_timers(e);
  for (auto& t : e->timers) t.used = 0;

  e->last = admm_run_summary{};
  e->last.steps = e->ctrl_host->steps;
  if (e->cg_st) {
    ADMM_HIP_TRY(hipMemcpy(e->cg_st_host, e->cg_st, sizeof(CgState), hipMemcpyDeviceToHost));
    e->cg_total_last = e->cg_st_host->total;
  }
  e->last.stopped_early = (e->ctrl_host->steps < N) ? 1 : 0;
  e->last.convtest_failed_at = e->ctrl_host->convfail;
  e->last.runtime_s = runtime;
  e->last.objopt = NAN;
  if (o.objevals && e->last.steps > 0) {  // admm.m:752-754: obj(x,z) at the final iterates == last objevals entry
    double v = NAN;
    ADMM_HIP_TRY(hipMemcpy(&v, e->objv + (e->last.steps - 1), sizeof(double), hipMemcpyDeviceToHost));
    e->last.objopt = v;
  }
  e->has_run = true;
  if (summary) *summary = e->last;
  return ADMM_OK;
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
    case ADMM_F_ZOPT: src = e->z; count = e->len; break;
    case ADMM_F_UOPT: src = e->u; count = e->len; break;
    case ADMM_F_XVALS: src = e->xhist; count = e->nA * steps; need_vec_hist = true; break;
    case ADMM_F_ZVALS: src = e->zhist; count = e->len * steps; need_vec_hist = true; break;
    case ADMM_F_UVALS: src = e->uhist; count = e->len * steps; need_vec_hist = true; break;
    case ADMM_F_VVALS: src = e->vhist; count = e->len * steps; need_vec_hist = true; need_fast = true; break;
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
    case ADMM_F_CG_ITERS: {
      if (e->xsolve != ADMM_XSOLVE_CG) return fail(ADMM_E_INVALID, "field exists only for xsolve = cg");
      if (cap < 1) return fail(ADMM_E_CAPACITY, "destination too small");
      dst[0] = static_cast<double>(e->cg_total_last);
      if (written) *written = 1;
      return ADMM_OK;
    }
    case ADMM_F_ZCONSENSUS:
      if (e->problem != ADMM_PROB_LASSO_CONSENSUS) return fail(ADMM_E_INVALID, "field exists only for consensus lasso");
      src = e->czc;
      count = e->nA;
      break;
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
