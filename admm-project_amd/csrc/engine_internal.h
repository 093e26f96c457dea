// engine_internal.h -- the engine object behind the opaque admm_engine handle and the host helpers shared by
// engine.hip (create / fetch / destroy: the reference solvers' one-time setup) and engine_run.hip (run: the loop).
#pragma once
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
int comm_stream_create(const admm_comm* comm, hipStream_t* out);  // a stream whose hardware queue no same-device peer rank shares (comm.hip)
int comm_check_error(admm_comm* comm, hipStream_t stream);  // P2P transport: a peer never arrived (synchronises)
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
  double* DplusT = nullptr;  // linear SVM with the caller's pseudo-inverse (args.Dplus): its transpose, m x n, ld = ldD
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

  // cached factor of the x-update and how it is applied (blocked triangular solves or explicit inverse; for a
  // rank-deficient linear-SVM D the pseudo-inverse of D'D): see build_slice_factor / factorize_pinv in engine.hip
  SliceFactor xfac{};
  int xsolve_requested = ADMM_XSOLVE_AUTO;  // what desc.xsolve asked for (AUTO / TRSV / INVERSE)
  double* F = nullptr;  // = xfac.F: lower Cholesky factor, nF x nF (null when the pseudo-inverse is in use)
  int64_t nF = 0, ldF = 0;

  // GEMV plans + partial buffers
  GemvNPlan planDN{};   // D*x
  GemvTPlan planDT{};   // D'*v
  GemvTPlan planSq{};   // square symmetric nA x nA GEMV (Minv or P) run as column dots: M*v == M'*v
  double *syN = nullptr, *syT = nullptr;  // partial-sum buffers of the lower-triangle kernel (shared by all factors)
  size_t sy_elems = 0;
  bool sy_split = false;  // multi-GPU: split the tiles of the x-solve over the ranks (decided by measurement at create)
  double *partDN = nullptr, *partDT = nullptr, *partSq = nullptr;

  // iterates
  double *x = nullptr, *z = nullptr, *u = nullptr, *rhs = nullptr, *dz = nullptr, *g = nullptr;
  int64_t ldg = 0;
  double* opG = nullptr;         // one-pass A = D iteration: [workgroups][ldg] partial rows of D'*(c + z - u)
  double *v = nullptr, *uhat = nullptr, *zprev = nullptr, *uprev = nullptr;
  double *tmpA = nullptr, *tmpB = nullptr;  // fat lasso scratch (m and n long)
  // total variation: forward-sweep intermediate, ping-pong partners of z/u, LDL' pivot prefix
  int64_t tv2_H = 0, tv2_W = 0;  // 2-D TV image shape
  Ctrl* ctrl_idle = nullptr;     // an all-zero control block for clean-up launches after the loop has stopped
  // lasso objective through the Gram matrix (desc.obj_gram): tile-padded copy of D'D, -D's, G*x scratch, 1/2*s's
  double half_ssq = 0.0;
  // obj_gram = 0 (automatic): calibrate against the literal form during the first batch, then decide (engine_run.hip)
  bool obj_alt = false;  // the solve-identity form of the lasso objective is available (1/2*s's is known)
  bool obj_auto = false, obj_gram_ok = false, obj_gram_bad = false;
  double obj_bound_seen = 0.0;  // largest cancellation bound of the right-hand-side objective form over all runs
  double* gobjpart = nullptr;  // [kMaxPartBlocks + 1] Gram-form partials during calibration; last entry: max discrepancy
  bool tv2_dct = false;          // spectral x-update instead of CG: the height a power of two (dct.h)
  bool tv2_rows_dct = false;     // ... and the width too: the row DCT exists as the fall-back of the Toeplitz row stage
  DctTables dctH{}, dctW{};
  double* tv_y2 = nullptr;  // ping-pong partner of tv_y (fused iteration kernel)
  double *tv_y = nullptr, *tv_zA = nullptr, *tv_uA = nullptr, *tv_zB = nullptr, *tv_uB = nullptr;
  double* tv_bprefix = nullptr;
  size_t tv_bprefix_cap = 0;
  double* tv_part = nullptr;    // tile partials of the fused 1-D kernels, kept across runs (grown when rho asks for more)
  size_t tv_part_cap = 0;
  double *tv_y3 = nullptr, *tv_v3 = nullptr;  // third rotation buffers of the deferred tail, allocated on first use
  // matrix-free x-update (xsolve = cg)
  double cg_tol = 1e-12;
  int32_t cg_maxit = 200;
  bool cg_maxit_auto = false;  // 2-D TV with the default cap: the cap follows rho (cond(I + rho*D'D) < 1 + 8*rho)
  bool cg_shift_is_rho = false;
  double *cg_r = nullptr, *cg_p = nullptr, *cg_q = nullptr, *cg_tmp = nullptr, *cg_part = nullptr;
  CgState* cg_st = nullptr;
  Ctrl* cg_skip = nullptr;  // Ctrl-shaped block whose .stop mirrors (CG converged || ctrl->stop): skips the operator kernels
  int cg_chunk = 8;         // inner iterations enqueued between polls: follows the last solve's count
  CgState* cg_st_host = nullptr;  // pinned
  int64_t cg_total_last = 0, cg_capped_last = 0;
  // consensus lasso (getProxOps.m:383-442, 1217-1343)
  std::vector<ConsSlice> cslices;
  int32_t cons_total = 0;  // slicenum over all ranks
  double *cX = nullptr, *cU = nullptr, *csums = nullptr, *czc = nullptr, *cxave = nullptr, *cxaveprev = nullptr,
         *cubar = nullptr, *cobjpart = nullptr, *cDts = nullptr, *cY = nullptr;
  double *csyN = nullptr, *csyT = nullptr;  // per-slice partial rows of the lower-triangle x-solves ([K][cpstride])
  int64_t cpstride = 0;
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
  // deferred finalize (engine_run.hip): the previous iteration's finalize arguments ride along with the packed x-solve
  const FinArgs* dfin = nullptr;
  bool dfin_pending = false;
  const double** cMptr = nullptr;  // consensus: device array of the K packed slice inverses (one batched x-solve launch)
  // two-launch unwrapped iteration (unwrapped.hip): pinv(D) as an n x m matrix and the double-buffered partial rows
  double* Dp = nullptr;
  int64_t ldDp = 0;
  double *uwG = nullptr, *uwAx = nullptr, *uwX = nullptr;  // partial rows of x, partial D*x, x double-buffered
  int32_t uwR = 0, uwnblk = 0, uwnchunk = 0;
  int64_t uwldg = 0, uwldax = 0;
  admm_operator_callback acb = nullptr, atcb = nullptr;  // options.A / options.At as function handles (no D)
  void *auser = nullptr, *atuser = nullptr;
  double* axbuf = nullptr;  // A(x) when A is a callback
  // options.B other than the shorthand -1 (admm.m:198-245; admm_engine_set_constraint_b): the caller's z lives in a
  // space of its own (nBz elements: zt, its previous value, the zming result, the fast-ADMM v) while the loop's "z"
  // buffers hold w = -B*z (len elements) -- B enters admm.m:515-700 linearly, so every fused kernel runs unchanged on w
  bool bgen = false;
  int64_t nBz = 0;
  double bscalar = -1.0;
  double* Bmat = nullptr;
  int64_t ldB = 0;
  GemvNPlan planBN{};
  double* partBN = nullptr;
  admm_operator_callback bcb = nullptr;
  void* buser = nullptr;
  double *zt = nullptr, *ztprev = nullptr, *ztnew = nullptr, *vt = nullptr, *btmp = nullptr;
  double *zthist = nullptr, *vthist = nullptr;
  admm_prox_callback xcb = nullptr, zcb = nullptr;
  admm_obj_callback ocb = nullptr;
  // options.altu / options.specialnorms as the caller's handles (admm_engine_set_hooks)
  admm_altu_callback altucb = nullptr;
  admm_norms_callback normscb = nullptr;
  void *altuuser = nullptr, *normsuser = nullptr;
  double *hk_uold = nullptr, *hk_bz = nullptr, *hk_unew = nullptr, *hk_zero = nullptr, *hk_norms = nullptr;
  void *xuser = nullptr, *zuser = nullptr, *ouser = nullptr;
  double* part = nullptr;     // [S_COUNT][kMaxPartBlocks]
  double* objpart = nullptr;  // [kMaxPartBlocks]
  Ctrl* ctrl = nullptr;
  Ctrl* ctrl_host = nullptr;  // pinned

  // histories of the last run
  int32_t hist_cap = 0;
  bool hist_vectors = false, hist_fast = false, hist_bgen = false;
  double *xhist = nullptr, *zhist = nullptr, *uhist = nullptr, *vhist = nullptr, *uhathist = nullptr;
  double *pnorm = nullptr, *dnorm = nullptr, *perr = nullptr, *derr = nullptr, *objv = nullptr, *hnorm = nullptr,
         *avals = nullptr, *dvals = nullptr, *restarted = nullptr;
  std::vector<void*> hist_ptrs;
  admm_options last_opts{};
  admm_run_summary last{};
  bool has_run = false;
  double setup_seconds = 0.0;

  uint32_t profiling = 0;  // bit k set: time kernel class k with HIP events
  int32_t prof_stride = 1;  // ... on every prof_stride-th launch group of the class (sampling keeps the cost of the
  uint32_t prof_tick[ADMM_K_COUNT] = {0, 0, 0, 0, 0};  // two event records out of the other iterations)
  KTimer timers[ADMM_K_COUNT];
};

namespace admm {

int upload(DevMem& mem, double** dst, const double* src, size_t elems, int memkind, hipStream_t stream);
// copy a column-major m x n matrix into a zero-padded device buffer with leading dimension ld
int upload_matrix(DevMem& mem, double** dst, int64_t* ld_out, const double* src, int64_t rows, int64_t cols,
                  int64_t ld_src, int memkind, hipStream_t stream);
void free_hist(admm_engine* e);
int hist_alloc(admm_engine* e, double** out, size_t elems);
void collect_timers(admm_engine* e);
// W (nF x nF, ld) holds an SPD matrix in its lower triangle -> F (in place), dinv, optionally Minv
int factorize(admm_engine* e, double* W, int64_t nF, int64_t ld, const double* Lgiven, int memkind);
int symv_apply(admm_engine* e, const double* y, double* out);
int solve_factor(admm_engine* e, const double* y, double* out);
int build_slice_factor(admm_engine* e, SliceFactor& f, double* W, int64_t n, int64_t ld, int want, const double* Lgiven,
                       int memkind);
int factorize_pinv(admm_engine* e, double* W, int64_t n, int64_t ld);
void apply_slice_factor(admm_engine* e, const SliceFactor& f, const double* y, double* out);

struct TimerScope {
  admm_engine* e;
  int which;
  bool on;
  size_t slot = 0;
  TimerScope(admm_engine* eng, int w) : e(eng), which(w), on((eng->profiling >> w) & 1u) {
    if (on && eng->prof_stride > 1) on = (eng->prof_tick[w]++ % static_cast<uint32_t>(eng->prof_stride)) == 0;
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

// what admm_engine_run's prologue hands to the per-problem iteration sequences
struct RunState {
  admm_options o;
  int alg;          // 0, 1 (strong), 2 (weak)
  int32_t N;        // maxiters
  int64_t len;      // length of z, u
  ProxArgs pa;
  FinArgs fa;
  ExtrapArgs xa;
};

// engine_run.hip
int cg_solve(admm_engine* e, const double* y);
// engine_run_tv.hip: total variation (totalvariation.m) and the 2-D extension
int cg_solve_tv2d(admm_engine* e, const double* y);
int run_total_variation(admm_engine* e, RunState& rs, admm_run_summary* summary);
int run_total_variation_2d(admm_engine* e, RunState& rs, admm_run_summary* summary);
// engine_run_consensus.hip: consensus lasso (getProxOps.m:1217-1343)
int run_consensus_lasso(admm_engine* e, RunState& rs, admm_run_summary* summary);

}  // namespace admm
