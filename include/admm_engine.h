/*
 * admm_engine.h -- C ABI of libadmm_hip.so, the MI355X (gfx950) ADMM iteration engine.
 *
 * Drop-in boundary for the hot path of PeterSutor/ADMM-Project: the loop of
 * `results = admm(xminf, zming, options)` (reference admm.m:24, loop 496-743) when
 * both prox handles come from `getproxops(problem, args)` (getProxOps.m:13).  The
 * reference has no native layer at all; these are the entry points a MEX (or ctypes)
 * binding for that path binds -- see INTEGRATION.md for the reference-side stub.
 *
 * Conventions: plain C, no C++/torch types.  All matrices are column-major fp64
 * (MATLAB layout).  Every function returns 0 on success and a negative ADMM_E_* code
 * on failure; admm_last_error() returns the thread-local message.  An engine handle
 * is not re-entrant; one host thread drives one device.
 */
#ifndef ADMM_ENGINE_H
#define ADMM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADMM_ABI_VERSION 5

/* ---- error codes ------------------------------------------------------------ */
enum {
  ADMM_OK = 0,
  ADMM_E_INVALID = -1,     /* bad argument (mirrors the reference's error() calls) */
  ADMM_E_UNSUPPORTED = -2, /* valid in the reference, not engine-native (yet) */
  ADMM_E_DEVICE = -3,      /* HIP runtime / no GPU */
  ADMM_E_NUMERIC = -4,     /* Cholesky breakdown (matrix not positive definite; lad.m:134 errors the same way) */
  ADMM_E_COMM = -5,        /* RCCL */
  ADMM_E_CAPACITY = -6     /* destination buffer too small */
};

/* ---- problem kinds: the cases of getproxops' switch (getProxOps.m:52-917) ---- */
enum {
  ADMM_PROB_LASSO = 1,            /* getProxOps.m:313-456 serial; x: 1192-1206, z: 455 */
  ADMM_PROB_LASSO_CONSENSUS = 2,  /* getProxOps.m:383-442, 1217-1343 (args.parallel=1) */
  ADMM_PROB_LAD = 3,              /* getProxOps.m:753-811, x: 1511-1515 */
  ADMM_PROB_HUBERFIT = 4,         /* getProxOps.m:814-912, z: 1529-1539 */
  ADMM_PROB_LINEARSVM = 5,        /* getProxOps.m:202-310, 1062-1180 */
  ADMM_PROB_TOTALVARIATION = 6,   /* getProxOps.m:145-199, 1044-1048 */
  ADMM_PROB_QP_BOUNDED = 7,       /* getProxOps.m:631-641, 1441-1474 */
  ADMM_PROB_BASISPURSUIT = 8,     /* getProxOps.m:98-142, 1027-1032 */
  ADMM_PROB_LINEARPROGRAM = 10,   /* getProxOps.m:459-542, x: 1357-1366 (KKT solve), z: 1378-1382 */
  ADMM_PROB_QP_STANDARD = 11,     /* getProxOps.m:624-630, x: 1397-1412 (KKT solve), z: 1422-1426 */
  ADMM_PROB_TV2D = 12,            /* engine-side extension: 2-D anisotropic TV of an m x n image (D = [Dv; Dh] forward
                                     differences); x-update of I + rho*D'D spectral (column DCT + row stage) where the
                                     height has a column transform, matrix-free CG otherwise or with ADMM_XSOLVE_CG; no
                                     reference counterpart (totalvariation.m is 1-D) -- BASELINE config 5 as literally written */
  ADMM_PROB_MODEL = 9             /* getProxOps.m:60-95, x: 952-979, z: 990-1013 (model.m); with both prox
                                     callbacks set and no data it is the generic admm(xminf, zming, options)
                                     of admm.m:24 for A = 1, B = -1 */
};

/* linear-SVM loss (getProxOps.m:1094: anything but '01' runs the hinge prox) */
enum { ADMM_LOSS_HINGE = 0, ADMM_LOSS_01 = 1,
       ADMM_LOSS_HINGE_OBJ01 = 2 /* any other lossfunction string (e.g. linearsvmtest.m:157 passes '0-1'): the hinge
                                    prox runs, but linearsvm.m:231-237 installs the 0-1 objective */ };

/* how the cached-factor x-update is applied every iteration */
enum {
  ADMM_XSOLVE_AUTO = 0,    /* INVERSE if it passes the accuracy probe below, else TRSV */
  ADMM_XSOLVE_TRSV = 1,    /* two triangular solves with the Cholesky factor (reference form: getProxOps.m:1200, 1514),
                              as blocked substitution: 2K + 1 bandwidth-bound launches, K = ceil(n / 2048) */
  ADMM_XSOLVE_INVERSE = 2, /* one symmetric n x n GEMV with the explicit inverse, built once.  Its forward error grows
                              like cond^1.5 * eps, so create() solves a system with a known solution with both forms
                              and keeps the explicit inverse only while it is as accurate as the triangular solves
                              (or below 1e-9); otherwise the triangular solves run (admm_engine_info reports which
                              form is in use and the probe errors) */
  ADMM_XSOLVE_CG = 3,      /* matrix-free conjugate gradients on (D'D + rho I), A-streaming */
  ADMM_XSOLVE_CALLBACK = 4,/* A = D problems (LAD shape: A x - z = c): no factor is built, D may have any shape;
                              every run needs an xminf callback (admm_engine_set_callbacks) -- the generic
                              results = admm(xminf, zming, options) with a matrix options.A (admm.m:117-120) */
  ADMM_XSOLVE_PINV = 5     /* linear SVM only: x = pinv(D)*(z-u) (linearsvm.m:185, unwrappedadmm.m:76-78) through the
                              pseudo-inverse of D'D (Jacobi eigen-decomposition, eigenvalues <= n*eps*max dropped).
                              AUTO / TRSV / INVERSE use chol(D'D) while D has full column rank and switch to this
                              form by themselves when the factorisation breaks down or its pivots reach the
                              rounding level (rank-deficient D: e.g. MNIST pixels that are zero in every sample) */
};

/* where the desc's data pointers live */
enum { ADMM_MEM_HOST = 0, ADMM_MEM_DEVICE = 1 };

/* stopcond (admm.m:69, 710-722) */
enum { ADMM_STOP_STANDARD = 0, ADMM_STOP_HNORM = 1, ADMM_STOP_BOTH = 2,
       ADMM_STOP_NONE = 3 /* any other string: the reference's strcmp chain matches nothing */ };

/* fast ADMM flavour (admm.m:63-64, 267-298) */
enum { ADMM_FAST_OFF = 0, ADMM_FAST_WEAK = 2 /* accelerated, alg 2 */, ADMM_FAST_STRONG = 1 /* alg 1 */ };

typedef struct admm_engine admm_engine; /* opaque */

/*
 * Caller-supplied proximal operators: the `xminf` / `zming` function handles of admm.m:24 when they are
 * NOT the library's own (examples/convergencechecking.m:125-136 mixes both kinds).  All pointers are
 * DEVICE pointers into the engine's state; the callback enqueues its work on `hip_stream` (a hipStream_t)
 * or synchronises before returning, and writes its result to `out` (never aliased with an input).
 *   xmin: out[nA] = xminf(x[nA], z[nB], u[nB], rho)   admm.m:502 (fast ADMM passes v, uhat: 506)
 *   zmin: out[nB] = zming(xh, z[nB], u[nB], rho)      admm.m:521-530; xh is x (nA elements) or, when
 *                                                      options.relax != 1, the relaxed Axhat (nB elements)
 *   obj : *out    = obj(x[nA], z[nB])                 admm.m:603-605
 * A non-zero return aborts the run with ADMM_E_INVALID.  Speculatively enqueued iterations after a stop
 * condition still invoke the callbacks; their outputs are discarded on the device.
 */
typedef int (*admm_prox_callback)(void* user, const double* x, const double* z, const double* u, double rho,
                                  double* out, int64_t nout, void* hip_stream);
typedef int (*admm_obj_callback)(void* user, const double* x, int64_t nA, const double* z, int64_t nB, double* out,
                                 void* hip_stream);
/* The two remaining caller-handle fields of the loop (ABI 4), same conventions (device pointers, hip_stream):
 *   altu : out[m] = options.altu(u[m], Ax[m], Bz[m], c[m])   admm.m:553-559 -- replaces the u-update; Ax is the relaxed
 *          Axhat when options.relax != 1 (admm.m:555), c is a zero vector when the constraint has none
 *   norms: out[0..1] = options.specialnorms(x[nA], z[nB], u[m], rho)   admm.m:612-616 -- replaces pnorm / dnorm (perr / derr
 *          are computed as always, admm.m:640-658) */
typedef int (*admm_altu_callback)(void* user, const double* u, const double* Ax, const double* Bz, const double* c,
                                  int64_t m, double* out, void* hip_stream);
typedef int (*admm_norms_callback)(void* user, const double* x, int64_t nA, const double* z, int64_t nB, const double* u,
                                   int64_t m, double rho, double* out2, void* hip_stream);
/* Caller-supplied constraint operators: options.A / options.At as function handles (admm.m:117-158).
 *   out[nout] = A(in[nin])  or  At(in[nin]);  device pointers, enqueue on hip_stream, non-zero return aborts the run */
typedef int (*admm_operator_callback)(void* user, const double* in, int64_t nin, double* out, int64_t nout,
                                      void* hip_stream);
typedef struct admm_comm admm_comm;     /* opaque RCCL communicator wrapper */

/*
 * Problem description = what lasso.m:181-189 / lad.m:129-134 / huberfit.m:161-166 /
 * linearsvm.m:210-214 / totalvariation.m:139-144 / quadraticprogram.m:212-218 pass to
 * getproxops in `args`, as flat pointers.  Unused pointers are NULL.  With
 * ADMM_MEM_HOST the engine copies everything at create (caller may free afterwards);
 * with ADMM_MEM_DEVICE the big matrix D (or P) is borrowed and must outlive the engine.
 */
typedef struct admm_problem_desc {
  int32_t struct_size; /* sizeof(admm_problem_desc), for ABI evolution */
  int32_t problem;     /* ADMM_PROB_* */
  int64_t m, n;        /* rows / columns of D (local rows when row-sharded); P is n x n */
  const double* D;     /* m x n, column-major */
  int64_t ldD;         /* leading dimension of D (>= m) */
  const double* s;     /* length m (LAD/Huber/lasso) or n (total variation signal) */
  const double* ell;   /* length m, labels +-1 (linear SVM) */
  const double* P;     /* n x n: QP matrix, or basis-pursuit projector */
  const double* q;     /* length n: QP linear term, or basis-pursuit offset */
  const double* lb;    /* length n (QP bounded) */
  const double* ub;    /* length n (QP bounded) */
  const double* L;     /* optional precomputed lower Cholesky factor (args.L, lasso.m:183) */
  double lambda;       /* lasso / TV regularisation */
  double C;            /* SVM regularisation */
  double r;            /* QP constant term */
  double rho;          /* rho the cached factor is built for (args.rho) */
  int32_t loss;        /* ADMM_LOSS_* */
  int32_t userelax;    /* args.userelax (lad.m:124-126) */
  int32_t xsolve;      /* ADMM_XSOLVE_* */
  int32_t mem;         /* ADMM_MEM_* */
  int32_t device;      /* HIP device ordinal */
  int32_t nslices;     /* consensus lasso: number of LOCAL row slices (>=1) */
  const int64_t* slices; /* their sizes (sum == m), slicemaker order (errorcheck.m:216-267) */
  admm_comm* comm;     /* NULL = single device; else rows are sharded across the ranks */
  double cg_tol;       /* ADMM_XSOLVE_CG: relative residual tolerance (default 1e-12) */
  int32_t cg_maxit;    /* ADMM_XSOLVE_CG: iteration cap per x-update (default 200) */
  int32_t obj_gram;    /* lasso (and, in the same way, the bounded QP's 1/2*x'Px + q'x, quadraticprogram.m:242), factor
                          built by the engine: how the objective's 1/2*||D*x - s||^2 (lasso.m:227) is
                          evaluated.  The Gram form is 1/2*x'Gx - x'D's + 1/2*s's with G = D'D; since x solves
                          (G + rho*I) x = y (y = rho*(z - u) + D's, getProxOps.m:1195), G x = y - rho*x and the form
                          is a sum over the element update's own operands -- no pass over D (8mn B) or G at all.  Its
                          absolute error is ~1e-16*||s||^2 plus x'(residual of the x-solve).
                          0 = automatic: the engine
                              evaluates BOTH forms during the first host batch of the first objevals run (the
                              literal values are the ones recorded) and switches to the Gram form only if they agreed
                              to 1e-11 relative; 1 = the Gram form always; -1 = the literal D*x form always */
  /* ADMM_PROB_MODEL (getProxOps.m:83-89): P = args.PtP, q = args.Ptr above; and */
  const double* Q;     /* n x n: args.QtQ */
  const double* qz;    /* length n: args.Qts */
  /* optional data of the model objective 1/2||P*x - r||^2 + 1/2||Q*z - s||^2 (model.m:133-134):
   * D = P (m x n), s = r above; and */
  const double* D2;    /* m2 x n: the matrix Q */
  int64_t m2, ldD2;
  const double* s2;    /* length m2: the vector s */
  const double* c;     /* optional constraint vector (length n) of x - z = c; NULL = 0 (model.m:127) */
  /* ADMM_PROB_LINEARPROGRAM / ADMM_PROB_QP_STANDARD: the KKT solve [M D'; D 0] \ [y; s] of
   * getProxOps.m:1363 / 1410 (M = rho*I or P + rho*I) reduced ONCE by the binding to the affine map
   * x = K*y + k0,  K = inv(M) - inv(M) D' inv(S) D inv(M) (symmetric, n x n),  k0 = inv(M) D' inv(S) s,
   * S = D inv(M) D'  (built for desc.rho).  y = rho*(z-u) - q with q = the linear cost (args.b / args.q). */
  const double* K;
  const double* k0;
  /* ADMM_PROB_LINEARSVM: optional pseudo-inverse computed by the caller, n x m column-major, ld = n
   * (args.Dplus = pinv(D), linearsvm.m:185-186).  When given, the x-update is literally x = Dplus*(z-u)
   * (getProxOps.m:1067): one GEMV with it, nothing is factored.  NULL: the engine forms the map from D'D itself. */
  const double* Dplus;
  /* ADMM_PROB_LASSO: optional D'*s (args.Dts, lasso.m:160, 182); the serial args struct carries no s
   * (getProxOps.m:445-451).  Without s the engine-native objective (objevals) is unavailable. */
  const double* Dts;
} admm_problem_desc;

/* POD mirror of the `options` struct read by admm.m:51-76 (defaults: setopt, 780-971). */
typedef struct admm_options {
  int32_t struct_size;
  int32_t maxiters;     /* default 1000 (admm.m:58, 334-339) */
  double rho;           /* default 1.0 */
  double relax;         /* default 1.0 (admm.m:60, 515-532) */
  double abstol;        /* default 1e-5 */
  double reltol;        /* default 1e-3 */
  double Hnormtol;      /* default 1e-6 (accepts the reference's Hreltol spelling host-side) */
  double convtol;       /* default 1e-10 */
  double restart;       /* default 0.999 */
  double dvaltol;       /* default 1e-8 */
  int32_t domaxiters;   /* default 0 */
  int32_t fast;         /* ADMM_FAST_* */
  int32_t objevals;     /* default 0 */
  int32_t convtest;     /* default 0 */
  int32_t stopcond;     /* ADMM_STOP_* */
  int32_t nodualerror;  /* default 0 */
  int32_t record_history; /* 1 = keep xvals/zvals/uvals like admm.m:608-610; 0 = perf opt-out */
  int32_t check_every;  /* iterations enqueued between host polls of the device stop flag (0 = auto) */
  const double* x0;     /* optional warm start, HOST pointers (admm.m:252-254) */
  const double* z0;
  const double* u0;
  int32_t stale_factor_ok; /* 1 = run with options.rho != the rho the cached factor was built with, as xminLASSO does
                              (getProxOps.m:1192-1206 never re-factors; needed by options.adaptive, admm.m:724-741);
                              0 = refuse the mismatch (default: it is almost always a caller's mistake) */
  int32_t reserved1;
} admm_options;

typedef struct admm_run_summary {
  int32_t steps;               /* results.steps (admm.m:746) */
  int32_t stopped_early;       /* 1 if a stop condition fired before maxiters */
  int32_t convtest_failed_at;  /* >0: iteration where the H-norm monotonicity test aborted (admm.m:686-701) */
  int32_t obj_gram_used;   /* 1: the run ended with the lasso objective in its Gram form (admm_problem_desc.obj_gram) */
  double runtime_s;            /* loop only: tic admm.m:315 .. toc admm.m:756 */
  double objopt;               /* results.objopt (admm.m:752-754), NaN if not evaluated */
} admm_run_summary;

/* result fields for admm_engine_fetch (reference names: admm.m:257-259, 582-616, 648-656, 681-682, 746-754) */
enum {
  ADMM_F_XOPT = 1, ADMM_F_ZOPT = 2, ADMM_F_UOPT = 3,
  ADMM_F_XVALS = 4, ADMM_F_ZVALS = 5, ADMM_F_UVALS = 6, /* len x steps, column-major */
  ADMM_F_PNORM = 7, ADMM_F_DNORM = 8, ADMM_F_PERR = 9, ADMM_F_DERR = 10,
  ADMM_F_OBJEVALS = 11, ADMM_F_HNORMSQ = 12,
  ADMM_F_AVALS = 13, ADMM_F_DVALS = 14, ADMM_F_RESTARTED = 15,
  ADMM_F_VVALS = 16, ADMM_F_UHATVALS = 17,
  ADMM_F_ZCONSENSUS = 18, /* consensus lasso: the true consensus z (the z handed to admm is 0, q9) */
  ADMM_F_FACTOR = 19,     /* n x n (or m x m, fat lasso) lower Cholesky factor */
  ADMM_F_CG_ITERS = 20,   /* matrix-free x-update: [0] inner iterations of the last run in total, [1] (capacity >= 2) the number of
                           * its x-updates that ended on cg_maxit with the residual still above cg_tol (an inexact iterate) */
  ADMM_F_CONS_X = 21,     /* consensus lasso: the local slices' x_k, n x K column-major (closure state xi{k}, getProxOps.m:1247) */
  ADMM_F_CONS_U = 22,     /* consensus lasso: the local slices' u_k (closure state ui{k}, getProxOps.m:1296) */
  ADMM_F_WVALS = 23       /* (nA + nB + nU) x steps: w = [x; z; rho*u] per iteration, results.wvals of the H-norm runs
                           * (admm.m:678-681); needs the vector histories */
};

/* ---- library ---------------------------------------------------------------- */
int admm_abi_version(void);
const char* admm_last_error(void);
int admm_device_count(int* count);
/* name[cap] gets the gcnArchName; hbm_bytes/cus may be NULL */
int admm_device_info(int device, char* name, size_t cap, int64_t* hbm_bytes, int32_t* cus);

/* ---- engine lifecycle --------------------------------------------------------
 * create   = the solver's one-time setup + getproxops (lasso.m:160-192, lad.m:129-137, ...):
 *            uploads data, builds Gram matrix / Cholesky factor / pseudo-inverse on device.
 * run      = admm(minx, minz, options): the iteration loop, entirely on device.
 * fetch    = copy one results.* field into caller memory (MEX: mxGetPr of the output).
 */
void admm_options_default(admm_options* opts);
void admm_problem_desc_default(admm_problem_desc* desc);
int admm_engine_create(const admm_problem_desc* desc, admm_engine** out);
/* replace the x- and/or z-update (and the objective hook) of an A = 1 problem (tall lasso, QP, LP, basis
 * pursuit, model) or an A = D problem (LAD, Huber, linear SVM: the zming handle of
 * unwrappedadmm(zming, D, options), unwrappedadmm.m:1) by caller-supplied callbacks; NULL keeps the
 * engine-native operator.  x has nA = n elements; z, u and zmin's first argument have nB elements (n or m).
 * ADMM_PROB_MODEL created without Gram data REQUIRES the corresponding callback. */
int admm_engine_set_callbacks(admm_engine* eng, admm_prox_callback xmin, void* xuser, admm_prox_callback zmin,
                              void* zuser, admm_obj_callback obj, void* objuser);
/* options.A / options.At as function handles: an engine created as ADMM_PROB_LAD with ADMM_XSOLVE_CALLBACK and
 * desc.D = NULL (desc.m = rows of A = length of z, u, c; desc.n = length of x; desc.s = c) has no matrix at all --
 * A*x, At*(.) of the dual residual / tolerance (admm.m:535, 624, 654) are these callbacks, both required; B = -1. */
int admm_engine_set_operators(admm_engine* eng, admm_operator_callback A, void* Auser, admm_operator_callback At,
                              void* Atuser);
/* options.altu / options.specialnorms as the CALLER's handles (admm.m:553-559, 612-616; the consensus-lasso hooks of
 * getproxops' `extra` are engine-native and need none of this).  Plain ADMM only (the reference never combines them with
 * fast / accelerated ADMM: q5), any engine of the general loop (not consensus lasso / total variation, not row-sharded).
 * With a hook set the iteration leaves the fused one-launch tails: element update, hook, a small fix-up kernel, finalize.
 * NULL, NULL restores the engine's own u-update and norms. */
int admm_engine_set_hooks(admm_engine* eng, admm_altu_callback altu, void* altu_user, admm_norms_callback norms,
                          void* norms_user);
/* options.B other than the shorthand -1 (admm.m:198-245): a scalar (B = NULL, Bop = NULL: B = scalar*I), an m x nB
 * matrix (column-major, leading dimension ldB, host or device pointer per memkind = ADMM_MEM_*), or a function handle
 * Bop(z[nB]) -> out[m].  Only for engines whose two prox operators are BOTH the caller's (ADMM_PROB_MODEL created
 * without Gram data, or ADMM_PROB_LAD with ADMM_XSOLVE_CALLBACK): the library's own operators are written for
 * B = -1.  Afterwards z, v and zming's result have nB elements (callbacks, options.z0, ADMM_F_ZOPT / ZVALS / VVALS);
 * u, c and the relaxed Axhat keep m.  B enters the loop linearly (admm.m:515, 536-552, 573, 621-658, 305), so the
 * device loop carries w = -B*z and runs the same fused kernels. */
int admm_engine_set_constraint_b(admm_engine* eng, const double* B, int64_t ldB, int64_t nB, int32_t memkind,
                                 double scalar, admm_operator_callback Bop, void* Buser);
int admm_engine_run(admm_engine* eng, const admm_options* opts, admm_run_summary* summary);
int admm_engine_fetch(admm_engine* eng, int field, double* dst, size_t cap, size_t* written);
/* what create() decided about the x-update factor (first slice for consensus lasso) */
typedef struct admm_engine_info_t {
  int32_t struct_size;       /* sizeof(admm_engine_info_t), set by the caller */
  int32_t xsolve_requested;  /* desc.xsolve as resolved: AUTO, TRSV or INVERSE */
  int32_t xsolve_used;       /* ADMM_XSOLVE_TRSV or ADMM_XSOLVE_INVERSE (CG / CALLBACK / 0 where no factor exists) */
  int32_t pinv_used;         /* 1: the explicit matrix is the pseudo-inverse of a rank-deficient D'D */
  int32_t probed;            /* 1: both forms were built and compared */
  int32_t trsv_blocks;       /* coarse blocks K of the blocked substitution (0 if not in use) */
  int32_t jacobi_sweeps;     /* sweeps of the eigen-solver (pinv) */
  int32_t unwrapped_fused;   /* 1: the two-launch unwrapped iteration with an explicit pinv(D) is available (linear SVM) */
  int64_t factor_n;          /* order of the factor (n; m for fat lasso) */
  int64_t rank;              /* numerical rank (== factor_n unless pinv_used) */
  double cond_estimate;      /* (max L_ii / min L_ii)^2, a lower bound of cond(L L'); pinv: lambda_max / lambda_min kept */
  double probe_err_inverse;  /* max-norm relative forward error of each form on (L L') x = L (L' x0); NaN if not probed */
  double probe_err_trsv;
  double probe_diff;         /* max-norm relative difference of the two forms on an unstructured right-hand side */
  /* bytes of the x-solve's matrix operand one iteration reads, split by cache policy (ABI 4): read with default loads
   * (meant to stay in the 256 MB Infinity Cache from one iteration to the next) / streamed non-temporally from HBM.
   * Consensus lasso: summed over the local slices.  0 / 0 where no n x n operand exists. */
  int64_t xsolve_cacheable_bytes;
  int64_t xsolve_stream_bytes;
  /* lasso / bounded-QP objective through the x-update's right-hand side (desc.obj_gram): the largest value of
   * eps * |cancelling terms| / |objective| any recorded objective of this engine has had (0: form never used); past
   * 1e-10 the engine has gone back to the literal form (obj_form_literal = 1) */
  double obj_bound_max;
  int32_t obj_form_literal;
  int32_t reserved0;
  /* ABI 5: the triangular solves' one-block form (the whole factor pre-inverted: two passes, two launches per pair;
   * trsv_blocks = 1 when it is in use): its forward error on the probe system next to probe_err_trsv, which is the
   * blocked substitution's (the yardstick); NaN where it was not built (n < 1536, or the explicit inverse runs) */
  double probe_err_trsv_one;
} admm_engine_info_t;
int admm_engine_info(admm_engine* eng, admm_engine_info_t* info);
/* seconds spent in create (upload + factorisation); solverruntime = setup + runtime */
int admm_engine_setup_seconds(admm_engine* eng, double* seconds);
/* per-kernel timing of the last run, measured with HIP events on the engine's stream:
 * which = ADMM_K_*; returns total milliseconds and launch count */
enum { ADMM_K_XSOLVE = 0, ADMM_K_GEMV_N = 1, ADMM_K_GEMV_T = 2, ADMM_K_PROX = 3, ADMM_K_FINALIZE = 4, ADMM_K_COUNT = 5 };
int admm_engine_kernel_time(admm_engine* eng, int which, double* total_ms, int64_t* launches);
/* per-kernel event timing for subsequent runs: mask = OR of (1 << ADMM_K_*), 0 = off (default),
 * negative = every class.  Each timed class costs two hipEventRecord per launch group. */
int admm_engine_set_profiling(admm_engine* eng, int mask);
/* time only every stride-th launch group of each selected class (default 1 = all): a sampled measurement costs the
 * loop 2/stride event records per iteration instead of 2 */
int admm_engine_set_profiling_stride(admm_engine* eng, int stride);
void admm_engine_destroy(admm_engine* eng);

/* ---- binding layer (ABI 5): the getproxops / admm argument structs seen through the C ABI -------------------------
 * A host language (the MATLAB MEX gateway, ctypes, ...) flattens its structs into admm_field entries; everything that
 * INTERPRETS them lives behind the ABI (csrc/binding.hip; host code, no GPU needed): which field means what for which
 * problem (getProxOps.m:52-917: the args struct of getproxops), the defaults of admm's options (admm.m:780-971), what
 * admm refuses before its loop, and the layout of the results struct (admm.m:257-767).  A gateway converts containers
 * and stages function handles; it decides nothing. */
enum { ADMM_FIELD_NUMERIC = 0,  /* full real double array: data, rows, cols (column-major) */
       ADMM_FIELD_TEXT = 1,     /* character vector: text */
       ADMM_FIELD_HANDLE = 2,   /* a function handle (its presence is what counts; the gateway keeps the callable) */
       ADMM_FIELD_SPARSE = 3,   /* real double CSC matrix: data = nonzeros, ir = their rows, jc = column starts (cols + 1) */
       ADMM_FIELD_OTHER = 4 };
typedef struct admm_field {
  const char* name;
  int32_t kind;
  int32_t reserved;
  const double* data;
  int64_t rows, cols;
  const char* text;
  const uint64_t* ir;
  const uint64_t* jc;
} admm_field;
typedef struct admm_binding admm_binding;
/* problem: the getproxops problem string ('lasso', 'lad', 'huberfit', 'linearsvm', 'totalvariation', 'quadraticprogram',
 * 'linearprogram', 'basispursuit', 'model'; extensions 'totalvariation2d', 'generic' = both prox operators the caller's);
 * args: the struct getproxops receives; handles: the caller's function handles by name (xminf, zming, obj, A, At, B, altu,
 * specialnorms) plus the scalars a solver keeps next to them (s, r, objnative).  The description borrows the numeric
 * arrays of `args`: they must stay valid until admm_engine_create has returned. */
int admm_binding_create(const char* problem, const admm_field* args, int32_t nargs, const admm_field* handles,
                        int32_t nhandles, admm_binding** out);
void admm_binding_destroy(admm_binding* b);
const admm_problem_desc* admm_binding_desc(const admm_binding* b);
typedef struct admm_binding_info {
  int32_t struct_size;  /* sizeof(admm_binding_info), set by the caller */
  int32_t problem;      /* ADMM_PROB_* */
  int64_t nA, nB, nU;   /* lengths of x, of z, and of u / c / A*x (admm.m:252-254) */
  int32_t a_handle;     /* generic: A and At are function handles (admm_engine_set_operators before a run) */
  int32_t b_kind;       /* generic: 0 = B is the shorthand -1, 1 = another scalar, 2 = a matrix, 3 = a function handle */
  double b_scalar;
  const double* b_matrix;
  int64_t b_ld;
} admm_binding_info;
int admm_binding_get_info(const admm_binding* b, admm_binding_info* info);
/* after admm_engine_create: a scalar or matrix B goes to the engine (admm_engine_set_constraint_b) */
int admm_binding_apply(const admm_binding* b, admm_engine* eng);
/* options struct -> admm_options (defaults of admm.m:780-971; x0 / z0 / u0 point into the fields) and the checks admm
 * makes before its loop -- callable BEFORE an engine exists, so that a refused call never holds device memory */
int admm_binding_options(const admm_binding* b, const admm_field* options, int32_t nopt, const admm_field* handles,
                         int32_t nhandles, admm_options* out);
/* the fields of results after a run, in the reference's order (admm.m:257-767) */
enum { ADMM_RES_FETCH = 0,   /* admm_engine_fetch(source = ADMM_F_*), rows x cols */
       ADMM_RES_SCALAR = 1,  /* `scalar` */
       ADMM_RES_START = 2 }; /* the start vector the run used: options.x0 / z0 / u0 (source 0 / 1 / 2) or zeros */
typedef struct admm_result_field {
  const char* name;  /* static storage */
  int32_t kind, source;
  int64_t rows, cols;
  double scalar;
} admm_result_field;
int admm_binding_results(admm_binding* b, const admm_options* opts, const admm_run_summary* summary,
                         admm_result_field* out, int32_t cap, int32_t* count);

/* Host <-> device copies on the engine's stream (hip_stream as handed to a callback, NULL = default stream), complete on
 * return.  For bindings whose callbacks live on the host (a MATLAB function handle staged by the MEX gateway): the
 * gateway needs no HIP headers. */
int admm_memcpy_d2h(void* host_dst, const void* device_src, size_t bytes, void* hip_stream);
int admm_memcpy_h2d(void* device_dst, const void* host_src, size_t bytes, void* hip_stream);

/* ---- stand-alone operators (kernel-level entry points; HOST pointers) ---------
 * The building blocks of the loop, callable on their own: used by the parity tests
 * and by bindings that run a user-supplied prox on the host but want the heavy
 * linear algebra on the device.  Each cites the reference op it replaces.
 */
/* y = D*x  (admm.m:120 `A*v`; getProxOps.m:810, 1088) */
int admm_op_gemv_n(const double* D, int64_t m, int64_t n, int64_t ldD, const double* x, double* y);
/* G(:,k) = D'*V(:,k), k < nrhs <= 4  (admm.m:119/167 `At*v`; getProxOps.m:1514) */
int admm_op_gemv_t(const double* D, int64_t m, int64_t n, int64_t ldD, const double* V, int64_t ldV,
                   int32_t nrhs, double* G, int64_t ldG);
/* W = D'*D (+ shift on the diagonal), n x n  (lasso.m:168, lad.m:134, unwrappedadmm.m:115) */
int admm_op_gram(const double* D, int64_t m, int64_t n, int64_t ldD, double shift, double* W);
/* in place lower Cholesky of the n x n SPD matrix A (chol(.,'lower'), lasso.m:168) */
int admm_op_cholesky(double* A, int64_t n, int64_t ldA);
/* x = L' \ (L \ y)  (getProxOps.m:1200, 1514) */
int admm_op_trsv_pair(const double* L, int64_t n, int64_t ldL, const double* y, double* x);
/* soft threshold sign(v).*max(abs(v)-t,0)  (getProxOps.m:933-938) */
int admm_op_soft_threshold(const double* v, int64_t n, double t, double* out);

/* ---- multi-GPU (one process per GPU; rows of D sharded; RCCL over xGMI) --------
 * unique id is created on rank 0 and handed to the other ranks by the host
 * (torch.distributed / MPI / a file).  errorcheck.m:216-267 defines the row partition.
 */
#define ADMM_COMM_ID_BYTES 128
/* transports: RCCL over xGMI (collectives enqueued on the engine's stream), or a host-staged
 * all-reduce through POSIX shared memory (any number of ranks may then share one GPU; used for
 * single-GPU testing of the sharded engines and as a fallback) */
enum { ADMM_COMM_RCCL = 0, ADMM_COMM_SHM = 1,
       /* one-shot peer-to-peer all-reduce for the loop's small payloads (80-240 KB: latency, not bandwidth): every rank
        * stores its contribution straight into a slot of every peer's device buffer (xGMI is a full point-to-point
        * mesh: one hop, where a ring makes 2(N-1)), raises a flag there, waits for the N flags in its own buffer and
        * sums the N slots in rank order -- ONE kernel per rank, on the engine's stream, no host synchronisation, bitwise
        * the same result on every rank.  Ranks of one node (processes or threads): buffers are shared through HIP IPC
        * handles exchanged over the shared-memory rendezvous.  Never run on real links in this repository's records. */
       ADMM_COMM_P2P = 2 };
int admm_comm_unique_id(char id[ADMM_COMM_ID_BYTES]);
int admm_comm_init(const char id[ADMM_COMM_ID_BYTES], int rank, int nranks, int device, int transport,
                   admm_comm** out);
int admm_comm_info(admm_comm* comm, int* rank, int* nranks, int* transport);
int admm_comm_allreduce_sum(admm_comm* comm, double* host_buf, size_t count); /* host convenience/test */
/* average wall time (microseconds) of one in-place sum all-reduce of `count` doubles on this communicator: `reps` of
 * them enqueued back to back on a stream of their own after 3 warm-up rounds (collective: every rank calls it with the
 * same arguments).  What the engine's per-iteration exchange costs on the links this process group really has. */
int admm_comm_measure_latency(admm_comm* comm, size_t count, int reps, double* microseconds);
void admm_comm_destroy(admm_comm* comm);

/* One host process driving several GPUs (a MATLAB session with a MEX gateway; the reference opens its pool from one
 * session: admm.m:347-356, unwrappedadmm.m:47).  The engines are the per-rank engines above; these calls run every
 * rank's create / run on a host thread of its own (each blocks inside its collectives) and join them.
 *   comm_init_all : nranks communicators of one group inside this process, comms[r] on devices[r] (the same device
 *                   may appear more than once with ADMM_COMM_SHM)
 *   create_all    : descs[r].comm = comms[r], descs[r].device = devices[r], each with its rows (slicemaker order)
 *   run_all       : opts[0] for every rank (opts_per_rank = 0) or opts[r]; summaries may be NULL
 * On failure the message of the first failing rank is the caller's admm_last_error(). */
int admm_comm_init_all(int nranks, const int* devices, int transport, admm_comm** comms);
int admm_engine_create_all(int nranks, const admm_problem_desc* descs, admm_engine** engines);
int admm_engine_run_all(int nranks, admm_engine* const* engines, const admm_options* opts, int opts_per_rank,
                        admm_run_summary* summaries);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_ENGINE_H */
