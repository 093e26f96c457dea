// admm_mex.cpp -- MATLAB MEX gateway to libadmm_hip.so (see INTEGRATION.md and matlab/admm.m, matlab/getproxops.m).
//
// MATLAB / mex.h do not exist in this repository's image; compile where MATLAB does:
//     mex -I../../include admm_mex.cpp -L.. -ladmm_hip
// The test-suite builds this very file against an executable stand-in of the MEX API
// (tests/c_abi/mex_stub/) and drives mexFunction on the GPU (tests/test_mex_gateway.py).
// All numerics live behind the C ABI (include/admm_engine.h); this file converts mxArrays <-> flat pointers and
// stages host-language function handles (mexCallMATLAB) through host memory.
//
//   ok      = admm_mex('available')
//   results = admm_mex('solve', problem, args, options, handles)
//        problem : the getproxops problem string ('lasso', 'lad', ..., getProxOps.m:52-917), or 'generic'
//        args    : the struct getproxops receives (getProxOps.m:83-89, 137-138, 190-194, 276-305, 384-387,
//                  445-451, 532-538, 624-651, 794-796, 891-893); 'generic': fields n (or nA, nB), A, c
//        options : the struct admm receives (admm.m:51-76)
//        handles : struct with optional fields xminf, zming, obj = MATLAB function handles that replace the
//                  engine-native operator (admm.m:24, 502, 521-530, 603-605); objnative = 1 selects the
//                  engine's own objective of the problem (the solver's options.obj restated on the device)
//   h = admm_mex('create', problem, args); results = admm_mex('run', h, options, handles); admm_mex('destroy', h)
//        the same in three steps, for callers that re-run one engine (rho sweeps with a cached factor)
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "admm_engine.h"
#include "mex.h"

namespace {

std::vector<admm_engine*> g_live;

void at_exit() {
  for (admm_engine* e : g_live) admm_engine_destroy(e);
  g_live.clear();
}

std::string to_string(const mxArray* a) {
  char* c = mxArrayToString(a);
  if (!c) mexErrMsgIdAndTxt("admm:arg", "expected a string");
  std::string s(c);
  mxFree(c);
  for (auto& ch : s) ch = static_cast<char>(tolower(ch));
  return s;
}

const mxArray* field(const mxArray* s, const char* name) {
  return (s && mxIsStruct(s)) ? mxGetField(s, 0, name) : nullptr;
}

bool is_dense_double(const mxArray* f) { return f && mxIsDouble(f) && !mxIsSparse(f) && !mxIsComplex(f); }

const double* opt_vec(const mxArray* s, const char* name, size_t* count = nullptr) {
  const mxArray* f = field(s, name);
  if (!is_dense_double(f) || mxGetNumberOfElements(f) == 0) return nullptr;
  if (count) *count = mxGetNumberOfElements(f);
  return mxGetPr(f);
}

double opt_scalar(const mxArray* s, const char* name, double dflt) {
  const mxArray* f = field(s, name);
  return (f && (mxIsNumeric(f) || mxIsLogical(f)) && mxGetNumberOfElements(f) == 1) ? mxGetScalar(f) : dflt;
}

bool str_is(const mxArray* s, const char* name, const char* value) {
  const mxArray* f = field(s, name);
  return f && mxIsChar(f) && to_string(f) == value;
}

bool is_handle(const mxArray* f) { return f && mxIsClass(f, "function_handle"); }

void check(int rc) {
  if (rc != ADMM_OK) mexErrMsgIdAndTxt("admm:engine", "%s", admm_last_error());
}

// ---- problem description ---------------------------------------------------------------------------------
struct Desc {
  admm_problem_desc d;
  std::vector<double> dense_L;    // args.L stored sparse (lasso.m:175) expanded column by column
  std::vector<int64_t> slices;    // args.slices (getProxOps.m:387) as integers
  int64_t nA = 0, nB = 0;         // lengths of x and of z
  int64_t mC = 0;                 // length of u and c; 0 = the same as nB (every problem but a generic one with a general B)
  bool a_matrix = false;          // the constraint matrix A is the data matrix (LAD shape)
  bool a_handle = false;          // generic: options.A / options.At are function handles (handles.A / handles.At)
  int b_kind = 0;                 // generic: 0 = the shorthand -1, 1 = another scalar, 2 = matrix, 3 = function handle
  double b_scalar = -1.0;
  const double* b_matrix = nullptr;
  int64_t b_ld = 0;
  std::vector<double> zeros;      // c = 0 for the engines that take c as a data vector
};

// a lower-triangular factor handed in as args.L / args.R: dense as is, sparse (CSC) expanded
const double* factor_arg(const mxArray* args, const char* name, Desc& ds, int64_t order) {
  const mxArray* f = field(args, name);
  if (!f || !mxIsDouble(f) || mxIsComplex(f)) return nullptr;
  if (static_cast<int64_t>(mxGetM(f)) != order || static_cast<int64_t>(mxGetN(f)) != order) return nullptr;
  if (!mxIsSparse(f)) return mxGetPr(f);
  const mwIndex* ir = mxGetIr(f);
  const mwIndex* jc = mxGetJc(f);
  const double* pr = mxGetPr(f);
  ds.dense_L.assign(static_cast<size_t>(order) * order, 0.0);
  for (int64_t j = 0; j < order; ++j)
    for (mwIndex k = jc[j]; k < jc[j + 1]; ++k) ds.dense_L[static_cast<size_t>(ir[k]) + static_cast<size_t>(j) * order] = pr[k];
  return ds.dense_L.data();
}

void describe(const std::string& p, const mxArray* args, const mxArray* handles, Desc& ds) {
  admm_problem_desc& d = ds.d;
  admm_problem_desc_default(&d);
  const mxArray* D = field(args, "D");
  const bool haveD = is_dense_double(D);
  if (haveD) {
    d.D = mxGetPr(D);
    d.m = static_cast<int64_t>(mxGetM(D));
    d.n = static_cast<int64_t>(mxGetN(D));
    d.ldD = d.m;
  }
  d.rho = opt_scalar(args, "rho", 1.0);
  d.device = static_cast<int32_t>(opt_scalar(args, "device", 0.0));
  if (str_is(args, "xsolve", "trsv")) d.xsolve = ADMM_XSOLVE_TRSV;
  else if (str_is(args, "xsolve", "inverse")) d.xsolve = ADMM_XSOLVE_INVERSE;
  else if (str_is(args, "xsolve", "cg")) d.xsolve = ADMM_XSOLVE_CG;
  else if (str_is(args, "xsolve", "pinv")) d.xsolve = ADMM_XSOLVE_PINV;

  if (p == "lasso") {
    if (!haveD) mexErrMsgIdAndTxt("admm:arg", "args.D must be a full real matrix");
    d.lambda = opt_scalar(args, "lambda", 0.0);
    ds.nA = ds.nB = d.n;
    d.s = opt_vec(args, "s");
    // the serial args struct has no s (getProxOps.m:445-451): the objective's s travels in handles.s
    if (!d.s) d.s = opt_vec(handles, "s");
    if (opt_scalar(args, "parallel", 0) != 0) {  // getProxOps.m:383-442
      d.problem = ADMM_PROB_LASSO_CONSENSUS;
      size_t k = 0;
      const double* sl = opt_vec(args, "slices", &k);
      if (!sl) mexErrMsgIdAndTxt("admm:arg", "consensus lasso needs args.slices (lasso.m:205-208)");
      ds.slices.resize(k);
      for (size_t i = 0; i < k; ++i) ds.slices[i] = static_cast<int64_t>(std::floor(sl[i] + 0.5));
      d.nslices = static_cast<int32_t>(k);
      d.slices = ds.slices.data();
    } else {
      d.problem = ADMM_PROB_LASSO;
      d.Dts = opt_vec(args, "Dts");
      d.L = factor_arg(args, "L", ds, d.m < d.n ? d.m : d.n);  // lasso.m:168 / 172
      d.obj_gram = static_cast<int32_t>(opt_scalar(args, "objgram", 0.0));
    }
  } else if (p == "lad" || p == "huberfit") {
    if (!haveD) mexErrMsgIdAndTxt("admm:arg", "args.D must be a full real matrix");
    d.problem = p == "lad" ? ADMM_PROB_LAD : ADMM_PROB_HUBERFIT;
    d.s = opt_vec(args, "s");
    d.L = factor_arg(args, "R", ds, d.n);  // lad.m:134, huberfit.m:166
    d.userelax = static_cast<int32_t>(opt_scalar(args, "userelax", 0.0));
    ds.nA = d.n;
    ds.nB = d.m;
    ds.a_matrix = true;
  } else if (p == "linearsvm") {
    if (!haveD) mexErrMsgIdAndTxt("admm:arg", "args.D must be a full real matrix");
    d.problem = ADMM_PROB_LINEARSVM;
    d.ell = opt_vec(args, "ell");
    d.C = opt_scalar(args, "C", 0.0);
    const mxArray* lf = field(args, "lossfunction");
    d.loss = !lf ? ADMM_LOSS_HINGE
                 : str_is(args, "lossfunction", "01") ? ADMM_LOSS_01
                 : str_is(args, "lossfunction", "hinge") ? ADMM_LOSS_HINGE : ADMM_LOSS_HINGE_OBJ01;  // linearsvmtest.m:160
    const mxArray* Dp = field(args, "Dplus");  // linearsvm.m:185-186
    if (is_dense_double(Dp) && static_cast<int64_t>(mxGetM(Dp)) == d.n && static_cast<int64_t>(mxGetN(Dp)) == d.m)
      d.Dplus = mxGetPr(Dp);
    ds.nA = d.n;
    ds.nB = d.m;
    ds.a_matrix = true;
  } else if (p == "totalvariation") {  // args.D is the sparse difference operator: implicit on the device
    size_t k = 0;
    d.s = opt_vec(args, "s", &k);
    d.problem = ADMM_PROB_TOTALVARIATION;
    d.D = nullptr;
    d.m = d.n = static_cast<int64_t>(k);
    d.lambda = opt_scalar(args, "lambda", 0.0);
    ds.nA = ds.nB = d.n;
    ds.a_matrix = true;
  } else if (p == "totalvariation2d") {  // engine-side extension: args.S is the H x W image
    const mxArray* S = field(args, "S");
    if (!is_dense_double(S)) mexErrMsgIdAndTxt("admm:arg", "args.S must be a full real image");
    d.problem = ADMM_PROB_TV2D;
    d.D = nullptr;
    d.s = mxGetPr(S);
    d.m = static_cast<int64_t>(mxGetM(S));
    d.n = static_cast<int64_t>(mxGetN(S));
    d.lambda = opt_scalar(args, "lambda", 0.0);
    ds.nA = d.m * d.n;
    ds.nB = 2 * ds.nA;
    ds.a_matrix = true;
  } else if (p == "quadraticprogram" && !str_is(args, "constraint", "standard")) {  // getProxOps.m:631-641
    const mxArray* P = field(args, "P");
    if (!is_dense_double(P)) mexErrMsgIdAndTxt("admm:arg", "args.P must be a full real matrix");
    d.problem = ADMM_PROB_QP_BOUNDED;
    d.D = nullptr;
    d.P = mxGetPr(P);
    d.m = d.n = static_cast<int64_t>(mxGetN(P));
    d.q = opt_vec(args, "q");
    d.lb = opt_vec(args, "lb");
    d.ub = opt_vec(args, "ub");
    d.r = opt_scalar(handles, "r", opt_scalar(args, "r", 0.0));
    ds.nA = ds.nB = d.n;
  } else if (p == "quadraticprogram" || p == "linearprogram") {
    // getProxOps.m:1363 / 1410 solve the (n+m) x (n+m) KKT system in every x-update; the engine eliminates the
    // multiplier once, on the device, from args.D / args.s for args.rho (a caller may also hand in the map: args.K, k0)
    const mxArray* K = field(args, "K");
    const bool qp = p == "quadraticprogram";
    d.problem = qp ? ADMM_PROB_QP_STANDARD : ADMM_PROB_LINEARPROGRAM;
    if (is_dense_double(K)) {
      d.D = nullptr;
      d.K = mxGetPr(K);
      d.k0 = opt_vec(args, "k0");
      d.m = d.n = static_cast<int64_t>(mxGetN(K));
    } else {
      if (!haveD) mexErrMsgIdAndTxt("admm:arg", "args.D must be a full real matrix");
      d.s = opt_vec(args, "s");
    }
    d.q = qp ? opt_vec(args, "q") : opt_vec(args, "b");
    if (qp) {
      const mxArray* P = field(args, "P");
      if (is_dense_double(P)) d.P = mxGetPr(P);
      d.r = opt_scalar(handles, "r", opt_scalar(args, "r", 0.0));
    }
    ds.nA = ds.nB = d.n;
  } else if (p == "basispursuit") {  // getProxOps.m:137-138
    // args = {P, q} as basispursuit.m:116-120 forms them; with args = {D, s} the engine forms both on the device
    const mxArray* P = field(args, "P");
    d.problem = ADMM_PROB_BASISPURSUIT;
    if (is_dense_double(P)) {
      d.D = nullptr;
      d.P = mxGetPr(P);
      d.m = d.n = static_cast<int64_t>(mxGetN(P));
      d.q = opt_vec(args, "q");
    } else {
      if (!haveD) mexErrMsgIdAndTxt("admm:arg", "args.P (or args.D and args.s) must be full real matrices");
      d.s = opt_vec(args, "s");
    }
    ds.nA = ds.nB = d.n;
  } else if (p == "model") {  // getProxOps.m:83-89
    d.problem = ADMM_PROB_MODEL;
    d.D = nullptr;
    d.n = static_cast<int64_t>(opt_scalar(args, "n", 0.0));
    d.m = d.n;
    d.P = opt_vec(args, "PtP");
    d.q = opt_vec(args, "Ptr");
    d.Q = opt_vec(args, "QtQ");
    d.qz = opt_vec(args, "Qts");
    d.c = opt_vec(args, "c");
    ds.nA = ds.nB = d.n;
  } else if (p == "generic") {
    // results = admm(xminf, zming, options) with both handles the caller's (admm.m:24).  A = 1 runs as the model
    // problem without data; a constraint matrix A, or function handles A / At (handles.A, handles.At; admm.m:113-195),
    // as the LAD shape without a factor.  B: the shorthand -1, another scalar, an m x nB matrix (args.B) or a function
    // handle (handles.B with args.nB) -- admm.m:198-245.
    const mxArray* A = field(args, "A");
    const bool a_fh = is_handle(field(handles, "A"));
    if (a_fh || (is_dense_double(A) && mxGetNumberOfElements(A) > 1)) {
      d.problem = ADMM_PROB_LAD;
      d.xsolve = ADMM_XSOLVE_CALLBACK;
      if (a_fh) {
        if (!is_handle(field(handles, "At")))
          mexErrMsgIdAndTxt("admm:arg", "options.A is a function handle: options.At must be one too (admm.m:139-158)");
        d.D = nullptr;
        d.m = static_cast<int64_t>(opt_scalar(args, "m", 0.0));
        d.n = static_cast<int64_t>(opt_scalar(args, "nA", opt_scalar(args, "n", 0.0)));
        if (d.m <= 0 || d.n <= 0)
          mexErrMsgIdAndTxt("admm:arg", "Matrix A is a function handle, but no number of columns nA (or rows m) "
                                        "specified for it");
        ds.a_handle = true;
      } else {
        d.D = mxGetPr(A);
        d.m = static_cast<int64_t>(mxGetM(A));
        d.n = static_cast<int64_t>(mxGetN(A));
        d.ldD = d.m;
      }
      d.s = opt_vec(args, "c");
      ds.zeros.assign(static_cast<size_t>(d.m), 0.0);
      if (!d.s) d.s = ds.zeros.data();  // c = 0
      ds.nA = d.n;
      ds.nB = d.m;
      ds.a_matrix = true;
    } else {
      d.problem = ADMM_PROB_MODEL;
      d.D = nullptr;
      d.n = static_cast<int64_t>(opt_scalar(args, "n", opt_scalar(args, "nA", 0.0)));
      d.m = d.n;
      d.c = opt_vec(args, "c");
      ds.nA = ds.nB = d.n;
    }
    const mxArray* B = field(args, "B");
    if (is_handle(field(handles, "B"))) {
      ds.b_kind = 3;
      ds.mC = ds.nB;
      ds.nB = static_cast<int64_t>(opt_scalar(args, "nB", 0.0));
      if (ds.nB <= 0)
        mexErrMsgIdAndTxt("admm:arg", "Matrix B is a function handle, but no number of columns nB specified for it; "
                                      "cannot infer nB - please specify it in options struct!");
    } else if (is_dense_double(B) && mxGetNumberOfElements(B) > 1) {
      if (static_cast<int64_t>(mxGetM(B)) != ds.nB)
        mexErrMsgIdAndTxt("admm:arg", "Number of rows in matrix B do not match length of column vector c in "
                                      "constraint Ax + Bz = c");
      ds.b_kind = 2;
      ds.b_matrix = mxGetPr(B);
      ds.b_ld = static_cast<int64_t>(mxGetM(B));
      ds.mC = ds.nB;
      ds.nB = static_cast<int64_t>(mxGetN(B));
    } else if (B && opt_scalar(args, "B", -1.0) != -1.0) {
      ds.b_kind = 1;
      ds.b_scalar = opt_scalar(args, "B", -1.0);
      ds.mC = ds.nB;
    }
  } else {
    mexErrMsgIdAndTxt("admm:problem", "Invalid input for problem - given string is not a solver!");
  }
}

// ---- host-language function handles as engine callbacks ------------------------------------------------------
// The engine hands device pointers; a MATLAB handle works on host mxArrays: stage down, feval, stage up.
struct Thunk {
  mxArray* fh = nullptr;
  size_t nfirst = 0, nB = 0, nU = 0;  // length of the first argument (x or the relaxed Ax-hat), of z and of u
  std::string failure;
};

// feval(handle, ...) with MATLAB's error trapped: an error inside the handle must come back HERE as a status, not unwind
// (long-jump in real MATLAB) through admm_engine_run's frames with the engine mid-iteration
int call_handle(Thunk* t, int nlhs, mxArray* lhs[], int nrhs, mxArray* rhs[]) {
  mxArray* exc = mexCallMATLABWithTrap(nlhs, lhs, nrhs, rhs, "feval");
  if (!exc) return 0;
  if (t->failure.empty()) t->failure = "a MATLAB function handle raised an error inside the ADMM loop";
  mxDestroyArray(exc);
  return 1;
}

mxArray* staged_vector(const double* dev, size_t n, void* stream) {
  mxArray* a = mxCreateDoubleMatrix(n, 1, mxREAL);
  if (admm_memcpy_d2h(mxGetPr(a), dev, n * sizeof(double), stream) != ADMM_OK) {
    mxDestroyArray(a);
    return nullptr;
  }
  return a;
}

int prox_thunk(void* user, const double* x, const double* z, const double* u, double rho, double* out, int64_t nout,
               void* stream) {
  Thunk* t = static_cast<Thunk*>(user);
  mxArray* rhs[5] = {t->fh, staged_vector(x, t->nfirst, stream), staged_vector(z, t->nB, stream),
                     staged_vector(u, t->nU, stream), mxCreateDoubleScalar(rho)};
  mxArray* lhs[1] = {nullptr};
  int rc = (rhs[1] && rhs[2] && rhs[3]) ? call_handle(t, 1, lhs, 5, rhs) : 1;
  if (rc == 0 && (!is_dense_double(lhs[0]) || mxGetNumberOfElements(lhs[0]) != static_cast<size_t>(nout))) {
    t->failure = "a proximal-operator handle returned something that is not a full real vector of the expected length";
    rc = 1;
  }
  if (rc == 0) rc = admm_memcpy_h2d(out, mxGetPr(lhs[0]), static_cast<size_t>(nout) * sizeof(double), stream);
  for (int k = 1; k < 5; ++k)
    if (rhs[k]) mxDestroyArray(rhs[k]);
  if (lhs[0]) mxDestroyArray(lhs[0]);
  return rc;
}

// options.A / At / B as MATLAB function handles of a single vector (admm.m:117-158, 206-216)
int op_thunk(void* user, const double* in, int64_t nin, double* out, int64_t nout, void* stream) {
  Thunk* t = static_cast<Thunk*>(user);
  mxArray* rhs[2] = {t->fh, staged_vector(in, static_cast<size_t>(nin), stream)};
  mxArray* lhs[1] = {nullptr};
  int rc = rhs[1] ? call_handle(t, 1, lhs, 2, rhs) : 1;
  if (rc == 0 && (!is_dense_double(lhs[0]) || mxGetNumberOfElements(lhs[0]) != static_cast<size_t>(nout))) {
    t->failure = "a constraint-operator handle (A, At or B) returned something that is not a full real vector of the "
                 "expected length";
    rc = 1;
  }
  if (rc == 0) rc = admm_memcpy_h2d(out, mxGetPr(lhs[0]), static_cast<size_t>(nout) * sizeof(double), stream);
  if (rhs[1]) mxDestroyArray(rhs[1]);
  if (lhs[0]) mxDestroyArray(lhs[0]);
  return rc;
}

// options.altu(u, Ax, Bz, c) and options.specialnorms(x, z, u, rho) as MATLAB handles (admm.m:553-559, 612-616)
int altu_thunk(void* user, const double* u, const double* ax, const double* bz, const double* c, int64_t m, double* out,
               void* stream) {
  Thunk* t = static_cast<Thunk*>(user);
  const size_t n = static_cast<size_t>(m);
  mxArray* rhs[5] = {t->fh, staged_vector(u, n, stream), staged_vector(ax, n, stream), staged_vector(bz, n, stream),
                     staged_vector(c, n, stream)};
  mxArray* lhs[1] = {nullptr};
  int rc = (rhs[1] && rhs[2] && rhs[3] && rhs[4]) ? call_handle(t, 1, lhs, 5, rhs) : 1;
  if (rc == 0 && (!is_dense_double(lhs[0]) || mxGetNumberOfElements(lhs[0]) != n)) {
    t->failure = "options.altu returned something that is not a full real vector of the length of u";
    rc = 1;
  }
  if (rc == 0) rc = admm_memcpy_h2d(out, mxGetPr(lhs[0]), n * sizeof(double), stream);
  for (int k = 1; k < 5; ++k)
    if (rhs[k]) mxDestroyArray(rhs[k]);
  if (lhs[0]) mxDestroyArray(lhs[0]);
  return rc;
}

int norms_thunk(void* user, const double* x, int64_t nA, const double* z, int64_t nB, const double* u, int64_t m,
                double rho, double* out2, void* stream) {
  Thunk* t = static_cast<Thunk*>(user);
  mxArray* rhs[5] = {t->fh, staged_vector(x, static_cast<size_t>(nA), stream), staged_vector(z, static_cast<size_t>(nB), stream),
                     staged_vector(u, static_cast<size_t>(m), stream), mxCreateDoubleScalar(rho)};
  mxArray* lhs[1] = {nullptr};
  int rc = (rhs[1] && rhs[2] && rhs[3]) ? call_handle(t, 1, lhs, 5, rhs) : 1;
  if (rc == 0 && (!is_dense_double(lhs[0]) || mxGetNumberOfElements(lhs[0]) < 2)) {
    t->failure = "options.specialnorms must return a vector of two values (admm.m:613-616)";
    rc = 1;
  }
  if (rc == 0) rc = admm_memcpy_h2d(out2, mxGetPr(lhs[0]), 2 * sizeof(double), stream);
  for (int k = 1; k < 5; ++k)
    if (rhs[k]) mxDestroyArray(rhs[k]);
  if (lhs[0]) mxDestroyArray(lhs[0]);
  return rc;
}

int obj_thunk(void* user, const double* x, int64_t nA, const double* z, int64_t nB, double* out, void* stream) {
  Thunk* t = static_cast<Thunk*>(user);
  mxArray* rhs[3] = {t->fh, staged_vector(x, static_cast<size_t>(nA), stream), staged_vector(z, static_cast<size_t>(nB), stream)};
  mxArray* lhs[1] = {nullptr};
  int rc = (rhs[1] && rhs[2]) ? call_handle(t, 1, lhs, 3, rhs) : 1;
  if (rc == 0 && (!lhs[0] || mxGetNumberOfElements(lhs[0]) != 1)) rc = 1;
  if (rc == 0) {
    const double v = mxGetScalar(lhs[0]);
    rc = admm_memcpy_h2d(out, &v, sizeof(double), stream);
  }
  for (int k = 1; k < 3; ++k)
    if (rhs[k]) mxDestroyArray(rhs[k]);
  if (lhs[0]) mxDestroyArray(lhs[0]);
  return rc;
}

// ---- results ----------------------------------------------------------------------------------------------------
mxArray* fetch_matrix(admm_engine* e, int fld, size_t rows, size_t cols) {
  mxArray* m = mxCreateDoubleMatrix(rows, cols, mxREAL);
  size_t n = 0;
  if (rows * cols == 0 || (admm_engine_fetch(e, fld, mxGetPr(m), rows * cols, &n) == ADMM_OK && n == rows * cols)) return m;
  mxDestroyArray(m);
  return nullptr;
}

void put(mxArray* res, const char* name, mxArray* v) {
  if (!v) return;
  mxAddField(res, name);
  mxSetField(res, 0, name, v);
}

void fetch_into(mxArray* res, admm_engine* e, const char* name, int fld, size_t rows, size_t cols) {
  put(res, name, fetch_matrix(e, fld, rows, cols));
}

mxArray* start_vector(const mxArray* op, const char* name, size_t n) {  // results.x0 / z0 / u0, admm.m:252-259
  mxArray* a = mxCreateDoubleMatrix(n, 1, mxREAL);
  size_t k = 0;
  const double* v = opt_vec(op, name, &k);
  if (v && k == n) std::memcpy(mxGetPr(a), v, n * sizeof(double));
  return a;
}

void read_options(const mxArray* op, admm_options& o) {
  admm_options_default(&o);  // setopt defaults, admm.m:780-971
  o.rho = opt_scalar(op, "rho", o.rho);
  o.maxiters = static_cast<int32_t>(std::ceil(opt_scalar(op, "maxiters", o.maxiters)));  // admm.m:334-339
  o.domaxiters = static_cast<int32_t>(opt_scalar(op, "domaxiters", 0));
  o.relax = opt_scalar(op, "relax", 1.0);
  o.abstol = opt_scalar(op, "abstol", o.abstol);
  o.reltol = opt_scalar(op, "reltol", o.reltol);
  o.Hnormtol = opt_scalar(op, "Hreltol", opt_scalar(op, "Hnormtol", o.Hnormtol));  // quirk q2: either name
  o.convtol = opt_scalar(op, "convtol", o.convtol);
  o.restart = opt_scalar(op, "restart", o.restart);
  o.dvaltol = opt_scalar(op, "dvaltol", o.dvaltol);
  o.objevals = static_cast<int32_t>(opt_scalar(op, "objevals", 0));
  o.convtest = static_cast<int32_t>(opt_scalar(op, "convtest", 0));
  o.nodualerror = static_cast<int32_t>(opt_scalar(op, "nodualerror", 0));
  o.record_history = static_cast<int32_t>(opt_scalar(op, "recordhistory", 1));
  o.stale_factor_ok = static_cast<int32_t>(opt_scalar(op, "stalefactorok", 0));
  if (opt_scalar(op, "fast", 0) != 0) o.fast = str_is(op, "fasttype", "strong") ? ADMM_FAST_STRONG : ADMM_FAST_WEAK;
  o.stopcond = str_is(op, "stopcond", "hnorm") ? ADMM_STOP_HNORM
               : str_is(op, "stopcond", "both") ? ADMM_STOP_BOTH
               : (field(op, "stopcond") && !str_is(op, "stopcond", "standard")) ? ADMM_STOP_NONE
                                                                                   : ADMM_STOP_STANDARD;
}

// Everything about a run that can be refused WITHOUT an engine: called before admm_engine_create in 'solve', so that a
// bad options struct never leaves device memory behind (at config 2's size an engine holds 8.4 GB).
void validate_run(const Desc& ds, const mxArray* op, const mxArray* handles) {
  if (!mxIsStruct(op)) mexErrMsgIdAndTxt("admm:arg", "Given options is not a struct! At least pass empty struct!");
  const size_t nA = static_cast<size_t>(ds.nA), nB = static_cast<size_t>(ds.nB);
  const size_t nU = ds.mC ? static_cast<size_t>(ds.mC) : nB;  // u, c, A*x (admm.m:252-254: zeros(m, 1))
  size_t k0 = 0;
  if (opt_vec(op, "x0", &k0) && k0 != nA) mexErrMsgIdAndTxt("admm:arg", "options.x0 has the wrong length");
  if (opt_vec(op, "z0", &k0) && k0 != nB) mexErrMsgIdAndTxt("admm:arg", "options.z0 has the wrong length");
  if (opt_vec(op, "u0", &k0) && k0 != nU) mexErrMsgIdAndTxt("admm:arg", "options.u0 has the wrong length");
  if (ds.a_handle && (!is_handle(field(handles, "A")) || !is_handle(field(handles, "At"))))
    mexErrMsgIdAndTxt("admm:arg", "this engine was created for function-handle operators: pass handles.A and handles.At");
  if (ds.b_kind == 3 && !is_handle(field(handles, "B")))
    mexErrMsgIdAndTxt("admm:arg", "this engine was created for a function-handle B: pass handles.B");
  for (const char* name : {"altu", "specialnorms"})  // the consensus hooks are descriptors, not handles (getproxops.m)
    if (is_handle(field(handles, name)) && ds.d.problem == ADMM_PROB_LASSO_CONSENSUS)
      mexErrMsgIdAndTxt("admm:unsupported", "consensus lasso runs with the hooks of its own getproxops call");
}

struct RunError {  // a failure after an engine exists: reported by the caller once the engine is dealt with
  std::string id, msg;
  bool set(const char* i, const std::string& m) {
    id = i;
    msg = m;
    return false;
  }
};

// the loop and the whole results struct of admm.m:257-767 (options / solverruntime are added by the callers).
// Never raises a MATLAB error itself: nullptr + err on failure (validate_run has vetted the arguments).
mxArray* run_engine(admm_engine* e, const Desc& ds, const mxArray* op, const mxArray* handles, RunError& err) {
  admm_options o;
  read_options(op, o);
  const size_t nA = static_cast<size_t>(ds.nA), nB = static_cast<size_t>(ds.nB);
  const size_t nU = ds.mC ? static_cast<size_t>(ds.mC) : nB;  // u, c, A*x (admm.m:252-254: zeros(m, 1))
  size_t k0 = 0;
  o.x0 = opt_vec(op, "x0", &k0);
  o.z0 = opt_vec(op, "z0", &k0);
  o.u0 = opt_vec(op, "u0", &k0);
  auto engine_ok = [&](int rc) { return rc == ADMM_OK || err.set("admm:engine", admm_last_error()); };

  // caller-supplied handles replace the engine-native operators (admm.m:502, 521-530, 603-605)
  Thunk tx, tz, tobj, ta, tat, tb, taltu, tnorms;
  if (ds.a_handle) {  // the thunks live as long as this run; a persistent engine gets fresh ones every run
    ta.fh = const_cast<mxArray*>(field(handles, "A"));
    tat.fh = const_cast<mxArray*>(field(handles, "At"));
    if (!engine_ok(admm_engine_set_operators(e, op_thunk, &ta, op_thunk, &tat))) return nullptr;
  }
  if (ds.b_kind == 3) {
    tb.fh = const_cast<mxArray*>(field(handles, "B"));
    if (!engine_ok(admm_engine_set_constraint_b(e, nullptr, 0, ds.nB, ADMM_MEM_HOST, 0.0, op_thunk, &tb))) return nullptr;
  }
  // options.altu / options.specialnorms as MATLAB handles (admm.m:553-559, 612-616): matlab/admm.m hands them over in
  // `handles` (the consensus descriptors of getproxops' extra stay in options and are the engine's own)
  if (is_handle(field(handles, "altu"))) taltu.fh = const_cast<mxArray*>(field(handles, "altu"));
  if (is_handle(field(handles, "specialnorms"))) tnorms.fh = const_cast<mxArray*>(field(handles, "specialnorms"));
  const bool hooks = taltu.fh || tnorms.fh;
  if (hooks && !engine_ok(admm_engine_set_hooks(e, taltu.fh ? altu_thunk : nullptr, &taltu,
                                                tnorms.fh ? norms_thunk : nullptr, &tnorms)))
    return nullptr;
  const mxArray* fx = field(handles, "xminf");
  const mxArray* fz = field(handles, "zming");
  const mxArray* fo = field(handles, "obj");
  const bool relaxed = o.relax != 1.0;
  if (is_handle(fx)) {
    tx.fh = const_cast<mxArray*>(fx);
    tx.nfirst = nA;
    tx.nB = nB;
    tx.nU = nU;
  }
  if (is_handle(fz)) {
    tz.fh = const_cast<mxArray*>(fz);
    tz.nfirst = relaxed ? nU : nA;  // zming(x, ...) or zming(Axhat, ...) (admm.m:521-530)
    tz.nB = nB;
    tz.nU = nU;
  }
  const bool obj_native = opt_scalar(handles, "objnative", 0.0) != 0.0;
  if (o.objevals && !obj_native && is_handle(fo)) {
    tobj.fh = const_cast<mxArray*>(fo);
  } else if (o.objevals && !obj_native) {
    o.objevals = 0;  // admm.m:603: objevals without options.obj records nothing
  }
  admm_run_summary s;
  int rc = ADMM_OK;
  if (tx.fh || tz.fh || tobj.fh)
    rc = admm_engine_set_callbacks(e, tx.fh ? prox_thunk : nullptr, &tx, tz.fh ? prox_thunk : nullptr, &tz,
                                   tobj.fh ? obj_thunk : nullptr, &tobj);
  if (rc == ADMM_OK) rc = admm_engine_run(e, &o, &s);
  const std::string engine_msg = rc == ADMM_OK ? std::string() : std::string(admm_last_error());
  if (tx.fh || tz.fh || tobj.fh) (void)admm_engine_set_callbacks(e, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (hooks) (void)admm_engine_set_hooks(e, nullptr, nullptr, nullptr, nullptr);
  if (rc != ADMM_OK) {
    for (const Thunk* t : {&tx, &tz, &tobj, &ta, &tat, &tb, &taltu, &tnorms})
      if (!t->failure.empty()) {
        err.set("admm:handle", t->failure);
        return nullptr;
      }
    err.set("admm:engine", engine_msg);
    return nullptr;
  }

  const size_t k = static_cast<size_t>(s.steps);
  const bool use_h = o.convtest || o.stopcond == ADMM_STOP_HNORM || o.stopcond == ADMM_STOP_BOTH;
  mxArray* res = mxCreateStructMatrix(1, 1, 0, nullptr);
  put(res, "x0", start_vector(op, "x0", nA));  // admm.m:257-259
  put(res, "z0", start_vector(op, "z0", nB));
  put(res, "u0", start_vector(op, "u0", nU));
  if (o.fast == ADMM_FAST_WEAK) put(res, "dvaltol", mxCreateDoubleScalar(o.dvaltol));  // admm.m:292
  if (use_h) put(res, "Hnormtol", mxCreateDoubleScalar(o.Hnormtol));                    // admm.m:312
  mxArray *xv = nullptr, *zv = nullptr, *uv = nullptr;
  if (o.record_history) {
    xv = fetch_matrix(e, ADMM_F_XVALS, nA, k);
    zv = fetch_matrix(e, ADMM_F_ZVALS, nB, k);
    uv = fetch_matrix(e, ADMM_F_UVALS, nU, k);
    if (use_h && xv && zv && uv) {  // admm.m:678-681  w = [x; z; rho*u]
      mxArray* w = mxCreateDoubleMatrix(nA + nB + nU, k, mxREAL);
      double* pw = mxGetPr(w);
      for (size_t i = 0; i < k; ++i) {
        double* col = pw + i * (nA + nB + nU);
        std::memcpy(col, mxGetPr(xv) + i * nA, nA * sizeof(double));
        std::memcpy(col + nA, mxGetPr(zv) + i * nB, nB * sizeof(double));
        const double* ui = mxGetPr(uv) + i * nU;
        for (size_t j = 0; j < nU; ++j) col[nA + nB + j] = o.rho * ui[j];
      }
      put(res, "wvals", w);
    }
    put(res, "xvals", xv);
    put(res, "zvals", zv);
    put(res, "uvals", uv);
    if (o.fast != ADMM_FAST_OFF) {
      fetch_into(res, e, "vvals", ADMM_F_VVALS, nB, k);
      fetch_into(res, e, "uhatvals", ADMM_F_UHATVALS, nU, k);
    }
  }
  if (o.fast != ADMM_FAST_WEAK) {  // q8: accelerated ADMM records no norms (admm.m:619-640)
    fetch_into(res, e, "pnorm", ADMM_F_PNORM, 1, k);
    fetch_into(res, e, "dnorm", ADMM_F_DNORM, 1, k);
    fetch_into(res, e, "perr", ADMM_F_PERR, 1, k);
    fetch_into(res, e, "derr", ADMM_F_DERR, 1, k);
  }
  if (o.objevals) fetch_into(res, e, "objevals", ADMM_F_OBJEVALS, 1, k);
  if (use_h) fetch_into(res, e, "Hnormsq", ADMM_F_HNORMSQ, 1, k);
  if (o.fast != ADMM_FAST_OFF) {
    fetch_into(res, e, "avals", ADMM_F_AVALS, 1, k);
    if (o.fast == ADMM_FAST_WEAK) {
      fetch_into(res, e, "dvals", ADMM_F_DVALS, 1, k);
      fetch_into(res, e, "restarted", ADMM_F_RESTARTED, 1, k);
    }
  }
  if (s.convtest_failed_at == 0) {  // q4: the reference returns early without these (admm.m:692-701)
    put(res, "steps", mxCreateDoubleScalar(s.steps));
    fetch_into(res, e, "xopt", ADMM_F_XOPT, nA, 1);
    fetch_into(res, e, "zopt", ADMM_F_ZOPT, nB, 1);
    fetch_into(res, e, "uopt", ADMM_F_UOPT, nU, 1);
    if (ds.d.problem == ADMM_PROB_LASSO_CONSENSUS) fetch_into(res, e, "zconsensus", ADMM_F_ZCONSENSUS, nA, 1);  // q9
    if (o.objevals) put(res, "objopt", mxCreateDoubleScalar(s.objopt));
    put(res, "runtime", mxCreateDoubleScalar(s.runtime_s));
  } else {
    put(res, "convtestfailedat", mxCreateDoubleScalar(s.convtest_failed_at));
  }
  double setup = 0.0;
  if (admm_engine_setup_seconds(e, &setup) == ADMM_OK) put(res, "enginesetupseconds", mxCreateDoubleScalar(setup));
  return res;
}

// engines kept alive between 'create' and 'destroy' remember what describe() learned
struct Live {
  admm_engine* e;
  Desc* ds;
};
std::vector<Live> g_handles;
Desc* g_solve = nullptr;  // the description of the 'solve' call in flight

mxArray* handle_to_mx(admm_engine* e) {
  mxArray* a = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
  *static_cast<uint64_t*>(mxGetData(a)) = reinterpret_cast<uint64_t>(e);
  return a;
}

Live* find_live(const mxArray* a) {
  if (!a || mxGetClassID(a) != mxUINT64_CLASS) mexErrMsgIdAndTxt("admm:handle", "bad engine handle");
  admm_engine* e = reinterpret_cast<admm_engine*>(*static_cast<uint64_t*>(mxGetData(a)));
  for (Live& l : g_handles)
    if (l.e == e) return &l;
  mexErrMsgIdAndTxt("admm:handle", "engine handle is not live");
  return nullptr;
}

void at_exit_all() {
  at_exit();
  delete g_solve;
  g_solve = nullptr;
  for (Live& l : g_handles) {
    admm_engine_destroy(l.e);
    delete l.ds;
  }
  g_handles.clear();
}

}  // namespace

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  (void)nlhs;
  if (nrhs < 1) mexErrMsgIdAndTxt("admm:arg", "usage: admm_mex(command, ...)");
  const std::string cmd = to_string(prhs[0]);
  if (cmd == "available") {
    int n = 0;
    plhs[0] = mxCreateLogicalScalar(admm_device_count(&n) == ADMM_OK && n > 0);
    return;
  }
  if (cmd == "livecount") {  // engines this gateway currently holds (tests: nothing may survive a failed solve)
    plhs[0] = mxCreateDoubleScalar(static_cast<double>(g_live.size() + g_handles.size()));
    return;
  }
  if (cmd == "solve") {
    if (nrhs < 4) mexErrMsgIdAndTxt("admm:arg", "admm_mex('solve', problem, args, options [, handles])");
    if (!mxIsStruct(prhs[2]))
      mexErrMsgIdAndTxt("admm:arg", "Given struct args is not a struct containing arguments needed for proximal "
                                    "operators for the given problem!");
    const mxArray* handles = nrhs > 4 ? prhs[4] : nullptr;
    // A Desc on the heap, owned by g_solve: a MATLAB error below (describe / validate_run: no engine yet) long-jumps
    // past every C++ destructor of this frame, and the next call (or mexAtExit) frees what it left
    delete g_solve;
    g_solve = new Desc();
    Desc& ds = *g_solve;
    describe(to_string(prhs[1]), prhs[2], handles, ds);
    validate_run(ds, prhs[3], handles);
    static bool registered = false;
    if (!registered) {
      mexAtExit(at_exit_all);
      registered = true;
    }
    admm_engine* e = nullptr;
    check(admm_engine_create(&ds.d, &e));  // (a failed create leaves nothing behind)
    g_live.push_back(e);  // (only an allocation failure inside MATLAB's own mx* calls can still leave it to mexAtExit)
    RunError err;
    mxArray* res = nullptr;
    if (ds.b_kind == 1 || ds.b_kind == 2) {
      if (admm_engine_set_constraint_b(e, ds.b_matrix, ds.b_ld, ds.b_kind == 2 ? ds.nB : 0, ADMM_MEM_HOST, ds.b_scalar,
                                       nullptr, nullptr) != ADMM_OK)
        err.set("admm:engine", admm_last_error());
    }
    if (err.id.empty()) res = run_engine(e, ds, prhs[3], handles, err);
    g_live.pop_back();
    admm_engine_destroy(e);  // before any error is raised: a failed solve returns every byte of device memory
    delete g_solve;
    g_solve = nullptr;
    if (!res) mexErrMsgIdAndTxt(err.id.empty() ? "admm:engine" : err.id.c_str(), "%s", err.msg.c_str());
    plhs[0] = res;
    return;
  }
  if (cmd == "create") {
    if (nrhs < 3 || !mxIsStruct(prhs[2])) mexErrMsgIdAndTxt("admm:arg", "admm_mex('create', problem, args)");
    Desc* ds = new Desc();
    describe(to_string(prhs[1]), prhs[2], nrhs > 3 ? prhs[3] : nullptr, *ds);
    admm_engine* e = nullptr;
    int rc = admm_engine_create(&ds->d, &e);
    if (rc == ADMM_OK && (ds->b_kind == 1 || ds->b_kind == 2)) {
      rc = admm_engine_set_constraint_b(e, ds->b_matrix, ds->b_ld, ds->b_kind == 2 ? ds->nB : 0, ADMM_MEM_HOST,
                                        ds->b_scalar, nullptr, nullptr);
      if (rc != ADMM_OK) admm_engine_destroy(e);
    }
    if (rc != ADMM_OK) {
      delete ds;
      check(rc);
    }
    // the engine copied every array at create: the borrowed pointers in ds->d must not be used again
    static bool registered = false;
    if (!registered) {
      mexAtExit(at_exit_all);
      registered = true;
    }
    g_handles.push_back(Live{e, ds});
    mexLock();
    plhs[0] = handle_to_mx(e);
    return;
  }
  if (cmd == "run") {
    if (nrhs < 3) mexErrMsgIdAndTxt("admm:arg", "admm_mex('run', handle, options [, handles])");
    Live* l = find_live(prhs[1]);
    const mxArray* hd = nrhs > 3 ? prhs[3] : nullptr;
    validate_run(*l->ds, prhs[2], hd);
    RunError err;
    mxArray* res = run_engine(l->e, *l->ds, prhs[2], hd, err);  // (the engine persists by design: 'destroy' frees it)
    if (!res) mexErrMsgIdAndTxt(err.id.c_str(), "%s", err.msg.c_str());
    plhs[0] = res;
    return;
  }
  if (cmd == "destroy") {
    if (nrhs < 2) mexErrMsgIdAndTxt("admm:arg", "admm_mex('destroy', handle)");
    Live* l = find_live(prhs[1]);
    admm_engine_destroy(l->e);
    delete l->ds;
    g_handles.erase(g_handles.begin() + (l - g_handles.data()));
    mexUnlock();
    return;
  }
  mexErrMsgIdAndTxt("admm:arg", "unknown command");
}
