// admm_mex.cpp -- MATLAB MEX gateway to libadmm_hip.so.
//
// NOT built in this repository's image (no MATLAB / mex.h here); compile where MATLAB exists:
//     mex -I../../include admm_mex.cpp -L.. -ladmm_hip
// All logic lives behind the C ABI (include/admm_engine.h), which is what the test-suite
// exercises; this file only converts mxArrays <-> flat pointers.
//
//   ok      = admm_mex('available')
//   h       = admm_mex('create', problem, args)     % args: the struct getproxops receives
//   results = admm_mex('run', h, options)           % options: the struct admm receives
//             admm_mex('destroy', h)
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

const double* opt_vec(const mxArray* s, const char* name) {
  const mxArray* f = field(s, name);
  if (!f || !mxIsDouble(f) || mxIsSparse(f) || mxIsComplex(f)) return nullptr;
  return mxGetPr(f);
}

double opt_scalar(const mxArray* s, const char* name, double dflt) {
  const mxArray* f = field(s, name);
  return (f && mxIsNumeric(f) && mxGetNumberOfElements(f) == 1) ? mxGetScalar(f) : dflt;
}

bool str_is(const mxArray* s, const char* name, const char* value) {
  const mxArray* f = field(s, name);
  return f && mxIsChar(f) && to_string(f) == value;
}

int problem_code(const std::string& p, const mxArray* args) {
  if (p == "lasso") return opt_scalar(args, "parallel", 0) != 0 ? ADMM_PROB_LASSO_CONSENSUS : ADMM_PROB_LASSO;
  if (p == "lad") return ADMM_PROB_LAD;
  if (p == "huberfit") return ADMM_PROB_HUBERFIT;
  if (p == "linearsvm") return ADMM_PROB_LINEARSVM;
  if (p == "totalvariation") return ADMM_PROB_TOTALVARIATION;
  if (p == "quadraticprogram") return ADMM_PROB_QP_BOUNDED;
  if (p == "basispursuit") return ADMM_PROB_BASISPURSUIT;
  if (p == "model") return ADMM_PROB_MODEL;
  mexErrMsgIdAndTxt("admm:problem", "Invalid input for problem - given string is not a solver!");
  return 0;
}

mxArray* handle_to_mx(admm_engine* e) {
  mxArray* a = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
  *static_cast<uint64_t*>(mxGetData(a)) = reinterpret_cast<uint64_t>(e);
  return a;
}

admm_engine* mx_to_handle(const mxArray* a) {
  if (!a || mxGetClassID(a) != mxUINT64_CLASS) mexErrMsgIdAndTxt("admm:handle", "bad engine handle");
  return reinterpret_cast<admm_engine*>(*static_cast<uint64_t*>(mxGetData(a)));
}

void check(int rc) {
  if (rc != ADMM_OK) mexErrMsgIdAndTxt("admm:engine", "%s", admm_last_error());
}

void fetch_into(mxArray* res, admm_engine* e, const char* name, int fld, size_t rows, size_t cols) {
  mxArray* m = mxCreateDoubleMatrix(rows, cols, mxREAL);
  size_t n = 0;
  if (admm_engine_fetch(e, fld, mxGetPr(m), rows * cols, &n) == ADMM_OK && n == rows * cols) {
    mxAddField(res, name);
    mxSetField(res, 0, name, m);
  } else {
    mxDestroyArray(m);
  }
}

void put_scalar(mxArray* res, const char* name, double v) {
  mxAddField(res, name);
  mxSetField(res, 0, name, mxCreateDoubleScalar(v));
}

}  // namespace

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 1) mexErrMsgIdAndTxt("admm:arg", "usage: admm_mex(command, ...)");
  const std::string cmd = to_string(prhs[0]);
  if (cmd == "available") {
    int n = 0;
    plhs[0] = mxCreateLogicalScalar(admm_device_count(&n) == ADMM_OK && n > 0);
    return;
  }
  if (cmd == "create") {
    if (nrhs != 3) mexErrMsgIdAndTxt("admm:arg", "admm_mex('create', problem, args)");
    const mxArray* args = prhs[2];
    if (!mxIsStruct(args))
      mexErrMsgIdAndTxt("admm:arg", "Given struct args is not a struct containing arguments needed for proximal "
                                    "operators for the given problem!");
    admm_problem_desc d;
    admm_problem_desc_default(&d);
    d.problem = problem_code(to_string(prhs[1]), args);
    const mxArray* D = field(args, "D");
    if (D && mxIsDouble(D) && !mxIsSparse(D)) {
      d.D = mxGetPr(D);
      d.m = static_cast<int64_t>(mxGetM(D));
      d.n = static_cast<int64_t>(mxGetN(D));
      d.ldD = d.m;
    }
    const mxArray* P = field(args, "P");
    if (P && mxIsDouble(P) && !mxIsSparse(P)) {
      d.P = mxGetPr(P);
      d.n = static_cast<int64_t>(mxGetN(P));
      if (!D) d.m = d.n;
    }
    const mxArray* s = field(args, "s");
    d.s = opt_vec(args, "s");
    if (!D && !P && s) d.m = d.n = static_cast<int64_t>(mxGetNumberOfElements(s));  // total variation
    d.ell = opt_vec(args, "ell");
    d.q = opt_vec(args, "q");
    d.lb = opt_vec(args, "lb");
    d.ub = opt_vec(args, "ub");
    d.L = opt_vec(args, "L");  // dense lower factor; a sparse L (lasso.m:175) is ignored -> factored on the GPU
    if (!d.L) d.L = opt_vec(args, "R");
    d.lambda = opt_scalar(args, "lambda", 0.0);
    d.C = opt_scalar(args, "C", 0.0);
    d.r = opt_scalar(args, "r", 0.0);
    d.rho = opt_scalar(args, "rho", 1.0);
    d.userelax = static_cast<int32_t>(opt_scalar(args, "userelax", 0.0));
    d.loss = str_is(args, "lossfunction", "01") ? ADMM_LOSS_01 : ADMM_LOSS_HINGE;
    if (d.problem == ADMM_PROB_MODEL) {  // getProxOps.m:83-89: args.PtP, Ptr, QtQ, Qts, n
      d.n = static_cast<int64_t>(opt_scalar(args, "n", 0.0));
      d.m = d.n;
      d.P = opt_vec(args, "PtP");
      d.q = opt_vec(args, "Ptr");
      d.Q = opt_vec(args, "QtQ");
      d.qz = opt_vec(args, "Qts");
    }
    d.device = static_cast<int32_t>(opt_scalar(args, "device", 0.0));
    admm_engine* e = nullptr;
    check(admm_engine_create(&d, &e));
    if (g_live.empty()) mexAtExit(at_exit);
    g_live.push_back(e);
    mexLock();
    plhs[0] = handle_to_mx(e);
    return;
  }
  if (cmd == "run") {
    if (nrhs != 3) mexErrMsgIdAndTxt("admm:arg", "admm_mex('run', handle, options)");
    admm_engine* e = mx_to_handle(prhs[1]);
    const mxArray* op = prhs[2];
    if (!mxIsStruct(op)) mexErrMsgIdAndTxt("admm:arg", "Given options is not a struct! At least pass empty struct!");
    admm_options o;
    admm_options_default(&o);  // setopt defaults, admm.m:780-971
    o.rho = opt_scalar(op, "rho", o.rho);
    o.maxiters = static_cast<int32_t>(opt_scalar(op, "maxiters", o.maxiters));
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
    if (opt_scalar(op, "fast", 0) != 0) o.fast = str_is(op, "fasttype", "strong") ? ADMM_FAST_STRONG : ADMM_FAST_WEAK;
    o.stopcond = str_is(op, "stopcond", "hnorm") ? ADMM_STOP_HNORM
                 : str_is(op, "stopcond", "both") ? ADMM_STOP_BOTH
                 : (field(op, "stopcond") && !str_is(op, "stopcond", "standard")) ? ADMM_STOP_NONE
                                                                                     : ADMM_STOP_STANDARD;
    o.x0 = opt_vec(op, "x0");
    o.z0 = opt_vec(op, "z0");
    o.u0 = opt_vec(op, "u0");
    admm_run_summary s;
    check(admm_engine_run(e, &o, &s));
    const size_t nA = static_cast<size_t>(opt_scalar(op, "nA", 0)), nB = static_cast<size_t>(opt_scalar(op, "nB", 0));
    const size_t k = static_cast<size_t>(s.steps);
    mxArray* res = mxCreateStructMatrix(1, 1, 0, nullptr);
    fetch_into(res, e, "xvals", ADMM_F_XVALS, nA, k);
    fetch_into(res, e, "zvals", ADMM_F_ZVALS, nB, k);
    fetch_into(res, e, "uvals", ADMM_F_UVALS, nB, k);
    fetch_into(res, e, "vvals", ADMM_F_VVALS, nB, k);
    fetch_into(res, e, "uhatvals", ADMM_F_UHATVALS, nB, k);
    if (o.fast != ADMM_FAST_WEAK) {  // q8: accelerated ADMM records no norms (admm.m:619-640)
      fetch_into(res, e, "pnorm", ADMM_F_PNORM, 1, k);
      fetch_into(res, e, "dnorm", ADMM_F_DNORM, 1, k);
      fetch_into(res, e, "perr", ADMM_F_PERR, 1, k);
      fetch_into(res, e, "derr", ADMM_F_DERR, 1, k);
    }
    if (o.objevals) fetch_into(res, e, "objevals", ADMM_F_OBJEVALS, 1, k);
    fetch_into(res, e, "Hnormsq", ADMM_F_HNORMSQ, 1, k);
    fetch_into(res, e, "avals", ADMM_F_AVALS, 1, k);
    fetch_into(res, e, "dvals", ADMM_F_DVALS, 1, k);
    fetch_into(res, e, "restarted", ADMM_F_RESTARTED, 1, k);
    if (s.convtest_failed_at == 0) {  // q4: the reference returns early without these (admm.m:692-701)
      put_scalar(res, "steps", s.steps);
      fetch_into(res, e, "xopt", ADMM_F_XOPT, nA, 1);
      fetch_into(res, e, "zopt", ADMM_F_ZOPT, nB, 1);
      fetch_into(res, e, "uopt", ADMM_F_UOPT, nB, 1);
      if (o.objevals) put_scalar(res, "objopt", s.objopt);
      put_scalar(res, "runtime", s.runtime_s);
    }
    plhs[0] = res;
    return;
  }
  if (cmd == "destroy") {
    admm_engine* e = mx_to_handle(prhs[1]);
    for (size_t i = 0; i < g_live.size(); ++i)
      if (g_live[i] == e) {
        g_live.erase(g_live.begin() + i);
        admm_engine_destroy(e);
        mexUnlock();
        break;
      }
    return;
  }
  mexErrMsgIdAndTxt("admm:arg", "unknown command");
}
