// binding.hip -- the getproxops / admm argument structs interpreted BEHIND the C ABI (admm_engine.h: admm_binding_*).
// A host language flattens its structs into admm_field entries; which field means what for which problem
// (getProxOps.m:52-917), the defaults of admm's options (admm.m:780-971), the argument checks admm makes before its loop
// and the layout of the results struct (admm.m:257-767) are decided here -- host code only, callable without a GPU --
// so that a gateway (csrc/admm_mex.cpp) is a converter of containers and nothing else.
#include <cctype>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

using namespace admm;

struct admm_binding {
  admm_problem_desc d;
  std::string problem;
  std::vector<double> dense_L;    // args.L stored sparse (lasso.m:175) expanded column by column
  std::vector<int64_t> slices;    // args.slices (getProxOps.m:387) as integers
  std::vector<double> zeros;      // c = 0 for the engines that take c as a data vector
  int64_t nA = 0, nB = 0;         // lengths of x and of z
  int64_t mC = 0;                 // length of u and c; 0 = the same as nB (every problem but a generic one with a general B)
  bool a_handle = false;          // generic: options.A / options.At are function handles
  int b_kind = 0;                 // generic: 0 = the shorthand -1, 1 = another scalar, 2 = matrix, 3 = function handle
  double b_scalar = -1.0;
  const double* b_matrix = nullptr;
  int64_t b_ld = 0;
};

namespace {

std::string lower(const char* s) {
  std::string r(s ? s : "");
  for (auto& ch : r) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
  return r;
}

struct View {  // one host struct
  const admm_field* f;
  int32_t n;
  const admm_field* get(const char* name) const {
    for (int32_t i = 0; i < n; ++i)
      if (f[i].name && std::strcmp(f[i].name, name) == 0) return &f[i];
    return nullptr;
  }
  static bool dense(const admm_field* x) { return x && x->kind == ADMM_FIELD_NUMERIC && x->data; }
  const double* vec(const char* name, size_t* count = nullptr) const {
    const admm_field* x = get(name);
    if (!dense(x) || x->rows * x->cols == 0) return nullptr;
    if (count) *count = static_cast<size_t>(x->rows * x->cols);
    return x->data;
  }
  double scalar(const char* name, double dflt) const {
    const admm_field* x = get(name);
    return (dense(x) && x->rows * x->cols == 1) ? x->data[0] : dflt;
  }
  bool text_is(const char* name, const char* value) const {
    const admm_field* x = get(name);
    return x && x->kind == ADMM_FIELD_TEXT && lower(x->text) == value;
  }
  bool handle(const char* name) const {
    const admm_field* x = get(name);
    return x && x->kind == ADMM_FIELD_HANDLE;
  }
};

// a lower-triangular factor handed in as args.L / args.R: dense as is, sparse (CSC) expanded
const double* factor_arg(const View& args, const char* name, admm_binding& b, int64_t order) {
  const admm_field* f = args.get(name);
  if (!f || (f->kind != ADMM_FIELD_NUMERIC && f->kind != ADMM_FIELD_SPARSE) || !f->data) return nullptr;
  if (f->rows != order || f->cols != order) return nullptr;
  if (f->kind == ADMM_FIELD_NUMERIC) return f->data;
  if (!f->ir || !f->jc) return nullptr;
  b.dense_L.assign(static_cast<size_t>(order) * order, 0.0);
  for (int64_t j = 0; j < order; ++j)
    for (uint64_t k = f->jc[j]; k < f->jc[j + 1]; ++k)
      b.dense_L[static_cast<size_t>(f->ir[k]) + static_cast<size_t>(j) * order] = f->data[k];
  return b.dense_L.data();
}

int describe(const std::string& p, const View& args, const View& handles, admm_binding& b) {
  admm_problem_desc& d = b.d;
  admm_problem_desc_default(&d);
  const admm_field* D = args.get("D");
  const bool haveD = View::dense(D);
  if (haveD) {
    d.D = D->data;
    d.m = D->rows;
    d.n = D->cols;
    d.ldD = d.m;
  }
  d.rho = args.scalar("rho", 1.0);
  d.device = static_cast<int32_t>(args.scalar("device", 0.0));
  if (args.text_is("xsolve", "trsv")) d.xsolve = ADMM_XSOLVE_TRSV;
  else if (args.text_is("xsolve", "inverse")) d.xsolve = ADMM_XSOLVE_INVERSE;
  else if (args.text_is("xsolve", "cg")) d.xsolve = ADMM_XSOLVE_CG;
  else if (args.text_is("xsolve", "pinv")) d.xsolve = ADMM_XSOLVE_PINV;
  const char* needD = "args.D must be a full real matrix";

  if (p == "lasso") {
    if (!haveD) return fail(ADMM_E_INVALID, needD);
    d.lambda = args.scalar("lambda", 0.0);
    b.nA = b.nB = d.n;
    d.s = args.vec("s");
    // the serial args struct has no s (getProxOps.m:445-451): the objective's s travels in handles.s
    if (!d.s) d.s = handles.vec("s");
    if (args.scalar("parallel", 0) != 0) {  // getProxOps.m:383-442
      d.problem = ADMM_PROB_LASSO_CONSENSUS;
      size_t k = 0;
      const double* sl = args.vec("slices", &k);
      if (!sl) return fail(ADMM_E_INVALID, "consensus lasso needs args.slices (lasso.m:205-208)");
      b.slices.resize(k);
      for (size_t i = 0; i < k; ++i) b.slices[i] = static_cast<int64_t>(std::floor(sl[i] + 0.5));
      d.nslices = static_cast<int32_t>(k);
      d.slices = b.slices.data();
    } else {
      d.problem = ADMM_PROB_LASSO;
      d.Dts = args.vec("Dts");
      d.L = factor_arg(args, "L", b, d.m < d.n ? d.m : d.n);  // lasso.m:168 / 172
      d.obj_gram = static_cast<int32_t>(args.scalar("objgram", 0.0));
    }
  } else if (p == "lad" || p == "huberfit") {
    if (!haveD) return fail(ADMM_E_INVALID, needD);
    d.problem = p == "lad" ? ADMM_PROB_LAD : ADMM_PROB_HUBERFIT;
    d.s = args.vec("s");
    d.L = factor_arg(args, "R", b, d.n);  // lad.m:134, huberfit.m:166
    d.userelax = static_cast<int32_t>(args.scalar("userelax", 0.0));
    b.nA = d.n;
    b.nB = d.m;
  } else if (p == "linearsvm") {
    if (!haveD) return fail(ADMM_E_INVALID, needD);
    d.problem = ADMM_PROB_LINEARSVM;
    d.ell = args.vec("ell");
    d.C = args.scalar("C", 0.0);
    d.loss = !args.get("lossfunction")            ? ADMM_LOSS_HINGE
             : args.text_is("lossfunction", "01") ? ADMM_LOSS_01
             : args.text_is("lossfunction", "hinge") ? ADMM_LOSS_HINGE
                                                     : ADMM_LOSS_HINGE_OBJ01;  // linearsvmtest.m:160
    const admm_field* Dp = args.get("Dplus");  // linearsvm.m:185-186
    if (View::dense(Dp) && Dp->rows == d.n && Dp->cols == d.m) d.Dplus = Dp->data;
    b.nA = d.n;
    b.nB = d.m;
  } else if (p == "totalvariation") {  // args.D is the sparse difference operator: implicit on the device
    size_t k = 0;
    d.s = args.vec("s", &k);
    d.problem = ADMM_PROB_TOTALVARIATION;
    d.D = nullptr;
    d.m = d.n = static_cast<int64_t>(k);
    d.lambda = args.scalar("lambda", 0.0);
    b.nA = b.nB = d.n;
  } else if (p == "totalvariation2d") {  // engine-side extension: args.S is the H x W image
    const admm_field* S = args.get("S");
    if (!View::dense(S)) return fail(ADMM_E_INVALID, "args.S must be a full real image");
    d.problem = ADMM_PROB_TV2D;
    d.D = nullptr;
    d.s = S->data;
    d.m = S->rows;
    d.n = S->cols;
    d.lambda = args.scalar("lambda", 0.0);
    b.nA = d.m * d.n;
    b.nB = 2 * b.nA;
  } else if (p == "quadraticprogram" && !args.text_is("constraint", "standard")) {  // getProxOps.m:631-641
    const admm_field* P = args.get("P");
    if (!View::dense(P)) return fail(ADMM_E_INVALID, "args.P must be a full real matrix");
    d.problem = ADMM_PROB_QP_BOUNDED;
    d.D = nullptr;
    d.P = P->data;
    d.m = d.n = P->cols;
    d.q = args.vec("q");
    d.lb = args.vec("lb");
    d.ub = args.vec("ub");
    d.r = handles.scalar("r", args.scalar("r", 0.0));
    b.nA = b.nB = d.n;
  } else if (p == "quadraticprogram" || p == "linearprogram") {
    // getProxOps.m:1363 / 1410 solve the (n+m) x (n+m) KKT system in every x-update; the engine eliminates the
    // multiplier once, on the device, from args.D / args.s for args.rho (a caller may also hand in the map: args.K, k0)
    const admm_field* K = args.get("K");
    const bool qp = p == "quadraticprogram";
    d.problem = qp ? ADMM_PROB_QP_STANDARD : ADMM_PROB_LINEARPROGRAM;
    if (View::dense(K)) {
      d.D = nullptr;
      d.K = K->data;
      d.k0 = args.vec("k0");
      d.m = d.n = K->cols;
    } else {
      if (!haveD) return fail(ADMM_E_INVALID, needD);
      d.s = args.vec("s");
    }
    d.q = qp ? args.vec("q") : args.vec("b");
    if (qp) {
      const admm_field* P = args.get("P");
      if (View::dense(P)) d.P = P->data;
      d.r = handles.scalar("r", args.scalar("r", 0.0));
    }
    b.nA = b.nB = d.n;
  } else if (p == "basispursuit") {  // getProxOps.m:137-138
    // args = {P, q} as basispursuit.m:116-120 forms them; with args = {D, s} the engine forms both on the device
    const admm_field* P = args.get("P");
    d.problem = ADMM_PROB_BASISPURSUIT;
    if (View::dense(P)) {
      d.D = nullptr;
      d.P = P->data;
      d.m = d.n = P->cols;
      d.q = args.vec("q");
    } else {
      if (!haveD) return fail(ADMM_E_INVALID, "args.P (or args.D and args.s) must be full real matrices");
      d.s = args.vec("s");
    }
    b.nA = b.nB = d.n;
  } else if (p == "model") {  // getProxOps.m:83-89
    d.problem = ADMM_PROB_MODEL;
    d.D = nullptr;
    d.n = static_cast<int64_t>(args.scalar("n", 0.0));
    d.m = d.n;
    d.P = args.vec("PtP");
    d.q = args.vec("Ptr");
    d.Q = args.vec("QtQ");
    d.qz = args.vec("Qts");
    d.c = args.vec("c");
    b.nA = b.nB = d.n;
  } else if (p == "generic") {
    // results = admm(xminf, zming, options) with both handles the caller's (admm.m:24).  A = 1 runs as the model
    // problem without data; a constraint matrix A, or function handles A / At (handles.A, handles.At; admm.m:113-195),
    // as the LAD shape without a factor.  B: the shorthand -1, another scalar, an m x nB matrix (args.B) or a function
    // handle (handles.B with args.nB) -- admm.m:198-245.
    const admm_field* A = args.get("A");
    const bool a_fh = handles.handle("A");
    if (a_fh || (View::dense(A) && A->rows * A->cols > 1)) {
      d.problem = ADMM_PROB_LAD;
      d.xsolve = ADMM_XSOLVE_CALLBACK;
      if (a_fh) {
        if (!handles.handle("At"))
          return fail(ADMM_E_INVALID, "options.A is a function handle: options.At must be one too (admm.m:139-158)");
        d.D = nullptr;
        d.m = static_cast<int64_t>(args.scalar("m", 0.0));
        d.n = static_cast<int64_t>(args.scalar("nA", args.scalar("n", 0.0)));
        if (d.m <= 0 || d.n <= 0)
          return fail(ADMM_E_INVALID, "Matrix A is a function handle, but no number of columns nA (or rows m) specified "
                                      "for it");
        b.a_handle = true;
      } else {
        d.D = A->data;
        d.m = A->rows;
        d.n = A->cols;
        d.ldD = d.m;
      }
      d.s = args.vec("c");
      b.zeros.assign(static_cast<size_t>(d.m), 0.0);
      if (!d.s) d.s = b.zeros.data();  // c = 0
      b.nA = d.n;
      b.nB = d.m;
    } else {
      d.problem = ADMM_PROB_MODEL;
      d.D = nullptr;
      d.n = static_cast<int64_t>(args.scalar("n", args.scalar("nA", 0.0)));
      d.m = d.n;
      d.c = args.vec("c");
      b.nA = b.nB = d.n;
    }
    const admm_field* B = args.get("B");
    if (handles.handle("B")) {
      b.b_kind = 3;
      b.mC = b.nB;
      b.nB = static_cast<int64_t>(args.scalar("nB", 0.0));
      if (b.nB <= 0)
        return fail(ADMM_E_INVALID, "Matrix B is a function handle, but no number of columns nB specified for it; "
                                    "cannot infer nB - please specify it in options struct!");
    } else if (View::dense(B) && B->rows * B->cols > 1) {
      if (B->rows != b.nB)
        return fail(ADMM_E_INVALID, "Number of rows in matrix B do not match length of column vector c in constraint "
                                    "Ax + Bz = c");
      b.b_kind = 2;
      b.b_matrix = B->data;
      b.b_ld = B->rows;
      b.mC = b.nB;
      b.nB = B->cols;
    } else if (B && args.scalar("B", -1.0) != -1.0) {
      b.b_kind = 1;
      b.b_scalar = args.scalar("B", -1.0);
      b.mC = b.nB;
    }
  } else {
    return fail(ADMM_E_INVALID, "Invalid input for problem - given string is not a solver!");
  }
  return ADMM_OK;
}

}  // namespace

extern "C" {

int admm_binding_create(const char* problem, const admm_field* args, int32_t nargs, const admm_field* handles,
                        int32_t nhandles, admm_binding** out) {
  if (!problem || !out || nargs < 0 || nhandles < 0 || (nargs > 0 && !args) || (nhandles > 0 && !handles))
    return fail(ADMM_E_INVALID, "admm_binding_create: bad argument");
  *out = nullptr;
  admm_binding* b = new admm_binding();
  b->problem = lower(problem);
  const int rc = describe(b->problem, View{args, nargs}, View{handles, nhandles}, *b);
  if (rc != ADMM_OK) {
    delete b;
    return rc;
  }
  *out = b;
  return ADMM_OK;
}

void admm_binding_destroy(admm_binding* b) { delete b; }

const admm_problem_desc* admm_binding_desc(const admm_binding* b) { return b ? &b->d : nullptr; }

int admm_binding_get_info(const admm_binding* b, admm_binding_info* info) {
  if (!b || !info) return fail(ADMM_E_INVALID, "NULL argument");
  if (info->struct_size != static_cast<int32_t>(sizeof(admm_binding_info)))
    return fail(ADMM_E_INVALID, "admm_binding_info.struct_size mismatch (ABI version skew)");
  info->problem = b->d.problem;
  info->nA = b->nA;
  info->nB = b->nB;
  info->nU = b->mC ? b->mC : b->nB;  // u, c, A*x (admm.m:252-254: zeros(m, 1))
  info->a_handle = b->a_handle ? 1 : 0;
  info->b_kind = b->b_kind;
  info->b_scalar = b->b_scalar;
  info->b_matrix = b->b_matrix;
  info->b_ld = b->b_ld;
  return ADMM_OK;
}

// options.B other than -1 as a scalar or a matrix: forwarded to the engine the binding was made for
int admm_binding_apply(const admm_binding* b, admm_engine* e) {
  if (!b || !e) return fail(ADMM_E_INVALID, "NULL argument");
  if (b->b_kind == 1 || b->b_kind == 2)
    return admm_engine_set_constraint_b(e, b->b_matrix, b->b_ld, b->b_kind == 2 ? b->nB : 0, ADMM_MEM_HOST, b->b_scalar,
                                        nullptr, nullptr);
  return ADMM_OK;
}

int admm_binding_options(const admm_binding* b, const admm_field* options, int32_t nopt, const admm_field* handles,
                         int32_t nhandles, admm_options* o) {
  if (!b || !o || nopt < 0 || nhandles < 0 || (nopt > 0 && !options) || (nhandles > 0 && !handles))
    return fail(ADMM_E_INVALID, "admm_binding_options: bad argument");
  const View op{options, nopt}, h{handles, nhandles};
  admm_options_default(o);  // setopt defaults, admm.m:780-971
  o->rho = op.scalar("rho", o->rho);
  o->maxiters = static_cast<int32_t>(std::ceil(op.scalar("maxiters", o->maxiters)));  // admm.m:334-339
  o->domaxiters = static_cast<int32_t>(op.scalar("domaxiters", 0));
  o->relax = op.scalar("relax", 1.0);
  o->abstol = op.scalar("abstol", o->abstol);
  o->reltol = op.scalar("reltol", o->reltol);
  o->Hnormtol = op.scalar("Hreltol", op.scalar("Hnormtol", o->Hnormtol));  // quirk q2: either name
  o->convtol = op.scalar("convtol", o->convtol);
  o->restart = op.scalar("restart", o->restart);
  o->dvaltol = op.scalar("dvaltol", o->dvaltol);
  o->objevals = static_cast<int32_t>(op.scalar("objevals", 0));
  o->convtest = static_cast<int32_t>(op.scalar("convtest", 0));
  o->nodualerror = static_cast<int32_t>(op.scalar("nodualerror", 0));
  o->record_history = static_cast<int32_t>(op.scalar("recordhistory", 1));
  o->stale_factor_ok = static_cast<int32_t>(op.scalar("stalefactorok", 0));
  if (op.scalar("fast", 0) != 0) o->fast = op.text_is("fasttype", "strong") ? ADMM_FAST_STRONG : ADMM_FAST_WEAK;
  o->stopcond = op.text_is("stopcond", "hnorm")  ? ADMM_STOP_HNORM
                : op.text_is("stopcond", "both") ? ADMM_STOP_BOTH
                : (op.get("stopcond") && !op.text_is("stopcond", "standard")) ? ADMM_STOP_NONE
                                                                                : ADMM_STOP_STANDARD;
  // what admm refuses before its loop -- here before any engine exists, so that a bad options struct never leaves device
  // memory behind (at config 2's size an engine holds 8.4 GB)
  const size_t nA = static_cast<size_t>(b->nA), nB = static_cast<size_t>(b->nB);
  const size_t nU = static_cast<size_t>(b->mC ? b->mC : b->nB);
  size_t k = 0;
  if ((o->x0 = op.vec("x0", &k)) != nullptr && k != nA) return fail(ADMM_E_INVALID, "options.x0 has the wrong length");
  if ((o->z0 = op.vec("z0", &k)) != nullptr && k != nB) return fail(ADMM_E_INVALID, "options.z0 has the wrong length");
  if ((o->u0 = op.vec("u0", &k)) != nullptr && k != nU) return fail(ADMM_E_INVALID, "options.u0 has the wrong length");
  if (b->a_handle && (!h.handle("A") || !h.handle("At")))
    return fail(ADMM_E_INVALID, "this engine was created for function-handle operators: pass handles.A and handles.At");
  if (b->b_kind == 3 && !h.handle("B"))
    return fail(ADMM_E_INVALID, "this engine was created for a function-handle B: pass handles.B");
  for (const char* name : {"altu", "specialnorms"})  // the consensus hooks are descriptors, not handles (getproxops.m)
    if (h.handle(name) && b->d.problem == ADMM_PROB_LASSO_CONSENSUS)
      return fail(ADMM_E_UNSUPPORTED, "consensus lasso runs with the hooks of its own getproxops call");
  // admm.m:603: objevals without options.obj records nothing (objnative = 1: the solver's own objective on the device)
  if (o->objevals && h.scalar("objnative", 0.0) == 0.0 && !h.handle("obj")) o->objevals = 0;
  return ADMM_OK;
}

int admm_binding_results(admm_binding* b, const admm_options* o, const admm_run_summary* s, admm_result_field* out,
                         int32_t cap, int32_t* count) {
  if (!b || !o || !s || !out || !count || cap < 0) return fail(ADMM_E_INVALID, "admm_binding_results: bad argument");
  const int64_t nA = b->nA, nB = b->nB, nU = b->mC ? b->mC : b->nB, k = s->steps;
  const bool use_h = o->convtest || o->stopcond == ADMM_STOP_HNORM || o->stopcond == ADMM_STOP_BOTH;
  std::vector<admm_result_field> r;
  auto add = [&](const char* name, int32_t kind, int32_t source, int64_t rows, int64_t cols, double scalar = 0.0) {
    r.push_back(admm_result_field{name, kind, source, rows, cols, scalar});
  };
  add("x0", ADMM_RES_START, 0, nA, 1);  // admm.m:257-259
  add("z0", ADMM_RES_START, 1, nB, 1);
  add("u0", ADMM_RES_START, 2, nU, 1);
  if (o->fast == ADMM_FAST_WEAK) add("dvaltol", ADMM_RES_SCALAR, 0, 1, 1, o->dvaltol);  // admm.m:292
  if (use_h) add("Hnormtol", ADMM_RES_SCALAR, 0, 1, 1, o->Hnormtol);                     // admm.m:312
  if (o->record_history) {
    if (use_h) add("wvals", ADMM_RES_FETCH, ADMM_F_WVALS, nA + nB + nU, k);  // admm.m:678-681  w = [x; z; rho*u]
    add("xvals", ADMM_RES_FETCH, ADMM_F_XVALS, nA, k);
    add("zvals", ADMM_RES_FETCH, ADMM_F_ZVALS, nB, k);
    add("uvals", ADMM_RES_FETCH, ADMM_F_UVALS, nU, k);
    if (o->fast != ADMM_FAST_OFF) {
      add("vvals", ADMM_RES_FETCH, ADMM_F_VVALS, nB, k);
      add("uhatvals", ADMM_RES_FETCH, ADMM_F_UHATVALS, nU, k);
    }
  }
  if (o->fast != ADMM_FAST_WEAK) {  // q8: accelerated ADMM records no norms (admm.m:619-640)
    add("pnorm", ADMM_RES_FETCH, ADMM_F_PNORM, 1, k);
    add("dnorm", ADMM_RES_FETCH, ADMM_F_DNORM, 1, k);
    add("perr", ADMM_RES_FETCH, ADMM_F_PERR, 1, k);
    add("derr", ADMM_RES_FETCH, ADMM_F_DERR, 1, k);
  }
  if (o->objevals) add("objevals", ADMM_RES_FETCH, ADMM_F_OBJEVALS, 1, k);
  if (use_h) add("Hnormsq", ADMM_RES_FETCH, ADMM_F_HNORMSQ, 1, k);
  if (o->fast != ADMM_FAST_OFF) {
    add("avals", ADMM_RES_FETCH, ADMM_F_AVALS, 1, k);
    if (o->fast == ADMM_FAST_WEAK) {
      add("dvals", ADMM_RES_FETCH, ADMM_F_DVALS, 1, k);
      add("restarted", ADMM_RES_FETCH, ADMM_F_RESTARTED, 1, k);
    }
  }
  if (s->convtest_failed_at == 0) {  // q4: the reference returns early without these (admm.m:692-701)
    add("steps", ADMM_RES_SCALAR, 0, 1, 1, static_cast<double>(s->steps));
    add("xopt", ADMM_RES_FETCH, ADMM_F_XOPT, nA, 1);
    add("zopt", ADMM_RES_FETCH, ADMM_F_ZOPT, nB, 1);
    add("uopt", ADMM_RES_FETCH, ADMM_F_UOPT, nU, 1);
    if (b->d.problem == ADMM_PROB_LASSO_CONSENSUS) add("zconsensus", ADMM_RES_FETCH, ADMM_F_ZCONSENSUS, nA, 1);  // q9
    if (o->objevals) add("objopt", ADMM_RES_SCALAR, 0, 1, 1, s->objopt);
    add("runtime", ADMM_RES_SCALAR, 0, 1, 1, s->runtime_s);
  } else {
    add("convtestfailedat", ADMM_RES_SCALAR, 0, 1, 1, static_cast<double>(s->convtest_failed_at));
  }
  *count = static_cast<int32_t>(r.size());
  if (static_cast<size_t>(cap) < r.size()) return fail(ADMM_E_CAPACITY, "admm_binding_results: destination too small");
  std::memcpy(out, r.data(), r.size() * sizeof(admm_result_field));
  return ADMM_OK;
}

}  // extern "C"
