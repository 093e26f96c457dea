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
#include <deque>
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
bool is_handle(const mxArray* f) { return f && mxIsClass(f, "function_handle"); }

void check(int rc) {
  if (rc != ADMM_OK) mexErrMsgIdAndTxt("admm:engine", "%s", admm_last_error());
}

// ---- a MATLAB struct as the admm_field array the binding layer reads (admm_engine.h: admm_binding_*) -------------------
// Which field means what is decided behind the C ABI; here a struct is flattened, nothing more.
struct Fields {
  std::vector<admm_field> f;
  std::deque<std::string> text;   // character vectors (deque: stable addresses)
  std::deque<double> scalars;     // logical / integer scalars as doubles
  void add(const mxArray* s) {
    if (!s || !mxIsStruct(s)) return;
    const int n = mxGetNumberOfFields(s);
    for (int i = 0; i < n; ++i) {
      const char* name = mxGetFieldNameByNumber(s, i);
      const mxArray* v = mxGetField(s, 0, name);
      if (!v) continue;
      admm_field e{};
      e.name = name;
      e.kind = ADMM_FIELD_OTHER;
      e.rows = static_cast<int64_t>(mxGetM(v));
      e.cols = static_cast<int64_t>(mxGetN(v));
      if (is_handle(v)) {
        e.kind = ADMM_FIELD_HANDLE;
      } else if (mxIsChar(v)) {
        text.push_back(to_string(v));
        e.kind = ADMM_FIELD_TEXT;
        e.text = text.back().c_str();
      } else if (mxIsDouble(v) && !mxIsComplex(v) && mxIsSparse(v)) {
        static_assert(sizeof(mwIndex) == sizeof(uint64_t), "CSC indices travel as 64-bit integers");
        e.kind = ADMM_FIELD_SPARSE;
        e.data = mxGetPr(v);
        e.ir = reinterpret_cast<const uint64_t*>(mxGetIr(v));
        e.jc = reinterpret_cast<const uint64_t*>(mxGetJc(v));
      } else if (is_dense_double(v)) {
        e.kind = ADMM_FIELD_NUMERIC;
        e.data = mxGetPr(v);
      } else if ((mxIsNumeric(v) || mxIsLogical(v)) && mxGetNumberOfElements(v) == 1) {
        scalars.push_back(mxGetScalar(v));
        e.kind = ADMM_FIELD_NUMERIC;
        e.data = &scalars.back();
        e.rows = e.cols = 1;
      }
      f.push_back(e);
    }
  }
  const admm_field* data() const { return f.empty() ? nullptr : f.data(); }
  int32_t size() const { return static_cast<int32_t>(f.size()); }
};

// a description and what the gateway has to remember next to it
struct Desc {
  admm_binding* b = nullptr;
  admm_binding_info info{};
  ~Desc() { admm_binding_destroy(b); }
};

// MATLAB error with the library's message; the caller has nothing on the device yet
void describe(const std::string& p, const mxArray* args, const mxArray* handles, Desc& ds) {
  Fields fa, fh;
  fa.add(args);
  fh.add(handles);
  if (admm_binding_create(p.c_str(), fa.data(), fa.size(), fh.data(), fh.size(), &ds.b) != ADMM_OK)
    mexErrMsgIdAndTxt(std::strstr(admm_last_error(), "is not a solver") ? "admm:problem" : "admm:arg", "%s", admm_last_error());
  ds.info.struct_size = static_cast<int32_t>(sizeof(ds.info));
  check(admm_binding_get_info(ds.b, &ds.info));
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

// out[0 .. nout) = feval(handle, in_0, ..., in_(nin-1) [, scalar]): device vectors staged down, the result staged up.
// exact: the result must be a full real vector of exactly nout values (else: of at least nout; nout = 1: any numeric
// scalar).  `what` names the handle in the failure text.
struct StagedIn {
  const double* dev;
  size_t n;
};
int staged_call(Thunk* t, const StagedIn* in, int nin, const double* scalar, double* out, size_t nout, bool exact,
                const char* what, void* stream) {
  mxArray* rhs[6] = {t->fh, nullptr, nullptr, nullptr, nullptr, nullptr};
  int nrhs = 1, rc = 0;
  for (int k = 0; k < nin; ++k, ++nrhs) {
    rhs[nrhs] = staged_vector(in[k].dev, in[k].n, stream);
    if (!rhs[nrhs]) rc = 1;
  }
  if (scalar) rhs[nrhs++] = mxCreateDoubleScalar(*scalar);
  mxArray* lhs[1] = {nullptr};
  if (rc == 0) rc = call_handle(t, 1, lhs, nrhs, rhs);
  if (rc == 0) {
    const size_t got = lhs[0] ? mxGetNumberOfElements(lhs[0]) : 0;
    const bool scalar_ok = nout == 1 && got == 1 && (mxIsNumeric(lhs[0]) || mxIsLogical(lhs[0]));
    if (!scalar_ok && (!is_dense_double(lhs[0]) || (exact ? got != nout : got < nout))) {
      t->failure = std::string(what) + " returned something that is not a full real vector of the expected length";
      rc = 1;
    } else if (scalar_ok) {
      const double v = mxGetScalar(lhs[0]);
      rc = admm_memcpy_h2d(out, &v, sizeof(double), stream);
    } else {
      rc = admm_memcpy_h2d(out, mxGetPr(lhs[0]), nout * sizeof(double), stream);
    }
  }
  for (int k = 1; k < nrhs; ++k)
    if (rhs[k]) mxDestroyArray(rhs[k]);
  if (lhs[0]) mxDestroyArray(lhs[0]);
  return rc;
}

int prox_thunk(void* user, const double* x, const double* z, const double* u, double rho, double* out, int64_t nout,
               void* stream) {
  Thunk* t = static_cast<Thunk*>(user);
  const StagedIn in[3] = {{x, t->nfirst}, {z, t->nB}, {u, t->nU}};
  return staged_call(t, in, 3, &rho, out, static_cast<size_t>(nout), true, "a proximal-operator handle", stream);
}

// options.A / At / B as MATLAB function handles of a single vector (admm.m:117-158, 206-216)
int op_thunk(void* user, const double* v, int64_t nin, double* out, int64_t nout, void* stream) {
  const StagedIn in[1] = {{v, static_cast<size_t>(nin)}};
  return staged_call(static_cast<Thunk*>(user), in, 1, nullptr, out, static_cast<size_t>(nout), true,
                     "a constraint-operator handle (A, At or B)", stream);
}

// options.altu(u, Ax, Bz, c) and options.specialnorms(x, z, u, rho) as MATLAB handles (admm.m:553-559, 612-616)
int altu_thunk(void* user, const double* u, const double* ax, const double* bz, const double* c, int64_t m, double* out,
               void* stream) {
  const size_t n = static_cast<size_t>(m);
  const StagedIn in[4] = {{u, n}, {ax, n}, {bz, n}, {c, n}};
  return staged_call(static_cast<Thunk*>(user), in, 4, nullptr, out, n, true, "options.altu", stream);
}

int norms_thunk(void* user, const double* x, int64_t nA, const double* z, int64_t nB, const double* u, int64_t m,
                double rho, double* out2, void* stream) {
  const StagedIn in[3] = {{x, static_cast<size_t>(nA)}, {z, static_cast<size_t>(nB)}, {u, static_cast<size_t>(m)}};
  return staged_call(static_cast<Thunk*>(user), in, 3, &rho, out2, 2, false, "options.specialnorms (two values, admm.m:613-616)",
                     stream);
}

int obj_thunk(void* user, const double* x, int64_t nA, const double* z, int64_t nB, double* out, void* stream) {
  const StagedIn in[2] = {{x, static_cast<size_t>(nA)}, {z, static_cast<size_t>(nB)}};
  return staged_call(static_cast<Thunk*>(user), in, 2, nullptr, out, 1, true, "options.obj", stream);
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

// options -> admm_options and every refusal that needs no engine (admm_binding_options); a MATLAB error on failure.
// `keep` owns what the options may point into besides MATLAB's own arrays (a start vector given as an integer scalar)
// and must live as long as `o` is used.
void read_options(const Desc& ds, const mxArray* op, const mxArray* handles, Fields& keep, admm_options& o) {
  if (!mxIsStruct(op)) mexErrMsgIdAndTxt("admm:arg", "Given options is not a struct! At least pass empty struct!");
  Fields fh;
  keep.add(op);
  fh.add(handles);
  const int rc = admm_binding_options(ds.b, keep.data(), keep.size(), fh.data(), fh.size(), &o);
  if (rc != ADMM_OK) mexErrMsgIdAndTxt(rc == ADMM_E_UNSUPPORTED ? "admm:unsupported" : "admm:arg", "%s", admm_last_error());
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
// Never raises a MATLAB error itself: nullptr + err on failure (read_options has vetted the arguments).
mxArray* run_engine(admm_engine* e, const Desc& ds, admm_options o, const mxArray* handles, RunError& err) {
  const size_t nA = static_cast<size_t>(ds.info.nA), nB = static_cast<size_t>(ds.info.nB),
               nU = static_cast<size_t>(ds.info.nU);
  auto engine_ok = [&](int rc) { return rc == ADMM_OK || err.set("admm:engine", admm_last_error()); };

  // caller-supplied handles replace the engine-native operators (admm.m:502, 521-530, 603-605)
  Thunk tx, tz, tobj, ta, tat, tb, taltu, tnorms;
  if (ds.info.a_handle) {  // the thunks live as long as this run; a persistent engine gets fresh ones every run
    ta.fh = const_cast<mxArray*>(field(handles, "A"));
    tat.fh = const_cast<mxArray*>(field(handles, "At"));
    if (!engine_ok(admm_engine_set_operators(e, op_thunk, &ta, op_thunk, &tat))) return nullptr;
  }
  if (ds.info.b_kind == 3) {
    tb.fh = const_cast<mxArray*>(field(handles, "B"));
    if (!engine_ok(admm_engine_set_constraint_b(e, nullptr, 0, ds.info.nB, ADMM_MEM_HOST, 0.0, op_thunk, &tb))) return nullptr;
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
  const mxArray* on = field(handles, "objnative");
  const bool obj_native = on && mxGetNumberOfElements(on) == 1 && mxGetScalar(on) != 0.0;
  if (is_handle(fx)) {
    tx.fh = const_cast<mxArray*>(fx);
    tx.nfirst = nA;
    tx.nB = nB;
    tx.nU = nU;
  }
  if (is_handle(fz)) {
    tz.fh = const_cast<mxArray*>(fz);
    tz.nfirst = o.relax != 1.0 ? nU : nA;  // zming(x, ...) or zming(Axhat, ...) (admm.m:521-530)
    tz.nB = nB;
    tz.nU = nU;
  }
  if (o.objevals && !obj_native && is_handle(fo)) tobj.fh = const_cast<mxArray*>(fo);  // (none: the binding cleared objevals)
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

  // results: which fields, from where and of what shape is the binding layer's answer (admm.m:257-767)
  admm_result_field rf[48];
  int32_t nrf = 0;
  if (!engine_ok(admm_binding_results(ds.b, &o, &s, rf, 48, &nrf))) return nullptr;
  mxArray* res = mxCreateStructMatrix(1, 1, 0, nullptr);
  for (int32_t i = 0; i < nrf; ++i) {
    const admm_result_field& r = rf[i];
    if (r.kind == ADMM_RES_FETCH) {
      put(res, r.name, fetch_matrix(e, r.source, static_cast<size_t>(r.rows), static_cast<size_t>(r.cols)));
    } else if (r.kind == ADMM_RES_SCALAR) {
      put(res, r.name, mxCreateDoubleScalar(r.scalar));
    } else {  // ADMM_RES_START: x0 / z0 / u0 as the run took them (admm.m:252-259)
      const double* v = r.source == 0 ? o.x0 : (r.source == 1 ? o.z0 : o.u0);
      mxArray* a = mxCreateDoubleMatrix(static_cast<size_t>(r.rows), 1, mxREAL);
      if (v) std::memcpy(mxGetPr(a), v, static_cast<size_t>(r.rows) * sizeof(double));
      put(res, r.name, a);
    }
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
    // A Desc on the heap, owned by g_solve: a MATLAB error below (describe / read_options: no engine yet) long-jumps
    // past every C++ destructor of this frame, and the next call (or mexAtExit) frees what it left
    delete g_solve;
    g_solve = new Desc();
    Desc& ds = *g_solve;
    describe(to_string(prhs[1]), prhs[2], handles, ds);
    admm_options o;
    static Fields opt_fields;  // (static: a MATLAB error below long-jumps past destructors; cleared on every call)
    opt_fields = Fields();
    read_options(ds, prhs[3], handles, opt_fields, o);  // (every refusal that needs no engine: nothing is on the device yet)
    static bool registered = false;
    if (!registered) {
      mexAtExit(at_exit_all);
      registered = true;
    }
    admm_engine* e = nullptr;
    check(admm_engine_create(admm_binding_desc(ds.b), &e));  // (a failed create leaves nothing behind)
    g_live.push_back(e);  // (only an allocation failure inside MATLAB's own mx* calls can still leave it to mexAtExit)
    RunError err;
    mxArray* res = nullptr;
    if (admm_binding_apply(ds.b, e) != ADMM_OK) err.set("admm:engine", admm_last_error());
    if (err.id.empty()) res = run_engine(e, ds, o, handles, err);
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
    int rc = admm_engine_create(admm_binding_desc(ds->b), &e);
    if (rc == ADMM_OK && (rc = admm_binding_apply(ds->b, e)) != ADMM_OK) admm_engine_destroy(e);
    if (rc != ADMM_OK) {
      delete ds;
      check(rc);
    }
    // the engine copied every array at create: the borrowed pointers of the description must not be used again
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
    admm_options o;
    static Fields opt_fields;
    opt_fields = Fields();
    read_options(*l->ds, prhs[2], hd, opt_fields, o);
    RunError err;
    mxArray* res = run_engine(l->e, *l->ds, o, hd, err);  // (the engine persists by design: 'destroy' frees it)
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
