// mex_runtime.cpp -- executable stand-in for the MEX / C Matrix API (see mex.h next to this file) plus a small C API
// ("mxh_*") through which the Python test-suite builds argument structs, registers function handles (ctypes callbacks),
// calls the gateway's mexFunction and reads the results.  mexErrMsgIdAndTxt throws; mxh_call catches and reports the
// identifier and message.  Test fixture only.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mex.h"

struct mxArray_tag {
  mxClassID cls = mxDOUBLE_CLASS;
  size_t m = 0, n = 0;
  std::vector<double> real;                 // double data (also the nonzeros of a sparse matrix)
  std::vector<uint64_t> u64;                // uint64 data
  std::vector<unsigned char> logical;
  std::string chars;
  std::vector<std::string> names;           // struct fields
  std::vector<mxArray*> values;
  bool sparse = false;
  std::vector<mwIndex> ir, jc;
  // function handle: a registered host callback
  typedef mxArray* (*callback_t)(void* user, int nrhs, mxArray** rhs);
  callback_t fn = nullptr;
  void* user = nullptr;
};

namespace {
struct MexError : std::runtime_error {
  std::string id;
  MexError(const std::string& i, const std::string& m) : std::runtime_error(m), id(i) {}
};
std::string g_err_id, g_err_msg;
void (*g_exit_fn)(void) = nullptr;
int g_lock = 0;
}  // namespace

extern "C" {

double* mxGetPr(const mxArray* pa) { return const_cast<double*>(pa->real.data()); }
void* mxGetData(const mxArray* pa) {
  if (pa->cls == mxUINT64_CLASS) return const_cast<uint64_t*>(pa->u64.data());
  if (pa->cls == mxLOGICAL_CLASS) return const_cast<unsigned char*>(pa->logical.data());
  return const_cast<double*>(pa->real.data());
}
double mxGetScalar(const mxArray* pa) {
  if (pa->cls == mxLOGICAL_CLASS) return pa->logical.empty() ? 0.0 : pa->logical[0];
  if (pa->cls == mxUINT64_CLASS) return pa->u64.empty() ? 0.0 : static_cast<double>(pa->u64[0]);
  return pa->real.empty() ? 0.0 : pa->real[0];
}
size_t mxGetM(const mxArray* pa) { return pa->m; }
size_t mxGetN(const mxArray* pa) { return pa->n; }
size_t mxGetNumberOfElements(const mxArray* pa) { return pa->m * pa->n; }
mxClassID mxGetClassID(const mxArray* pa) { return pa->cls; }
bool mxIsStruct(const mxArray* pa) { return pa && pa->cls == mxSTRUCT_CLASS; }
bool mxIsSparse(const mxArray* pa) { return pa && pa->sparse; }
bool mxIsDouble(const mxArray* pa) { return pa && pa->cls == mxDOUBLE_CLASS; }
bool mxIsNumeric(const mxArray* pa) { return pa && (pa->cls == mxDOUBLE_CLASS || pa->cls == mxUINT64_CLASS); }
bool mxIsLogical(const mxArray* pa) { return pa && pa->cls == mxLOGICAL_CLASS; }
bool mxIsComplex(const mxArray*) { return false; }
bool mxIsChar(const mxArray* pa) { return pa && pa->cls == mxCHAR_CLASS; }
bool mxIsClass(const mxArray* pa, const char* name) {
  if (!pa) return false;
  if (!std::strcmp(name, "function_handle")) return pa->cls == mxFUNCTION_CLASS;
  if (!std::strcmp(name, "double")) return pa->cls == mxDOUBLE_CLASS;
  if (!std::strcmp(name, "struct")) return pa->cls == mxSTRUCT_CLASS;
  if (!std::strcmp(name, "char")) return pa->cls == mxCHAR_CLASS;
  return false;
}
mwIndex* mxGetIr(const mxArray* pa) { return const_cast<mwIndex*>(pa->ir.data()); }
mwIndex* mxGetJc(const mxArray* pa) { return const_cast<mwIndex*>(pa->jc.data()); }

mxArray* mxGetField(const mxArray* pa, mwIndex, const char* fieldname) {
  if (!pa || pa->cls != mxSTRUCT_CLASS) return nullptr;
  for (size_t k = 0; k < pa->names.size(); ++k)
    if (pa->names[k] == fieldname) return pa->values[k];
  return nullptr;
}
int mxAddField(mxArray* pa, const char* fieldname) {
  for (size_t k = 0; k < pa->names.size(); ++k)
    if (pa->names[k] == fieldname) return static_cast<int>(k);
  pa->names.push_back(fieldname);
  pa->values.push_back(nullptr);
  return static_cast<int>(pa->names.size()) - 1;
}
void mxSetField(mxArray* pa, mwIndex, const char* fieldname, mxArray* value) {
  const int k = mxAddField(pa, fieldname);
  if (pa->values[k] && pa->values[k] != value) mxDestroyArray(pa->values[k]);
  pa->values[k] = value;
}
int mxGetNumberOfFields(const mxArray* pa) { return static_cast<int>(pa->names.size()); }
const char* mxGetFieldNameByNumber(const mxArray* pa, int n) { return pa->names[n].c_str(); }
char* mxArrayToString(const mxArray* pa) {
  if (!pa || pa->cls != mxCHAR_CLASS) return nullptr;
  char* c = static_cast<char*>(std::malloc(pa->chars.size() + 1));
  std::memcpy(c, pa->chars.c_str(), pa->chars.size() + 1);
  return c;
}
void mxFree(void* ptr) { std::free(ptr); }
void mxDestroyArray(mxArray* pa) {
  if (!pa) return;
  for (mxArray* v : pa->values) mxDestroyArray(v);
  delete pa;
}
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames) {
  mxArray* a = new mxArray_tag();
  a->cls = mxSTRUCT_CLASS;
  a->m = m;
  a->n = n;
  for (int k = 0; k < nfields; ++k) mxAddField(a, fieldnames[k]);
  return a;
}
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity) {
  mxArray* a = new mxArray_tag();
  a->cls = classid;
  a->m = m;
  a->n = n;
  if (classid == mxUINT64_CLASS) a->u64.assign(m * n, 0);
  else a->real.assign(m * n, 0.0);
  return a;
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity f) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, f); }
mxArray* mxCreateDoubleScalar(double value) {
  mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL);
  a->real[0] = value;
  return a;
}
mxArray* mxCreateLogicalScalar(bool value) {
  mxArray* a = new mxArray_tag();
  a->cls = mxLOGICAL_CLASS;
  a->m = a->n = 1;
  a->logical.assign(1, value ? 1 : 0);
  return a;
}
mxArray* mxCreateString(const char* str) {
  mxArray* a = new mxArray_tag();
  a->cls = mxCHAR_CLASS;
  a->chars = str;
  a->m = 1;
  a->n = a->chars.size();
  return a;
}
mxArray* mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity) {
  mxArray* a = new mxArray_tag();
  a->cls = mxDOUBLE_CLASS;
  a->sparse = true;
  a->m = m;
  a->n = n;
  a->real.assign(nzmax, 0.0);
  a->ir.assign(nzmax, 0);
  a->jc.assign(n + 1, 0);
  return a;
}

void mexErrMsgIdAndTxt(const char* identifier, const char* err_msg, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, err_msg);
  std::vsnprintf(buf, sizeof buf, err_msg, ap);
  va_end(ap);
  throw MexError(identifier ? identifier : "", buf);
}
int mexCallMATLAB(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* functionName) {
  // only feval(handle, args...) exists here: the handle is a registered host callback
  if (std::strcmp(functionName, "feval") != 0 || nrhs < 1 || !prhs[0] || prhs[0]->cls != mxFUNCTION_CLASS) return 1;
  mxArray* out = prhs[0]->fn(prhs[0]->user, nrhs - 1, prhs + 1);
  if (!out) return 1;
  if (nlhs >= 1) plhs[0] = out;
  else mxDestroyArray(out);
  return 0;
}
mxArray* mexCallMATLABWithTrap(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* functionName) {
  if (mexCallMATLAB(nlhs, plhs, nrhs, prhs, functionName) == 0) return nullptr;
  return mxCreateDoubleScalar(-1.0);  // stands for the MException object
}
void mexLock(void) { ++g_lock; }
void mexUnlock(void) { --g_lock; }
int mexAtExit(void (*exit_fcn)(void)) {
  g_exit_fn = exit_fcn;
  return 0;
}

// ---- harness API used by tests/test_mex_gateway.py (ctypes) ---------------------------------------------------
mxArray* mxh_struct(void) { return mxCreateStructMatrix(1, 1, 0, nullptr); }
mxArray* mxh_matrix(const double* data, size_t m, size_t n) {
  mxArray* a = mxCreateDoubleMatrix(m, n, mxREAL);
  if (data && m * n > 0) std::memcpy(a->real.data(), data, m * n * sizeof(double));
  return a;
}
// dense column-major lower-triangular matrix -> sparse CSC (what lasso.m:175 `sparse(L)` hands to getproxops)
mxArray* mxh_sparse_from_dense(const double* data, size_t m, size_t n) {
  size_t nnz = 0;
  for (size_t k = 0; k < m * n; ++k) nnz += data[k] != 0.0;
  mxArray* a = mxCreateSparse(m, n, nnz, mxREAL);
  size_t p = 0;
  for (size_t j = 0; j < n; ++j) {
    a->jc[j] = p;
    for (size_t i = 0; i < m; ++i)
      if (data[i + j * m] != 0.0) {
        a->ir[p] = i;
        a->real[p] = data[i + j * m];
        ++p;
      }
  }
  a->jc[n] = p;
  return a;
}
mxArray* mxh_string(const char* s) { return mxCreateString(s); }
mxArray* mxh_scalar(double v) { return mxCreateDoubleScalar(v); }
mxArray* mxh_function(mxArray_tag::callback_t fn, void* user) {
  mxArray* a = new mxArray_tag();
  a->cls = mxFUNCTION_CLASS;
  a->m = a->n = 1;
  a->fn = fn;
  a->user = user;
  return a;
}
void mxh_set(mxArray* s, const char* name, mxArray* v) { mxSetField(s, 0, name, v); }
mxArray* mxh_get(const mxArray* s, const char* name) { return mxGetField(s, 0, name); }
int mxh_nfields(const mxArray* s) { return mxIsStruct(s) ? mxGetNumberOfFields(s) : 0; }
const char* mxh_fieldname(const mxArray* s, int k) { return mxGetFieldNameByNumber(s, k); }
size_t mxh_rows(const mxArray* a) { return a->m; }
size_t mxh_cols(const mxArray* a) { return a->n; }
int mxh_class(const mxArray* a) { return static_cast<int>(a->cls); }
const double* mxh_data(const mxArray* a) { return a->real.data(); }
double mxh_value(const mxArray* a) { return mxGetScalar(a); }
void mxh_free(mxArray* a) { mxDestroyArray(a); }
const char* mxh_last_error_id(void) { return g_err_id.c_str(); }
const char* mxh_last_error_msg(void) { return g_err_msg.c_str(); }
// calls the gateway; returns 0 and *out (may be NULL) or 1 with the error recorded
int mxh_call(int nrhs, const mxArray** prhs, mxArray** out) {
  mxArray* plhs[1] = {nullptr};
  g_err_id.clear();
  g_err_msg.clear();
  try {
    mexFunction(1, plhs, nrhs, prhs);
  } catch (const MexError& e) {
    g_err_id = e.id;
    g_err_msg = e.what();
    return 1;
  }
  if (out) *out = plhs[0];
  return 0;
}
void mxh_shutdown(void) {
  if (g_exit_fn) g_exit_fn();
  g_exit_fn = nullptr;
}
int mxh_lock_count(void) { return g_lock; }

}  // extern "C"
