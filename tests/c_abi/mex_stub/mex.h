/* Stand-in for MATLAB's mex.h / matrix.h: the entry points admm_mex.cpp uses, with the signatures documented in
 * MathWorks' "C Matrix API" / "MEX API" reference.  The real header comes with MATLAB (extern/include/mex.h), which
 * does not exist in the build image.  mex_runtime.cpp next to this file IMPLEMENTS them (malloc-backed mxArrays,
 * function handles as registered C callbacks) so that the test-suite can execute the gateway, not only compile it.
 * Test fixture only: nothing here is shipped or linked into libadmm_hip.so. */
#ifndef ADMM_TEST_MEX_STUB_H
#define ADMM_TEST_MEX_STUB_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum {
  mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS,
  mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS,
  mxINT64_CLASS, mxUINT64_CLASS, mxFUNCTION_CLASS
} mxClassID;

double* mxGetPr(const mxArray* pa);
void* mxGetData(const mxArray* pa);
double mxGetScalar(const mxArray* pa);
size_t mxGetM(const mxArray* pa);
size_t mxGetN(const mxArray* pa);
size_t mxGetNumberOfElements(const mxArray* pa);
mxClassID mxGetClassID(const mxArray* pa);
bool mxIsStruct(const mxArray* pa);
bool mxIsSparse(const mxArray* pa);
bool mxIsDouble(const mxArray* pa);
bool mxIsNumeric(const mxArray* pa);
bool mxIsLogical(const mxArray* pa);
bool mxIsComplex(const mxArray* pa);
bool mxIsChar(const mxArray* pa);
bool mxIsClass(const mxArray* pa, const char* name);
mwIndex* mxGetIr(const mxArray* pa);
mwIndex* mxGetJc(const mxArray* pa);
mxArray* mxGetField(const mxArray* pa, mwIndex i, const char* fieldname);
void mxSetField(mxArray* pa, mwIndex i, const char* fieldname, mxArray* value);
int mxAddField(mxArray* pa, const char* fieldname);
int mxGetNumberOfFields(const mxArray* pa);
const char* mxGetFieldNameByNumber(const mxArray* pa, int n);
char* mxArrayToString(const mxArray* pa);
void mxFree(void* ptr);
void mxDestroyArray(mxArray* pa);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double value);
mxArray* mxCreateLogicalScalar(bool value);
mxArray* mxCreateString(const char* str);
mxArray* mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity flag);

void mexErrMsgIdAndTxt(const char* identifier, const char* err_msg, ...);
int mexCallMATLAB(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* functionName);
/* as mexCallMATLAB, but an error inside the callee comes back as an MException object (NULL = success) */
mxArray* mexCallMATLABWithTrap(int nlhs, mxArray* plhs[], int nrhs, mxArray* prhs[], const char* functionName);
void mexLock(void);
void mexUnlock(void);
int mexAtExit(void (*exit_fcn)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#ifdef __cplusplus
}
#endif
#endif
