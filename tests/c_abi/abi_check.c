/* abi_check.c -- the drop-in boundary used from plain C (no Python, no C++): compiles against
 * include/admm_engine.h with a C99 compiler, links libadmm_hip.so, and runs a 4 x 2 lasso
 * (solvers/lasso.m semantics) through create / run / fetch / destroy.
 *   exit 0 and "OK steps=..."        on a machine with a HIP device
 *   exit 0 and "NO_DEVICE rc=-3 ..." where there is none (the engine must refuse loudly, never fall back)
 *   exit 1                           on any other outcome
 */
#include <stdio.h>
#include <string.h>

#include "admm_engine.h"

int main(void) {
  /* column-major 4 x 2 */
  const double D[8] = {0.5, -0.5, 0.5, 0.5, 0.1, 0.7, -0.7, 0.1};
  const double s[4] = {1.0, -0.2, 0.3, 0.8};
  admm_problem_desc d;
  admm_options o;
  admm_run_summary sum;
  admm_engine* eng = NULL;
  double x[2] = {0.0, 0.0}, pn[64];
  size_t n = 0;
  int rc, ndev = 0;

  if (admm_abi_version() != ADMM_ABI_VERSION) {
    printf("ABI version mismatch: header %d, library %d\n", ADMM_ABI_VERSION, admm_abi_version());
    return 1;
  }
  admm_problem_desc_default(&d);
  if (d.struct_size != (int)sizeof(admm_problem_desc)) return 1;
  d.problem = ADMM_PROB_LASSO;
  d.m = 4;
  d.n = 2;
  d.D = D;
  d.ldD = 4;
  d.s = s;
  d.lambda = 0.05;
  d.rho = 1.0;
  d.xsolve = ADMM_XSOLVE_TRSV;
  rc = admm_engine_create(&d, &eng);
  (void)admm_device_count(&ndev);
  if (rc != ADMM_OK) {
    if (rc == ADMM_E_DEVICE && ndev <= 0 && eng == NULL) {
      printf("NO_DEVICE rc=%d msg=%s\n", rc, admm_last_error());
      return 0;
    }
    printf("create failed rc=%d msg=%s\n", rc, admm_last_error());
    return 1;
  }
  admm_options_default(&o);
  o.maxiters = 50;
  o.objevals = 1;
  rc = admm_engine_run(eng, &o, &sum);
  if (rc != ADMM_OK) {
    printf("run failed rc=%d msg=%s\n", rc, admm_last_error());
    return 1;
  }
  if (admm_engine_fetch(eng, ADMM_F_XOPT, x, 2, &n) != ADMM_OK || n != 2) return 1;
  if (admm_engine_fetch(eng, ADMM_F_PNORM, pn, 64, &n) != ADMM_OK || (int)n != sum.steps) return 1;
  /* a buffer that is too small must be refused, not overrun */
  if (admm_engine_fetch(eng, ADMM_F_XOPT, x, 1, &n) != ADMM_E_CAPACITY) return 1;
  printf("OK steps=%d x=[%.12g %.12g] objopt=%.12g pnorm_last=%.3e\n", sum.steps, x[0], x[1], sum.objopt,
         pn[sum.steps - 1]);
  admm_engine_destroy(eng);
  return 0;
}
