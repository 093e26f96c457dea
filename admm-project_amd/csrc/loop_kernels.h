// loop_kernels.h -- argument blocks of the per-iteration fused kernels (loop_kernels.hip).
#pragma once
#include "common.h"

namespace admm {

// z-prox flavours (getProxOps.m): the argument is always v = (Ax^ + u) - c
enum ProxKind : int32_t {
  PROX_SOFT = 0,   // sign(v).*max(abs(v)-t,0)                         933-938
  PROX_HUBER = 1,  // (rho*v + soft(v, 1+1/rho))/(1+rho)               1529-1539
  PROX_HINGE = 2,  // v + ell.*max(min(1-ell.*v, C/rho),0)             1084-1096
  PROX_01 = 3,     // ell.*minz01(ell.*v, rho/C)                       1100, 1158-1180
  PROX_BOX = 4,    // min(ub,max(lb,v))                                1470-1474
  PROX_POS = 6,    // max(v, 0)   (CVX pos)                            1378-1382, 1422-1426
  PROX_GIVEN = 5   // z was computed outside the fused kernel (a linear solve: zminModel 990-1013,
                   // or a caller-supplied zming callback) and is read from ProxArgs::zgiven
};

// what the NEXT x-update's right-hand side is, written by the prox/extrapolation kernel
enum RhsKind : int32_t {
  RHS_NONE = 0,
  RHS_RHO_DTS = 1,   // y = rho*(zx-ux) + Dts           lasso   getProxOps.m:1195
  RHS_RHO_MINUS_Q = 2,  // y = rho*(zx-ux) - q          QP      getProxOps.m:1455
  RHS_DIFF = 3,      // y = zx-ux                       basis pursuit 1031
  RHS_T1 = 4         // t1 = c + zx - ux (then D'*t1)   LAD/Huber/SVM 1514, 1067
};

// z-side objective terms accumulated by the prox kernel
enum ObjZKind : int32_t { OBJZ_NONE = 0, OBJZ_ABS = 1, OBJZ_HUBER = 2 };
// Ax-side / x-side terms
enum ObjXKind : int32_t { OBJX_NONE = 0, OBJX_HINGE = 1, OBJX_ZEROONE = 2, OBJX_ABS = 3,
                          OBJX_DOT = 4 /* sum ell_i * x_i: b'*x, linearprogram.m:178 */,
                          // lasso, A = I: x solves (G + rho*I) x = y with y = rho*(zx-ux) + D's still in a.rhs, so
                          // G x = y - rho*x and 1/2*x'Gx - x'D's = sum x_i*(1/2*(y_i - rho*x_i) - (D's)_i): the data
                          // term of lasso.m:227 (plus 1/2*s's) without touching D or G
                          OBJX_SOLVE = 5,
                          // bounded QP: (P + rho*I) x = y = rho*(zx-ux) - q, so 1/2*x'Px + q'x (quadraticprogram.m:242)
                          // = sum x_i*(1/2*(y_i - rho*x_i) + q_i)
                          OBJX_SOLVE_QP = 6 };

// reduction slots (per-block partials, summed in block order by the finalize kernel)
enum Slot : int32_t {
  S_R2 = 0,    // ||Ax + Bz - c||^2                    admm.m:621
  S_AX2 = 1,   // ||Ax||^2                             admm.m:649
  S_Z2 = 2,    // ||Bz||^2 = ||z||^2
  S_DZ2 = 3,   // ||z - zprev||^2                      admm.m:624 (A=I), 305
  S_U2 = 4,    // ||u||^2                              admm.m:654 (A=I)
  S_DU2 = 5,   // ||u - uprev||^2                      admm.m:306
  S_OBJZ = 6,
  S_OBJX = 7,
  S_DUH2 = 8,  // ||u - uhat||^2                       admm.m:572
  S_DZV2 = 9,  // ||z - v||^2                          admm.m:573
  S_G2 = 10,   // ||D'(z - zprev)||^2 via stencil (TV)   admm.m:624
  S_G3 = 11,   // ||D'u||^2 via stencil (TV)             admm.m:654
  S_OBJA = 12, // sum of |terms| of the S_OBJX sum (OBJX_SOLVE / _QP): the size of what cancels in it
  S_COUNT = 13
};

struct ProxArgs {
  int64_t len;             // nB = m (B = -I)
  const double* axsrc;     // Ax: [naxpart][axld] partials (naxpart >= 1)
  const double* ax_t;      // one-launch tail after the lower-triangle x-solve: its T-part partial rows (axsrc = the
                           // N-part rows, naxpart = tiles, axld = row stride); null otherwise
  int32_t naxpart;
  int32_t ax_tri;          // after the one-block triangular solves (symv.hip: tri1_*): x_i = sum of the rows p >= i / 128
                           // of ax_t alone (naxpart = tiles; axsrc unused)
  int64_t axld;
  double* x_out;           // A = I: x_i = sum of partials is stored here (may alias axsrc when naxpart == 1)
  const double* c;         // nullable (c = 0)
  const double* ell;
  const double* lb;
  const double* ub;
  const double* zgiven;    // PROX_GIVEN: the new z
  double* z;
  double* u;
  double* uhat;            // fast only
  double* v;               // fast only
  double* zprev;           // fast only (alg 2 needs it after the prox)
  double* uprev;           // fast only
  double* dz;              // A = D: z - zprev for the dual residual
  double* rhs;             // y / t1 for the next x-update (len elements)
  const double* rhs_add;   // Dts or q
  double* zhist;           // [len][maxiters] or null
  double* uhist;
  double* xhist;           // A = I only ([len][maxiters]) or null
  double* vhist;           // fast only
  double* uhathist;        // fast only
  double* part;            // [S_COUNT][kMaxPartBlocks]
  double rho, relax, t;    // t: soft threshold | C/rho (hinge) | rho/C (01)
  double rho_solve;        // OBJX_SOLVE: the shift of the factor the x-update solved with (a stale factor under adaptive
                           // rho keeps its own: getProxOps.m:1190-1200 caches L, U once)
  int32_t prox;
  int32_t rhs_kind;
  int32_t alg;             // 0 plain, 1 fast (strong), 2 accelerated (weak)
  int32_t objz, objx;
  int32_t a_identity;
};

struct ExtrapArgs {  // alg 2 second phase (admm.m:576-591) -> v, uhat and the next rhs
  int64_t len;
  const double* z;
  const double* u;
  const double* zprev;
  const double* uprev;
  const double* c;
  double* v;
  double* uhat;
  double* rhs;
  const double* rhs_add;
  double* vhist;
  double* uhathist;
  double rho;
  int32_t rhs_kind;
};

struct FinArgs {
  int64_t len;             // nB
  int64_t nA;
  const double* part;      // [S_COUNT][kMaxPartBlocks]
  int32_t nblk;            // partial blocks used by the prox kernel
  const double* g;         // A = D: [3][ldg] = D'*[t1, dz, u]; null when A = I
  int64_t ldg;
  const double* x;         // A = D: current x (history copy, ||x||^2)
  double* xhist;
  const double* objpart;   // extra objective partials (e.g. ||Dx-s||^2 blocks) or null
  int32_t nobjpart;
  double obj_scale_part;   // multiplies sum(objpart)
  double obj_scale_z;      // multiplies S_OBJZ
  double obj_scale_x;      // multiplies S_OBJX
  double obj_half_xnorm;   // adds this * ||x||^2
  double obj_const;
  double cnorm;            // ||c||
  double rho, rhoH;
  double abstol, reltol, Hnormtol, convtol, restart, dvaltol;
  int32_t alg, a_identity, nodualerror, objevals, use_h, convtest, stopcond, domaxiters, maxiters;
  int32_t dual_from_slots;  // dual norms come from S_G2/S_G3 (stencil operators) instead of g
  int32_t specialnorms;     // consensus lasso: pnorm/dnorm are lassonorms' squared sums (q10)
  int32_t nslices_total;    // slicenum over all ranks
  const double* cons_q;     // sharded consensus lasso: sum over ALL slices of ||x_k - xave_prev||^2 (consensus.hip);
                            // lassonorms' first value is *cons_q - N*S_G2 instead of the rank-local S_R2
  // row-sharded runs: sums already reduced over blocks AND ranks (all-reduced), else null
  const double* slots_reduced;  // [16]
  const double* objp_reduced;   // [1]
  int64_t len_global;           // numel(Ax) over all ranks (admm.m:644-645); 0 = use len
  double* pnorm;
  double* dnorm;
  double* perr;
  double* derr;
  double* objv;
  double* hnorm;
  double* avals;
  double* dvals;
  double* restarted;
  Ctrl* ctrl;
  const double* norms_given;    // options.specialnorms as the caller's handle: pnorm, dnorm (admm.m:612-616), else null
  int32_t obj_track_bound;      // the objective is the right-hand-side form (OBJX_SOLVE): track eps*|terms|/|objective|
};

// First half of a split z-update (PROX_GIVEN): what the z-prox is called with (admm.m:515-530).
//   xh  = Axhat = relax*Ax - (1-relax)*(B zprev - c)   (= Ax when relax == 1)
//   rz  = add + rho*((xh + uo) - c)                    right-hand side of zminModel (getProxOps.m:1012)
struct PreZArgs {
  int64_t len;
  const double* axsrc;  // Ax partials, as ProxArgs
  int32_t naxpart;
  int64_t axld;
  const double* c;
  const double* z;      // z_prev
  const double* uo;     // u (alg 0) or uhat
  const double* add;    // nullable
  double* xh;
  double* rz;           // nullable
  double rho, relax;
};
void launch_prez(const PreZArgs& a, const Ctrl* ctrl, hipStream_t stream);
// caller-supplied options.altu (admm.m:553-559), plain ADMM: bz = -z (what the handle gets as Bz) ...
void launch_negate(const double* z, double* bz, int64_t len, const Ctrl* ctrl, hipStream_t stream);
// ... and, with the handle's result: u = unew, its history column, the u-dependent partial sums (||u||^2, ||u - u_old||^2)
// in the prox kernel's block layout, and the next x-update's right-hand side from (z, unew)
struct UFixArgs {
  int64_t len;
  const double* unew;
  const double* uold;
  const double* z;
  const double* c;        // nullable
  const double* rhs_add;  // nullable
  double* u;
  double* uhist;          // nullable
  double* rhs;            // nullable
  double* part;           // [S_COUNT][kMaxPartBlocks]
  int32_t nblk;           // blocks the prox kernel wrote
  int32_t rhs_kind;
  double rho;
};
void launch_ufix(const UFixArgs& a, const Ctrl* ctrl, hipStream_t stream);
void launch_prox(const ProxArgs& a, const Ctrl* ctrl, int* nblk_out, hipStream_t stream);
// A = I, alg 0 / 1: z/u update AND the finalize logic in one launch (the last workgroup to arrive finalizes);
// a.len <= 128 * kMaxPartBlocks
void launch_prox_fin(const ProxArgs& a, const FinArgs& f, Ctrl* ctrl, int* nblk_out, hipStream_t stream,
                     bool defer = false);
// the finalize arguments launch_prox_fin hands to the last workgroup (or, deferred, to the next launch)
FinArgs prox_fin_args(const ProxArgs& a, const FinArgs& f);
void launch_fast_decide(const FinArgs& a, hipStream_t stream);   // alg 2: d, restart decision, alpha
void launch_extrapolate(const ExtrapArgs& a, const Ctrl* ctrl, hipStream_t stream);
void launch_finalize(const FinArgs& a, hipStream_t stream);
// out[slot] = sum over blocks of part[slot][.] (slot < 16): the payload of the per-iteration all-reduce
void launch_pack_slots(const double* part, int32_t nblk, double* out16, const Ctrl* ctrl, hipStream_t stream);
// out[0] = sum of v[0..n)
void launch_pack_sum(const double* v, int32_t n, double* out1, const Ctrl* ctrl, hipStream_t stream);
// rhs for the very first x-update from (zx, ux): same formulas as the fused epilogue
void launch_initial_rhs(int64_t len, int rhs_kind, double rho, const double* zx, const double* ux, const double* c,
                        const double* rhs_add, double* rhs, hipStream_t stream);
// sum of squares of (sum_c part[c][i] - s[i]) per block -> objpart[block]; returns blocks used
void launch_residual_sq(const double* part, int32_t nchunk, int64_t ld, const double* s, int64_t len, double* objpart,
                        int* nblk_out, const Ctrl* ctrl, hipStream_t stream);
// objpart[block] = sum_i x_i*(0.5*(sum_c part[c][i]) + q_i)   (1/2 x'Px + q'x, quadraticprogram.m:242)
void launch_qp_objective(const double* part, int32_t nchunk, int64_t ld, const double* x, const double* q,
                         int64_t len, double* objpart, int* nblk_out, const Ctrl* ctrl, hipStream_t stream);
// The packed lower-triangle x-solve (symv.hip) carrying the finalize logic of the previous iteration in workgroup 0;
// the partial rows stay unsummed (prox_fin_kernel takes them)
// part_rank / part_count: the tile split over the ranks (launch_symv_lower); y != null: also sum the partial rows into y
void launch_symv_lower_fin(const struct SymvPlan& p, const double* M, const double* x, double* npart, double* tpart,
                           const FinArgs& f, bool fin_pending, const Ctrl* ctrl, hipStream_t stream, int part_rank = 0,
                           int part_count = 1, double* y = nullptr);

// g[r][j] = sum_c gpart[c][r][j] (gemv.hip) with the deferred finalize logic in one extra workgroup
void launch_sum_partials_t_fin(const struct GemvTPlan& p, const double* gpart, int nrhs, double* g, int64_t ldg_out,
                               const FinArgs& f, const Ctrl* ctrl, hipStream_t stream);

// Two-launch iteration of unwrapped ADMM with an explicit pseudo-inverse (unwrapped.hip)
struct UwArgs {
  const double* D;   // m x n, column-major
  int64_t ldD;
  const double* Dp;  // n x m, column-major: pinv(D) (linearsvm.m:185)
  int64_t ldP;
  int64_t m, n;
  int32_t R;         // rows per workgroup of uw_prox_kernel (multiple of 64)
  int32_t nblk;      // its workgroups = partial rows of x
  double* G;         // [2][nblk][ldg] partial rows of Dp*(c + z - u), double-buffered on the iteration parity
  int64_t ldg;
  double* axpart;    // [nchunk][ldax] partial D*x per 64-column chunk
  int64_t ldax;
  int32_t nchunk;    // <= 16
  double* xbuf;      // [2][ldx] the x of the iteration (stored by uw_ax_kernel), double-buffered on its parity
  int64_t ldx;
  int64_t iter;      // 0-based iteration this launch belongs to (the host's count)
  int32_t fin_pending;  // uw_ax_kernel: run the finalize logic of iteration iter - 1 in its extra workgroup
  int32_t init;      // uw_prox_kernel: 1 = only the partial rows of Dp*(c + z0 - u0), before the first iteration
};
int uw_rows_per_block(int64_t m);
int uw_chunks(int64_t n);
bool uw_supported(int64_t m, int64_t n);
FinArgs uw_fin_args(const UwArgs& a, const FinArgs& f);
void launch_uw_ax(const UwArgs& a, const FinArgs& f, Ctrl* ctrl, hipStream_t stream);
void launch_uw_prox(const UwArgs& a, const ProxArgs& pa, const Ctrl* ctrl, hipStream_t stream);

// One pass over a tall, NARROW D per A = D iteration without a dual residual (unwrapped.hip: ad_onepass_kernel)
struct OnePassArgs {
  const double* D;   // m x n, column-major
  int64_t ldD, m, n;
  const double* x;   // the iteration's x (n)
  double* gpart;     // [workgroups][ldg] partial rows of D'*(c + z - u): the next x-update's right-hand side
  int64_t ldg;
};
bool onepass_supported(int64_t m, int64_t n);
int onepass_workgroups(int64_t m);
void launch_ad_onepass(const OnePassArgs& a, const ProxArgs& pa, const Ctrl* ctrl, int* nblk_out, hipStream_t stream);

// State of the caller's z when options.B is general (the loop's z buffers hold w = -B*z): after zming returned znew,
// zprev <- z, z <- znew, zvals(:, i) = znew and -- fast ADMM -- v = z + coef*(z - zprev) (admm.m:568, 579) or the
// restart value zprev (admm.m:586).  phase 0: state + history (+ v for alg 1, coefficient from ctrl->acurr as the fused
// kernel takes it); phase 1: v of accelerated ADMM after launch_fast_decide published ctrl->coef / restart_flag.
struct ZStateArgs {
  int64_t len;
  double *z, *zprev, *v;
  const double* znew;
  double *zhist, *vhist;
  int32_t alg, phase;
};
void launch_zstate(const ZStateArgs& a, const Ctrl* ctrl, hipStream_t stream);
// disc[0] = max(disc[0], |a - b| / |a|) for a = sa*sum(pa[0..na)) + ca, b = sb*sum(pb[0..nb)) + cb (one workgroup, fixed
// order): the calibration of the Gram-form lasso objective against the literal one
void launch_obj_compare(const double* pa, int na, double sa, double ca, const double* pb, int nb, double sb, double cb,
                        double* disc, const Ctrl* ctrl, hipStream_t stream);
// Start of a run from zero iterates: x, z, u, v, uhat, the nine scalar histories and the control block in one launch
struct RunInitArgs {
  double *x, *z, *u, *v, *uhat;
  int64_t nA, len;
  double* scal[9];
  int32_t N;
  Ctrl* ctrl;
  Ctrl c0;
};
void launch_run_init(const RunInitArgs& a, hipStream_t stream);
// x = alpha*(sum_c part[c][i]) + beta*y[i] + add[i]  (add/y nullable)
void launch_combine(const double* part, int32_t nchunk, int64_t ld, double alpha, const double* y, double beta,
                    const double* add, double* x, int64_t len, const Ctrl* ctrl, hipStream_t stream);

}  // namespace admm
