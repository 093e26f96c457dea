// kernels.h -- host-side launchers of the HIP kernels (all enqueue on `stream`, never sync).
#pragma once
#include <vector>

#include "common.h"

namespace admm {

// ---------------------------------------------------------------- GEMV (gemv.hip)
// Plan for y = D*x with column-chunk partials: ypart is [nchunk][ldy].
struct GemvNPlan {
  int64_t m, n, ld;
  int64_t ldy;          // >= m, row stride between chunk partials
  int32_t nchunk;       // number of column chunks (partials to sum)
  int64_t cols_per_chunk;
  size_t part_elems() const { return static_cast<size_t>(nchunk) * ldy; }
};
GemvNPlan gemv_n_plan(int64_t m, int64_t n, int64_t ld);
// ypart[c][i] = sum_{j in chunk c} D[i,j]*x[j]; x may be given as the sum alpha*(xa - xb) + xc
void launch_gemv_n(const GemvNPlan& p, const double* D, const double* x, double* ypart, const Ctrl* ctrl,
                   hipStream_t stream);
// y[i] = sum_c ypart[c][i]  (stand-alone reduce, used outside the fused consumers)
void launch_sum_partials(const double* part, int32_t nchunk, int64_t ld, int64_t len, double* y, const Ctrl* ctrl,
                         hipStream_t stream);

// Plan for G = D'*[v0 v1 v2] with row-chunk partials: gpart is [nchunk][nrhs][ldg].
struct GemvTPlan {
  int64_t m, n, ld;
  int64_t ldg;           // >= n
  int32_t nchunk;        // row chunks
  int32_t rows_per_chunk;
  size_t part_elems(int nrhs) const { return static_cast<size_t>(nchunk) * nrhs * ldg; }
};
GemvTPlan gemv_t_plan(int64_t m, int64_t n, int64_t ld);
void launch_gemv_t(const GemvTPlan& p, const double* D, const double* v0, const double* v1, const double* v2,
                   int nrhs, double* gpart, const Ctrl* ctrl, hipStream_t stream);
// g[r][j] = sum_c gpart[c][r][j]; g has row stride ldg_out
void launch_sum_partials_t(const GemvTPlan& p, const double* gpart, int nrhs, double* g, int64_t ldg_out,
                           const Ctrl* ctrl, hipStream_t stream);

// y = M*x for symmetric M from its lower triangle only (half the bytes of a full GEMV).
// M must be stored npad x npad (npad = round_up(n,128), ld >= npad) with zeros outside n x n.
struct SymvPlan {
  int64_t n, npad;
  int64_t ldp;     // row stride of the partial arrays (= npad)
  int32_t ntile;   // 128-row wave chunks = 128-column groups: npart and tpart are [ntile][ldp]
  bool packed;     // M is tile-packed (launch_symv_pack): lower-triangle 128x128 tiles back to back, ld unused
  int64_t ncached; // tiles (linear index < ncached) read with default loads; the rest stream non-temporally
  size_t npart_elems() const { return static_cast<size_t>(ntile) * ldp; }
  size_t tpart_elems() const { return static_cast<size_t>(ntile) * ldp; }
};
SymvPlan symv_plan(int64_t n);
// Share of a re-read symmetric matrix that can stay in the 256 MB Infinity Cache next to the rest of an iteration's
// traffic (measured plateau 105 .. 290 MB at n = 10000, dev/symv_packed.hip)
constexpr int64_t kSymvCacheBytes = int64_t{168} << 20;
int64_t symv_tiles(const SymvPlan& p);          // lower-triangle tiles
size_t symv_packed_elems(const SymvPlan& p);    // doubles of the tile-packed storage
int64_t symv_cached_tiles(const SymvPlan& p, int64_t budget_bytes);
// K tile-packed matrices of one plan in one launch (partial rows left unsummed): Ms_dev = device array of K pointers
// fin != null: workgroup (0, 0) runs the finalize logic of the previous iteration with these arguments
struct FinArgs;
void launch_symv_lower_batch(const SymvPlan& p, const double* const* Ms_dev, int32_t K, const double* x0, int64_t xstride,
                             double* npart0, double* tpart0, int64_t pstride, const Ctrl* ctrl, hipStream_t stream,
                             const FinArgs* fin = nullptr);
// P (symv_packed_elems doubles) <- the lower-triangle tiles of the padded column-major M (npad x npad, ld)
void launch_symv_pack(const SymvPlan& p, const double* M, int64_t ld, double* P, hipStream_t stream);
// y = M*x for a small symmetric M (full storage, ld even, 16-byte aligned; x 16-byte aligned): one wave per column
void launch_symv_small(const double* M, int64_t n, int64_t ld, const double* x, double* y, const Ctrl* ctrl,
                       hipStream_t stream);
// part_rank / part_count: this launch processes every part_count-th lower-triangle tile (multi-GPU split of
// the x-solve); npart / tpart must have been zero-filled once, y is then this rank's PARTIAL result.
void launch_symv_lower(const SymvPlan& p, const double* M, int64_t ld, const double* x, double* npart, double* tpart,
                       double* y, const Ctrl* ctrl, hipStream_t stream, int part_rank = 0, int part_count = 1,
                       bool reduce = true);

// ---------------------------------------------------------------- dense setup (dense.hip)
// C = alpha*op(A)*op(B) + beta*C, column-major fp64 on the MFMA f64 path.  transA/transB: 0 = N, 1 = T.
// lower_only: compute only tiles that touch the lower triangle (SYRK-style symmetric results).
void launch_gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                 const double* B, int64_t ldb, double beta, double* C, int64_t ldc, bool lower_only,
                 hipStream_t stream);
// In-place blocked lower Cholesky; info (device int) gets the failing pivot (1-based) or 0.
// dinv (optional, [ceil(n/64)][64*64]) receives the inverted 64x64 diagonal blocks of the factor.
int cholesky_lower(double* A, int64_t n, int64_t lda, int32_t* info_dev, double* dinv, hipStream_t stream);
// invert the 64x64 diagonal blocks of an existing lower factor
void launch_trtri_diag(const double* L, int64_t n, int64_t ldl, double* dinv, hipStream_t stream);
// X = inv(L) (lower, upper part zero) from L and its inverted diagonal blocks
// clear = false: X (lower block and its upper-right scratch) is already zero
int trtri_lower_from_diag(const double* L, int64_t n, int64_t ldl, const double* dinv, double* X, int64_t ldx,
                          hipStream_t stream, bool clear = true);
// y = L (L' x), tmp: n doubles (setup-time probe of the x-solve forms)
void launch_llt_apply(const double* L, int64_t n, int64_t ld, const double* x, double* tmp, double* y,
                      hipStream_t stream);
// mirror the lower triangle into the upper one
void launch_symmetrize_lower(double* A, int64_t n, int64_t lda, hipStream_t stream);
void launch_add_diag(double* A, int64_t n, int64_t lda, double shift, hipStream_t stream);
void launch_fill(double* p, size_t n, double v, hipStream_t stream);

// ---------------------------------------------------------------- TRSV (trsv.hip)
// x = L' \ (L \ y) as blocked substitution over K coarse blocks whose diagonal-block inverses are folded
// into the panels (one bandwidth-bound launch per block and sweep).
struct TrsvPlan {
  int64_t n, npad;
  int64_t ldm, ldp;
  int32_t ntile, nblk, bt;   // 128-row tiles, coarse blocks, tiles per block
  double* Fm;                // forward panels  [inv(L_kk); -L_below,k inv(L_kk)]   (lower triangle, tile-packed)
  double* Um;                // backward panels [-L_k,above' inv(L_kk)'; inv(L_kk)'] (upper triangle, tile-packed)
  int64_t ncached;           // tiles of each triangle read with default loads (the rest stream non-temporally)
  double* P[2];              // [bt][npad] column-tile partials: the step in flight and the previous one
  double *v, *w;             // running right-hand sides (npad)
  bool streaming;            // non-temporal matrix loads (matrix larger than the caches)
  // one-launch form (trsv.hip: tri_persist_kernel): every step's partials and running right-hand side in buffers of
  // their own (written once per launch), per-(step, row tile) completion counters, the ordered work list
  bool persist;              // available (built) and not disabled by ADMM_TRSV_STEPS
  double* PP;                // [2*nblk][bt][npad] column-tile partials of every step
  double* BB;                // [2*nblk][npad] running right-hand side as every step leaves it
  int32_t* sync;             // [0] ticket, [1] done, [2] error, [3] pad, then counters [2*nblk][ntile]
  const void* items;         // device: TpItem[nitems]
  const int32_t* chunks;     // device: first item of every ticket, [nchunks + 1]
  int32_t nitems, nchunks, grid;
  // one-block form (symv.hip: tri1_*): the whole factor as ONE pre-inverted block, X = inv(L) tile-packed, applied as
  // w = X y (N-part pass + fold) and x = X' w (T-part pass): 3 launches per pair.  When built, Fm / Um / P / the one-launch
  // buffers above are absent (nblk = 1 and they would be the same matrix twice).
  bool one;
  double* X1;                // tile-packed lower triangle of inv(L)
  double *np1, *tp1;         // [ntile][ldp] partial rows of the forward / backward pass
  double* w1;                // [npad] forward result
};
// arguments of the one-block form's kernels
struct Tri1Args {
  const double* X;
  int64_t n;
  const double* y;           // forward input (n elements, 8-byte aligned)
  double* npart;
  double* tpart;
  double* w;
  int64_t ldp;
  uint32_t ncached;          // tiles with linear index < ncached are read with default loads (Infinity Cache share)
  uint32_t ntri;             // lower-triangle tiles
  int32_t ntile;
};
void launch_tri1_forward(const Tri1Args& a, const FinArgs* fin, bool fin_pending, const Ctrl* ctrl, hipStream_t stream);
void launch_tri1_backward(const Tri1Args& a, const Ctrl* ctrl, hipStream_t stream);
void launch_tri1_reduce(const Tri1Args& a, double* x, const Ctrl* ctrl, hipStream_t stream);
Tri1Args tri1_args(const TrsvPlan& p, const double* y);
// both passes; x == nullptr leaves the backward pass's partial rows to the consumer (prox_fin_kernel)
void launch_tri1_pair(const TrsvPlan& p, const double* y, double* x, const FinArgs* fin, bool fin_pending,
                      const Ctrl* ctrl, hipStream_t stream);
// doubles the plan needs in one caller-owned device buffer
// form 0: blocked substitution over K = ceil(n / 2048) coarse blocks; 1: the one-block form (n >= 256, else blocked).
// trsv_resolve_form: `form` unless ADMM_TRSV_FORM=blocked|one says otherwise (tests, A/B measurements) -- consulted by
// the callers that have a choice (engine.hip: choose_trsv_form; ops.hip), never by trsv_build itself.
constexpr int kTrsvBlocked = 0, kTrsvOne = 1;
int trsv_resolve_form(int64_t n, int form);
size_t trsv_plan_elems(int64_t n, int form = kTrsvBlocked);
// dinv64: the inverted 64x64 diagonal blocks of L (cholesky_lower / launch_trtri_diag)
int trsv_build(const double* L, int64_t n, int64_t ldl, const double* dinv64, double* buf, TrsvPlan* plan,
               hipStream_t stream, int form = kTrsvBlocked);
void launch_trsv_pair(const TrsvPlan& p, const double* y, double* x, const Ctrl* ctrl, hipStream_t stream);
// the one-launch form's error word (synchronises the stream): ADMM_OK, or ADMM_E_DEVICE after a poll gave up
int trsv_check_error(const TrsvPlan& p, hipStream_t stream);

// ---------------------------------------------------------------- symmetric eigen-decomposition (jacobi.hip)
// W (n x n symmetric PSD, FULL storage) is overwritten by W*V; V gets the eigenvectors, lam_host the eigenvalues
// (paired with V's columns, unsorted).  lam_dev: n doubles, rot: one int32 (device scratch).
int jacobi_eig_psd(double* W, int64_t n, int64_t ldw, double* V, int64_t ldv, double* lam_dev, int32_t* rot,
                   std::vector<double>* lam_host, int* sweeps_out, hipStream_t stream);
void launch_scale_cols(double* V, int64_t ldv, int64_t n, const double* scale, hipStream_t stream);

}  // namespace admm
