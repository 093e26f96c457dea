// dct.h -- direct x-update of 2-D total variation: (I + rho*D'D) is diagonalised by the 2-D DCT-II (D'D is the
// 5-point Laplacian with Neumann boundaries), so x = C' diag(1/(1 + rho*(lam_i + lam_j))) C b in five streaming
// passes instead of ~35 CG iterations.  Engine-side extension (the reference has no 2-D solver; tv2d.h).
#pragma once
#include "common.h"

namespace admm {

struct FinArgs;

struct DctTables {        // device tables of one transform length n (= 2^log2n, or any n with the chirp tables below)
  int32_t n, log2n;
  const admm_double2* tw;   // e^{-2 pi i k / n},     k < n/2        (FFT twiddles; chirp form: of length bm)
  const admm_double2* c4;   // e^{-i pi k / (2n)},    k <= n/2       (FFT -> DCT-II rotation; chirp form: every k < n)
  const double* lam;        // 4 sin^2(pi k / (2n)),  k < n          (eigenvalues of the 1-D Neumann Laplacian)
  // n not a power of two: the n-point DFT as a circular convolution of length bm = 2^log2bm >= 2n - 1 (Bluestein)
  int32_t bm, log2bm;          // 0: plain power-of-two network
  const admm_double2* chirp;   // c_j = e^{-pi i j^2 / n}, j < n
  const admm_double2* hbr;     // FFT_bm of h_m = conj(c_|m|) (wrapped), at the network's output positions, times 1/bm
  // the column kernels transform two real columns as one complex sequence; an odd width leaves one column without a
  // partner: pair `odd_pair` (set by the launchers, -1 otherwise) takes its single column twice
  int32_t odd_pair;
};

// host side of the tables (long double trigonometry); buffers have n/2, n/2 + 1 and n entries
void dct_fill_tables(int32_t n, admm_double2* tw, admm_double2* c4, double* lam);
// n is a power of two the LDS-resident transform supports
bool dct_length_ok(int64_t n);
// any other length the column transform supports through the chirp form (2n - 1 <= 8192), and its FFT length
bool dct_chirp_length_ok(int64_t n);
int32_t dct_chirp_fft_length(int64_t n);
// host side of the chirp tables: tw (bm/2), c4 (n: every k), lam (n), chirp (n), hbr (bm)
void dct_fill_chirp_tables(int32_t n, admm_double2* tw, admm_double2* c4, double* lam, admm_double2* chirp,
                           admm_double2* hbr);

// img (H x W, column-major, ld = H), in place: every column -> its DCT-II (unnormalised), two columns per workgroup
void launch_dct_cols_forward(double* img, int64_t H, int64_t W, const DctTables& th, const Ctrl* ctrl,
                             hipStream_t stream);
// ... with one extra workgroup that runs the finalize logic `f` of the previous iteration (when fin_pending)
void launch_dct_cols_forward_fin(double* img, int64_t H, int64_t W, const DctTables& th, const FinArgs& f,
                                 bool fin_pending, const Ctrl* ctrl, hipStream_t stream);
// the fused 2-D TV pass (tv2d.h: launch_tv2d_fused) whose right-hand side goes straight into the forward column
// transform: bhat (H x W) receives the DCT-II of every column of b = s + rho*D'(z+ - u+); a.W even, a.H = th.n
struct Tv2Args;
void launch_tv2d_fused_dct(const Tv2Args& a, bool state_in, double* bhat, const DctTables& th, const Ctrl* ctrl,
                           int* nblk_out, hipStream_t stream);
// inverse of the above (DCT-III with the 1/H factor): src -> dst (may alias)
void launch_dct_cols_inverse(const double* src, double* dst, int64_t H, int64_t W, const DctTables& th,
                             const Ctrl* ctrl, hipStream_t stream);
// t (W x H, column-major: the transposed, column-transformed image), in place: every column (one image row
// frequency i) -> DCT-II along W, divide by 1 + rho*(lamH[i] + lamW[k]), DCT-III back
void launch_dct_rows_solve(double* t, int64_t H, int64_t W, double rho, const DctTables& th, const DctTables& tw,
                           const Ctrl* ctrl, hipStream_t stream);
// the same on the untransposed image img (H x W, column-major), row pairs read and written at stride H: replaces
// transpose -> rows_solve -> transpose
void launch_dct_rows_solve_strided(double* img, int64_t H, int64_t W, double rho, const DctTables& th,
                                   const DctTables& tw, const Ctrl* ctrl, hipStream_t stream);
// The row stage without a transform: dst = inv((1 + rho*lamH[i]) I + rho*L_W) applied along every row i of src (H x W,
// column-major, the column-transformed image), as the truncated Toeplitz kernel on the mirrored row (dct.hip);
// src != dst.  tv2d_rows_green_taps(rho) = terms per side; usable while that is well below W.
int tv2d_rows_green_taps(double rho);
// fin != nullptr: one extra workgroup runs the finalize logic `*fin` of the previous iteration (when fin_pending)
void launch_tv2d_rows_green(const double* src, double* dst, int64_t H, int64_t W, double rho, const DctTables& th,
                            const Ctrl* ctrl, hipStream_t stream, const FinArgs* fin = nullptr, bool fin_pending = false);

// the same row stage as an exact Thomas elimination along the rows (any width, any rho): setup once per run and rho
// (cp, inv: H x W scratch images holding the elimination factors), then src -> dst per iteration (may alias)
void launch_tv2d_rows_thomas_setup(int64_t H, int64_t W, double rho, const DctTables& th, double* cp, double* inv,
                                   hipStream_t stream);
void launch_tv2d_rows_thomas(const double* src, double* dst, int64_t H, int64_t W, double rho, const double* cp,
                             const double* inv, const Ctrl* ctrl, hipStream_t stream);
// dst (cols x rows, column-major) = src (rows x cols, column-major) transposed
void launch_transpose(const double* src, double* dst, int64_t rows, int64_t cols, const Ctrl* ctrl,
                      hipStream_t stream);

}  // namespace admm
