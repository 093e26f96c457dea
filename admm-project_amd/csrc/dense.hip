// dense.hip -- one-time dense setup on the fp64 matrix cores (gfx950):
//   gemm      C = alpha*op(A)*op(B) + beta*C      v_mfma_f64_16x16x4_f64, LDS-tiled 128x128
//   cholesky  blocked right-looking chol(.,'lower') (lasso.m:168, lad.m:134, getProxOps.m:434)
//   trtri     inverse of the lower factor, recursive-doubling over batched GEMMs
// These replace MATLAB's `D'*D`, `chol` and the implicit factor inverse used by `\`.
// MFMA is used here and only here (north_star: "MFMA used only for the one-time AtA build").
#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace admm {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int BMP = BM + 16;  // row stride (doubles) of the k-major LDS tiles: 144 = 16 mod 32
constexpr int BNP = BN + 16;  //   -> the two k-rows a half-wave touches land on disjoint banks

// Batched/strided GEMM descriptor.  All matrices column-major.  Reads outside
// [0,rowsX) x [0,colsX) of a matrix return 0, stores outside C's extent are dropped, so
// callers may issue full-size tiles on ragged edges.
struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc;
  int64_t M, N, K;           // logical problem per batch entry
  int64_t a_rows, a_cols;    // valid extent of the stored A (before op), relative to A + batch offset
  int64_t b_rows, b_cols;
  int64_t c_rows, c_cols;
  int64_t strideA, strideB, strideC;  // batch strides (elements)
  int64_t shrinkA_r, shrinkA_c, shrinkB_r, shrinkB_c, shrinkC_r, shrinkC_c;  // extent lost per batch index
  double alpha, beta;
  int lower_only;            // skip tiles strictly above the diagonal (square C)
};

// One 128 x 16 operand tile, element (t, k) of op(X):
//   KMAJOR storage (contiguous along k): X[k + t*ld]  -- op(A) with transA, op(B) without transB
//   TMAJOR storage (contiguous along t): X[t + k*ld]
// Each thread moves 4 pairs (16 bytes each): global -> registers (tile_load, issued one k-step ahead of the
// MFMAs that consume it) and registers -> LDS (tile_store), LDS layout S[k*BTP + t].
// FAST: the whole tile lies inside the matrix and pairs are 16-byte aligned -> no guards at all.
template <bool KMAJOR, bool FAST>
__device__ __forceinline__ void tile_load(const double* __restrict__ X, int64_t ld, int64_t t0, int64_t k0,
                                          int64_t tvalid, int64_t kvalid, int tid, double2_t (&r)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (KMAJOR) {
      const int kp = (tid & 7) * 2, tt = (tid >> 3) + s * 32;
      const int64_t gt = t0 + tt, gk = k0 + kp;
      const double* p = X + gk + gt * ld;
      if (FAST) {
        r[s] = *reinterpret_cast<const double2_t*>(p);
      } else {
        double2_t v{0.0, 0.0};
        if (gt < tvalid) {
          if (gk < kvalid) v.x = p[0];
          if (gk + 1 < kvalid) v.y = p[1];
        }
        r[s] = v;
      }
    } else {
      const int tp = (tid & 63) * 2, kk = (tid >> 6) + s * 4;
      const int64_t gt = t0 + tp, gk = k0 + kk;
      const double* p = X + gt + gk * ld;
      if (FAST) {
        r[s] = *reinterpret_cast<const double2_t*>(p);
      } else {
        double2_t v{0.0, 0.0};
        if (gk < kvalid) {
          if (gt < tvalid) v.x = p[0];
          if (gt + 1 < tvalid) v.y = p[1];
        }
        r[s] = v;
      }
    }
  }
}

// LDS image of a tile: element (k, t) at S[k*BMP + (t ^ swz(k))], swz(k) = ((k >> 1) & 3) * 8.
// BMP = 16 mod 32 makes the MFMA operand reads (4 k-rows x 16 consecutive t per wave) exactly 2-way, which
// is the floor for 512 bytes; the XOR spreads the TRANSPOSED stores of the k-major case (8 lanes hold 8
// different even k of one t) over all banks instead of one (8-way conflict without it) and only permutes
// 8-aligned groups, so the read pattern keeps its bank set.
__device__ __forceinline__ int swz(int k) { return ((k >> 1) & 3) << 3; }

template <bool KMAJOR>
__device__ __forceinline__ void tile_store(double* __restrict__ S, int tid, const double2_t (&r)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (KMAJOR) {
      const int kp = (tid & 7) * 2, tt = (tid >> 3) + s * 32;
      S[kp * BMP + (tt ^ swz(kp))] = r[s].x;        // swz(kp) == swz(kp + 1): kp is even
      S[(kp + 1) * BMP + (tt ^ swz(kp))] = r[s].y;
    } else {
      const int tp = (tid & 63) * 2, kk = (tid >> 6) + s * 4;
      *reinterpret_cast<double2_t*>(&S[kk * BMP + (tp ^ swz(kk))]) = r[s];
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_f64_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double As[2][BK * BMP];  // double-buffered: one barrier per k-step
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * BNP];
  static_assert(BMP == BNP, "tile_store assumes one row stride");
  const int bz = blockIdx.z;
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * BM;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * BN;
  if (g.lower_only && i0 + BM - 1 < j0) return;
  const double* __restrict__ A = g.A + bz * g.strideA;
  const double* __restrict__ B = g.B + bz * g.strideB;
  double* __restrict__ C = g.C + bz * g.strideC;
  const int64_t a_rows = g.a_rows - bz * g.shrinkA_r, a_cols = g.a_cols - bz * g.shrinkA_c;
  const int64_t b_rows = g.b_rows - bz * g.shrinkB_r, b_cols = g.b_cols - bz * g.shrinkB_c;
  const int64_t c_rows = g.c_rows - bz * g.shrinkC_r, c_cols = g.c_cols - bz * g.shrinkC_c;
  // valid extents of op(A) (M x K) and op(B) (K x N) in (tile, k) coordinates
  const int64_t a_t = TA ? (g.M < a_cols ? g.M : a_cols) : (g.M < a_rows ? g.M : a_rows);
  const int64_t a_k = TA ? (g.K < a_rows ? g.K : a_rows) : (g.K < a_cols ? g.K : a_cols);
  const int64_t b_t = TB ? (g.N < b_rows ? g.N : b_rows) : (g.N < b_cols ? g.N : b_cols);
  const int64_t b_k = TB ? (g.K < b_cols ? g.K : b_cols) : (g.K < b_rows ? g.K : b_rows);
  const bool a_in = (i0 + BM <= a_t) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((g.lda & 1) == 0);
  const bool b_in = (j0 + BN <= b_t) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ((g.ldb & 1) == 0);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wi = (wid & 1) * 64, wj = (wid >> 1) * 64;  // wave's 64x64 sub-tile
  const int l15 = lane & 15, lq = lane >> 4;

  double4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};

  // Pipeline per k-step: the MFMAs of tile k run from LDS buffer `cur` while (a) tile k+1 moves from registers
  // into the other LDS buffer and (b) the global loads of tile k+2 are in flight -- both issued between the
  // MFMA groups, so the matrix pipe (64 cycles per f64 MFMA) hides them; one barrier per k-step.  The loop
  // body is branch-free (loads past the end are clamped onto the last tile and never consumed): with control
  // flow inside it the register allocator shuttles the 128 accumulator registers between AGPRs and VGPRs on
  // every k-step -- and it still does so for the loop-carried values unless the MFMAs take their C/D operand
  // in VGPRs (this file is built with -mllvm -amdgpu-mfma-vgpr-form, see the Makefile: at one wave per SIMD
  // the 512-entry unified register file has room for the accumulators and everything else).  FAST = interior tile, aligned, K a multiple of BK: no guards on the loads.
  auto mainloop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    double2_t ra[4], rb[4];
    const int64_t klast = (g.K > BK) ? ((g.K - 1) / BK) * BK : 0;  // first k of the last tile
    auto load = [&](int64_t k0) {
      const int64_t kc = k0 < klast ? k0 : klast;
      tile_load<TA, FAST>(A, g.lda, i0, kc, a_t, a_k, tid, ra);
      tile_load<!TB, FAST>(B, g.ldb, j0, kc, b_t, b_k, tid, rb);
    };
    load(0);
    tile_store<TA>(As[0], tid, ra);
    tile_store<!TB>(Bs[0], tid, rb);
    __syncthreads();
    load(BK);
    int cur = 0;
#pragma unroll 1
    for (int64_t k0 = 0; k0 < g.K; k0 += BK, cur ^= 1) {
      const double* __restrict__ Ac = As[cur];
      const double* __restrict__ Bc = Bs[cur];
      // operand fragments are read one sub-step ahead of the MFMAs that use them
      double af[2][4], bf[2][4];
      auto read_frag = [&](int slot, int kk) {
        const int krow = kk + lq, sw = swz(krow);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          af[slot][t] = Ac[krow * BMP + ((wi + t * 16 + l15) ^ sw)];
          bf[slot][t] = Bc[krow * BNP + ((wj + t * 16 + l15) ^ sw)];
        }
      };
      read_frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 4) {
        const int slot = (kk >> 2) & 1;
        if (kk + 4 < BK) read_frag(slot ^ 1, kk + 4);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[slot][a], bf[slot][b], acc[a][b], 0, 0, 0);
        if (kk == 0) {  // tile k+1: registers -> the other buffer (nobody reads it during this step)
          tile_store<TA>(As[cur ^ 1], tid, ra);
          tile_store<!TB>(Bs[cur ^ 1], tid, rb);
        }
        if (kk == 4) load(k0 + 2 * BK);
      }
      __syncthreads();
    }
  };
  if (a_in && b_in && (g.K % BK) == 0 && g.K <= a_k && g.K <= b_k) mainloop(std::true_type{});
  else mainloop(std::false_type{});
  // ---- epilogue.  f64 16x16 C/D map: col = lane&15, row = (lane>>4) + 4*reg.
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t gj = j0 + wj + b * 16 + l15;
      if (gj >= g.N || gj >= c_cols) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gi = i0 + wi + a * 16 + lq + 4 * r;
        if (gi >= g.M || gi >= c_rows) continue;
        if (g.lower_only && gi < gj) continue;
        double* cp = C + gi + gj * g.ldc;
        double v = g.alpha * acc[a][b][r];
        if (g.beta != 0.0) v += g.beta * (*cp);
        *cp = v;
      }
    }
  }
}

static void launch_gemm_args(int transA, int transB, const GemmArgs& g, int batch, hipStream_t stream) {
  dim3 grid(static_cast<unsigned>(ceil_div(g.M, BM)), static_cast<unsigned>(ceil_div(g.N, BN)),
            static_cast<unsigned>(batch));
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return;
  if (!transA && !transB)
    hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, dim3(kBlock), 0, stream, g);
  else if (transA && !transB)
    hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, dim3(kBlock), 0, stream, g);
  else if (!transA && transB)
    hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, dim3(kBlock), 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, dim3(kBlock), 0, stream, g);
}

void launch_gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                 const double* B, int64_t ldb, double beta, double* C, int64_t ldc, bool lower_only,
                 hipStream_t stream) {
  GemmArgs g{};
  g.A = A;
  g.B = B;
  g.C = C;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.M = M;
  g.N = N;
  g.K = K;
  g.a_rows = transA ? K : M;
  g.a_cols = transA ? M : K;
  g.b_rows = transB ? N : K;
  g.b_cols = transB ? K : N;
  g.c_rows = M;
  g.c_cols = N;
  g.alpha = alpha;
  g.beta = beta;
  g.lower_only = lower_only ? 1 : 0;
  launch_gemm_args(transA, transB, g, 1, stream);
}

// ------------------------------------------------------------------------------------
// Cholesky: NB = 64 right-looking.  Per block column k:
//   potrf_diag : one workgroup factors A_kk in LDS and also writes inv(L_kk)
//   trsm_panel : A[k+1:, k] <- A[k+1:, k] * inv(L_kk)'      (64-row slabs, in place)
//   gemm NT    : A[k+1:, k+1:] -= P * P'  (lower tiles only)
// ------------------------------------------------------------------------------------
constexpr int NB = 64;
constexpr int NBP = NB + 1;

// Factor the nb x nb (nb <= 64) block at A (ld) in place (lower), write its inverse (lower,
// ld = NB, zero upper) to `inv`.  info: first non-positive pivot (global 1-based index) or unchanged.
__global__ __launch_bounds__(kBlock) void potrf_diag_kernel(double* __restrict__ A, int64_t ld, int nb,
                                                            double* __restrict__ inv, int32_t* info,
                                                            int32_t pivot_base) {
  __shared__ double S[NB * NBP];   // S[j*NBP + i] = A(i,j)
  __shared__ double X[NB * NBP];   // inverse, X[j*NBP + i]
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int idx = tid; idx < NB * NB; idx += kBlock) {
    const int i = idx % NB, j = idx / NB;
    S[j * NBP + i] = (i < nb && j < nb && i >= j) ? A[i + j * ld] : ((i == j) ? 1.0 : 0.0);
    X[j * NBP + i] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int k = 0; k < nb; ++k) {
    const double akk = S[k * NBP + k];
    if (!(akk > 0.0)) {
      if (tid == 0) {
        bad = 1;
        if (*info == 0) *info = pivot_base + k + 1;
      }
    }
    __syncthreads();
    if (bad) break;
    const double d = sqrt(akk);
    __syncthreads();
    if (tid < NB) {
      if (tid == k) S[k * NBP + k] = d;
      else if (tid > k) S[k * NBP + tid] /= d;
    }
    __syncthreads();
    // trailing update of the lower triangle: A(i,j) -= L(i,k)*L(j,k), k < j <= i
    for (int idx = tid; idx < NB * NB; idx += kBlock) {
      const int i = idx % NB, j = idx / NB;
      if (j > k && i >= j) S[j * NBP + i] -= S[k * NBP + i] * S[k * NBP + j];
    }
    __syncthreads();
  }
  if (bad) return;
  // inverse by forward substitution, one thread per column of the identity
  if (tid < NB) {
    const int j = tid;
    for (int i = j; i < NB; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = j; k < i; ++k) s -= S[k * NBP + i] * X[j * NBP + k];
      X[j * NBP + i] = s / S[i * NBP + i];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < NB * NB; idx += kBlock) {
    const int i = idx % NB, j = idx / NB;
    if (i < nb && j < nb && i >= j) A[i + j * ld] = S[j * NBP + i];
    inv[i + j * NB] = (i >= j && i < nb && j < nb) ? X[j * NBP + i] : 0.0;
  }
}

// P[rows, 0:nb] <- P[rows, 0:nb] * inv', inv is nb x nb lower (ld NB).  One 64-row slab per block.
__global__ __launch_bounds__(kBlock) void trsm_panel_kernel(double* __restrict__ P, int64_t ld, int64_t rows, int nb,
                                                            const double* __restrict__ inv) {
  __shared__ double Sp[NB * NBP];  // Sp[c*NBP + r] = P(r, c)
  __shared__ double Si[NB * NBP];  // Si[j*NBP + i] = inv(i, j)
  const int tid = threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * NB;
  for (int idx = tid; idx < NB * NB; idx += kBlock) {
    const int r = idx % NB, c = idx / NB;
    Sp[c * NBP + r] = (r0 + r < rows && c < nb) ? P[r0 + r + c * ld] : 0.0;
    Si[c * NBP + r] = inv[r + c * NB];
  }
  __syncthreads();
  // out(r, j) = sum_{c<=j} P(r,c) * inv(j,c)
  double out[NB * NB / kBlock];
#pragma unroll
  for (int s = 0; s < NB * NB / kBlock; ++s) {
    const int idx = tid + s * kBlock;
    const int r = idx % NB, j = idx / NB;
    double acc = 0.0;
    for (int c = 0; c <= j; ++c) acc = __builtin_fma(Sp[c * NBP + r], Si[c * NBP + j], acc);
    out[s] = acc;
  }
#pragma unroll
  for (int s = 0; s < NB * NB / kBlock; ++s) {
    const int idx = tid + s * kBlock;
    const int r = idx % NB, j = idx / NB;
    if (r0 + r < rows && j < nb) P[r0 + r + j * ld] = out[s];
  }
}

// dinv: optional device buffer [ceil(n/64)][64*64] that receives the inverted diagonal blocks.
int cholesky_lower(double* A, int64_t n, int64_t lda, int32_t* info_dev, double* dinv, hipStream_t stream) {
  double* tmp = nullptr;
  if (!dinv) ADMM_HIP_TRY(hipMalloc(&tmp, sizeof(double) * NB * NB));
  ADMM_HIP_TRY(hipMemsetAsync(info_dev, 0, sizeof(int32_t), stream));
  const int64_t nblk = ceil_div(n, NB);
  for (int64_t k = 0; k < nblk; ++k) {
    const int64_t k0 = k * NB;
    const int nb = static_cast<int>((n - k0 < NB) ? n - k0 : NB);
    double* inv = dinv ? dinv + k * NB * NB : tmp;
    double* Akk = A + k0 + k0 * lda;
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(kBlock), 0, stream, Akk, lda, nb, inv, info_dev,
                       static_cast<int32_t>(k0));
    const int64_t rows = n - k0 - nb;
    if (rows > 0) {
      double* P = A + (k0 + nb) + k0 * lda;
      hipLaunchKernelGGL(trsm_panel_kernel, dim3(static_cast<unsigned>(ceil_div(rows, NB))), dim3(kBlock), 0, stream,
                         P, lda, rows, nb, inv);
      double* T = A + (k0 + nb) + (k0 + nb) * lda;
      launch_gemm(0, 1, rows, rows, nb, -1.0, P, lda, P, lda, 1.0, T, lda, true, stream);
    }
  }
  if (tmp) {
    ADMM_HIP_TRY(hipStreamSynchronize(stream));
    ADMM_HIP_TRY(hipFree(tmp));
  }
  return ADMM_OK;
}

// Invert the 64x64 diagonal blocks of an existing lower factor (when L is supplied by the caller).
__global__ __launch_bounds__(kBlock) void trtri_diag_kernel(const double* __restrict__ L, int64_t ld, int64_t n,
                                                            double* __restrict__ dinv) {
  __shared__ double S[NB * NBP];
  __shared__ double X[NB * NBP];
  const int tid = threadIdx.x;
  const int64_t k0 = static_cast<int64_t>(blockIdx.x) * NB;
  const int nb = static_cast<int>((n - k0 < NB) ? n - k0 : NB);
  const double* A = L + k0 + k0 * ld;
  double* inv = dinv + static_cast<int64_t>(blockIdx.x) * NB * NB;
  for (int idx = tid; idx < NB * NB; idx += kBlock) {
    const int i = idx % NB, j = idx / NB;
    S[j * NBP + i] = (i < nb && j < nb && i >= j) ? A[i + j * ld] : ((i == j) ? 1.0 : 0.0);
  }
  __syncthreads();
  if (tid < NB) {
    const int j = tid;
    for (int i = j; i < NB; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = j; k < i; ++k) s -= S[k * NBP + i] * X[j * NBP + k];
      X[j * NBP + i] = s / S[i * NBP + i];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < NB * NB; idx += kBlock) {
    const int i = idx % NB, j = idx / NB;
    inv[i + j * NB] = (i >= j && i < nb && j < nb) ? X[j * NBP + i] : 0.0;
  }
}

void launch_trtri_diag(const double* L, int64_t n, int64_t ldl, double* dinv, hipStream_t stream) {
  hipLaunchKernelGGL(trtri_diag_kernel, dim3(static_cast<unsigned>(ceil_div(n, NB))), dim3(kBlock), 0, stream, L, ldl,
                     n, dinv);
}

// ------------------------------------------------------------------------------------
// Full inverse of the lower factor by recursive doubling:
//   inv([L11 0; L21 L22]) = [X11 0; -X22*L21*X11  X22]
// Level l pairs diagonal blocks of size b = 64*2^l; all pairs of a level run as one batched
// GEMM (uniform stride 2b*(ld+1)); ragged last pairs are handled by the GEMM's extent guards.
// The scratch product T = L21*X11 lives in the (unused) upper-right block of each pair.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void scatter_diag_inv_kernel(const double* __restrict__ dinv, int64_t n,
                                                                  double* __restrict__ X, int64_t ldx) {
  const int64_t k0 = static_cast<int64_t>(blockIdx.x) * NB;
  const double* inv = dinv + static_cast<int64_t>(blockIdx.x) * NB * NB;
  for (int idx = threadIdx.x; idx < NB * NB; idx += kBlock) {
    const int i = idx % NB, j = idx / NB;
    if (k0 + i < n && k0 + j < n) X[k0 + i + (k0 + j) * ldx] = inv[i + j * NB];
  }
}

// zero the scratch block (rows [o, o+b), cols [o+b, o+2b), o = pair*2b) of every pair of a level
__global__ __launch_bounds__(kBlock) void zero_scratch_kernel(double* __restrict__ X, int64_t n, int64_t ldx,
                                                              int64_t b) {
  const int64_t o = static_cast<int64_t>(blockIdx.y) * 2 * b;
  const int64_t j = o + b + blockIdx.x;
  if (j >= n) return;
  for (int64_t i = threadIdx.x; i < b; i += kBlock)
    if (o + i < n) X[o + i + j * ldx] = 0.0;
}

int trtri_lower_from_diag(const double* L, int64_t n, int64_t ldl, const double* dinv, double* X, int64_t ldx,
                          hipStream_t stream, bool clear) {
  if (clear) ADMM_HIP_TRY(hipMemsetAsync(X, 0, sizeof(double) * static_cast<size_t>(ldx) * n, stream));
  hipLaunchKernelGGL(scatter_diag_inv_kernel, dim3(static_cast<unsigned>(ceil_div(n, NB))), dim3(kBlock), 0, stream,
                     dinv, n, X, ldx);
  for (int64_t b = NB; b < n; b *= 2) {
    const int64_t npair = ceil_div(n, 2 * b);
    // Pairs whose second block is empty or ragged are handled by the GEMM's extent guards.
    // The scratch holds T' = X11' * L21' (size rows(block1) x rows(block2)), which is exactly the
    // shape of the pair's upper-right block even when block 2 is ragged.
    GemmArgs g{};
    g.M = b;  // i: columns of X11
    g.N = b;  // j: rows of L21
    g.K = b;
    g.alpha = 1.0;
    g.beta = 0.0;
    g.A = X;  // X11, used transposed (stored K x M)
    g.lda = ldx;
    g.strideA = 2 * b * (ldx + 1);
    g.a_rows = n;
    g.a_cols = n;
    g.shrinkA_r = 2 * b;
    g.shrinkA_c = 2 * b;
    g.B = L + b;  // L21 (rows b.., cols 0..), used transposed (stored N x K)
    g.ldb = ldl;
    g.strideB = 2 * b * (ldl + 1);
    g.b_rows = n - b;
    g.b_cols = n;
    g.shrinkB_r = 2 * b;
    g.shrinkB_c = 2 * b;
    g.C = X + b * ldx;  // upper-right block of the pair
    g.ldc = ldx;
    g.strideC = 2 * b * (ldx + 1);
    g.c_rows = n;
    g.c_cols = n - b;
    g.shrinkC_r = 2 * b;
    g.shrinkC_c = 2 * b;
    launch_gemm_args(1, 1, g, static_cast<int>(npair), stream);
    // X21 = -X22 * T = -X22 * (T')'
    GemmArgs h{};
    h.M = b;
    h.N = b;
    h.K = b;
    h.alpha = -1.0;
    h.beta = 0.0;
    h.A = X + b + b * ldx;  // X22
    h.lda = ldx;
    h.strideA = 2 * b * (ldx + 1);
    h.a_rows = n - b;
    h.a_cols = n - b;
    h.shrinkA_r = 2 * b;
    h.shrinkA_c = 2 * b;
    h.B = X + b * ldx;  // T' (stored N x K)
    h.ldb = ldx;
    h.strideB = 2 * b * (ldx + 1);
    h.b_rows = n;
    h.b_cols = n - b;
    h.shrinkB_r = 2 * b;
    h.shrinkB_c = 2 * b;
    h.C = X + b;  // X21
    h.ldc = ldx;
    h.strideC = 2 * b * (ldx + 1);
    h.c_rows = n - b;
    h.c_cols = n;
    h.shrinkC_r = 2 * b;
    h.shrinkC_c = 2 * b;
    launch_gemm_args(0, 1, h, static_cast<int>(npair), stream);
    // X11 of the next level must be strictly lower-triangular again: clear the scratch
    hipLaunchKernelGGL(zero_scratch_kernel, dim3(static_cast<unsigned>(b), static_cast<unsigned>(npair)),
                       dim3(kBlock), 0, stream, X, n, ldx, b);
  }
  return ADMM_OK;
}

// ------------------------------------------------------------------------------------
// y = L (L' x) for a lower factor L (strict upper part of the buffer ignored): builds a right-hand side whose exact
// solution is known, for the x-solve probe at setup (engine.hip).  Cold path: one wave per column, one thread per row.
__global__ __launch_bounds__(kBlock) void lt_apply_kernel(const double* __restrict__ L, int64_t ld, int64_t n,
                                                          const double* __restrict__ x, double* __restrict__ t) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 4 + wid;
  if (j >= n) return;
  const double* __restrict__ col = L + j * ld;
  double s = 0.0;
  for (int64_t i = j + lane; i < n; i += 64) s = __builtin_fma(col[i], x[i], s);
  s = wave_sum(s);
  if (lane == 0) t[j] = s;
}

__global__ __launch_bounds__(kBlock) void l_apply_kernel(const double* __restrict__ L, int64_t ld, int64_t n,
                                                         const double* __restrict__ t, double* __restrict__ y) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t ic = i < n ? i : n - 1;
  // every thread of the block walks the columns up to the block's last row (uniform trip count, coalesced loads)
  const int64_t jend = std::min<int64_t>(n, (static_cast<int64_t>(blockIdx.x) + 1) * kBlock);
  double s = 0.0;
#pragma unroll 8
  for (int64_t j = 0; j < jend; ++j) {
    const double v = L[ic + j * ld];
    s = (j <= ic) ? __builtin_fma(v, t[j], s) : s;
  }
  if (i < n) y[i] = s;
}

void launch_llt_apply(const double* L, int64_t n, int64_t ld, const double* x, double* tmp, double* y,
                      hipStream_t stream) {
  hipLaunchKernelGGL(lt_apply_kernel, dim3(static_cast<unsigned>(ceil_div(n, 4))), dim3(kBlock), 0, stream, L, ld, n, x,
                     tmp);
  hipLaunchKernelGGL(l_apply_kernel, dim3(static_cast<unsigned>(ceil_div(n, kBlock))), dim3(kBlock), 0, stream, L, ld, n,
                     tmp, y);
}

// ------------------------------------------------------------------------------------ utilities
__global__ __launch_bounds__(kBlock) void symmetrize_lower_kernel(double* __restrict__ A, int64_t n, int64_t lda) {
  // tile-transpose copy: A(j,i) = A(i,j) for i > j
  __shared__ double T[32][33];
  const int64_t bi = blockIdx.x, bj = blockIdx.y;
  if (bi < bj) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = bi * 32 + tx, j = bj * 32 + r;
    T[r][tx] = (i < n && j < n) ? A[i + j * lda] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    // write A(bj*32 + tx, bi*32 + r) = A(bi*32 + r, bj*32 + tx) = T[tx][r]
    const int64_t i = bj * 32 + tx, j = bi * 32 + r;
    if (i < n && j < n && j > i) A[i + j * lda] = T[tx][r];
  }
}

void launch_symmetrize_lower(double* A, int64_t n, int64_t lda, hipStream_t stream) {
  const unsigned nb = static_cast<unsigned>(ceil_div(n, 32));
  hipLaunchKernelGGL(symmetrize_lower_kernel, dim3(nb, nb), dim3(kBlock), 0, stream, A, n, lda);
}

__global__ void add_diag_kernel(double* __restrict__ A, int64_t n, int64_t lda, double shift) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) A[i + i * lda] += shift;
}

void launch_add_diag(double* A, int64_t n, int64_t lda, double shift, hipStream_t stream) {
  hipLaunchKernelGGL(add_diag_kernel, dim3(static_cast<unsigned>(ceil_div(n, kBlock))), dim3(kBlock), 0, stream, A, n,
                     lda, shift);
}

__global__ void fill_kernel(double* __restrict__ p, size_t n, double v) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<size_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

void launch_fill(double* p, size_t n, double v, hipStream_t stream) {
  size_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, p, n, v);
}

}  // namespace admm
