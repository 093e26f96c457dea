// gemv.hip -- HBM-streaming fp64 GEMV kernels for column-major D (gfx950, wave64).
//
//   gemv_n : y = D*x      (reference: admm.m:120 `A*v`, getProxOps.m:810/911/1088/1128, lasso.m:227)
//   gemv_t : G = D'*[v..] (reference: admm.m:119/167 `At*v`, getProxOps.m:1514, unwrappedadmm.m:130)
//
// Both are pure streaming kernels (0.25 flop/byte): every element of D is loaded exactly
// once per launch with 16-byte-per-lane coalesced loads straight into registers (no LDS
// round trip for the matrix -- it has no reuse), many loads in flight per wave.  Only the
// small vectors are staged: x through the scalar cache (gemv_n), v through LDS (gemv_t).
// Partial results are written per chunk and summed by the consumer in a fixed order, so
// results are bitwise reproducible (no float atomics).
#include "finalize_device.h"
#include "kernels.h"

namespace admm {

typedef double double2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------
// gemv_n: thread owns a pair of rows, loops over the columns of its chunk.
//   grid.x = row blocks (512 rows each), grid.y = column chunks.
// ------------------------------------------------------------------------------------
template <int UNROLL, bool NT>
__global__ __launch_bounds__(kBlock) void gemv_n_kernel(const double* __restrict__ D, int64_t m, int64_t n,
                                                        int64_t ld, const double* __restrict__ x,
                                                        double* __restrict__ ypart, int64_t ldy,
                                                        int64_t cols_per_chunk, const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  const int64_t pair = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t row = pair * 2;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * cols_per_chunk;
  const int64_t j1 = (j0 + cols_per_chunk < n) ? j0 + cols_per_chunk : n;
  if (row >= m) return;
  double* yout = ypart + static_cast<int64_t>(blockIdx.y) * ldy + row;
  if (row + 1 < m) {
    const double* p = D + row + j0 * ld;
    double2_t acc[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) acc[k] = double2_t{0.0, 0.0};
    int64_t j = j0;
    for (; j + UNROLL <= j1; j += UNROLL) {
      double2_t d[UNROLL];
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) d[k] = load2<NT>(p + k * ld);
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) {
        const double xj = x[j + k];  // wave-uniform -> scalar load
        acc[k].x = __builtin_fma(d[k].x, xj, acc[k].x);
        acc[k].y = __builtin_fma(d[k].y, xj, acc[k].y);
      }
      p += UNROLL * ld;
    }
    for (; j < j1; ++j) {
      const double2_t d = load2<NT>(p);
      const double xj = x[j];
      acc[0].x = __builtin_fma(d.x, xj, acc[0].x);
      acc[0].y = __builtin_fma(d.y, xj, acc[0].y);
      p += ld;
    }
    double2_t s = acc[0];
#pragma unroll
    for (int k = 1; k < UNROLL; ++k) {
      s.x += acc[k].x;
      s.y += acc[k].y;
    }
    *reinterpret_cast<double2_t*>(yout) = s;
  } else {  // odd m: the last row stands alone
    const double* p = D + row + j0 * ld;
    double s = 0.0;
    for (int64_t j = j0; j < j1; ++j) {
      s = __builtin_fma(*p, x[j], s);
      p += ld;
    }
    *yout = s;
  }
}

GemvNPlan gemv_n_plan(int64_t m, int64_t n, int64_t ld) {
  GemvNPlan p{};
  p.m = m;
  p.n = n;
  p.ld = ld;
  p.ldy = round_up(m, 2);
  const int64_t rowblocks = ceil_div(ceil_div(m, 2), kBlock);
  // aim for ~4096 workgroups (256 CUs x 8 blocks x 2) but keep >= 32 columns per chunk
  int64_t want = ceil_div(4096, rowblocks);
  int64_t maxchunk = n / 32 > 0 ? n / 32 : 1;
  if (want > maxchunk) want = maxchunk;
  if (want < 1) want = 1;
  if (want > 65535) want = 65535;
  p.cols_per_chunk = ceil_div(n, want);
  p.nchunk = static_cast<int32_t>(ceil_div(n, p.cols_per_chunk));
  return p;
}

void launch_gemv_n(const GemvNPlan& p, const double* D, const double* x, double* ypart, const Ctrl* ctrl,
                   hipStream_t stream) {
  dim3 grid(static_cast<unsigned>(ceil_div(ceil_div(p.m, 2), kBlock)), static_cast<unsigned>(p.nchunk));
  if (stream_hint(8 * p.m * p.n))
    hipLaunchKernelGGL((gemv_n_kernel<8, true>), grid, dim3(kBlock), 0, stream, D, p.m, p.n, p.ld, x, ypart, p.ldy,
                       p.cols_per_chunk, ctrl);
  else
    hipLaunchKernelGGL((gemv_n_kernel<8, false>), grid, dim3(kBlock), 0, stream, D, p.m, p.n, p.ld, x, ypart, p.ldy,
                       p.cols_per_chunk, ctrl);
}

__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const double* __restrict__ part, int32_t nchunk,
                                                              int64_t ld, int64_t len, double* __restrict__ y,
                                                              const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < len;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    y[i] = gather_chunks(part, nchunk, ld, i);
  }
}

void launch_sum_partials(const double* part, int32_t nchunk, int64_t ld, int64_t len, double* y, const Ctrl* ctrl,
                         hipStream_t stream) {
  int64_t blocks = ceil_div(len, kBlock);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, part, nchunk,
                     ld, len, y, ctrl);
}

// ------------------------------------------------------------------------------------
// gemv_t: block = 32 columns x one row chunk.  The chunk of V (NRHS vectors) is staged in
// LDS once per block; each wave then streams 4 columns at a time (16-B loads, 4 deep), so
// one LDS read of V feeds 4 matrix columns.  Cross-lane sums by wave shuffles.
//   grid.x = column tiles (32 columns), grid.y = row chunks.
// ------------------------------------------------------------------------------------
constexpr int kTCols = 32;  // columns per block
constexpr int kTColsPerWavePass = 4;

template <int NRHS, bool NT>
__global__ __launch_bounds__(kBlock) void gemv_t_kernel(const double* __restrict__ D, int64_t m, int64_t n,
                                                        int64_t ld, const double* __restrict__ v0,
                                                        const double* __restrict__ v1,
                                                        const double* __restrict__ v2, double* __restrict__ gpart,
                                                        int64_t ldg, int32_t rows_per_chunk,
                                                        const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  extern __shared__ __attribute__((aligned(16))) double sV[];  // [NRHS][rows_per_chunk]
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_chunk;
  const int64_t rem = m - r0;
  const int rows = rem < rows_per_chunk ? static_cast<int>(rem) : rows_per_chunk;
  const double* vs[3] = {v0, v1, v2};
#pragma unroll
  for (int r = 0; r < NRHS; ++r) {
    for (int i = threadIdx.x; i < rows_per_chunk; i += kBlock)
      sV[r * rows_per_chunk + i] = (i < rows) ? vs[r][r0 + i] : 0.0;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int npairs = rows >> 1;
  const int64_t jtile = static_cast<int64_t>(blockIdx.x) * kTCols;
#pragma unroll 1
  for (int pass = 0; pass < kTCols / (4 * kTColsPerWavePass); ++pass) {
    const int64_t jb = jtile + (pass * 4 + wid) * kTColsPerWavePass;
    if (jb >= n) break;
    // clamp out-of-range columns onto the last valid one (their results are not stored)
    const double* col[kTColsPerWavePass];
#pragma unroll
    for (int c = 0; c < kTColsPerWavePass; ++c) {
      const int64_t j = (jb + c < n) ? jb + c : n - 1;
      col[c] = D + r0 + j * ld;
    }
    double acc[kTColsPerWavePass][NRHS];
#pragma unroll
    for (int c = 0; c < kTColsPerWavePass; ++c)
#pragma unroll
      for (int r = 0; r < NRHS; ++r) acc[c][r] = 0.0;

    constexpr int U = 4;  // row steps in flight: 4 cols x 4 steps = 16 KiB per wave
    int p = lane;
    for (; p + 64 * (U - 1) < npairs; p += 64 * U) {
      double2_t d[U][kTColsPerWavePass];
#pragma unroll
      for (int k = 0; k < U; ++k)
#pragma unroll
        for (int c = 0; c < kTColsPerWavePass; ++c)
          d[k][c] = load2<NT>(col[c] + 2 * (p + 64 * k));
#pragma unroll
      for (int k = 0; k < U; ++k) {
#pragma unroll
        for (int r = 0; r < NRHS; ++r) {
          const double2_t vv = *reinterpret_cast<const double2_t*>(&sV[r * rows_per_chunk + 2 * (p + 64 * k)]);
#pragma unroll
          for (int c = 0; c < kTColsPerWavePass; ++c) {
            acc[c][r] = __builtin_fma(d[k][c].x, vv.x, acc[c][r]);
            acc[c][r] = __builtin_fma(d[k][c].y, vv.y, acc[c][r]);
          }
        }
      }
    }
    for (; p < npairs; p += 64) {
#pragma unroll
      for (int r = 0; r < NRHS; ++r) {
        const double2_t vv = *reinterpret_cast<const double2_t*>(&sV[r * rows_per_chunk + 2 * p]);
#pragma unroll
        for (int c = 0; c < kTColsPerWavePass; ++c) {
          const double2_t dd = load2<NT>(col[c] + 2 * p);
          acc[c][r] = __builtin_fma(dd.x, vv.x, acc[c][r]);
          acc[c][r] = __builtin_fma(dd.y, vv.y, acc[c][r]);
        }
      }
    }
    if ((rows & 1) && lane == 0) {  // odd tail row of the chunk
      const int i = rows - 1;
#pragma unroll
      for (int r = 0; r < NRHS; ++r) {
        const double vv = sV[r * rows_per_chunk + i];
#pragma unroll
        for (int c = 0; c < kTColsPerWavePass; ++c) acc[c][r] = __builtin_fma(col[c][i], vv, acc[c][r]);
      }
    }
#pragma unroll
    for (int c = 0; c < kTColsPerWavePass; ++c) {
#pragma unroll
      for (int r = 0; r < NRHS; ++r) {
        const double s = wave_sum(acc[c][r]);
        if (lane == 0 && jb + c < n)
          gpart[(static_cast<int64_t>(blockIdx.y) * NRHS + r) * ldg + jb + c] = s;
      }
    }
  }
}

GemvTPlan gemv_t_plan(int64_t m, int64_t n, int64_t ld) {
  GemvTPlan p{};
  p.m = m;
  p.n = n;
  p.ld = ld;
  p.ldg = round_up(n, 2);
  const int64_t coltiles = ceil_div(n, kTCols);
  // row chunk: 2048 rows (48 KiB of LDS at 3 RHS -> 3 blocks/CU) unless the grid would be
  // too small to fill 256 CUs, then shrink down to 256 rows.
  int32_t rc = 2048;
  while (rc > 256 && coltiles * ceil_div(m, rc) < 2048) rc >>= 1;
  p.rows_per_chunk = rc;
  p.nchunk = static_cast<int32_t>(ceil_div(m, rc));
  return p;
}

void launch_gemv_t(const GemvTPlan& p, const double* D, const double* v0, const double* v1, const double* v2,
                   int nrhs, double* gpart, const Ctrl* ctrl, hipStream_t stream) {
  dim3 grid(static_cast<unsigned>(ceil_div(p.n, kTCols)), static_cast<unsigned>(p.nchunk));
  const size_t lds = static_cast<size_t>(nrhs) * p.rows_per_chunk * sizeof(double);
  const bool nt = stream_hint(8 * p.m * p.n);
#define ADMM_LAUNCH_GEMV_T(NR, NTV)                                                                                 \
  hipLaunchKernelGGL((gemv_t_kernel<NR, NTV>), grid, dim3(kBlock), lds, stream, D, p.m, p.n, p.ld, v0, v1, v2, gpart, \
                     p.ldg, p.rows_per_chunk, ctrl)
  switch (nrhs) {
    case 1:
      if (nt) ADMM_LAUNCH_GEMV_T(1, true);
      else ADMM_LAUNCH_GEMV_T(1, false);
      break;
    case 2:
      if (nt) ADMM_LAUNCH_GEMV_T(2, true);
      else ADMM_LAUNCH_GEMV_T(2, false);
      break;
    default:
      if (nt) ADMM_LAUNCH_GEMV_T(3, true);
      else ADMM_LAUNCH_GEMV_T(3, false);
      break;
  }
#undef ADMM_LAUNCH_GEMV_T
}

// g[r][j] = sum_c gpart[c][r][j].  Workgroup = 16 consecutive columns x 16 slots that split the chunk
// index; the slots are combined through LDS in a fixed order (bitwise reproducible).  A serial loop over
// the chunks per column (the first version) cost 30 us at 235 chunks -- a fifth of an SVM 60000x400 iteration.
__global__ __launch_bounds__(kBlock) void sum_partials_t_kernel(const double* __restrict__ gpart, int32_t nchunk,
                                                                int nrhs, int64_t ldg, int64_t n,
                                                                double* __restrict__ g, int64_t ldg_out,
                                                                const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double sacc[16][17];
  const int jj = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int64_t tiles_per_rhs = (n + 15) / 16;
  const int r = static_cast<int>(blockIdx.x / tiles_per_rhs);
  const int64_t j = (blockIdx.x - static_cast<int64_t>(r) * tiles_per_rhs) * 16 + jj;
  double s = 0.0;
  if (j < n) {
#pragma unroll 4
    for (int32_t c = slot; c < nchunk; c += 16) s += gpart[(static_cast<int64_t>(c) * nrhs + r) * ldg + j];
  }
  sacc[slot][jj] = s;
  __syncthreads();
  if (slot == 0 && j < n) {
    double t = sacc[0][jj];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sacc[k][jj];
    g[r * ldg_out + j] = t;
  }
}

// the same with a passenger: the last workgroup runs the deferred finalize logic of the iteration whose element update
// has just run (engine_run.hip: defer_fin_ad) next to the partial sums instead of serially inside that update's launch
__global__ __launch_bounds__(kBlock) void sum_partials_t_fin_kernel(const double* __restrict__ gpart, int32_t nchunk,
                                                                    int nrhs, int64_t ldg, int64_t n,
                                                                    double* __restrict__ g, int64_t ldg_out, FinArgs f,
                                                                    const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  if (blockIdx.x == gridDim.x - 1) {
    finalize_body<false>(f);
    return;
  }
  __shared__ double sacc[16][17];
  const int jj = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int64_t tiles_per_rhs = (n + 15) / 16;
  const int r = static_cast<int>(blockIdx.x / tiles_per_rhs);
  const int64_t j = (blockIdx.x - static_cast<int64_t>(r) * tiles_per_rhs) * 16 + jj;
  double s = 0.0;
  if (j < n) {
#pragma unroll 4
    for (int32_t c = slot; c < nchunk; c += 16) s += gpart[(static_cast<int64_t>(c) * nrhs + r) * ldg + j];
  }
  sacc[slot][jj] = s;
  __syncthreads();
  if (slot == 0 && j < n) {
    double t = sacc[0][jj];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sacc[k][jj];
    g[r * ldg_out + j] = t;
  }
}

void launch_sum_partials_t_fin(const GemvTPlan& p, const double* gpart, int nrhs, double* g, int64_t ldg_out,
                               const FinArgs& f, const Ctrl* ctrl, hipStream_t stream) {
  const int64_t blocks = ceil_div(p.n, 16) * nrhs + 1;
  hipLaunchKernelGGL(sum_partials_t_fin_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, gpart,
                     p.nchunk, nrhs, p.ldg, p.n, g, ldg_out, f, ctrl);
}

void launch_sum_partials_t(const GemvTPlan& p, const double* gpart, int nrhs, double* g, int64_t ldg_out,
                           const Ctrl* ctrl, hipStream_t stream) {
  const int64_t blocks = ceil_div(p.n, 16) * nrhs;
  hipLaunchKernelGGL(sum_partials_t_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, gpart,
                     p.nchunk, nrhs, p.ldg, p.n, g, ldg_out, ctrl);
}

}  // namespace admm
