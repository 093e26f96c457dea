// common.h -- shared host/device helpers for libadmm_hip (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/admm_engine.h"

namespace admm {

constexpr int kWave = 64;        // CDNA wavefront width
constexpr int kBlock = 256;      // default workgroup: 4 waves, one per SIMD
constexpr int kMaxPartBlocks = 1024;  // cap on blocks that emit reduction partials

// ---- error plumbing -------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define ADMM_HIP_TRY(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      return ::admm::fail(ADMM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    }                                                                                   \
  } while (0)

#define ADMM_TRY(expr)       \
  do {                       \
    int _rc = (expr);        \
    if (_rc != ADMM_OK) return _rc; \
  } while (0)

inline int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Device control block: lives in device memory, mirrored to pinned host memory when
// the host polls.  Every loop kernel starts with `if (ctrl->stop) return;` so that
// iterations enqueued speculatively after a stop condition fired are no-ops.
struct Ctrl {
  int32_t stop;            // a stop condition fired (admm.m:706-722) or convtest aborted
  int32_t iter;            // completed iterations == 0-based index of the one in flight
  int32_t steps;           // results.steps (admm.m:746)
  int32_t convfail;        // iteration (1-based) at which convtest aborted (admm.m:692-701)
  int32_t cg_total;        // accumulated inner CG iterations
  int32_t arrive;          // arrival counter of the one-launch tail (prox_fin_kernel); zero between launches
  double acurr, aprev;     // fast ADMM alpha (admm.m:271-272, 567, 578)
  double d, dprev;         // accelerated ADMM restart value (admm.m:278-279, 572-588)
  double coef;             // (aprev-1)/acurr for the extrapolation in flight
  double restart_flag;     // 1 if the iteration in flight restarted
  double obj_bound;        // largest cancellation bound of the right-hand-side objective form seen in this run
                           // (finalize_device.h; 0 when that form is not in use)
};

// Matrices larger than this are streamed with non-temporal loads: they cannot stay in L2 / Infinity
// Cache between passes anyway, and the streaming hint is worth +13-15 % read bandwidth on MI355X
// (dev/read_bw.hip: 6.2 -> 7.1 TB/s).  Smaller ones keep normal loads so that they stay cache-resident.
constexpr int64_t kStreamBytes = int64_t{192} << 20;
inline bool stream_hint(int64_t bytes) { return bytes > kStreamBytes; }

#ifdef __HIPCC__
typedef double admm_double2 __attribute__((ext_vector_type(2)));
// 16-byte matrix load, NT = non-temporal (streaming) hint
template <bool NT>
__device__ __forceinline__ admm_double2 load2(const double* p) {
  const admm_double2* q = reinterpret_cast<const admm_double2*>(p);
  return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ __forceinline__ void store2(double* p, admm_double2 v) {
  admm_double2* q = reinterpret_cast<admm_double2*>(p);
  if (NT) __builtin_nontemporal_store(v, q);
  else *q = v;
}
template <bool NT>
__device__ __forceinline__ double load1(const double* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// ---- wave / block reductions (wave64 shuffles, then LDS across the 4 waves) -------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// Sum over the block; result valid in thread 0.  `scratch` holds >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // protect scratch reuse across consecutive calls
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) r += scratch[w];
  }
  return r;
}
#endif

// sum of chunk partials for element i, in chunk order (same rounding as the plain loop); four loads in flight at a
// time instead of one dependent load + add per chunk
__device__ __forceinline__ double gather_chunks(const double* __restrict__ src, int32_t nchunk, int64_t ld, int64_t i) {
  double ax = 0.0;
  int32_t c = 0;
  for (; c + 4 <= nchunk; c += 4) {
    const double p0 = src[static_cast<int64_t>(c) * ld + i], p1 = src[static_cast<int64_t>(c + 1) * ld + i];
    const double p2 = src[static_cast<int64_t>(c + 2) * ld + i], p3 = src[static_cast<int64_t>(c + 3) * ld + i];
    ax = (((ax + p0) + p1) + p2) + p3;
  }
  for (; c < nchunk; ++c) ax += src[static_cast<int64_t>(c) * ld + i];
  return ax;
}



}  // namespace admm
