// tv2d.hip -- 2-D anisotropic total-variation ADMM kernels (engine-side extension of totalvariation.m to the
// "4096 x 4096 image, matrix-free x-update" of BASELINE config 5; the reference's solver is 1-D).
//
//   minimise 1/2*||x - s||^2 + lambda*||D x||_1,   D = [Dv; Dh]  (vertical / horizontal forward differences)
//   x-update   (I + rho*D'D) x = s + rho*D'(z - u)      5-point Neumann Laplacian: warm-started CG (cg.hip),
//                                                       the operator is the stencil below, nothing is stored
//   z-update   z = soft(u + D x, lambda/rho);  u += D x - z       (getProxOps.m:199 / admm.m:548, c = 0)
// Everything is a streaming stencil over the image; neighbour reads hit L2.
#include "kernels.h"
#include "loop_kernels.h"
#include "tv2d.h"

namespace admm {

static int tv2_blocks(int64_t n) {
  int64_t b = ceil_div(n, kBlock);
  if (b > kMaxPartBlocks) b = kMaxPartBlocks;
  if (b < 1) b = 1;
  return static_cast<int>(b);
}

// w = rho * D'D p:  (D'D p)[i,j] = sum over the existing 4-neighbours of (p[i,j] - p[nb])
__global__ __launch_bounds__(kBlock) void tv2d_laplace_kernel(int64_t H, int64_t W, double rho,
                                                              const double* __restrict__ p, double* __restrict__ w,
                                                              const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t N = H * W;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < N;
       idx += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t j = idx / H, i = idx - j * H;
    const double c = p[idx];
    const double cu = p[i > 0 ? idx - 1 : idx], cd = p[i < H - 1 ? idx + 1 : idx];
    const double cl = p[j > 0 ? idx - H : idx], cr = p[j < W - 1 ? idx + H : idx];
    double acc = 0.0;
    if (i > 0) acc += c - cu;
    if (i < H - 1) acc += c - cd;
    if (j > 0) acc += c - cl;
    if (j < W - 1) acc += c - cr;
    w[idx] = rho * acc;
  }
}

__global__ __launch_bounds__(kBlock) void tv2d_cg_pq_kernel(int64_t H, int64_t W, double rho, CgArgs a,
                                                            double* __restrict__ p_new, int nblk, int first) {
  if (a.ctrl->stop || a.st->done) return;
  __shared__ double scratch[4];
  __shared__ double bc;
  double beta = 0.0;
  if (!first) {  // (r.r)_new from the update kernel's block partials, (r.r)_old still in the state block
    double sacc = 0.0;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) sacc += a.part[kMaxPartBlocks + b];
    const double t = block_sum(sacc, scratch);
    if (threadIdx.x == 0) bc = t / a.st->rs;
    __syncthreads();
    beta = bc;
  }
  const double* __restrict__ r = a.r;
  const double* __restrict__ po = a.p;
  const int64_t N = H * W;
  double acc = 0.0;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < N;
       idx += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t j = idx / H, i = idx - j * H;
    auto pv = [&](int64_t k) { return first ? r[k] : __builtin_fma(beta, po[k], r[k]); };
    // clamped neighbour indices: the ten loads are issued together instead of one wait per guarded load
    const double c = pv(idx);
    const double cu = pv(i > 0 ? idx - 1 : idx), cd = pv(i < H - 1 ? idx + 1 : idx);
    const double cl = pv(j > 0 ? idx - H : idx), cr = pv(j < W - 1 ? idx + H : idx);
    double lap = 0.0;
    if (i > 0) lap += c - cu;
    if (i < H - 1) lap += c - cd;
    if (j > 0) lap += c - cl;
    if (j < W - 1) lap += c - cr;
    const double qv = c + rho * lap;
    p_new[idx] = c;
    a.q[idx] = qv;
    acc += c * qv;
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) a.part[blockIdx.x] = t;
}

// (D'w)[i,j] for w = [wv; wh] given as two accessors; rows of D that do not exist (i = H-1 / j = W-1) are zero
template <typename FV, typename FH>
__device__ __forceinline__ double tv2_dt(int64_t i, int64_t j, int64_t H, int64_t W, int64_t idx, FV wv, FH wh) {
  double acc = 0.0;
  if (i < H - 1) acc += wv(idx);
  if (i > 0) acc -= wv(idx - 1);
  if (j < W - 1) acc += wh(idx);
  if (j > 0) acc -= wh(idx - H);
  return acc;
}

__global__ __launch_bounds__(kBlock) void tv2d_rhs_kernel(Tv2Args a, double* __restrict__ b,
                                                          const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t H = a.H, W = a.W, N = H * W;
  const double* __restrict__ z = a.z;
  const double* __restrict__ u = a.u;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < N;
       idx += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t j = idx / H, i = idx - j * H;
    const double dt = tv2_dt(i, j, H, W, idx, [&](int64_t k) { return z[k] - u[k]; },
                             [&](int64_t k) { return z[N + k] - u[N + k]; });
    b[idx] = a.s[idx] + a.rho * dt;
  }
}

// the u half of the soft threshold: clamp(v, -t, t); z = soft(v, t) = v - clamp(v, -t, t) has the rounding of the textbook
// sign(v)*max(|v| - t, 0) (one subtraction v -+ t), in three instructions instead of eleven
__device__ __forceinline__ double tv2_clamp(double v, double t) {
  return __builtin_fmin(__builtin_fmax(v, -t), t);
}

template <int WAVES = 4>
__device__ __forceinline__ void tv2_block_partials(const double (&acc)[S_COUNT], double* part, int first, int last) {
  __shared__ double sred[WAVES][S_COUNT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) {
    const double w = wave_sum(acc[s]);
    if (lane == 0) sred[wid][s] = w;
  }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) >= first && static_cast<int>(threadIdx.x) <= last) {
    const int s = threadIdx.x;
    double tot = sred[0][s];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) tot += sred[w][s];
    part[s * kMaxPartBlocks + blockIdx.x] = tot;
  }
}

// One pass per iteration instead of three: the z/u update, the D' stencils of the dual residual / tolerance AND the
// next x-update's right-hand side b = s + rho*D'(z+ - u+).  The stencils need the NEW values of the rows i-1 (of Dv)
// and j-1 (of Dh): each thread recomputes those two neighbour updates from the old iterates (same formula, same
// inputs -> bit-identical to what the neighbour stores); the extra reads are cache hits (row i-1 sits in the same
// line, column j-1 was streamed 16 workgroups earlier on the same XCD).
//
// Compact dual state: the loop carries v = z + u (2N) instead of z and u (4N).  With z+ = soft(v+) and
// u+ = v+ - z+ (admm.m:548 up to the rounding of one subtraction), where v+ = u + D x, both old iterates are
// functions of the stored v: z = soft(v), u = v - z, evaluated with the threshold they were made with, so bitwise the
// values the previous pass used.  HBM traffic: x, v (2N), s read, v+ (2N) and b written = 7N per iteration, against
// 11N with z and u stored apart (and 9N + 6N + 6N for separate prox, dual and rhs kernels).  VIN = false is a run's
// first iteration: z, u come from the engine's iterates as given (warm starts need not satisfy z = soft(z + u)).
// launch_tv2d_expand writes z, u back out of the last v when the run ends.
constexpr int kTv2Tile = 256;      // rows of a column per workgroup step of the fused pass
constexpr int kTv2Resident = 4;    // workgroups per CU (512-row tiles with 3 per CU, 6 waves per SIMD: 201 against 196 us)
template <bool VIN>
__global__ __launch_bounds__(kTv2Tile, kTv2Resident * kTv2Tile / 256) void tv2d_fused_kernel(Tv2Args a, double* __restrict__ bnext,
                                                            const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t it = ctrl->iter;
  const int64_t H = a.H, W = a.W, N = H * W;
  const double t = a.thresh;
  // A workgroup walks tiles of kTv2Tile consecutive rows of one column; tile T = column T / cpc, row chunk T % cpc, both
  // advanced on the scalar unit.  Every address is a scalar base (column / neighbour column / second half) plus the
  // thread's 32-bit row offset, and the image borders are clamped offsets or scalar base selects, not branches.
  const uint32_t cpc = static_cast<uint32_t>((H + kTv2Tile - 1) / kTv2Tile);
  const uint32_t dj = gridDim.x / cpc, dc = gridDim.x - dj * cpc;
  int64_t j = blockIdx.x / cpc;
  uint32_t c = blockIdx.x - static_cast<uint32_t>(j) * cpc;
  const uint32_t tid = threadIdx.x;
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  for (; j < W; j += dj, c += dc) {
    if (c >= cpc) {
      c -= cpc;
      if (++j >= W) break;
    }
    const int64_t row0 = static_cast<int64_t>(c) * kTv2Tile, base = j * H + row0;
    const int64_t i = row0 + tid;
    if (i >= H) continue;
    const bool hasv = i < H - 1, up = i > 0;
    const bool hash = j < W - 1, left = j > 0;  // uniform
    const uint32_t o_up = up ? tid : tid + 1;   // relative to the element before the tile: own row when there is none
    const uint32_t o_dn = hasv ? tid + 1 : tid;
    const int64_t lbase = left ? base - H : base, rbase = hash ? base + H : base;
    const double* __restrict__ xc = a.x + base;
    const double* __restrict__ zc = a.z + base;  // VIN: v
    const double xi = xc[tid], x_dn = xc[o_dn], x_up = (xc - 1)[o_up];
    const double x_lf = (a.x + lbase)[tid], x_rt = (a.x + rbase)[tid];
    const double in0 = zc[tid], in1 = (zc + N)[tid], in_up = (zc - 1)[o_up], in_lf = (a.z + N + lbase)[tid];
    double z_own[2], u_own[2], z_up, u_up, z_lf, u_lf;
    if (VIN) {
      u_own[0] = tv2_clamp(in0, t);
      u_own[1] = tv2_clamp(in1, t);
      u_up = tv2_clamp(in_up, t);
      u_lf = tv2_clamp(in_lf, t);
      z_own[0] = in0 - u_own[0];
      z_own[1] = in1 - u_own[1];
      z_up = in_up - u_up;
      z_lf = in_lf - u_lf;
    } else {
      const double* __restrict__ uc = a.u + base;
      z_own[0] = in0;
      z_own[1] = in1;
      z_up = in_up;
      z_lf = in_lf;
      u_own[0] = uc[tid];
      u_own[1] = (uc + N)[tid];
      u_up = (uc - 1)[o_up];
      u_lf = (a.u + N + lbase)[tid];
    }
    const double si = (a.s + base)[tid];
    const double d[2] = {hasv ? xi - x_dn : 0.0, hash ? xi - x_rt : 0.0};
    double zn[2], un[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const double ax = d[part];
      const double uo = u_own[part];
      const double vn = uo + ax;
      un[part] = tv2_clamp(vn, t);
      zn[part] = vn - un[part];
      const double r = ax - zn[part], dz = zn[part] - z_own[part], du = un[part] - uo;
      acc[S_R2] += r * r;
      acc[S_AX2] += ax * ax;
      acc[S_Z2] += zn[part] * zn[part];
      acc[S_DZ2] += dz * dz;
      acc[S_U2] += un[part] * un[part];
      acc[S_DU2] += du * du;
      if (a.objevals) acc[S_OBJZ] += fabs(ax);
      (a.zo + part * N + base)[tid] = vn;
      if (a.zhist) {
        a.zhist[it * 2 * N + part * N + base + tid] = zn[part];
        a.uhist[it * 2 * N + part * N + base + tid] = un[part];
      }
    }
    if (a.objevals) {
      const double e = xi - si;
      acc[S_OBJX] += e * e;
    }
    if (a.xhist) a.xhist[it * N + base + tid] = xi;
    // the rows above / to the left: new z, u of (i-1, j) in the vertical part and of (i, j-1) in the horizontal one
    const double vnu = u_up + (x_up - xi), vnl = u_lf + (x_lf - xi);
    const double unu = tv2_clamp(vnu, t), unl = tv2_clamp(vnl, t);
    const double znu = vnu - unu, znl = vnl - unl;
    // D'w at (i, j), in tv2_dt's order of operations
    double g2 = 0.0, g3 = 0.0, gb = 0.0;
    if (hasv) {
      g2 += zn[0] - z_own[0];
      g3 += un[0];
      gb += zn[0] - un[0];
    }
    if (up) {
      g2 -= znu - z_up;
      g3 -= unu;
      gb -= znu - unu;
    }
    if (hash) {
      g2 += zn[1] - z_own[1];
      g3 += un[1];
      gb += zn[1] - un[1];
    }
    if (left) {
      g2 -= znl - z_lf;
      g3 -= unl;
      gb -= znl - unl;
    }
    acc[S_G2] += g2 * g2;
    acc[S_G3] += g3 * g3;
    (bnext + base)[tid] = si + a.rho * gb;
  }
  tv2_block_partials<kTv2Tile / kWave>(acc, a.part, 0, S_COUNT - 1);
}

// u = clamp(v), z = v - u over the 2N elements of the compact state (the iterates a run hands back)
__global__ __launch_bounds__(kBlock) void tv2d_expand_kernel(const double* __restrict__ v, double t, int64_t len,
                                                             double* __restrict__ z, double* __restrict__ u) {
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; k < len;
       k += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double vk = v[k], uk = tv2_clamp(vk, t);
    z[k] = vk - uk;
    u[k] = uk;
  }
}

void launch_tv2d_expand(const double* v, double thresh, int64_t len, double* z, double* u, hipStream_t stream) {
  int64_t blocks = ceil_div(len, kBlock);
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(tv2d_expand_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, v, thresh, len,
                     z, u);
}

// ---- building blocks of the fast / accelerated ADMM variants (the generic prox kernel does z, u, v, uhat)
// ax = D*x as a 2N-vector, the objective 1/2||x - s||^2 + lambda*||D x||_1 in block partials, the x history column
__global__ __launch_bounds__(kBlock) void tv2d_dx_kernel(int64_t H, int64_t W, double lambda, int objevals,
                                                         const double* __restrict__ x, const double* __restrict__ s,
                                                         double* __restrict__ ax, double* __restrict__ objpart,
                                                         double* __restrict__ xhist, const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  __shared__ double scratch[4];
  const int64_t it = ctrl->iter;
  const int64_t N = H * W;
  double acc = 0.0;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < N;
       idx += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t j = idx / H, i = idx - j * H;
    const double xi = x[idx];
    const double dv = (i < H - 1) ? xi - x[idx + 1] : 0.0, dh = (j < W - 1) ? xi - x[idx + H] : 0.0;
    ax[idx] = dv;
    ax[N + idx] = dh;
    if (objevals) {
      const double e = xi - s[idx];
      acc += 0.5 * (e * e) + lambda * (fabs(dv) + fabs(dh));
    }
    if (xhist) xhist[it * N + idx] = xi;
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) objpart[blockIdx.x] = t;
}

void launch_tv2d_dx(int64_t H, int64_t W, double lambda, int objevals, const double* x, const double* s, double* ax,
                    double* objpart, int* nobj_out, double* xhist, const Ctrl* ctrl, hipStream_t stream) {
  const int nb = tv2_blocks(H * W);
  *nobj_out = nb;
  hipLaunchKernelGGL(tv2d_dx_kernel, dim3(nb), dim3(kBlock), 0, stream, H, W, lambda, objevals, x, s, ax, objpart, xhist,
                     ctrl);
}

// ||D'dz||^2 (admm.m:624, 632) and ||D'u||^2 (admm.m:654) from the vectors dz = z - zprev and u, into the S_G2 / S_G3
// rows of the prox kernel's partial array (nblk blocks, as that kernel used)
__global__ __launch_bounds__(kBlock) void tv2d_dual_vec_kernel(int64_t H, int64_t W, const double* __restrict__ dz,
                                                               const double* __restrict__ u, double* __restrict__ part,
                                                               const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t N = H * W;
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < N;
       idx += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t j = idx / H, i = idx - j * H;
    const double g2 = tv2_dt(i, j, H, W, idx, [&](int64_t k) { return dz[k]; }, [&](int64_t k) { return dz[N + k]; });
    const double g3 = tv2_dt(i, j, H, W, idx, [&](int64_t k) { return u[k]; }, [&](int64_t k) { return u[N + k]; });
    acc[S_G2] += g2 * g2;
    acc[S_G3] += g3 * g3;
  }
  tv2_block_partials(acc, part, S_G2, S_G3);
}

void launch_tv2d_dual_vec(int64_t H, int64_t W, const double* dz, const double* u, double* part, int nblk,
                          const Ctrl* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(tv2d_dual_vec_kernel, dim3(nblk), dim3(kBlock), 0, stream, H, W, dz, u, part, ctrl);
}

void launch_tv2d_fused(const Tv2Args& a, bool state_in, double* bnext, const Ctrl* ctrl, int* nblk_out,
                       hipStream_t stream) {
  // every workgroup resident at once (kTv2Resident per CU), walking the tiles with a grid stride
  const int64_t tiles = ceil_div(a.H, kTv2Tile) * a.W;
  int64_t cap = static_cast<int64_t>(kTv2Resident) * 256;  // MI355X: 256 CUs
  if (cap > kMaxPartBlocks) cap = kMaxPartBlocks;
  const int nb = static_cast<int>(tiles < cap ? tiles : cap);
  *nblk_out = nb;
  if (state_in) hipLaunchKernelGGL(tv2d_fused_kernel<true>, dim3(nb), dim3(kTv2Tile), 0, stream, a, bnext, ctrl);
  else hipLaunchKernelGGL(tv2d_fused_kernel<false>, dim3(nb), dim3(kTv2Tile), 0, stream, a, bnext, ctrl);
}

void launch_tv2d_laplace(int64_t H, int64_t W, double rho, const double* p, double* w, const Ctrl* ctrl,
                         hipStream_t stream) {
  int64_t blocks = ceil_div(H * W, kBlock);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tv2d_laplace_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, H, W, rho, p, w,
                     ctrl);
}

void launch_tv2d_cg_pq(int64_t H, int64_t W, double rho, const CgArgs& a, double* p_new, bool first,
                       hipStream_t stream) {
  const int nb = cg_num_blocks(a.n);  // the partial-sum count launch_cg_update / launch_cg_advance expect
  hipLaunchKernelGGL(tv2d_cg_pq_kernel, dim3(nb), dim3(kBlock), 0, stream, H, W, rho, a, p_new, nb, first ? 1 : 0);
}

void launch_tv2d_rhs(const Tv2Args& a, double* b, const Ctrl* ctrl, hipStream_t stream) {
  int64_t blocks = ceil_div(a.H * a.W, kBlock);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tv2d_rhs_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, a, b, ctrl);
}

}  // namespace admm
