// dct.hip -- LDS-resident fp64 DCT-II / DCT-III (length 2^p <= 8192) and the spectral x-update of 2-D total
// variation (dct.h).  One workgroup transforms TWO real sequences at once as one complex FFT (z = a + i*b):
//   Makhoul's permutation   v[n] = x[2n], v[N-1-n] = x[2n+1]        turns the DCT-II into an N-point FFT of v,
//   X_k = Re(e^{-i pi k/(2N)} V_k),  X_{N-k} = -Im(e^{-i pi k/(2N)} V_k),
// and V_a, V_b come out of the joint spectrum by the usual even/odd split.  The FFT is an in-place radix-2
// decimation-in-frequency network, four stages fused per pass in registers (16 points per thread: three LDS round
// trips for 4096 points), natural order in -> bit-reversed order out; the inverse runs the same network backwards
// (decimation in time, conjugate twiddles), bit-reversed in -> natural out, so nothing is ever reordered: the
// spectral step addresses position bitrev(k) directly.
#include <cmath>
#include <utility>
#include <vector>

#include "dct.h"
#include "finalize_device.h"
#include "kernels.h"
#include "tv2d_pixel.h"

namespace admm {

using c64 = admm_double2;  // (re, im)

__device__ __forceinline__ c64 cmul(c64 a, c64 b) {
  return c64{__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x)};
}
__device__ __forceinline__ c64 cmulc(c64 a, c64 b) {  // a * conj(b)
  return c64{__builtin_fma(a.x, b.x, a.y * b.y), __builtin_fma(a.y, b.x, -(a.x * b.y))};
}
__device__ __forceinline__ int bitrev(int k, int log2n) { return static_cast<int>(__brev(static_cast<unsigned>(k)) >> (32 - log2n)); }

// LDS position of element a: the low four index bits are XORed with bits 4-7 and 8-11, which makes every access
// pattern of this file conflict-free per 16-lane group: consecutive a, a = 16*lane + e (the last pass of a
// 4096-point transform), and a = 256*c + const (bit-reversed order seen from consecutive k).
__device__ __forceinline__ int swz(int a) { return a ^ ((a >> 4) & 15) ^ ((a >> 8) & 15); }

template <int R>
__device__ __forceinline__ c64 unit_root(int x) {  // e^{-2 pi i x / R}, x < R/2 (compile-time after unrolling)
  constexpr double kC16[8] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173,
                              0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613};
  constexpr double kS16[8] = {0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613,
                              -1.0, -0.92387953251128675613, -0.70710678118654752440, -0.38268343236508977173};
  const int i = x * (16 / R);
  return c64{kC16[i], kS16[i]};
}

// log2(R) fused radix-2 stages on blocks of B points: thread q owns the R points base + e*B/R.  DIF (forward) pairs
// (e, e + half) with twiddle w_n^{(j + el*B/R) * (n/B) * 2^s} = tw[j*(n/B) << s] * e^{-2 pi i (el << s) / R};
// INV runs the stages backwards with conjugate twiddles (the exact inverse network, times R).
// tws = 1: tw is the table of a transform of length 2n (an n-point transform inside a 2n-point real one)
template <int R, bool INV>
__device__ __forceinline__ void fft_pass(c64* zs, int n, int B, const c64* __restrict__ tw, int tws = 0) {
  constexpr int LR = (R == 16) ? 4 : (R == 8) ? 3 : (R == 4) ? 2 : 1;
  const int Bq = B / R;
  const int T = n / B;
  const int lg = 31 - __clz(Bq);
  for (int q = threadIdx.x; q < n / R; q += blockDim.x) {
    const int j = q & (Bq - 1);
    const int base = (q >> lg) * B + j;
    c64 v[R];
#pragma unroll
    for (int e = 0; e < R; ++e) v[e] = zs[swz(base + e * Bq)];
#pragma unroll
    for (int ss = 0; ss < LR; ++ss) {
      const int s = INV ? LR - 1 - ss : ss;
      const int half = R >> (s + 1);
      const c64 wj = tw[((j * T) << s) << tws];
#pragma unroll
      for (int e = 0; e < R; ++e) {
        if (e & half) continue;
        const int el = e & (half - 1);
        const c64 w = (el == 0) ? wj : cmul(wj, unit_root<R>(el << s));
        const c64 a = v[e], b = v[e + half];
        if (!INV) {
          v[e] = a + b;
          v[e + half] = cmul(a - b, w);
        } else {
          const c64 t = cmulc(b, w);
          v[e] = a + t;
          v[e + half] = a - t;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < R; ++e) zs[swz(base + e * Bq)] = v[e];
  }
}

// callers synchronise before; every pass ends with a barrier.  Radix plan: 16 while four or more stages remain, then
// one pass of 8 / 4 / 2 on the smallest blocks; the inverse replays the same passes in reverse.
template <bool INV>
__device__ __forceinline__ void fft_network(c64* zs, int n, int log2n, const c64* __restrict__ tw, int tws = 0) {
  const int rem = log2n & 3;        // stages of the odd pass (blocks of 2^rem points)
  if (!INV) {
    int B = n;
    for (; B >= 16; B >>= 4) {
      fft_pass<16, false>(zs, n, B, tw, tws);
      __syncthreads();
    }
    if (rem == 3) fft_pass<8, false>(zs, n, 8, tw, tws);
    else if (rem == 2) fft_pass<4, false>(zs, n, 4, tw, tws);
    else if (rem == 1) fft_pass<2, false>(zs, n, 2, tw, tws);
    if (rem) __syncthreads();
  } else {
    if (rem == 3) fft_pass<8, true>(zs, n, 8, tw, tws);
    else if (rem == 2) fft_pass<4, true>(zs, n, 4, tw, tws);
    else if (rem == 1) fft_pass<2, true>(zs, n, 2, tw, tws);
    if (rem) __syncthreads();
    for (int B = 16 << rem; B <= n; B <<= 4) {
      fft_pass<16, true>(zs, n, B, tw, tws);
      __syncthreads();
    }
  }
}


// (a, b) -> LDS in Makhoul order: x[2i] -> v[i], x[2i+1] -> v[n-1-i], one 16-byte load per column and thread
__device__ __forceinline__ void load_pair(c64* zs, const double* a, const double* b, int n) {
#pragma unroll 8
  for (int i = threadIdx.x; i < (n >> 1); i += blockDim.x) {
    const admm_double2 va = load2<false>(a + 2 * i), vb = load2<false>(b + 2 * i);
    zs[swz(i)] = c64{va.x, vb.x};
    zs[swz(n - 1 - i)] = c64{va.y, vb.y};
  }
}
// LDS (after the inverse network) -> (a, b), scaled
__device__ __forceinline__ void store_pair(const c64* zs, double* a, double* b, int n, double scale) {
#pragma unroll 4
  for (int i = threadIdx.x; i < (n >> 1); i += blockDim.x) {
    const c64 v0 = zs[swz(i)], v1 = zs[swz(n - 1 - i)];
    store2<false>(a + 2 * i, admm_double2{v0.x * scale, v1.x * scale});
    store2<false>(b + 2 * i, admm_double2{v0.y * scale, v1.y * scale});
  }
}

// joint bit-reversed spectrum -> the four DCT coefficients of a pair (k, n-k), 0 < k < n/2
__device__ __forceinline__ void spectrum_to_dct(c64 zk, c64 zn, c64 ck, double& xak, double& xan, double& xbk,
                                                double& xbn) {
  const c64 va = c64{0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y)};
  const c64 vb = c64{0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x)};
  const c64 pa = cmul(ck, va), pb = cmul(ck, vb);
  xak = pa.x;
  xan = -pa.y;
  xbk = pb.x;
  xbn = -pb.y;
}
__device__ __forceinline__ void dct_to_spectrum(double xak, double xan, double xbk, double xbn, c64 ck, c64& zk,
                                                c64& zn) {
  const c64 va = cmulc(c64{xak, -xan}, ck), vb = cmulc(c64{xbk, -xbn}, ck);
  zk = c64{va.x - vb.y, va.y + vb.x};
  zn = c64{va.x + vb.y, vb.x - va.y};
}

// block partials of the S_COUNT sums of a 4-wave workgroup -> part[s][blockIdx.x]
__device__ __forceinline__ void tv2_block_partials_lds(const double (&acc)[S_COUNT], double* __restrict__ part) {
  __shared__ double sred[kBlock / kWave][S_COUNT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) {
    const double w = wave_sum(acc[s]);
    if (lane == 0) sred[wid][s] = w;
  }
  __syncthreads();
  if (threadIdx.x < S_COUNT) {
    const int s = threadIdx.x;
    double tot = sred[0][s];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w) tot += sred[w][s];
    part[s * kMaxPartBlocks + blockIdx.x] = tot;
  }
}

constexpr double kSqrtHalf = 0.70710678118654752440;
constexpr double kSqrt2 = 1.41421356237309504880;

// zs holds the pair (a, b) in Makhoul order (callers synchronise before): FFT, then the DCT-II coefficients to a, b
__device__ __forceinline__ void dct_forward_from_lds(c64* zs, double* __restrict__ a, double* __restrict__ b,
                                                     const DctTables& t) {
  const int n = t.n, p = t.log2n;
  fft_network<false>(zs, n, p, t.tw);
#pragma unroll 2
  for (int k = threadIdx.x; k <= (n >> 1); k += blockDim.x) {
    if (k == 0) {
      a[0] = zs[0].x;
      b[0] = zs[0].y;
    } else if (k == (n >> 1)) {
      const c64 h = zs[1];  // bitrev(n/2) = 1
      a[k] = kSqrtHalf * h.x;
      b[k] = kSqrtHalf * h.y;
    } else {
      double xak, xan, xbk, xbn;
      spectrum_to_dct(zs[swz(bitrev(k, p))], zs[swz(bitrev(n - k, p))], t.c4[k], xak, xan, xbk, xbn);
      a[k] = xak;
      a[n - k] = xan;
      b[k] = xbk;
      b[n - k] = xbn;
    }
  }
}

__device__ __forceinline__ void dct_cols_forward_body(double* __restrict__ img, int64_t H, const DctTables& t,
                                                      unsigned pair) {
  extern __shared__ c64 zs[];
  double* a = img + static_cast<int64_t>(2 * pair) * H;
  double* b = (static_cast<int32_t>(pair) == t.odd_pair) ? a : a + H;  // (an odd width's last column: both halves of the pair)
  load_pair(zs, a, b, t.n);
  __syncthreads();
  dct_forward_from_lds(zs, a, b, t);
}

// ---------------------------------------------------------------- fused 2-D TV pass -> forward column transform
// The fused pass (tv2d.hip) ends in the next x-update's right-hand side b, and the x-update begins with the column DCT
// of b: here b never reaches memory.  A workgroup owns column pairs; for a pair it runs the pixel update of both columns
// (tv2d_pixel.h: the same arithmetic; x, v, s from global memory, the neighbours' reads are cache hits), drops b into
// the FFT's LDS image at its Makhoul position, transforms, and stores the coefficients where the row stage expects
// them.  Against the two kernels it replaces: one write and one read of the image less per iteration (11 N doubles
// instead of 13 N), one launch less.
template <bool VIN, bool ODDW>
__global__ __launch_bounds__(kBlock) void tv2d_fused_dct_kernel(Tv2Args a, double* __restrict__ bhat, DctTables t,
                                                                const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  extern __shared__ c64 zs[];
  const int64_t it = ctrl->iter;
  const int64_t H = a.H, W = a.W, N = H * W;
  const int n = t.n;  // == H
  const uint32_t tid = threadIdx.x;
  double acc[S_COUNT];
#pragma unroll
  for (int s = 0; s < S_COUNT; ++s) acc[s] = 0.0;
  // ODDW (an odd width): the last pair is its single column twice.  A separate instantiation: the even-width body has
  // no control flow between the loads of its two columns and their updates (with the uniform branch of the odd form in
  // it the launch took 241 instead of 215 us at 4096^2)
  const int64_t npairs = ODDW ? (W + 1) >> 1 : W >> 1;
  // Neighbouring pairs share two columns of x and one of v (the stencil's halo).  Workgroup b runs on XCD b mod 8
  // (round-robin placement; speed only): within every 64 consecutive pairs, XCD c takes the 8 adjacent pairs
  // [8c, 8c + 8), so that seven of eight halos are hits in its own L2 while all XCDs still work inside one 128-column
  // window.  (Dealt round-robin, every XCD fetched its own copy of the halo: 743 MB read per launch at 4096^2 where x,
  // v and s are 537 MB.  Whole eighths of the image per XCD were 30 us SLOWER than that: the eight streams sit
  // 16 MB apart and meet in the same HBM channels.)
  const bool xcd_groups = (npairs & 63) == 0 && (gridDim.x & 63u) == 0;
  for (int64_t w = blockIdx.x; w < npairs; w += gridDim.x) {
    const int64_t b = w & 63;
    const int64_t pair = xcd_groups ? (w - b) + (b & 7) * 8 + (b >> 3) : w;
    const int64_t j0 = 2 * pair;
    // NC row chunks of kBlock rows per round, the loads of all of them (2 columns x 10 values each) issued before any
    // update: a column pair is a chain of H / (NC*kBlock) dependent round trips to memory, and two workgroups per CU
    // (64 KB of LDS each at H = 4096) do not hide them
    auto rows = [&](auto nc_tag, int64_t row0) {
      constexpr int NC = decltype(nc_tag)::value;
      Tv2Px px[NC][2];
      int64_t base[NC][2];
      uint32_t o_up[NC], o_dn[NC];
      bool hasv[NC], up[NC];
#pragma unroll
      for (int h = 0; h < NC; ++h) {
        const int64_t r0 = row0 + h * kBlock, i = r0 + tid;
        hasv[h] = i < H - 1;
        up[h] = i > 0;
        o_up[h] = up[h] ? tid : tid + 1;  // relative to the element before the chunk: own row when there is none
        o_dn[h] = hasv[h] ? tid + 1 : tid;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int64_t j = (!ODDW || j0 + c < W) ? j0 + c : j0;
          base[h][c] = j * H + r0;
          const int64_t lbase = j > 0 ? base[h][c] - H : base[h][c], rbase = j < W - 1 ? base[h][c] + H : base[h][c];
          tv2px_load<VIN>(a, N, base[h][c], lbase, rbase, tid, o_up[h], o_dn[h], px[h][c]);
        }
      }
#pragma unroll
      for (int h = 0; h < NC; ++h) {
        const int64_t i = row0 + h * kBlock + tid;
        const int k = static_cast<int>(i >> 1);
        const int pos = swz((i & 1) ? n - 1 - k : k);  // Makhoul order: x[2k] -> v[k], x[2k+1] -> v[n-1-k]
        double bv0 = 0.0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int64_t j = j0 + c;
          if (!ODDW || j < W) {  // (uniform)
            const double bv = tv2px_apply<VIN>(a, N, it, base[h][c], tid, hasv[h], up[h], j < W - 1, j > 0, px[h][c], acc);
            if (c == 0) bv0 = bv;
            reinterpret_cast<double*>(&zs[pos])[c] = bv;
          } else {
            reinterpret_cast<double*>(&zs[pos])[c] = bv0;
          }
        }
      }
    };
    if (H >= 2 * kBlock) {  // (H is a power of two: whole double chunks)
      for (int64_t row0 = 0; row0 < H; row0 += 2 * kBlock) rows(std::integral_constant<int, 2>{}, row0);
    } else {
      for (int64_t row0 = 0; row0 < H; row0 += kBlock)
        if (row0 + tid < H) rows(std::integral_constant<int, 1>{}, row0);
    }
    __syncthreads();
    double* ca = bhat + j0 * H;
    dct_forward_from_lds(zs, ca, (!ODDW || j0 + 1 < W) ? ca + H : ca, t);
    __syncthreads();  // zs is the next pair's
  }
  tv2_block_partials_lds(acc, a.part);
}

__global__ __launch_bounds__(kBlock) void dct_cols_forward_kernel(double* __restrict__ img, int64_t H, DctTables t,
                                                                  const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  dct_cols_forward_body(img, H, t, blockIdx.x);
}

// The same with a passenger: workgroup 0 (dispatched first) runs the finalize logic of the PREVIOUS 2-D TV iteration
// while the others transform the new right-hand side in place.  Nothing in this launch depends on that decision, and
// the three launches behind it (row stage, inverse transform, fused z/u pass) start after it and no-op when it has
// raised ctrl->stop -- x and the dual state of the last real iteration stay as they are (engine_run_tv.hip).
__global__ __launch_bounds__(kBlock) void dct_cols_forward_fin_kernel(double* __restrict__ img, int64_t H, DctTables t,
                                                                      FinArgs f, int32_t fin_pending,
                                                                      const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  if (blockIdx.x == 0) {
    if (fin_pending) finalize_body<false>(f);
    return;
  }
  dct_cols_forward_body(img, H, t, blockIdx.x - 1u);
}

__global__ __launch_bounds__(kBlock) void dct_cols_inverse_kernel(const double* __restrict__ src,
                                                                  double* __restrict__ dst, int64_t H, DctTables t,
                                                                  const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  extern __shared__ c64 zs[];
  const int n = t.n, p = t.log2n;
  const bool odd = static_cast<int32_t>(blockIdx.x) == t.odd_pair;
  const double* a = src + static_cast<int64_t>(2 * blockIdx.x) * H;
  const double* b = odd ? a : a + H;
  // coefficient pairs (k, n - k), 0 <= k < n/2: kInvQ of them per thread and round, every load of a round issued before
  // any is used (clamped indices, no branches around loads: with the loads inside the three cases of the loop body the
  // kernel fetched two pairs per round trip -- four dependent trips for a 4096-point column pair)
  constexpr int kInvQ = 8;
  const int half = n >> 1;
  for (int k0 = threadIdx.x; k0 < half; k0 += kInvQ * blockDim.x) {
    double ak[kInvQ], an[kInvQ], bk[kInvQ], bn[kInvQ];
    c64 ck[kInvQ];
#pragma unroll
    for (int q = 0; q < kInvQ; ++q) {
      const int k = k0 + q * static_cast<int>(blockDim.x);
      const int kc = k < half ? k : half - 1, nk = kc == 0 ? half : n - kc;  // (k = 0 pairs with n/2: the two special ones)
      ak[q] = a[kc];
      an[q] = a[nk];
      bk[q] = b[kc];
      bn[q] = b[nk];
      ck[q] = t.c4[kc];
    }
#pragma unroll
    for (int q = 0; q < kInvQ; ++q) {
      const int k = k0 + q * static_cast<int>(blockDim.x);
      if (k >= half) continue;
      if (k == 0) {
        zs[0] = c64{ak[q], bk[q]};
        zs[1] = c64{kSqrt2 * an[q], kSqrt2 * bn[q]};  // k = n/2 (bitrev(n/2) = 1)
      } else {
        c64 zk, zn;
        dct_to_spectrum(ak[q], an[q], bk[q], bn[q], ck[q], zk, zn);
        zs[swz(bitrev(k, p))] = zk;
        zs[swz(bitrev(n - k, p))] = zn;
      }
    }
  }
  __syncthreads();
  fft_network<true>(zs, n, p, t.tw);
  double* oa = dst + static_cast<int64_t>(2 * blockIdx.x) * H;
  store_pair(zs, oa, odd ? oa : oa + H, n, 1.0 / static_cast<double>(n));
}

// ---------------------------------------------------------------- column transforms of ANY length (chirp form)
// For a height that is not a power of two the n-point DFT behind the DCT is taken as a circular convolution (Bluestein):
// with c_j = e^{-pi i j^2/n},  V_k = c_k * sum_j (v_j c_j) conj(c_(k-j)):  one forward FFT of length bm >= 2n - 1 of the
// chirped, zero-padded sequence, a pointwise product with the filter's spectrum (precomputed, 1/bm folded in, stored at
// the network's bit-reversed output positions), one inverse FFT, and the chirp again.  The inverse DFT is the same with
// conjugated chirps and the conjugated filter spectrum (the filter is even).  Two real columns per workgroup as one
// complex sequence, as in the power-of-two kernels; ~4.5 x their work per column, and still an order of magnitude ahead
// of the ~33 CG iterations the x-update costs otherwise.  Scalar loads and stores: any n, any column alignment.
__device__ __forceinline__ int makhoul_pos(int j, int n) { return (j & 1) ? n - 1 - (j >> 1) : (j >> 1); }  // x index -> v index

__device__ __forceinline__ void chirp_convolve(c64* zs, const DctTables& t, bool conj_filter) {
  const int M = t.bm;
  __syncthreads();
  fft_network<false>(zs, M, t.log2bm, t.tw);
  for (int p = threadIdx.x; p < M; p += blockDim.x) {
    const c64 h = t.hbr[p];
    zs[swz(p)] = conj_filter ? cmulc(zs[swz(p)], h) : cmul(zs[swz(p)], h);
  }
  __syncthreads();
  fft_network<true>(zs, M, t.log2bm, t.tw);
}

// (512 threads where the FFT has 8192 points -- 128 KB of LDS, one workgroup per CU: every radix-16 pass then has one
// butterfly group per thread instead of two rounds of 256)
constexpr int kChirpMaxThreads = 512;
static int chirp_threads(int M) { return M >= 8192 ? kChirpMaxThreads : kBlock; }

template <bool FIN>
__global__ __launch_bounds__(kChirpMaxThreads) void dct_cols_forward_chirp_kernel(double* __restrict__ img, int64_t H, DctTables t,
                                                                        FinArgs f, int32_t fin_pending,
                                                                        const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  if (FIN && blockIdx.x == 0) {
    if (fin_pending && threadIdx.x < kBlock) finalize_body<false>(f);  // (written for one 256-thread workgroup)
    return;
  }
  extern __shared__ c64 zs[];
  const int n = t.n, M = t.bm;
  const unsigned pair = blockIdx.x - (FIN ? 1u : 0u);
  double* a = img + static_cast<int64_t>(2 * pair) * H;
  double* b = (static_cast<int32_t>(pair) == t.odd_pair) ? a : a + H;
  for (int j = threadIdx.x; j < M; j += blockDim.x) {  // v_j c_j in Makhoul order, zeros behind
    c64 y{0.0, 0.0};
    if (j < n) {
      const int v = makhoul_pos(j, n);
      y = cmul(c64{a[j], b[j]}, t.chirp[v]);
      zs[swz(v)] = y;
    } else {
      zs[swz(j)] = y;
    }
  }
  chirp_convolve(zs, t, false);
  for (int k = threadIdx.x; k < n; k += blockDim.x) zs[swz(k)] = cmul(zs[swz(k)], t.chirp[k]);  // V_k (of a + i b)
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += blockDim.x) {  // X_k = Re(e^{-i pi k/(2n)} V_k) for either column
    const c64 zk = zs[swz(k)], zn = zs[swz(k == 0 ? 0 : n - k)];
    const c64 va = c64{0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y)};
    const c64 vb = c64{0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x)};
    const c64 ck = t.c4[k];
    a[k] = cmul(ck, va).x;
    b[k] = cmul(ck, vb).x;
  }
}

__global__ __launch_bounds__(kChirpMaxThreads) void dct_cols_inverse_chirp_kernel(const double* __restrict__ src,
                                                                        double* __restrict__ dst, int64_t H, DctTables t,
                                                                        const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  extern __shared__ c64 zs[];
  const int n = t.n, M = t.bm;
  const bool odd = static_cast<int32_t>(blockIdx.x) == t.odd_pair;
  const double* a = src + static_cast<int64_t>(2 * blockIdx.x) * H;
  const double* b = odd ? a : a + H;
  for (int k = threadIdx.x; k < M; k += blockDim.x) {  // Z_k conj(c_k), zeros behind
    c64 y{0.0, 0.0};
    if (k < n) {
      const int nk = k == 0 ? 0 : n - k;
      const c64 ck = t.c4[k];
      c64 va = cmulc(c64{a[k], -a[nk]}, ck), vb = cmulc(c64{b[k], -b[nk]}, ck);  // V = e^{+i pi k/(2n)} (X_k - i X_(n-k))
      if (k == 0) {
        va = c64{a[0], 0.0};
        vb = c64{b[0], 0.0};
      }
      y = cmulc(c64{va.x - vb.y, va.y + vb.x}, t.chirp[k]);
    }
    zs[swz(k)] = y;
  }
  chirp_convolve(zs, t, true);
  double* oa = dst + static_cast<int64_t>(2 * blockIdx.x) * H;
  double* ob = odd ? oa : oa + H;
  const double scale = 1.0 / static_cast<double>(n);
  for (int j = threadIdx.x; j < n; j += blockDim.x) {  // x_j = v_(pos(j)),  v = conj(c) * convolution / n
    const int v = makhoul_pos(j, n);
    const c64 z = cmulc(zs[swz(v)], t.chirp[v]);
    oa[j] = z.x * scale;
    ob[j] = z.y * scale;
  }
}

// (Round 3 also built the inverse transform with ONE real column per workgroup -- an n/2-point complex FFT of
// z[m] = v[2m] + i v[2m+1], half the LDS, four workgroups per CU instead of two: correct, and 9 us SLOWER per launch at
// 4096^2 (74 against 65 us): the transform is bound by its own instruction stream, not by the overlap of its phases.
// Removed; the commit that carried it: "2-D TV row stage: 16 waves of a workgroup ...".)

// columns of t have length W (= tw.n) and belong to the row frequencies i = 2*blockIdx.x, +1 of the image
__global__ __launch_bounds__(kBlock) void dct_rows_solve_kernel(double* __restrict__ tm, int64_t W, double rho,
                                                                const double* __restrict__ lamH, DctTables t,
                                                                const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  extern __shared__ c64 zs[];
  const int n = t.n, p = t.log2n;
  double* a = tm + static_cast<int64_t>(2 * blockIdx.x) * W;
  double* b = a + W;
  load_pair(zs, a, b, n);
  __syncthreads();
  fft_network<false>(zs, n, p, t.tw);
  const double la = lamH[2 * blockIdx.x], lb = lamH[2 * blockIdx.x + 1];
  const double* __restrict__ lam = t.lam;
#pragma unroll 2
  for (int k = threadIdx.x; k <= (n >> 1); k += blockDim.x) {
    if (k == 0) {
      const c64 z0 = zs[0];
      zs[0] = c64{z0.x / (1.0 + rho * la), z0.y / (1.0 + rho * lb)};  // lam[0] = 0
    } else if (k == (n >> 1)) {
      const c64 h = zs[1];  // X = sqrt(1/2) V and V' = sqrt(2) X': the rotation cancels
      zs[1] = c64{h.x / (1.0 + rho * (la + lam[k])), h.y / (1.0 + rho * (lb + lam[k]))};
    } else {
      const int rk = swz(bitrev(k, p)), rn = swz(bitrev(n - k, p));
      const c64 ck = t.c4[k];
      double xak, xan, xbk, xbn;
      spectrum_to_dct(zs[rk], zs[rn], ck, xak, xan, xbk, xbn);
      const double lk = lam[k], ln = lam[n - k];
      xak /= 1.0 + rho * (la + lk);
      xan /= 1.0 + rho * (la + ln);
      xbk /= 1.0 + rho * (lb + lk);
      xbn /= 1.0 + rho * (lb + ln);
      c64 zk, zn;
      dct_to_spectrum(xak, xan, xbk, xbn, ck, zk, zn);
      zs[rk] = zk;
      zs[rn] = zn;
    }
  }
  __syncthreads();
  fft_network<true>(zs, n, p, t.tw);
  store_pair(zs, a, b, n, 1.0 / static_cast<double>(n));
}

// 64 x 64 tiles through LDS (65-double pitch: conflict-free both ways); fully coalesced on both sides
// The same solve on the image where it lies (H x W, column-major): the workgroup of row pair (i, i+1) reads, for every
// column j, the 16 bytes img[i..i+1, j] -- already the complex pair (a_j, b_j) the joint FFT wants -- at stride H,
// and writes them back the same way: no transposed copy of the image, two passes over it instead of six.  The access
// is 16 bytes per 8H-byte stride; what keeps it off HBM is that the 8 workgroups sharing a 128-byte line run at the
// same time on ONE XCD (row pairs are dealt to the XCDs in contiguous ranges: workgroup b -> XCD b mod 8 under
// round-robin placement, speed only) and that the image (134 MB at 4096^2) sits in the 256 MB Infinity Cache right
// after the column pass wrote it.
__global__ __launch_bounds__(kBlock) void dct_rows_solve_strided_kernel(double* __restrict__ img, int64_t H, double rho,
                                                                        const double* __restrict__ lamH, DctTables t,
                                                                        const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  extern __shared__ c64 zs[];
  const int n = t.n, p = t.log2n;
  const unsigned npairs = gridDim.x;
  // contiguous ranges of row pairs per XCD (where the pairs divide by 8; otherwise in order)
  const unsigned b = blockIdx.x;
  const unsigned pair = (npairs >= 8 && (npairs & 7u) == 0) ? (b & 7u) * (npairs >> 3) + (b >> 3) : b;
  double* __restrict__ base = img + 2 * static_cast<int64_t>(pair);
  // Makhoul order: x[2k] -> v[k], x[2k+1] -> v[n-1-k]; all loads of a thread are independent
#pragma unroll 8
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const admm_double2 v = load2<false>(base + static_cast<int64_t>(j) * H);
    const int k = j >> 1;
    zs[swz((j & 1) ? n - 1 - k : k)] = c64{v.x, v.y};
  }
  __syncthreads();
  fft_network<false>(zs, n, p, t.tw);
  const double la = lamH[2 * pair], lb = lamH[2 * pair + 1];
  const double* __restrict__ lam = t.lam;
#pragma unroll 2
  for (int k = threadIdx.x; k <= (n >> 1); k += blockDim.x) {
    if (k == 0) {
      const c64 z0 = zs[0];
      zs[0] = c64{z0.x / (1.0 + rho * la), z0.y / (1.0 + rho * lb)};  // lam[0] = 0
    } else if (k == (n >> 1)) {
      const c64 h = zs[1];
      zs[1] = c64{h.x / (1.0 + rho * (la + lam[k])), h.y / (1.0 + rho * (lb + lam[k]))};
    } else {
      const int rk = swz(bitrev(k, p)), rn = swz(bitrev(n - k, p));
      const c64 ck = t.c4[k];
      double xak, xan, xbk, xbn;
      spectrum_to_dct(zs[rk], zs[rn], ck, xak, xan, xbk, xbn);
      const double lk = lam[k], ln = lam[n - k];
      xak /= 1.0 + rho * (la + lk);
      xan /= 1.0 + rho * (la + ln);
      xbk /= 1.0 + rho * (lb + lk);
      xbn /= 1.0 + rho * (lb + ln);
      c64 zk, zn;
      dct_to_spectrum(xak, xan, xbk, xbn, ck, zk, zn);
      zs[rk] = zk;
      zs[rn] = zn;
    }
  }
  __syncthreads();
  fft_network<true>(zs, n, p, t.tw);
  const double scale = 1.0 / static_cast<double>(n);
#pragma unroll 8
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const int k = j >> 1;
    const c64 v = zs[swz((j & 1) ? n - 1 - k : k)];
    store2<false>(base + static_cast<int64_t>(j) * H, admm_double2{v.x * scale, v.y * scale});
  }
}

__global__ __launch_bounds__(kBlock) void transpose_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                           int64_t rows, int64_t cols, const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stop) return;
  __shared__ double tile[64][65];
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 64, c0 = static_cast<int64_t>(blockIdx.y) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < 64; c += 4) {
    const int64_t r = r0 + tx, cc = c0 + c;
    if (r < rows && cc < cols) tile[c][tx] = src[r + cc * rows];
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int64_t cc = c0 + tx, rr = r0 + r;
    if (cc < cols && rr < rows) dst[cc + rr * cols] = tile[tx][r];
  }
}

bool dct_length_ok(int64_t n) { return n >= 8 && n <= 8192 && (n & (n - 1)) == 0; }

void dct_fill_tables(int32_t n, admm_double2* tw, admm_double2* c4, double* lam) {
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int32_t k = 0; k < n / 2; ++k) {
    const long double ang = -2.0L * pi * k / n;
    tw[k] = admm_double2{static_cast<double>(cosl(ang)), static_cast<double>(sinl(ang))};
  }
  for (int32_t k = 0; k <= n / 2; ++k) {
    const long double ang = -pi * k / (2.0L * n);
    c4[k] = admm_double2{static_cast<double>(cosl(ang)), static_cast<double>(sinl(ang))};
  }
  for (int32_t k = 0; k < n; ++k) {
    const long double sv = sinl(pi * k / (2.0L * n));
    lam[k] = static_cast<double>(4.0L * sv * sv);
  }
}

bool dct_chirp_length_ok(int64_t n) { return n >= 8 && n <= 4096 && !dct_length_ok(n); }

int32_t dct_chirp_fft_length(int64_t n) {
  int32_t m = 16;
  while (m < 2 * n - 1) m <<= 1;
  return m;
}

void dct_fill_chirp_tables(int32_t n, admm_double2* tw, admm_double2* c4, double* lam, admm_double2* chirp,
                           admm_double2* hbr) {
  const long double pi = 3.141592653589793238462643383279502884L;
  const int32_t M = dct_chirp_fft_length(n);
  int log2m = 0;
  while ((1 << log2m) < M) ++log2m;
  for (int32_t k = 0; k < M / 2; ++k) {
    const long double ang = -2.0L * pi * k / M;
    tw[k] = admm_double2{static_cast<double>(cosl(ang)), static_cast<double>(sinl(ang))};
  }
  for (int32_t k = 0; k < n; ++k) {
    const long double ang = -pi * k / (2.0L * n);
    c4[k] = admm_double2{static_cast<double>(cosl(ang)), static_cast<double>(sinl(ang))};
    const long double sv = sinl(pi * k / (2.0L * n));
    lam[k] = static_cast<double>(4.0L * sv * sv);
  }
  // c_j = e^{-pi i j^2/n}: j^2 reduced modulo 2n exactly before the angle is formed
  std::vector<long double> hr(static_cast<size_t>(M), 0.0L), hi(static_cast<size_t>(M), 0.0L);
  for (int32_t j = 0; j < n; ++j) {
    const int64_t q = (static_cast<int64_t>(j) * j) % (2 * static_cast<int64_t>(n));
    const long double ang = -pi * static_cast<long double>(q) / n;
    const long double cr = cosl(ang), ci = sinl(ang);
    chirp[j] = admm_double2{static_cast<double>(cr), static_cast<double>(ci)};
    hr[j] = cr;  // h_m = conj(c_|m|) at m and at M - m
    hi[j] = -ci;
    if (j > 0) {
      hr[M - j] = cr;
      hi[M - j] = -ci;
    }
  }
  // H = FFT_M(h): iterative radix-2 in long double (bit-reversal permutation first), then H[bitrev(p)] / M at position p
  auto brev = [&](int32_t k) {
    int32_t r = 0;
    for (int b = 0; b < log2m; ++b) r |= ((k >> b) & 1) << (log2m - 1 - b);
    return r;
  };
  for (int32_t k = 0; k < M; ++k) {
    const int32_t r = brev(k);
    if (r > k) {
      std::swap(hr[k], hr[r]);
      std::swap(hi[k], hi[r]);
    }
  }
  for (int32_t len = 2; len <= M; len <<= 1) {
    const long double ang = -2.0L * pi / len;
    for (int32_t i0 = 0; i0 < M; i0 += len)
      for (int32_t j = 0; j < len / 2; ++j) {
        const long double wr = cosl(ang * j), wi = sinl(ang * j);
        const int32_t p = i0 + j, q = p + len / 2;
        const long double tr = hr[q] * wr - hi[q] * wi, ti = hr[q] * wi + hi[q] * wr;
        hr[q] = hr[p] - tr;
        hi[q] = hi[p] - ti;
        hr[p] += tr;
        hi[p] += ti;
      }
  }
  for (int32_t p = 0; p < M; ++p) {
    const int32_t k = brev(p);
    hbr[p] = admm_double2{static_cast<double>(hr[k] / M), static_cast<double>(hi[k] / M)};
  }
}

static size_t dct_lds_bytes(int n) { return sizeof(c64) * static_cast<size_t>(n); }
static unsigned col_pairs(int64_t W) { return static_cast<unsigned>((W + 1) / 2); }
static DctTables with_pairs(const DctTables& th, int64_t W) {  // the tables as a launch over the columns of a W-wide image sees them
  DctTables t = th;
  t.odd_pair = (W & 1) ? static_cast<int32_t>(W / 2) : -1;
  return t;
}

// n = 8192 needs 128 KB of the CU's 160 KB LDS (one workgroup per CU): beyond the 64 KB a launch gets by default
template <typename K>
static void dct_allow_lds(K kernel, size_t bytes) {
  if (bytes > (64u << 10)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

void launch_dct_cols_forward(double* img, int64_t H, int64_t W, const DctTables& th, const Ctrl* ctrl,
                             hipStream_t stream) {
  const DctTables t2 = with_pairs(th, W);
  if (th.bm) {
    dct_allow_lds(dct_cols_forward_chirp_kernel<false>, dct_lds_bytes(th.bm));
    hipLaunchKernelGGL(dct_cols_forward_chirp_kernel<false>, dim3(col_pairs(W)), dim3(chirp_threads(th.bm)),
                       dct_lds_bytes(th.bm), stream, img, H, t2, FinArgs{}, 0, ctrl);
    return;
  }
  dct_allow_lds(dct_cols_forward_kernel, dct_lds_bytes(th.n));
  hipLaunchKernelGGL(dct_cols_forward_kernel, dim3(col_pairs(W)), dim3(kBlock), dct_lds_bytes(th.n), stream, img, H, t2,
                     ctrl);
}

void launch_tv2d_fused_dct(const Tv2Args& a, bool state_in, double* bhat, const DctTables& th, const Ctrl* ctrl,
                           int* nblk_out, hipStream_t stream) {
  int64_t nb = (a.W + 1) / 2;
  if (nb > kMaxPartBlocks) nb = kMaxPartBlocks;  // (a workgroup then walks several pairs: one set of partials each)
  *nblk_out = static_cast<int>(nb);
  const size_t lds = dct_lds_bytes(th.n);
  auto go = [&](auto kernel) {
    dct_allow_lds(kernel, lds);
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(nb)), dim3(kBlock), lds, stream, a, bhat, th, ctrl);
  };
  const bool odd = (a.W & 1) != 0;
  if (state_in) odd ? go(tv2d_fused_dct_kernel<true, true>) : go(tv2d_fused_dct_kernel<true, false>);
  else odd ? go(tv2d_fused_dct_kernel<false, true>) : go(tv2d_fused_dct_kernel<false, false>);
}

void launch_dct_cols_forward_fin(double* img, int64_t H, int64_t W, const DctTables& th, const FinArgs& f,
                                 bool fin_pending, const Ctrl* ctrl, hipStream_t stream) {
  const DctTables t2 = with_pairs(th, W);
  if (th.bm) {
    dct_allow_lds(dct_cols_forward_chirp_kernel<true>, dct_lds_bytes(th.bm));
    hipLaunchKernelGGL(dct_cols_forward_chirp_kernel<true>, dim3(col_pairs(W) + 1u), dim3(chirp_threads(th.bm)),
                       dct_lds_bytes(th.bm), stream, img, H, t2, f, fin_pending ? 1 : 0, ctrl);
    return;
  }
  dct_allow_lds(dct_cols_forward_fin_kernel, dct_lds_bytes(th.n));
  hipLaunchKernelGGL(dct_cols_forward_fin_kernel, dim3(col_pairs(W) + 1u), dim3(kBlock), dct_lds_bytes(th.n), stream, img, H,
                     t2, f, fin_pending ? 1 : 0, ctrl);
}

void launch_dct_cols_inverse(const double* src, double* dst, int64_t H, int64_t W, const DctTables& th,
                             const Ctrl* ctrl, hipStream_t stream) {
  const DctTables t2 = with_pairs(th, W);
  if (th.bm) {
    dct_allow_lds(dct_cols_inverse_chirp_kernel, dct_lds_bytes(th.bm));
    hipLaunchKernelGGL(dct_cols_inverse_chirp_kernel, dim3(col_pairs(W)), dim3(chirp_threads(th.bm)), dct_lds_bytes(th.bm),
                       stream, src, dst, H, t2, ctrl);
    return;
  }
  dct_allow_lds(dct_cols_inverse_kernel, dct_lds_bytes(th.n));
  hipLaunchKernelGGL(dct_cols_inverse_kernel, dim3(col_pairs(W)), dim3(kBlock), dct_lds_bytes(th.n), stream, src, dst, H, t2,
                     ctrl);
}

void launch_dct_rows_solve(double* t, int64_t H, int64_t W, double rho, const DctTables& th, const DctTables& tw,
                           const Ctrl* ctrl, hipStream_t stream) {
  dct_allow_lds(dct_rows_solve_kernel, dct_lds_bytes(tw.n));
  hipLaunchKernelGGL(dct_rows_solve_kernel, dim3(static_cast<unsigned>(H / 2)), dim3(kBlock), dct_lds_bytes(tw.n),
                     stream, t, W, rho, th.lam, tw, ctrl);
}

void launch_dct_rows_solve_strided(double* img, int64_t H, int64_t W, double rho, const DctTables& th,
                                   const DctTables& tw, const Ctrl* ctrl, hipStream_t stream) {
  (void)W;
  dct_allow_lds(dct_rows_solve_strided_kernel, dct_lds_bytes(tw.n));
  hipLaunchKernelGGL(dct_rows_solve_strided_kernel, dim3(static_cast<unsigned>(H / 2)), dim3(kBlock),
                     dct_lds_bytes(tw.n), stream, img, H, rho, th.lam, tw, ctrl);
}

// ---------------------------------------------------------------- the row stage without a transform
// After the column DCT, row i of the image (vertical frequency i) has to be multiplied by
//   inv( (1 + rho*lamH[i]) I + rho*L_W ),   L_W = the 1-D Neumann Laplacian along the row.
// That matrix is tridiag(-rho, d_i, -rho), d_i = 1 + rho*(lamH[i] + 2), with the two end entries d_i - rho: by the
// method of images (half-sample even extension -- the symmetry of the DCT-II) its inverse is EXACTLY the Toeplitz
// kernel applied to the mirrored row:  x_j = A_i * sum_m r_i^|m| b_mirror(j+m),  r_i = 2 rho / (d_i + sqrt(d_i^2 - 4 rho^2)),
// A_i = 1/sqrt(d_i^2 - 4 rho^2); r_i <= r_0 < 1, so |m| <= K (r_0^K < 1e-18) is the whole sum to rounding.
// Lanes run along i (contiguous in the column-major image: every load and store of a wave is one 512-byte run), a
// wave owns a run of kRowRun consecutive columns j: one K-term Horner sum on each side, then the two first-order
// recurrences across the run -- no LDS, no cross-lane traffic, no 16-byte strided accesses (the strided row DCT moved
// 2N doubles in 137 us at 4096^2; this moves the same 2N in coalesced runs).  src -> dst, not in place.
constexpr int kRowRun = 32;   // columns whose causal sums a lane keeps in registers
constexpr int kRowRuns = 4;   // consecutive runs per wave: the causal recurrence carries over, and the K columns a run
                              // looks ahead are the next run's own columns (cache hits): ~1.7x instead of 3.6x reads

// FIN: one more column of workgroups in front (blockIdx.x == 0); its first one is a passenger that runs the finalize
// logic of the PREVIOUS iteration while the others work (as dct_cols_forward_fin_kernel does when the iteration begins
// with the column transform).  Nothing in this launch depends on that decision, and dst is a scratch image here: the
// launches behind it -- the inverse transform into x, the fused pass -- no-op when it has raised ctrl->stop.
template <bool FIN>
__global__ __launch_bounds__(kBlock, 2) void tv2d_rows_green_kernel(const double* __restrict__ src,
                                                                    double* __restrict__ dst, int64_t H, int64_t W,
                                                                    double rho, const double* __restrict__ lamH, int K,
                                                                    FinArgs f, int32_t fin_pending,
                                                                    const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  if (FIN && blockIdx.x == 0) {
    if (blockIdx.y == 0 && fin_pending) finalize_body<false>(f);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = static_cast<int64_t>(blockIdx.x - (FIN ? 1u : 0u)) * 64 + lane;
  const int64_t jseg = (static_cast<int64_t>(blockIdx.y) * 4 + wave) * (kRowRun * kRowRuns);
  if (jseg >= W) return;  // wave-uniform
  const int64_t ic = i < H ? i : H - 1;
  const double d = 1.0 + rho * (lamH[ic] + 2.0);
  const double disc = sqrt(d * d - 4.0 * rho * rho);
  const double r = 2.0 * rho / (d + disc), A = 1.0 / disc;
  // the wave's own truncation, from the largest ratio among its 64 rows: K (the host's bound, from r_0) is needed by
  // the lowest vertical frequencies only -- at lambda = 4, rho = 1, 22 terms do what 44 do at lambda = 0
  {
    double rmax = r;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, off, 64));
    const int kw = static_cast<int>(-41.4465316738928 / log(rmax)) + 2;  // ln(1e-18)
    K = __builtin_amdgcn_readfirstlane(kw < K ? kw : K);
  }
  const double* __restrict__ row = src + ic;
  auto at = [&](int64_t j) -> double {  // mirrored column index (K < W)
    const int64_t jm = j < 0 ? -1 - j : (j >= W ? 2 * W - 1 - j : j);
    return row[jm * H];
  };
  // causal sum just left of the segment: cprev = sum_{m=0..K} r^m b(jseg - 1 - m)   (then c_j = r*c_(j-1) + b_j)
  // Loads are unconditional (taps past the sum re-read tap 0 / the run's last column) and a group of kTapGroup is issued
  // before any is used: written as `cond ? load : 0` hipcc puts every load into its own basic block behind a scalar
  // branch and waits per group of 8 -- six dependent round trips per side and run instead of three.
  constexpr int kTapGroup = 16;
  double cprev = 0.0;
  for (int m0 = K; m0 >= 0; m0 -= kTapGroup) {
    double v[kTapGroup];
#pragma unroll
    for (int q = 0; q < kTapGroup; ++q) v[q] = at(jseg - 1 - (m0 - q >= 0 ? m0 - q : 0));
#pragma unroll
    for (int q = 0; q < kTapGroup; ++q) {
      const double c2 = __builtin_fma(r, cprev, v[q]);
      cprev = (m0 - q >= 0) ? c2 : cprev;
    }
  }
#pragma unroll 1
  for (int run = 0; run < kRowRuns; ++run) {
    const int64_t j0 = jseg + static_cast<int64_t>(run) * kRowRun;
    if (j0 >= W) break;  // wave-uniform
    const int nrun = (W - j0 < kRowRun) ? static_cast<int>(W - j0) : kRowRun;
    // anticausal sum at the last column of the run: a = sum_{m=1..K} r^m b(jl + m)
    const int64_t jl = j0 + nrun - 1;
    double bs[kRowRun], cs[kRowRun];
#pragma unroll
    for (int t = 0; t < kRowRun; ++t) {  // the run's own columns first: in flight together with the first tap group
      const double val = row[(t < nrun ? j0 + t : jl) * H];
      bs[t] = (t < nrun) ? val : 0.0;
    }
    double ac = 0.0;
    for (int m0 = K; m0 >= 1; m0 -= kTapGroup) {
      double v[kTapGroup];
#pragma unroll
      for (int q = 0; q < kTapGroup; ++q) v[q] = at(jl + (m0 - q >= 1 ? m0 - q : 1));
#pragma unroll
      for (int q = 0; q < kTapGroup; ++q) {
        const double a2 = __builtin_fma(r, ac, v[q]);
        ac = (m0 - q >= 1) ? a2 : ac;
      }
    }
    ac *= r;
    cs[0] = __builtin_fma(r, cprev, bs[0]);
#pragma unroll
    for (int t = 1; t < kRowRun; ++t) cs[t] = __builtin_fma(r, cs[t - 1], bs[t]);
    cprev = cs[kRowRun - 1];  // (a short run is the last one of the row: never used again)
#pragma unroll
    for (int t = kRowRun - 1; t >= 0; --t) {
      if (t < nrun) {
        if (i < H) dst[i + (j0 + t) * H] = A * (cs[t] + ac);
        ac = r * (ac + bs[t]);
      }
    }
  }
}

// The same row stage with the waves of a workgroup working together (default when the truncation fits 64 columns).
// In the kernel above a wave pays for its two carries with K-term Horner sums over its neighbours' columns: 2.1 x the
// image in reads (PMC).  Here a workgroup is 16 waves on the SAME 64 rows, wave q holding the 32-column run q in
// registers from its loads to its stores:
//   local sums   S_q = sum_t r^(31-t) b_t   (what the run hands to the right),  P_q = sum_t r^t b_t   (... to the left)
//   carries      c(j_q0 - 1) = S_(q-1) + r^32 S_(q-2),   a(j_q31) = r (P_(q+1) + r^32 P_(q+2))      (r^64 < 1e-18)
// through 16 KB of LDS and one barrier; waves 0, 1 and 14, 15 hold the halo runs left and right of the 384 output
// columns (mirror images at the ends of the row; columns farther than the block's own truncation K are not loaded),
// so the halo costs 2K / 384 of the image instead of 2K / 128 + 32/32.  The second pass runs the two first-order
// recurrences from the exact carries; c_t overwrites b_t and the anticausal pass takes b_(t+1) = c_(t+1) - r c_t
// (one rounding of size eps |b| / (1 - r): a second 32-double array would halve the occupancy).
constexpr int kCoopRun = 32, kCoopWaves = 16, kCoopHalo = 2, kCoopOut = kCoopWaves - 2 * kCoopHalo;

template <bool FIN>
__global__ __launch_bounds__(kCoopWaves* kWave) void tv2d_rows_coop_kernel(const double* __restrict__ src,
                                                                          double* __restrict__ dst, int64_t H, int64_t W,
                                                                          double rho, const double* __restrict__ lamH,
                                                                          int K, FinArgs f, int32_t fin_pending,
                                                                          const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  if (FIN && blockIdx.x == 0) {
    if (blockIdx.y == 0 && fin_pending && threadIdx.x < kBlock) finalize_body<false>(f);
    return;
  }
  __shared__ double Ssh[kCoopWaves][kWave], Psh[kCoopWaves][kWave];
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));  // (scalar: the column addresses are)
  const int64_t i = static_cast<int64_t>(blockIdx.x - (FIN ? 1u : 0u)) * 64 + lane;
  const int64_t ic = i < H ? i : H - 1;
  const double d = 1.0 + rho * (lamH[ic] + 2.0);
  const double disc = sqrt(d * d - 4.0 * rho * rho);
  const double r = 2.0 * rho / (d + disc), A = 1.0 / disc;
  {  // the block's own truncation, from the largest ratio among its 64 rows (K: the host's bound, from r_0)
    double rmax = r;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, off, 64));
    const int kw = static_cast<int>(-41.4465316738928 / log(rmax)) + 2;  // ln(1e-18)
    K = __builtin_amdgcn_readfirstlane(kw < K ? kw : K);
  }
  const int64_t Jout = static_cast<int64_t>(blockIdx.y) * (kCoopOut * kCoopRun);  // first output column of the workgroup
  const int64_t Jend = Jout + kCoopOut * kCoopRun;                                 // one past its last
  const int64_t j0 = Jout + static_cast<int64_t>(q - kCoopHalo) * kCoopRun;        // this wave's first column (may be < 0)
  const double* __restrict__ row = src + ic;
  double bs[kCoopRun];
#pragma unroll
  for (int t = 0; t < kCoopRun; ++t) {
    int64_t j = j0 + t;
    // halo columns beyond the truncation are not needed: re-read the nearest needed one (a cache hit), count as zero
    bool need = true;
    if (j < Jout - K) {
      j = Jout - K;
      need = false;
    }
    if (j >= Jend + K) {
      j = Jend + K - 1;
      need = false;
    }
    if (j >= 2 * W) {  // (beyond the mirror image of the row: only in halo runs right of a short last block)
      j = 2 * W - 1;
      need = false;
    }
    const int64_t jm = j < 0 ? -1 - j : (j >= W ? 2 * W - 1 - j : j);  // mirrored column (K < W)
    const double v = row[jm * H];
    bs[t] = need ? v : 0.0;
  }
  double S = 0.0, P = 0.0;
#pragma unroll
  for (int t = 0; t < kCoopRun; ++t) S = __builtin_fma(r, S, bs[t]);
#pragma unroll
  for (int t = kCoopRun - 1; t >= 0; --t) P = __builtin_fma(r, P, bs[t]);
  Ssh[q][lane] = S;
  Psh[q][lane] = P;
  __syncthreads();
  if (q < kCoopHalo || q >= kCoopWaves - kCoopHalo || j0 >= W) return;  // halo runs; runs right of the image
  double r32 = r * r;  // r^2
  r32 *= r32;          // r^4
  r32 *= r32;          // r^8
  r32 *= r32;          // r^16
  r32 *= r32;          // r^32
  double c = __builtin_fma(r32, Ssh[q - 2][lane], Ssh[q - 1][lane]);
  double a = r * __builtin_fma(r32, Psh[q + 2][lane], Psh[q + 1][lane]);
#pragma unroll
  for (int t = 0; t < kCoopRun; ++t) {
    c = __builtin_fma(r, c, bs[t]);
    bs[t] = c;
  }
  const bool rowok = i < H;
#pragma unroll
  for (int t = kCoopRun - 1; t >= 0; --t) {
    if (rowok && j0 + t < W) dst[i + (j0 + t) * H] = A * (bs[t] + a);
    if (t > 0) a = r * (a + (bs[t] - r * bs[t - 1]));
  }
}

// the cooperative form covers truncations up to its two halo runs
static bool tv2d_rows_coop_ok(int taps, int64_t W) {
  return taps <= kCoopHalo * kCoopRun && W >= kCoopOut * kCoopRun && std::getenv("ADMM_HIP_TV2D_ROWS_WAVE") == nullptr;
}

int tv2d_rows_green_taps(double rho) {
  const double d0 = 1.0 + 2.0 * rho;
  const double r0 = 2.0 * rho / (d0 + std::sqrt(d0 * d0 - 4.0 * rho * rho));
  return static_cast<int>(std::ceil(std::log(1e-18) / std::log(r0))) + 1;
}

void launch_tv2d_rows_green(const double* src, double* dst, int64_t H, int64_t W, double rho, const DctTables& th,
                            const Ctrl* ctrl, hipStream_t stream, const FinArgs* fin, bool fin_pending) {
  const int taps = tv2d_rows_green_taps(rho);
  if (tv2d_rows_coop_ok(taps, W)) {
    const unsigned cx = static_cast<unsigned>(ceil_div(H, int64_t{64})), cy = static_cast<unsigned>(ceil_div(W, int64_t{kCoopOut * kCoopRun}));
    if (fin)
      hipLaunchKernelGGL(tv2d_rows_coop_kernel<true>, dim3(cx + 1u, cy), dim3(kCoopWaves * kWave), 0, stream, src, dst, H, W,
                         rho, th.lam, taps, *fin, fin_pending ? 1 : 0, ctrl);
    else
      hipLaunchKernelGGL(tv2d_rows_coop_kernel<false>, dim3(cx, cy), dim3(kCoopWaves * kWave), 0, stream, src, dst, H, W, rho,
                         th.lam, taps, FinArgs{}, 0, ctrl);
    return;
  }
  const unsigned gx = static_cast<unsigned>(ceil_div(H, int64_t{64})), gy = static_cast<unsigned>(ceil_div(W, int64_t{4 * kRowRun * kRowRuns}));
  if (fin)
    hipLaunchKernelGGL(tv2d_rows_green_kernel<true>, dim3(gx + 1u, gy), dim3(kBlock), 0, stream, src, dst, H, W, rho, th.lam,
                       tv2d_rows_green_taps(rho), *fin, fin_pending ? 1 : 0, ctrl);
  else
    hipLaunchKernelGGL(tv2d_rows_green_kernel<false>, dim3(gx, gy), dim3(kBlock), 0, stream, src, dst, H, W, rho, th.lam,
                       tv2d_rows_green_taps(rho), FinArgs{}, 0, ctrl);
}

// ---------------------------------------------------------------- the row stage as an exact tridiagonal solve
// Where the Toeplitz form does not apply -- a width below four times its tap count, or a rho whose kernel decays over
// thousands of columns -- and the width has no row transform either, the row systems tridiag(-rho, d_i, -rho) (end
// entries d_i - rho) are solved as they stand: Thomas elimination along the row, one LANE per row i (contiguous in the
// column-major image: every access of a wave is one 512-byte run), the W steps of a row sequential.  The elimination
// factors depend on (i, j) and rho only: a setup launch per run stores c'_j and 1/denominator_j (two images of scratch);
// the solve is then two fused-multiply-add recurrences, kThomasChunk columns of loads in flight before each stretch of
// the chain.  H lanes is all the parallelism there is (64 waves at H = 4096), so this runs at the latency of its chain,
// not at bandwidth -- against the thousands of CG steps it replaces at large rho (cond(I + rho*D'D) ~ 8 rho) it is exact
// and one to two orders of magnitude faster.  src -> dst (may alias).
constexpr int kThomasChunk = 16;

__global__ __launch_bounds__(kWave) void tv2d_rows_thomas_setup_kernel(int64_t H, int64_t W, double rho,
                                                                       const double* __restrict__ lamH,
                                                                       double* __restrict__ cp, double* __restrict__ inv) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kWave + threadIdx.x;
  if (i >= H) return;
  const double d = 1.0 + rho * (lamH[i] + 2.0);
  double c = 0.0;  // c'_(j-1)
  for (int64_t j = 0; j < W; ++j) {
    const double diag = d - (j == 0 ? rho : 0.0) - (j == W - 1 ? rho : 0.0);
    const double r = 1.0 / (diag + rho * c);  // denominator of row j: diag - (-rho)*c'_(j-1)
    c = -rho * r;
    inv[i + j * H] = r;
    cp[i + j * H] = c;
  }
}

__global__ __launch_bounds__(kWave) void tv2d_rows_thomas_kernel(const double* src, double* dst,
                                                                 int64_t H, int64_t W, double rho,
                                                                 const double* __restrict__ cp,
                                                                 const double* __restrict__ inv,
                                                                 const Ctrl* __restrict__ ctrl) {
  if (ctrl->stop) return;
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * kWave + threadIdx.x;
  const int64_t i = i0 < H ? i0 : H - 1;  // (clamped lanes compute row H-1 again and store nothing)
  const bool live = i0 < H;
  double carry = 0.0;  // d'_(j-1)
  for (int64_t j0 = 0; j0 < W; j0 += kThomasChunk) {  // forward: d'_j = (b_j + rho*d'_(j-1)) / denominator_j
    double b[kThomasChunk], r[kThomasChunk];
#pragma unroll
    for (int k = 0; k < kThomasChunk; ++k) {
      const int64_t j = j0 + k < W ? j0 + k : W - 1;
      b[k] = src[i + j * H];
      r[k] = inv[i + j * H];
    }
#pragma unroll
    for (int k = 0; k < kThomasChunk; ++k) {
      if (j0 + k < W) {
        carry = __builtin_fma(rho, carry, b[k]) * r[k];
        if (live) dst[i + (j0 + k) * H] = carry;
      }
    }
  }
  double x = 0.0;  // x_(j+1); c'_(W-1) multiplies nothing: the last row has no super-diagonal
  for (int64_t j1 = W; j1 > 0; j1 -= kThomasChunk) {  // backward: x_j = d'_j - c'_j x_(j+1)
    double dpv[kThomasChunk], c[kThomasChunk];
#pragma unroll
    for (int k = 0; k < kThomasChunk; ++k) {
      const int64_t j = j1 - 1 - k >= 0 ? j1 - 1 - k : 0;
      dpv[k] = dst[i + j * H];
      c[k] = cp[i + j * H];
    }
#pragma unroll
    for (int k = 0; k < kThomasChunk; ++k) {
      const int64_t j = j1 - 1 - k;
      if (j >= 0) {
        x = (j == W - 1) ? dpv[k] : __builtin_fma(-c[k], x, dpv[k]);
        if (live) dst[i + j * H] = x;
      }
    }
  }
}

void launch_tv2d_rows_thomas_setup(int64_t H, int64_t W, double rho, const DctTables& th, double* cp, double* inv,
                                   hipStream_t stream) {
  hipLaunchKernelGGL(tv2d_rows_thomas_setup_kernel, dim3(static_cast<unsigned>(ceil_div(H, int64_t{kWave}))), dim3(kWave), 0,
                     stream, H, W, rho, th.lam, cp, inv);
}

void launch_tv2d_rows_thomas(const double* src, double* dst, int64_t H, int64_t W, double rho, const double* cp,
                             const double* inv, const Ctrl* ctrl, hipStream_t stream) {
  hipLaunchKernelGGL(tv2d_rows_thomas_kernel, dim3(static_cast<unsigned>(ceil_div(H, int64_t{kWave}))), dim3(kWave), 0,
                     stream, src, dst, H, W, rho, cp, inv, ctrl);
}

void launch_transpose(const double* src, double* dst, int64_t rows, int64_t cols, const Ctrl* ctrl,
                      hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>(ceil_div(rows, 64)), static_cast<unsigned>(ceil_div(cols, 64)));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(kBlock), 0, stream, src, dst, rows, cols, ctrl);
}

}  // namespace admm
