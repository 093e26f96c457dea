// tv.h -- argument block and launchers of the 1-D total-variation kernels (tv.hip).
#pragma once
#include <vector>

#include "common.h"

namespace admm {

struct TvArgs {
  int64_t n;
  double rho, thresh;      // thresh = lambda/rho   (getProxOps.m:199)
  const double* s;         // noisy signal
  const double* z;         // current z, u (read)
  const double* u;
  double* zo;              // next z, u (written by the prox kernel; ping-pong)
  double* uo;
  double* x;
  double* y;               // forward-sweep intermediate
  const double* bprefix;   // leading LDL' pivots until they become stationary
  int64_t nprefix;
  double bstar;            // stationary pivot
  int32_t halo, elems, tile;
  int32_t objevals;
  double* xhist;
  double* zhist;
  double* uhist;
  double* part;            // [S_COUNT][kMaxPartBlocks]
};

// pivots of I + rho*D'D and the launch geometry for a given rho
int tv_plan(double rho, int64_t n, std::vector<double>* prefix, double* bstar, int* halo, int* elems, int* tile);
void launch_tv_sweep(const TvArgs& a, bool backward, const Ctrl* ctrl, hipStream_t stream);
void launch_tv_prox(const TvArgs& a, const Ctrl* ctrl, int* nblk_out, hipStream_t stream);

}  // namespace admm
