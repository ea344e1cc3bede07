// Bounded L-BFGS of the joint fit with its vector arithmetic on the device (kernel K10 of SURVEY.md 8(a); `north_star`:
// "L-BFGS parameter updates fused on-device").  Replaces what scipy's L-BFGS-B does on the host for
// STARRED's Optimizer(method='l-bfgs-b') (reference call sites: lightcurver/processes/roi_modelling.py:278-280 - the
// translations / fluxes stage of the ROI fit - and utilities/starred_utilities.py:33-34): the free parameters, the
// gradient, the search direction and the (s, y) history never leave the device; per trial point the host reads back three
// scalars (loss, g . step, and after an accepted step s . y with the two norms) and decides.  The algorithm is the
// projected L-BFGS of csrc/lbfgs_host.h (two-loop recursion on the variables that are not held by a bound, Armijo
// backtracking on the projected step); iterate-level parity with scipy is not a goal, the optimum is.
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int kLbThreads = 1024;
constexpr int kLbMem = 10;

struct LbfgsDev {
  int D;
  float *x, *g, *xt, *gt, *dir;   // [D]
  const float *lo, *hi;           // [D]
  float *S, *Y;                   // [kLbMem][D] ring buffers
  float *rho;                     // [kLbMem]
  float *scal;                    // [8] results for the host
};

__device__ inline float lb_block_sum(float v, float *sh) {
  // fixed-order reduction: lanes by shuffles, then the waves in order
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < kLbThreads / 64; ++w) t += sh[w];
  return t;
}

__device__ __forceinline__ bool lb_active(float x, float g, float lo, float hi) {
  return (x <= lo && g > 0.f) || (x >= hi && g < 0.f);
}

// dir = -H g on the variables not held by a bound (two-loop recursion over the m newest pairs; `order[k]` = ring slot of
// the k-th oldest pair); scal[0] = g . dir, scal[1] = |dir|^2, scal[2] = max |projected gradient|
__global__ __launch_bounds__(kLbThreads) void lb_direction_kernel(LbfgsDev L, int m, const int *order_in) {
  __shared__ float sh[kLbThreads / 64];
  __shared__ float a[kLbMem];
  __shared__ int order[kLbMem];
  const int tid = threadIdx.x, D = L.D;
  if (tid < kLbMem) order[tid] = (tid < m) ? order_in[tid] : 0;
  __syncthreads();
  float pg = 0.f;
  for (int i = tid; i < D; i += kLbThreads) {
    const float q = lb_active(L.x[i], L.g[i], L.lo[i], L.hi[i]) ? 0.f : L.g[i];
    L.dir[i] = q;
    pg = fmaxf(pg, fabsf(q));
  }
  // max via the sum machinery on a monotone transform would lose exactness: do a plain max reduction
  for (int off = 32; off > 0; off >>= 1) pg = fmaxf(pg, __shfl_down(pg, off, 64));
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = pg;
  __syncthreads();
  float pgmax = 0.f;
  for (int w = 0; w < kLbThreads / 64; ++w) pgmax = fmaxf(pgmax, sh[w]);
  __syncthreads();
  for (int k = m - 1; k >= 0; --k) {
    const float *s = L.S + (size_t)order[k] * D, *y = L.Y + (size_t)order[k] * D;
    float dot = 0.f;
    for (int i = tid; i < D; i += kLbThreads) dot += s[i] * L.dir[i];
    dot = lb_block_sum(dot, sh);
    const float ak = L.rho[order[k]] * dot;
    if (tid == 0) a[k] = ak;
    for (int i = tid; i < D; i += kLbThreads) L.dir[i] -= ak * y[i];
    __syncthreads();
  }
  float gamma = 1.f;
  if (m > 0) {
    const float *s = L.S + (size_t)order[m - 1] * D, *y = L.Y + (size_t)order[m - 1] * D;
    float sy = 0.f, yy = 0.f;
    for (int i = tid; i < D; i += kLbThreads) {
      sy += s[i] * y[i];
      yy += y[i] * y[i];
    }
    sy = lb_block_sum(sy, sh);
    yy = lb_block_sum(yy, sh);
    gamma = sy / yy;
  }
  for (int i = tid; i < D; i += kLbThreads) L.dir[i] *= gamma;
  __syncthreads();
  for (int k = 0; k < m; ++k) {
    const float *s = L.S + (size_t)order[k] * D, *y = L.Y + (size_t)order[k] * D;
    float dot = 0.f;
    for (int i = tid; i < D; i += kLbThreads) dot += y[i] * L.dir[i];
    dot = lb_block_sum(dot, sh);
    const float c = a[k] - L.rho[order[k]] * dot;
    for (int i = tid; i < D; i += kLbThreads) L.dir[i] += s[i] * c;
    __syncthreads();
  }
  float gd = 0.f, dd = 0.f;
  for (int i = tid; i < D; i += kLbThreads) {
    const float d = lb_active(L.x[i], L.g[i], L.lo[i], L.hi[i]) ? 0.f : -L.dir[i];
    L.dir[i] = d;
    gd += d * L.g[i];
    dd += d * d;
  }
  gd = lb_block_sum(gd, sh);
  dd = lb_block_sum(dd, sh);
  if (tid == 0) {
    L.scal[0] = gd;
    L.scal[1] = dd;
    L.scal[2] = pgmax;
  }
}

// steepest-descent restart: dir = -projected gradient; same scalars
__global__ __launch_bounds__(kLbThreads) void lb_steepest_kernel(LbfgsDev L) {
  __shared__ float sh[kLbThreads / 64];
  const int tid = threadIdx.x, D = L.D;
  float gd = 0.f, dd = 0.f;
  for (int i = tid; i < D; i += kLbThreads) {
    const float d = lb_active(L.x[i], L.g[i], L.lo[i], L.hi[i]) ? 0.f : -L.g[i];
    L.dir[i] = d;
    gd += d * L.g[i];
    dd += d * d;
  }
  gd = lb_block_sum(gd, sh);
  dd = lb_block_sum(dd, sh);
  if (tid == 0) {
    L.scal[0] = gd;
    L.scal[1] = dd;
  }
}

// trial point xt = clip(x + alpha dir); scal[3] = g . (xt - x)
__global__ __launch_bounds__(kLbThreads) void lb_trial_kernel(LbfgsDev L, float alpha) {
  __shared__ float sh[kLbThreads / 64];
  const int tid = threadIdx.x, D = L.D;
  float dec = 0.f;
  for (int i = tid; i < D; i += kLbThreads) {
    const float v = fminf(fmaxf(L.x[i] + alpha * L.dir[i], L.lo[i]), L.hi[i]);
    L.xt[i] = v;
    dec += L.g[i] * (v - L.x[i]);
  }
  dec = lb_block_sum(dec, sh);
  if (tid == 0) L.scal[3] = dec;
}

// starting point into the box: x = clip(x), in place (reads nothing but x and the bounds)
__global__ __launch_bounds__(kLbThreads) void lb_clip_kernel(LbfgsDev L) {
  for (int i = threadIdx.x; i < L.D; i += kLbThreads) L.x[i] = fminf(fmaxf(L.x[i], L.lo[i]), L.hi[i]);
}

// accepted step: (s, y) into ring slot `slot`, x <- xt, g <- gt; scal[4] = s . y, scal[5] = |s|^2, scal[6] = |y|^2
__global__ __launch_bounds__(kLbThreads) void lb_accept_kernel(LbfgsDev L, int slot) {
  __shared__ float sh[kLbThreads / 64];
  const int tid = threadIdx.x, D = L.D;
  float *s = L.S + (size_t)slot * D, *y = L.Y + (size_t)slot * D;
  float sy = 0.f, ss = 0.f, yy = 0.f;
  for (int i = tid; i < D; i += kLbThreads) {
    const float si = L.xt[i] - L.x[i], yi = L.gt[i] - L.g[i];
    s[i] = si;
    y[i] = yi;
    sy += si * yi;
    ss += si * si;
    yy += yi * yi;
    L.x[i] = L.xt[i];
    L.g[i] = L.gt[i];
  }
  sy = lb_block_sum(sy, sh);
  ss = lb_block_sum(ss, sh);
  yy = lb_block_sum(yy, sh);
  if (tid == 0) {
    L.scal[4] = sy;
    L.scal[5] = ss;
    L.scal[6] = yy;
    // a pair that fails the curvature test stays neutral in the two-loop recursion (rho = 0) should its slot be listed
    L.rho[slot] = (sy > 0.f && sy > 1e-10f * sqrtf(ss * yy)) ? 1.f / sy : 0.f;
  }
}

}  // namespace lc
