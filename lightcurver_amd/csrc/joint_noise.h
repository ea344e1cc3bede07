// Noise propagation of the joint fit on the device (SURVEY.md kernel K11; SPEC in DESIGN.md section 3):
//   W_j(x)^2 = sum_e ( up0(w_e) (*) kappa_{e,j}^2(. + ss p* - c) )(x),  kappa_{e,j} = starlet scale j of r_e,
//   r_e[u'][v'] = sum_{(u,v) in block(p*)} s_e[u - u' + c][v - v' + c]   (adjoint of D_ss . conv_same(., s_e) at the
//   central data pixel p*).  Replaces starred.utils.noise_utils.propagate_noise(method='SLIT', likelihood_type='chi2')
//   as called at lightcurver/processes/star_photometry.py:108-110 and roi_modelling.py:299-301.
// The responses are not separable, so the convolutions go through the FFT pipeline of the epoch kernel
// (joint_epoch_kernel<C, true>): per scale one launch turns the squared, shifted coefficients of every epoch into
// spectra, one launch convolves them with the zero-inserted weights, and the epoch reduction kernel sums the epochs.
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int kNzThreads = 256;

// r_e on the N x N grid; grid (ceil(N*N / 256), E)
__global__ void nz_response_kernel(int N, int ss, const float *psf, float *r) {
  const int e = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N * N) return;
  const int n = N / ss, c = (N - 1) / 2, b0 = ss * (n / 2);
  const int up = k / N, vp = k % N;
  const float *s = psf + (size_t)e * N * N;
  float acc = 0.f;
  for (int du = 0; du < ss; ++du)
    for (int dv = 0; dv < ss; ++dv) {
      const int a = b0 + du - up + c, b = b0 + dv - vp + c;
      if (a >= 0 && a < N && b >= 0 && b < N) acc += s[a * N + b];
    }
  r[(size_t)e * N * N + k] = acc;
}

// one edge-replicating 5-tap a-trous pass (dilation d) along rows (axis 1) or columns (axis 0), batched over epochs
__global__ void nz_pass_kernel(int N, int d, int axis, const float *in, float *out) {
  const int e = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N * N) return;
  const float *src = in + (size_t)e * N * N;
  const int u = k / N, v = k % N;
  const float b3[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
  float acc = 0.f;
#pragma unroll
  for (int t = -2; t <= 2; ++t) {
    const int uu = axis == 0 ? min(max(u + t * d, 0), N - 1) : u;
    const int vv = axis == 1 ? min(max(v + t * d, 0), N - 1) : v;
    acc = fmaf(b3[t + 2], src[uu * N + vv], acc);
  }
  out[(size_t)e * N * N + k] = acc;
}

// scene[e][u][v] = kappa^2[u + shift][v + shift] (zero where either index leaves the grid), kappa = c - cn, or = c for
// the coarse scale (cn == null)
__global__ void nz_kappa2_kernel(int N, int shift, const float *c, const float *cn, float *scene) {
  const int e = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N * N) return;
  const int u = k / N + shift, v = k % N + shift;
  float val = 0.f;
  if (u >= 0 && u < N && v >= 0 && v < N) {
    const size_t o = (size_t)e * N * N + (size_t)u * N + v;
    const float kap = cn ? c[o] - cn[o] : c[o];
    val = kap * kap;
  }
  scene[(size_t)e * N * N + k] = val;
}

// zero-insertion up-sampling of the inverse variances
__global__ void nz_up0_kernel(int N, int ss, const float *wgt, float *up) {
  const int e = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N * N) return;
  const int n = N / ss, u = k / N, v = k % N;
  float val = 0.f;
  if (u % ss == 0 && v % ss == 0) {
    const float w = wgt[(size_t)e * n * n + (size_t)(u / ss) * n + v / ss];
    val = (w > 0.f && w < 3.0e38f) ? w : 0.f;
  }
  up[(size_t)e * N * N + k] = val;
}

__global__ void nz_sqrt_kernel(int NN, const float *w2, float *W) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < NN) W[k] = sqrtf(fmaxf(w2[k], 0.f));
}

}  // namespace lc
