// Noise propagation of the PSF fit on the device (SURVEY.md kernel K11; SPEC in DESIGN.md section 3):
//   W_j(x)^2 = sum_s ( up0(w_s) (*) kappa_{s,j}^2 )(x),   kappa_{s,j} = starlet scale j of r_s,
//   r_s = a_s * ry_s (x) rx_s : the response of dL/dB to a unit of whitened noise in the central data pixel of star s.
// Replaces the propagate_noise call inside build_psf (lightcurver/processes/psf_modelling.py:164-171).
//
// r_s is separable and so is every smoothed scale c_j = cy_j (x) cx_j of its (separable, edge-replicating) starlet;
// with the 1-D details dy_j = cy_j - cy_{j+1}, dx_j = cx_j - cx_{j+1}
//   kappa_j = cy_j (x) dx_j + dy_j (x) cx_{j+1}
//   kappa_j^2 = cy_j^2 (x) dx_j^2 + 2 cy_j dy_j (x) dx_j cx_{j+1} + dy_j^2 (x) cx_{j+1}^2      (three rank-1 terms,
// no cancellation between large numbers), so the 2-D convolution with the zero-inserted weights is three pairs of
// small dense products per (star, scale):  tmp = w (n x n) . KX (n x N),  W^2 += KY^T (N x n) . tmp.
// Kernel 1 builds the 1-D tables in double precision, kernel 2 does the products in fp32 (one block per frame x scale).
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int kNoiseThreads = 256;

// tables [F][S][J][3 terms][2: ky, kx][N] (float); block = (frame, star), N threads
__global__ void psf_noise_tables_kernel(int S, int N, int ss, int J, const float *stars, float *tables) {
  extern __shared__ double nsh[];  // cur[2 axes][N], nxt[2 axes][N]
  double *cur = nsh, *nxt = nsh + 2 * N;
  const int fs = blockIdx.x, x = threadIdx.x;
  const float *sp = stars + (size_t)fs * 4;
  const double a = sp[0];
  const int n = N / ss, b0 = ss * (n / 2), shift = b0 - (N - 1) / 2;
  const double c_off = (N % 2 == 0) ? 0.5 : 0.0, is2 = 1.0 / ((double)kSigmaG * kSigmaG);
  const double nrm = 0.3989422804014327 / (double)kSigmaG;
  if (x < N) {
    for (int ax = 0; ax < 2; ++ax) {
      // ax 0: rows (y, star parameter 2), ax 1: columns (x, star parameter 1)
      const double delta = ss * (double)sp[ax == 0 ? 2 : 1] + c_off;
      const int o = (int)nearbyint(delta);
      double v = 0.0;
      for (int du = 0; du < ss; ++du) {
        const int t = b0 + du - x;  // sample of the star's Gaussian that maps pixel x onto the central data pixel
        if (t >= o - kRg && t <= o + kRg) v += nrm * exp(-0.5 * (t - delta) * (t - delta) * is2);
      }
      cur[ax * N + x] = v;
    }
  }
  __syncthreads();
  const double b3[5] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
  for (int j = 0; j < J; ++j) {
    const int d = 1 << j;
    if (x < N) {
      for (int ax = 0; ax < 2; ++ax) {
        double acc = 0.0;
        for (int t = -2; t <= 2; ++t) acc += b3[t + 2] * cur[ax * N + min(max(x + t * d, 0), N - 1)];
        nxt[ax * N + x] = acc;
      }
      const double cy = cur[x], cyn = nxt[x], cx = cur[N + x], cxn = nxt[N + x];
      const double dy = cy - cyn, dx = cx - cxn;
      float *T = tables + (((size_t)fs * J + j) * 3) * 2 * N;
      // kappa^2 enters shifted by ss * (n / 2) - c with zero fill (SPEC: kappa^2(. + ss p* - c)), which drops its first
      // `shift` rows and columns; kappa is proportional to the star's amplitude
      const double a2 = (x >= shift) ? a * a : 0.0;
      T[(0 * 2 + 0) * N + x] = (float)(a2 * cy * cy);
      const double keep = (x >= shift) ? 1.0 : 0.0;
      T[(0 * 2 + 1) * N + x] = (float)(keep * dx * dx);
      T[(1 * 2 + 0) * N + x] = (float)(a2 * 2.0 * cy * dy);
      T[(1 * 2 + 1) * N + x] = (float)(keep * dx * cxn);
      T[(2 * 2 + 0) * N + x] = (float)(a2 * dy * dy);
      T[(2 * 2 + 1) * N + x] = (float)(keep * cxn * cxn);
    }
    __syncthreads();
    if (x < N) {
      cur[x] = nxt[x];
      cur[N + x] = nxt[N + x];
    }
    __syncthreads();
  }
}

// W [F][J][N*N]; block = (frame, scale)
template <int N, int SS>
__global__ __launch_bounds__(kNoiseThreads) void psf_noise_accumulate_kernel(int S, int J, const float *stars,
                                                                             const float *wgt, const float *tables,
                                                                             float *W) {
  constexpr int n = N / SS, b0 = SS * (n / 2), NPT = (N * N + kNoiseThreads - 1) / kNoiseThreads;
  constexpr int TN = (n * N + kNoiseThreads - 1) / kNoiseThreads;
  extern __shared__ float nshf[];
  float *wS = nshf;                   // [n][n]
  float *tmp = wS + n * n;            // [n][N]
  float *KY = tmp + n * N;            // [3N]: zeros, table, zeros  (index N + position)
  float *KX = KY + 3 * N;             // [3N]
  const int f = blockIdx.x / J, j = blockIdx.x % J, tid = threadIdx.x;
  float acc[NPT];
#pragma unroll
  for (int m = 0; m < NPT; ++m) acc[m] = 0.f;
  for (int k = tid; k < 6 * N; k += kNoiseThreads) KY[k] = 0.f;  // KY and KX are contiguous
  for (int s = 0; s < S; ++s) {
    if (stars[((size_t)f * S + s) * 4] == 0.f) continue;  // padding star (block-uniform)
    __syncthreads();
    const float *ws = wgt + ((size_t)f * S + s) * n * n;
    for (int k = tid; k < n * n; k += kNoiseThreads) {
      const float wv = ws[k];
      wS[k] = (wv > 0.f && wv < 3.0e38f) ? wv : 0.f;
    }
    for (int term = 0; term < 3; ++term) {
      const float *T = tables + ((((size_t)f * S + s) * J + j) * 3 + term) * 2 * N;
      __syncthreads();  // previous term's products are done with KY / tmp
      for (int k = tid; k < N; k += kNoiseThreads) {
        KY[N + k] = T[k];
        KX[N + k] = T[N + k];
      }
      __syncthreads();
      // tmp[i][v] = sum_jx w[i][jx] * kx[v + b0 - ss jx]
#pragma unroll
      for (int m = 0; m < TN; ++m) {
        const int k = tid + m * kNoiseThreads;
        if (k < n * N) {
          const int i = k / N, v = k % N;
          const float *kx = KX + N + v + b0;
          const float *wr = wS + i * n;
          float t = 0.f;
#pragma unroll 8
          for (int jx = 0; jx < n; ++jx) t = fmaf(wr[jx], kx[-SS * jx], t);
          tmp[k] = t;
        }
      }
      __syncthreads();
      // W^2[u][v] += sum_i ky[u + b0 - ss i] * tmp[i][v]
#pragma unroll
      for (int m = 0; m < NPT; ++m) {
        const int k = tid + m * kNoiseThreads;
        if (k < N * N) {
          const int u = k / N, v = k % N;
          const float *ky = KY + N + u + b0;
          float t = 0.f;
#pragma unroll 8
          for (int i = 0; i < n; ++i) t = fmaf(ky[-SS * i], tmp[i * N + v], t);
          acc[m] += t;
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < NPT; ++m) {
    const int k = tid + m * kNoiseThreads;
    if (k < N * N) W[((size_t)f * J + j) * N * N + k] = sqrtf(fmaxf(acc[m], 0.f));
  }
}

}  // namespace lc
