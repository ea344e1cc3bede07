// Field distortion inside the PSF fit (build_psf(field_distortion=True), reference call site
// lightcurver/processes/psf_modelling.py:164-171 with `field_distortion` / `stamp_coordinates`; kernel K13 of SURVEY.md 8(a)).
// Model frozen in DESIGN.md section 3 (unverified against STARRED):
//   star i of a frame sits at the rescaled frame coordinates (x_i, y_i); A_i = [[1 + dil_x, shear], [shear, 1 + dil_y]] with
//   every entry c0 + c1 x_i + c2 y_i;  its PSF is  T_i = Moffat_i + W_i[B]  where
//     Moffat_i(u) ~ (1 + (u - c)^T Q_i (u - c))^-beta, Q_i = A_i^-T Q A_i^-1 (the analytic Moffat seen through the distortion,
//                   unit sum on the grid), and
//     W_i[B](u)   = bilinear_0(B, c + A_i^-1 (u - c)) / det A_i   (the pixel grid resampled, flux conserving).
// This header holds the device side: the Moffat raster / parameter gradient in terms of Q (smooth everywhere, unlike
// (fwhm_x, fwhm_y, phi) of a nearly round Moffat), the resampling of the grid for every star and its exact adjoint as an
// ordered gather (no atomics: results do not depend on scheduling).
#pragma once
#include "lc_common.h"

namespace lc {

__device__ __forceinline__ void distort_matrix(const float *coef, float x, float y, float &a00, float &a01, float &a11) {
  a00 = 1.f + coef[0] + coef[1] * x + coef[2] * y;
  a11 = 1.f + coef[3] + coef[4] * x + coef[5] * y;
  a01 = coef[6] + coef[7] * x + coef[8] * y;
}

// ---- Moffat from its quadratic form: q = (q11, q12, q22, beta), M(x, y) = (1 + q11 x^2 + 2 q12 x y + q22 y^2)^-beta / sum ----
__device__ inline double block_sum_dd(double v, double *sh) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ void moffat_q_raster_kernel(int N, const float *q, float *Tm) {
  __shared__ double sh[256];
  const int f = blockIdx.x;
  const double q11 = q[f * 4], q12 = q[f * 4 + 1], q22 = q[f * 4 + 2], beta = q[f * 4 + 3];
  const int c = (N - 1) / 2;
  double acc = 0;
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    const double x = i % N - c, y = i / N - c;
    acc += pow(1.0 + q11 * x * x + 2.0 * q12 * x * y + q22 * y * y, -beta);
  }
  const double S = block_sum_dd(acc, sh);
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    const double x = i % N - c, y = i / N - c;
    Tm[(size_t)f * N * N + i] = (float)(pow(1.0 + q11 * x * x + 2.0 * q12 * x * y + q22 * y * y, -beta) / S);
  }
}

// d loss / d (q11, q12, q22, beta) from d loss / d T (gT), through the unit-sum normalisation
__global__ void moffat_q_grad_kernel(int N, const float *q, const float *gT, float *gq) {
  __shared__ double sh[256];
  const int f = blockIdx.x;
  const double q11 = q[f * 4], q12 = q[f * 4 + 1], q22 = q[f * 4 + 2], beta = q[f * 4 + 3];
  const int c = (N - 1) / 2;
  double sM = 0, gM = 0, sd[4] = {0, 0, 0, 0}, gd[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    const double x = i % N - c, y = i / N - c;
    const double base = 1.0 + q11 * x * x + 2.0 * q12 * x * y + q22 * y * y;
    const double M = pow(base, -beta), Mb1 = -beta * M / base;
    const double d[4] = {Mb1 * x * x, Mb1 * 2.0 * x * y, Mb1 * y * y, -log(base) * M};
    const double g = gT[(size_t)f * N * N + i];
    sM += M;
    gM += g * M;
    for (int k = 0; k < 4; ++k) {
      sd[k] += d[k];
      gd[k] += g * d[k];
    }
  }
  const double S = block_sum_dd(sM, sh), G = block_sum_dd(gM, sh);
  for (int k = 0; k < 4; ++k) {
    const double a = block_sum_dd(sd[k], sh), b = block_sum_dd(gd[k], sh);
    if (threadIdx.x == 0) gq[f * 4 + k] = (float)((b - G * a / S) / S);
  }
}

// ---- resampling of the pixel grid: frame f, star i -> image (f * S + i) ---------------------------------------------------
// sample position of destination pixel (u, v) in the source grid
__device__ __forceinline__ void warp_sample(int u, int v, float c, float i00, float i01, float i11, float &X, float &Y) {
  const float qx = (float)v - c, qy = (float)u - c;
  X = c + i00 * qx + i01 * qy;
  Y = c + i01 * qx + i11 * qy;
}

__global__ __launch_bounds__(256) void psf_warp_kernel(int N, int S, const float *coef, const float *xy, const float *B, float *Bw) {
  const int img = blockIdx.x, f = img / S;
  float a00, a01, a11;
  distort_matrix(coef + f * 9, xy[2 * img], xy[2 * img + 1], a00, a01, a11);
  const float det = a00 * a11 - a01 * a01, idet = 1.f / det;
  const float i00 = a11 * idet, i01 = -a01 * idet, i11 = a00 * idet;
  const float c = (float)((N - 1) / 2);
  const float *src = B + (size_t)f * N * N;
  float *dst = Bw + (size_t)img * N * N;
  for (int i = threadIdx.x; i < N * N; i += 256) {
    const int u = i / N, v = i % N;
    float X, Y;
    warp_sample(u, v, c, i00, i01, i11, X, Y);
    const float x0f = floorf(X), y0f = floorf(Y);
    const float fx = X - x0f, fy = Y - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < N && xx >= 0 && xx < N) ? src[yy * N + xx] : 0.f; };
    const float top = (1.f - fx) * at(y0, x0) + fx * at(y0, x0 + 1);
    const float bot = (1.f - fx) * at(y0 + 1, x0) + fx * at(y0 + 1, x0 + 1);
    dst[i] = ((1.f - fy) * top + fy * bot) * idet;
  }
}

// exact adjoint, as a gather in a fixed order: source pixel k collects every destination pixel whose 2 x 2 footprint
// contains it.  The distortion is close to the identity (|entries - identity| <= 0.25, enforced by the host), so those
// lie within +-2 of the forward image of k.
__global__ __launch_bounds__(256) void psf_warp_adjoint_kernel(int N, int S, const float *coef, const float *xy, const float *g,
                                                               float *gsrc) {
  // grid (frames, ceil(N^2 / 256)): one source pixel per thread (one block per frame left 60 % of the CUs idle and took
  // 400 us of the 500 us iteration at 100 frames x 8 stars x 64^2; the sums per pixel are unchanged).  The matrices of
  // the frame's stars are made once per block; the candidate window of a star is as wide as its matrix asks for
  // (|d - A k| < |a00| + |a01| + 1/2 along x, |a01| + |a11| + 1/2 along y: 3 x 3 for distortions of a few per cent) -
  // the candidates left out have weight zero.
  __shared__ float sm[16][8];
  const int f = blockIdx.x;
  const float c = (float)((N - 1) / 2);
  if ((int)threadIdx.x < S) {
    const int img = f * S + threadIdx.x;
    float a00, a01, a11;
    distort_matrix(coef + f * 9, xy[2 * img], xy[2 * img + 1], a00, a01, a11);
    const float det = a00 * a11 - a01 * a01, idet = 1.f / det;
    float *m = sm[threadIdx.x];
    m[0] = a00; m[1] = a01; m[2] = a11;
    m[3] = a11 * idet; m[4] = -a01 * idet; m[5] = a00 * idet; m[6] = idet;
    const int hv = min((int)floorf(fabsf(a00) + fabsf(a01) + 0.501f), 2), hu = min((int)floorf(fabsf(a01) + fabsf(a11) + 0.501f), 2);
    m[7] = __int_as_float(hv | (hu << 8));
  }
  __syncthreads();
  for (int i = blockIdx.y * 256 + threadIdx.x; i < N * N; i += 256 * gridDim.y) {
    const int ky = i / N, kx = i % N;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
      const int img = f * S + s;
      const float *m = sm[s];
      const float a00 = m[0], a01 = m[1], a11 = m[2], i00 = m[3], i01 = m[4], i11 = m[5], idet = m[6];
      const int hw = __float_as_int(m[7]), hv = hw & 255, hu = hw >> 8;
      // forward image of the source pixel: u = c + A (k - c)
      const float rx = (float)kx - c, ry = (float)ky - c;
      const int vc = (int)nearbyintf(c + a00 * rx + a01 * ry), uc = (int)nearbyintf(c + a01 * rx + a11 * ry);
      const float *gi = g + (size_t)img * N * N;
      float part = 0.f;
      for (int u = max(uc - hu, 0); u <= min(uc + hu, N - 1); ++u)
        for (int v = max(vc - hv, 0); v <= min(vc + hv, N - 1); ++v) {
          float X, Y;
          warp_sample(u, v, c, i00, i01, i11, X, Y);
          const float x0f = floorf(X), y0f = floorf(Y);
          const float fx = X - x0f, fy = Y - y0f;
          const int x0 = (int)x0f, y0 = (int)y0f;
          const float wx = (x0 == kx ? 1.f - fx : 0.f) + (x0 + 1 == kx ? fx : 0.f);
          const float wy = (y0 == ky ? 1.f - fy : 0.f) + (y0 + 1 == ky ? fy : 0.f);
          part = fmaf(wx * wy, gi[u * N + v], part);
        }
      acc = fmaf(part, idet, acc);
    }
    gsrc[(size_t)f * N * N + i] = acc;
  }
}

}  // namespace lc
