// Field distortion of the narrow PSF (kernel K13 of SURVEY.md 8(a)): the resampling STARRED's apply_distortion performs
// for the reference at lightcurver/processes/star_photometry.py:291-304 and roi_file_preparation.py:169-180, with the
// distortion model frozen in DESIGN.md section 3 (unverified against STARRED, like the rest of the SPEC):
//   kwargs_distortion = {dilation_x, dilation_y, shear}, each a first-order polynomial c0 + c1 x + c2 y of the star's
//   rescaled frame coordinates (x, y) in [-0.5, 0.5] (lightcurver/utilities/image_coordinates.py:6-27);
//   A = [[1 + dilation_x, shear], [shear, 1 + dilation_y]];
//   out[u][v] = bilinear_0(psf, c + A^-1 ((v, u) - c)),  c = (N - 1) // 2 (the zero-lag index: the PSF centre stays
//   put), bilinear_0 = order-1 interpolation with zeros outside the grid; the result is renormalised to unit sum.
#include "lc_common.h"

using namespace lc;

namespace {

__device__ __forceinline__ void distortion_matrix(const float *coef, float x, float y, float &a00, float &a01, float &a11) {
  a00 = 1.f + coef[0] + coef[1] * x + coef[2] * y;
  a11 = 1.f + coef[3] + coef[4] * x + coef[5] * y;
  a01 = coef[6] + coef[7] * x + coef[8] * y;
}

// one workgroup per position k: warp, block sum (fixed order), normalise
__global__ __launch_bounds__(256) void distort_psf_kernel(int N, const float *psf, const float *coef, const float *xy, float *out) {
  __shared__ double sh[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  float a00, a01, a11;
  distortion_matrix(coef, xy[2 * k], xy[2 * k + 1], a00, a01, a11);
  const float det = a00 * a11 - a01 * a01;
  const float i00 = a11 / det, i01 = -a01 / det, i11 = a00 / det;  // A^-1 (symmetric)
  const float c = (float)((N - 1) / 2);
  float *o = out + (size_t)k * N * N;
  double acc = 0.0;
  for (int i = tid; i < N * N; i += 256) {
    const int u = i / N, v = i % N;
    const float qx = (float)v - c, qy = (float)u - c;
    const float X = c + i00 * qx + i01 * qy, Y = c + i01 * qx + i11 * qy;
    const float x0f = floorf(X), y0f = floorf(Y);
    const float fx = X - x0f, fy = Y - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < N && xx >= 0 && xx < N) ? psf[yy * N + xx] : 0.f; };
    const float top = (1.f - fx) * at(y0, x0) + fx * at(y0, x0 + 1);
    const float bot = (1.f - fx) * at(y0 + 1, x0) + fx * at(y0 + 1, x0 + 1);
    const float val = (1.f - fy) * top + fy * bot;
    o[i] = val;
    acc += val;
  }
  sh[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  const double total = sh[0];
  const float inv = (total != 0.0) ? (float)(1.0 / total) : 0.f;
  for (int i = tid; i < N * N; i += 256) o[i] *= inv;
}

}  // namespace

extern "C" int lc_apply_distortion(lc_ctx *ctx, int N, int K, const float *narrow_psf, const float *coeffs, const float *xy,
                                   float *out) {
  if (!ctx) return LC_ERR_INVALID;
  if (!narrow_psf || !coeffs || !xy || !out || N <= 0 || K <= 0) LC_FAIL(ctx, LC_ERR_INVALID, "lc_apply_distortion: invalid argument");
  LC_ENTER(ctx);
  for (int i = 0; i < 9; ++i)
    if (!std::isfinite(coeffs[i])) LC_FAIL(ctx, LC_ERR_INVALID, "lc_apply_distortion: non-finite coefficient");
  const size_t NN = (size_t)N * N;
  float *dpsf = nullptr, *dcoef = nullptr, *dxy = nullptr, *dout = nullptr;
  struct Guard {
    float **p[4];
    ~Guard() {
      for (auto q : p)
        if (*q) (void)hipFree(*q);
    }
  } guard{{&dpsf, &dcoef, &dxy, &dout}};
  LC_HIP(ctx, hipMalloc((void **)&dpsf, NN * sizeof(float)));
  LC_HIP(ctx, hipMalloc((void **)&dcoef, 9 * sizeof(float)));
  LC_HIP(ctx, hipMalloc((void **)&dxy, (size_t)K * 2 * sizeof(float)));
  LC_HIP(ctx, hipMalloc((void **)&dout, (size_t)K * NN * sizeof(float)));
  hipStream_t q = ctx->stream;
  LC_HIP(ctx, hipMemcpyAsync(dpsf, narrow_psf, NN * sizeof(float), hipMemcpyHostToDevice, q));
  LC_HIP(ctx, hipMemcpyAsync(dcoef, coeffs, 9 * sizeof(float), hipMemcpyHostToDevice, q));
  LC_HIP(ctx, hipMemcpyAsync(dxy, xy, (size_t)K * 2 * sizeof(float), hipMemcpyHostToDevice, q));
  hipLaunchKernelGGL(distort_psf_kernel, dim3(K), dim3(256), 0, q, N, dpsf, dcoef, dxy, dout);
  LC_HIP(ctx, hipGetLastError());
  LC_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)K * NN * sizeof(float), hipMemcpyDeviceToHost, q));
  LC_HIP(ctx, hipStreamSynchronize(q));
  // a singular or inverting matrix makes a meaningless PSF: report it instead of returning zeros
  for (int k = 0; k < K; ++k) {
    const float x = xy[2 * k], y = xy[2 * k + 1];
    const float a00 = 1.f + coeffs[0] + coeffs[1] * x + coeffs[2] * y, a11 = 1.f + coeffs[3] + coeffs[4] * x + coeffs[5] * y,
                a01 = coeffs[6] + coeffs[7] * x + coeffs[8] * y;
    if (!(a00 * a11 - a01 * a01 > 1e-3f)) LC_FAIL(ctx, LC_ERR_INVALID, "lc_apply_distortion: distortion matrix is singular or inverting");
  }
  return LC_OK;
}
