// Stamp pre-processing in one pass over the stack and a copy-bandwidth probe.
//
// One workgroup per stamp.  Per pixel it restates what the reference does on the host before the fits:
//   cutout_making.py:43-51             noisemap = max(sqrt((t rms)^2 + |t data|), 1e-7) / t      (t = exposure time)
//   roi_file_preparation.py:162-164    data, noisemap /= coefficient
//   psf_modelling.py:139-143           both NaN -> data 0, noisemap 1, pixel masked
//   roi_file_preparation.py:194-201    both NaN -> data 0, noisemap 1e7; noisemap[flagged] *= 1000
//   star_photometry.py:309-316         same, but a flagged pixel boosts the WHOLE epoch, once
//   psf_modelling.py:144-153           number of masked pixels per stamp (the caller applies the 40 % cut)
// and emits what the fit objects take: cleaned data, noise map, and weight = good / sigma^2.
#include <cmath>
#include <cstring>
#include <vector>

#include "lc_common.h"
#include "../../include/lcmi.h"

namespace lc {

constexpr int kPrepThreads = 256;

struct PrepArgs {
  int K, npix;
  const float *data, *noisemap, *rms, *exptime, *coefficient;
  const uint8_t *bad;
  float nan_noise, boost;
  int boost_whole_stamp;
  float *data_out, *noisemap_out, *weight_out;
  int *masked_count;
};

__device__ __forceinline__ void prep_pixel(const PrepArgs &A, int k, size_t idx, float inv_coef, float t, float trms2,
                                           float &d, float &s, bool &flagged, bool &both_nan) {
  d = A.data[idx];
  if (A.noisemap) {
    s = A.noisemap[idx];
  } else {
    // electrons: data * t; back to electrons / second at the end (NaN data keeps the noise NaN, as in numpy)
    const float e = d * t;
    s = fmaxf(sqrtf(trms2 + fabsf(e)), 1e-7f) / t;
    if (e != e) s = e;
  }
  d /= inv_coef;
  s /= inv_coef;
  both_nan = (d != d) && (s != s);
  if (both_nan) {
    d = 0.f;
    s = A.nan_noise;
  }
  flagged = A.bad ? (A.bad[idx] != 0) : false;
}

__global__ __launch_bounds__(kPrepThreads) void prep_stamps_kernel(PrepArgs A) {
  __shared__ int cnt[kPrepThreads / 64], flg[kPrepThreads / 64];
  __shared__ int tot[2];
  const int k = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)k * A.npix;
  const float inv_coef = A.coefficient ? A.coefficient[k] : 1.0f;  // the divisor itself: divisions are kept, as in the reference
  const float t = A.exptime ? A.exptime[k] : 1.0f;
  const float trms = A.rms ? t * A.rms[k] : 0.f;
  const float trms2 = trms * trms;
  // pass 1: masked pixels of the stamp (flagged or NaN in both inputs), and whether anything is flagged
  int masked = 0, any = 0;
  for (int p = tid; p < A.npix; p += kPrepThreads) {
    float d, s;
    bool flagged, both_nan;
    prep_pixel(A, k, base + p, inv_coef, t, trms2, d, s, flagged, both_nan);
    masked += (flagged || both_nan) ? 1 : 0;
    any |= flagged ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    masked += __shfl_down(masked, off, 64);
    any |= __shfl_down(any, off, 64);
  }
  if ((tid & 63) == 0) {
    cnt[tid >> 6] = masked;
    flg[tid >> 6] = any;
  }
  __syncthreads();
  if (tid == 0) {
    int m = 0, a = 0;
    for (int w = 0; w < kPrepThreads / 64; ++w) {
      m += cnt[w];
      a |= flg[w];
    }
    tot[0] = m;
    tot[1] = a;
    if (A.masked_count) A.masked_count[k] = m;
  }
  __syncthreads();
  const bool stamp_flagged = tot[1] != 0;
  // pass 2: outputs (the stamp is L2-resident from pass 1)
  for (int p = tid; p < A.npix; p += kPrepThreads) {
    float d, s;
    bool flagged, both_nan;
    prep_pixel(A, k, base + p, inv_coef, t, trms2, d, s, flagged, both_nan);
    if (A.boost > 0.f && (A.boost_whole_stamp ? stamp_flagged : flagged)) s *= A.boost;
    const bool good = !(flagged || both_nan) && (d == d) && (s == s) && s > 0.f && fabsf(s) < 3.0e38f && fabsf(d) < 3.0e38f;
    if (A.data_out) A.data_out[base + p] = d;
    if (A.noisemap_out) A.noisemap_out[base + p] = s;
    if (A.weight_out) A.weight_out[base + p] = good ? 1.0f / (s * s) : 0.f;
  }
}

__global__ void copy_kernel(const float4 *src, float4 *dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace lc

using namespace lc;

extern "C" {

int lc_prepare_stamps(lc_ctx *ctx, int K, int npix, const float *data, const float *noisemap, const float *rms,
                      const float *exptime, const float *coefficient, const uint8_t *bad, float nan_noise,
                      float noise_boost, int boost_whole_stamp, float *data_out, float *noisemap_out,
                      float *weight_out, int32_t *masked_count, float *kernel_ms) {
  if (!ctx) return LC_ERR_INVALID;
  if (K <= 0 || npix <= 0 || !data) LC_FAIL(ctx, LC_ERR_INVALID, "lc_prepare_stamps: invalid argument");
  if (!noisemap && !(rms && exptime))
    LC_FAIL(ctx, LC_ERR_INVALID, "lc_prepare_stamps: without a noise map, rms and exptime are required");
  LC_HIP(ctx, hipSetDevice(ctx->device));
  const size_t tot = (size_t)K * npix;
  std::vector<void *> dev;
  auto up = [&](const void *h, size_t bytes, void **d) -> hipError_t {
    hipError_t e = hipMalloc(d, bytes);
    if (e != hipSuccess) return e;
    dev.push_back(*d);
    return h ? hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
  };
  auto cleanup = [&]() {
    for (void *p : dev) (void)hipFree(p);
  };
#define PREP_TRY(call)                                                \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);   \
      cleanup();                                                      \
      return LC_ERR_DEVICE;                                           \
    }                                                                 \
  } while (0)
  PrepArgs A;
  std::memset(&A, 0, sizeof(A));
  A.K = K;
  A.npix = npix;
  A.nan_noise = nan_noise;
  A.boost = noise_boost;
  A.boost_whole_stamp = boost_whole_stamp;
  void *p = nullptr;
  PREP_TRY(up(data, tot * 4, &p));
  A.data = (const float *)p;
  if (noisemap) {
    PREP_TRY(up(noisemap, tot * 4, &p));
    A.noisemap = (const float *)p;
  } else {
    PREP_TRY(up(rms, (size_t)K * 4, &p));
    A.rms = (const float *)p;
    PREP_TRY(up(exptime, (size_t)K * 4, &p));
    A.exptime = (const float *)p;
  }
  if (coefficient) {
    PREP_TRY(up(coefficient, (size_t)K * 4, &p));
    A.coefficient = (const float *)p;
  }
  if (bad) {
    PREP_TRY(up(bad, tot, &p));
    A.bad = (const uint8_t *)p;
  }
  if (data_out) {
    PREP_TRY(up(nullptr, tot * 4, &p));
    A.data_out = (float *)p;
  }
  if (noisemap_out) {
    PREP_TRY(up(nullptr, tot * 4, &p));
    A.noisemap_out = (float *)p;
  }
  if (weight_out) {
    PREP_TRY(up(nullptr, tot * 4, &p));
    A.weight_out = (float *)p;
  }
  if (masked_count) {
    PREP_TRY(up(nullptr, (size_t)K * 4, &p));
    A.masked_count = (int *)p;
  }
  PREP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(prep_stamps_kernel, dim3(K), dim3(kPrepThreads), 0, ctx->stream, A);
  PREP_TRY(hipGetLastError());
  PREP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  if (data_out) PREP_TRY(hipMemcpyAsync(data_out, A.data_out, tot * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (noisemap_out) PREP_TRY(hipMemcpyAsync(noisemap_out, A.noisemap_out, tot * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (weight_out) PREP_TRY(hipMemcpyAsync(weight_out, A.weight_out, tot * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (masked_count)
    PREP_TRY(hipMemcpyAsync(masked_count, A.masked_count, (size_t)K * 4, hipMemcpyDeviceToHost, ctx->stream));
  PREP_TRY(hipStreamSynchronize(ctx->stream));
  if (kernel_ms) PREP_TRY(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
  cleanup();
#undef PREP_TRY
  return LC_OK;
}

int lc_copy_bandwidth(lc_ctx *ctx, int64_t bytes, int reps, float *gb_per_s) {
  if (!ctx || !gb_per_s || bytes < 1024 || reps <= 0) return LC_ERR_INVALID;
  LC_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n4 = (size_t)bytes / 16;
  float4 *a = nullptr, *b = nullptr;
  LC_HIP(ctx, hipMalloc((void **)&a, n4 * 16));
  if (hipMalloc((void **)&b, n4 * 16) != hipSuccess) {
    (void)hipFree(a);
    LC_FAIL(ctx, LC_ERR_DEVICE, "lc_copy_bandwidth: out of memory");
  }
  (void)hipMemsetAsync(a, 0, n4 * 16, ctx->stream);
  const int blocks = ctx->n_cu > 0 ? ctx->n_cu * 8 : 2048;
  hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a, b, n4);  // warm-up
  (void)hipEventRecord(ctx->ev0, ctx->stream);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a, b, n4);
  (void)hipEventRecord(ctx->ev1, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  (void)hipFree(a);
  (void)hipFree(b);
  if (e != hipSuccess) LC_FAIL(ctx, LC_ERR_DEVICE, std::string("lc_copy_bandwidth: ") + hipGetErrorString(e));
  *gb_per_s = (float)(2.0 * (double)n4 * 16.0 * reps / (ms * 1e-3) / 1e9);  // read + write
  return LC_OK;
}

}  // extern "C"
