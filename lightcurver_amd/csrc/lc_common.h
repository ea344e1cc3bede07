// Shared host/device definitions of liblcmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/lcmi.h"

namespace lc {

// Target resolution of the two-channel (MCS-style) scheme: Gaussian of FWHM 2 high-res pixels.
constexpr float kGaussFwhm = 2.0f;
constexpr float kSigmaG = 0.84932180028801907f;  // 2 / (2 sqrt(2 ln 2))
constexpr int kRg = 5;                           // half support of the sampled Gaussian: the first dropped sample is >= 5.5 px = 6.5 sigma away (8e-10 of the peak)
constexpr int kWave = 64;

__host__ __device__ constexpr int ntaps(int ss) { return 2 * kRg + 1 + 2 * (ss - 1); }
__host__ __device__ constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
};

}  // namespace lc

struct lc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  int n_cu = 0;
};

// learning rate and bias corrections of AdaBelief iteration t (0-based), in double on the host: the kernels take
// them as numbers instead of evaluating three double-precision pow() in one lane per iteration
inline void adabelief_schedule(const lc_adabelief_cfg &ab, int t, float &lr, float &bc1, float &bc2) {
  double l = ab.init_learning_rate;
  if (ab.schedule_learning_rate) l *= std::pow((double)ab.decay_rate, (double)t / (double)ab.transition_steps);
  lr = (float)l;
  bc1 = (float)(1.0 / (1.0 - std::pow((double)ab.b1, (double)(t + 1))));
  bc2 = (float)(1.0 / (1.0 - std::pow((double)ab.b2, (double)(t + 1))));
}

#define LC_HIP(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return LC_ERR_DEVICE;                                                                \
    }                                                                                      \
  } while (0)

// Every entry point that can allocate, launch or copy makes its context's device current first: the calling thread
// may have another device current (two contexts in one process, a call from another thread).
#define LC_ENTER(ctx)                                                                      \
  do {                                                                                     \
    hipError_t e_ = hipSetDevice((ctx)->device);                                           \
    if (e_ != hipSuccess) {                                                                \
      (ctx)->err = std::string("hipSetDevice: ") + hipGetErrorString(e_);                  \
      return LC_ERR_DEVICE;                                                                \
    }                                                                                      \
  } while (0)

#define LC_FAIL(ctx, code, msg) \
  do {                          \
    (ctx)->err = (msg);         \
    return (code);              \
  } while (0)
