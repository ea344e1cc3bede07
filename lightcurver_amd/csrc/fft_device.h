// Wave-level complex FFT in LDS (Stockham radix-2 autosort) for the zero-padded convolutions of the
// joint forward model.  One wave transforms one length-L sequence; the four waves of a workgroup
// work on different rows / columns, so no workgroup barrier is needed inside a transform.
#pragma once
#include "lc_common.h"

namespace lc {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {  // a * conj(b)
  return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}

// LDS traffic of one wave is issued and completed in order; this only stops the compiler from
// moving LDS accesses of different lanes' data across the point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a: L complex samples (natural order), b: L complex scratch, tw[k] = exp(-2 pi i k / L), k < L/2.
// Returns the buffer (a or b) that holds the transform, natural order, unscaled.
template <int L, bool INV>
__device__ __forceinline__ float2 *wave_fft(float2 *a, float2 *b, const float2 *tw, int lane) {
  float2 *in = a, *out = b;
#pragma unroll
  for (int Ns = 1; Ns < L; Ns <<= 1) {
    wave_lds_sync();
#pragma unroll
    for (int j0 = 0; j0 < L / 2; j0 += kWave) {
      const int j = j0 + lane;
      if (L / 2 >= kWave || j < L / 2) {
        const int k = j & (Ns - 1);
        float2 w = tw[k * (L / (2 * Ns))];
        if (INV) w.y = -w.y;
        const float2 x = in[j];
        const float2 y = cmul(in[j + L / 2], w);
        const int o = ((j - k) << 1) + k;
        out[o] = make_float2(x.x + y.x, x.y + y.y);
        out[o + Ns] = make_float2(x.x - y.x, x.y - y.y);
      }
    }
    float2 *t = in;
    in = out;
    out = t;
  }
  wave_lds_sync();
  return in;
}

}  // namespace lc
