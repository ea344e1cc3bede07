// Wave-level complex FFT in LDS (Stockham radix-2 autosort) for the zero-padded convolutions of the
// joint forward model.  One wave transforms one length-L sequence; the four waves of a workgroup
// work on different rows / columns, so no workgroup barrier is needed inside a transform.
#pragma once
#include "lc_common.h"
#include "starlet_device.h"

namespace lc {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {  // a * conj(b)
  return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}

// a: L complex samples (natural order), b: L complex scratch, tw[m] = exp(-2 pi i m / L), m < L.
// Radix-4 Stockham stages (one LDS round trip per two bits of L), one radix-2 stage when log2 L is odd.
// Returns the buffer (a or b) that holds the transform, natural order, unscaled.
template <int L, bool INV>
__device__ __forceinline__ float2 *wave_fft(float2 *a, float2 *b, const float2 *tw, int lane) {
  float2 *in = a, *out = b;
  int Ns = 1;
#pragma unroll
  for (int stage = 0; stage < 16; ++stage) {
    if (Ns * 4 > L) break;
    wave_lds_sync();
    constexpr int Q = L / 4;
#pragma unroll
    for (int j0 = 0; j0 < Q; j0 += kWave) {
      const int j = j0 + lane;
      if (Q >= kWave || j < Q) {
        const int k = j & (Ns - 1);
        const int step = k * (L / (4 * Ns));
        float2 w1 = tw[step], w2 = tw[2 * step], w3 = tw[3 * step];
        if (INV) {
          w1.y = -w1.y;
          w2.y = -w2.y;
          w3.y = -w3.y;
        }
        const float2 x0 = in[j];
        const float2 x1 = cmul(in[j + Q], w1), x2 = cmul(in[j + 2 * Q], w2), x3 = cmul(in[j + 3 * Q], w3);
        const float2 s02 = make_float2(x0.x + x2.x, x0.y + x2.y), d02 = make_float2(x0.x - x2.x, x0.y - x2.y);
        const float2 s13 = make_float2(x1.x + x3.x, x1.y + x3.y), d13 = make_float2(x1.x - x3.x, x1.y - x3.y);
        // forward: -i * d13 = (d13.y, -d13.x); inverse: +i * d13 = (-d13.y, d13.x)
        const float2 r13 = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
        const int o = ((j - k) << 2) + k;
        out[o] = make_float2(s02.x + s13.x, s02.y + s13.y);
        out[o + Ns] = make_float2(d02.x + r13.x, d02.y + r13.y);
        out[o + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
        out[o + 3 * Ns] = make_float2(d02.x - r13.x, d02.y - r13.y);
      }
    }
    float2 *t = in;
    in = out;
    out = t;
    Ns <<= 2;
  }
  if (Ns < L) {  // remaining factor of two
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
