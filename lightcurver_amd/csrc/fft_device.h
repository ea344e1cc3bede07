// Register-resident complex FFTs for the zero-padded convolutions of the joint forward model: a group of 16
// lanes transforms one length-L sequence (four sequences per wave), lane exchanges by DPP, no LDS round trip.
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

// ---- quarter-wave register FFT ------------------------------------------------------------------------
// One length-L transform per group of 16 lanes (four per wave, side by side), data in registers:
//   "stride" layout : lane n1 (0..15), register n2 (0..L/16-1) holds sample  n = n1 + 16 n2
//   "block"  layout : lane l,          register k2              holds bin     k = k2 + (L/16) * bitrev4(l)
// forward  = in-lane FFT over n2, twiddle W_L^(n1 k2), 16-point decimation-in-frequency FFT across the lanes
//            (lane exchanges by xor-shuffles): stride in -> block out;
// inverse  = 16-point decimation-in-time FFT across the lanes, conjugate twiddle, in-lane inverse FFT:
//            block in -> stride out (unnormalised).  No bit-reversal pass and no LDS round trip is needed.
__device__ __forceinline__ int bitrev4(int x) {
  return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3);
}
__device__ __forceinline__ float2 w32(int k) {  // exp(-2 pi i k / 32), k in [0, 16): compile-time k folds to literals
  constexpr float c[9] = {1.f, 0.9807852804032304f, 0.9238795325112867f, 0.8314696123025452f, 0.7071067811865476f,
                          0.5555702330196022f, 0.3826834323650898f, 0.19509032201612825f, 0.f};
  // cos(pi k / 16) = c[k] for k <= 8, -c[16 - k] above; sin(pi k / 16) = c[8 - k] for k <= 8, c[k - 8] above
  const float co = (k <= 8) ? c[k] : -c[16 - k];
  const float si = (k <= 8) ? c[8 - k] : c[k - 8];
  return make_float2(co, -si);
}
__host__ __device__ constexpr int brev_bits(int i, int bits) {
  int r = 0;
  for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
  return r;
}
template <int N2, bool INV>
__device__ __forceinline__ void inlane_fft(float2 (&x)[N2]) {
  constexpr int BITS = ilog2(N2);
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    const int j = brev_bits(i, BITS);
    if (i < j) {
      const float2 t = x[i];
      x[i] = x[j];
      x[j] = t;
    }
  }
#pragma unroll
  for (int h = 1; h < N2; h <<= 1) {
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      if ((i & h) == 0) {
        float2 w = w32((i & (h - 1)) * (32 / (2 * h)));
        if (INV) w.y = -w.y;
        const float2 a = x[i], b = cmul(x[i + h], w);
        x[i] = make_float2(a.x + b.x, a.y + b.y);
        x[i + h] = make_float2(a.x - b.x, a.y - b.y);
      }
    }
  }
}
__device__ __forceinline__ float2 shfl2(float2 v, int src) {
  return make_float2(__shfl(v.x, src, 64), __shfl(v.y, src, 64));
}
// value of lane (l ^ H) inside the 16-lane row, H in {1, 2, 4, 8}: DPP only, no LDS crossbar
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_take(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                CTRL, 0xF, BANK, false));
}
template <int H>
__device__ __forceinline__ float row_xor(float v) {
  if constexpr (H == 1) return dpp_take<0xB1, 0xF>(v, v);        // quad_perm [1,0,3,2]
  else if constexpr (H == 2) return dpp_take<0x4E, 0xF>(v, v);   // quad_perm [2,3,0,1]
  else if constexpr (H == 8) return dpp_take<0x128, 0xF>(v, v);  // row_ror:8
  else {
    // lanes with bit 2 set (banks 1, 3) read lane - 4 = row_ror:4, the others lane + 4 = row_ror:12
    const float t = dpp_take<0x124, 0xA>(v, v);
    return dpp_take<0x12C, 0x5>(t, v);
  }
}
template <int H>
__device__ __forceinline__ float2 row_xor2(float2 v) {
  return make_float2(row_xor<H>(v.x), row_xor<H>(v.y));
}

// One decimation-in-frequency stage across the lanes (span H): lower lane a + b, upper lane (a - b) w.
// Written branch- and select-free: t = p + s x with s = +-1, then a multiplication by w (lower lanes: 1).
template <int L, int H>
__device__ __forceinline__ void dif_stage(float2 (&x)[L / 16], int l16, const float2 *tw) {
  const bool upper = (l16 & H) != 0;
  const float2 wt = tw[(l16 & (H - 1)) * (L / (2 * H))];
  const float sgn = upper ? -1.f : 1.f;
  const float2 w = upper ? wt : make_float2(1.f, 0.f);
#pragma unroll
  for (int k2 = 0; k2 < L / 16; ++k2) {
    const float2 p = row_xor2<H>(x[k2]);
    const float2 t = make_float2(fmaf(sgn, x[k2].x, p.x), fmaf(sgn, x[k2].y, p.y));
    x[k2] = cmul(t, w);
  }
}
// One decimation-in-time stage (inverse direction): upper lane pre-multiplied by conj w, then a + b / a - b.
template <int L, int H>
__device__ __forceinline__ void dit_stage_inv(float2 (&x)[L / 16], int l16, const float2 *tw) {
  const bool upper = (l16 & H) != 0;
  const float2 wt = tw[(l16 & (H - 1)) * (L / (2 * H))];
  const float sgn = upper ? -1.f : 1.f;
  const float2 w = upper ? make_float2(wt.x, -wt.y) : make_float2(1.f, 0.f);
#pragma unroll
  for (int k2 = 0; k2 < L / 16; ++k2) {
    const float2 t = cmul(x[k2], w);
    const float2 p = row_xor2<H>(t);
    x[k2] = make_float2(fmaf(sgn, t.x, p.x), fmaf(sgn, t.y, p.y));
  }
}
// tw[m] = exp(-2 pi i m / L), m < L (LDS)
template <int L>
__device__ __forceinline__ void quarter_fft_fwd(float2 (&x)[L / 16], int l16, const float2 *tw) {
  constexpr int N2 = L / 16;
  if constexpr (L >= 512) LC_LAUNDER(l16);  // keep the 2 * N2 twiddle registers from being hoisted out of the caller's loops
  inlane_fft<N2, false>(x);
#pragma unroll
  for (int k2 = 1; k2 < N2; ++k2) x[k2] = cmul(x[k2], tw[l16 * k2]);
  dif_stage<L, 8>(x, l16, tw);
  dif_stage<L, 4>(x, l16, tw);
  dif_stage<L, 2>(x, l16, tw);
  dif_stage<L, 1>(x, l16, tw);
}
template <int L>
__device__ __forceinline__ void quarter_fft_inv(float2 (&x)[L / 16], int l16, const float2 *tw) {
  constexpr int N2 = L / 16;
  if constexpr (L >= 512) LC_LAUNDER(l16);
  dit_stage_inv<L, 1>(x, l16, tw);
  dit_stage_inv<L, 2>(x, l16, tw);
  dit_stage_inv<L, 4>(x, l16, tw);
  dit_stage_inv<L, 8>(x, l16, tw);
#pragma unroll
  for (int k2 = 1; k2 < N2; ++k2) {
    float2 w = tw[l16 * k2];
    w.y = -w.y;
    x[k2] = cmul(x[k2], w);
  }
  inlane_fft<N2, true>(x);
}

}  // namespace lc
