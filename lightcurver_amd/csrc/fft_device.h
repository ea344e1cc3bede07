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
// (L = 16 * N2 with N2 = 2^m or 3 * 2^m: 32, 48, 96, 192, 384 ... -- the smallest alias-free length of an N-point 'same'
//  convolution is 3N/2, so the factor 3 saves a quarter of the work of the next power of two)
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
// exp(-2 pi i m / 24), m in [0, 24): the twiddles of the radix-3 level for lengths 3, 6, 12, 24 (literals for compile-time m)
__device__ __forceinline__ float2 w24(int m) {
  constexpr float c[7] = {1.f, 0.9659258262890683f, 0.8660254037844387f, 0.7071067811865476f, 0.5f, 0.25881904510252074f, 0.f};
  // cos(pi m / 12) and sin(pi m / 12) from the first quadrant
  const int q = m / 6, r = m % 6;  // quadrant, position inside it
  const float cr = c[r], sr = c[6 - r];
  float co, si;
  if (q == 0) { co = cr; si = sr; }
  else if (q == 1) { co = -sr; si = cr; }
  else if (q == 2) { co = -cr; si = -sr; }
  else { co = sr; si = -cr; }
  return make_float2(co, -si);
}
// In-lane FFT of N2 = 2^m or 3 * 2^m points (natural order in and out).  The factor 3 is taken by one decimation-in-time
// level on top of three power-of-two transforms: X[k + P q] = sum_r W3^(r q) W_N2^(r k) FFT_P(x[3 m + r])[k].
template <int N2, bool INV>
__device__ __forceinline__ void inlane_fft_any(float2 (&x)[N2]) {
  if constexpr ((N2 & (N2 - 1)) == 0) {
    inlane_fft<N2, INV>(x);
  } else {
    static_assert(N2 % 3 == 0 && ((N2 / 3) & (N2 / 3 - 1)) == 0 && 24 % N2 == 0, "length 3, 6, 12 or 24");
    constexpr int P = N2 / 3;
    float2 y0[P], y1[P], y2[P];
#pragma unroll
    for (int m = 0; m < P; ++m) {
      y0[m] = x[3 * m];
      y1[m] = x[3 * m + 1];
      y2[m] = x[3 * m + 2];
    }
    inlane_fft<P, INV>(y0);
    inlane_fft<P, INV>(y1);
    inlane_fft<P, INV>(y2);
    constexpr float kS3 = 0.8660254037844386f;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      float2 w1 = w24(k * (24 / N2)), w2 = w24(2 * k * (24 / N2));
      if (INV) {
        w1.y = -w1.y;
        w2.y = -w2.y;
      }
      const float2 t1 = cmul(y1[k], w1), t2 = cmul(y2[k], w2);
      const float2 sm = make_float2(t1.x + t2.x, t1.y + t2.y), df = make_float2(t1.x - t2.x, t1.y - t2.y);
      const float2 a0 = y0[k];
      const float2 mid = make_float2(fmaf(-0.5f, sm.x, a0.x), fmaf(-0.5f, sm.y, a0.y));
      // forward: X1 = mid - i c df, X2 = mid + i c df;  inverse: the conjugate roots
      const float2 rot = INV ? make_float2(-kS3 * df.y, kS3 * df.x) : make_float2(kS3 * df.y, -kS3 * df.x);
      x[k] = make_float2(a0.x + sm.x, a0.y + sm.y);
      x[k + P] = make_float2(mid.x + rot.x, mid.y + rot.y);
      x[k + 2 * P] = make_float2(mid.x - rot.x, mid.y - rot.y);
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
  if constexpr (N2 >= 24) LC_LAUNDER(l16);  // keep the 2 * N2 twiddle registers from being hoisted out of the caller's loops
  inlane_fft_any<N2, false>(x);
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
  if constexpr (N2 >= 24) LC_LAUNDER(l16);
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
  inlane_fft_any<N2, true>(x);
}

}  // namespace lc
