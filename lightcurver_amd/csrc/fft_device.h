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

// ---- quarter-wave (or half-wave) register FFT ------------------------------------------------------------------------
// One length-L transform per group of 16 lanes (four per wave, side by side), data in registers:
// (L = 16 * N2 with N2 = 2^m or 3 * 2^m: 32, 48, 96, 192, 384 ... -- the smallest alias-free length of an N-point 'same'
//  convolution is 3N/2, so the factor 3 saves a quarter of the work of the next power of two)
//   "stride" layout : lane n1 (0..15), register n2 (0..L/16-1) holds sample  n = n1 + 16 n2
//   "block"  layout : lane l,          register k2              holds bin     k = k2 + (L/16) * bitrev4(l)
// forward  = in-lane FFT over n2, twiddle W_L^(n1 k2), 16-point decimation-in-frequency FFT across the lanes
//            (lane exchanges by DPP, see fft_index below: "lane" n1 / l above means the lane's transform index);
//            stride in -> block out;
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
// Lane exchanges of the 16-point transform across a row of 16 lanes.  Every stage pairs a lane with one partner, and
// the exchange rides on the arithmetic instruction itself (v_fmac_f32_dpp: x += s * x[partner], one instruction per
// component): the DPP controls that are involutions of a row are quad_perm [1,0,3,2] (lane ^ 1), quad_perm [2,3,0,1]
// (lane ^ 2), row_half_mirror (lane ^ 7) and row_ror:8 (lane ^ 8) -- there is none for lane ^ 4.  So the transform
// index of a lane is not its lane number: fft_index(lane) = lane ^ (lane & 4 ? 3 : 0) (an involution), under which the
// partner across index bit 2 is lane ^ 7.  Callers take their sample / bin positions from fft_index().
__device__ __forceinline__ int fft_index(int lane16) { return lane16 ^ ((lane16 & 4) ? 3 : 0); }
// x += s * x[partner] for both components of NE complex registers, in place.  The compiler folds a DPP move into VOP2
// users but not into a fused multiply-add, hence the assembly; one block covers up to six elements and opens with the two
// wait states a DPP read needs after a VALU write of the same register (nothing inside a block reads what it wrote).
#define LC_XF1(a, c) "v_fmac_f32_dpp %" #a ", %" #a ", %" #c " "
#define LC_DEF_EXCHANGE(NAME, CTL)                                                                                         \
  __device__ __forceinline__ void NAME##_1(float2 &a, float s) {                                                          \
    asm("s_nop 1\n\t" LC_XF1(0, 2) CTL "\n\t" LC_XF1(1, 2) CTL : "+v"(a.x), "+v"(a.y) : "v"(s));                           \
  }                                                                                                                        \
  __device__ __forceinline__ void NAME##_6(float2 &a, float2 &b, float2 &c, float2 &d, float2 &e, float2 &f, float s) {    \
    asm("s_nop 1\n\t" LC_XF1(0, 12) CTL "\n\t" LC_XF1(1, 12) CTL "\n\t" LC_XF1(2, 12) CTL "\n\t" LC_XF1(3, 12) CTL "\n\t"     \
        LC_XF1(4, 12) CTL "\n\t" LC_XF1(5, 12) CTL "\n\t" LC_XF1(6, 12) CTL "\n\t" LC_XF1(7, 12) CTL "\n\t"                   \
        LC_XF1(8, 12) CTL "\n\t" LC_XF1(9, 12) CTL "\n\t" LC_XF1(10, 12) CTL "\n\t" LC_XF1(11, 12) CTL                        \
        : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y), "+v"(c.x), "+v"(c.y), "+v"(d.x), "+v"(d.y), "+v"(e.x), "+v"(e.y),   \
          "+v"(f.x), "+v"(f.y)                                                                                            \
        : "v"(s));                                                                                                         \
  }
LC_DEF_EXCHANGE(exch_b0, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
LC_DEF_EXCHANGE(exch_b1, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
LC_DEF_EXCHANGE(exch_b2, "row_half_mirror row_mask:0xf bank_mask:0xf")
LC_DEF_EXCHANGE(exch_b3, "row_ror:8 row_mask:0xf bank_mask:0xf")
#undef LC_DEF_EXCHANGE
#undef LC_XF1
// x[k2] += s * (x[k2] of the lane whose fft_index differs in bit H), all k2
template <int H, int N2>
__device__ __forceinline__ void exchange_add(float2 (&x)[N2], float s) {
  if constexpr (H == 16) {  // half-wave transforms: the partner sits in the neighbouring DPP row (lane ^ 16), through the LDS crossbar
#pragma unroll
    for (int k = 0; k < N2; ++k) {
      const float px = __shfl_xor(x[k].x, 16, 64), py = __shfl_xor(x[k].y, 16, 64);
      x[k].x = fmaf(px, s, x[k].x);
      x[k].y = fmaf(py, s, x[k].y);
    }
    return;
  }
  constexpr int N6 = N2 / 6 * 6;
#pragma unroll
  for (int k = 0; k < N6; k += 6) {
    if constexpr (H == 1) exch_b0_6(x[k], x[k + 1], x[k + 2], x[k + 3], x[k + 4], x[k + 5], s);
    else if constexpr (H == 2) exch_b1_6(x[k], x[k + 1], x[k + 2], x[k + 3], x[k + 4], x[k + 5], s);
    else if constexpr (H == 4) exch_b2_6(x[k], x[k + 1], x[k + 2], x[k + 3], x[k + 4], x[k + 5], s);
    else exch_b3_6(x[k], x[k + 1], x[k + 2], x[k + 3], x[k + 4], x[k + 5], s);
  }
#pragma unroll
  for (int k = N6; k < N2; ++k) {
    if constexpr (H == 1) exch_b0_1(x[k], s);
    else if constexpr (H == 2) exch_b1_1(x[k], s);
    else if constexpr (H == 4) exch_b2_1(x[k], s);
    else exch_b3_1(x[k], s);
  }
}

// One decimation-in-frequency stage across the lanes (span H): lower lane a + b, upper lane (a - b) w.
// Branch- and select-free: t = x + s x[partner] with s = +-1 (the upper lane gets b - a, the sign goes into its
// twiddle), then a multiplication by w (lower lanes: 1).  j = transform index of the lane (fft_index_n).
template <int L, int H, int LPF, bool BARE = false>
__device__ __forceinline__ void dif_stage(float2 (&x)[L / LPF], int j, const float2 *tw) {
  const bool upper = (j & H) != 0;
  if constexpr (BARE) {  // H = 1: the twiddle is 1, so w = -+1; the caller takes the sign of the odd lanes elsewhere
    static_assert(H == 1, "bare stage");
    exchange_add<H>(x, upper ? -1.f : 1.f);
    return;
  }
  float2 wt = tw[(j & (H - 1)) * (L / (2 * H))];
  asm volatile("" : "+v"(wt.x), "+v"(wt.y));  // every lane loads: no branch around the read
  const float sgn = upper ? -1.f : 1.f;
  const float2 w = upper ? make_float2(-wt.x, -wt.y) : make_float2(1.f, 0.f);
  exchange_add<H>(x, sgn);
#pragma unroll
  for (int k2 = 0; k2 < L / LPF; ++k2) x[k2] = cmul(x[k2], w);
}
// One decimation-in-time stage (inverse direction): upper lane pre-multiplied by conj w, then a + b / a - b.
// The upper lane carries -t through the exchange (sign in its twiddle), so that both lanes do x = t - s t[partner].
template <int L, int H, int LPF, bool BARE = false>
__device__ __forceinline__ void dit_stage_inv(float2 (&x)[L / LPF], int j, const float2 *tw) {
  const bool upper = (j & H) != 0;
  if constexpr (BARE) {  // H = 1: the odd lanes arrive already negated
    static_assert(H == 1, "bare stage");
    exchange_add<H>(x, upper ? 1.f : -1.f);
    return;
  }
  float2 wt = tw[(j & (H - 1)) * (L / (2 * H))];
  asm volatile("" : "+v"(wt.x), "+v"(wt.y));
  const float msg = upper ? 1.f : -1.f;
  const float2 w = upper ? make_float2(-wt.x, wt.y) : make_float2(1.f, 0.f);
#pragma unroll
  for (int k2 = 0; k2 < L / LPF; ++k2) x[k2] = cmul(x[k2], w);
  exchange_add<H>(x, msg);
}
// Transforms over LPF = 16 lanes (four per wave) or 32 lanes (two per wave; the longest grids, whose 24 registers per
// lane at LPF = 16 would not leave room for anything else): transform index of a lane / lane of an index, and the bit
// reversal of the block layout
template <int LPF>
__device__ __forceinline__ int fft_index_n(int l) {
  if constexpr (LPF == 16) return fft_index(l);
  else return fft_index(l & 15) | (l & 16);
}
template <int LPF>
__device__ __forceinline__ int bitrev_n(int x) {
  if constexpr (LPF == 16) return bitrev4(x);
  else return (bitrev4(x & 15) << 1) | ((x >> 4) & 1);
}
// tw[m] = exp(-2 pi i m / L), m < L (LDS)
// ODDNEG: the last forward stage (span 1) has twiddle 1, i.e. multiplies the odd-index lanes by -1, and the first inverse
// stage undoes exactly that.  A forward transform that is followed by lane-local arithmetic and the inverse transform (the
// column passes: multiply by a spectrum) may therefore leave the odd lanes negated (ODDNEG forward) and hand them to an
// ODDNEG inverse as they are: two complex multiplications per element less, same numbers (products with +-1 are exact).
template <int L, int LPF = 16, bool ODDNEG = false>
__device__ __forceinline__ void group_fft_fwd(float2 (&x)[L / LPF], int j, const float2 *tw) {
  constexpr int N2 = L / LPF;
  if constexpr (N2 >= 24) LC_LAUNDER(j);  // keep the 2 * N2 twiddle registers from being hoisted out of the caller's loops
  inlane_fft_any<N2, false>(x);
#pragma unroll
  for (int k2 = 1; k2 < N2; ++k2) x[k2] = cmul(x[k2], tw[j * k2]);
  if constexpr (LPF == 32) dif_stage<L, 16, LPF>(x, j, tw);
  dif_stage<L, 8, LPF>(x, j, tw);
  dif_stage<L, 4, LPF>(x, j, tw);
  dif_stage<L, 2, LPF>(x, j, tw);
  dif_stage<L, 1, LPF, ODDNEG>(x, j, tw);
}
template <int L, int LPF = 16, bool ODDNEG = false>
__device__ __forceinline__ void group_fft_inv(float2 (&x)[L / LPF], int j, const float2 *tw) {
  constexpr int N2 = L / LPF;
  if constexpr (N2 >= 24) LC_LAUNDER(j);
  dit_stage_inv<L, 1, LPF, ODDNEG>(x, j, tw);
  dit_stage_inv<L, 2, LPF>(x, j, tw);
  dit_stage_inv<L, 4, LPF>(x, j, tw);
  dit_stage_inv<L, 8, LPF>(x, j, tw);
  if constexpr (LPF == 32) dit_stage_inv<L, 16, LPF>(x, j, tw);
#pragma unroll
  for (int k2 = 1; k2 < N2; ++k2) {
    float2 w = tw[j * k2];
    w.y = -w.y;
    x[k2] = cmul(x[k2], w);
  }
  inlane_fft_any<N2, true>(x);
}

}  // namespace lc
