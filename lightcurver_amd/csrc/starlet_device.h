// Starlet l1 regulariser of an N x N image held PX-pixels-per-thread by one workgroup:
//   value  lam_hf * sum W_0 |w_0| + lam_sc * sum_{1<=j<J} W_j |w_j|   (coarse scale unpenalised)
//   and its sub-gradient through the exact adjoint of the edge-replicating a-trous transform.
// Used by the PSF fit (on the pixel grid B) and by the joint fit (on the background h).
// Restates STARRED's Loss(regularization_terms='l1_starlet') as frozen in DESIGN.md "SPEC"
// (reference call sites: lightcurver/processes/star_photometry.py:95-111, roi_modelling.py:313-321).
#pragma once
#include <utility>

#include "lc_common.h"

namespace lc {

// Issue priority by progress (k = units of work this wave has finished, wave-uniform): of the two waves of a SIMD the one that
// is behind wins the arbitration, so both reach the barrier together instead of the older one early and the younger one
// alone at 1 / 1.75 of the pair's rate (MI355X_MICROARCH.md, two waves per SIMD: priority, then age).
__device__ __forceinline__ void progress_prio(int k) {
#ifndef LC_NO_PROGRESS_PRIO
  if (k <= 0) __builtin_amdgcn_s_setprio(3);
  else if (k == 1) __builtin_amdgcn_s_setprio(2);
  else if (k == 2) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
}
// The same with the four levels spent on the LAST units of a longer loop (unit `done` of `total`): the lead the older wave
// builds early is taken back at the end, where it would otherwise turn into waiting at the barrier.
__device__ __forceinline__ void progress_prio_end(int done, int total) {
#ifndef LC_NO_PROGRESS_PRIO
  const int left = total - 1 - done;
  if (left >= 3) __builtin_amdgcn_s_setprio(3);
  else if (left == 2) __builtin_amdgcn_s_setprio(2);
  else if (left == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
}

// Packed fp32 arithmetic (v_pk_fma_f32: two FMAs per lane and instruction).  The separable passes are bound by VALU
// issue, so they are written in forms whose operands pair up in aligned 64-bit registers: two taps of one output
// (row pass, transposed column pass), or value and derivative taps of one sample (column pass, transposed row pass).
typedef float lc_v2f __attribute__((ext_vector_type(2)));
// -DLC_SCALAR_FMA -fno-slp-vectorize builds the same passes from v_fma_f32 only: measured slower on gfx950 (C2 16.5 vs 16.0 us
// per iteration, C4 133 vs 125 us), so the packed form is the default.
#ifdef LC_SCALAR_FMA
__device__ __forceinline__ lc_v2f pk_fma(lc_v2f a, lc_v2f b, lc_v2f c) { return (lc_v2f){fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#else
__device__ __forceinline__ lc_v2f pk_fma(lc_v2f a, lc_v2f b, lc_v2f c) { return __builtin_elementwise_fma(a, b, c); }
#endif
__device__ __forceinline__ lc_v2f pk_bcast(float v) { return (lc_v2f){v, v}; }

#ifndef LC_LAUNDER
#define LC_LAUNDER(x) asm volatile("" : "+v"(x))
#endif

// LDS traffic of one wave is issued and completed in order; this only stops the compiler from
// moving LDS accesses of different lanes' data across the point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ inline float wave_sum_shfl(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// starred's first-generation starlet uses the B3 spline [1,4,6,4,1]/16.
__device__ __forceinline__ float b3tap(int t) {  // t in [-2, 2]
  return (t == 0) ? 0.375f : ((t == 1 || t == -1) ? 0.25f : 0.0625f);
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Sum over the LPR (4, 8 or 16) consecutive lanes that own one image line, result in every lane.
// xor-1 / xor-2 butterflies as quad permutes, then the half-row and row mirrors (DPP, no LDS).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int iv = __builtin_bit_cast(int, v);
  const int r = __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xF, 0xF, false);
  return v + __builtin_bit_cast(float, r);
}
template <int LPR>
__device__ __forceinline__ float line_sum(float v) {
  v = dpp_add<0xB1>(v);  // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);  // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) v = dpp_add<0x141>(v);   // row_half_mirror
  if constexpr (LPR >= 16) v = dpp_add<0x140>(v);  // row_mirror
  return v;
}

// Sum over the 64 lanes of the wave (valid in every lane): DPP butterflies inside each row of 16 lanes,
// then the four row totals through scalar broadcasts.  All lanes must be active.
__device__ __forceinline__ float wave_sum(float v) {
  v = line_sum<16>(v);
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}

// Zero-filling DPP lane shifts inside one image line (LPR consecutive lanes, LPR | 16).
template <int CTRL>
__device__ __forceinline__ float dpp_mov0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int K, int LPR>
__device__ __forceinline__ float line_from_lower(float v, int lil) {  // value of lane - K of the same line, else 0
  if constexpr (K >= LPR) {
    return 0.f;
  } else {
    float r = dpp_mov0<0x110 + K>(v);  // row_shr:K
    if constexpr (LPR < 16) r = (lil >= K) ? r : 0.f;
    return r;
  }
}
template <int K, int LPR>
__device__ __forceinline__ float line_from_upper(float v, int lil) {  // value of lane + K of the same line, else 0
  if constexpr (K >= LPR) {
    return 0.f;
  } else {
    float r = dpp_mov0<0x100 + K>(v);  // row_shl:K
    if constexpr (LPR < 16) r = (lil + K < LPR) ? r : 0.f;
    return r;
  }
}

// Adjoint of one edge-replicating 5-tap a-trous pass (dilation D) along a line of N samples:
//   out[x] = sum_t b3[t] * sum_{x'': clamp(x'' + t D) = x} g[x'']
// The calling thread holds samples x0 .. x0+PX-1 of the line in own[]; the LPR = N / PX threads of the
// line are consecutive lanes (lil = lane index inside the line).  Interior samples gather g[x -/+ D],
// g[x -/+ 2D] with zero outside the line: pure DPP lane shifts, no LDS.  The two end samples also collect
// every sample that clamps onto them: masked partial sums reduced over the line's lanes by DPP.
template <int N, int PX, int LPR, int D>
__device__ __forceinline__ void adjoint_line_dpp(const float own[PX], int x0, int lil, float out[PX]) {
  if constexpr (D < PX) {
    float w[3 * PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      w[p] = line_from_lower<1, LPR>(own[p], lil);
      w[PX + p] = own[p];
      w[2 * PX + p] = line_from_upper<1, LPR>(own[p], lil);
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      float acc = 0.375f * own[p];
      acc = fmaf(0.25f, w[PX + p - D] + w[PX + p + D], acc);
      acc = fmaf(0.0625f, w[PX + p - 2 * D] + w[PX + p + 2 * D], acc);
      out[p] = acc;
    }
  } else {
    constexpr int K1 = D / PX, K2 = 2 * D / PX;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      float acc = 0.375f * own[p];
      acc = fmaf(0.25f, line_from_lower<K1, LPR>(own[p], lil) + line_from_upper<K1, LPR>(own[p], lil), acc);
      acc = fmaf(0.0625f, line_from_lower<K2, LPR>(own[p], lil) + line_from_upper<K2, LPR>(own[p], lil), acc);
      out[p] = acc;
    }
  }
  // head sums over x <= D and x <= 2D, tail sums over x >= N-1-D and x >= N-1-2D (the samples that clamp onto the
  // two ends).  x = lil * PX + p, so for a threshold T a lane contributes all of its samples (lil < T / PX), a
  // compile-time prefix of them (lil == T / PX) or nothing: one shared full sum, four short partial sums, selects
  // per lane instead of per sample.
  float full = own[0];
#pragma unroll
  for (int p = 1; p < PX; ++p) full += own[p];
  auto head = [&](auto tc) {
    constexpr int T = decltype(tc)::value, Q = T / PX, R = T % PX;
    float part = own[0];
#pragma unroll
    for (int p = 1; p <= R; ++p) part += own[p];
    return (lil < Q) ? full : ((lil == Q) ? part : 0.f);
  };
  auto tail = [&](auto tc) {
    constexpr int T = decltype(tc)::value, Q = T / PX, R = T % PX;
    float part = own[PX - 1];
#pragma unroll
    for (int p = 1; p <= R; ++p) part += own[PX - 1 - p];
    const int lir = LPR - 1 - lil;
    return (lir < Q) ? full : ((lir == Q) ? part : 0.f);
  };
  float s1 = head(std::integral_constant<int, D>{}), s2 = head(std::integral_constant<int, 2 * D>{});
  float e1 = tail(std::integral_constant<int, D>{}), e2 = tail(std::integral_constant<int, 2 * D>{});
  s1 = line_sum<LPR>(s1);
  s2 = line_sum<LPR>(s2);
  e1 = line_sum<LPR>(e1);
  e2 = line_sum<LPR>(e2);
  if (x0 == 0) out[0] = 0.375f * own[0] + 0.25f * s1 + 0.0625f * s2;
  if (x0 + PX == N) out[PX - 1] = 0.375f * own[PX - 1] + 0.25f * e1 + 0.0625f * e2;
}

// Same operator for line layouts DPP cannot serve (LPR not in {4, 8, 16}): gathers from LDS, serial edge sums.
template <int N, int PX>
__device__ __forceinline__ void adjoint_line_lds(const float *in, int base, int stride, int x0, int d, float out[PX]) {
  float own[PX];
#pragma unroll
  for (int p = 0; p < PX; ++p) own[p] = in[base + (x0 + p) * stride];
  float s1 = 0.f, s2 = 0.f, e1 = 0.f, e2 = 0.f;
  if (x0 == 0) {
    const int m1 = min(d, N - 1), m2 = min(2 * d, N - 1);
    for (int x = 0; x <= m2; ++x) {
      const float gv = in[base + x * stride];
      if (x <= m1) s1 += gv;
      s2 += gv;
    }
  }
  if (x0 + PX == N) {
    const int m1 = max(N - 1 - d, 0), m2 = max(N - 1 - 2 * d, 0);
    for (int x = m2; x <= N - 1; ++x) {
      const float gv = in[base + x * stride];
      if (x >= m1) e1 += gv;
      e2 += gv;
    }
  }
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    const int x = x0 + p;
    float acc = 0.f;
#pragma unroll
    for (int t = -2; t <= 2; ++t) {
      const int xx = x - t * d;
      const int cx = min(max(xx, 0), N - 1);
      const float gv = (t == 0) ? own[p] : in[base + cx * stride];
      acc = fmaf(b3tap(t), (xx >= 0 && xx <= N - 1) ? gv : 0.f, acc);
    }
    if (x == 0) acc = 0.375f * own[p] + 0.25f * s1 + 0.0625f * s2;
    if (x == N - 1) acc = 0.375f * own[p] + 0.25f * e1 + 0.0625f * e2;
    out[p] = acc;
  }
}

template <int... Is, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}


// LDS needed by starlet_l1_grad (floats): two ping-pong images at the 16-byte aligned row stride.
template <int N>
struct StarletLds {
  // forward sweep rows: [4 x first sample][N samples][4 x last sample], 16-byte aligned, so that a neighbour quad
  // that falls off either end is read as the replicated edge through a clamped address, without selects
  static constexpr int TS = N + 1, TSS = N + 8;
  static constexpr int FLOATS = 2 * N * TSS;
};

// img[PX]: the calling thread's pixels (row pu = tid / (N/PX), columns pv = (tid % (N/PX)) * PX ..).
// Wf: [J][N*N] weights or null (then norms[j] is used); qscr: [J][N*N] thread-private scratch, only
// touched when the sub-gradients do not fit in registers.  All N*N/PX threads of the block must call.
// On return l1 holds this thread's share of the value and z[PX] the sub-gradient at its pixels.
template <int N, int PX, int JUSE>
__device__ __forceinline__ void starlet_l1_grad_lds(const float img[PX], const float *Wf, const float *norms, float *qscr,
                                                    float lam_sc, float lam_hf, float *lds, int tid, float &l1,
                                                    float z[PX]) {
  constexpr int J = JUSE;  // detail scales entering the penalty (the PSF / background terms use all of them)
  constexpr int TS = StarletLds<N>::TS;
  float *bufA = lds, *bufB = lds + N * TS;
  const int pu = tid / (N / PX);
  const int pv = (tid % (N / PX)) * PX;
  l1 = 0.f;
#pragma unroll
  for (int p = 0; p < PX; ++p) z[p] = 0.f;
  const int pu_ = pu, pv_ = pv;
  constexpr bool QREG = (J * PX <= 64);  // sub-gradients stay in registers when they fit
  constexpr int TSS = StarletLds<N>::TSS;  // 16-byte aligned rows for the forward sweep
  float *fA = lds, *fB = lds + N * TSS;
  float qreg[QREG ? J : 1][PX];
  float c[PX];
#pragma unroll
  for (int q = 0; q < PX / 4; ++q) {
    c[4 * q] = img[4 * q];
    c[4 * q + 1] = img[4 * q + 1];
    c[4 * q + 2] = img[4 * q + 2];
    c[4 * q + 3] = img[4 * q + 3];
  }
  // one row of the forward buffers: own samples, plus the replicated edge quads from the two end threads
  auto put_row = [&](float *buf, int pu, int pv, const float (&v)[PX]) {
    float *row = buf + pu * TSS + 4;
#pragma unroll
    for (int q = 0; q < PX / 4; ++q)
      *(float4 *)&row[pv + 4 * q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    if (pv == 0) *(float4 *)&row[-4] = make_float4(v[0], v[0], v[0], v[0]);
    if (pv + PX == N) *(float4 *)&row[N] = make_float4(v[PX - 1], v[PX - 1], v[PX - 1], v[PX - 1]);
  };
  put_row(fA, pu, pv, c);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < J; ++j) {
    constexpr int dummy = 0;
    (void)dummy;
    const int d = 1 << j;
    int pu = pu_, pv = pv_;
    LC_LAUNDER(pu);
    LC_LAUNDER(pv);
    // weights of this scale: issued before the row pass so the latency hides behind it
    float wj[PX];
    if (Wf) {
      const float4 *wp = (const float4 *)(Wf + (size_t)j * N * N + (size_t)pu * N + pv);
#pragma unroll
      for (int q = 0; q < PX / 4; ++q) {
        const float4 w4 = wp[q];
        wj[4 * q] = w4.x;
        wj[4 * q + 1] = w4.y;
        wj[4 * q + 2] = w4.z;
        wj[4 * q + 3] = w4.w;
      }
    } else {
      const float nv = norms[j];
#pragma unroll
      for (int p = 0; p < PX; ++p) wj[p] = nv;
    }
    // row pass: r = Row_j c (edge replicating), own samples from registers, neighbours by 128-bit reads
    float r[PX];
    const float *row = fA + pu * TSS + 4;
    if (d < 4) {
      float w[PX + 8];
      const float4 Lc = ld4(row + pv - 4);
      const float4 Rc = ld4(row + pv + PX);
      w[0] = Lc.x;
      w[1] = Lc.y;
      w[2] = Lc.z;
      w[3] = Lc.w;
#pragma unroll
      for (int p = 0; p < PX; ++p) w[4 + p] = c[p];
      w[PX + 4] = Rc.x;
      w[PX + 5] = Rc.y;
      w[PX + 6] = Rc.z;
      w[PX + 7] = Rc.w;
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        float acc = 0.375f * w[4 + p];
        acc = fmaf(0.25f, w[4 + p - d] + w[4 + p + d], acc);
        acc = fmaf(0.0625f, w[4 + p - 2 * d] + w[4 + p + 2 * d], acc);
        r[p] = acc;
      }
    } else {
#pragma unroll
      for (int q = 0; q < PX / 4; ++q) {
        float a4[4] = {0.375f * c[4 * q], 0.375f * c[4 * q + 1], 0.375f * c[4 * q + 2], 0.375f * c[4 * q + 3]};
#pragma unroll
        for (int t = -2; t <= 2; ++t) {
          if (t == 0) continue;
          // d is a multiple of 4 here: a neighbour quad lies entirely inside the row or entirely beyond one end
          const int idx = min(max(pv + 4 * q + t * d, -4), N);
          const float4 v4 = ld4(row + idx);
          const float bt = b3tap(t);
          a4[0] = fmaf(bt, v4.x, a4[0]);
          a4[1] = fmaf(bt, v4.y, a4[1]);
          a4[2] = fmaf(bt, v4.z, a4[2]);
          a4[3] = fmaf(bt, v4.w, a4[3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) r[4 * q + e] = a4[e];
      }
    }
#pragma unroll
    for (int q = 0; q < PX / 4; ++q)
      *(float4 *)&fB[pu * TSS + 4 + pv + 4 * q] = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
    __syncthreads();
    // column pass: c_{j+1} = Col_j r; detail coefficients, l1 value and sub-gradient
    const float lam = (j == 0) ? lam_hf : lam_sc;
    float q[PX];
#pragma unroll
    for (int qq = 0; qq < PX / 4; ++qq) {
      float a4[4] = {0.375f * r[4 * qq], 0.375f * r[4 * qq + 1], 0.375f * r[4 * qq + 2], 0.375f * r[4 * qq + 3]};
#pragma unroll
      for (int t = -2; t <= 2; ++t) {
        if (t == 0) continue;
        const int uu = min(max(pu + t * d, 0), N - 1);
        const float4 v4 = ld4(fB + uu * TSS + 4 + pv + 4 * qq);
        const float bt = b3tap(t);
        a4[0] = fmaf(bt, v4.x, a4[0]);
        a4[1] = fmaf(bt, v4.y, a4[1]);
        a4[2] = fmaf(bt, v4.z, a4[2]);
        a4[3] = fmaf(bt, v4.w, a4[3]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = 4 * qq + e;
        const float w = c[p] - a4[e];
        const float lw = lam * wj[p];
        l1 = fmaf(lw, fabsf(w), l1);
        q[p] = (w > 0.f) ? lw : ((w < 0.f) ? -lw : 0.f);
        c[p] = a4[e];
      }
    }
    if constexpr (QREG) {
#pragma unroll
      for (int p = 0; p < PX; ++p) qreg[j][p] = q[p];
    } else {
      float4 *qp = (float4 *)(qscr + (size_t)j * N * N + (size_t)pu * N + pv);
#pragma unroll
      for (int qq = 0; qq < PX / 4; ++qq) qp[qq] = make_float4(q[4 * qq], q[4 * qq + 1], q[4 * qq + 2], q[4 * qq + 3]);
    }
    if (j + 1 < J) put_row(fA, pu, pv, c);
    __syncthreads();
  }
  // backward: z_J = 0; z_j = q_j + H_j^T (z_{j+1} - q_j), H_j^T = Row^T Col^T (edge-replicating adjoint).
  // Col^T runs with a column-major thread mapping and Row^T with the row-major one, so that each
  // line (column resp. row) is owned by LPR consecutive lanes and both passes are lane shifts.
  constexpr int LPR = N / PX;
  constexpr bool FAST = (LPR == 4 || LPR == 8 || LPR == 16);
  const int cu0_ = (tid % LPR) * PX;  // column-major mapping: first owned row ...
  const int cv_ = tid / LPR;          // ... of this column
  const int lil_ = tid % LPR;
  static_for(std::make_integer_sequence<int, J>{}, [&](auto jc) {
    constexpr int j = J - 1 - decltype(jc)::value;
    constexpr int d = 1 << j;
    int pu = pu_, pv = pv_, cu0 = cu0_, cv = cv_, lil = lil_;
    LC_LAUNDER(pu);
    LC_LAUNDER(pv);
    LC_LAUNDER(cu0);
    LC_LAUNDER(cv);
    LC_LAUNDER(lil);
    float q[PX];
    if constexpr (QREG) {
#pragma unroll
      for (int p = 0; p < PX; ++p) q[p] = qreg[j][p];
    } else {
      const float4 *qp = (const float4 *)(qscr + (size_t)j * N * N + (size_t)pu * N + pv);
#pragma unroll
      for (int qq = 0; qq < PX / 4; ++qq) {
        const float4 v4 = qp[qq];
        q[4 * qq] = v4.x;
        q[4 * qq + 1] = v4.y;
        q[4 * qq + 2] = v4.z;
        q[4 * qq + 3] = v4.w;
      }
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) bufA[pu * TS + pv + p] = z[p] - q[p];
    __syncthreads();
    float ct[PX];
    if constexpr (FAST) {
      float own[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) own[p] = bufA[(cu0 + p) * TS + cv];
      adjoint_line_dpp<N, PX, LPR, d>(own, cu0, lil, ct);  // Col^T: line = column cv
    } else {
      adjoint_line_lds<N, PX>(bufA, cv, TS, cu0, d, ct);
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) bufB[(cu0 + p) * TS + cv] = ct[p];
    __syncthreads();
    float rt[PX];
    if constexpr (FAST) {
      float own[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) own[p] = bufB[pu * TS + pv + p];
      adjoint_line_dpp<N, PX, LPR, d>(own, pv, lil, rt);  // Row^T: line = row pu
    } else {
      adjoint_line_lds<N, PX>(bufB, pu * TS, 1, pv, d, rt);
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) z[p] = q[p] + rt[p];
  });
}


// ---- register / DPP form (line layouts DPP can serve: LPR = N / PX in {4, 8, 16}) ------------------------------------
// One 5-tap B3 pass at dilation D along a line of N samples held by LPR consecutive lanes (PX samples each, lil = lane
// index inside the line), entirely in registers.  ADJ = false: the edge-replicating pass r[x] = sum_t b_t c[clamp(x + t D)];
// ADJ = true: its exact adjoint.  Both are the zero-padded symmetric stencil plus a boundary term:
//   forward:  r[x] += c[0] * sum_{t: x + t D < 0} b_t + c[N-1] * sum_{t: x + t D > N-1} b_t
//   adjoint:  g[0] += b_1 sum_{x < D} y[x] + b_2 sum_{x < 2D} y[x],   g[N-1] likewise from the other end.
// Neighbours in other lanes arrive by zero-filling DPP row shifts; a neighbour that lies in another line of the same
// 16-lane DPP row is switched off through its tap coefficient (per lane, a handful of selects per pass) instead of a
// select per fetched value, so that a tap is one multiply-add on a DPP operand.
struct LineLane {
  int lil;
  float is_first, is_last;  // 1 in the first / last lane of the line, else 0
};
template <int LPR>
__device__ __forceinline__ float swz_line_first(float v) {
  constexpr int pat = (~(LPR - 1)) & 0x1F;
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), pat));
}
template <int LPR>
__device__ __forceinline__ float swz_line_last(float v) {
  constexpr int pat = ((~(LPR - 1)) & 0x1F) | ((LPR - 1) << 5);
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), pat));
}
template <int K>
__device__ __forceinline__ float dpp_lower(float v) {  // value of lane - K (same DPP row), 0 beyond the row
  if constexpr (K >= 16) return 0.f; else return dpp_mov0<0x110 + K>(v);
}
template <int K>
__device__ __forceinline__ float dpp_upper(float v) {  // value of lane + K (same DPP row), 0 beyond the row
  if constexpr (K >= 16) return 0.f; else return dpp_mov0<0x100 + K>(v);
}
template <int LPR>
__device__ __forceinline__ float line_sum_fused(float v) {  // sum over the line's lanes, in every lane
  v += dpp_mov0<0xB1>(v);                             // quad_perm [1,0,3,2]
  v += dpp_mov0<0x4E>(v);                             // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) v += dpp_mov0<0x141>(v);    // row_half_mirror
  if constexpr (LPR >= 16) v += dpp_mov0<0x140>(v);   // row_mirror
  return v;
}

template <int N, int PX, int LPR, int D, bool ADJ>
__device__ __forceinline__ void line_pass(const float (&own)[PX], const LineLane &L, float (&out)[PX]) {
  constexpr float b0 = 0.375f, b1 = 0.25f, b2 = 0.0625f;
  const int lil = L.lil;
  if constexpr (D >= PX) {
    constexpr int K1 = D / PX, K2 = 2 * D / PX;
    // taps: lanes lil -+ K1, lil -+ K2 of the same line
    const float cl1 = (K1 < LPR && lil >= K1) ? b1 : 0.f, cu1 = (K1 < LPR && lil + K1 < LPR) ? b1 : 0.f;
    const float cl2 = (K2 < LPR && lil >= K2) ? b2 : 0.f, cu2 = (K2 < LPR && lil + K2 < LPR) ? b2 : 0.f;
    // boundary weights: how much of the stencil of this lane's samples reaches beyond either end
    const float ef = ((lil < K1) ? b1 : 0.f) + ((lil < K2) ? b2 : 0.f);
    const float el = ((lil + K1 >= LPR) ? b1 : 0.f) + ((lil + K2 >= LPR) ? b2 : 0.f);
    float e = 0.f;
    if constexpr (!ADJ) e = fmaf(ef, swz_line_first<LPR>(own[0]), el * swz_line_last<LPR>(own[PX - 1]));
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      float acc = fmaf(b0, own[p], e);
      if constexpr (K1 < LPR) {
        acc = fmaf(cl1, dpp_lower<K1>(own[p]), acc);
        acc = fmaf(cu1, dpp_upper<K1>(own[p]), acc);
      }
      if constexpr (K2 < LPR) {
        acc = fmaf(cl2, dpp_lower<K2>(own[p]), acc);
        acc = fmaf(cu2, dpp_upper<K2>(own[p]), acc);
      }
      out[p] = acc;
    }
    if constexpr (ADJ) {
      float full = own[0];
#pragma unroll
      for (int p = 1; p < PX; ++p) full += own[p];
      const float hs = line_sum_fused<LPR>(full * ef), ts = line_sum_fused<LPR>(full * el);
      out[0] = fmaf(L.is_first, hs, out[0]);
      out[PX - 1] = fmaf(L.is_last, ts, out[PX - 1]);
    }
  } else {
    static_assert(2 * D <= PX, "dilations below PX reach the adjacent lane only");
    const float has_lower = 1.f - L.is_first, has_upper = 1.f - L.is_last;
    float w[3 * PX];  // [lower lane | own | upper lane], only the entries the stencil touches are formed
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      w[q] = (q >= PX - 2 * D) ? has_lower * dpp_lower<1>(own[q]) : 0.f;
      w[PX + q] = own[q];
      w[2 * PX + q] = (q < 2 * D) ? has_upper * dpp_upper<1>(own[q]) : 0.f;
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      float acc = b0 * own[p];
      acc = fmaf(b1, w[PX + p - D] + w[PX + p + D], acc);
      acc = fmaf(b2, w[PX + p - 2 * D] + w[PX + p + 2 * D], acc);
      out[p] = acc;
    }
    if constexpr (!ADJ) {
      const float cf = L.is_first * own[0], cl = L.is_last * own[PX - 1];  // only the end lanes reach beyond the line
#pragma unroll
      for (int p = 0; p < 2 * D; ++p) {
        out[p] = fmaf((p < D) ? b1 + b2 : b2, cf, out[p]);
        out[PX - 1 - p] = fmaf((p < D) ? b1 + b2 : b2, cl, out[PX - 1 - p]);
      }
    } else {
      float h1 = 0.f, h2 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int p = 0; p < 2 * D; ++p) {
        if (p < D) {
          h1 += own[p];
          t1 += own[PX - 1 - p];
        }
        h2 += own[p];
        t2 += own[PX - 1 - p];
      }
      out[0] = fmaf(L.is_first, fmaf(b1, h1, b2 * h2), out[0]);
      out[PX - 1] = fmaf(L.is_last, fmaf(b1, t1, b2 * t2), out[PX - 1]);
    }
  }
}

// Starlet l1 value and sub-gradient with every pass in registers: the image alternates between the row-major layout
// (thread = row pu, samples pv .. pv + PX - 1) and the column-major one (thread = column cv, samples cu0 .. cu0 + PX - 1)
// through two LDS transposes per scale and direction; weights, coefficients and sub-gradients live in the row-major one.
template <int N, int PX, int JUSE>
__device__ __forceinline__ void starlet_l1_grad_dpp(const float img[PX], const float *Wf, const float *norms, float *qscr,
                                                    float lam_sc, float lam_hf, float *lds, int tid, float &l1,
                                                    float z[PX]) {
  constexpr int J = JUSE, TS = StarletLds<N>::TS, LPR = N / PX;
  float *bufA = lds, *bufB = lds + N * TS;
  const int pu_ = tid / LPR, pv_ = (tid % LPR) * PX;  // row-major: row, first column
  const int cv_ = tid / LPR, cu0_ = (tid % LPR) * PX; // column-major: column, first row
  LineLane L;
  L.lil = tid % LPR;
  L.is_first = (L.lil == 0) ? 1.f : 0.f;
  L.is_last = (L.lil == LPR - 1) ? 1.f : 0.f;
  l1 = 0.f;
  constexpr bool QREG = (J * PX <= 64);  // sub-gradients stay in registers when they fit
  float qreg[QREG ? J : 1][PX];
  float c[PX];
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    c[p] = img[p];
    z[p] = 0.f;
  }
  // transposes: row-major -> column-major through bufA, back through bufB (one barrier each: a buffer is rewritten
  // only after a barrier that every reader of its previous contents has passed)
  // Bank swizzle (64 x 64, 8 samples per thread): a row-major access puts lanes t % 8 = a at columns 8 a + p - banks 8 a + row,
  // so a and a + 4 collide - and a column-major one lanes a at rows 8 a + p - 65 * 8 a = 8 a (mod 32): the same collision.
  // Every LDS access of the 24 transposes of an iteration was a 2-way conflict (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS
  // = 0.97, profiles/r03_psf_sq_wave_time.txt).  Element (r, c) therefore sits at column c ^ 4 ((c >> 5) ^ (r >> 5)): the
  // upper half of the columns and the lower half of the rows shift by four banks - conflict-free in both directions (the
  // flip stays inside a thread's eight samples: same values in the same registers, only their LDS addresses differ).
  // The flip is the same for all eight samples of a thread (its columns share c >> 5, its rows r >> 5), so it costs no
  // per-element address arithmetic: a row-major thread swaps the two halves of its eight columns - two base addresses,
  // immediate offsets as before - and a column-major thread moves its one column by four.  (Computed per element, the flip
  // took the immediate offsets away: 20.0 against 14.8 us per iteration.)
  // MEASURED AND NOT KEPT (built with -DLC_STARLET_SWZ only): A / B on one MI355X, same run - 100 frames in the two-workgroup
  // form 14.9 us per iteration with the swizzle against 15.0 without (the convolution role is the longer of the two: a faster
  // starlet does not show), 256 frames in the one-workgroup form 28.0 - 28.2 against 26.4 - 26.6, 100 frames one workgroup each
  // 26.1 against 25.0: the 2-way conflicts of the transposes are NOT what the starlet waits for (its 16 LDS instructions per
  // transpose sit between register passes of ~300 vector instructions), and the second base address costs registers.
#ifdef LC_STARLET_SWZ   // (A / B build: make alt ALTFLAGS=-DLC_STARLET_SWZ)
  constexpr bool SWZ = (N == 64 && PX == 8);
#else
  constexpr bool SWZ = false;
#endif
  auto row_bases = [&](int pu, int pv, int &lo, int &hi) {   // samples p < 4 at lo + p, samples p >= 4 at hi + p
    const int k4 = SWZ ? ((((pv >> 5) ^ (pu >> 5)) & 1) << 2) : 0;
    lo = pu * TS + pv + k4;
    hi = pu * TS + pv - k4;
  };
  auto col_base = [&](int cu0, int cv) {                      // sample p at base + p * TS
    const int k4 = SWZ ? ((((cv >> 5) ^ (cu0 >> 5)) & 1) << 2) : 0;
    return cu0 * TS + (cv ^ k4);
  };
  auto to_columns = [&](const float (&v)[PX], float (&o)[PX]) {
    int pu = pu_, pv = pv_, cu0 = cu0_, cv = cv_;
    LC_LAUNDER(pu);
    LC_LAUNDER(pv);
    LC_LAUNDER(cu0);
    LC_LAUNDER(cv);
    int lo, hi;
    row_bases(pu, pv, lo, hi);
    const int cb = col_base(cu0, cv);
#pragma unroll
    for (int p = 0; p < PX; ++p) bufA[((SWZ && p >= 4) ? hi : lo) + p] = v[p];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PX; ++p) o[p] = bufA[cb + p * TS];
  };
  auto to_rows = [&](const float (&v)[PX], float (&o)[PX]) {
    int pu = pu_, pv = pv_, cu0 = cu0_, cv = cv_;
    LC_LAUNDER(pu);
    LC_LAUNDER(pv);
    LC_LAUNDER(cu0);
    LC_LAUNDER(cv);
    int lo, hi;
    row_bases(pu, pv, lo, hi);
    const int cb = col_base(cu0, cv);
#pragma unroll
    for (int p = 0; p < PX; ++p) bufB[cb + p * TS] = v[p];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PX; ++p) o[p] = bufB[((SWZ && p >= 4) ? hi : lo) + p];
  };
  static_for(std::make_integer_sequence<int, J>{}, [&](auto jc) {
    constexpr int j = decltype(jc)::value, d = 1 << j;
    // weights of this scale: requested before the passes so that the latency hides behind them
    float wj[PX];
    if (Wf) {
      int pu = pu_, pv = pv_;
      LC_LAUNDER(pu);
      LC_LAUNDER(pv);
      const float4 *wp = (const float4 *)(Wf + (size_t)j * N * N + (size_t)pu * N + pv);
#pragma unroll
      for (int q = 0; q < PX / 4; ++q) {
        const float4 w4 = wp[q];
        wj[4 * q] = w4.x;
        wj[4 * q + 1] = w4.y;
        wj[4 * q + 2] = w4.z;
        wj[4 * q + 3] = w4.w;
      }
    } else {
      const float nv = norms[j];
#pragma unroll
      for (int p = 0; p < PX; ++p) wj[p] = nv;
    }
    float r[PX], rc[PX], cc[PX], cn[PX];
    line_pass<N, PX, LPR, d, false>(c, L, r);    // rows
    to_columns(r, rc);
    line_pass<N, PX, LPR, d, false>(rc, L, cc);  // columns
    to_rows(cc, cn);
    const float lam = (j == 0) ? lam_hf : lam_sc;
    float q[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      const float w = c[p] - cn[p];
      const float lw = lam * wj[p];
      l1 = fmaf(lw, fabsf(w), l1);
      q[p] = (w > 0.f) ? lw : ((w < 0.f) ? -lw : 0.f);
      c[p] = cn[p];
    }
    if constexpr (QREG) {
#pragma unroll
      for (int p = 0; p < PX; ++p) qreg[j][p] = q[p];
    } else {
      float4 *qp = (float4 *)(qscr + (size_t)j * N * N + (size_t)pu_ * N + pv_);
#pragma unroll
      for (int qq = 0; qq < PX / 4; ++qq) qp[qq] = make_float4(q[4 * qq], q[4 * qq + 1], q[4 * qq + 2], q[4 * qq + 3]);
    }
  });
  // backward: z_J = 0; z_j = q_j + Col_j^T Row_j^T (z_{j+1} - q_j)   (the two adjoint passes commute)
  static_for(std::make_integer_sequence<int, J>{}, [&](auto jc) {
    constexpr int j = J - 1 - decltype(jc)::value, d = 1 << j;
    float q[PX];
    if constexpr (QREG) {
#pragma unroll
      for (int p = 0; p < PX; ++p) q[p] = qreg[j][p];
    } else {
      const float4 *qp = (const float4 *)(qscr + (size_t)j * N * N + (size_t)pu_ * N + pv_);
#pragma unroll
      for (int qq = 0; qq < PX / 4; ++qq) {
        const float4 v4 = qp[qq];
        q[4 * qq] = v4.x;
        q[4 * qq + 1] = v4.y;
        q[4 * qq + 2] = v4.z;
        q[4 * qq + 3] = v4.w;
      }
    }
    float y[PX], rt[PX], yc[PX], ct[PX], back[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) y[p] = z[p] - q[p];
    line_pass<N, PX, LPR, d, true>(y, L, rt);    // Row^T
    to_columns(rt, yc);
    line_pass<N, PX, LPR, d, true>(yc, L, ct);   // Col^T
    to_rows(ct, back);
#pragma unroll
    for (int p = 0; p < PX; ++p) z[p] = q[p] + back[p];
  });
}

template <int N, int PX, int JUSE = ilog2(N)>
__device__ __forceinline__ void starlet_l1_grad(const float img[PX], const float *Wf, const float *norms, float *qscr,
                                                float lam_sc, float lam_hf, float *lds, int tid, float &l1,
                                                float z[PX]) {
  constexpr int LPR = N / PX;
  if constexpr ((LPR == 4 || LPR == 8 || LPR == 16) && PX >= 2 && PX <= 8)  // PX = 16 (1024 threads, 128 registers): the LDS form spills less
    starlet_l1_grad_dpp<N, PX, JUSE>(img, Wf, norms, qscr, lam_sc, lam_hf, lds, tid, l1, z);
  else
    starlet_l1_grad_lds<N, PX, JUSE>(img, Wf, norms, qscr, lam_sc, lam_hf, lds, tid, l1, z);
}

}  // namespace lc
