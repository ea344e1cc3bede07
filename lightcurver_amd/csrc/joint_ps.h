// Joint forward model without a background: the scene of an epoch is only its M point sources (h == 0 and fixed), the
// default of the reference's star photometry (star_photometry.py:74-87, config `star_photometry_starlet_global_background:
// false`).  Then  f_e = sum_i a_i D_ss[ G(X_i, Y_i) (*) s_e ] + mean_e  is a separable 13-tap Gaussian filtering of the
// epoch's narrow PSF, and chi2, d/da, d/dX, d/dY, d/dmean follow from the value and the two derivative filters: no FFT
// and no adjoint convolution.  One workgroup per epoch; same outputs as joint_epoch_kernel (joint_kernels.h), so the
// reduction and update kernels are shared.  G is sampled where the FFT path has it non-negligible (|t - centre| <= kRg,
// relative tail 1e-9) and, like there, only on the N x N scene grid.
#pragma once
#include "joint_kernels.h"

namespace lc {

constexpr int kPsThreads = 256;

// taps of one (source, axis): Phi(m) = sum_{dv < SS} phi(m + dv), phi(t) = N(t; delta, sigma) for |t - round(delta)| <= kRg
// and t + c inside the scene grid; dPhi = d/d delta.  m = bq * SS + k.
template <int N, int SS, int NT>
__device__ __forceinline__ void ps_tap(float delta, int k, float &tap, float &dtap, int &bq) {
  constexpr int c = (N - 1) / 2;
  const float dcl = fminf(fmaxf(delta, -2.0f * N), 2.0f * N);  // a runaway position must not overflow the indices
  const int o = (int)nearbyintf(dcl);
  const int base = o - kRg - (SS - 1);
  int q = base / SS;
  if (q * SS > base) --q;
  bq = q;
  const int m = q * SS + k;
  const float inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm = 0.3989422804014327f / kSigmaG;
  float a = 0.f, d = 0.f;
#pragma unroll
  for (int dv = 0; dv < SS; ++dv) {
    const int t = m + dv;
    if (t >= o - kRg && t <= o + kRg && t + c >= 0 && t + c < N) {
      const float x = (float)t - delta;
      const float p = nrm * expf(-0.5f * x * x * inv_s2);
      a += p;
      d += p * x * inv_s2;
    }
  }
  tap = a;
  dtap = d;
}

struct JointPsArgs {
  JointArgs J;
  const float *psf;  // [E][N][N] narrow PSFs
  float *F;          // [E][M][3][n*n] scratch: value, d/dX, d/dY filter outputs of every source
};

template <int N, int SS>
__global__ __launch_bounds__(kPsThreads) void joint_ps_kernel(JointPsArgs P) {
  constexpr int n = N / SS, NT = ntaps(SS), TS = N + 1, RS = n + 1, nn = n * n, NWV = kPsThreads / 64;
  constexpr int NQ = 4 + 3 * kMaxSources;
  const JointArgs &A = P.J;
  extern __shared__ float plds[];
  float *Sx = plds;                    // [N][TS]  PSF of the epoch
  float *R = Sx + N * TS;              // [N][RS]  row pass, value taps
  float *Rx = R + N * RS;              // [N][RS]  row pass, derivative taps
  float *TAP = Rx + N * RS;            // [4][NT]  tx, dtx, ty, dty
  float *RED = TAP + 4 * NT;           // [NWV + 1][NQ]
  __shared__ int BQ[2];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, M = A.M;
  const float c0 = (N - 1) * 0.5f, c_off = c0 - (float)((N - 1) / 2);
  const float al = A.alpha[e] * 0.017453292519943295f;
  const float ca = cosf(al), sa = sinf(al), dxe = A.dx[e], dye = A.dy[e], meane = A.mean[e];
  const float *se = P.psf + (size_t)e * N * N;
  for (int k = tid; k < N * N; k += kPsThreads) Sx[(k / N) * TS + (k % N)] = se[k];
  float *Fe = P.F + (size_t)e * M * 3 * nn;
  for (int i = 0; i < M; ++i) {
    __syncthreads();  // PSF tile loaded; previous source's passes done with TAP / R
    if (tid < 2 * NT) {
      const int ax = tid / NT, k = tid % NT;
      const float X = SS * (ca * A.cx[i] - sa * A.cy[i] + dxe), Y = SS * (sa * A.cx[i] + ca * A.cy[i] + dye);
      float tap, dtap;
      int bq;
      ps_tap<N, SS, NT>((ax == 0 ? X : Y) + c_off, k, tap, dtap, bq);
      TAP[(2 * ax) * NT + k] = tap;
      TAP[(2 * ax + 1) * NT + k] = dtap;
      if (k == 0) BQ[ax] = bq;
    }
    __syncthreads();
    const int bqx = BQ[0], bqy = BQ[1];
    // row pass fused with the column down-sampling: R[r][a] = sum_k tx[k] s[r][SS (a - bqx) - k]
    for (int it = tid; it < N * n; it += kPsThreads) {
      const int r = it / n, a = it % n;
      const int v0 = SS * (a - bqx);
      float acc = 0.f, accd = 0.f;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int v = v0 - k;
        const float sv = (v >= 0 && v < N) ? Sx[r * TS + v] : 0.f;
        acc = fmaf(TAP[k], sv, acc);
        accd = fmaf(TAP[NT + k], sv, accd);
      }
      R[r * RS + a] = acc;
      Rx[r * RS + a] = accd;
    }
    __syncthreads();
    // column pass fused with the row down-sampling: value, d/dX and d/dY filter outputs of source i at every data pixel
    for (int px = tid; px < nn; px += kPsThreads) {
      const int I = px / n, a = px % n;
      const int r0 = SS * (I - bqy);
      float fv = 0.f, fx = 0.f, fy = 0.f;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int r = r0 - k;
        const bool ok = (r >= 0 && r < N);
        const float rv = ok ? R[r * RS + a] : 0.f, rx = ok ? Rx[r * RS + a] : 0.f;
        fv = fmaf(TAP[2 * NT + k], rv, fv);
        fy = fmaf(TAP[3 * NT + k], rv, fy);
        fx = fmaf(TAP[2 * NT + k], rx, fx);
      }
      Fe[((size_t)i * 3 + 0) * nn + px] = fv;
      Fe[((size_t)i * 3 + 1) * nn + px] = fx;
      Fe[((size_t)i * 3 + 2) * nn + px] = fy;
    }
  }
  __syncthreads();  // this workgroup's filter outputs are visible to all of its threads
  // residuals and reductions
  float vals[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) vals[q] = 0.f;
  const float *de = A.data + (size_t)e * nn, *we = A.wgt + (size_t)e * nn;
  for (int px = tid; px < nn; px += kPsThreads) {
    const float w = we[px];
    if (A.mode == 2) {
      const float fv = Fe[((size_t)A.isrc * 3) * nn + px];
      vals[0] = fmaf(w * fv, fv, vals[0]);
      continue;
    }
    float model = meane;
    for (int i = 0; i < M; ++i) model = fmaf(A.a[e * M + i], Fe[((size_t)i * 3) * nn + px], model);
    if (A.model_out) A.model_out[(size_t)e * nn + px] = model;
    const float res = model - de[px], rw = w * res;
    vals[0] = fmaf(rw, res, vals[0]);
    vals[1] += rw;
    if (A.mode == 0) {
#pragma unroll
      for (int i = 0; i < kMaxSources; ++i)
        if (i < M) {
          vals[4 + 3 * i] = fmaf(rw, Fe[((size_t)i * 3 + 0) * nn + px], vals[4 + 3 * i]);
          vals[5 + 3 * i] = fmaf(rw, Fe[((size_t)i * 3 + 1) * nn + px], vals[5 + 3 * i]);
          vals[6 + 3 * i] = fmaf(rw, Fe[((size_t)i * 3 + 2) * nn + px], vals[6 + 3 * i]);
        }
    }
  }
  const int nq = (A.mode == 0) ? 4 + 3 * M : 2;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q < nq) {
      const float s = wave_sum(vals[q]);
      if (lane == 0) RED[wid * NQ + q] = s;
    }
  }
  __syncthreads();
  float *TOT = RED + NWV * NQ;
  if (tid < nq) {
    float acc = 0.f;
    for (int w = 0; w < NWV; ++w) acc += RED[w * NQ + tid];
    TOT[tid] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    if (A.mode == 2) {
      A.fisher_out[e * M + A.isrc] = 1.0f / sqrtf(TOT[0]);
      return;
    }
    A.chi2_e[e] = TOT[0];
    if (A.mode == 1) return;
    A.g_mean[e] = TOT[1];
    float gdx = 0.f, gdy = 0.f;
    for (int i = 0; i < M; ++i) {
      const float ai = A.a[e * M + i];
      const float gX = ai * TOT[5 + 3 * i], gY = ai * TOT[6 + 3 * i];  // d chi2/2 / d(position in high-res pixels)
      A.g_a[e * M + i] = TOT[4 + 3 * i];
      gdx += SS * gX;
      gdy += SS * gY;
      A.g_cx_e[e * M + i] = SS * (ca * gX + sa * gY);
      A.g_cy_e[e * M + i] = SS * (ca * gY - sa * gX);
    }
    A.g_dx[e] = gdx;
    A.g_dy[e] = gdy;
  }
}

template <int N, int SS>
constexpr int joint_ps_lds_bytes() {
  constexpr int n = N / SS, NT = ntaps(SS);
  return (N * (N + 1) + 2 * N * (n + 1) + 4 * NT + (kPsThreads / 64 + 1) * (4 + 3 * kMaxSources)) * (int)sizeof(float);
}

}  // namespace lc
