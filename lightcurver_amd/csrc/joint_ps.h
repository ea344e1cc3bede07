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
  // persistent form (PERSIST = true): T AdaBelief iterations of this epoch's own parameters inside one launch
  int T;
  const float *sched;                  // [T][3] learning rate and bias corrections of every iteration (host, adabelief_schedule)
  lc_adabelief_cfg ab;
  int free_a, free_dx, free_dy, free_mean;
  float lam_pos_ps;
  float *par_a, *par_dx, *par_dy, *par_mean;  // [E*M], [E], [E], [E]: read at the start, written at the end
  float *pm_a, *ps_a, *pm_dx, *ps_dx, *pm_dy, *ps_dy, *pm_mean, *ps_mean;  // AdaBelief moments, same shapes
  float *hist_e;                       // [E][T] this epoch's share of the loss before every update
  // return_param_history: rows [T][phist_P] of the device-resident history; offsets of the free blocks in a row (-1: fixed)
  float *phist;
  int phist_P, poff_a, poff_dx, poff_dy, poff_mean;
  int e_off;  // first epoch of this launch (a launch over a range of the epochs: batched star photometry, two halves on two streams)
};

// PERSIST: when every free parameter belongs to one epoch (fluxes, shifts, sky levels; the shared positions c_x, c_y held
// fixed - the reference's default star photometry leaves them free and takes the launch-per-iteration form - and no
// flux-uniformity / point-source / prior term) the epochs are independent fits, and the whole
// optimisation loop runs inside this kernel: the PSF tile stays in LDS, the parameters and their AdaBelief moments in
// LDS / registers, and the loss history leaves as one value per epoch and iteration (summed over the epochs afterwards).
// Same arithmetic per iteration as the launch-per-iteration form (same filters, same reductions, same update).
template <int N, int SS, bool PERSIST = false>
__global__ __launch_bounds__(kPsThreads) __attribute__((amdgpu_waves_per_eu((N <= 64 && !PERSIST) ? 4 : 1, (N <= 64 && !PERSIST) ? 4 : 8))) void joint_ps_kernel(JointPsArgs P) {
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
  const int e = blockIdx.x + P.e_off, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, M = A.M;
  const float c0 = (N - 1) * 0.5f, c_off = c0 - (float)((N - 1) / 2);
  const float al = A.alpha[e] * 0.017453292519943295f;
  const float ca = cosf(al), sa = sinf(al);
  float dxe = A.dx[e], dye = A.dy[e], meane = A.mean[e];
  const float *se = P.psf + (size_t)e * N * N;
  // (16-byte loads: a quarter of the requests; the padded LDS rows take the four values one by one)
  for (int k4 = tid; k4 < N * N / 4; k4 += kPsThreads) {
    const float4 v = ((const float4 *)se)[k4];
    float *d = Sx + ((4 * k4) / N) * TS + (4 * k4) % N;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  float *Fe = P.F + (size_t)e * M * 3 * nn;
  // persistent form: parameters [a_0 .. a_{M-1}, dx, dy, mean] and their moments, one thread each
  __shared__ float PP[kMaxSources + 3];
  float my_p = 0.f, my_m = 0.f, my_s = 0.f;
  bool my_free = false;
  if constexpr (PERSIST) {
    if (tid < M + 3) {
      const int k = tid - M;  // < 0: flux tid
      const float *src = (k < 0) ? P.par_a + e * M + tid : (k == 0 ? P.par_dx + e : (k == 1 ? P.par_dy + e : P.par_mean + e));
      const float *sm = (k < 0) ? P.pm_a + e * M + tid : (k == 0 ? P.pm_dx + e : (k == 1 ? P.pm_dy + e : P.pm_mean + e));
      const float *sv = (k < 0) ? P.ps_a + e * M + tid : (k == 0 ? P.ps_dx + e : (k == 1 ? P.ps_dy + e : P.ps_mean + e));
      my_p = *src;
      my_m = *sm;
      my_s = *sv;
      my_free = (k < 0) ? P.free_a != 0 : (k == 0 ? P.free_dx != 0 : (k == 1 ? P.free_dy != 0 : P.free_mean != 0));
      PP[tid] = my_p;
    }
  }
  const int n_it = PERSIST ? P.T : 1;
  for (int iter = 0; iter < n_it; ++iter) {
  if constexpr (PERSIST) {
    __syncthreads();  // parameters of this iteration in PP; the previous iteration's readers of TOT / R are done
    dxe = PP[M];
    dye = PP[M + 1];
    meane = PP[M + 2];
  }
  auto flux = [&](int i) { return PERSIST ? PP[i] : A.a[e * M + i]; };
  // One point source (every star-photometry fit): the filter outputs of a thread's pixels stay in its registers from the
  // column pass to the residuals, instead of travelling through the global scratch F and back (3 n^2 floats each way per
  // epoch and iteration: half of the memory traffic of a batch of thousands of epochs).
  constexpr int LSF = 4, NITF = ((n / LSF) * n + kPsThreads - 1) / kPsThreads;
  const bool single = (M == 1);
  float fvr[NITF][LSF], fxr[NITF][LSF], fyr[NITF][LSF];
#pragma unroll
  for (int t = 0; t < NITF; ++t)
#pragma unroll
    for (int jj = 0; jj < LSF; ++jj) fvr[t][jj] = fxr[t][jj] = fyr[t][jj] = 0.f;
  for (int i = 0; i < M; ++i) {
    __syncthreads();  // PSF tile loaded; previous source's passes done with TAP / R
    if (tid < 2 * NT) {
      const int ax = tid / NT, k = tid % NT;
      const int gi = (A.group ? A.group[e] : 0) * M + i;  // (batched star photometry: the positions of this epoch's star)
      const float X = SS * (ca * A.cx[gi] - sa * A.cy[gi] + dxe), Y = SS * (sa * A.cx[gi] + ca * A.cy[gi] + dye);
      float tap, dtap;
      int bq;
      ps_tap<N, SS, NT>((ax == 0 ? X : Y) + c_off, k, tap, dtap, bq);
      TAP[(2 * ax) * NT + k] = tap;
      TAP[(2 * ax + 1) * NT + k] = dtap;
      if (k == 0) BQ[ax] = bq;
    }
    __syncthreads();
    const int bqx = BQ[0], bqy = BQ[1];
    // (the taps of the source in registers for both passes: read from LDS per multiply-add they were three of every five
    //  LDS reads of this kernel, which is what a batch of thousands of epochs waits for; same values, same order of the sums)
    // value and derivative tap of an axis as one packed operand: a window sample enters both filters with one v_pk_fma_f32
    // (the same two fused multiply-adds, one instruction instead of two)
    lc_v2f txd[NT], tyd[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      txd[k] = (lc_v2f){TAP[k], TAP[NT + k]};
      tyd[k] = (lc_v2f){TAP[2 * NT + k], TAP[3 * NT + k]};
    }
    // row pass fused with the column down-sampling: R[r][a] = sum_k tx[k] s[r][SS (a - bqx) - k].  A thread takes a strip
    // of LS consecutive outputs of a row and reads their common window of the PSF row once (19 LDS reads for 4 outputs
    // instead of 52); every output adds its taps in the same order as before.
    constexpr int LS = 4, WL = SS * (LS - 1) + NT;
    static_assert(n % LS == 0, "strips");
    for (int it = tid; it < N * (n / LS); it += kPsThreads) {
      const int r = it / (n / LS), a0 = (it % (n / LS)) * LS;
      const int w0 = SS * (a0 - bqx) - (NT - 1);
      float win[WL];
#pragma unroll
      for (int q = 0; q < WL; ++q) {
        const int v = w0 + q;
        win[q] = (v >= 0 && v < N) ? Sx[r * TS + v] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < LS; ++j) {
        lc_v2f acc = {0.f, 0.f};  // (value, x-derivative)
#pragma unroll
        for (int k = 0; k < NT; ++k) acc = pk_fma(txd[k], pk_bcast(win[SS * j + NT - 1 - k]), acc);
        R[r * RS + a0 + j] = acc.x;
        Rx[r * RS + a0 + j] = acc.y;
      }
    }
    __syncthreads();
    // column pass fused with the row down-sampling: value, d/dX and d/dY filter outputs of source i at every data pixel;
    // a thread takes LS consecutive data rows of one column (lanes = consecutive columns)
    static_assert(LS == LSF, "strip length");
#pragma unroll
    for (int t = 0; t < NITF; ++t) {
      const int it = tid + t * kPsThreads;
      if (it >= (n / LS) * n) break;
      const int I0 = (it / n) * LS, a = it % n;
      const int w0 = SS * (I0 - bqy) - (NT - 1);
      float wr[WL], wx[WL];
#pragma unroll
      for (int q = 0; q < WL; ++q) {
        const int r = w0 + q;
        const bool ok = (r >= 0 && r < N);
        wr[q] = ok ? R[r * RS + a] : 0.f;
        wx[q] = ok ? Rx[r * RS + a] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < LS; ++j) {
        lc_v2f fvy = {0.f, 0.f};  // (value, y-derivative)
        float fx = 0.f;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const float rv = wr[SS * j + NT - 1 - k], rx = wx[SS * j + NT - 1 - k];
          fvy = pk_fma(tyd[k], pk_bcast(rv), fvy);
          fx = fmaf(tyd[k].x, rx, fx);
        }
        const float fv = fvy.x, fy = fvy.y;
        const int px = (I0 + j) * n + a;
        if (single) {
          fvr[t][j] = fv;
          fxr[t][j] = fx;
          fyr[t][j] = fy;
        } else {
          Fe[((size_t)i * 3 + 0) * nn + px] = fv;
          Fe[((size_t)i * 3 + 1) * nn + px] = fx;
          Fe[((size_t)i * 3 + 2) * nn + px] = fy;
        }
      }
    }
  }
  __syncthreads();  // this workgroup's filter outputs are visible to all of its threads
  // residuals and reductions
  float vals[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) vals[q] = 0.f;
  const float *de = A.data + (size_t)e * nn, *we = A.wgt + (size_t)e * nn;
  if (single) {  // the thread's own pixels, filter outputs from registers
    const float a0f = (A.mode == 2) ? 0.f : flux(0);
#pragma unroll
    for (int t = 0; t < NITF; ++t) {
      const int it = tid + t * kPsThreads;
      if (it >= (n / LSF) * n) break;
      const int I0 = (it / n) * LSF, a = it % n;
#pragma unroll
      for (int jj = 0; jj < LSF; ++jj) {
        const int px = (I0 + jj) * n + a;
        const float w = we[px], fv = fvr[t][jj];
        if (A.mode == 2) {
          vals[0] = fmaf(w * fv, fv, vals[0]);
          continue;
        }
        const float model = fmaf(a0f, fv, meane);
        if (A.model_out) A.model_out[(size_t)e * nn + px] = model;
        const float res = model - de[px], rw = w * res;
        vals[0] = fmaf(rw, res, vals[0]);
        vals[1] += rw;
        if (A.mode == 0) {
          vals[4] = fmaf(rw, fv, vals[4]);
          vals[5] = fmaf(rw, fxr[t][jj], vals[5]);
          vals[6] = fmaf(rw, fyr[t][jj], vals[6]);
        }
      }
    }
  }
  for (int px = tid; px < (single ? 0 : nn); px += kPsThreads) {
    const float w = we[px];
    if (A.mode == 2) {
      const float fv = Fe[((size_t)A.isrc * 3) * nn + px];
      vals[0] = fmaf(w * fv, fv, vals[0]);
      continue;
    }
    float model = meane;
    for (int i = 0; i < M; ++i) model = fmaf(flux(i), Fe[((size_t)i * 3) * nn + px], model);
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
  if constexpr (PERSIST) {
    // gradient of this thread's parameter (the rules of gm_small_blocks, joint_gm.h), loss share, AdaBelief step
    if (tid < M + 3) {
      const int k = tid - M;
      float g;
      if (k < 0) {
        g = TOT[4 + 3 * tid];
        if (P.lam_pos_ps != 0.f && my_p < 0.f) g -= P.lam_pos_ps;
      } else if (k == 2) {
        g = TOT[1];
      } else {
        g = 0.f;
        for (int i = 0; i < M; ++i) g += SS * (PP[i] * TOT[(k == 0 ? 5 : 6) + 3 * i]);
      }
      if (tid == 0) {
        float loss = 0.5f * TOT[0];
        if (P.lam_pos_ps != 0.f)
          for (int i = 0; i < M; ++i) loss += (PP[i] < 0.f) ? -P.lam_pos_ps * PP[i] : 0.f;
        P.hist_e[(size_t)e * P.T + iter] = loss;
      }
      __builtin_amdgcn_wave_barrier();  // (threads 0 .. M + 2 sit in one wave: everyone has read PP before anyone rewrites it)
      if (my_free) {
        adabelief_step(my_p, my_m, my_s, g, P.sched[3 * iter], P.sched[3 * iter + 1], P.sched[3 * iter + 2], P.ab);
        PP[tid] = my_p;
        if (P.phist) {
          const int off = (k < 0) ? P.poff_a + e * M + tid : ((k == 0 ? P.poff_dx : (k == 1 ? P.poff_dy : P.poff_mean)) + e);
          P.phist[(size_t)iter * P.phist_P + off] = my_p;
        }
      }
    }
    continue;
  }
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
  }  // iterations
  if constexpr (PERSIST) {
    if (tid < M + 3) {
      const int k = tid - M;
      float *dst = (k < 0) ? P.par_a + e * M + tid : (k == 0 ? P.par_dx + e : (k == 1 ? P.par_dy + e : P.par_mean + e));
      float *dm = (k < 0) ? P.pm_a + e * M + tid : (k == 0 ? P.pm_dx + e : (k == 1 ? P.pm_dy + e : P.pm_mean + e));
      float *dv = (k < 0) ? P.ps_a + e * M + tid : (k == 0 ? P.ps_dx + e : (k == 1 ? P.ps_dy + e : P.ps_mean + e));
      *dst = my_p;
      *dm = my_m;
      *dv = my_s;
    }
  }
}

// hist[t0 + it] = sum over the epochs of hist_e[e][it] (lanes stride over the epochs, fixed combine order)
__global__ void joint_ps_hist_kernel(int E, int T, const float *hist_e, float *hist) {
  const int it = blockIdx.x, lane = threadIdx.x;
  float acc = 0.f;
  for (int e = lane; e < E; e += 64) acc += hist_e[(size_t)e * T + it];
  acc = wave_sum_shfl(acc);
  if (lane == 0) hist[it] = acc;
}

template <int N, int SS>
constexpr int joint_ps_lds_bytes() {
  constexpr int n = N / SS, NT = ntaps(SS);
  return (N * (N + 1) + 2 * N * (n + 1) + 4 * NT + (kPsThreads / 64 + 1) * (4 + 3 * kMaxSources)) * (int)sizeof(float);
}

}  // namespace lc
