// Stage A of build_psf (STARRED Optimizer(method='l-bfgs-b') on Moffat + a, x0, y0 with the pixel grid fixed: reference call
// site lightcurver/processes/psf_modelling.py:164-171, `n_iter_analytic`) with the optimiser ON THE DEVICE: every frame is
// an independent bounded L-BFGS problem in D = 4 + 3 S <= 64 unknowns, one wave per frame, one lane per unknown.  The
// kernel below is the whole state machine of csrc/lbfgs_host.h (projected gradient test, two-loop recursion on the free
// variables, steepest-descent restart, Armijo back-tracking on the projected step, curvature-guarded history update) in
// double precision; between two calls the library evaluates loss and gradient of all frames at the trial points the
// kernel wrote into the parameter blocks.  The host only counts rounds and reads one integer every few of them.
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int kPsfLbfgsMem = 10;

struct PsfLbfgsDev {
  int F, S, D, maxiter;
  double gtol, ftol;
  double *x, *g, *d, *xt, *lo, *hi;  // [F][D]
  double *Sm, *Ym;                   // [F][mem][D], oldest pair first
  double *rho;                       // [F][mem]
  double *f, *alpha;                 // [F]
  int *state, *ls, *iters, *m;       // [F]; state 0 = needs a direction, 1 = in line search, 2 = done
  float *moffat, *stars;             // parameter blocks the evaluation reads: [F][4], [F][S][4]
  const float *o_loss, *o_gmoffat, *o_gstars;
  int *n_active;                     // [rounds] frames still working after each round
};

__device__ __forceinline__ double lb_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double lb_wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

// mode 0: the evaluation was at x itself (start); 1: at the trial points; 2: leave x in the parameter blocks (end)
__global__ __launch_bounds__(64) void psf_lbfgs_step_kernel(PsfLbfgsDev P, int mode, int round) {
  const int fr = blockIdx.x, i = threadIdx.x, D = P.D, S = P.S;
  const bool on = i < D;
  const size_t o = (size_t)fr * D + (on ? i : 0);
  // parameter slot of unknown i: Moffat (4) | a (S) | x0 (S) | y0 (S)
  float *slot = nullptr;
  const float *gslot = nullptr;
  if (on) {
    if (i < 4) {
      slot = P.moffat + fr * 4 + i;
      gslot = P.o_gmoffat + fr * 4 + i;
    } else {
      const int c = (i - 4) / S, s = (i - 4) % S;
      slot = P.stars + ((size_t)fr * S + s) * 4 + c;
      gslot = P.o_gstars + ((size_t)fr * S + s) * 4 + c;
    }
  }
  double x = on ? P.x[o] : 0.0, g = on ? P.g[o] : 0.0, d = on ? P.d[o] : 0.0;
  const double lo = on ? P.lo[o] : 0.0, hi = on ? P.hi[o] : 0.0;
  double f = P.f[fr], alpha = P.alpha[fr];
  int state = P.state[fr], ls = P.ls[fr], iters = P.iters[fr], m = P.m[fr];
  double *Sf = P.Sm + (size_t)fr * kPsfLbfgsMem * D, *Yf = P.Ym + (size_t)fr * kPsfLbfgsMem * D, *rf = P.rho + fr * kPsfLbfgsMem;
  if (mode == 2) {
    if (on) *slot = (float)x;
    return;
  }
  double r[kPsfLbfgsMem];  // 1 / (s . y) of the stored pairs: wave-uniform, kept in registers while this call changes them
#pragma unroll
  for (int k = 0; k < kPsfLbfgsMem; ++k) r[k] = rf[k];
  if (mode == 0) {
    f = (double)P.o_loss[fr];
    g = on ? (double)*gslot : 0.0;
    state = isfinite(f) ? 0 : 2;
  } else if (state == 1) {
    const double ft = (double)P.o_loss[fr], gt = on ? (double)*gslot : 0.0, xt = on ? P.xt[o] : 0.0;
    const double dec = lb_wave_sum(g * (xt - x));  // g . (projected step)
    if (isfinite(ft) && ft <= f + 1e-4 * dec) {
      const double s = xt - x, y = gt - g;
      const double sy = lb_wave_sum(s * y), ss = lb_wave_sum(s * s), yy = lb_wave_sum(y * y);
      const double fold = f;
      x = xt;
      g = gt;
      f = ft;
      ++iters;
      if (sy > 1e-10 * sqrt(ss * yy) && sy > 0.0) {
        if (m == kPsfLbfgsMem) {  // drop the oldest pair (every lane moves its own column of the history)
#pragma unroll
          for (int k = 0; k + 1 < kPsfLbfgsMem; ++k) {
            if (on) {
              Sf[k * D + i] = Sf[(k + 1) * D + i];
              Yf[k * D + i] = Yf[(k + 1) * D + i];
            }
            r[k] = r[k + 1];
          }
          m = kPsfLbfgsMem - 1;
        }
        if (on) {
          Sf[m * D + i] = s;
          Yf[m * D + i] = y;
        }
#pragma unroll
        for (int k = 0; k < kPsfLbfgsMem; ++k)
          if (k == m) r[k] = 1.0 / sy;
        ++m;
      }
      state = 0;
      if (fabs(fold - f) <= P.ftol * fmax(fmax(fabs(fold), fabs(f)), 1.0)) state = 2;
    } else {
      alpha *= 0.5;
      if (++ls >= 25) state = 2;  // line search failed: keep the last accepted point
    }
  }
  if (state == 0) {
    if (iters >= P.maxiter) {
      state = 2;
    } else {
      const bool act = on && ((x <= lo && g > 0.0) || (x >= hi && g < 0.0));
      double q = (on && !act) ? g : 0.0;
      const double pgmax = lb_wave_max(fabs(q));
      if (pgmax < P.gtol * fmax(1.0, fabs(f))) {
        state = 2;
      } else {
        double a[kPsfLbfgsMem];
#pragma unroll
        for (int k = kPsfLbfgsMem - 1; k >= 0; --k) {
          a[k] = 0.0;
          if (k < m) {
            const double sk = on ? Sf[k * D + i] : 0.0, yk = on ? Yf[k * D + i] : 0.0;
            a[k] = r[k] * lb_wave_sum(sk * q);
            q -= a[k] * yk;
          }
        }
        if (m > 0) {
          const double sk = on ? Sf[(m - 1) * D + i] : 0.0, yk = on ? Yf[(m - 1) * D + i] : 0.0;
          q *= lb_wave_sum(sk * yk) / lb_wave_sum(yk * yk);
        }
#pragma unroll
        for (int k = 0; k < kPsfLbfgsMem; ++k) {
          if (k < m) {
            const double sk = on ? Sf[k * D + i] : 0.0, yk = on ? Yf[k * D + i] : 0.0;
            const double beta = r[k] * lb_wave_sum(yk * q);
            q += sk * (a[k] - beta);
          }
        }
        d = (on && !act) ? -q : 0.0;
        double gdot = lb_wave_sum(d * g);
        if (!(gdot < 0.0)) {  // not a descent direction: restart from steepest descent
          m = 0;
          d = (on && !act) ? -g : 0.0;
          gdot = lb_wave_sum(d * g);
        }
        if (m == 0) {
          const double nrm = lb_wave_sum(d * d);
          alpha = fmin(1.0, 1.0 / sqrt(fmax(nrm, 1e-300)));
        } else {
          alpha = 1.0;
        }
        ls = 0;
        state = 1;
      }
    }
  }
  // next evaluation point: the trial of a frame in line search, the accepted point of a finished one
  const double xt = (state == 1) ? fmin(fmax(x + alpha * d, lo), hi) : x;
  if (on) {
    P.x[o] = x;
    P.g[o] = g;
    P.d[o] = d;
    P.xt[o] = xt;
    *slot = (float)xt;
  }
  if (i == 0) {
    P.f[fr] = f;
    P.alpha[fr] = alpha;
    P.state[fr] = state;
    P.ls[fr] = ls;
    P.iters[fr] = iters;
    P.m[fr] = m;
#pragma unroll
    for (int k = 0; k < kPsfLbfgsMem; ++k) rf[k] = r[k];
    if (state != 2) atomicAdd(P.n_active + round, 1);
  }
}

}  // namespace lc
