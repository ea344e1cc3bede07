/* CPU restatement (plain C, OpenMP over epochs) of the point-source-only joint fit: the hot loop of the reference's star
 * photometry.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py): the "port" CPU baseline bench.py times beside the
 * star-photometry entry (cpu_baseline.kind = "port") and a second, independent checker in tests/.  The product path never
 * links, loads or calls it.  PARITY UNPINNED against STARRED itself: it restates the same frozen
 * SPEC as oracle/model.py, and tests/test_joint_ps_cpu_port_cpu.py pins it to that float64 oracle.
 *
 * What it restates (reference call site: lightcurver/processes/star_photometry.py:66-122 - setup_model with one point
 * source, background fixed at zero (config.yaml:258 star_photometry_starlet_global_background: false), Optimizer 'adabelief'):
 *     f_e = D_ss[ s_e (*) sum_i a_ei G(c_i + (dx_e, dy_e)) ] + mean_e         (oracle/model.py deconv_model, h = 0, alpha = 0)
 *     L   = 1/2 sum_e sum_pix (d - f)^2 / sigma^2                              (deconv_loss, chi2 term)
 *     AdaBelief (optax: b1 .9, b2 .999, eps 1e-16, eps_root 1e-16, lr_t = lr0 * 0.99^(t/10)) on a, c_x, c_y, dx, dy
 *     (and mean when asked)                                                    (oracle/optim.py adabelief)
 * in the direct separable form the HIP kernel (csrc/joint_ps.h) also uses: the convolution of the epoch's narrow PSF with
 * a FWHM-2 Gaussian at a sub-pixel position is a 1-D row pass fused with the column down-sampling, then a 1-D column pass
 * fused with the row down-sampling; the derivatives with respect to the position ride on derivative taps, so the whole
 * gradient is a handful of dot products with the weighted residual (no adjoint passes).  oracle/model.py takes the other
 * route (full-frame Gaussian raster, FFT convolution, autograd).  The Gaussian is truncated at +-KRG samples around its
 * rounded centre (exp(-29) relative).
 *
 * Two builds (oracle/Makefile): libjointpscpu.so with real = float (the arithmetic of the HIP path: the CPU baseline) and
 * libjointpscpu_f64.so with real = double (the checker).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef JPS_CPU_DOUBLE
typedef double real;
#define R(x) x
#define EXP exp
#define SQRT sqrt
#define NEARBYINT nearbyint
#else
typedef float real;
#define R(x) x##f
#define EXP expf
#define SQRT sqrtf
#define NEARBYINT nearbyintf
#endif

#define KRG 6
#define SIGMA_G R(0.84932180028801907)
#define MAXT (2 * KRG + 1)
#define MAXSS 4
#define MAXPHI (MAXT + MAXSS - 1)

/* Down-sampling taps of one axis: with o = round(pos), Phi[j + ss - 1] = sum_{q < ss, 0 <= j + q <= 2 KRG} g(o - KRG + j + q - pos)
 * for j = -(ss - 1) .. 2 KRG, and the same sums of dg/dpos.  A data sample J then is sum_j Phi[j] * line[ss J + cr - o + KRG - j]. */
static void taps(real pos, int ss, int *o, real *phi, real *dphi) {
  const real inv_s2 = R(1.0) / (SIGMA_G * SIGMA_G), nrm = R(0.3989422804014327) / SIGMA_G;
  real g[MAXT], dg[MAXT];
  *o = (int)NEARBYINT(pos);
  for (int k = 0; k < MAXT; ++k) {
    const real x = (real)(*o - KRG + k) - pos;
    g[k] = nrm * EXP(R(-0.5) * x * x * inv_s2);
    dg[k] = g[k] * x * inv_s2;
  }
  for (int j = -(ss - 1); j <= 2 * KRG; ++j) {
    real a = 0, d = 0;
    for (int q = 0; q < ss; ++q)
      if (j + q >= 0 && j + q <= 2 * KRG) {
        a += g[j + q];
        d += dg[j + q];
      }
    phi[j + ss - 1] = a;
    dphi[j + ss - 1] = d;
  }
}

typedef struct {
  real *Rv, *Rx;          /* [N][n] row-pass outputs (value taps, x-derivative taps) */
  real *fv, *fx, *fy;     /* [M][n][n] unit-flux model and its position derivatives */
} Work;

static int work_alloc(Work *w, int M, int n, int N) {
  w->Rv = (real *)malloc(sizeof(real) * (size_t)N * n);
  w->Rx = (real *)malloc(sizeof(real) * (size_t)N * n);
  w->fv = (real *)malloc(sizeof(real) * (size_t)M * n * n);
  w->fx = (real *)malloc(sizeof(real) * (size_t)M * n * n);
  w->fy = (real *)malloc(sizeof(real) * (size_t)M * n * n);
  return (w->Rv && w->Rx && w->fv && w->fx && w->fy) ? 0 : -1;
}
static void work_free(Work *w) { free(w->Rv); free(w->Rx); free(w->fv); free(w->fx); free(w->fy); }

/* One epoch: model, chi2 / 2 and the gradient with respect to this epoch's fluxes, shifts, sky level and its share of the
 * gradient with respect to the sources' positions (gX[i], gY[i]: d/d(position in data pixels)). */
static double epoch_eval(int M, int n, int ss, const real *data, const real *wgt, const real *psf, const real *a, const real *cx,
                         const real *cy, real dx, real dy, real mean, Work *w, real *ga, real *gX, real *gY, real *gmean,
                         real *model_out) {
  const int N = n * ss, cr = (N - 1) / 2, NP = MAXT + ss - 1;
  const real c0 = (real)(N - 1) / R(2.0);
  for (int i = 0; i < M; ++i) {
    int ox, oy;
    real px[MAXPHI], dpx[MAXPHI], py[MAXPHI], dpy[MAXPHI];
    taps(c0 + (real)ss * (cx[i] + dx), ss, &ox, px, dpx);
    taps(c0 + (real)ss * (cy[i] + dy), ss, &oy, py, dpy);
    /* row pass over every PSF row, fused with the column down-sampling */
    for (int r = 0; r < N; ++r) {
      const real *row = psf + (size_t)r * N;
      for (int J = 0; J < n; ++J) {
        const int base = ss * J + cr - ox + KRG + (ss - 1);   /* PSF column of tap 0 */
        real av = 0, ax = 0;
        const int k0 = base - (N - 1) > 0 ? base - (N - 1) : 0, k1 = base < NP - 1 ? base : NP - 1;   /* 0 <= base - k < N */
        for (int k = k0; k <= k1; ++k) {
          av += px[k] * row[base - k];
          ax += dpx[k] * row[base - k];
        }
        w->Rv[(size_t)r * n + J] = av;
        w->Rx[(size_t)r * n + J] = ax;
      }
    }
    /* column pass fused with the row down-sampling */
    real *fv = w->fv + (size_t)i * n * n, *fx = w->fx + (size_t)i * n * n, *fy = w->fy + (size_t)i * n * n;
    for (int I = 0; I < n; ++I) {
      const int base = ss * I + cr - oy + KRG + (ss - 1);
      for (int J = 0; J < n; ++J) fv[I * n + J] = fx[I * n + J] = fy[I * n + J] = 0;
      for (int k = 0; k < NP; ++k) {
        const int m = base - k;
        if (m < 0 || m >= N) continue;
        const real tv = py[k], td = dpy[k];
        const real *rv = w->Rv + (size_t)m * n, *rx = w->Rx + (size_t)m * n;
        for (int J = 0; J < n; ++J) {
          fv[I * n + J] += tv * rv[J];
          fx[I * n + J] += tv * rx[J];
          fy[I * n + J] += td * rv[J];
        }
      }
    }
  }
  double chi = 0, gm = 0;
  double sa[16], sx[16], sy[16];
  for (int i = 0; i < M; ++i) sa[i] = sx[i] = sy[i] = 0;
  for (int p = 0; p < n * n; ++p) {
    real f = mean;
    for (int i = 0; i < M; ++i) f += a[i] * w->fv[(size_t)i * n * n + p];
    if (model_out) model_out[p] = f;
    const real r = f - data[p], rw = wgt[p] * r;
    chi += (double)(rw * r);
    gm += (double)rw;
    for (int i = 0; i < M; ++i) {
      sa[i] += (double)(rw * w->fv[(size_t)i * n * n + p]);
      sx[i] += (double)(rw * w->fx[(size_t)i * n * n + p]);
      sy[i] += (double)(rw * w->fy[(size_t)i * n * n + p]);
    }
  }
  for (int i = 0; i < M; ++i) {
    ga[i] = (real)sa[i];
    gX[i] = (real)(sx[i] * (double)a[i] * ss);
    gY[i] = (real)(sy[i] * (double)a[i] * ss);
  }
  *gmean = (real)gm;
  return 0.5 * chi;
}

/* Loss and gradient at the given parameters.  a [E][M]; cx, cy [M]; dx, dy, mean [E]; psf [E][N][N]; wgt = 1 / sigma^2.
 * Outputs (any may be NULL except loss): ga [E][M], gcx, gcy [M], gdx, gdy, gmean [E], model [E][n][n]. */
int jps_cpu_eval(int E, int M, int n, int ss, const real *data, const real *wgt, const real *psf, const real *a, const real *cx,
                 const real *cy, const real *dx, const real *dy, const real *mean, double *loss, real *ga, real *gcx, real *gcy,
                 real *gdx, real *gdy, real *gmean, real *model, int n_threads) {
  if (M < 1 || M > 16 || ss < 1 || ss > MAXSS) return -2;
  const int N = n * ss;
  real *tga = (real *)malloc(sizeof(real) * (size_t)E * M), *tgX = (real *)malloc(sizeof(real) * (size_t)E * M);
  real *tgY = (real *)malloc(sizeof(real) * (size_t)E * M), *tgm = (real *)malloc(sizeof(real) * (size_t)E);
  double *tl = (double *)malloc(sizeof(double) * (size_t)E);
  int fail = !(tga && tgX && tgY && tgm && tl);
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  if (!fail) {
#pragma omp parallel
    {
      Work w;
      const int bad = work_alloc(&w, M, n, N);
      if (bad) {
#pragma omp atomic write
        fail = 1;
      }
#pragma omp barrier
      if (!fail) {
#pragma omp for schedule(dynamic, 1)
        for (int e = 0; e < E; ++e)
          tl[e] = epoch_eval(M, n, ss, data + (size_t)e * n * n, wgt + (size_t)e * n * n, psf + (size_t)e * N * N, a + (size_t)e * M, cx,
                             cy, dx[e], dy[e], mean[e], &w, tga + (size_t)e * M, tgX + (size_t)e * M, tgY + (size_t)e * M, tgm + e,
                             model ? model + (size_t)e * n * n : NULL);
      }
      work_free(&w);
    }
  }
  if (!fail) {
    double L = 0;
    for (int e = 0; e < E; ++e) L += tl[e];
    *loss = L;
    for (int i = 0; i < M; ++i) {
      double sx = 0, sy = 0;
      for (int e = 0; e < E; ++e) {
        sx += (double)tgX[(size_t)e * M + i];
        sy += (double)tgY[(size_t)e * M + i];
      }
      if (gcx) gcx[i] = (real)sx;
      if (gcy) gcy[i] = (real)sy;
    }
    for (int e = 0; e < E; ++e) {
      double sx = 0, sy = 0;
      for (int i = 0; i < M; ++i) {
        sx += (double)tgX[(size_t)e * M + i];
        sy += (double)tgY[(size_t)e * M + i];
        if (ga) ga[(size_t)e * M + i] = tga[(size_t)e * M + i];
      }
      if (gdx) gdx[e] = (real)sx;
      if (gdy) gdy[e] = (real)sy;
      if (gmean) gmean[e] = tgm[e];
    }
  }
  free(tga); free(tgX); free(tgY); free(tgm); free(tl);
  return fail ? -1 : 0;
}

static void adabelief_step(real *p, real *m, real *s, real g, real lr, real bc1, real bc2) {
  const real b1 = R(0.9), b2 = R(0.999), eps = R(1e-16), eps_root = R(1e-16);
  const real mn = b1 * *m + (R(1.) - b1) * g;
  const real dg = g - mn;
  const real sn = b2 * *s + (R(1.) - b2) * dg * dg + eps_root;
  *m = mn;
  *s = sn;
  *p -= lr * (mn * bc1) / (SQRT(sn * bc2) + eps);
}

/* n_iter AdaBelief iterations on a, c_x, c_y, dx, dy (and mean when free_mean) - parameters and both moments in / out, the
 * moments in the layout [a (E M) | cx (M) | cy (M) | dx (E) | dy (E) | mean (E)].  loss_hist[n_iter + 1]: loss before every
 * update, then the loss of the final parameters.  Epochs are spread over the threads; the shared sums are taken in epoch
 * order by one thread (same bits for any thread count). */
int jps_cpu_run(int E, int M, int n, int ss, const real *data, const real *wgt, const real *psf, real *a, real *cx, real *cy,
                real *dx, real *dy, real *mean, real *mom_m, real *mom_s, int free_mean, real lr0, int schedule, int t0, int n_iter,
                double *loss_hist, int n_threads) {
  if (M < 1 || M > 16 || ss < 1 || ss > MAXSS) return -2;
  const int N = n * ss;
  const size_t EM = (size_t)E * M;
  real *tga = (real *)malloc(sizeof(real) * EM), *tgX = (real *)malloc(sizeof(real) * EM), *tgY = (real *)malloc(sizeof(real) * EM);
  real *tgm = (real *)malloc(sizeof(real) * (size_t)E);
  double *tl = (double *)malloc(sizeof(double) * (size_t)E);
  int fail = !(tga && tgX && tgY && tgm && tl);
  real *m_a = mom_m, *m_cx = m_a + EM, *m_cy = m_cx + M, *m_dx = m_cy + M, *m_dy = m_dx + E, *m_mean = m_dy + E;
  real *s_a = mom_s, *s_cx = s_a + EM, *s_cy = s_cx + M, *s_dx = s_cy + M, *s_dy = s_dx + E, *s_mean = s_dy + E;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  if (!fail) {
#pragma omp parallel
    {
      Work w;
      const int bad = work_alloc(&w, M, n, N);
      if (bad) {
#pragma omp atomic write
        fail = 1;
      }
#pragma omp barrier
      if (!fail) {
        for (int it = 0; it <= n_iter; ++it) {
#pragma omp for schedule(static)
          for (int e = 0; e < E; ++e)
            tl[e] = epoch_eval(M, n, ss, data + (size_t)e * n * n, wgt + (size_t)e * n * n, psf + (size_t)e * N * N, a + (size_t)e * M,
                               cx, cy, dx[e], dy[e], mean[e], &w, tga + (size_t)e * M, tgX + (size_t)e * M, tgY + (size_t)e * M,
                               tgm + e, NULL);
          /* (implicit barrier) the update: per-epoch parameters in parallel, the shared sums by one thread in epoch order */
          const int t = t0 + it;
          const double lr = schedule ? (double)lr0 * pow(0.99, (double)t / 10.0) : (double)lr0;
          const real bc1 = (real)(1.0 / (1.0 - pow(0.9, t + 1.0))), bc2 = (real)(1.0 / (1.0 - pow(0.999, t + 1.0)));
#pragma omp single
          {
            double L = 0;
            for (int e = 0; e < E; ++e) L += tl[e];
            loss_hist[it] = L;
            if (it < n_iter)
              for (int i = 0; i < M; ++i) {
                double sx = 0, sy = 0;
                for (int e = 0; e < E; ++e) {
                  sx += (double)tgX[(size_t)e * M + i];
                  sy += (double)tgY[(size_t)e * M + i];
                }
                adabelief_step(&cx[i], &m_cx[i], &s_cx[i], (real)sx, (real)lr, bc1, bc2);
                adabelief_step(&cy[i], &m_cy[i], &s_cy[i], (real)sy, (real)lr, bc1, bc2);
              }
          }
          /* (implicit barrier of the single) */
          if (it < n_iter) {
#pragma omp for schedule(static)
            for (int e = 0; e < E; ++e) {
              double sx = 0, sy = 0;
              for (int i = 0; i < M; ++i) {
                sx += (double)tgX[(size_t)e * M + i];
                sy += (double)tgY[(size_t)e * M + i];
                adabelief_step(&a[(size_t)e * M + i], &m_a[(size_t)e * M + i], &s_a[(size_t)e * M + i], tga[(size_t)e * M + i], (real)lr, bc1,
                               bc2);
              }
              adabelief_step(&dx[e], &m_dx[e], &s_dx[e], (real)sx, (real)lr, bc1, bc2);
              adabelief_step(&dy[e], &m_dy[e], &s_dy[e], (real)sy, (real)lr, bc1, bc2);
              if (free_mean) adabelief_step(&mean[e], &m_mean[e], &s_mean[e], tgm[e], (real)lr, bc1, bc2);
            }
          }
        }
      }
      work_free(&w);
    }
  }
  free(tga); free(tgX); free(tgY); free(tgm); free(tl);
  return fail ? -1 : 0;
}
