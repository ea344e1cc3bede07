/* CPU restatement (plain C, fp32, OpenMP over frames or (frame, star) units) of the pixel-grid stage of the PSF fit.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py): this file is the "port" CPU baseline of
 * bench.py (cpu_baseline.kind = "port") and a second, independent checker of the HIP path in tests/.  The product
 * path never links, loads or calls it.  PARITY UNPINNED against STARRED itself: it restates
 * the same frozen SPEC as oracle/model.py, and tests/test_psf_cpu_port_cpu.py pins it to that float64 oracle.
 *
 * What it restates (reference call site: lightcurver/processes/psf_modelling.py:164-171, build_psf(...,
 * n_iter_adabelief=...) -- stage B of STARRED's build_psf):
 *     f_i = a_i * D_ss[ G(x0_i, y0_i) (*) (Moffat + B) ] + sky_i                      (oracle/model.py psf_model)
 *     L   = 1/2 sum w (d - f)^2 + lam_hf sum W_0 |w_0(B)| + lam_sc sum_{1<=j<J} W_j |w_j(B)|   (psf_loss)
 *     AdaBelief (optax: b1 .9, b2 .999, eps 1e-16, eps_root 1e-16, lr_t = lr0 * 0.99^(t/10)) on B, a, x0, y0
 *                                                                                      (oracle/optim.py adabelief)
 * in the direct separable form: G is the FWHM-2 Gaussian, so the convolution is a 1-D row pass fused with the
 * column down-sampling, then a 1-D column pass fused with the row down-sampling; the gradient runs the two
 * transposed passes; the starlet is the edge-replicating a-trous B3 transform with its exact adjoint.
 * The Gaussian is truncated at +-KRG samples around its rounded centre (exp(-29) relative: below fp64 resolution of
 * the sums it enters).
 *
 * Two builds of this file (oracle/Makefile): libpsfcpu.so with real = float (the CPU baseline: the arithmetic the HIP
 * path uses) and libpsfcpu_f64.so with real = double (-DPSF_CPU_DOUBLE): a second float64 implementation, independent
 * of oracle/model.py in every choice of algorithm (direct separable sums and hand-derived adjoints here, FFT
 * convolution and autograd there), used to show that the two agree to ~1e-11 over thousands of iterations while
 * fp32 trajectories of the same fit drift apart at the 1e-4 level (tests/test_psf_cpu_port_cpu.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef PSF_CPU_DOUBLE
typedef double real;
#define R(x) x
#define EXP exp
#define FABS fabs
#define SQRT sqrt
#define NEARBYINT nearbyint
#else
typedef float real;
#define R(x) x##f
#define EXP expf
#define FABS fabsf
#define SQRT sqrtf
#define NEARBYINT nearbyintf
#endif

#define KRG 6
#define SIGMA_G R(0.84932180028801907)
#define MAXT (2 * KRG + 1)

static int ilog2i(int n) { int j = 0; while (n > 1) { n >>= 1; ++j; } return j; }

/* taps g(t - delta), dg/ddelta for t = o - KRG .. o + KRG, o = round(delta) */
static void taps(real delta, int *o, real *g, real *dg) {
  const real inv_s2 = R(1.0) / (SIGMA_G * SIGMA_G), nrm = R(0.3989422804014327) / SIGMA_G;
  *o = (int)NEARBYINT(delta);
  for (int k = 0; k < MAXT; ++k) {
    const real x = (real)(*o - KRG + k) - delta;
    g[k] = nrm * EXP(-R(0.5) * x * x * inv_s2);
    dg[k] = g[k] * x * inv_s2;
  }
}

/* one edge-replicating 5-tap a-trous pass (dilation d) along rows (axis = 1) or columns (axis = 0) */
static void atrous(const real *in, real *out, int N, int d, int axis) {
  static const real b3[5] = {R(0.0625), R(0.25), R(0.375), R(0.25), R(0.0625)};
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      real acc = R(0.);
      for (int t = -2; t <= 2; ++t) {
        int uu = u, vv = v;
        if (axis) { vv = v + t * d; vv = vv < 0 ? 0 : (vv > N - 1 ? N - 1 : vv); }
        else { uu = u + t * d; uu = uu < 0 ? 0 : (uu > N - 1 ? N - 1 : uu); }
        acc += b3[t + 2] * in[uu * N + vv];
      }
      out[u * N + v] = acc;
    }
}
/* exact adjoint of atrous(): scatter through the same clamped indices */
static void atrous_adj(const real *gout, real *gin, int N, int d, int axis) {
  static const real b3[5] = {R(0.0625), R(0.25), R(0.375), R(0.25), R(0.0625)};
  memset(gin, 0, sizeof(real) * (size_t)N * N);
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      const real g = gout[u * N + v];
      for (int t = -2; t <= 2; ++t) {
        int uu = u, vv = v;
        if (axis) { vv = v + t * d; vv = vv < 0 ? 0 : (vv > N - 1 ? N - 1 : vv); }
        else { uu = u + t * d; uu = uu < 0 ? 0 : (uu > N - 1 ? N - 1 : uu); }
        gin[uu * N + vv] += b3[t + 2] * g;
      }
    }
}

typedef struct {
  real *T, *gB, *tmp, *tmpx, *V, *res, *c, *r, *cn, *q, *z, *y, *y2;
} Work;

static int work_alloc(Work *w, int N, int n, int J) {
  const size_t NN = (size_t)N * N;
  w->T = malloc(sizeof(real) * NN);
  w->gB = malloc(sizeof(real) * NN);
  w->tmp = malloc(sizeof(real) * (size_t)N * n);
  w->tmpx = malloc(sizeof(real) * (size_t)N * n);
  w->V = malloc(sizeof(real) * (size_t)N * n);
  w->res = malloc(sizeof(real) * (size_t)n * n);
  w->c = malloc(sizeof(real) * NN);
  w->r = malloc(sizeof(real) * NN);
  w->cn = malloc(sizeof(real) * NN);
  w->q = malloc(sizeof(real) * NN * (size_t)J);
  w->z = malloc(sizeof(real) * NN);
  w->y = malloc(sizeof(real) * NN);
  w->y2 = malloc(sizeof(real) * NN);
  return w->T && w->gB && w->tmp && w->tmpx && w->V && w->res && w->c && w->r && w->cn && w->q && w->z && w->y && w->y2;
}
static void work_free(Work *w) {
  free(w->T); free(w->gB); free(w->tmp); free(w->tmpx); free(w->V); free(w->res); free(w->c); free(w->r); free(w->cn);
  free(w->q); free(w->z); free(w->y); free(w->y2);
}

/* One star of a frame: forward model, chi2 and the star's gradients gs[3] = dL/da, dL/dx0, dL/dy0; its share of
 * dchi2/dB is ADDED to gB.  T = Moffat + B of the frame (read only); tmp, tmpx, V, res of `w` are scratch.
 * model_out (nullable) [n][n].  Returns the star's chi2. */
static double star_eval(int s, int n, int ss, const real *data, const real *wgt, const real *T, const real *stars, Work *w,
                        real *gB, real *gs, real *model_out) {
  const int N = n * ss;
  const real c_off = (N % 2 == 0) ? R(0.5) : R(0.0);
  double chi2 = 0.0;
  {
    const real a = stars[s * 4 + 0], x0 = stars[s * 4 + 1], y0 = stars[s * 4 + 2], sky = stars[s * 4 + 3];
    real gx[MAXT], dgx[MAXT], gy[MAXT], dgy[MAXT];
    int ox, oy;
    taps(ss * x0 + c_off, &ox, gx, dgx);
    taps(ss * y0 + c_off, &oy, gy, dgy);
    /* row pass + column down-sampling: tmp[u][jd] = sum_dv sum_k gx[k] T[u][ss jd + dv - t_k] */
    for (int u = 0; u < N; ++u)
      for (int jd = 0; jd < n; ++jd) {
        real acc = R(0.), accx = R(0.);
        for (int dv = 0; dv < ss; ++dv) {
          const int base = ss * jd + dv - ox + KRG;  /* v = base - k must lie in [0, N) */
          const int k0 = base - (N - 1) > 0 ? base - (N - 1) : 0, k1 = base < MAXT - 1 ? base : MAXT - 1;
          const real *Tr = T + u * N + base;
          for (int k = k0; k <= k1; ++k) { acc += gx[k] * Tr[-k]; accx += dgx[k] * Tr[-k]; }
        }
        w->tmp[u * n + jd] = acc;
        w->tmpx[u * n + jd] = accx;
      }
    /* column pass + row down-sampling, residuals, chi2, star gradients */
    double ga = 0.0, ggx = 0.0, ggy = 0.0;
    const real *d = data, *wg = wgt;  /* (the caller passes this star's stamp) */
    for (int id = 0; id < n; ++id)
      for (int jd = 0; jd < n; ++jd) {
        real fv = R(0.), fx = R(0.), fy = R(0.);
        for (int du = 0; du < ss; ++du) {
          const int base = ss * id + du - oy + KRG;
          const int k0 = base - (N - 1) > 0 ? base - (N - 1) : 0, k1 = base < MAXT - 1 ? base : MAXT - 1;
          for (int k = k0; k <= k1; ++k) {
            const int u = base - k;
            fv += gy[k] * w->tmp[u * n + jd];
            fy += dgy[k] * w->tmp[u * n + jd];
            fx += gy[k] * w->tmpx[u * n + jd];
          }
        }
        const real model = a * fv + sky;
        const real r = model - d[id * n + jd], rw = wg[id * n + jd] * r;
        chi2 += (double)rw * r;
        ga += (double)rw * fv;
        ggx += (double)rw * fx;
        ggy += (double)rw * fy;
        w->res[id * n + jd] = rw;
        if (model_out) model_out[id * n + jd] = model;
      }
    gs[0] = (real)ga;
    gs[1] = (real)(ggx * a * ss);
    gs[2] = (real)(ggy * a * ss);
    /* transposed column pass: V[u][jd] = sum_id gy(ss id + du - u) rw[id][jd] */
    memset(w->V, 0, sizeof(real) * (size_t)N * n);
    for (int id = 0; id < n; ++id)
      for (int du = 0; du < ss; ++du)
        for (int k = 0; k < MAXT; ++k) {
          const int u = ss * id + du - (oy - KRG + k);
          if (u < 0 || u >= N) continue;
          const real g = gy[k];
          for (int jd = 0; jd < n; ++jd) w->V[u * n + jd] += g * w->res[id * n + jd];
        }
    /* transposed row pass into dchi2/dB */
    for (int u = 0; u < N; ++u)
      for (int jd = 0; jd < n; ++jd) {
        const real vv = a * w->V[u * n + jd];
        for (int dv = 0; dv < ss; ++dv) {
          const int base = ss * jd + dv - ox + KRG;
          const int k0 = base - (N - 1) > 0 ? base - (N - 1) : 0, k1 = base < MAXT - 1 ? base : MAXT - 1;
          real *gr = gB + u * N + base;
          for (int k = k0; k <= k1; ++k) gr[-k] += gx[k] * vv;
        }
      }
  }
  return chi2;
}

/* starlet l1 of B: value (returned), sub-gradient w->z through the exact adjoint */
static double starlet_eval(int N, const real *W, const real *B, real lam_sc, real lam_hf, Work *w) {
  const int J = ilog2i(N);
  const size_t NN = (size_t)N * N;
  double l1 = 0.0;
  memset(w->z, 0, sizeof(real) * NN);
  if (lam_sc != R(0.) || lam_hf != R(0.)) {
    memcpy(w->c, B, sizeof(real) * NN);
    for (int j = 0; j < J; ++j) {
      const int dd = 1 << j;
      const real lam = (j == 0) ? lam_hf : lam_sc;
      atrous(w->c, w->r, N, dd, 1);
      atrous(w->r, w->cn, N, dd, 0);
      real *q = w->q + (size_t)j * NN;
      const real *Wj = W + (size_t)j * NN;
      for (size_t i = 0; i < NN; ++i) {
        const real wv = w->c[i] - w->cn[i], lw = lam * Wj[i];
        l1 += (double)lw * FABS(wv);
        q[i] = (wv > R(0.)) ? lw : ((wv < R(0.)) ? -lw : R(0.));
      }
      memcpy(w->c, w->cn, sizeof(real) * NN);
    }
    /* z_J = 0; z_j = q_j + Row_j^T Col_j^T (z_{j+1} - q_j) */
    for (int j = J - 1; j >= 0; --j) {
      const int dd = 1 << j;
      const real *q = w->q + (size_t)j * NN;
      for (size_t i = 0; i < NN; ++i) w->y[i] = w->z[i] - q[i];
      atrous_adj(w->y, w->y2, N, dd, 0);
      atrous_adj(w->y2, w->y, N, dd, 1);
      for (size_t i = 0; i < NN; ++i) w->z[i] = q[i] + w->y[i];
    }
  }
  return l1;
}

/* loss and gradients of one frame at the current parameters.  gs[S][3] = dL/da, dL/dx0, dL/dy0; w->gB = chi2 part,
 * w->z = l1 sub-gradient.  model_out (nullable) [S][n][n]. */
static real frame_eval(int S, int n, int ss, const real *data, const real *wgt, const real *Tm, const real *W,
                        const real *B, const real *stars, real lam_sc, real lam_hf, Work *w, real *gs,
                        real *model_out, real *chi2_out) {
  const int N = n * ss;
  const size_t NN = (size_t)N * N, nn = (size_t)n * n;
  for (size_t i = 0; i < NN; ++i) { w->T[i] = Tm[i] + B[i]; w->gB[i] = R(0.); }
  double chi2 = 0.0;
  for (int s = 0; s < S; ++s) {  /* every star's share on its own, then added: the order psf_cpu_run uses (stars in parallel) */
    memset(w->y, 0, sizeof(real) * NN);
    chi2 += star_eval(s, n, ss, data + (size_t)s * nn, wgt + (size_t)s * nn, w->T, stars, w, w->y, gs + s * 3,
                      model_out ? model_out + (size_t)s * nn : NULL);
    for (size_t i = 0; i < NN; ++i) w->gB[i] += w->y[i];
  }
  const double l1 = starlet_eval(N, W, B, lam_sc, lam_hf, w);
  if (chi2_out) *chi2_out = (real)chi2;
  return (real)(0.5 * chi2 + l1);
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

/* n_iter AdaBelief iterations on B, a, x0, y0 of every frame (state in / out).  loss_hist[F][n_iter + 1]:
 * loss before each update, then the loss of the final parameters.  Returns 0, or -1 on allocation failure.
 * With no more threads than frames every thread takes whole frames (no barrier).  With more threads than frames the work is
 * split further: (frame, star) units for the convolution part, frames for the starlet term and the update - an iteration is
 * F * (S + 1) independent units, three barriers per iteration.  Both forms add the stars' shares in the same order: same bits. */
int psf_cpu_run(int F, int S, int n, int ss, const real *data, const real *wgt, const real *Tm, const real *W,
                real *B, real *mB, real *sB, real *stars, real *stars_m, real *stars_s, real lam_sc,
                real lam_hf, real lr0, int schedule, int t0, int n_iter, real *loss_hist, int n_threads) {
  const int N = n * ss, J = ilog2i(N);
  const size_t NN = (size_t)N * N, nn = (size_t)n * n;
  int fail = 0;
  int threads = 1;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
  threads = omp_get_max_threads();
#endif
  if (threads <= F) {
    /* no more threads than frames: every thread takes whole frames through all their iterations, no barrier at all */
#pragma omp parallel
    {
      Work w;
      real *gs = malloc(sizeof(real) * (size_t)S * 3);
      const int ok = work_alloc(&w, N, n, J) && gs;
      if (!ok) {
#pragma omp atomic write
        fail = 1;
      } else {
#pragma omp for schedule(dynamic, 1)
        for (int f = 0; f < F; ++f) {
          const real *df = data + (size_t)f * S * nn, *wf = wgt + (size_t)f * S * nn, *Tf = Tm + (size_t)f * NN;
          const real *Wf = W + (size_t)f * J * NN;
          real *Bf = B + (size_t)f * NN, *mf = mB + (size_t)f * NN, *sf = sB + (size_t)f * NN;
          real *st = stars + (size_t)f * S * 4, *stm = stars_m + (size_t)f * S * 4, *sts = stars_s + (size_t)f * S * 4;
          for (int it = 0; it <= n_iter; ++it) {
            const real loss = frame_eval(S, n, ss, df, wf, Tf, Wf, Bf, st, lam_sc, lam_hf, &w, gs, NULL, NULL);
            loss_hist[(size_t)f * (n_iter + 1) + it] = loss;
            if (it == n_iter) break;
            const int t = t0 + it;
            const double lr = schedule ? (double)lr0 * pow(0.99, (double)t / 10.0) : (double)lr0;
            const real bc1 = (real)(1.0 / (1.0 - pow(0.9, t + 1))), bc2 = (real)(1.0 / (1.0 - pow(0.999, t + 1)));
            for (size_t i = 0; i < NN; ++i) adabelief_step(&Bf[i], &mf[i], &sf[i], w.gB[i] + w.z[i], (real)lr, bc1, bc2);
            for (int s = 0; s < S; ++s)
              for (int q = 0; q < 3; ++q)
                adabelief_step(&st[s * 4 + q], &stm[s * 4 + q], &sts[s * 4 + q], gs[s * 3 + q], (real)lr, bc1, bc2);
          }
        }
      }
      if (ok) work_free(&w);
      free(gs);
    }
    return fail ? -1 : 0;
  }
  /* more threads than frames: (frame, star) work units */
  real *T_all = malloc(sizeof(real) * F * NN), *gBs = malloc(sizeof(real) * (size_t)F * S * NN), *z_all = malloc(sizeof(real) * F * NN);
  real *gs_all = malloc(sizeof(real) * (size_t)F * S * 3);
  double *chi_all = malloc(sizeof(double) * (size_t)F * S), *l1_all = malloc(sizeof(double) * F);
  if (!T_all || !gBs || !z_all || !gs_all || !chi_all || !l1_all) fail = 1;
#pragma omp parallel
  {
    Work w;
    int ok = work_alloc(&w, N, n, J);
    if (!ok) {
#pragma omp atomic write
      fail = 1;
    }
#pragma omp barrier
    if (!fail) {
      for (int it = 0; it <= n_iter; ++it) {
#pragma omp for schedule(static)
        for (int f = 0; f < F; ++f)
          for (size_t i = 0; i < NN; ++i) T_all[(size_t)f * NN + i] = Tm[(size_t)f * NN + i] + B[(size_t)f * NN + i];
#pragma omp for schedule(dynamic, 1)
        for (int unit = 0; unit < F * (S + 1); ++unit) {
          const int f = unit / (S + 1), u = unit % (S + 1);
          if (u < S) {
            real *g = gBs + ((size_t)f * S + u) * NN;
            memset(g, 0, sizeof(real) * NN);
            chi_all[(size_t)f * S + u] = star_eval(u, n, ss, data + ((size_t)f * S + u) * nn, wgt + ((size_t)f * S + u) * nn,
                                                  T_all + (size_t)f * NN, stars + (size_t)f * S * 4, &w, g,
                                                  gs_all + ((size_t)f * S + u) * 3, NULL);
          } else {
            l1_all[f] = starlet_eval(N, W + (size_t)f * J * NN, B + (size_t)f * NN, lam_sc, lam_hf, &w);
            memcpy(z_all + (size_t)f * NN, w.z, sizeof(real) * NN);
          }
        }
        const int t = t0 + it;
        const double lr = schedule ? (double)lr0 * pow(0.99, (double)t / 10.0) : (double)lr0;
        const real bc1 = (real)(1.0 / (1.0 - pow(0.9, t + 1))), bc2 = (real)(1.0 / (1.0 - pow(0.999, t + 1)));
#pragma omp for schedule(static)
        for (int f = 0; f < F; ++f) {
          double chi2 = 0.0;
          for (int s = 0; s < S; ++s) chi2 += chi_all[(size_t)f * S + s];
          loss_hist[(size_t)f * (n_iter + 1) + it] = (real)(0.5 * chi2 + l1_all[f]);
          if (it == n_iter) continue;
          real *Bf = B + (size_t)f * NN, *mf = mB + (size_t)f * NN, *sf = sB + (size_t)f * NN;
          real *st = stars + (size_t)f * S * 4, *stm = stars_m + (size_t)f * S * 4, *sts = stars_s + (size_t)f * S * 4;
          const real *zf = z_all + (size_t)f * NN;
          for (size_t i = 0; i < NN; ++i) {
            real g = R(0.);
            for (int s = 0; s < S; ++s) g += gBs[((size_t)f * S + s) * NN + i];  /* the stars' shares in order */
            adabelief_step(&Bf[i], &mf[i], &sf[i], g + zf[i], (real)lr, bc1, bc2);
          }
          for (int s = 0; s < S; ++s)
            for (int q = 0; q < 3; ++q)
              adabelief_step(&st[s * 4 + q], &stm[s * 4 + q], &sts[s * 4 + q], gs_all[((size_t)f * S + s) * 3 + q], (real)lr, bc1, bc2);
        }
      }
    }
    if (ok) work_free(&w);
  }
  free(T_all); free(gBs); free(z_all); free(gs_all); free(chi_all); free(l1_all);
  return fail ? -1 : 0;
}

/* one evaluation per frame: loss [F], chi2 [F], grad_grid [F][N*N] (regularisation included), grad_stars [F][S][3],
 * model [F][S][n][n] (nullable outputs) */
int psf_cpu_eval(int F, int S, int n, int ss, const real *data, const real *wgt, const real *Tm, const real *W,
                 const real *B, const real *stars, real lam_sc, real lam_hf, real *loss, real *chi2,
                 real *grad_grid, real *grad_stars, real *model) {
  const int N = n * ss, J = ilog2i(N);
  const size_t NN = (size_t)N * N, nn = (size_t)n * n;
  Work w;
  real *gs = malloc(sizeof(real) * (size_t)S * 3);
  if (!work_alloc(&w, N, n, J) || !gs) return -1;
  for (int f = 0; f < F; ++f) {
    real c2 = R(0.);
    const real L = frame_eval(S, n, ss, data + (size_t)f * S * nn, wgt + (size_t)f * S * nn, Tm + (size_t)f * NN,
                               W + (size_t)f * J * NN, B + (size_t)f * NN, stars + (size_t)f * S * 4, lam_sc, lam_hf, &w,
                               gs, model ? model + (size_t)f * S * nn : NULL, &c2);
    if (loss) loss[f] = L;
    if (chi2) chi2[f] = c2;
    if (grad_grid)
      for (size_t i = 0; i < NN; ++i) grad_grid[(size_t)f * NN + i] = w.gB[i] + w.z[i];
    if (grad_stars) memcpy(grad_stars + (size_t)f * S * 3, gs, sizeof(real) * (size_t)S * 3);
  }
  work_free(&w);
  free(gs);
  return 0;
}
