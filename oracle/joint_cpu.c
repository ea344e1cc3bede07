/* CPU restatement (plain C, OpenMP over epochs) of the joint multi-epoch forward-model fit WITH the pixelated background:
 * the hot loop of the reference's ROI modelling.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py): the "port" CPU baseline bench.py times beside the joint-fit
 * entries (cpu_baseline.kind = "port") and a third, independent checker in tests/.  The product path never links, loads or
 * calls it.  PARITY UNPINNED against STARRED itself: it restates the same frozen SPEC as
 * oracle/model.py, and tests/test_joint_cpu_port_cpu.py pins its float64 build to that float64 oracle (loss and every
 * gradient block to 1e-9, AdaBelief trajectories to 1e-9).
 *
 * What it restates (reference call site: lightcurver/processes/roi_modelling.py:213-334 - setup_model with M point sources
 * and the background h, Loss with the starlet / positivity / point-source / flux-scatter regularisers (:308-321), Optimizer
 * 'adabelief', 2000 iterations (:326-334)):
 *     f_e = D_ss[ s_e (*) ( T_e[h] + sum_i a_ei G(R_e c_i + d_e) ) ] + mean_e              (oracle/model.py deconv_model)
 *     L   = 1/2 sum (d - f)^2 / sigma^2 + lam_hf sum W_0 |w_0(h)| + lam sum_{1 <= j < J} W_j |w_j(h)|
 *           + lam_pos sum max(-h, 0) + lam_pos_ps sum max(-a, 0) + lam_pts sum W_0 |w_0(Pbar)| + lam_fu sum_i std_e(a_ei)
 *                                                                                           (deconv_loss; no prior term)
 *     AdaBelief (optax: b1 .9, b2 .999, eps 1e-16, eps_root 1e-16, lr_t = lr0 * 0.99^(t/10) when scheduled)
 * by its own route: radix-2 FFTs written here (length L = the power of two >= 2 N: fully linear, two real rows per complex
 * transform, half spectra), hand-derived adjoints - the transposed convolution as a correlation, T_e^T as a scatter of the
 * four bilinear weights, the adjoint of the a-trous cascade by the recursion z_j = q_j + S_j^T (z_{j+1} - q_j) - where
 * oracle/model.py has torch.fft and autograd, and the HIP kernels length-3N/2 register FFTs with binned rows.
 *
 * Two builds (oracle/Makefile): libjointcpu.so with real = float (the arithmetic of the HIP path: the CPU baseline) and
 * libjointcpu_f64.so with real = double (the checker).  Sums over the epochs are taken in epoch order by one thread: same
 * bits for any thread count.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef JC_CPU_DOUBLE
typedef double real;
#define R(x) x
#define EXP exp
#define SQRT sqrt
#define FLOOR floor
#define COS cos
#define SIN sin
#define FABS fabs
#else
typedef float real;
#define R(x) x##f
#define EXP expf
#define SQRT sqrtf
#define FLOOR floorf
#define COS cosf
#define SIN sinf
#define FABS fabsf
#endif

#define SIGMA_G R(0.84932180028801907)
#define MAXM 16

typedef struct { real re, im; } cplx;

struct JcWork_;
typedef struct {
  int E, M, n, ss, N, L, LH, J;
  const real *data, *wgt;   /* [E][n][n] (borrowed) */
  cplx *K;                  /* [E][LH][L] PSF spectra (x frequency major), 1 / L^2 folded in */
  cplx *tw;                 /* [L / 2] exp(-2 pi i k / L) */
  int *rev;                 /* [L] bit reversal */
  struct JcWork_ *works;    /* per-thread work space, kept across calls (fresh allocations page-fault on every iteration) */
  int nworks;
} JcCtx;

typedef struct {
  real lam_scales, lam_hf, lam_pos, lam_pos_ps, lam_pts, lam_fu;
} JcLoss;

/* ---- FFT ------------------------------------------------------------------------------------------------------------ */
static void fft1d(cplx *x, int L, const cplx *tw, const int *rev, int inverse) {
  for (int i = 0; i < L; ++i) {
    const int j = rev[i];
    if (i < j) {
      const cplx t = x[i];
      x[i] = x[j];
      x[j] = t;
    }
  }
  for (int half = 1; half < L; half <<= 1) {
    const int step = L / (2 * half);
    for (int i = 0; i < L; i += 2 * half)
      for (int k = 0; k < half; ++k) {
        const cplx w = tw[k * step];
        const real wi = inverse ? -w.im : w.im;
        cplx *a = x + i + k, *b = a + half;
        const real tr = b->re * w.re - b->im * wi, ti = b->re * wi + b->im * w.re;
        b->re = a->re - tr;
        b->im = a->im - ti;
        a->re += tr;
        a->im += ti;
      }
  }
}

typedef struct JcWork_ {
  cplx *spec;    /* [LH][L] */
  cplx *row;     /* [L] */
  cplx *half;    /* [N + pad][LH] row spectra before the column pass */
  real *scene, *conv, *gs, *up;   /* [N][N] */
  real *hslab;   /* unused when the caller gives a slab */
} JcWork;

static int work_alloc(JcWork *w, const JcCtx *c) {
  const size_t NN = (size_t)c->N * c->N;
  w->spec = (cplx *)malloc(sizeof(cplx) * (size_t)c->LH * c->L);
  w->row = (cplx *)malloc(sizeof(cplx) * (size_t)c->L);
  w->half = (cplx *)malloc(sizeof(cplx) * (size_t)c->L * c->LH);
  w->scene = (real *)malloc(sizeof(real) * NN);
  w->conv = (real *)malloc(sizeof(real) * NN);
  w->gs = (real *)malloc(sizeof(real) * NN);
  w->up = (real *)malloc(sizeof(real) * NN);
  w->hslab = NULL;
  return (w->spec && w->row && w->half && w->scene && w->conv && w->gs && w->up) ? 0 : -1;
}
static void work_free(JcWork *w) {
  free(w->spec); free(w->row); free(w->half); free(w->scene); free(w->conv); free(w->gs); free(w->up);
}

/* half spectrum of the real N x N image `in`, placed at (off, off) of the zero L x L frame  ->  spec [LH][L] */
static void fwd2d(const JcCtx *c, const real *in, int off, JcWork *w) {
  const int N = c->N, L = c->L, LH = c->LH;
  /* rows, two per complex transform: Z = FFT(x1 + i x2), X1 = (Z[k] + conj Z[L-k]) / 2, X2 = (Z[k] - conj Z[L-k]) / 2i */
  for (int r = 0; r < N; r += 2) {
    memset(w->row, 0, sizeof(cplx) * (size_t)L);
    for (int v = 0; v < N; ++v) {
      w->row[v + off].re = in[(size_t)r * N + v];
      w->row[v + off].im = (r + 1 < N) ? in[(size_t)(r + 1) * N + v] : 0;
    }
    fft1d(w->row, L, c->tw, c->rev, 0);
    for (int k = 0; k < LH; ++k) {
      const cplx zk = w->row[k], zc = w->row[(L - k) % L];
      cplx *h1 = w->half + (size_t)r * LH + k, *h2 = w->half + (size_t)(r + 1) * LH + k;
      h1->re = R(0.5) * (zk.re + zc.re);
      h1->im = R(0.5) * (zk.im - zc.im);
      if (r + 1 < N) {
        h2->re = R(0.5) * (zk.im + zc.im);
        h2->im = R(-0.5) * (zk.re - zc.re);
      }
    }
  }
  /* columns */
  for (int k = 0; k < LH; ++k) {
    cplx *col = w->spec + (size_t)k * L;
    memset(col, 0, sizeof(cplx) * (size_t)L);
    for (int r = 0; r < N; ++r) col[r + off] = w->half[(size_t)r * LH + k];
    fft1d(col, L, c->tw, c->rev, 0);
  }
}

/* spec [LH][L] (already multiplied)  ->  the N x N window at (off, off) of the real L x L image, unnormalised */
static void inv2d(const JcCtx *c, int off, real *out, JcWork *w) {
  const int N = c->N, L = c->L, LH = c->LH;
  for (int k = 0; k < LH; ++k) {
    cplx *col = w->spec + (size_t)k * L;
    fft1d(col, L, c->tw, c->rev, 1);
    for (int r = 0; r < N; ++r) w->half[(size_t)r * LH + k] = col[r + off];
  }
  for (int r = 0; r < N; r += 2) {
    const cplx *h1 = w->half + (size_t)r * LH, *h2 = (r + 1 < N) ? w->half + (size_t)(r + 1) * LH : NULL;
    for (int k = 0; k < LH; ++k) {
      const real x2r = h2 ? h2[k].re : 0, x2i = h2 ? h2[k].im : 0;
      w->row[k].re = h1[k].re - x2i;            /* X1 + i X2 */
      w->row[k].im = h1[k].im + x2r;
      if (k > 0 && k < L - k) {                 /* Hermitian extension: conj(X1[k]) + i conj(X2[k]) */
        w->row[L - k].re = h1[k].re + x2i;
        w->row[L - k].im = -h1[k].im + x2r;
      }
    }
    fft1d(w->row, L, c->tw, c->rev, 1);
    for (int v = 0; v < N; ++v) {
      out[(size_t)r * N + v] = w->row[v + off].re;
      if (r + 1 < N) out[(size_t)(r + 1) * N + v] = w->row[v + off].im;
    }
  }
}

static void times_spectrum(const JcCtx *c, const cplx *K, int conj, JcWork *w) {
  const size_t n = (size_t)c->LH * c->L;
  for (size_t i = 0; i < n; ++i) {
    const cplx a = w->spec[i], b = K[i];
    const real bi = conj ? -b.im : b.im;
    w->spec[i].re = a.re * b.re - a.im * bi;
    w->spec[i].im = a.re * bi + a.im * b.re;
  }
}

/* ---- context ---------------------------------------------------------------------------------------------------------- */
void jc_cpu_destroy(JcCtx *c) {
  if (!c) return;
  for (int t = 0; t < c->nworks; ++t) work_free(&c->works[t]);
  free(c->works);
  free(c->K); free(c->tw); free(c->rev);
  free(c);
}
/* work space of at least nthr threads (grown by one thread, outside parallel regions) */
static int ensure_works(JcCtx *c, int nthr) {
  if (nthr <= c->nworks) return 0;
  JcWork *nw = (JcWork *)realloc(c->works, sizeof(JcWork) * (size_t)nthr);
  if (!nw) return -1;
  c->works = nw;
  for (int t = c->nworks; t < nthr; ++t) {
    if (work_alloc(&c->works[t], c)) {
      work_free(&c->works[t]);
      return -1;
    }
    c->nworks = t + 1;
  }
  return 0;
}
static int max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
static int thread_num(void) {
#ifdef _OPENMP
  return omp_get_thread_num();
#else
  return 0;
#endif
}

JcCtx *jc_cpu_create(int E, int M, int n, int ss, const real *data, const real *wgt, const real *psf, int n_threads) {
  if (E < 1 || M < 0 || M > MAXM || n < 2 || ss < 1) return NULL;
  JcCtx *c = (JcCtx *)calloc(1, sizeof(JcCtx));
  if (!c) return NULL;
  c->E = E; c->M = M; c->n = n; c->ss = ss; c->N = n * ss;
  c->L = 1;
  while (c->L < 2 * c->N) c->L <<= 1;
  c->LH = c->L / 2 + 1;
  c->J = 0;
  while ((2 << c->J) <= c->N) c->J += 1;       /* floor(log2 N) */
  c->data = data; c->wgt = wgt;
  const int L = c->L;
  c->tw = (cplx *)malloc(sizeof(cplx) * (size_t)(L / 2));
  c->rev = (int *)malloc(sizeof(int) * (size_t)L);
  c->K = (cplx *)malloc(sizeof(cplx) * (size_t)E * c->LH * L);
  if (!c->tw || !c->rev || !c->K) { jc_cpu_destroy(c); return NULL; }
  for (int k = 0; k < L / 2; ++k) {
    const double ang = -2.0 * 3.14159265358979323846 * k / L;
    c->tw[k].re = (real)cos(ang);
    c->tw[k].im = (real)sin(ang);
  }
  int bits = 0;
  while ((1 << bits) < L) ++bits;
  for (int i = 0; i < L; ++i) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
    c->rev[i] = r;
  }
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  if (ensure_works(c, max_threads())) { jc_cpu_destroy(c); return NULL; }
#pragma omp parallel
  {
    JcWork *w = &c->works[thread_num()];
    const real sc = R(1.0) / ((real)L * (real)L);
#pragma omp for schedule(static)
    for (int e = 0; e < E; ++e) {
      fwd2d(c, psf + (size_t)e * c->N * c->N, 0, w);
      cplx *Ke = c->K + (size_t)e * c->LH * L;
      for (size_t i = 0; i < (size_t)c->LH * L; ++i) {
        Ke[i].re = w->spec[i].re * sc;
        Ke[i].im = w->spec[i].im * sc;
      }
    }
  }
  return c;
}

/* ---- starlet -------------------------------------------------------------------------------------------------------- */
static const real B3[5] = {R(0.0625), R(0.25), R(0.375), R(0.25), R(0.0625)};
static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* out = Col_d Row_d in (edge replicating); tmp: N x N */
static void smooth(int N, int d, const real *in, real *tmp, real *out) {
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      real acc = 0;
      for (int t = -2; t <= 2; ++t) acc += B3[t + 2] * in[(size_t)clampi(u + t * d, 0, N - 1) * N + v];
      tmp[(size_t)u * N + v] = acc;
    }
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      real acc = 0;
      for (int t = -2; t <= 2; ++t) acc += B3[t + 2] * tmp[(size_t)u * N + clampi(v + t * d, 0, N - 1)];
      out[(size_t)u * N + v] = acc;
    }
}
/* out = (Col_d Row_d)^T in = Row_d^T Col_d^T in: the transposes scatter where the forward passes gather */
static void smooth_adjoint(int N, int d, const real *in, real *tmp, real *out) {
  memset(tmp, 0, sizeof(real) * (size_t)N * N);
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      const real y = in[(size_t)u * N + v];
      for (int t = -2; t <= 2; ++t) tmp[(size_t)u * N + clampi(v + t * d, 0, N - 1)] += B3[t + 2] * y;
    }
  memset(out, 0, sizeof(real) * (size_t)N * N);
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      const real y = tmp[(size_t)u * N + v];
      for (int t = -2; t <= 2; ++t) out[(size_t)clampi(u + t * d, 0, N - 1) * N + v] += B3[t + 2] * y;
    }
}

/* value and sub-gradient of  lam_hf sum W_0 |w_0| + lam sum_{1 <= j < jmax} W_j |w_j|  of the N x N image x.
 * W: [>= jmax][N][N] or NULL (then norms[j]).  g (N x N) receives the sub-gradient (sign(0) = 0).  buf: 5 N^2 reals. */
static double l1_starlet_grad(int N, int jmax, const real *x, const real *W, const real *norms, real lam_hf, real lam_sc, real *g,
                              real *buf, real *qbuf /* [jmax][N][N] */) {
  const size_t NN = (size_t)N * N;
  real *c = buf, *cn = buf + NN, *tmp = buf + 2 * NN, *z = buf + 3 * NN, *y = buf + 4 * NN;
  double val = 0;
  memcpy(c, x, sizeof(real) * NN);
  for (int j = 0; j < jmax; ++j) {
    const real lam = (j == 0) ? lam_hf : lam_sc;
    smooth(N, 1 << j, c, tmp, cn);
    real *q = qbuf + (size_t)j * NN;
    double v = 0;
    for (size_t k = 0; k < NN; ++k) {
      const real wv = c[k] - cn[k], lw = lam * (W ? W[(size_t)j * NN + k] : norms[j]);
      q[k] = (wv > 0) ? lw : ((wv < 0) ? -lw : 0);
      v += (double)(lw * FABS(wv));
    }
    val += v;
    memcpy(c, cn, sizeof(real) * NN);
  }
  memset(z, 0, sizeof(real) * NN);                      /* z_jmax = 0 */
  for (int j = jmax - 1; j >= 0; --j) {                  /* z_j = q_j + S_j^T (z_{j+1} - q_j) */
    const real *q = qbuf + (size_t)j * NN;
    for (size_t k = 0; k < NN; ++k) y[k] = z[k] - q[k];
    smooth_adjoint(N, 1 << j, y, tmp, z);
    for (size_t k = 0; k < NN; ++k) z[k] += q[k];
  }
  memcpy(g, z, sizeof(real) * NN);
  return val;
}

/* norms of the starlet atoms (dirac at the zero-lag index of the N x N grid, edge effects included): the weights without W */
static void starlet_norms(int N, int J, real *norms, real *buf) {
  const size_t NN = (size_t)N * N;
  real *c = buf, *cn = buf + NN, *tmp = buf + 2 * NN;
  memset(c, 0, sizeof(real) * NN);
  const int cr = (N - 1) / 2;
  c[(size_t)cr * N + cr] = 1;
  for (int j = 0; j < J; ++j) {
    smooth(N, 1 << j, c, tmp, cn);
    double s = 0;
    for (size_t k = 0; k < NN; ++k) s += (double)((c[k] - cn[k]) * (c[k] - cn[k]));
    norms[j] = (real)sqrt(s);
    memcpy(c, cn, sizeof(real) * NN);
  }
}

/* ---- one epoch ------------------------------------------------------------------------------------------------------- */
/* chi2 / 2 of the epoch; gradients with respect to its fluxes, its share of d/dc (already rotated back), its shifts and sky
 * level; T_e^T of the scene gradient into hslab (N x N, overwritten). */
static double epoch_eval(const JcCtx *c, int e, const real *a, const real *cx, const real *cy, real dx, real dy, real alpha,
                         real mean, const real *h, JcWork *w, real *ga, real *gcx, real *gcy, real *gdx, real *gdy, real *gmean,
                         real *hslab, real *model_out) {
  const int N = c->N, n = c->n, ss = c->ss, M = c->M, cr = (N - 1) / 2;
  const real c0 = (real)(N - 1) / R(2.0), inv_s2 = R(1.0) / (SIGMA_G * SIGMA_G), nrm2 = R(0.15915494309189535) * inv_s2;
  const real al = alpha * R(0.017453292519943295), ca = COS(al), sa = SIN(al);
  real gx[MAXM][512], gy[MAXM][512], X[MAXM], Y[MAXM];
  /* scene = T_e[h] + point sources */
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      const real px = ((real)v - c0) - (real)ss * dx, py = ((real)u - c0) - (real)ss * dy;
      const real Xs = c0 + ca * px + sa * py, Ys = c0 - sa * px + ca * py;
      const real x0f = FLOOR(Xs), y0f = FLOOR(Ys), fx = Xs - x0f, fy = Ys - y0f;
      const int x0 = (int)x0f, y0 = (int)y0f;
      const int xa = clampi(x0, 0, N - 1), xb = clampi(x0 + 1, 0, N - 1), ya = clampi(y0, 0, N - 1), yb = clampi(y0 + 1, 0, N - 1);
      const real top = (1 - fx) * h[(size_t)ya * N + xa] + fx * h[(size_t)ya * N + xb];
      const real bot = (1 - fx) * h[(size_t)yb * N + xa] + fx * h[(size_t)yb * N + xb];
      w->scene[(size_t)u * N + v] = (1 - fy) * top + fy * bot;
    }
  for (int i = 0; i < M; ++i) {
    X[i] = c0 + (real)ss * (ca * cx[i] - sa * cy[i] + dx);
    Y[i] = c0 + (real)ss * (sa * cx[i] + ca * cy[i] + dy);
    for (int p = 0; p < N; ++p) {
      const real tx = (real)p - X[i], ty = (real)p - Y[i];
      gx[i][p] = EXP(R(-0.5) * tx * tx * inv_s2);
      gy[i][p] = EXP(R(-0.5) * ty * ty * inv_s2);
    }
    const real amp = a[i] * nrm2;
    for (int u = 0; u < N; ++u) {
      const real ay = amp * gy[i][u];
      for (int v = 0; v < N; ++v) w->scene[(size_t)u * N + v] += ay * gx[i][v];
    }
  }
  /* convolution ('same' window: zero lag at cr), down-sampling, residual */
  fwd2d(c, w->scene, 0, w);
  times_spectrum(c, c->K + (size_t)e * c->LH * c->L, 0, w);
  inv2d(c, cr, w->conv, w);
  const real *de = c->data + (size_t)e * n * n, *we = c->wgt + (size_t)e * n * n;
  double chi = 0, gm = 0;
  for (int I = 0; I < n; ++I)
    for (int Jd = 0; Jd < n; ++Jd) {
      real f = mean;
      for (int p = 0; p < ss; ++p)
        for (int q = 0; q < ss; ++q) f += w->conv[(size_t)(ss * I + p) * N + ss * Jd + q];
      if (model_out) model_out[(size_t)I * n + Jd] = f;
      const real r = f - de[(size_t)I * n + Jd], rw = we[(size_t)I * n + Jd] * r;
      chi += (double)(rw * r);
      gm += (double)rw;
      for (int p = 0; p < ss; ++p)
        for (int q = 0; q < ss; ++q) w->up[(size_t)(ss * I + p) * N + ss * Jd + q] = rw;
    }
  *gmean = (real)gm;
  /* adjoint: the up-sampled weighted residual correlated with the PSF -> gradient with respect to the scene */
  fwd2d(c, w->up, cr, w);
  times_spectrum(c, c->K + (size_t)e * c->LH * c->L, 1, w);
  inv2d(c, 0, w->gs, w);
  /* point sources */
  double sdx = 0, sdy = 0;
  for (int i = 0; i < M; ++i) {
    double s0 = 0, sX = 0, sY = 0;
    for (int u = 0; u < N; ++u) {
      double r0 = 0, rX = 0;
      for (int v = 0; v < N; ++v) {
        const real g = w->gs[(size_t)u * N + v] * gx[i][v];
        r0 += (double)g;
        rX += (double)(g * ((real)v - X[i]) * inv_s2);
      }
      s0 += r0 * (double)gy[i][u];
      sX += rX * (double)gy[i][u];
      sY += r0 * (double)(gy[i][u] * ((real)u - Y[i]) * inv_s2);
    }
    ga[i] = (real)(s0 * (double)nrm2);
    const double gX = (double)a[i] * (double)nrm2 * sX, gY = (double)a[i] * (double)nrm2 * sY;
    gcx[i] = (real)((double)ss * ((double)ca * gX + (double)sa * gY));
    gcy[i] = (real)((double)ss * ((double)ca * gY - (double)sa * gX));
    sdx += (double)ss * gX;
    sdy += (double)ss * gY;
  }
  /* background: shifts through the interpolation, T_e^T by scattering the four weights */
  memset(hslab, 0, sizeof(real) * (size_t)N * N);
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      const real g = w->gs[(size_t)u * N + v];
      const real px = ((real)v - c0) - (real)ss * dx, py = ((real)u - c0) - (real)ss * dy;
      const real Xs = c0 + ca * px + sa * py, Ys = c0 - sa * px + ca * py;
      const real x0f = FLOOR(Xs), y0f = FLOOR(Ys), fx = Xs - x0f, fy = Ys - y0f;
      const int x0 = (int)x0f, y0 = (int)y0f;
      const int xa = clampi(x0, 0, N - 1), xb = clampi(x0 + 1, 0, N - 1), ya = clampi(y0, 0, N - 1), yb = clampi(y0 + 1, 0, N - 1);
      const real h00 = h[(size_t)ya * N + xa], h01 = h[(size_t)ya * N + xb], h10 = h[(size_t)yb * N + xa], h11 = h[(size_t)yb * N + xb];
      const real dHx = (1 - fy) * (h01 - h00) + fy * (h11 - h10);
      const real dHy = ((1 - fx) * h10 + fx * h11) - ((1 - fx) * h00 + fx * h01);
      sdx += (double)(g * (real)ss * (sa * dHy - ca * dHx));
      sdy += (double)(g * -(real)ss * (sa * dHx + ca * dHy));
      hslab[(size_t)ya * N + xa] += g * (1 - fx) * (1 - fy);
      hslab[(size_t)ya * N + xb] += g * fx * (1 - fy);
      hslab[(size_t)yb * N + xa] += g * (1 - fx) * fy;
      hslab[(size_t)yb * N + xb] += g * fx * fy;
    }
  *gdx = (real)sdx;
  *gdy = (real)sdy;
  return 0.5 * chi;
}

/* ---- the terms that couple the epochs: regularisers of h, point-source term, flux scatter ------------------------------------
 * adds their gradients to gh [N^2], ga [E][M], gcx, gcy [M]; returns their value.  buf: (5 + J) N^2 reals. */
static double shared_terms(const JcCtx *c, const JcLoss *lo, const real *W, const real *a, const real *cx, const real *cy, const real *h,
                           real *gh, real *ga, real *gcx, real *gcy, real *buf) {
  const int N = c->N, E = c->E, M = c->M, J = c->J, ss = c->ss;
  const size_t NN = (size_t)N * N;
  real norms[32];
  real *g = buf, *work = buf + NN, *qbuf = buf + 6 * NN;
  double val = 0;
  const int need_norms = !W && (lo->lam_scales != 0 || lo->lam_hf != 0 || lo->lam_pts != 0);
  if (need_norms) starlet_norms(N, J, norms, work);
  if (lo->lam_scales != 0 || lo->lam_hf != 0) {
    val += l1_starlet_grad(N, J, h, W, norms, lo->lam_hf, lo->lam_scales, g, work, qbuf);
    for (size_t k = 0; k < NN; ++k) gh[k] += g[k];
  }
  if (lo->lam_pos != 0) {
    double v = 0;
    for (size_t k = 0; k < NN; ++k)
      if (h[k] < 0) {
        v += (double)(-lo->lam_pos * h[k]);
        gh[k] -= lo->lam_pos;
      }
    val += v;
  }
  if (lo->lam_pos_ps != 0) {
    double v = 0;
    for (size_t k = 0; k < (size_t)E * M; ++k)
      if (a[k] < 0) {
        v += (double)(-lo->lam_pos_ps * a[k]);
        ga[k] -= lo->lam_pos_ps;
      }
    val += v;
  }
  if (lo->lam_pts != 0 && M > 0) {
    const real c0 = (real)(N - 1) / R(2.0), inv_s2 = R(1.0) / (SIGMA_G * SIGMA_G), nrm2 = R(0.15915494309189535) * inv_s2;
    real *pbar = buf + 5 * NN;   /* (free while l1_starlet_grad is not running: it uses buf + NN .. buf + 6 NN as work) */
    real abar[MAXM];
    real *gxs = (real *)malloc(sizeof(real) * (size_t)M * N), *gys = (real *)malloc(sizeof(real) * (size_t)M * N);
    real *pb = (real *)malloc(sizeof(real) * NN);
    if (gxs && gys && pb) {
      (void)pbar;
      memset(pb, 0, sizeof(real) * NN);
      for (int i = 0; i < M; ++i) {
        double s = 0;
        for (int e = 0; e < E; ++e) s += (double)a[(size_t)e * M + i];
        abar[i] = (real)(s / E);
        const real Xi = c0 + (real)ss * cx[i], Yi = c0 + (real)ss * cy[i];
        for (int p = 0; p < N; ++p) {
          const real tx = (real)p - Xi, ty = (real)p - Yi;
          gxs[(size_t)i * N + p] = EXP(R(-0.5) * tx * tx * inv_s2);
          gys[(size_t)i * N + p] = EXP(R(-0.5) * ty * ty * inv_s2);
        }
        for (int u = 0; u < N; ++u)
          for (int v = 0; v < N; ++v) pb[(size_t)u * N + v] += abar[i] * nrm2 * gys[(size_t)i * N + u] * gxs[(size_t)i * N + v];
      }
      /* scale 0 only, with the scale-0 weights */
      val += l1_starlet_grad(N, 1, pb, W, norms, lo->lam_pts, 0, g, work, qbuf);
      for (int i = 0; i < M; ++i) {
        const real Xi = c0 + (real)ss * cx[i], Yi = c0 + (real)ss * cy[i];
        double s0 = 0, sX = 0, sY = 0;
        for (int u = 0; u < N; ++u)
          for (int v = 0; v < N; ++v) {
            const double gq = (double)(g[(size_t)u * N + v] * nrm2 * gys[(size_t)i * N + u] * gxs[(size_t)i * N + v]);
            s0 += gq;
            sX += gq * (double)(((real)v - Xi) * inv_s2);
            sY += gq * (double)(((real)u - Yi) * inv_s2);
          }
        for (int e = 0; e < E; ++e) ga[(size_t)e * M + i] += (real)(s0 / E);
        gcx[i] += (real)((double)ss * (double)abar[i] * sX);
        gcy[i] += (real)((double)ss * (double)abar[i] * sY);
      }
    }
    free(gxs); free(gys); free(pb);
  }
  if (lo->lam_fu != 0 && E > 1) {
    for (int i = 0; i < M; ++i) {
      double s = 0, s2 = 0;
      for (int e = 0; e < E; ++e) s += (double)a[(size_t)e * M + i];
      const double mu = s / E;
      for (int e = 0; e < E; ++e) s2 += ((double)a[(size_t)e * M + i] - mu) * ((double)a[(size_t)e * M + i] - mu);
      const double sd = sqrt(s2 / E);
      val += (double)lo->lam_fu * sd;
      if (sd > 0)
        for (int e = 0; e < E; ++e) ga[(size_t)e * M + i] += (real)((double)lo->lam_fu * ((double)a[(size_t)e * M + i] - mu) / (E * sd));
    }
  }
  return val;
}

/* ---- loss and gradient --------------------------------------------------------------------------------------------------- */
/* a [E][M]; cx, cy [M]; dx, dy, alpha, mean [E]; h [N^2]; W [J][N^2] or NULL.  Outputs may be NULL except loss. */
static int eval_all(const JcCtx *c, const JcLoss *lo, const real *W, const real *a, const real *cx, const real *cy, const real *dx,
                    const real *dy, const real *alpha, const real *mean, const real *h, double *loss, real *ga, real *gcx, real *gcy,
                    real *gdx, real *gdy, real *gmean, real *gh, real *model, real *slabs, real *tcx, real *tcy, double *tl,
                    real *shared_buf, int in_parallel) {
  const int E = c->E, M = c->M, N = c->N, n = c->n;
  const size_t NN = (size_t)N * N;
  (void)in_parallel;
  if (ensure_works((JcCtx *)c, max_threads())) return -1;
#pragma omp parallel
  {
    JcWork *w = &c->works[thread_num()];
#pragma omp for schedule(static)
    for (int e = 0; e < E; ++e)
      tl[e] = epoch_eval(c, e, a + (size_t)e * M, cx, cy, dx[e], dy[e], alpha[e], mean[e], h, w, ga + (size_t)e * M, tcx + (size_t)e * M,
                         tcy + (size_t)e * M, gdx + e, gdy + e, gmean + e, slabs + (size_t)e * NN, model ? model + (size_t)e * n * n : NULL);
    /* the sum over the epochs of the T_e^T slabs: pixels in parallel, epochs in order */
#pragma omp for schedule(static)
    for (long k = 0; k < (long)NN; ++k) {
      real acc = 0;
      for (int e = 0; e < E; ++e) acc += slabs[(size_t)e * NN + k];
      gh[k] = acc;
    }
  }
  double L = 0;
  for (int e = 0; e < E; ++e) L += tl[e];
  for (int i = 0; i < M; ++i) {
    double sx = 0, sy = 0;
    for (int e = 0; e < E; ++e) {
      sx += (double)tcx[(size_t)e * M + i];
      sy += (double)tcy[(size_t)e * M + i];
    }
    gcx[i] = (real)sx;
    gcy[i] = (real)sy;
  }
  L += shared_terms(c, lo, W, a, cx, cy, h, gh, ga, gcx, gcy, shared_buf);
  *loss = L;
  return 0;
}

typedef struct {
  real *ga, *gcx, *gcy, *gdx, *gdy, *gmean, *gh, *slabs, *tcx, *tcy, *shared_buf;
  double *tl;
} JcScratch;
static int scratch_alloc(JcScratch *s, const JcCtx *c) {
  const size_t NN = (size_t)c->N * c->N, EM = (size_t)c->E * (c->M > 0 ? c->M : 1);
  s->ga = (real *)calloc(EM, sizeof(real)); s->gcx = (real *)calloc(MAXM, sizeof(real)); s->gcy = (real *)calloc(MAXM, sizeof(real));
  s->gdx = (real *)calloc((size_t)c->E, sizeof(real)); s->gdy = (real *)calloc((size_t)c->E, sizeof(real));
  s->gmean = (real *)calloc((size_t)c->E, sizeof(real)); s->gh = (real *)calloc(NN, sizeof(real));
  s->slabs = (real *)malloc(sizeof(real) * (size_t)c->E * NN);
  s->tcx = (real *)calloc(EM, sizeof(real)); s->tcy = (real *)calloc(EM, sizeof(real));
  s->shared_buf = (real *)malloc(sizeof(real) * (size_t)(7 + c->J) * NN);
  s->tl = (double *)calloc((size_t)c->E, sizeof(double));
  return (s->ga && s->gcx && s->gcy && s->gdx && s->gdy && s->gmean && s->gh && s->slabs && s->tcx && s->tcy && s->shared_buf && s->tl) ? 0 : -1;
}
static void scratch_free(JcScratch *s) {
  free(s->ga); free(s->gcx); free(s->gcy); free(s->gdx); free(s->gdy); free(s->gmean); free(s->gh); free(s->slabs);
  free(s->tcx); free(s->tcy); free(s->shared_buf); free(s->tl);
}

int jc_cpu_eval(const JcCtx *c, const JcLoss *lo, const real *W, const real *a, const real *cx, const real *cy, const real *dx,
                const real *dy, const real *alpha, const real *mean, const real *h, double *loss, real *ga, real *gcx, real *gcy,
                real *gdx, real *gdy, real *gmean, real *gh, real *model, int n_threads) {
  if (!c || !lo || !loss) return -2;
  JcScratch s;
  if (scratch_alloc(&s, c)) { scratch_free(&s); return -1; }
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  const int rc = eval_all(c, lo, W, a, cx, cy, dx, dy, alpha, mean, h, loss, s.ga, s.gcx, s.gcy, s.gdx, s.gdy, s.gmean, s.gh, model,
                          s.slabs, s.tcx, s.tcy, s.tl, s.shared_buf, 0);
  if (!rc) {
    const size_t NN = (size_t)c->N * c->N;
    if (ga) memcpy(ga, s.ga, sizeof(real) * (size_t)c->E * c->M);
    if (gcx) memcpy(gcx, s.gcx, sizeof(real) * (size_t)c->M);
    if (gcy) memcpy(gcy, s.gcy, sizeof(real) * (size_t)c->M);
    if (gdx) memcpy(gdx, s.gdx, sizeof(real) * (size_t)c->E);
    if (gdy) memcpy(gdy, s.gdy, sizeof(real) * (size_t)c->E);
    if (gmean) memcpy(gmean, s.gmean, sizeof(real) * (size_t)c->E);
    if (gh) memcpy(gh, s.gh, sizeof(real) * NN);
  }
  scratch_free(&s);
  return rc;
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

/* n_iter AdaBelief iterations.  free_mask[7]: a, c_x, c_y, dx, dy, mean, h.  Parameters and both moments in / out, the
 * moments in the layout [a (E M) | cx (M) | cy (M) | dx (E) | dy (E) | mean (E) | h (N^2)].  loss_hist[n_iter + 1]: the loss
 * before every update, then the loss of the final parameters. */
int jc_cpu_run(const JcCtx *c, const JcLoss *lo, const real *W, real *a, real *cx, real *cy, real *dx, real *dy, const real *alpha,
               real *mean, real *h, real *mom_m, real *mom_s, const int *free_mask, real lr0, int schedule, int t0, int n_iter,
               double *loss_hist, int n_threads) {
  if (!c || !lo || !loss_hist) return -2;
  const int E = c->E, M = c->M;
  const size_t EM = (size_t)E * M, NN = (size_t)c->N * c->N;
  JcScratch s;
  if (scratch_alloc(&s, c)) { scratch_free(&s); return -1; }
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  real *m_a = mom_m, *m_cx = m_a + EM, *m_cy = m_cx + M, *m_dx = m_cy + M, *m_dy = m_dx + E, *m_mean = m_dy + E, *m_h = m_mean + E;
  real *s_a = mom_s, *s_cx = s_a + EM, *s_cy = s_cx + M, *s_dx = s_cy + M, *s_dy = s_dx + E, *s_mean = s_dy + E, *s_h = s_mean + E;
  int rc = 0;
  for (int it = 0; it <= n_iter && !rc; ++it) {
    double L = 0;
    rc = eval_all(c, lo, W, a, cx, cy, dx, dy, alpha, mean, h, &L, s.ga, s.gcx, s.gcy, s.gdx, s.gdy, s.gmean, s.gh, NULL, s.slabs, s.tcx,
                  s.tcy, s.tl, s.shared_buf, 0);
    if (rc) break;
    loss_hist[it] = L;
    if (it == n_iter) break;
    const int t = t0 + it;
    const real lr = (real)(schedule ? (double)lr0 * pow(0.99, (double)t / 10.0) : (double)lr0);
    const real bc1 = (real)(1.0 / (1.0 - pow(0.9, t + 1.0))), bc2 = (real)(1.0 / (1.0 - pow(0.999, t + 1.0)));
    if (free_mask[0]) for (size_t k = 0; k < EM; ++k) adabelief_step(&a[k], &m_a[k], &s_a[k], s.ga[k], lr, bc1, bc2);
    if (free_mask[1]) for (int i = 0; i < M; ++i) adabelief_step(&cx[i], &m_cx[i], &s_cx[i], s.gcx[i], lr, bc1, bc2);
    if (free_mask[2]) for (int i = 0; i < M; ++i) adabelief_step(&cy[i], &m_cy[i], &s_cy[i], s.gcy[i], lr, bc1, bc2);
    if (free_mask[3]) for (int e = 0; e < E; ++e) adabelief_step(&dx[e], &m_dx[e], &s_dx[e], s.gdx[e], lr, bc1, bc2);
    if (free_mask[4]) for (int e = 0; e < E; ++e) adabelief_step(&dy[e], &m_dy[e], &s_dy[e], s.gdy[e], lr, bc1, bc2);
    if (free_mask[5]) for (int e = 0; e < E; ++e) adabelief_step(&mean[e], &m_mean[e], &s_mean[e], s.gmean[e], lr, bc1, bc2);
    if (free_mask[6]) {
#pragma omp parallel for schedule(static)
      for (long k = 0; k < (long)NN; ++k) adabelief_step(&h[k], &m_h[k], &s_h[k], s.gh[k], lr, bc1, bc2);
    }
  }
  scratch_free(&s);
  return rc;
}
