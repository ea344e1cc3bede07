// PSF-fit batch object behind the C ABI (include/lcmi.h, "PSF fit" section).
#include <cmath>
#include <cstring>

#include "lbfgs_host.h"
#include <thread>

#include "noise_host.h"
#include "starlet_norms.h"
#include <cstdlib>

#include "psf_kernels.h"
#include "psf_noise.h"
#include "psf_distort.h"
#include "psf_lbfgs.h"

using namespace lc;

struct lc_psf_batch {
  lc_ctx *ctx = nullptr;
  int F = 0, S = 0, n = 0, ss = 0, N = 0, J = 0;
  float *data = nullptr, *wgt = nullptr, *W = nullptr, *norms = nullptr, *Tm = nullptr;
  float *B = nullptr, *mB = nullptr, *sB = nullptr;
  float *stars = nullptr, *stars_m = nullptr, *stars_s = nullptr, *moffat = nullptr;
  float *hist = nullptr, *qscratch = nullptr;
  float *sched = nullptr;  // AdaBelief schedule table of the current launch [cap][3]
  int sched_cap = 0;
  lc_adabelief_cfg sched_cfg{};
  float *ntab = nullptr;  // noise propagation: 1-D starlet tables [F][S][J][3][2][N]
  float *B1 = nullptr, *mB1 = nullptr, *sB1 = nullptr;  // role 1's copy of the pixel state (n = 64)
  float *xch = nullptr;   // two-workgroup form: exchange slabs, flags, abort word
  int *xflags = nullptr;
  bool split_used = false;
  int launch_seq = 0;       // sequence number of the two-workgroup launches (abort word protocol, psf_kernels.h)
  float *bkB = nullptr, *bkmB = nullptr, *bksB = nullptr, *bkstars = nullptr, *bkstars_m = nullptr, *bkstars_s = nullptr;
  int split_blocks_per_cu = -1;  // occupancy of the two-workgroup kernel (queried once)
  // field distortion (csrc/psf_distort.h): Moffat given by its quadratic form, distortion of the stars of every frame,
  // external gradient of B
  bool moffat_is_q = false;
  float *moffat_q = nullptr, *o_gq = nullptr;  // [F][4]
  float *dist_coef = nullptr, *dist_xy = nullptr, *ext_grad = nullptr;  // [F][9], [F][S_stars][2], [F][N*N]
  int dist_S = 0;
  bool use_ext_grad = false;
  float *o_loss = nullptr, *o_chi2 = nullptr, *o_gstars = nullptr, *o_ggrid = nullptr, *o_gT = nullptr,
        *o_model = nullptr, *o_gmoffat = nullptr;
  float *narrow = nullptr, *full = nullptr, *resid = nullptr, *redchi2 = nullptr;
  bool have_W = false;
  float lam_sc = 0.f, lam_hf = 0.f;
  int hist_stride = 0, iters_done = 0;
  std::vector<void *> allocs;
};

namespace {

template <class T>
int dmalloc(lc_psf_batch *b, T **p, size_t count) {
  LC_HIP(b->ctx, hipMalloc((void **)p, count * sizeof(T)));
  b->allocs.push_back(*p);
  LC_HIP(b->ctx, hipMemsetAsync(*p, 0, count * sizeof(T), b->ctx->stream));
  return LC_OK;
}

// ---- Moffat rasterisation and its parameter gradient (double precision, one block per frame) ----
// (the frame's constants - 2^(1/beta), the axis scales, the rotation - are formed once per thread, not per pixel: the
//  stage-A optimiser rasterises all frames once per trial point)
struct MoffatFrame {
  double beta, ax, ay, cs, sn, s_over_kb, dax_db, day_db;
  int c;
};
__device__ inline MoffatFrame moffat_frame(int N, int ss, const float *par) {
  MoffatFrame m;
  const double fx = par[0], fy = par[1], phi = par[2];
  m.beta = par[3];
  m.c = (N - 1) / 2;
  const double p2 = pow(2.0, 1.0 / m.beta);
  const double kb = 2.0 * sqrt(p2 - 1.0);
  m.ax = ss * fx / kb;
  m.ay = ss * fy / kb;
  m.cs = cos(phi);
  m.sn = sin(phi);
  m.s_over_kb = ss / kb;
  const double dkb_db = (1.0 / sqrt(p2 - 1.0)) * p2 * log(2.0) * (-1.0 / (m.beta * m.beta));
  m.dax_db = (-ss * fx / (kb * kb)) * dkb_db;
  m.day_db = (-ss * fy / (kb * kb)) * dkb_db;
  return m;
}
__device__ inline void moffat_terms(const MoffatFrame &m, int u, int v, double &M, double dM[4]) {
  const double x = v - m.c, y = u - m.c;
  const double xr = x * m.cs + y * m.sn, yr = -x * m.sn + y * m.cs;
  const double A = xr * xr / (m.ax * m.ax) + yr * yr / (m.ay * m.ay);
  M = pow(1.0 + A, -m.beta);
  const double Mb1 = M / (1.0 + A);  // (1+A)^(-beta-1)
  const double dM_dax = 2.0 * m.beta * Mb1 * xr * xr / (m.ax * m.ax * m.ax);
  const double dM_day = 2.0 * m.beta * Mb1 * yr * yr / (m.ay * m.ay * m.ay);
  dM[0] = dM_dax * m.s_over_kb;
  dM[1] = dM_day * m.s_over_kb;
  dM[2] = -m.beta * Mb1 * 2.0 * xr * yr * (1.0 / (m.ax * m.ax) - 1.0 / (m.ay * m.ay));
  dM[3] = -log(1.0 + A) * M + dM_dax * m.dax_db + dM_day * m.day_db;
}

__device__ inline double block_sum_d(double v, double *sh) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ void moffat_raster_kernel(int N, int ss, const float *par, float *Tm) {
  __shared__ double sh[256];
  const int f = blockIdx.x;
  const MoffatFrame mf = moffat_frame(N, ss, par + f * 4);
  double acc = 0;
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    double M, dM[4];
    moffat_terms(mf, i / N, i % N, M, dM);
    acc += M;
  }
  const double S = block_sum_d(acc, sh);
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    double M, dM[4];
    moffat_terms(mf, i / N, i % N, M, dM);
    Tm[(size_t)f * N * N + i] = (float)(M / S);
  }
}

__global__ void moffat_grad_kernel(int N, int ss, const float *par, const float *gT, float *gmoffat) {
  __shared__ double sh[256];
  const int f = blockIdx.x;
  const MoffatFrame mf = moffat_frame(N, ss, par + f * 4);
  double sM = 0, sdM[4] = {0, 0, 0, 0}, gM = 0, gdM[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    double M, dM[4];
    moffat_terms(mf, i / N, i % N, M, dM);
    const double g = gT[(size_t)f * N * N + i];
    sM += M;
    gM += g * M;
    for (int k = 0; k < 4; ++k) {
      sdM[k] += dM[k];
      gdM[k] += g * dM[k];
    }
  }
  const double S = block_sum_d(sM, sh);
  const double G = block_sum_d(gM, sh);
  for (int k = 0; k < 4; ++k) {
    const double a = block_sum_d(sdM[k], sh);
    const double b = block_sum_d(gdM[k], sh);
    if (threadIdx.x == 0) gmoffat[f * 4 + k] = (float)((b - G * a / S) / S);
  }
}

// ---- outputs: narrow / full PSF, residuals, reduced chi2 ------------------------------------------
__global__ void psf_finalize_kernel(int N, const float *Tm, const float *B, float *narrow, float *full) {
  __shared__ double sh[256];
  const int f = blockIdx.x;
  const size_t o = (size_t)f * N * N;
  double acc = 0;
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) acc += (double)Tm[o + i] + (double)B[o + i];
  const double S = block_sum_d(acc, sh);
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) narrow[o + i] = (float)(((double)Tm[o + i] + (double)B[o + i]) / S);
  __syncthreads();
  // full = conv_same(G(centre), narrow): separable taps phi(t) = N(t; delta, sigma), delta = (N-1)/2 - (N-1)//2
  const double delta = (N % 2 == 0) ? 0.5 : 0.0;
  double tp[2 * kRg + 2];
  const int o0 = (int)nearbyint(delta);
  for (int k = 0; k <= 2 * kRg; ++k) {
    const double x = (o0 - kRg + k) - delta;
    tp[k] = exp(-0.5 * x * x / ((double)kSigmaG * kSigmaG)) / (sqrt(2.0 * M_PI) * kSigmaG);
  }
  acc = 0;
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    const int u = i / N, v = i % N;
    double a = 0;
    for (int ku = 0; ku <= 2 * kRg; ++ku) {
      const int uu = u - (o0 - kRg + ku);
      if (uu < 0 || uu >= N) continue;
      double r = 0;
      for (int kv = 0; kv <= 2 * kRg; ++kv) {
        const int vv = v - (o0 - kRg + kv);
        if (vv < 0 || vv >= N) continue;
        r += tp[kv] * narrow[o + (size_t)uu * N + vv];
      }
      a += tp[ku] * r;
    }
    full[o + i] = (float)a;
    acc += a;
  }
  const double SF = block_sum_d(acc, sh);
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) full[o + i] = (float)(full[o + i] / SF);
}

__global__ void psf_residual_kernel(int S, int n, const float *data, const float *wgt, const float *model,
                                    float *resid, float *redchi2) {
  __shared__ double sh[256];
  const int f = blockIdx.x;
  const size_t o = (size_t)f * S * n * n;
  double chi = 0, cnt = 0;
  for (int i = threadIdx.x; i < S * n * n; i += blockDim.x) {
    const float r = data[o + i] - model[o + i];
    resid[o + i] = r;
    chi += (double)wgt[o + i] * r * r;
    cnt += wgt[o + i] > 0.f ? 1.0 : 0.0;
  }
  const double C = block_sum_d(chi, sh);
  const double K = block_sum_d(cnt, sh);
  if (threadIdx.x == 0) redchi2[f] = (float)(C / (K > 0 ? K : 1.0));
}

typedef void (*psf_kernel_fn)(PsfArgs);
struct PsfVariant {
  int n, ss;
  psf_kernel_fn fn;
  int nthr, lds_bytes;
  psf_kernel_fn fn_split;  // two workgroups per frame, or null
};

template <class C, bool WITH_SPLIT = false>
PsfVariant make_variant() {
  psf_kernel_fn split = nullptr;
  if constexpr (WITH_SPLIT) split = psf_fit_kernel<C, true>;
  return PsfVariant{C::n, C::SS, psf_fit_kernel<C>, C::NTHR, (int)(C::LDS_FLOATS * sizeof(float)), split};
}

const PsfVariant *find_variant(int n, int ss) {
  static const PsfVariant table[] = {
      make_variant<PsfCfg<16, 1, 4, 8, true>>(),    // n = 16, ss = 1 (reference test fixture size)
      make_variant<PsfCfg<32, 2, 4, 8, true>, true>(),    // n = 16, ss = 2
      make_variant<PsfCfg<48, 2, 4, 8, true>, true>(),    // n = 24 (lightcurver default stamp_size_stars)
      make_variant<PsfCfg<64, 2, 8, 8, true>, true>(),    // n = 32 (C1, C2): all (up to) 8 stars in one group
      make_variant<PsfCfg<128, 2, 16, 1>, true>(),  // n = 64 (C3)
  };
  for (const auto &v : table)
    if (v.n == n && v.ss == ss) return &v;
  return nullptr;
}

int launch_psf(lc_psf_batch *b, int mode, int n_iter, const lc_adabelief_cfg *cfg, bool want_outputs,
               bool reg) {
  const PsfVariant *v = find_variant(b->n, b->ss);
  if (!v) LC_FAIL(b->ctx, LC_ERR_UNSUPPORTED, "no PSF kernel instantiated for this stamp size");
  PsfArgs A;
  std::memset(&A, 0, sizeof(A));
  A.F = b->F;
  A.S = b->S;
  A.n_iter = n_iter;
  A.t0 = b->iters_done;
  A.hist_stride = b->hist_stride;
  A.mode = mode;
  A.data = b->data;
  A.wgt = b->wgt;
  A.W = b->have_W ? b->W : nullptr;
  A.norms = b->norms;
  A.Tm = b->Tm;
  A.B = b->B;
  A.mB = b->mB;
  A.sB = b->sB;
  A.stars = b->stars;
  A.stars_m = b->stars_m;
  A.stars_s = b->stars_s;
  A.hist = b->hist;
  A.qscratch = b->qscratch;
  if (want_outputs) {
    A.out_loss = b->o_loss;
    A.out_chi2 = b->o_chi2;
    A.out_gstars = b->o_gstars;
    A.out_ggrid = b->o_ggrid;
    A.out_gT = b->o_gT;
    A.out_model = b->o_model;
  }
  A.ext_grad = b->use_ext_grad ? b->ext_grad : nullptr;
  {
    const char *xf = std::getenv("LCMI_PSF_XCD_FAST");
    A.xcd_fast = !(xf && xf[0] == '0');
  }
  A.lam_sc = reg ? b->lam_sc : 0.f;
  A.lam_hf = reg ? b->lam_hf : 0.f;
  if (cfg) A.ab = *cfg; else lc_adabelief_defaults(&A.ab);
  if (mode == 1) {
    // learning rate and bias corrections by absolute iteration, evaluated here in double; the table is rebuilt only
    // when the optimiser settings change or the iteration count outgrows it, so back-to-back launches stay asynchronous
    const int need = A.t0 + n_iter;
    if (need > b->sched_cap || std::memcmp(&A.ab, &b->sched_cfg, sizeof(A.ab)) != 0) {
      const int cap = std::max(need * 2, 4096);
      if (cap > b->sched_cap) {
        b->sched = nullptr;  // the old table stays in the allocation list until the object goes
        int rc = dmalloc(b, &b->sched, (size_t)3 * cap);
        if (rc) return rc;
        b->sched_cap = cap;
      }
      std::vector<float> tab((size_t)3 * b->sched_cap);
      for (int t = 0; t < b->sched_cap; ++t) adabelief_schedule(A.ab, t, tab[3 * t], tab[3 * t + 1], tab[3 * t + 2]);
      LC_HIP(b->ctx, hipMemcpyAsync(b->sched, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, b->ctx->stream));
      LC_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));  // `tab` is pageable and goes out of scope
      b->sched_cfg = A.ab;
    }
    A.sched = b->sched;
  }
  // Two workgroups per frame when the optimisation loop would otherwise leave more than half of the CUs idle.
  // Every workgroup of that grid must be resident at once (partners wait for each other): the grid has to fit in
  // (resident workgroups per CU, from the occupancy calculator) x (number of CUs).  Another process or stream can
  // still hold CUs; a partner that does not show up makes the launch give up and the fall-back launch behind it redo
  // the work in the one-workgroup form (abort word protocol, psf_kernels.h), so the result never depends on it.
  const int split_grid = ((b->F + 7) / 8) * 16;
  // (not for launches of a few iterations - the step-by-step drive of the distortion fit: the copies of the pre-launch state
  // the fall-back needs cost ~35 us per launch, the two-workgroup form gains ~9 us per iteration)
  bool split = mode == 1 && n_iter >= 4 && v->fn_split && (A.lam_sc != 0.f || A.lam_hf != 0.f) && !std::getenv("LCMI_PSF_SINGLE_WG");
  if (split) {
    LC_HIP(b->ctx, hipFuncSetAttribute((const void *)v->fn_split, hipFuncAttributeMaxDynamicSharedMemorySize, v->lds_bytes));
    if (b->split_blocks_per_cu < 0) {
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)v->fn_split, v->nthr, v->lds_bytes) != hipSuccess) per_cu = 0;
      b->split_blocks_per_cu = per_cu;
    }
    split = (long long)split_grid <= (long long)b->split_blocks_per_cu * b->ctx->n_cu;
  }
  if (split) {
    const size_t FNN = (size_t)b->F * b->N * b->N, FS4 = (size_t)b->F * b->S * 4;
    if (!b->xch) {
      int rc = dmalloc(b, &b->xch, (size_t)b->F * 4 * ((size_t)b->N * b->N + 64));
      if (rc) return rc;
      if ((rc = dmalloc(b, &b->xflags, (size_t)b->F * 2 * kXFlagStride + 16))) return rc;
      if ((rc = dmalloc(b, &b->B1, FNN)) || (rc = dmalloc(b, &b->mB1, FNN)) || (rc = dmalloc(b, &b->sB1, FNN))) return rc;
      if ((rc = dmalloc(b, &b->bkB, FNN)) || (rc = dmalloc(b, &b->bkmB, FNN)) || (rc = dmalloc(b, &b->bksB, FNN)) ||
          (rc = dmalloc(b, &b->bkstars, FS4)) || (rc = dmalloc(b, &b->bkstars_m, FS4)) || (rc = dmalloc(b, &b->bkstars_s, FS4)))
        return rc;
    }
    hipStream_t q = b->ctx->stream;
    // the 2F iteration flags restart at zero; the abort word and the fall-back counter behind them are never cleared
    LC_HIP(b->ctx, hipMemsetAsync(b->xflags, 0, (size_t)b->F * 2 * kXFlagStride * sizeof(int), q));
    // pre-launch state, for the fall-back launch
    LC_HIP(b->ctx, hipMemcpyAsync(b->bkB, b->B, FNN * sizeof(float), hipMemcpyDeviceToDevice, q));
    LC_HIP(b->ctx, hipMemcpyAsync(b->bkmB, b->mB, FNN * sizeof(float), hipMemcpyDeviceToDevice, q));
    LC_HIP(b->ctx, hipMemcpyAsync(b->bksB, b->sB, FNN * sizeof(float), hipMemcpyDeviceToDevice, q));
    LC_HIP(b->ctx, hipMemcpyAsync(b->bkstars, b->stars, FS4 * sizeof(float), hipMemcpyDeviceToDevice, q));
    LC_HIP(b->ctx, hipMemcpyAsync(b->bkstars_m, b->stars_m, FS4 * sizeof(float), hipMemcpyDeviceToDevice, q));
    LC_HIP(b->ctx, hipMemcpyAsync(b->bkstars_s, b->stars_s, FS4 * sizeof(float), hipMemcpyDeviceToDevice, q));
    A.xch = b->xch;
    A.xflags = b->xflags;
    A.xabort = b->xflags + (size_t)b->F * 2 * kXFlagStride;
    A.heal_count = b->xflags + (size_t)b->F * 2 * kXFlagStride + 1;
    A.B1 = b->B1;
    A.mB1 = b->mB1;
    A.sB1 = b->sB1;
    A.launch_seq = ++b->launch_seq;
    A.heal = 0;
    A.force_abort_it = -1;
    if (const char *fa = std::getenv("LCMI_PSF_FORCE_ABORT")) A.force_abort_it = std::atoi(fa);  // test hook
    A.bkB = b->bkB;
    A.bkmB = b->bkmB;
    A.bksB = b->bksB;
    A.bkstars = b->bkstars;
    A.bkstars_m = b->bkstars_m;
    A.bkstars_s = b->bkstars_s;
    b->split_used = true;
    hipLaunchKernelGGL(v->fn_split, dim3(split_grid), dim3(v->nthr), v->lds_bytes, q, A);
    LC_HIP(b->ctx, hipGetLastError());
    // fall-back: the same launch in the one-workgroup form; every workgroup returns at once unless the launch above gave up
    A.heal = 1;
    LC_HIP(b->ctx, hipFuncSetAttribute((const void *)v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, v->lds_bytes));
    hipLaunchKernelGGL(v->fn, dim3(b->F), dim3(v->nthr), v->lds_bytes, q, A);
  } else {
    A.force_abort_it = -1;
    LC_HIP(b->ctx, hipFuncSetAttribute((const void *)v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, v->lds_bytes));
    hipLaunchKernelGGL(v->fn, dim3(b->F), dim3(v->nthr), v->lds_bytes, b->ctx->stream, A);
  }
  LC_HIP(b->ctx, hipGetLastError());
  return LC_OK;
}

// how many two-workgroup launches of this batch gave up and were redone in the one-workgroup form
int read_split_fallbacks(lc_psf_batch *b, int *count) {
  *count = 0;
  if (!b->split_used) return LC_OK;
  LC_HIP(b->ctx, hipMemcpyAsync(count, b->xflags + (size_t)b->F * 2 * kXFlagStride + 1, sizeof(int), hipMemcpyDeviceToHost, b->ctx->stream));
  LC_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));
  return LC_OK;
}

int ensure_hist(lc_psf_batch *b, int needed) {
  if (needed <= b->hist_stride) return LC_OK;
  int ns = std::max(needed, 2 * b->hist_stride + 64);
  float *nh = nullptr;
  LC_HIP(b->ctx, hipMalloc((void **)&nh, (size_t)b->F * ns * sizeof(float)));
  LC_HIP(b->ctx, hipMemsetAsync(nh, 0, (size_t)b->F * ns * sizeof(float), b->ctx->stream));
  if (b->hist) {
    LC_HIP(b->ctx, hipMemcpy2DAsync(nh, ns * sizeof(float), b->hist, b->hist_stride * sizeof(float),
                                     b->hist_stride * sizeof(float), b->F, hipMemcpyDeviceToDevice, b->ctx->stream));
    LC_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));
    hipFree(b->hist);
  }
  b->hist = nh;
  b->hist_stride = ns;
  return LC_OK;
}

int h2d(lc_psf_batch *b, void *dst, const void *src, size_t bytes) {
  LC_HIP(b->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, b->ctx->stream));
  LC_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));
  return LC_OK;
}
int d2h(lc_psf_batch *b, void *dst, const void *src, size_t bytes) {
  LC_HIP(b->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, b->ctx->stream));
  LC_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));
  return LC_OK;
}

}  // namespace

extern "C" {

int lc_psf_supported(int n, int ss) { return find_variant(n, ss) != nullptr; }

int lc_psf_batch_create(lc_ctx *ctx, int F, int S_max, int n, int ss, const float *data, const float *weight,
                        lc_psf_batch **out) {
  if (!ctx || !out || !data || !weight || F <= 0 || S_max <= 0) {
    if (ctx) ctx->err = "lc_psf_batch_create: invalid argument";
    return LC_ERR_INVALID;
  }
  if (S_max > 16) LC_FAIL(ctx, LC_ERR_UNSUPPORTED, "at most 16 stars per frame");
  if (!find_variant(n, ss)) LC_FAIL(ctx, LC_ERR_UNSUPPORTED, "no PSF kernel instantiated for this stamp size");
  LC_HIP(ctx, hipSetDevice(ctx->device));
  lc_psf_batch *b = new lc_psf_batch();
  b->ctx = ctx;
  b->F = F;
  b->S = S_max;
  b->n = n;
  b->ss = ss;
  b->N = n * ss;
  b->J = ilog2(b->N);
  const size_t NN = (size_t)b->N * b->N, nn = (size_t)n * n;
  int rc = 0;
#define TRY(x)            \
  if ((rc = (x)) != 0) {  \
    lc_psf_batch_destroy(b); \
    return rc;            \
  }
  TRY(dmalloc(b, &b->data, F * S_max * nn));
  TRY(dmalloc(b, &b->wgt, F * S_max * nn));
  TRY(dmalloc(b, &b->W, F * (size_t)b->J * NN));
  TRY(dmalloc(b, &b->norms, b->J + 1));
  TRY(dmalloc(b, &b->Tm, F * NN));
  TRY(dmalloc(b, &b->B, F * NN));
  TRY(dmalloc(b, &b->mB, F * NN));
  TRY(dmalloc(b, &b->sB, F * NN));
  TRY(dmalloc(b, &b->stars, (size_t)F * S_max * 4));
  TRY(dmalloc(b, &b->stars_m, (size_t)F * S_max * 4));
  TRY(dmalloc(b, &b->stars_s, (size_t)F * S_max * 4));
  TRY(dmalloc(b, &b->moffat, (size_t)F * 4));
  TRY(dmalloc(b, &b->qscratch, F * (size_t)b->J * NN));
  TRY(dmalloc(b, &b->o_loss, F));
  TRY(dmalloc(b, &b->o_chi2, F));
  TRY(dmalloc(b, &b->o_gstars, (size_t)F * S_max * 4));
  TRY(dmalloc(b, &b->o_ggrid, F * NN));
  TRY(dmalloc(b, &b->o_gT, F * NN));
  TRY(dmalloc(b, &b->o_model, F * S_max * nn));
  TRY(dmalloc(b, &b->o_gmoffat, (size_t)F * 4));
  TRY(dmalloc(b, &b->narrow, F * NN));
  TRY(dmalloc(b, &b->full, F * NN));
  TRY(dmalloc(b, &b->resid, F * S_max * nn));
  TRY(dmalloc(b, &b->redchi2, F));
  TRY(ensure_hist(b, 64));
  // sanitise inputs: non-finite data or weight -> weight 0 (NaN handling of psf_modelling.py:136-140)
  {
    std::vector<float> d(data, data + F * S_max * nn), w(weight, weight + F * S_max * nn);
    for (size_t i = 0; i < d.size(); ++i)
      if (!std::isfinite(d[i]) || !std::isfinite(w[i]) || w[i] < 0.f) {
        d[i] = 0.f;
        w[i] = 0.f;
      }
    TRY(h2d(b, b->data, d.data(), d.size() * sizeof(float)));
    TRY(h2d(b, b->wgt, w.data(), w.size() * sizeof(float)));
  }
  // starlet scale norms (the l1 weights used when no weight map is set)
  {
    std::vector<float> norms;
    starlet_scale_norms(b->N, b->J, norms);
    TRY(h2d(b, b->norms, norms.data(), norms.size() * sizeof(float)));
  }
#undef TRY
  *out = b;
  return LC_OK;
}

void lc_psf_batch_destroy(lc_psf_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->ctx->device);
  hipStreamSynchronize(b->ctx->stream);
  for (void *p : b->allocs) hipFree(p);
  if (b->hist) hipFree(b->hist);
  delete b;
}

int lc_psf_batch_set_moffat(lc_psf_batch *b, const float *moffat) {
  if (!b || !moffat) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  int rc = h2d(b, b->moffat, moffat, (size_t)b->F * 4 * sizeof(float));
  if (rc) return rc;
  hipLaunchKernelGGL(moffat_raster_kernel, dim3(b->F), dim3(256), 0, b->ctx->stream, b->N, b->ss, b->moffat, b->Tm);
  LC_HIP(b->ctx, hipGetLastError());
  b->moffat_is_q = false;
  return LC_OK;
}
int lc_psf_batch_set_moffat_q(lc_psf_batch *b, const float *q) {
  if (!b || !q) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  int rc;
  if (!b->moffat_q && (rc = dmalloc(b, &b->moffat_q, (size_t)b->F * 4))) return rc;
  if ((rc = h2d(b, b->moffat_q, q, (size_t)b->F * 4 * sizeof(float)))) return rc;
  hipLaunchKernelGGL(moffat_q_raster_kernel, dim3(b->F), dim3(256), 0, b->ctx->stream, b->N, b->moffat_q, b->Tm);
  LC_HIP(b->ctx, hipGetLastError());
  b->moffat_is_q = true;
  return LC_OK;
}
int lc_psf_batch_set_distortion(lc_psf_batch *b, int S_stars, const float *coeffs, const float *xy) {
  if (!b || !coeffs || !xy || S_stars <= 0 || S_stars > 16) return LC_ERR_INVALID;  // (16 stars per frame: the library's limit)
  LC_ENTER(b->ctx);
  for (size_t i = 0; i < (size_t)b->F * 9; ++i)
    if (!std::isfinite(coeffs[i]) || std::fabs(coeffs[i]) > 0.25f)
      LC_FAIL(b->ctx, LC_ERR_INVALID, "lc_psf_batch_set_distortion: coefficients must be finite and within +-0.25");
  int rc;
  if (b->dist_S != S_stars) {
    b->dist_coef = b->dist_xy = nullptr;  // earlier buffers stay in the allocation list until the object goes
    if ((rc = dmalloc(b, &b->dist_coef, (size_t)b->F * 9)) || (rc = dmalloc(b, &b->dist_xy, (size_t)b->F * S_stars * 2))) return rc;
    if (!b->ext_grad && (rc = dmalloc(b, &b->ext_grad, (size_t)b->F * b->N * b->N))) return rc;
    b->dist_S = S_stars;
  }
  if ((rc = h2d(b, b->dist_coef, coeffs, (size_t)b->F * 9 * sizeof(float)))) return rc;
  return h2d(b, b->dist_xy, xy, (size_t)b->F * S_stars * 2 * sizeof(float));
}
// frames->B resampled at every star -> stars->B (stars: one single-star frame per (frame, star), frame-major)
int lc_psf_distortion_forward(lc_psf_batch *frames, lc_psf_batch *stars) {
  if (!frames || !stars) return LC_ERR_INVALID;
  LC_ENTER(frames->ctx);
  if (frames->ctx != stars->ctx || frames->dist_S <= 0 || stars->F != frames->F * frames->dist_S || stars->N != frames->N)
    LC_FAIL(frames->ctx, LC_ERR_INVALID, "lc_psf_distortion_forward: the star batch must hold F * S single-star frames of the same grid");
  hipLaunchKernelGGL(psf_warp_kernel, dim3(stars->F), dim3(256), 0, frames->ctx->stream, frames->N, frames->dist_S,
                     frames->dist_coef, frames->dist_xy, frames->B, stars->B);
  LC_HIP(frames->ctx, hipGetLastError());
  return LC_OK;
}
// adjoint: the gradients d loss / d B of the star batch (from its last step / evaluation) -> frames' external gradient
int lc_psf_distortion_backward(lc_psf_batch *frames, lc_psf_batch *stars) {
  if (!frames || !stars) return LC_ERR_INVALID;
  LC_ENTER(frames->ctx);
  if (frames->ctx != stars->ctx || frames->dist_S <= 0 || stars->F != frames->F * frames->dist_S || stars->N != frames->N)
    LC_FAIL(frames->ctx, LC_ERR_INVALID, "lc_psf_distortion_backward: the star batch must hold F * S single-star frames of the same grid");
  hipLaunchKernelGGL(psf_warp_adjoint_kernel, dim3(frames->F, (frames->N * frames->N + 255) / 256), dim3(256), 0, frames->ctx->stream, frames->N, frames->dist_S,
                     frames->dist_coef, frames->dist_xy, stars->o_ggrid, frames->ext_grad);
  LC_HIP(frames->ctx, hipGetLastError());
  frames->use_ext_grad = true;
  return LC_OK;
}
int lc_psf_batch_get_ext_grad(lc_psf_batch *b, float *grad) {
  if (!b || !grad) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  if (!b->ext_grad) LC_FAIL(b->ctx, LC_ERR_INVALID, "no external gradient: lc_psf_batch_set_distortion first");
  return d2h(b, grad, b->ext_grad, (size_t)b->F * b->N * b->N * sizeof(float));
}
// one AdaBelief iteration; export_grad: also leave d loss / d B of that iteration on the device for
// lc_psf_distortion_backward; use_ext_grad: add the external gradient to d loss / d B
int lc_psf_batch_step_adabelief(lc_psf_batch *b, const lc_adabelief_cfg *cfg, int use_ext_grad, int export_grad) {
  if (!b) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  if (use_ext_grad && !b->ext_grad) LC_FAIL(b->ctx, LC_ERR_INVALID, "no external gradient: lc_psf_batch_set_distortion first");
  int rc = ensure_hist(b, b->iters_done + 2);
  if (rc) return rc;
  b->use_ext_grad = use_ext_grad != 0;
  rc = launch_psf(b, 1, 1, cfg, export_grad != 0, true);
  if (rc) return rc;
  b->iters_done += 1;
  return LC_OK;
}
// The pixel-grid stage of build_psf(field_distortion=True) as ONE call: n_iter times { forward, step of `stars` with its
// gradient exported, backward, step of `frames` with the external gradient }, then one more forward so that the star batch
// holds the final resampled grid - the loop the host language used to drive call by call (4 n_iter ABI calls).
int lc_psf_distortion_run(lc_psf_batch *frames, lc_psf_batch *stars, int n_iter, const lc_adabelief_cfg *cfg) {
  if (!frames || !stars || n_iter < 0) return LC_ERR_INVALID;
  int rc = LC_OK;
  for (int it = 0; it < n_iter && !rc; ++it) {
    if ((rc = lc_psf_distortion_forward(frames, stars))) break;
    if ((rc = lc_psf_batch_step_adabelief(stars, cfg, 0, 1))) break;
    if ((rc = lc_psf_distortion_backward(frames, stars))) break;
    rc = lc_psf_batch_step_adabelief(frames, cfg, 1, 0);
  }
  if (!rc) rc = lc_psf_distortion_forward(frames, stars);
  return rc;
}
int lc_psf_batch_get_moffat(lc_psf_batch *b, float *moffat) {
  if (!b || !moffat) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  return d2h(b, moffat, b->moffat, (size_t)b->F * 4 * sizeof(float));
}
int lc_psf_batch_set_stars(lc_psf_batch *b, const float *stars) {
  if (!b || !stars) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  return h2d(b, b->stars, stars, (size_t)b->F * b->S * 4 * sizeof(float));
}
int lc_psf_batch_get_stars(lc_psf_batch *b, float *stars) {
  if (!b || !stars) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  return d2h(b, stars, b->stars, (size_t)b->F * b->S * 4 * sizeof(float));
}
int lc_psf_batch_set_grid(lc_psf_batch *b, const float *grid) {
  if (!b) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  const size_t bytes = (size_t)b->F * b->N * b->N * sizeof(float);
  if (grid) return h2d(b, b->B, grid, bytes);
  LC_HIP(b->ctx, hipMemsetAsync(b->B, 0, bytes, b->ctx->stream));
  LC_HIP(b->ctx, hipMemsetAsync(b->mB, 0, bytes, b->ctx->stream));
  LC_HIP(b->ctx, hipMemsetAsync(b->sB, 0, bytes, b->ctx->stream));
  LC_HIP(b->ctx, hipMemsetAsync(b->stars_m, 0, (size_t)b->F * b->S * 4 * sizeof(float), b->ctx->stream));
  LC_HIP(b->ctx, hipMemsetAsync(b->stars_s, 0, (size_t)b->F * b->S * 4 * sizeof(float), b->ctx->stream));
  b->iters_done = 0;
  return LC_OK;
}
int lc_psf_batch_get_grid(lc_psf_batch *b, float *grid) {
  if (!b || !grid) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  return d2h(b, grid, b->B, (size_t)b->F * b->N * b->N * sizeof(float));
}
int lc_psf_batch_set_regularization(lc_psf_batch *b, const float *W, float lam_scales, float lam_hf) {
  if (!b) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  b->lam_sc = lam_scales;
  b->lam_hf = lam_hf;
  if (W) {
    int rc = h2d(b, b->W, W, (size_t)b->F * b->J * b->N * b->N * sizeof(float));
    if (rc) return rc;
    b->have_W = true;
  }
  return LC_OK;
}
namespace {
typedef void (*noise_fn)(int, int, const float *, const float *, const float *, float *);
noise_fn find_noise_kernel(int N, int ss) {
  if (N == 16 && ss == 1) return psf_noise_accumulate_kernel<16, 1>;
  if (N == 32 && ss == 2) return psf_noise_accumulate_kernel<32, 2>;
  if (N == 48 && ss == 2) return psf_noise_accumulate_kernel<48, 2>;
  if (N == 64 && ss == 2) return psf_noise_accumulate_kernel<64, 2>;
  if (N == 128 && ss == 2) return psf_noise_accumulate_kernel<128, 2>;
  return nullptr;
}
}  // namespace

int lc_psf_batch_propagate_noise(lc_psf_batch *b) {
  if (!b) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  const int F = b->F, S = b->S, N = b->N, n = b->n, ss = b->ss, J = b->J;
  const size_t NN = (size_t)N * N, nn = (size_t)n * n;
  if (!std::getenv("LCMI_NOISE_HOST")) {
    // device path (psf_noise.h): 1-D starlet tables of every star in double, then the rank-1 products in fp32
    noise_fn fn = find_noise_kernel(N, ss);
    if (!fn) LC_FAIL(b->ctx, LC_ERR_UNSUPPORTED, "no noise-propagation kernel for this stamp size");
    if (!b->ntab) {
      int rc0 = dmalloc(b, &b->ntab, (size_t)F * S * J * 6 * N);
      if (rc0) return rc0;
    }
    hipLaunchKernelGGL(psf_noise_tables_kernel, dim3(F * S), dim3(((N + 63) / 64) * 64), 4 * N * sizeof(double),
                       b->ctx->stream, S, N, ss, J, b->stars, b->ntab);
    LC_HIP(b->ctx, hipGetLastError());
    const int lds = (int)((nn + (size_t)n * N + 6 * N) * sizeof(float));
    LC_HIP(b->ctx, hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(fn, dim3(F * J), dim3(kNoiseThreads), lds, b->ctx->stream, S, J, b->stars, b->wgt, b->ntab, b->W);
    LC_HIP(b->ctx, hipGetLastError());
    b->have_W = true;
    return LC_OK;
  }
  // host path (double precision FFT convolutions), kept as an independent cross-check: LCMI_NOISE_HOST=1
  std::vector<float> st((size_t)F * S * 4), w((size_t)F * S * nn);
  int rc;
  if ((rc = d2h(b, st.data(), b->stars, st.size() * sizeof(float)))) return rc;
  if ((rc = d2h(b, w.data(), b->wgt, w.size() * sizeof(float)))) return rc;
  std::vector<float> Wall((size_t)F * J * NN);
  // contributor = star of the frame; response of dL/dB to a unit of whitened noise in data pixel p*:
  //   r_i[u'][v'] = a_i * sum_{(u,v) in block(p*)} phi_y(u - u') phi_x(v - v'),  phi = sampled Gaussian of the star
  const int T = noise_threads(F);
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)
    pool.emplace_back([&, t]() {
      std::vector<double> r(NN), py(2 * kRg + 1 + 2), px(2 * kRg + 1 + 2);
      std::vector<float> Wf((size_t)(J + 1) * NN);
      const double c_off = (N % 2 == 0) ? 0.5 : 0.0, is2 = 1.0 / ((double)kSigmaG * kSigmaG),
                   nrm = 1.0 / (std::sqrt(2.0 * M_PI) * kSigmaG);
      const int b0 = ss * (n / 2);
      for (int f = t; f < F; f += T) {
        NoiseAccumulator acc(N, ss);
        for (int s = 0; s < S; ++s) {
          const float *sp = &st[((size_t)f * S + s) * 4];
          const double a = sp[0], dx = ss * (double)sp[1] + c_off, dy = ss * (double)sp[2] + c_off;
          const int ox = (int)std::nearbyint(dx), oy = (int)std::nearbyint(dy);
          std::fill(r.begin(), r.end(), 0.0);
          if (a != 0.0)
            for (int du = 0; du < ss; ++du)
              for (int tu = oy - kRg; tu <= oy + kRg; ++tu) {
                const int up = b0 + du - tu;
                if (up < 0 || up >= N) continue;
                const double gy = nrm * std::exp(-0.5 * (tu - dy) * (tu - dy) * is2);
                for (int dv = 0; dv < ss; ++dv)
                  for (int tv = ox - kRg; tv <= ox + kRg; ++tv) {
                    const int vp = b0 + dv - tv;
                    if (vp < 0 || vp >= N) continue;
                    r[(size_t)up * N + vp] += a * gy * nrm * std::exp(-0.5 * (tv - dx) * (tv - dx) * is2);
                  }
              }
          acc.add(r, &w[((size_t)f * S + s) * nn]);
        }
        acc.finalize(Wf.data());
        std::memcpy(&Wall[(size_t)f * J * NN], Wf.data(), (size_t)J * NN * sizeof(float));
      }
    });
  for (auto &th : pool) th.join();
  if ((rc = h2d(b, b->W, Wall.data(), Wall.size() * sizeof(float)))) return rc;
  b->have_W = true;
  return LC_OK;
}
int lc_psf_batch_get_weights(lc_psf_batch *b, float *W) {
  if (!b || !W) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  return d2h(b, W, b->W, (size_t)b->F * b->J * b->N * b->N * sizeof(float));
}

int lc_psf_batch_eval(lc_psf_batch *b, float *loss, float *chi2, float *grad_moffat, float *grad_stars,
                      float *grad_grid, float *model) {
  if (!b) return LC_ERR_INVALID;
  int rc = ensure_hist(b, b->iters_done + 1);
  if (rc) return rc;
  rc = launch_psf(b, 0, 1, nullptr, true, true);
  if (rc) return rc;
  if (grad_moffat) {
    if (b->moffat_is_q)  // d loss / d (q11, q12, q22, beta) of the quadratic-form Moffat (lc_psf_batch_set_moffat_q)
      hipLaunchKernelGGL(moffat_q_grad_kernel, dim3(b->F), dim3(256), 0, b->ctx->stream, b->N, b->moffat_q, b->o_gT, b->o_gmoffat);
    else
      hipLaunchKernelGGL(moffat_grad_kernel, dim3(b->F), dim3(256), 0, b->ctx->stream, b->N, b->ss, b->moffat, b->o_gT,
                         b->o_gmoffat);
    LC_HIP(b->ctx, hipGetLastError());
  }
  const size_t NN = (size_t)b->N * b->N, nn = (size_t)b->n * b->n;
  if (loss && (rc = d2h(b, loss, b->o_loss, b->F * sizeof(float)))) return rc;
  if (chi2 && (rc = d2h(b, chi2, b->o_chi2, b->F * sizeof(float)))) return rc;
  if (grad_moffat && (rc = d2h(b, grad_moffat, b->o_gmoffat, (size_t)b->F * 4 * sizeof(float)))) return rc;
  if (grad_stars && (rc = d2h(b, grad_stars, b->o_gstars, (size_t)b->F * b->S * 4 * sizeof(float)))) return rc;
  if (grad_grid && (rc = d2h(b, grad_grid, b->o_ggrid, b->F * NN * sizeof(float)))) return rc;
  if (model && (rc = d2h(b, model, b->o_model, b->F * b->S * nn * sizeof(float)))) return rc;
  LC_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));
  return LC_OK;
}

int lc_psf_batch_fit_moffat(lc_psf_batch *b, int n_iter, float *final_loss) {
  if (!b || n_iter < 0) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  const int F = b->F, S = b->S, D = 4 + 3 * S;
  std::vector<float> mof((size_t)F * 4), st((size_t)F * S * 4);
  int rc;
  if ((rc = d2h(b, mof.data(), b->moffat, mof.size() * sizeof(float)))) return rc;
  if ((rc = d2h(b, st.data(), b->stars, st.size() * sizeof(float)))) return rc;
  std::vector<double> x((size_t)F * D), lo((size_t)F * D), hi((size_t)F * D);
  const double inf = std::numeric_limits<double>::infinity();
  for (int f = 0; f < F; ++f) {
    double *xf = &x[(size_t)f * D], *l = &lo[(size_t)f * D], *h = &hi[(size_t)f * D];
    for (int k = 0; k < 4; ++k) xf[k] = mof[f * 4 + k];
    l[0] = l[1] = 0.5 / b->ss;
    h[0] = h[1] = b->n / 2.0;
    l[2] = -M_PI;
    h[2] = M_PI;
    l[3] = 1.1;
    h[3] = 50.0;
    for (int s = 0; s < S; ++s) {
      xf[4 + s] = st[((size_t)f * S + s) * 4 + 0];
      xf[4 + S + s] = st[((size_t)f * S + s) * 4 + 1];
      xf[4 + 2 * S + s] = st[((size_t)f * S + s) * 4 + 2];
      l[4 + s] = 0.0;
      h[4 + s] = inf;
      l[4 + S + s] = l[4 + 2 * S + s] = -b->n / 4.0;
      h[4 + S + s] = h[4 + 2 * S + s] = b->n / 4.0;
    }
  }
  const bool had_W = b->have_W;
  hipStream_t q = b->ctx->stream;
  if (D <= 64 && !b->moffat_is_q && !std::getenv("LCMI_LBFGS_HOST")) {
    // the optimiser on the device (csrc/psf_lbfgs.h): per round one evaluation of all frames at the points the step
    // kernel left in the parameter blocks, then the step kernel; the host reads one integer every 8 rounds
    const size_t FD = (size_t)F * D;
    const int max_rounds = std::max(1, n_iter) * 4 + 32;  // accepted steps + back-tracking trials; stops earlier when all frames are done
    double *dbl = nullptr;
    int *ints = nullptr;
    const size_t n_dbl = 6 * FD + 2 * FD * kPsfLbfgsMem + (size_t)F * kPsfLbfgsMem + 2 * (size_t)F;
    const size_t n_int = 4 * (size_t)F + (size_t)max_rounds + 1;
    LC_HIP(b->ctx, hipMalloc((void **)&dbl, n_dbl * sizeof(double)));
    struct DevGuard {
      void *p;
      ~DevGuard() { (void)hipFree(p); }
    } g1{dbl};
    LC_HIP(b->ctx, hipMalloc((void **)&ints, n_int * sizeof(int)));
    DevGuard g2{ints};
    LC_HIP(b->ctx, hipMemsetAsync(dbl, 0, n_dbl * sizeof(double), q));
    LC_HIP(b->ctx, hipMemsetAsync(ints, 0, n_int * sizeof(int), q));
    PsfLbfgsDev P;
    P.F = F;
    P.S = S;
    P.D = D;
    P.maxiter = n_iter;
    P.gtol = 1e-7;
    P.ftol = 1e-12;
    P.x = dbl;
    P.g = P.x + FD;
    P.d = P.g + FD;
    P.xt = P.d + FD;
    P.lo = P.xt + FD;
    P.hi = P.lo + FD;
    P.Sm = P.hi + FD;
    P.Ym = P.Sm + FD * kPsfLbfgsMem;
    P.rho = P.Ym + FD * kPsfLbfgsMem;
    P.f = P.rho + (size_t)F * kPsfLbfgsMem;
    P.alpha = P.f + F;
    P.state = ints;
    P.ls = P.state + F;
    P.iters = P.ls + F;
    P.m = P.iters + F;
    P.n_active = P.m + F;
    P.moffat = b->moffat;
    P.stars = b->stars;
    P.o_loss = b->o_loss;
    P.o_gmoffat = b->o_gmoffat;
    P.o_gstars = b->o_gstars;
    for (size_t i = 0; i < x.size(); ++i) x[i] = std::min(std::max(x[i], lo[i]), hi[i]);
    for (auto &v : hi) v = std::min(v, 1e300);  // (infinite bounds travel as huge finite ones)
    LC_HIP(b->ctx, hipMemcpyAsync(P.x, x.data(), FD * sizeof(double), hipMemcpyHostToDevice, q));
    LC_HIP(b->ctx, hipMemcpyAsync(P.lo, lo.data(), FD * sizeof(double), hipMemcpyHostToDevice, q));
    LC_HIP(b->ctx, hipMemcpyAsync(P.hi, hi.data(), FD * sizeof(double), hipMemcpyHostToDevice, q));
    auto evaluate = [&]() -> int {
      hipLaunchKernelGGL(moffat_raster_kernel, dim3(F), dim3(256), 0, q, b->N, b->ss, b->moffat, b->Tm);
      int r = ensure_hist(b, b->iters_done + 1);
      if (r) return r;
      if ((r = launch_psf(b, 0, 1, nullptr, true, false))) return r;  // stage A: no regularisation (B fixed)
      hipLaunchKernelGGL(moffat_grad_kernel, dim3(F), dim3(256), 0, q, b->N, b->ss, b->moffat, b->o_gT, b->o_gmoffat);
      LC_HIP(b->ctx, hipGetLastError());
      return LC_OK;
    };
    hipLaunchKernelGGL(psf_lbfgs_step_kernel, dim3(F), dim3(64), 0, q, P, 2, 0);  // clipped start -> parameter blocks
    if ((rc = evaluate())) return rc;
    hipLaunchKernelGGL(psf_lbfgs_step_kernel, dim3(F), dim3(64), 0, q, P, 0, 0);
    for (int round = 1; round <= max_rounds; ++round) {
      if ((rc = evaluate())) return rc;
      hipLaunchKernelGGL(psf_lbfgs_step_kernel, dim3(F), dim3(64), 0, q, P, 1, round);
      LC_HIP(b->ctx, hipGetLastError());
      if (round % 8 == 0 || round == max_rounds) {
        int active = 0;
        if ((rc = d2h(b, &active, P.n_active + round, sizeof(int)))) return rc;
        if (active == 0) break;
      }
    }
    hipLaunchKernelGGL(psf_lbfgs_step_kernel, dim3(F), dim3(64), 0, q, P, 2, 0);  // leave the device at the accepted optimum
    if ((rc = evaluate())) return rc;
    b->have_W = had_W;
    if (final_loss && (rc = d2h(b, final_loss, b->o_loss, (size_t)F * sizeof(float)))) return rc;
    LC_HIP(b->ctx, hipStreamSynchronize(q));
    return LC_OK;
  }
  // host form (LCMI_LBFGS_HOST=1, the quadratic-form Moffat of the distortion fit): the same state machine in
  // csrc/lbfgs_host.h, pinned staging: parameters up, loss + gradients down, one synchronisation per evaluation
  const size_t n_in = (size_t)F * 4 + (size_t)F * S * 4, n_out = (size_t)F + (size_t)F * 4 + (size_t)F * S * 4;
  float *pin = nullptr;
  LC_HIP(b->ctx, hipHostMalloc((void **)&pin, (n_in + n_out) * sizeof(float), hipHostMallocDefault));
  struct PinGuard {
    float *p;
    ~PinGuard() { (void)hipHostFree(p); }
  } guard{pin};
  float *pmof = pin, *pst = pin + (size_t)F * 4, *loss = pin + n_in, *gm = loss + F, *gs = gm + (size_t)F * 4;
  std::memcpy(pst, st.data(), st.size() * sizeof(float));  // sky (column 3) stays as it is
  auto eval = [&](const std::vector<double> &X, std::vector<double> &Fv, std::vector<double> &G) -> int {
    for (int f = 0; f < F; ++f) {
      const double *xf = &X[(size_t)f * D];
      for (int k = 0; k < 4; ++k) pmof[f * 4 + k] = (float)xf[k];
      for (int s = 0; s < S; ++s) {
        pst[((size_t)f * S + s) * 4 + 0] = (float)xf[4 + s];
        pst[((size_t)f * S + s) * 4 + 1] = (float)xf[4 + S + s];
        pst[((size_t)f * S + s) * 4 + 2] = (float)xf[4 + 2 * S + s];
      }
    }
    int r;
    LC_HIP(b->ctx, hipMemcpyAsync(b->moffat, pmof, (size_t)F * 4 * sizeof(float), hipMemcpyHostToDevice, q));
    LC_HIP(b->ctx, hipMemcpyAsync(b->stars, pst, (size_t)F * S * 4 * sizeof(float), hipMemcpyHostToDevice, q));
    hipLaunchKernelGGL(moffat_raster_kernel, dim3(F), dim3(256), 0, q, b->N, b->ss, b->moffat, b->Tm);
    if ((r = ensure_hist(b, b->iters_done + 1))) return r;
    if ((r = launch_psf(b, 0, 1, nullptr, true, false))) return r;  // stage A: no regularisation (B fixed)
    hipLaunchKernelGGL(moffat_grad_kernel, dim3(F), dim3(256), 0, q, b->N, b->ss, b->moffat, b->o_gT, b->o_gmoffat);
    LC_HIP(b->ctx, hipGetLastError());
    LC_HIP(b->ctx, hipMemcpyAsync(loss, b->o_loss, (size_t)F * sizeof(float), hipMemcpyDeviceToHost, q));
    LC_HIP(b->ctx, hipMemcpyAsync(gm, b->o_gmoffat, (size_t)F * 4 * sizeof(float), hipMemcpyDeviceToHost, q));
    LC_HIP(b->ctx, hipMemcpyAsync(gs, b->o_gstars, (size_t)F * S * 4 * sizeof(float), hipMemcpyDeviceToHost, q));
    LC_HIP(b->ctx, hipStreamSynchronize(q));
    for (int f = 0; f < F; ++f) {
      Fv[f] = loss[f];
      double *g = &G[(size_t)f * D];
      for (int k = 0; k < 4; ++k) g[k] = gm[f * 4 + k];
      for (int s = 0; s < S; ++s) {
        g[4 + s] = gs[((size_t)f * S + s) * 4 + 0];
        g[4 + S + s] = gs[((size_t)f * S + s) * 4 + 1];
        g[4 + 2 * S + s] = gs[((size_t)f * S + s) * 4 + 2];
      }
    }
    return 0;
  };
  LbfgsResult res;
  rc = batched_lbfgs(F, D, x, lo, hi, n_iter, eval, res);
  if (rc) return rc;
  // leave the device at the accepted optimum
  std::vector<double> Fv(F), G((size_t)F * D);
  if ((rc = eval(x, Fv, G))) return rc;
  b->have_W = had_W;
  if (final_loss)
    for (int f = 0; f < F; ++f) final_loss[f] = (float)Fv[f];
  return LC_OK;
}

int lc_psf_batch_run_adabelief(lc_psf_batch *b, int n_iter, const lc_adabelief_cfg *cfg) {
  if (!b || n_iter <= 0) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  int rc = ensure_hist(b, b->iters_done + n_iter + 1);
  if (rc) return rc;
  rc = launch_psf(b, 1, n_iter, cfg, false, true);
  if (rc) return rc;
  b->iters_done += n_iter;
  return LC_OK;
}
int lc_psf_batch_iterations_done(lc_psf_batch *b) { return b ? b->iters_done : LC_ERR_INVALID; }

int lc_psf_batch_get_loss_history(lc_psf_batch *b, float *history, int stride) {
  if (!b || !history || stride < b->iters_done + 1) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  int rc = launch_psf(b, 0, 1, nullptr, false, true);  // loss of the final parameters -> hist[T]
  if (rc) return rc;
  LC_HIP(b->ctx, hipMemcpy2DAsync(history, stride * sizeof(float), b->hist, b->hist_stride * sizeof(float),
                                   (b->iters_done + 1) * sizeof(float), b->F, hipMemcpyDeviceToHost, b->ctx->stream));
  LC_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));
  return LC_OK;
}

int lc_psf_batch_split_fallbacks(lc_psf_batch *b, int *count) {
  if (!b || !count) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  return read_split_fallbacks(b, count);
}

int lc_psf_batch_get_results(lc_psf_batch *b, float *narrow_psf, float *full_psf, float *residuals, float *chi2) {
  if (!b) return LC_ERR_INVALID;
  LC_ENTER(b->ctx);
  int rc = ensure_hist(b, b->iters_done + 1);
  if (rc) return rc;
  rc = launch_psf(b, 0, 1, nullptr, true, true);
  if (rc) return rc;
  hipLaunchKernelGGL(psf_finalize_kernel, dim3(b->F), dim3(256), 0, b->ctx->stream, b->N, b->Tm, b->B, b->narrow,
                     b->full);
  hipLaunchKernelGGL(psf_residual_kernel, dim3(b->F), dim3(256), 0, b->ctx->stream, b->S, b->n, b->data, b->wgt,
                     b->o_model, b->resid, b->redchi2);
  LC_HIP(b->ctx, hipGetLastError());
  const size_t NN = (size_t)b->N * b->N, nn = (size_t)b->n * b->n;
  if (narrow_psf && (rc = d2h(b, narrow_psf, b->narrow, b->F * NN * sizeof(float)))) return rc;
  if (full_psf && (rc = d2h(b, full_psf, b->full, b->F * NN * sizeof(float)))) return rc;
  if (residuals && (rc = d2h(b, residuals, b->resid, b->F * b->S * nn * sizeof(float)))) return rc;
  if (chi2 && (rc = d2h(b, chi2, b->redchi2, b->F * sizeof(float)))) return rc;
  LC_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));
  return LC_OK;
}

#ifdef LC_STAMPS
int lc_debug_get_stamps(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc::g_stamps), 128 * sizeof(long long)) == hipSuccess ? 0 : -2;
}
#endif

}  // extern "C"
