// Joint multi-epoch forward-model object behind the C ABI (include/lcmi.h, "joint" section).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <complex>
#include <thread>
#include <cstdio>
#include <cstring>

#include "joint_kernels.h"
#include "joint_gm.h"
#include "joint_reduce_peer.h"
#include "joint_reg_mfma.h"
#include "joint_reg_fused.h"
#include "joint_reg_rows.h"
#include "joint_ps.h"
#include "joint_noise.h"
#include "joint_lbfgs.h"
#include "noise_host.h"
#include "starlet_norms.h"

using namespace lc;

constexpr int kMaxParts = 16;  // workgroups per epoch of the phased launches
typedef void (*epoch_fn)(JointArgs);
typedef void (*update_fn)(JointUpdArgs);
struct JointVariant {
  int n, ss, L;
  epoch_fn ek;
  int e_lds, e_thr;
  update_fn uk;  // null: regulariser + update run as global-memory kernels (joint_gm.h)
  int u_thr, u_lds;
  bool gspec;
  epoch_fn ek_aux;  // plain convolution / spectrum modes of the same pipeline (noise propagation)
  epoch_fn ek_tile = nullptr;  // global-spectrum kernels: the variant whose column passes go through an LDS tile
  int e_lds_tile = 0;
  // global-spectrum kernels: one phase (A, B, C, B', C', D) per launch on a grid (E, parts); [1] / [3] also in the tile build
  epoch_fn ek_phase[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  epoch_fn ek_phase_tile[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int e_lds_lite = 0, e_lds_lite_tile = 0;  // LDS of the column phases and of phase D (JointCfg::LDS_LITE)
  int lpf = 16;                             // lanes per transform (JointCfg::LPF)
  // cluster form (joint_kernels.h, PHASE = 7): several workgroups per epoch in ONE launch, spectrum in the global scratch;
  // a build of its own beside an LDS-spectrum kernel (same N, SS, L: same spectra)
  epoch_fn ek_cluster = nullptr;
  int cl_lds = 0, cl_thr = 0, cl_lpf = 16;
};

typedef void (*mreg_fn)(MregArgs);
typedef void (*mreg_mm_fn)(MmBatch);
typedef void (*mreg_chain_fn)(MregChainArgs);
struct MregKernels {
  int N;
  mreg_fn fwd, adj;
  int lds_fwd, lds_adj, nthr;
  mreg_mm_fn mm;  // batched tiled products of the second form of the chain (joint_reg_mfma.h)
  mreg_chain_fn chain;  // the whole second form as one launch (null: not built for this N)
  void (*rows)(MregRowsArgs) = nullptr;  // third form: row blocks, two stages (joint_reg_rows.h; null: not built for this N)
  int rows_lds = 0;
  void (*mmx[4])(MmxArgs) = {nullptr, nullptr, nullptr, nullptr};  // fourth form (default): f1', a1', a2' and the plain product of joint_reg_fused.h
};

struct lc_joint {
  lc_ctx *ctx = nullptr;
  const JointVariant *v = nullptr;
  int E = 0, M = 0, n = 0, ss = 0, N = 0, L = 0, J = 0, KH = 0;
  float *data = nullptr, *wgt = nullptr;
  float2 *St = nullptr, *twid = nullptr;
  float *par[LC_P_COUNT] = {}, *pm[LC_P_COUNT] = {}, *ps[LC_P_COUNT] = {}, *gout[LC_P_COUNT] = {};
  int psize[LC_P_COUNT] = {};
  float *tabs = nullptr, *HG = nullptr;
  float *part = nullptr;  // [E][kMaxParts][4 + 3 kMaxSources] partial sums of phased launches
  float *tshift = nullptr;  // [E][2] shifts used by the last gradient evaluation (JointArgs::tshift)
  float *chi2_e = nullptr, *g_a = nullptr, *g_cx_e = nullptr, *g_cy_e = nullptr, *g_dx = nullptr, *g_dy = nullptr,
        *g_mean = nullptr;
  float *model = nullptr, *fisher = nullptr, *shared = nullptr, *W = nullptr, *norms = nullptr,
        *qscr = nullptr, *out_loss = nullptr, *hist = nullptr, *scene2 = nullptr;
  float *prior = nullptr;  // [4][M]
  float *a_ref = nullptr;  // [kMaxSources] reference fluxes the flux moments of the shared block are centred on
  int shared_count = 0, hist_cap = 0, iters_done = 0, n_prior = 0;
  int free_mask[LC_P_COUNT] = {};
  bool have_W = false, h_nonzero = false;
  lc_joint_loss_cfg cfg{};
  float *greg = nullptr, *regs = nullptr;
  float2 *spec = nullptr;             // [E][N][KS] spectrum scratch of the large-grid epoch kernel
  float *nz_a = nullptr, *nz_b = nullptr, *nz_c = nullptr, *nz_up = nullptr, *nz_scene = nullptr;  // noise propagation [E][N*N]
  float2 *St_alt = nullptr;                                                                        // [E][KH][L]
  float *psf_dev = nullptr, *psF = nullptr;  // point-source-only path: narrow PSFs [E][N*N], filter outputs [E][M][3][n*n]
  float *gm_c = nullptr, *gm_t = nullptr, *gm_n = nullptr, *gm_y = nullptr, *gm_l1 = nullptr, *gm_pos = nullptr;
  float *gm_edge = nullptr;  // [N][4] end sums of the adjoint passes
  float *gm_pts = nullptr;  // [8] mean fluxes + [blocks][8][3] partial inner products of the point-source term
  // matrix-core form of the regulariser (joint_reg_mfma.h), N >= 128
  const struct MregKernels *mreg = nullptr;
  float *mr_A = nullptr, *mr_AT = nullptr, *mr_C = nullptr, *mr_Z = nullptr, *mr_l1 = nullptr, *mr_pos = nullptr,
        *mr_part = nullptr, *mr_pbar = nullptr;
  float *mr_l1b = nullptr, *mr_posb = nullptr, *mr_S = nullptr, *mr_T = nullptr;  // second form of the chain: per-block values, S planes, product scratch
  float *rr_Zp = nullptr;  // third form (row blocks): partial sub-gradient planes [N / 16 * kRrMaxParts][N^2]
  hipStream_t streamB = nullptr;      // the h regulariser runs here, concurrently with the epoch kernel
  hipStream_t streamC = nullptr;      // ... and the point-source starlet term here, beside the regulariser chain (created on first use)
  hipEvent_t evPts = nullptr;
  hipEvent_t evReg = nullptr, evUpd = nullptr;
  const void *ps_attr_fn = nullptr;   // point-source kernel whose dynamic-LDS attribute has been set
  std::vector<hipStream_t> gstreams;  // batched star photometry: the parts of the batch beyond the first run on these
  std::vector<hipEvent_t> gevents;
  bool reg_pending = false;
  hipEvent_t *tl_events = nullptr;  // LCMI_TIMELINE diagnostic (lc_joint_run_adabelief)
  bool in_device_loop = false, fuse_pending = false;
  unsigned int *reg_flag = nullptr;  // [0] sequence number of the last finished regulariser chain, [1] a flag wait ran out
  unsigned int reg_seq = 0;
  bool flag_sync = false;            // this iteration's update checks reg_flag itself instead of waiting for evReg
  // four-launch chain (joint_reg_fused.h): planes_pred = the consumer of this iteration's chain will be the fused reduction +
  // update, which adds the planes itself (set by lc_joint_step_local before the chain is enqueued); reg_planes = this
  // iteration's chain left planes and per-tile values only (no greg / regs yet); reg_noflag = it wrote greg / regs but raises
  // no completion flag (its consumer waits for the event)
  unsigned int *pts_ctr = nullptr;   // [0] arrival counter of the point-source blocks inside the update launch, [1] a wait ran out
  unsigned int pts_seq = 0;
  bool pts_tail_used = false;
  // "the update is complete" word ([0]: sequence number stored by the first thread of the following epoch launch; [1]: a gate
  // wait ran out) and its host-side value; upd_gate_pending: no event was recorded for the last update (the next chain starts
  // behind a gate kernel); upd_signal_due: a gate is waiting for the next epoch launch to raise the word
  unsigned int *upd_ctr = nullptr;
  unsigned int upd_seq = 0;
  bool upd_gate_pending = false, upd_gate_used = false, upd_signal_due = false;
  bool evreg_recorded = true;   // evReg stands behind the last chain enqueued on the second stream
  // force_events: an in-kernel wait of this object ran out once (a stream held up for longer than its bound: ~1 s); the run was
  // redone from a copy of the state and the object synchronises its streams with events only from then on
  bool force_events = false;
  int wait_fallbacks = 0;
  int streams_overlap = -1;   // -1: not probed yet; 1: the chain's stream runs beside the main stream; 0: they share a hardware queue
  // epoch_wait_due: this iteration's chain ends in a counting launch and the epoch launch may carry the wait for it (an extra
  // block: JointArgs::chain_flag); epoch_waited: it did - the update behind it needs no synchronisation of its own
  bool epoch_wait_due = false, epoch_waited = false, epoch_wait_used = false;
  bool planes_pred = false, reg_planes = false, reg_noflag = false;
  // reg_counter: this iteration's chain ends in a launch that counts its blocks into reg_flag; defer_event: lc_joint_step_update
  // left the cross-stream wait for the chain to launch_update, which drops it when the consumer checks the counter itself
  bool reg_counter = false, defer_event = false, gm_flag_used = false;
  RegPlanes planes;
  bool fuse_full = false;  // lc_joint_run_adabelief, background free: reduction over the epochs and update in one launch
  bool fuse_stencil = false;  // ... and the T_e^T step too: no phase D, no slabs (global-spectrum kernels, translated epochs)
  bool any_rotation = false;  // some alpha != 0
  bool pts_pending = false;  // the point-source starlet term of this iteration was evaluated by the stream-B launch  // lc_joint_run_adabelief: scalar reduction fused into the update
  // batched star photometry (lc_joint_create_groups): G stars, star g owns the epochs [gstart[g], gstart[g + 1]) and its own
  // shared positions c_x, c_y [g * M ..]; the update runs one block per star (joint_update_groups_kernel)
  int G = 0;
  std::vector<int> gstart;
  int *group_dev = nullptr;
  JointUpdArgs *views_dev = nullptr;
  float *ghist = nullptr, *shared_g = nullptr, *a_ref_g = nullptr, *out_loss_g = nullptr;
  int ghist_cap = 0;
  // cluster launches of the epoch kernel (few epochs per GPU): flag words [E][kClStride], then the abort word
  unsigned int *cl_ctr = nullptr;
  unsigned int cl_base = 0;
  unsigned int *chain_flags = nullptr;  // one-launch regulariser chain: [kChainBlocks] sync words, then the abort word
  unsigned int chain_base = 0;
  bool chain_off = false, chain_used = false;
  bool cluster_off = false, in_sharded_loop = false;
  // sharded loop over the library's own peer group: the reduction over the epochs publishes straight into the exchange region
  // (joint_reduce_peer_kernel: one launch for reduction + all-reduce); peer_fused_done: this iteration's launch did both
  lc_peer_group *peer_fuse = nullptr;
  bool peer_fused_done = false;
  int cl_parts_last = 0, cl_fallbacks = 0;
  // return_param_history: device-resident [phist_cap][phist_P] rows of the free blocks, one per AdaBelief update
  float *phist = nullptr;
  int phist_cap = 0, phist_P = 0, phist_rows = 0, phist_off[LC_P_COUNT] = {};
  std::vector<float> h_sigma2, h_psf;  // host copies for the one-time noise propagation
  std::vector<void *> allocs;
};

namespace {

template <class T>
int dmalloc(lc_joint *j, T **p, size_t count) {
  LC_HIP(j->ctx, hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
  j->allocs.push_back(*p);
  LC_HIP(j->ctx, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), j->ctx->stream));
  return LC_OK;
}
int h2d(lc_joint *j, void *dst, const void *src, size_t bytes) {
  LC_HIP(j->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, j->ctx->stream));
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  return LC_OK;
}
int d2h(lc_joint *j, void *dst, const void *src, size_t bytes) {
  LC_HIP(j->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, j->ctx->stream));
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  return LC_OK;
}

template <int N, int SS, int L, int PX, int NW, int LPF = 16>
JointVariant make_jv() {
  typedef JointCfg<N, SS, L, NW, false, LPF> C;
  return JointVariant{C::n, SS, L, joint_epoch_kernel<C>, C::LDS_BYTES, C::NTHR, joint_update_kernel<N, PX>, N * N / PX,
                      (int)(StarletLds<N>::FLOATS * sizeof(float)), false, joint_epoch_kernel<C, true>};
}
// ... with the cluster form beside it: four-wave workgroups (one wave per SIMD), spectrum in the global scratch
template <int N, int SS, int L, int PX, int NW, int LPF, int CNW, int CLPF>
JointVariant make_jv_cl() {
  JointVariant v = make_jv<N, SS, L, PX, NW, LPF>();
  typedef JointCfg<N, SS, L, CNW, true, CLPF> CC;
  v.ek_cluster = joint_epoch_kernel<CC, false, 7>;
  v.cl_lds = CC::LDS_BYTES;
  v.cl_thr = CC::NTHR;
  v.cl_lpf = CLPF;
  return v;
}
// stamp sizes beside the tuned ones (any multiple of 8 up to 64 at ss = 2): the same epoch kernel at that N, regulariser and
// update as the run-time-N multi-block kernels (joint_gm.h) instead of a single-workgroup kernel built per size
template <int N, int SS, int L, int NW, int LPF = 16>
JointVariant make_jv_plain() {
  typedef JointCfg<N, SS, L, NW, false, LPF> C;
  return JointVariant{C::n, SS, L, joint_epoch_kernel<C>, C::LDS_BYTES, C::NTHR, nullptr, 0, 0, false, joint_epoch_kernel<C, true>};
}
// large grids: spectrum scratch in HBM, starlet / update as multi-block kernels
template <int N, int SS, int L, int NW, int LPF = 16>
JointVariant make_jv_gm() {
  typedef JointCfg<N, SS, L, NW, true, LPF> C;
  typedef JointCfg<N, SS, L, NW, true, LPF, true> CT;
  JointVariant v{C::n, SS, L, joint_epoch_kernel<C>, C::LDS_BYTES, C::NTHR, nullptr, 0, 0, true, joint_epoch_kernel<C, true>,
                 joint_epoch_kernel<CT>, CT::LDS_BYTES};
  v.lpf = LPF;
  v.ek_phase[0] = joint_epoch_kernel<C, false, 1>;
  v.ek_phase[1] = joint_epoch_kernel<C, false, 2>;
  v.ek_phase[2] = joint_epoch_kernel<C, false, 3>;
  v.ek_phase[3] = joint_epoch_kernel<C, false, 4>;
  v.ek_phase[4] = joint_epoch_kernel<C, false, 5>;
  v.ek_phase[5] = joint_epoch_kernel<C, false, 6>;
  v.ek_phase_tile[1] = joint_epoch_kernel<CT, false, 2>;
  v.ek_phase_tile[3] = joint_epoch_kernel<CT, false, 4>;
  v.e_lds_lite = C::LDS_LITE;
  v.e_lds_lite_tile = CT::LDS_LITE;
  return v;
}
int g_debug_global = 0;  // lc_joint_set_debug_global: small stamps through the large-grid kernels (parity tests)
// E, n_cu: epochs of the fit and compute units of the device (0: the default kernel of the stamp size)
const JointVariant *find_jv(int n, int ss, int E = 0, int n_cu = 0) {
  if (g_debug_global) {
    static const JointVariant dbg[] = {
        make_jv_gm<32, 2, 48, 4>(),
        make_jv_gm<64, 2, 96, 8>(),
    };
    for (const auto &v : dbg)
      if (v.n == n && v.ss == ss) return &v;
  }
  // n = 64 (C4) with the epoch spread over several workgroups, one launch per phase, spectrum in global memory - the form
  // the n = 128 kernels take below 129 epochs.  Built, tested (tests/test_joint_gpu.py, test_joint_paths_gpu.py) and NOT the
  // default: measured on MI355X at the epoch counts a sharded C4 leaves per GPU it loses to one workgroup per epoch at every
  // count (25 epochs: 72 against 68 us per iteration, 50: 85 / 69, 100: 107 / 71; with two-wave workgroups 80 / 99 / 142).
  // rocprofv3: each of the six phase launches takes 8 - 11 us for 1 - 2 us of transforms (parameters -> tables -> rows ->
  // transform -> store is a chain of dependent memory round trips per launch), 58 us per iteration against 55 us for the
  // whole epoch in one workgroup.  LCMI_N128_SPLIT=1 selects it.
  if (n == 64 && ss == 2 && E > 0) {
    const char *sp = std::getenv("LCMI_N128_SPLIT");
    if (sp && std::atoi(sp) != 0) {
      static const JointVariant gm4 = make_jv_gm<128, 2, 192, 4, 16>();
      return &gm4;
    }
  }
  static const JointVariant table[] = {
      // FFT length L = 16 * N2, the smallest of 2^m and 3 * 2^m that keeps the 'same' window alias free (>= 3N/2)
      make_jv<16, 1, 32, 4, 8>(),     // n = 16, ss = 1 (reference test fixture)
      make_jv<32, 2, 48, 4, 16>(),    // n = 16, ss = 2
      make_jv<48, 2, 96, 4, 8>(),     // n = 24 (default stamp_size_stars)
      make_jv<64, 2, 96, 8, 16>(),    // n = 32 (default stamp_size_ROI)
      make_jv_plain<80, 2, 128, 8>(),    // n = 40
      make_jv_plain<96, 2, 192, 8>(),    // n = 48
      make_jv_plain<112, 2, 192, 8>(),   // n = 56
      // n = 64 (C4); 16 waves with 32-lane transforms (4 waves per SIMD, 128 registers, 34 spilled) measured 2 % slower
      make_jv_cl<128, 2, 192, 16, 8, 16, 4, 16>(),
      // n = 128 (C5): transforms over 32 lanes (12 registers per lane like the n = 64 kernel; at 16 lanes the 24-register
      // transforms spilled 200+ registers, and 4 waves with 436 registers each measured 1.23 x slower than that)
      // (the cluster form at this size - two to eight 8-wave workgroups per epoch in one launch - was built in round 4 and measured
      //  slower than the phased launches: 125 epochs x 2: 292.9 against 228.7 us per iteration, 64 x 4: 208.4 / 153.3, 32 x 8:
      //  168.8 / 116.4; 256 registers with 39 spilled, and every hand-off goes to the memory side; not instantiated)
      make_jv_gm<256, 2, 384, 8, 32>(),
  };
  for (const auto &v : table)
    if (v.n == n && v.ss == ss) return &v;
  return nullptr;
}

__global__ void joint_scene_kernel(int N, int ss, int M, int e, const float *a, const float *cx, const float *cy,
                                   const float *dx, const float *dy, const float *alpha, const float *h, float *scene,
                                   float *background) {
  const float c0 = (N - 1) * 0.5f;
  const float al = alpha[e] * 0.017453292519943295f;
  const float ca = cosf(al), sa = sinf(al);
  const float sdx = ss * dx[e], sdy = ss * dy[e];
  const float inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N * N; k += gridDim.x * blockDim.x) {
    const int u = k / N, v = k % N;
    float Xs, Ys;
    sample_coords(u, v, c0, ca, sa, sdx, sdy, Xs, Ys);
    const float x0f = floorf(Xs), y0f = floorf(Ys);
    const float fx = Xs - x0f, fy = Ys - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int xa = min(max(x0, 0), N - 1), xb = min(max(x0 + 1, 0), N - 1);
    const int ya = min(max(y0, 0), N - 1), yb = min(max(y0 + 1, 0), N - 1);
    const float top = h[ya * N + xa] + fx * (h[ya * N + xb] - h[ya * N + xa]);
    const float bot = h[yb * N + xa] + fx * (h[yb * N + xb] - h[yb * N + xa]);
    const float bg = top + fy * (bot - top);
    float ps = 0.f;
    for (int i = 0; i < M; ++i) {
      const float X = c0 + ss * (ca * cx[i] - sa * cy[i] + dx[e]);
      const float Y = c0 + ss * (sa * cx[i] + ca * cy[i] + dy[e]);
      const float tx = v - X, ty = u - Y;
      ps += a[e * M + i] * nrm2 * expf(-0.5f * (tx * tx + ty * ty) * inv_s2);
    }
    scene[k] = ps + bg;
    background[k] = bg;
  }
}

int ensure_hist(lc_joint *j, int needed) {
  if (needed <= j->hist_cap) return LC_OK;
  const int nc = std::max(needed, 2 * j->hist_cap + 64);
  float *nh = nullptr;
  LC_HIP(j->ctx, hipMalloc((void **)&nh, (size_t)nc * sizeof(float)));
  LC_HIP(j->ctx, hipMemsetAsync(nh, 0, (size_t)nc * sizeof(float), j->ctx->stream));
  if (j->hist) {
    LC_HIP(j->ctx, hipMemcpyAsync(nh, j->hist, (size_t)j->hist_cap * sizeof(float), hipMemcpyDeviceToDevice, j->ctx->stream));
    LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
    hipFree(j->hist);
  }
  j->hist = nh;
  j->hist_cap = nc;
  return LC_OK;
}

bool reg_h_on(const lc_joint *j) {
  const bool lam = (j->cfg.lam_scales != 0.f || j->cfg.lam_hf != 0.f || j->cfg.lam_positivity != 0.f);
  return lam && (j->free_mask[LC_P_H] || j->h_nonzero);
}

typedef void (*ps_fn)(JointPsArgs);
void find_ps_kernel(int N, int ss, ps_fn *fn, int *lds, bool persist = false) {
  *fn = nullptr;
#define LC_PS(NN_, SS_)                                                                   \
  if (N == NN_ && ss == SS_) {                                                            \
    *fn = persist ? joint_ps_kernel<NN_, SS_, true> : joint_ps_kernel<NN_, SS_, false>;   \
    *lds = joint_ps_lds_bytes<NN_, SS_>();                                                \
  }
  LC_PS(16, 1)
  LC_PS(32, 2)
  LC_PS(48, 2)
  LC_PS(64, 2)
  LC_PS(80, 2)
  LC_PS(96, 2)
  LC_PS(112, 2)
  LC_PS(128, 2)
#undef LC_PS
}

// Workgroups per epoch of the cluster form for the next forward + backward launch of this object (0: the one-workgroup
// kernel).  Inside the library's own loops only, and all workgroups of the launch resident together, one per CU.  P
// four-wave workgroups per epoch: six turn every phase into one sweep (24 waves x 4 transforms = the 96 column transforms).
// Measured on MI355X at 64 x 64 (profiles/r04_cluster_*): the epoch kernel itself takes 42 us as a cluster of six against 53.5
// us with one workgroup per epoch.  With the eight-launch regulariser chain (~60 us on the second stream) that gained nothing:
// the chain was the iteration (66.8 against 65.0 us at 25 epochs).  With the four-launch chain (joint_reg_fused.h, ~32 us) it
// does: 8 / 16 / 25 / 32 epochs 55.3 / 55.9 / 56.2 / 56.7 us per iteration against 64.8 / 65.1 / 65.0 / 65.3.
// DEFAULT ("auto", also when LCMI_CLUSTER is unset) since then: six workgroups per epoch whenever they leave kChainBlocks CUs
// to the chain (6 E + 64 <= CUs: up to 32 epochs - a rank's share of a sharded C4).  Not beyond: clusters that fill the
// machine starve the chain (42 epochs x 6: 87.5 us, 50 x 5: 101.9, 85 x 3: 122.5), and fewer workgroups per epoch gain
// little or lose (36 x 5: 64.1, 40 x 4: 64.0, 56 x 3: 72.8, 80 x 2: 85.9 against 65 - 67 us; again after the event behind the
// update had gone: 34 x 5: 60.1, 40 x 4: 60.0, 48 x 4: 60.3, 56 x 3: 70.9 against 60.7 - 61.7).  LCMI_CLUSTER=0 switches the
// form off, LCMI_CLUSTER=<P> forces P workgroups per epoch (as many as fit one per CU).
int cluster_parts(const lc_joint *j) {
  const JointVariant *v = j->v;
  if (!(v->ek_cluster && j->cl_ctr && !j->cluster_off && (j->in_device_loop || j->in_sharded_loop) && !j->fuse_stencil)) return 0;
  const char *cl = std::getenv("LCMI_CLUSTER");
  const int per_wg = v->cl_thr / v->cl_lpf, full = (j->L / 2 + per_wg - 1) / per_wg;
  if (!cl || std::strcmp(cl, "auto") == 0) {
    if (full > kMaxParts || full < 2 || j->E * full + kChainBlocks > j->ctx->n_cu) return 0;
    // (the chain must be the short one for the form to pay: the background regulariser on the matrix cores)
    return (j->mreg || !reg_h_on(j)) ? full : 0;
  }
  const int want = std::max(0, std::atoi(cl));
  const int P = std::min({kMaxParts, want, j->ctx->n_cu / std::max(j->E, 1)});
  return P >= 2 ? P : 0;
}

// (e0, e1, stream: the point-source-only kernel over the epochs [e0, e1) on `stream` - batched star photometry; default: all epochs)
int launch_epochs(lc_joint *j, int mode, int isrc, bool want_hgrad, float *model_out, int e0 = 0, int e1 = -1,
                  hipStream_t ps_stream = nullptr) {
  const JointVariant *v = j->v;
  JointArgs A;
  std::memset(&A, 0, sizeof(A));
  A.E = j->E;
  A.M = j->M;
  A.mode = mode;
  A.isrc = isrc;
  A.h_active = (j->free_mask[LC_P_H] || j->h_nonzero || want_hgrad) ? 1 : 0;
  A.need_hgrad = (mode == 0 && (j->free_mask[LC_P_H] || want_hgrad)) ? 1 : 0;
  A.data = j->data;
  A.wgt = j->wgt;
  A.St = j->St;
  A.spec = j->spec;
  A.twid = j->twid;
  A.a = j->par[LC_P_A];
  A.cx = j->par[LC_P_CX];
  A.cy = j->par[LC_P_CY];
  A.dx = j->par[LC_P_DX];
  A.dy = j->par[LC_P_DY];
  A.alpha = j->par[LC_P_ALPHA];
  A.h = j->par[LC_P_H];
  A.mean = j->par[LC_P_MEAN];
  A.tabs = j->tabs;
  A.HG = j->HG;
  A.chi2_e = j->chi2_e;
  A.g_a = j->g_a;
  A.g_cx_e = j->g_cx_e;
  A.g_cy_e = j->g_cy_e;
  A.g_dx = j->g_dx;
  A.g_dy = j->g_dy;
  A.g_mean = j->g_mean;
  A.model_out = model_out;
  A.part = j->part;
  A.fisher_out = j->fisher;
  A.group = j->group_dev;
  A.skip_D = (mode == 0 && j->fuse_stencil) ? 1 : 0;
  A.tshift = j->tshift;
  // the LAST launch below carries the wait for this iteration's chain (one extra block), where one is due
  const bool carry_wait = j->epoch_wait_due && !ps_stream && mode == 0;
  j->epoch_wait_due = false;
  auto attach_wait = [&](int wait_block) {
    A.chain_flag = j->reg_flag;
    A.chain_seq = j->reg_seq;
    A.chain_err = j->reg_flag + 1;
    A.wait_block = wait_block;
    j->epoch_waited = j->epoch_wait_used = true;
  };
  if (j->upd_signal_due && !ps_stream) {   // the first launch below opens the gate of this iteration's chain
    A.upd_signal = j->upd_ctr;
    A.upd_value = j->upd_seq;
    j->upd_signal_due = false;
  }
  // no background in the scene: separable Gaussian filtering of the epoch PSFs instead of the FFT pipeline
  if (!A.h_active && j->M > 0 && j->psf_dev && !std::getenv("LCMI_JOINT_FFT_ONLY")) {
    ps_fn pk = nullptr;
    int plds = 0;
    find_ps_kernel(j->N, j->ss, &pk, &plds);
    if (pk) {
      if (!j->psF) {
        int rc = dmalloc(j, &j->psF, (size_t)j->E * j->M * 3 * j->n * j->n);
        if (rc) return rc;
      }
      JointPsArgs P;
      std::memset(&P, 0, sizeof(P));  // (fields a launch does not use are passed as zeros, not as stack contents)
      P.J = A;
      P.psf = j->psf_dev;
      P.F = j->psF;
      if (j->ps_attr_fn != (const void *)pk) {  // (once per kernel, not per launch: the batched loop enqueues thousands)
        LC_HIP(j->ctx, hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, plds));
        j->ps_attr_fn = (const void *)pk;
      }
      if (e1 < 0) e1 = j->E;
      P.e_off = e0;
      if (A.upd_signal)   // (this kernel does not raise the word itself)
        hipLaunchKernelGGL(mreg_signal_kernel, dim3(1), dim3(64), 0, j->ctx->stream, A.upd_signal, A.upd_value);
      hipLaunchKernelGGL(pk, dim3(e1 - e0), dim3(kPsThreads), plds, ps_stream ? ps_stream : j->ctx->stream, P);
      LC_HIP(j->ctx, hipGetLastError());
      return 0;
    }
  }
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "batched star photometry runs the point-source-only kernel (LCMI_JOINT_FFT_ONLY is set?)");
  // global-spectrum kernels: with many workgroups in flight the 8 / 16-byte column accesses saturate L2 / Infinity Cache and
  // the LDS-tile variant of the column passes wins (128 x 128 ROIs: 32 epochs +1.4 %, 64 +-0, 125 -1 %, 160 -7 %, 200 -16 %,
  // 1000 -15 %; LCMI_TILE_COLS=0/1 overrides the choice)
  // ... and an epoch that would leave CUs idle is spread over several workgroups, one launch per phase (LCMI_EPOCH_PARTS)
  // (C5 shard, 125 epochs: 425 us per iteration as one kernel, 295 us with two workgroups per epoch; 64 epochs 409 -> 198 us
  //  with four; 150 epochs and up: the single kernel, whose workgroups overlap each other's phases, is as fast or faster -
  //  200 epochs 451 against 469 us.  LCMI_EPOCH_PARTS_COL: the count for the column phases alone, a tuning switch.)
  if (const int P = (mode == 0 && !A.skip_D) ? cluster_parts(j) : 0) {
    A.cl_ctr = j->cl_ctr;
    A.cl_abort = j->cl_ctr + (size_t)j->E * kClStride;
    A.cl_base = j->cl_base;
    A.cl_parts = P;
    const int groups = (j->E + 7) / 8;
    LC_HIP(j->ctx, hipFuncSetAttribute((const void *)v->ek_cluster, hipFuncAttributeMaxDynamicSharedMemorySize, v->cl_lds));
    if (carry_wait) attach_wait(8 * groups * P);
    hipLaunchKernelGGL(v->ek_cluster, dim3(8 * groups * P + (carry_wait ? 1 : 0)), dim3(v->cl_thr), v->cl_lds, j->ctx->stream, A);
    LC_HIP(j->ctx, hipGetLastError());
    j->cl_base += (unsigned int)kClBarriers;
    j->cl_parts_last = P;
    return A.need_hgrad;
  }
  j->cl_parts_last = 0;
  int parts = 1, parts_col = 1;
  if (v->ek_phase[0] && mode == 0 && j->part) {
    // as many workgroups per epoch as leave no CU idle, and no more than one sweep of a phase has work for: the row phases
    // of an epoch are N / 2 row pairs, its column phases L / 2 columns, a workgroup takes e_thr / LPF of them per sweep
    const int per_wg = v->e_thr / v->lpf;
    const int cap_row = std::max(1, (j->N / 2 + per_wg - 1) / per_wg), cap_col = std::max(1, (j->L / 2 + per_wg - 1) / per_wg);
    const int fit = std::max(1, j->ctx->n_cu / std::max(j->E, 1));
    parts = std::min({kMaxParts, fit, cap_row});
    if (const char *ep = std::getenv("LCMI_EPOCH_PARTS")) parts = std::min(kMaxParts, std::max(1, std::atoi(ep)));
    parts_col = (parts > 1) ? std::min({kMaxParts, std::max(fit, parts), cap_col}) : 1;
    if (std::getenv("LCMI_EPOCH_PARTS")) parts_col = parts;
    if (const char *ep = std::getenv("LCMI_EPOCH_PARTS_COL")) parts_col = std::min(kMaxParts, std::max(1, std::atoi(ep)));
  }
  const bool phased = parts > 1;
  bool tile = v->ek_tile && j->E * parts >= 96;
  if (const char *tc = std::getenv("LCMI_TILE_COLS")) tile = v->ek_tile && std::atoi(tc) != 0;
  if (phased) {
    for (int ph = 0; ph < 6; ++ph) {
      if (ph == 5 && (!(A.h_active && A.need_hgrad) || A.skip_D)) continue;
      const bool tl = tile && v->ek_phase_tile[ph];
      const bool col = (ph == 1 || ph == 3), lite = col || ph == 5;
      epoch_fn pk = tl ? v->ek_phase_tile[ph] : v->ek_phase[ph];
      const int plds = lite ? (tl ? v->e_lds_lite_tile : v->e_lds_lite) : v->e_lds;
      const int np = col ? parts_col : parts;
      LC_HIP(j->ctx, hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, plds));
      const bool last = (ph == 5) && carry_wait;
      if (last) attach_wait(j->E);
      hipLaunchKernelGGL(pk, dim3(j->E + (last ? 1 : 0), np), dim3(v->e_thr), plds, j->ctx->stream, A);
      A.upd_signal = nullptr;   // (the first phase's launch has raised the word)
    }
    if (!(A.h_active && A.need_hgrad) || A.skip_D)  // (phase D's first workgroup of an epoch adds up the partial sums otherwise)
      hipLaunchKernelGGL(joint_epoch_finish_kernel, dim3(j->E), dim3(64), 0, j->ctx->stream, A, j->ss, parts);
    LC_HIP(j->ctx, hipGetLastError());
    return A.need_hgrad;
  }
  epoch_fn ek = tile ? v->ek_tile : v->ek;
  const int e_lds = tile ? v->e_lds_tile : v->e_lds;
  LC_HIP(j->ctx, hipFuncSetAttribute((const void *)ek, hipFuncAttributeMaxDynamicSharedMemorySize, e_lds));
  if (carry_wait) attach_wait(j->E);
  hipLaunchKernelGGL(ek, dim3(j->E + (carry_wait ? 1 : 0)), dim3(v->e_thr), e_lds, j->ctx->stream, A);
  LC_HIP(j->ctx, hipGetLastError());
  return A.need_hgrad;
}

int launch_reduce(lc_joint *j, int need_h) {
  const int NN = j->N * j->N;
  if (j->peer_fuse) {
    lc_peer::PeerArgs P;
    unsigned int *arrive = nullptr, fcall = 0;
    int rc = lc_peer_next_call(j->peer_fuse, j->shared_count, &P, &arrive, &fcall);
    if (rc) {
      j->ctx->err = j->peer_fuse->ctx->err;
      return rc;
    }
    hipLaunchKernelGGL(joint_reduce_peer_kernel, dim3(NN / (kRedPix * kRpTiles) + 1), dim3(kRedThreads), 0, j->ctx->stream, j->E, j->M, NN,
                       need_h, j->HG, j->g_cx_e, j->g_cy_e, j->chi2_e, j->par[LC_P_A], j->a_ref, j->shared, P, arrive, fcall);
    LC_HIP(j->ctx, hipGetLastError());
    j->peer_fused_done = true;
    return LC_OK;
  }
  const int nimg = NN / kRedPix;
  hipLaunchKernelGGL(joint_reduce_kernel, dim3(nimg + 1), dim3(kRedThreads), 0, j->ctx->stream, j->E, j->M, NN,
                     need_h, j->HG, j->g_cx_e, j->g_cy_e, j->chi2_e, j->par[LC_P_A], j->a_ref, j->shared);
  LC_HIP(j->ctx, hipGetLastError());
  return LC_OK;
}

// starlet l1 + positivity of h as multi-block kernels: -> greg, regs (same contract as reg_mode 1 of joint_update_kernel)
// pts_only: the background part (greg, regs[0..1]) was evaluated by the chain on the second stream; only the point-source
// starlet term, which in a sharded fit needs the all-reduced mean fluxes, is evaluated here (12 short launches instead of
// the whole cascade once more: 125 epochs of 128 x 128 in the sharded loop 541 -> see DESIGN.md section 6)
int launch_reg_gm(lc_joint *j, hipStream_t stream, bool with_pts, bool abar_from_shared, bool pts_only = false) {
  const int N = j->N, NN = N * N, J = j->J, nb = (NN + kGmThreads - 1) / kGmThreads;
  const dim3 grid(nb), block(kGmThreads);
  const float *W = j->have_W ? j->W : nullptr;
  const bool l1_on = (j->cfg.lam_scales != 0.f || j->cfg.lam_hf != 0.f);
  if (!pts_only) {
  LC_HIP(j->ctx, hipMemcpyAsync(j->gm_c, j->par[LC_P_H], (size_t)NN * sizeof(float), hipMemcpyDeviceToDevice, stream));
  if (l1_on) {
    for (int s = 0; s < J; ++s) {
      const int d = 1 << s;
      const float lam = (s == 0) ? j->cfg.lam_hf : j->cfg.lam_scales;
      hipLaunchKernelGGL(gm_pass_kernel, grid, block, 0, stream, N, d, 1, j->gm_c, j->gm_t);
      hipLaunchKernelGGL(gm_pass_kernel, grid, block, 0, stream, N, d, 0, j->gm_t, j->gm_n);
      hipLaunchKernelGGL(gm_coef_kernel, grid, block, 0, stream, N, j->gm_n, j->gm_c, W ? W + (size_t)s * NN : nullptr,
                         j->norms + s, lam, j->qscr + (size_t)s * NN, j->gm_l1 + (size_t)s * nb);
    }
    // z_J = 0; z_j = q_j + Row_j^T Col_j^T (z_{j+1} - q_j)
    LC_HIP(j->ctx, hipMemsetAsync(j->greg, 0, (size_t)NN * sizeof(float), stream));
    for (int s = J - 1; s >= 0; --s) {
      const int d = 1 << s;
      const float *q = j->qscr + (size_t)s * NN;
      hipLaunchKernelGGL(gm_sub_kernel, grid, block, 0, stream, NN, j->greg, q, j->gm_y);
      hipLaunchKernelGGL(gm_edge_kernel, dim3(N), dim3(64), 0, stream, N, d, 0, j->gm_y, j->gm_edge);
      hipLaunchKernelGGL(gm_pass_adjoint_kernel, grid, block, 0, stream, N, d, 0, j->gm_y, j->gm_edge, (const float *)nullptr, j->gm_t);
      hipLaunchKernelGGL(gm_edge_kernel, dim3(N), dim3(64), 0, stream, N, d, 1, j->gm_t, j->gm_edge);
      hipLaunchKernelGGL(gm_pass_adjoint_kernel, grid, block, 0, stream, N, d, 1, j->gm_t, j->gm_edge, q, j->greg);
    }
  } else {
    LC_HIP(j->ctx, hipMemsetAsync(j->greg, 0, (size_t)NN * sizeof(float), stream));
  }
  hipLaunchKernelGGL(gm_positivity_kernel, grid, block, 0, stream, NN, j->par[LC_P_H], j->cfg.lam_positivity, j->greg, j->gm_pos);
  hipLaunchKernelGGL(gm_regs_kernel, dim3(1), dim3(64), 0, stream, l1_on ? J * nb : 0, nb, j->gm_l1, j->gm_pos, j->regs);
  }  // !pts_only
  if (with_pts && pts_only && abar_from_shared && N % kPtT == 0 && !std::getenv("LCMI_PTS_CHAIN")) {  // (LCMI_PTS_CHAIN=1: the twelve-launch form, the cross-check)
    // behind the all-reduce of a sharded fit, on the critical path: the term in one launch (joint_gm.h)
    float *part = j->gm_pts + 8, *l1p = j->gm_l1 + (size_t)J * nb;
    hipLaunchKernelGGL(gm_pts_direct_kernel, dim3((N / kPtT) * (N / kPtT)), block, 0, stream, N, j->ss, j->M, j->a_ref, j->shared,
                       j->par[LC_P_CX], j->par[LC_P_CY], W, j->norms, j->cfg.lam_pts_source, part, l1p);
    hipLaunchKernelGGL(gm_pts_final_kernel, dim3(1), dim3(1024), 0, stream, (N / kPtT) * (N / kPtT), j->M, part, l1p, j->regs);
  } else if (with_pts) {
    // point-source starlet term: scale 0 only, on Pbar (the work buffers of the chain above are free again)
    float *abar = j->gm_pts, *part = j->gm_pts + 8, *qp = j->qscr + (size_t)J * NN, *l1p = j->gm_l1 + (size_t)J * nb;
    hipLaunchKernelGGL(gm_abar_kernel, dim3(1), dim3(256), 0, stream, j->E, j->M, NN, j->par[LC_P_A], j->a_ref, j->shared,
                       abar_from_shared ? 1 : 0, abar);
    hipLaunchKernelGGL(gm_pbar_kernel, grid, block, 0, stream, N, j->ss, j->M, abar, j->par[LC_P_CX], j->par[LC_P_CY], j->gm_c);
    hipLaunchKernelGGL(gm_pass_kernel, grid, block, 0, stream, N, 1, 1, j->gm_c, j->gm_t);
    hipLaunchKernelGGL(gm_pass_kernel, grid, block, 0, stream, N, 1, 0, j->gm_t, j->gm_n);
    hipLaunchKernelGGL(gm_coef_kernel, grid, block, 0, stream, N, j->gm_n, j->gm_c, W, j->norms, j->cfg.lam_pts_source, qp, l1p);
    hipLaunchKernelGGL(gm_edge_kernel, dim3(N), dim3(64), 0, stream, N, 1, 0, qp, j->gm_edge);
    hipLaunchKernelGGL(gm_pass_adjoint_kernel, grid, block, 0, stream, N, 1, 0, qp, j->gm_edge, (const float *)nullptr, j->gm_t);
    hipLaunchKernelGGL(gm_edge_kernel, dim3(N), dim3(64), 0, stream, N, 1, 1, j->gm_t, j->gm_edge);
    hipLaunchKernelGGL(gm_pass_adjoint_kernel, grid, block, 0, stream, N, 1, 1, j->gm_t, j->gm_edge, (const float *)nullptr, j->gm_y);
    hipLaunchKernelGGL(gm_pts_inner_kernel, grid, block, 0, stream, N, j->ss, j->M, qp, j->gm_y, j->par[LC_P_CX], j->par[LC_P_CY], part);
    hipLaunchKernelGGL(gm_pts_final_kernel, dim3(1), dim3(1024), 0, stream, nb, j->M, part, l1p, j->regs);
  }
  LC_HIP(j->ctx, hipGetLastError());
  return LC_OK;
}

template <int N>
MregKernels make_mreg() {
  MregKernels k{N, mreg_forward_kernel<N>, mreg_adjoint_kernel<N>, MregCfg<N>::LDS_FWD, MregCfg<N>::LDS_ADJ, MregCfg<N>::NTHR,
                mreg_mm_kernel<N>, nullptr};
  k.mmx[0] = mreg_mmx_kernel<N, 0>;
  k.mmx[1] = mreg_mmx_kernel<N, 1>;
  k.mmx[2] = mreg_mmx_kernel<N, 2>;
  k.mmx[3] = mreg_mmx_kernel<N, 3>;
  return k;
}
template <int N>
MregKernels make_mreg_chain() {
  MregKernels k = make_mreg<N>();
  k.chain = mreg_chain_kernel<N>;
  k.rows = mreg_rows_kernel<N>;
  k.rows_lds = RrCfg<N>::LDS_BYTES;
  return k;
}
const MregKernels *find_mreg(int N) {
  static const MregKernels table[] = {make_mreg_chain<128>(), make_mreg<256>()};
  if (std::getenv("LCMI_REG_CASCADE")) return nullptr;  // the a-trous cascade kernels instead (cross-check)
  for (const auto &k : table)
    if (k.N == N) return &k;
  return nullptr;
}
// cumulative 1-D smoothing operators A_0 = I, A_{s+1} = R_s A_s (R_s: edge-replicating B3 filter at dilation 2^s), in
// double on the host; A and A^T as [J + 1][N][N] floats
void build_cumulative_operators(int N, int J, std::vector<float> &A, std::vector<float> &AT) {
  const size_t NN = (size_t)N * N;
  std::vector<double> cur(NN, 0.0), nxt(NN);
  for (int r = 0; r < N; ++r) cur[(size_t)r * N + r] = 1.0;
  A.assign((size_t)(J + 1) * NN, 0.f);
  AT.assign((size_t)(J + 1) * NN, 0.f);
  const double b[5] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
  for (int s = 0; s <= J; ++s) {
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) {
        A[(size_t)s * NN + (size_t)r * N + c] = (float)cur[(size_t)r * N + c];
        AT[(size_t)s * NN + (size_t)c * N + r] = (float)cur[(size_t)r * N + c];
      }
    if (s == J) break;
    const int d = 1 << s;
    std::fill(nxt.begin(), nxt.end(), 0.0);
    for (int r = 0; r < N; ++r)
      for (int t = -2; t <= 2; ++t) {
        const int rr = std::min(std::max(r + t * d, 0), N - 1);
        const double bt = b[t + 2];
        for (int c = 0; c < N; ++c) nxt[(size_t)r * N + c] += bt * cur[(size_t)rr * N + c];
      }
    cur.swap(nxt);
  }
}

// starlet l1 + positivity of h (+ the point-source starlet term, mean fluxes from the parameters) on the matrix cores:
// -> greg, regs (same contract as reg_mode 1 of joint_update_kernel)
int launch_reg_mfma(lc_joint *j, hipStream_t stream, bool with_pts) {
  const MregKernels *k = j->mreg;
  const int N = j->N, NN = N * N, J = j->J, nb = (NN + kGmThreads - 1) / kGmThreads;
  const bool l1_on = (j->cfg.lam_scales != 0.f || j->cfg.lam_hf != 0.f);
  MregArgs A;
  std::memset(&A, 0, sizeof(A));
  A.J = J;
  A.has_pts = with_pts ? 1 : 0;
  A.s0 = l1_on ? 0 : J;
  A.A = j->mr_A;
  A.AT = j->mr_AT;
  A.X = j->par[LC_P_H];
  A.P = j->mr_pbar;
  A.C = j->mr_C;
  A.W = j->have_W ? j->W : nullptr;
  A.norms = j->norms;
  A.lam_sc = j->cfg.lam_scales;
  A.lam_hf = j->cfg.lam_hf;
  A.lam_pts = j->cfg.lam_pts_source;
  A.Z = j->mr_Z;
  A.l1p = j->mr_l1;
  if (const char *dly = std::getenv("LCMI_REG_DELAY_US"))  // test hook: a chain that finishes after the epoch kernel
    hipLaunchKernelGGL(mreg_delay_kernel, dim3(1), dim3(64), 0, stream, (long long)(std::atof(dly) * 100.0));  // wall_clock64: 100 MHz
  auto launch_pbar = [&]() {
    if (with_pts)
      hipLaunchKernelGGL(mreg_pbar_kernel, dim3(nb), dim3(kGmThreads), 0, stream, N, j->ss, j->E, j->M, j->par[LC_P_A],
                         j->par[LC_P_CX], j->par[LC_P_CY], j->mr_pbar);
  };
  // Third form (LCMI_REG_ROWS=1; N = 128): the regulariser cut by ROWS - one launch does forward products, S rows and adjoint
  // products of a row block for a group of scales, a second adds the partial planes and the values; the point-source term as
  // tiles in a launch of its own in front.  Three launches for the eight of the batched-product form (joint_reg_rows.h).
  // Correct (tests/test_joint_paths_gpu.py: against the cascade and the second form) and NOT the default: measured on MI355X
  // (profiles/r04_rows_*) its row kernel takes 45 us (35 us with a rolled product loop), the chain 57 - 63 us against ~60 us
  // for the eight launches - a workgroup's 18 dependent 16-row products cost ~5 k cycles each, of which ~2.5 k remain with the
  // operand loads and the matrix products switched off (LCMI_REG_ROWS_DBG=7): a few hundred instructions per product issued
  // by two waves per SIMD, not memory and not the matrix pipe.  C4 83.9 against 71.1 us per iteration, its 25-epoch shard 76.8
  // against 65.0.
  // (its completion signal is a counter - every block of the last launch adds one - so only the flag-reading update of the
  //  device loop and the event-waiting forms can follow it: both read reg_flag / wait for the stream)
  {
    const char *rr_env = std::getenv("LCMI_REG_ROWS");
    if (k->rows && j->rr_Zp && l1_on && J == 7 && rr_env && std::atoi(rr_env) != 0 && !std::getenv("LCMI_REG_MFMA_V1")) {
      // scale groups per row block, of about equal cost (the products of a scale cover its band: the first scales are cheap);
      // at most kRrBatch scales each: three groups {1 .. 4}, {5, 6}, {7} (default) or two {1 .. 4}, {5 .. 7}
      int nparts = 3;
      if (const char *pe = std::getenv("LCMI_REG_ROWS_PARTS")) nparts = std::min(std::max(2, std::atoi(pe)), kRrMaxParts);
      MregRowsArgs Q;
      std::memset(&Q, 0, sizeof(Q));
      Q.J = J;
      Q.nparts = nparts;
      const int lo3[3] = {1, 5, 7}, hi3[3] = {4, 6, 7}, lo2[2] = {1, 5}, hi2[2] = {4, 7};
      for (int p = 0; p < nparts; ++p) {
        Q.s_lo[p] = (nparts == 3) ? lo3[p] : lo2[p];
        Q.s_hi[p] = (nparts == 3) ? hi3[p] : hi2[p];
      }
      Q.A = j->mr_A;
      Q.AT = j->mr_AT;
      Q.X = j->par[LC_P_H];
      Q.W = j->have_W ? j->W : nullptr;
      Q.norms = j->norms;
      Q.lam_sc = j->cfg.lam_scales;
      Q.lam_hf = j->cfg.lam_hf;
      Q.lam_pos = j->cfg.lam_positivity;
      Q.Zp = j->rr_Zp;
      Q.S0 = j->mr_S;
      Q.vals = j->mr_l1b;
      if (const char *dg = std::getenv("LCMI_REG_ROWS_DBG")) Q.dbg = std::atoi(dg);
      const int tiles = (N / kPtT) * (N / kPtT), nwg = (N / kRrRows) * nparts;
      if (with_pts)
        hipLaunchKernelGGL(gm_pts_direct_kernel, dim3(tiles), dim3(kGmThreads), 0, stream, N, j->ss, j->M, j->a_ref, (const float *)nullptr,
                           j->par[LC_P_CX], j->par[LC_P_CY], Q.W, j->norms, j->cfg.lam_pts_source, j->mr_part, j->mr_posb,
                           (const float *)j->par[LC_P_A], j->E);
      LC_HIP(j->ctx, hipFuncSetAttribute((const void *)k->rows, hipFuncAttributeMaxDynamicSharedMemorySize, k->rows_lds));
      hipLaunchKernelGGL(k->rows, dim3(nwg), dim3(kRrThreads), k->rows_lds, stream, Q);
      const int nfin = NN / kGmThreads + 1;
      j->reg_seq += (unsigned int)nfin;
      hipLaunchKernelGGL(mreg_rows_finish_kernel, dim3(nfin), dim3(kGmThreads), 0, stream, NN, nwg, j->rr_Zp, j->mr_S, j->greg,
                         j->mr_l1b, with_pts ? 1 : 0, tiles, j->M, j->mr_part, j->mr_posb, j->regs, j->reg_flag);
      LC_HIP(j->ctx, hipGetLastError());
      return LC_OK;
    }
  }
  if (!std::getenv("LCMI_REG_MFMA_V1")) {
    // second form: batched tiled products over the scales, telescoped adjoint
    const size_t NNs = (size_t)NN;
    auto At = [&](int s) { return j->mr_AT + (size_t)s * NNs; };
    auto Ap = [&](int s) { return j->mr_A + (size_t)s * NNs; };
    // LCMI_PTS_SIDE=1: the point-source starlet term (scale 0 only) beside the chain instead of inside it - as tiles in ONE
    // launch (gm_pts_direct_kernel, joint_gm.h; mean fluxes from the fluxes themselves) on a third stream, joined in front of
    // the sums; inside the chain it is a launch of its own for Pbar plus one more product in each of the four batches.  Equal
    // to the batched form to fp32 rounding (tests/test_joint_paths_gpu.py).  Built to take the Pbar launch off a chain that
    // ends just after the epoch kernel; measured SLOWER and not the default: C4 81.0 against 70.9 us per iteration, its
    // 25-epoch shard 66.4 / 64.4, the C5 shard 235.5 / 230.7 - the third stream's fork and join (two more cross-stream event
    // waits per iteration, a third hardware queue) cost more than the launch they remove.  (In a process that has created many
    // streams the runtime may map two of the three onto one hardware queue; with the update's in-kernel wait for the chain
    // (the default) a run of this form then timed out once in the test suite: use it with LCMI_EVENT_SYNC=1.)
    const char *ps_env = std::getenv("LCMI_PTS_SIDE");
    const bool pts_side = with_pts && stream == j->streamB && N % kPtT == 0 && ps_env && std::atoi(ps_env) != 0 && !std::getenv("LCMI_REG_CHAIN");
    if (pts_side) {
      if (!j->streamC) {
        LC_HIP(j->ctx, hipStreamCreate(&j->streamC));
        LC_HIP(j->ctx, hipEventCreateWithFlags(&j->evPts, hipEventDisableTiming));
      }
      LC_HIP(j->ctx, hipStreamWaitEvent(j->streamC, j->evUpd, 0));   // (what the chain's own stream waited for)
      const int tiles = (N / kPtT) * (N / kPtT);
      hipLaunchKernelGGL(gm_pts_direct_kernel, dim3(tiles), dim3(kGmThreads), 0, j->streamC, N, j->ss, j->M, j->a_ref, (const float *)nullptr,
                         j->par[LC_P_CX], j->par[LC_P_CY], j->have_W ? j->W : nullptr, j->norms, j->cfg.lam_pts_source, j->mr_part,
                         j->mr_l1b + (size_t)(J + 1) * nb, (const float *)j->par[LC_P_A], j->E);
      LC_HIP(j->ctx, hipEventRecord(j->evPts, j->streamC));
    }
    // Fourth form (default; LCMI_REG_FUSED=0: the eight launches below): the element-wise launches folded into the products and
    // the point-source term evaluated from separable tables instead of a ninth product (joint_reg_fused.h) - f1 (+ the
    // point-source blocks), f2, a1' (S planes in the operand fetch, values per tile), a2' (completion counter); the fused
    // reduction + update adds the planes, any other consumer gets greg / regs from one more launch.
    const char *rc_env = std::getenv("LCMI_REG_CHAIN");
    const char *fu_env = std::getenv("LCMI_REG_FUSED");
    const bool fused = k->mmx[0] && l1_on && J <= kPlanesMaxJ && J <= 12 && !pts_side && !(rc_env && std::atoi(rc_env) != 0) &&
                       !(fu_env && std::atoi(fu_env) == 0) && NN % (kPtsBlocks * kMmThreads) == 0;
    const bool pts_batch = with_pts && !pts_side && !fused;
    MmBatch f1, f2, a1, a2;
    std::memset(&f1, 0, sizeof(f1));
    f2 = a1 = a2 = f1;
    int nbch = 0;
    auto add = [&](int s, const float *in, float *cplane, const float *splane, float *zplane) {
      float *T = j->mr_T + (size_t)nbch * NNs;
      f1.A[nbch] = in;      f1.B[nbch] = At(s); f1.C[nbch] = T;        // T = X AT_s
      f2.A[nbch] = Ap(s);   f2.B[nbch] = T;     f2.C[nbch] = cplane;   // c_s = A_s T
      a1.A[nbch] = splane;  a1.B[nbch] = Ap(s); a1.C[nbch] = T;        // T' = S_s A_s
      a2.A[nbch] = At(s);   a2.B[nbch] = T;     a2.C[nbch] = zplane;   // Z_s = AT_s T'
      // A_s and its transpose are banded with half-width 2 (2^s - 1): as the second operand the band follows the tile's
      // columns, as the first its rows (LCMI_REG_DENSE=1: all K slices, the cross-check)
      const int hw = 2 * ((1 << s) - 1), on = std::getenv("LCMI_REG_DENSE") ? 0 : 1;
      f1.band[nbch] = a1.band[nbch] = on * 1;
      f2.band[nbch] = a2.band[nbch] = on * 2;
      f1.hw[nbch] = f2.hw[nbch] = a1.hw[nbch] = a2.hw[nbch] = hw;
      ++nbch;
    };
    if (l1_on)
      for (int s = 1; s <= J; ++s) add(s, j->par[LC_P_H], j->mr_C + (size_t)s * NNs, j->mr_S + (size_t)s * NNs, j->mr_Z + (size_t)s * NNs);
    if (pts_batch) add(1, j->mr_pbar, j->mr_C + (size_t)(J + 1) * NNs, j->mr_S + (size_t)(J + 1) * NNs, j->mr_Z + (size_t)(J + 1) * NNs);
    f1.nb = f2.nb = a1.nb = a2.nb = nbch;
    // The whole chain as ONE launch (mreg_chain_kernel; LCMI_REG_CHAIN=1) where its kChainBlocks workgroups are resident beside
    // the epoch kernel's.  Same stages, same bits (tests/test_joint_cluster_gpu.py).  Built on the premise that the eight
    // launches (5 - 7.5 us each for 1 - 2 us of work, ~60 us per iteration at N = 128) are launch overhead; measured, they
    // are not: the one launch takes 59.4 us per iteration (profiles/r04_cluster_*) - every stage boundary is a hand-off
    // between CUs (write-through stores, drain, flag, L1-bypassing loads from the memory side), 5 - 7 us whether a launch
    // boundary or an in-kernel sync delivers it.  NOT the default; what shortens the chain is fewer stages, not cheaper ones.
    if (fused) {
      MmxArgs Q;
      std::memset(&Q, 0, sizeof(Q));
      for (int b = 0; b < nbch; ++b) Q.scale[b] = b + 1;
      Q.J = J;
      Q.ntile = (N / 64) * (N / 64);
      Q.X = j->par[LC_P_H]; Q.C = j->mr_C; Q.W = j->have_W ? j->W : nullptr; Q.norms = j->norms;
      Q.lam_sc = j->cfg.lam_scales; Q.lam_hf = j->cfg.lam_hf; Q.lam_pos = j->cfg.lam_positivity;
      Q.S = j->mr_S;
      Q.vals = j->mr_l1b;
      const dim3 mgrid(N / 64, N / 64, nbch), mblock(kMmThreads);
      dim3 g1 = mgrid;
      if (with_pts) {   // the point-source blocks ride in the first launch (they depend on nothing the chain computes)
        PtsSepArgs &P = Q.pts;
        P.N = N; P.ss = j->ss; P.E = j->E; P.M = j->M;
        P.a = j->par[LC_P_A]; P.cx = j->par[LC_P_CX]; P.cy = j->par[LC_P_CY];
        P.W0 = j->have_W ? j->W : nullptr; P.norms = j->norms; P.lam_pts = j->cfg.lam_pts_source;
        P.part = j->mr_part;
        g1.z += (kPtsBlocks + Q.ntile - 1) / Q.ntile;
      }
      Q.mm = f1;
      hipLaunchKernelGGL(k->mmx[0], g1, mblock, 0, stream, Q);
      Q.mm = f2;
      hipLaunchKernelGGL(k->mmx[3], mgrid, mblock, 0, stream, Q);   // (the plain product, without mreg_mm_kernel's 48 bytes of scratch per lane)
      Q.mm = a1;
      hipLaunchKernelGGL(k->mmx[1], mgrid, mblock, 0, stream, Q);
      Q.mm = a2;
      // who adds the planes up: the fused update itself when the chain is the longer path of the iteration (beside the cluster
      // form of the epoch kernel: one stage less on the chain, ~1.3 us more in the update), otherwise a fifth launch of the
      // chain, which then has the time (it ends ~15 us before a one-workgroup epoch kernel does)
      const bool planes = j->planes_pred && j->reg_flag && cluster_parts(j) >= 2;
      Q.done = planes ? j->reg_flag : nullptr;
      hipLaunchKernelGGL(k->mmx[2], mgrid, mblock, 0, stream, Q);
      RegPlanes &P = j->planes;
      P.on = 1; P.J = J; P.ntile = Q.ntile; P.npts = with_pts ? kPtsBlocks : 0;
      P.S0 = j->mr_S; P.Z = j->mr_Z; P.vals = j->mr_l1b; P.pts_part = j->mr_part;
      if (planes) {
        j->reg_seq += (unsigned int)(mgrid.x * mgrid.y * mgrid.z);
        j->reg_planes = true;
      } else {
        // (the completion counter also serves the multi-block update of the sharded drive, which checks it in its kernel)
        unsigned int *done = ((j->planes_pred || j->in_sharded_loop) && j->reg_flag) ? j->reg_flag : nullptr;
        hipLaunchKernelGGL(mreg_finish3_kernel, dim3(nb + 1), dim3(kGmThreads), 0, stream, NN, nb, j->M, P, j->greg, j->regs, done);
        if (done) j->reg_seq += (unsigned int)(nb + 1);
        else j->reg_noflag = true;
        j->reg_counter = done != nullptr;
      }
      LC_HIP(j->ctx, hipGetLastError());
      return LC_OK;
    }
    const int epoch_wgs = j->E * std::max(1, cluster_parts(j));
    if (k->chain && j->chain_flags && j->reg_flag && !j->chain_off && epoch_wgs + kChainBlocks <= j->ctx->n_cu && rc_env && std::atoi(rc_env) != 0) {
      MregChainArgs Q;
      std::memset(&Q, 0, sizeof(Q));
      Q.mm[0] = f1; Q.mm[1] = f2; Q.mm[2] = a1; Q.mm[3] = a2;
      Q.xa[0] = 0; Q.xb[0] = 0;   // T = X AT_s (h or Pbar: Pbar is a product of this launch)
      Q.xa[1] = 0; Q.xb[1] = 1;   // c_s = A_s T
      Q.xa[2] = 1; Q.xb[2] = 0;   // T' = S_s A_s
      Q.xa[3] = 0; Q.xb[3] = 1;   // Z_s = AT_s T'
      if (with_pts) Q.xa[0] = 2;  // (the last product of the batch reads Pbar: see chain_mm_tile's caller)
      Q.G.B = A;
      Q.G.lam_pos = j->cfg.lam_positivity;
      Q.G.has_l1 = l1_on ? 1 : 0;
      Q.G.S = j->mr_S;
      Q.G.l1b = j->mr_l1b;
      Q.G.posb = j->mr_posb;
      Q.sslots = (l1_on ? J + 1 : 1) + (with_pts ? 1 : 0);
      Q.with_pts = with_pts ? 1 : 0;
      Q.N = N; Q.ss = j->ss; Q.E = j->E; Q.M = j->M; Q.J = J; Q.has_l1 = l1_on ? 1 : 0;
      Q.a = j->par[LC_P_A]; Q.cx = j->par[LC_P_CX]; Q.cy = j->par[LC_P_CY];
      Q.pbar = j->mr_pbar;
      Q.S = j->mr_S; Q.Z = j->mr_Z;
      Q.greg = j->greg; Q.pts_part = j->mr_part;
      Q.regs = j->regs;
      j->reg_seq += 1;
      Q.done_flag = j->reg_flag; Q.done_seq = j->reg_seq;
      Q.flags = j->chain_flags; Q.base = j->chain_base;
      j->chain_base = (j->chain_base + (unsigned int)kChainSyncs) & 0x0fffffffu;
      hipLaunchKernelGGL(k->chain, dim3(kChainBlocks), dim3(kMmThreads), 0, stream, Q);
      LC_HIP(j->ctx, hipGetLastError());
      j->chain_used = true;
      return LC_OK;
    }
    const dim3 mgrid(N / 64, N / 64, nbch), mblock(kMmThreads);
    if (pts_batch) launch_pbar();
    if (nbch > 0) {
      hipLaunchKernelGGL(k->mm, mgrid, mblock, 0, stream, f1);
      hipLaunchKernelGGL(k->mm, mgrid, mblock, 0, stream, f2);
    }
    MregSArgs G;
    std::memset(&G, 0, sizeof(G));
    G.B = A;
    G.lam_pos = j->cfg.lam_positivity;
    G.has_l1 = l1_on ? 1 : 0;
    G.S = j->mr_S;
    G.l1b = j->mr_l1b;
    G.posb = j->mr_posb;
    G.B.has_pts = pts_batch ? 1 : 0;
    const int sslots = (l1_on ? J + 1 : 1) + (pts_batch ? 1 : 0);  // slot 0 always: it carries the positivity term
    hipLaunchKernelGGL(mreg_splanes_kernel, dim3(nb, sslots), dim3(kGmThreads), 0, stream, G, NN);
    if (nbch > 0) {
      hipLaunchKernelGGL(k->mm, mgrid, mblock, 0, stream, a1);
      hipLaunchKernelGGL(k->mm, mgrid, mblock, 0, stream, a2);
    }
    // (the tiles of the point-source term, on the third stream, are joined here: the sums below read their partials)
    if (pts_side) LC_HIP(j->ctx, hipStreamWaitEvent(stream, j->evPts, 0));
    hipLaunchKernelGGL(mreg_finish2_kernel, dim3(nb), dim3(kGmThreads), 0, stream, N, J, l1_on ? 1 : 0, pts_batch ? 1 : 0, j->ss, j->M,
                       j->mr_S, j->mr_Z, j->par[LC_P_CX], j->par[LC_P_CY], j->greg, j->mr_part);
    j->reg_seq += 1;
    hipLaunchKernelGGL(mreg_regs2_kernel, dim3(1), dim3(64), 0, stream, J, l1_on ? 1 : 0, with_pts ? 1 : 0, nb, j->M, j->mr_l1b,
                       j->mr_posb, j->mr_part, j->regs, j->reg_flag, j->reg_seq);
    LC_HIP(j->ctx, hipGetLastError());
    return LC_OK;
  }
  launch_pbar();
  const int slots = (l1_on ? J : 0) + (with_pts ? 1 : 0);
  if (slots > 0) {
    hipLaunchKernelGGL(k->fwd, dim3(N / 32, slots), dim3(k->nthr), k->lds_fwd, stream, A);
    hipLaunchKernelGGL(k->adj, dim3(N / 32, slots), dim3(k->nthr), k->lds_adj, stream, A);
  }
  hipLaunchKernelGGL(mreg_finish_kernel, dim3(nb), dim3(kGmThreads), 0, stream, N, J, l1_on ? 1 : 0, with_pts ? 1 : 0, j->ss,
                     j->M, j->mr_Z, j->par[LC_P_H], j->cfg.lam_positivity, j->par[LC_P_CX], j->par[LC_P_CY], j->greg,
                     j->mr_pos, j->mr_part);
  j->reg_seq += 1;
  hipLaunchKernelGGL(mreg_regs_kernel, dim3(1), dim3(64), 0, stream, J, l1_on ? 1 : 0, with_pts ? 1 : 0, nb, j->M, j->mr_l1,
                     j->mr_pos, j->mr_part, j->regs, j->reg_flag, j->reg_seq);
  LC_HIP(j->ctx, hipGetLastError());
  return LC_OK;
}

// May the event behind an update be left out?  Inside the library's loops, on the main stream, where its only reader is the
// next iteration's chain, which can start behind a gate kernel instead (LCMI_UPD_EVENT=1: the event, the cross-check; the
// third stream of LCMI_PTS_SIDE waits for the event too).
// One probe per object (mreg_probe_kernel): do the chain's stream and the main stream run side by side?
static int probe_streams(lc_joint *j) {
  if (j->streams_overlap >= 0 || !j->upd_ctr) return LC_OK;
  unsigned int *word = j->upd_ctr + 2, *seen = j->upd_ctr + 3;
  LC_HIP(j->ctx, hipMemsetAsync(word, 0, 2 * sizeof(unsigned int), j->ctx->stream));
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  LC_HIP(j->ctx, hipStreamSynchronize(j->streamB));
  hipLaunchKernelGGL(mreg_probe_kernel, dim3(1), dim3(64), 0, j->streamB, word, seen, 50000ll);   // at most 500 us
  hipLaunchKernelGGL(mreg_signal_kernel, dim3(1), dim3(64), 0, j->ctx->stream, word, 1u);
  LC_HIP(j->ctx, hipGetLastError());
  LC_HIP(j->ctx, hipStreamSynchronize(j->streamB));
  unsigned int ok = 0;
  int rc = d2h(j, &ok, seen, sizeof(ok));
  if (rc) return rc;
  j->streams_overlap = ok ? 1 : 0;
  if (std::getenv("LCMI_DEBUG_STREAMS")) std::fprintf(stderr, "lc_joint: second stream beside the main stream: %s\n", ok ? "yes" : "NO (shared hardware queue): events instead of the gate kernel");
  return LC_OK;
}
static bool upd_gate_ok(const lc_joint *j, hipStream_t stream) {
  if (j->streams_overlap != 1 || j->force_events) return false;
  // (LCMI_EVENT_SYNC=1 - what counter collection sets, which runs one kernel at a time: a gate kernel alone on the machine
  //  would wait for an epoch kernel that cannot start - keeps the event as well)
  return j->upd_ctr && (j->in_device_loop || j->in_sharded_loop) && stream == j->ctx->stream && !std::getenv("LCMI_UPD_EVENT") &&
         !std::getenv("LCMI_EVENT_SYNC") && !std::getenv("LCMI_PTS_SIDE");
}
// Wait (on `stream`) for the regulariser chain of this iteration through its event.  The event is recorded HERE, behind the chain
// on its own stream, the first time somebody needs it - the iterations whose consumer learns of the chain's completion in a
// kernel (completion counter, the epoch launch's extra block) never record it.
static int wait_for_chain_event(lc_joint *j, hipStream_t stream) {
  if (!j->evreg_recorded) {
    LC_HIP(j->ctx, hipEventRecord(j->evReg, j->streamB));
    j->evreg_recorded = true;
  }
  LC_HIP(j->ctx, hipStreamWaitEvent(stream, j->evReg, 0));
  return LC_OK;
}
int launch_update(lc_joint *j, int mode, int t, const lc_adabelief_cfg *cfg, bool write_hist, bool all_grads,
                  int reg_mode = 0, hipStream_t stream = nullptr) {
  if (!stream) stream = j->ctx->stream;
  const JointVariant *v = j->v;
  bool gm_pts_done = false;
  if (reg_mode == 1 && j->mreg) return launch_reg_mfma(j, stream, j->pts_pending);
  // a cross-stream wait for the chain that lc_joint_step_update left to this function: taken here by every consumer except the
  // multi-block update, which checks the chain's completion counter in its kernel
  auto take_event = [&]() -> int {
    if (j->defer_event) {
      j->defer_event = false;
      return wait_for_chain_event(j, stream);
    }
    return LC_OK;
  };
  if (j->defer_event && (j->fuse_full || j->fuse_pending)) {
    int rc = take_event();
    if (rc) return rc;
  }
  // Behind the chain of a sharded / step-by-step iteration (reg_mode 2) the point-source starlet term is still to do - it needs
  // the all-reduced mean fluxes.  Where the separable form applies (N >= 128) it rides INSIDE the multi-block update launch
  // (PtsTail, joint_gm.h): no launch of its own on the tail behind the all-reduce (two before: gm_pts_direct_kernel and its
  // final sums; LCMI_PTS_TAIL=0 brings them back, the cross-check).
  bool pts_tail = false;
  {
    const char *pt = std::getenv("LCMI_PTS_TAIL");
    const bool want_pts = j->cfg.lam_pts_source != 0.f && j->M > 0;
    pts_tail = reg_mode == 2 && want_pts && !j->pts_pending && j->pts_ctr && (j->N * j->N) % (kPtsBlocks * kGmThreads) == 0 &&
               (mode == 1 || !v->uk) && !(pt && std::atoi(pt) == 0) && !std::getenv("LCMI_PTS_CHAIN") && (size_t)4 * j->M * j->N * sizeof(float) <= 48 * 1024;
    if (pts_tail) gm_pts_done = true;
  }
  if (pts_tail) {
  } else if (!v->uk) {
    const bool want_pts = j->cfg.lam_pts_source != 0.f && j->M > 0;
    if (reg_mode == 1) return launch_reg_gm(j, stream, j->pts_pending, false);
    if (reg_mode == 2) gm_pts_done = j->pts_pending;  // evaluated by the stream-B chain of this iteration
    if ((reg_mode == 0 && (reg_h_on(j) || want_pts)) || (reg_mode == 2 && want_pts && !gm_pts_done)) {
      // inline (gradient evaluations, step-by-step / sharded drive): the mean fluxes come from the reduced block; behind a
      // chain that has already evaluated the background part (reg_mode 2) only the point-source term is left to do
      int rc = take_event();
      if (rc) return rc;
      rc = launch_reg_gm(j, stream, want_pts, true, reg_mode == 2);
      if (rc) return rc;
      reg_mode = 2;
      gm_pts_done = want_pts;
    }
  } else if (reg_mode == 2 && mode == 1 && j->gm_c && j->cfg.lam_pts_source != 0.f && j->M > 0 && !j->pts_pending) {
    // LDS-spectrum sizes in the sharded drive: the chain on the second stream has the background part, the point-source
    // term follows the all-reduce here - and the multi-block update takes over from the one-workgroup kernel, which would
    // evaluate everything once more by itself (95 us at N = 128)
    int rc = take_event();
    if (rc) return rc;
    rc = launch_reg_gm(j, stream, true, true, true);
    if (rc) return rc;
    gm_pts_done = true;
  }
  JointUpdArgs A;
  std::memset(&A, 0, sizeof(A));
  A.E = j->E;
  A.M = j->M;
  A.mode = mode;
  A.t = t;
  A.ss = j->ss;
  A.reg_mode = reg_mode;
  A.greg = j->greg;
  A.regs = j->regs;
  for (int k = 0; k < LC_P_COUNT; ++k) {
    A.free_mask[k] = j->free_mask[k];
    A.par[k] = j->par[k];
    A.pm[k] = j->pm[k];
    A.ps[k] = j->ps[k];
    A.gout[k] = (mode == 0 && all_grads) ? j->gout[k] : nullptr;
  }
  A.shared = j->shared;
  A.a_ref = j->a_ref;
  A.h = j->par[LC_P_H];
  A.mh = j->pm[LC_P_H];
  A.sh = j->ps[LC_P_H];
  A.W = j->have_W ? j->W : nullptr;
  A.norms = j->norms;
  A.qscr = j->qscr;
  A.g_a = j->g_a;
  A.g_dx = j->g_dx;
  A.g_dy = j->g_dy;
  A.g_mean = j->g_mean;
  A.hist = write_hist ? j->hist : nullptr;
  A.out_loss = j->out_loss;
  const bool rh = reg_h_on(j);
  A.lam_sc = rh ? j->cfg.lam_scales : 0.f;
  A.lam_hf = rh ? j->cfg.lam_hf : 0.f;
  A.lam_pos = rh ? j->cfg.lam_positivity : 0.f;
  A.lam_pos_ps = j->cfg.lam_positivity_ps;
  A.lam_pts = j->cfg.lam_pts_source;
  A.lam_fu = j->cfg.lam_flux_uniformity;
  A.n_prior = j->n_prior;
  A.prior_cx_mean = j->prior;
  A.prior_cx_sigma = j->prior + j->M;
  A.prior_cy_mean = j->prior + 2 * j->M;
  A.prior_cy_sigma = j->prior + 3 * j->M;
  if (cfg) A.ab = *cfg; else lc_adabelief_defaults(&A.ab);
  adabelief_schedule(A.ab, t, A.lr, A.bc1, A.bc2);
  if (mode == 1 && reg_mode != 1 && j->phist && j->phist_rows < j->phist_cap) {
    A.phist = j->phist + (size_t)j->phist_rows * j->phist_P;
    for (int k = 0; k < LC_P_COUNT; ++k) A.poff[k] = j->phist_off[k];
    j->phist_rows += 1;
  }
  // the multi-block update serves the large grids always, and the LDS variants whenever nothing is left for one
  // workgroup to do alone (h regulariser already evaluated on the second stream, no point-source starlet term):
  // N^2 / 256 blocks finish the AdaBelief sweep of h in a fraction of the single-workgroup latency
  A.pts_early = (reg_mode == 1) ? (j->pts_pending ? 1 : 0) : (((j->pts_pending && mode == 1) || gm_pts_done) ? 2 : 0);
  if (!v->uk || ((A.lam_pts == 0.f || A.pts_early == 2) && (reg_mode == 2 || !rh))) {
    const int NN = j->N * j->N;
    int nblk = (NN + kGmThreads - 1) / kGmThreads;
    if (j->fuse_full && mode == 1) {
      if (j->flag_sync) {
        A.wait_flag = j->reg_flag;
        A.wait_seq = j->reg_seq;
        A.wait_err = j->reg_flag + 1;
      }
      if (j->reg_planes && reg_mode == 2 && !j->fuse_stencil) A.planes = j->planes;
      A.g_cx_e = j->g_cx_e;
      A.g_cy_e = j->g_cy_e;
      A.chi2_e = j->chi2_e;
      A.shared_w = j->shared;
      // (16-pixel tiles per block; 256 x 256 grids: four - 4096 one-tile blocks spend more on block turnover than on the
      //  sums, C5 shard 294 -> 287 us per iteration; LCMI_UPDATE_TILES overrides)
      if (j->fuse_stencil) {
        StencilSrc S;
        const int KS = (j->KH + 15) / 16 * 16;
        S.gs = (const float *)j->spec;
        S.epoch_stride = (size_t)j->N * KS * 2;
        S.row_stride = 2 * KS;
        S.shifts = j->tshift;
        S.ss = j->ss;
        hipLaunchKernelGGL(joint_stencil_update_kernel, dim3(NN / kStPix + 2), dim3(kRedThreads), 0, stream, A, j->N, S);
        LC_HIP(j->ctx, hipGetLastError());
        return LC_OK;
      }
      int tiles = (NN / kRedPix >= 4096) ? 4 : (j->flag_sync ? 2 : 1);
      if (const char *tl = std::getenv("LCMI_UPDATE_TILES")) tiles = std::min(std::max(1, std::atoi(tl)), kUpdMaxTiles);
      if ((NN / kRedPix) % tiles) tiles = 1;
      const int nimg = NN / kRedPix / tiles;
      hipLaunchKernelGGL(joint_reduce_update_kernel, dim3(nimg + 2), dim3(kRedThreads), 0, stream, A, j->N, j->HG, tiles);
      LC_HIP(j->ctx, hipGetLastError());
      return LC_OK;
    }
    if (j->fuse_pending && mode == 1) {
      A.fuse_scalar_reduce = 1;
      A.g_cx_e = j->g_cx_e;
      A.g_cy_e = j->g_cy_e;
      A.chi2_e = j->chi2_e;
      A.shared_w = j->shared;
      nblk = 1;
    }
    if (j->defer_event) {   // the chain's counter instead of the event
      j->defer_event = false;
      A.wait_flag = j->reg_flag;
      A.wait_seq = j->reg_seq;
      A.wait_err = j->reg_flag + 1;
      j->gm_flag_used = true;
    }
    PtsTail T;
    std::memset(&T, 0, sizeof(T));
    size_t dyn = 0;
    if (pts_tail) {
      T.on = 1;
      PtsSepArgs &P = T.P;
      P.N = j->N; P.ss = j->ss; P.E = j->E; P.M = j->M;
      P.a = j->par[LC_P_A]; P.cx = j->par[LC_P_CX]; P.cy = j->par[LC_P_CY];
      P.abar_sum = j->shared + NN + 2 * j->M; P.a_ref = j->a_ref; P.n_total = j->shared + NN + 4 * j->M + 1;
      P.W0 = j->have_W ? j->W : nullptr; P.norms = j->norms; P.lam_pts = j->cfg.lam_pts_source;
      P.part = j->mr_part;
      P.write_through = 1;
      j->pts_seq += (unsigned int)kPtsBlocks;
      T.ctr = j->pts_ctr; T.seq = j->pts_seq; T.err = j->pts_ctr + 1;
      dyn = (size_t)4 * j->M * j->N * sizeof(float);
      j->pts_tail_used = true;
    }
    hipLaunchKernelGGL(joint_update_gm_kernel, dim3(nblk + (pts_tail ? kPtsBlocks : 0)), dim3(kGmThreads), dyn, stream, A, j->N, T);
    LC_HIP(j->ctx, hipGetLastError());
    return LC_OK;
  }
  {
    int rc = take_event();
    if (rc) return rc;
  }
  LC_HIP(j->ctx, hipFuncSetAttribute((const void *)v->uk, hipFuncAttributeMaxDynamicSharedMemorySize, v->u_lds));
  hipLaunchKernelGGL(v->uk, dim3(1), dim3(v->u_thr), v->u_lds, stream, A);
  LC_HIP(j->ctx, hipGetLastError());
  return LC_OK;
}

// one launch of the auxiliary epoch kernel: mode 4 (scene -> spectrum) or mode 3 (scene (*) spectrum -> image)
int launch_aux(lc_joint *j, int mode, const float *scene, const float2 *St_in, float2 *St_out, float *conv_out) {
  const JointVariant *v = j->v;
  JointArgs A;
  std::memset(&A, 0, sizeof(A));
  A.E = j->E;
  A.M = 0;
  A.mode = mode;
  A.St = St_in;
  A.spec = j->spec;
  A.twid = j->twid;
  A.a = j->par[LC_P_A];
  A.cx = j->par[LC_P_CX];
  A.cy = j->par[LC_P_CY];
  A.dx = j->par[LC_P_DX];
  A.dy = j->par[LC_P_DY];
  A.alpha = j->par[LC_P_ALPHA];
  A.h = j->par[LC_P_H];
  A.mean = j->par[LC_P_MEAN];
  A.tabs = j->tabs;
  A.scene_in = scene;
  A.St_out = St_out;
  A.conv_out = conv_out;
  LC_HIP(j->ctx, hipFuncSetAttribute((const void *)v->ek_aux, hipFuncAttributeMaxDynamicSharedMemorySize, v->e_lds));
  hipLaunchKernelGGL(v->ek_aux, dim3(j->E), dim3(v->e_thr), v->e_lds, j->ctx->stream, A);
  LC_HIP(j->ctx, hipGetLastError());
  return LC_OK;
}

}  // namespace

extern "C" {

int lc_joint_supported(int n, int ss) { return find_jv(n, ss) != nullptr; }
int lc_joint_set_debug_global(int on) {
  g_debug_global = on ? 1 : 0;
  return LC_OK;
}

static int joint_create_impl(lc_ctx *ctx, int E, int M, int n, int ss, const float *data, const float *sigma2,
                             const float *psf, int G, const int32_t *epochs_per_group, lc_joint **out);
int lc_joint_create(lc_ctx *ctx, int E, int M, int n, int ss, const float *data, const float *sigma2,
                    const float *psf, lc_joint **out) {
  return joint_create_impl(ctx, E, M, n, ss, data, sigma2, psf, 0, nullptr, out);
}
int lc_joint_create_groups(lc_ctx *ctx, int G, const int32_t *epochs_per_group, int M, int n, int ss, const float *data,
                           const float *sigma2, const float *psf, lc_joint **out) {
  if (!ctx || G <= 0 || !epochs_per_group) {
    if (ctx) ctx->err = "lc_joint_create_groups: invalid argument";
    return LC_ERR_INVALID;
  }
  long long E = 0;
  for (int g = 0; g < G; ++g) {
    if (epochs_per_group[g] <= 0) LC_FAIL(ctx, LC_ERR_INVALID, "lc_joint_create_groups: every star needs at least one epoch");
    E += epochs_per_group[g];
  }
  if (E > (1 << 24)) LC_FAIL(ctx, LC_ERR_INVALID, "lc_joint_create_groups: too many epochs");
  if (M <= 0) LC_FAIL(ctx, LC_ERR_INVALID, "lc_joint_create_groups: at least one point source per star");
  return joint_create_impl(ctx, (int)E, M, n, ss, data, sigma2, psf, G, epochs_per_group, out);
}
static int joint_create_impl(lc_ctx *ctx, int E, int M, int n, int ss, const float *data, const float *sigma2,
                             const float *psf, int G, const int32_t *epochs_per_group, lc_joint **out) {
  if (!ctx || !out || !data || !sigma2 || !psf || E <= 0 || M < 0) {
    if (ctx) ctx->err = "lc_joint_create: invalid argument";
    return LC_ERR_INVALID;
  }
  // grouped objects run the point-source-only kernel and nothing else: no spectra, no background work space
  const bool lean = G > 0;
  if (M > kMaxSources) LC_FAIL(ctx, LC_ERR_UNSUPPORTED, "at most 8 point sources");
  const JointVariant *v = find_jv(n, ss, E, ctx->n_cu);
  if (!v) LC_FAIL(ctx, LC_ERR_UNSUPPORTED, "no joint-fit kernel instantiated for this stamp size");
  LC_HIP(ctx, hipSetDevice(ctx->device));
  lc_joint *j = new lc_joint();
  j->ctx = ctx;
  j->v = v;
  j->E = E;
  j->M = M;
  j->n = n;
  j->ss = ss;
  j->N = n * ss;
  j->L = v->L;
  j->KH = j->L / 2 + 1;
  j->J = ilog2(j->N);
  const int N = j->N, L = j->L, KH = j->KH;
  const size_t NN = (size_t)N * N, nn = (size_t)n * n;
  int rc = 0;
#define TRY(x)             \
  if ((rc = (x)) != 0) {   \
    lc_joint_destroy(j);   \
    return rc;             \
  }
  const int GM = std::max(G, 1) * M;
  const int sizes[LC_P_COUNT] = {E * M, GM, GM, E, E, E, (int)NN, E};
  j->G = G;
  for (int k = 0; k < LC_P_COUNT; ++k) {
    j->psize[k] = sizes[k];
    TRY(dmalloc(j, &j->par[k], sizes[k]));
    TRY(dmalloc(j, &j->pm[k], sizes[k]));
    TRY(dmalloc(j, &j->ps[k], sizes[k]));
    TRY(dmalloc(j, &j->gout[k], sizes[k]));
  }
  TRY(dmalloc(j, &j->data, E * nn));
  TRY(dmalloc(j, &j->wgt, E * nn));
  if (!lean) TRY(dmalloc(j, &j->St, (size_t)E * KH * L));
  TRY(dmalloc(j, &j->twid, L));
  if (!lean) TRY(dmalloc(j, &j->tabs, (size_t)E * 4 * std::max(M, 1) * N));
  if (!lean) TRY(dmalloc(j, &j->HG, E * NN));
  TRY(dmalloc(j, &j->chi2_e, E));
  TRY(dmalloc(j, &j->g_a, E * M));
  TRY(dmalloc(j, &j->g_cx_e, E * M));
  TRY(dmalloc(j, &j->g_cy_e, E * M));
  TRY(dmalloc(j, &j->g_dx, E));
  TRY(dmalloc(j, &j->g_dy, E));
  TRY(dmalloc(j, &j->g_mean, E));
  TRY(dmalloc(j, &j->model, E * nn));
  TRY(dmalloc(j, &j->fisher, E * M));
  j->shared_count = (int)NN + 4 * M + 2;
  TRY(dmalloc(j, &j->shared, j->shared_count));
  TRY(dmalloc(j, &j->W, lean ? 1 : (size_t)(j->J + 1) * NN));
  TRY(dmalloc(j, &j->norms, j->J + 1));
  TRY(dmalloc(j, &j->qscr, lean ? 1 : (size_t)(j->J + 1) * NN));
  TRY(dmalloc(j, &j->out_loss, 4));
  TRY(dmalloc(j, &j->greg, NN));
  TRY(dmalloc(j, &j->regs, 4 + 3 * kMaxSources + 4));
  const bool cluster_ok = v->ek_cluster && !lean && E <= ctx->n_cu / 2;  // (at least two workgroups per epoch, all resident)
  if ((v->gspec || cluster_ok) && !lean) TRY(dmalloc(j, &j->spec, (size_t)E * N * ((KH + 15) / 16 * 16)));  // rows padded to 128-byte lines (JointCfg::KS)
  if ((v->gspec || cluster_ok) && !lean) TRY(dmalloc(j, &j->part, (size_t)E * kMaxParts * (4 + 3 * kMaxSources)));
  if (cluster_ok) TRY(dmalloc(j, &j->cl_ctr, (size_t)(E + 1) * kClStride));
  if (!lean) TRY(dmalloc(j, &j->chain_flags, kChainBlocks + 32));
  if (v->gspec && !lean) TRY(dmalloc(j, &j->tshift, (size_t)E * 2));
  if (lean) {
    std::vector<int> grp(E);
    j->gstart.assign(G + 1, 0);
    for (int g = 0; g < G; ++g) {
      j->gstart[g + 1] = j->gstart[g] + epochs_per_group[g];
      for (int e = j->gstart[g]; e < j->gstart[g + 1]; ++e) grp[e] = g;
    }
    TRY(dmalloc(j, &j->group_dev, E));
    TRY(h2d(j, j->group_dev, grp.data(), grp.size() * sizeof(int)));
    TRY(dmalloc(j, &j->views_dev, G));
    TRY(dmalloc(j, &j->shared_g, (size_t)G * (4 * M + 2)));
    TRY(dmalloc(j, &j->a_ref_g, (size_t)G * kMaxSources));
    TRY(dmalloc(j, &j->out_loss_g, G));
  }
  if (!lean) {  // (all sizes: the sharded drive evaluates the point-source term with these kernels at every N)
    const size_t nb = (NN + kGmThreads - 1) / kGmThreads;
    TRY(dmalloc(j, &j->gm_c, NN));
    TRY(dmalloc(j, &j->gm_t, NN));
    TRY(dmalloc(j, &j->gm_n, NN));
    TRY(dmalloc(j, &j->gm_y, NN));
    TRY(dmalloc(j, &j->gm_l1, (size_t)(j->J + 1) * nb));
    TRY(dmalloc(j, &j->gm_pos, nb));
    TRY(dmalloc(j, &j->gm_pts, 8 + nb * 3 * kMaxSources));
    TRY(dmalloc(j, &j->gm_edge, (size_t)N * 4));
  }
  j->mreg = lean ? nullptr : find_mreg(N);
  if (j->mreg) {
    const size_t nb = (NN + kGmThreads - 1) / kGmThreads;
    std::vector<float> A, AT;
    build_cumulative_operators(N, j->J, A, AT);
    TRY(dmalloc(j, &j->mr_A, A.size()));
    TRY(dmalloc(j, &j->mr_AT, AT.size()));
    TRY(h2d(j, j->mr_A, A.data(), A.size() * sizeof(float)));
    TRY(h2d(j, j->mr_AT, AT.data(), AT.size() * sizeof(float)));
    TRY(dmalloc(j, &j->mr_C, (size_t)(j->J + 2) * NN));
    TRY(dmalloc(j, &j->mr_Z, (size_t)(j->J + 2) * NN));
    TRY(dmalloc(j, &j->mr_l1b, (size_t)(j->J + 2) * nb));
    TRY(dmalloc(j, &j->mr_posb, nb));
    TRY(dmalloc(j, &j->mr_S, (size_t)(j->J + 2) * NN));
    TRY(dmalloc(j, &j->mr_T, (size_t)(j->J + 1) * NN));
    TRY(dmalloc(j, &j->mr_l1, j->J + 1));
    TRY(dmalloc(j, &j->mr_pos, nb));
    TRY(dmalloc(j, &j->mr_part, std::max<size_t>((size_t)nb * 3 * kMaxSources, (size_t)kPtsBlocks * kPtsStride)));
    TRY(dmalloc(j, &j->mr_pbar, NN));
    if (j->mreg->rows) TRY(dmalloc(j, &j->rr_Zp, (size_t)(N / kRrRows) * kRrMaxParts * NN));
    TRY(dmalloc(j, &j->pts_ctr, 4));
    TRY(dmalloc(j, &j->upd_ctr, 4));
    TRY(dmalloc(j, &j->reg_flag, 4));  // [0] completion flag, [1] a wait ran out, [2] ticket of the finishing launch
    LC_HIP(ctx, hipFuncSetAttribute((const void *)j->mreg->fwd, hipFuncAttributeMaxDynamicSharedMemorySize, j->mreg->lds_fwd));
    LC_HIP(ctx, hipFuncSetAttribute((const void *)j->mreg->adj, hipFuncAttributeMaxDynamicSharedMemorySize, j->mreg->lds_adj));

  }
  LC_HIP(ctx, hipStreamCreate(&j->streamB));
  LC_HIP(ctx, hipEventCreateWithFlags(&j->evReg, hipEventDisableTiming));
  LC_HIP(ctx, hipEventCreateWithFlags(&j->evUpd, hipEventDisableTiming));
  LC_HIP(ctx, hipEventRecord(j->evUpd, ctx->stream));
  TRY(dmalloc(j, &j->scene2, 2 * NN));
  TRY(dmalloc(j, &j->prior, 4 * std::max(M, 1)));
  TRY(dmalloc(j, &j->a_ref, kMaxSources));
  TRY(ensure_hist(j, 64));
  {
    std::vector<float> d(data, data + E * nn), w(E * nn);
    j->h_sigma2.assign(sigma2, sigma2 + E * nn);
    for (size_t i = 0; i < d.size(); ++i) {
      const float s2 = sigma2[i];
      if (!std::isfinite(d[i]) || !std::isfinite(s2) || s2 <= 0.f) {
        d[i] = 0.f;
        w[i] = 0.f;
      } else {
        w[i] = 1.0f / s2;
      }
    }
    TRY(h2d(j, j->data, d.data(), d.size() * sizeof(float)));
    TRY(h2d(j, j->wgt, w.data(), w.size() * sizeof(float)));
  }
  {
    std::vector<float2> tw(L);
    for (int k = 0; k < L; ++k) tw[k] = make_float2((float)std::cos(-2.0 * M_PI * k / L), (float)std::sin(-2.0 * M_PI * k / L));
    TRY(h2d(j, j->twid, tw.data(), tw.size() * sizeof(float2)));
    std::vector<float> norms;
    starlet_scale_norms(N, j->J, norms);
    TRY(h2d(j, j->norms, norms.data(), norms.size() * sizeof(float)));
  }
  {
    // PSF spectra: FFT2 of the zero-padded narrow PSF, stored transposed and pre-divided by L^2
    if (!lean) j->h_psf.assign(psf, psf + E * NN);
    TRY(dmalloc(j, &j->psf_dev, (size_t)E * NN));
    TRY(h2d(j, j->psf_dev, psf, (size_t)E * NN * sizeof(float)));
    if (lean) {
      ps_fn pk = nullptr;
      int plds = 0;
      find_ps_kernel(N, ss, &pk, &plds);
      if (!pk) {
        ctx->err = "lc_joint_create_groups: no point-source kernel for this stamp size";
        lc_joint_destroy(j);
        return LC_ERR_UNSUPPORTED;
      }
    } else if (!std::getenv("LCMI_SPECTRA_HOST")) {
      // on the device: the row / column FFT passes of the epoch kernel in its spectrum mode
      TRY(launch_aux(j, 4, j->psf_dev, nullptr, j->St, nullptr));
    } else {
      // host (double precision, threaded): independent cross-check
      std::vector<float2> st((size_t)E * KH * L);
      const double sc = 1.0 / ((double)L * L);
      const int T = noise_threads(E);
      std::vector<std::thread> pool;
      for (int t = 0; t < T; ++t)
        pool.emplace_back([&, t]() {
          std::vector<cd> a((size_t)L * L);
          for (int e = t; e < E; e += T) {
            std::fill(a.begin(), a.end(), cd(0, 0));
            for (int r = 0; r < N; ++r)
              for (int c = 0; c < N; ++c) a[(size_t)r * L + c] = psf[(size_t)e * NN + (size_t)r * N + c];
            host_fft2d(a, L, 0, N, false);
            float2 *se = st.data() + (size_t)e * KH * L;
            for (int k = 0; k < KH; ++k)
              for (int r = 0; r < L; ++r)
                se[(size_t)k * L + r] = make_float2((float)(a[(size_t)r * L + k].real() * sc), (float)(a[(size_t)r * L + k].imag() * sc));
          }
        });
      for (auto &th : pool) th.join();
      TRY(h2d(j, j->St, st.data(), st.size() * sizeof(float2)));
    }
  }
  for (int k = 0; k < LC_P_COUNT; ++k) j->free_mask[k] = 0;
#undef TRY
  *out = j;
  return LC_OK;
}

void lc_joint_destroy(lc_joint *j) {
  if (!j) return;
  (void)hipSetDevice(j->ctx->device);
  hipStreamSynchronize(j->ctx->stream);
  if (j->streamB) {
    hipStreamSynchronize(j->streamB);
    hipStreamDestroy(j->streamB);
  }
  if (j->streamC) {
    hipStreamSynchronize(j->streamC);
    hipStreamDestroy(j->streamC);
  }
  if (j->evPts) hipEventDestroy(j->evPts);
  for (hipStream_t st : j->gstreams) {
    hipStreamSynchronize(st);
    hipStreamDestroy(st);
  }
  for (hipEvent_t ev : j->gevents) hipEventDestroy(ev);
  if (j->evReg) hipEventDestroy(j->evReg);
  if (j->evUpd) hipEventDestroy(j->evUpd);
  for (void *p : j->allocs) hipFree(p);
  if (j->hist) hipFree(j->hist);
  if (j->ghist) hipFree(j->ghist);
  if (j->phist) hipFree(j->phist);
  delete j;
}

int lc_joint_set_param(lc_joint *j, int which, const float *values, int count) {
  if (!j || which < 0 || which >= LC_P_COUNT || !values) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (count != j->psize[which]) LC_FAIL(j->ctx, LC_ERR_INVALID, "lc_joint_set_param: wrong element count");
  if (which == LC_P_ALPHA) {
    j->any_rotation = false;
    for (int i = 0; i < count; ++i)
      if (values[i] != 0.f) j->any_rotation = true;
  }
  if (which == LC_P_H) {
    j->h_nonzero = false;
    for (int i = 0; i < count; ++i)
      if (values[i] != 0.f) {
        j->h_nonzero = true;
        break;
      }
  }
  if (j->G > 0 && which == LC_P_H && j->h_nonzero) {
    j->h_nonzero = false;  // (nothing was stored)
    LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "batched star photometry has no background: h must stay zero");
  }
  if (j->G > 0 && which == LC_P_A) {  // one reference per star and source: the mean over that star's epochs
    std::vector<float> ref((size_t)j->G * kMaxSources, 0.f);
    for (int g = 0; g < j->G; ++g)
      for (int i = 0; i < j->M; ++i) {
        double acc = 0.0;
        for (int e = j->gstart[g]; e < j->gstart[g + 1]; ++e) acc += values[(size_t)e * j->M + i];
        ref[(size_t)g * kMaxSources + i] = (float)(acc / (j->gstart[g + 1] - j->gstart[g]));
      }
    int rc = h2d(j, j->a_ref_g, ref.data(), ref.size() * sizeof(float));
    if (rc) return rc;
  } else if (which == LC_P_A && j->M > 0) {
    // reference fluxes of the centred flux moments: the mean of the values set here (a sharded fit overrides them
    // with one reference for all ranks, lc_joint_set_flux_reference)
    float ref[kMaxSources] = {};
    for (int i = 0; i < j->M; ++i) {
      double acc = 0.0;
      for (int e = 0; e < j->E; ++e) acc += values[(size_t)e * j->M + i];
      ref[i] = (float)(acc / j->E);
    }
    int rc = h2d(j, j->a_ref, ref, sizeof(ref));
    if (rc) return rc;
  }
  return h2d(j, j->par[which], values, (size_t)count * sizeof(float));
}
int lc_joint_set_flux_reference(lc_joint *j, const float *ref, int count) {
  if (!j || !ref) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_set_flux_reference: a batched object keeps one reference per star (set with the fluxes)");
  if (count != j->M) LC_FAIL(j->ctx, LC_ERR_INVALID, "lc_joint_set_flux_reference: one reference flux per point source");
  float r[kMaxSources] = {};
  for (int i = 0; i < count; ++i) r[i] = ref[i];
  return h2d(j, j->a_ref, r, sizeof(r));
}
int lc_joint_get_flux_reference(lc_joint *j, float *ref, int count) {
  if (!j || !ref) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_get_flux_reference: a batched object keeps one reference per star");
  if (count != j->M) LC_FAIL(j->ctx, LC_ERR_INVALID, "lc_joint_get_flux_reference: one reference flux per point source");
  return d2h(j, ref, j->a_ref, (size_t)count * sizeof(float));
}
int lc_joint_get_param(lc_joint *j, int which, float *values, int count) {
  if (!j || which < 0 || which >= LC_P_COUNT || !values) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (count != j->psize[which]) LC_FAIL(j->ctx, LC_ERR_INVALID, "lc_joint_get_param: wrong element count");
  return d2h(j, values, j->par[which], (size_t)count * sizeof(float));
}
int lc_joint_set_free(lc_joint *j, const int32_t *free_mask) {
  if (!j || !free_mask) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (free_mask[LC_P_ALPHA]) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "alpha is never optimised (roi_modelling.py:221-222)");
  if (j->G > 0 && free_mask[LC_P_H]) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "batched star photometry has no background grid");
  for (int k = 0; k < LC_P_COUNT; ++k) j->free_mask[k] = free_mask[k] ? 1 : 0;
  lc_joint_param_history_end(j);  // (its row layout follows the free blocks)
  // a new optimisation starts: reset the moments and the iteration counter
  for (int k = 0; k < LC_P_COUNT; ++k) {
    LC_HIP(j->ctx, hipMemsetAsync(j->pm[k], 0, (size_t)std::max(j->psize[k], 1) * sizeof(float), j->ctx->stream));
    LC_HIP(j->ctx, hipMemsetAsync(j->ps[k], 0, (size_t)std::max(j->psize[k], 1) * sizeof(float), j->ctx->stream));
  }
  j->iters_done = 0;
  return LC_OK;
}
int lc_joint_set_loss(lc_joint *j, const lc_joint_loss_cfg *cfg, const float *W) {
  if (!j || !cfg) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0 && (cfg->lam_pts_source != 0.f || cfg->n_prior > 0 || W))
    LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "batched star photometry: no point-source starlet term, prior or weight cube");
  j->cfg = *cfg;
  j->n_prior = 0;
  if (cfg->n_prior > 0) {
    if (!cfg->prior_cx_mean || !cfg->prior_cx_sigma || !cfg->prior_cy_mean || !cfg->prior_cy_sigma || cfg->n_prior != j->M)
      LC_FAIL(j->ctx, LC_ERR_INVALID, "prior arrays must have M entries");
    std::vector<float> p(4 * j->M);
    for (int i = 0; i < j->M; ++i) {
      p[i] = cfg->prior_cx_mean[i];
      p[j->M + i] = cfg->prior_cx_sigma[i];
      p[2 * j->M + i] = cfg->prior_cy_mean[i];
      p[3 * j->M + i] = cfg->prior_cy_sigma[i];
    }
    int rc = h2d(j, j->prior, p.data(), p.size() * sizeof(float));
    if (rc) return rc;
    j->n_prior = j->M;
  }
  j->cfg.prior_cx_mean = j->cfg.prior_cx_sigma = j->cfg.prior_cy_mean = j->cfg.prior_cy_sigma = nullptr;
  if (W) {
    int rc = h2d(j, j->W, W, (size_t)j->J * j->N * j->N * sizeof(float));
    if (rc) return rc;
    j->have_W = true;
  } else {
    j->have_W = false;
  }
  return LC_OK;
}

namespace {
int propagate_noise_device(lc_joint *j) {
  const int N = j->N, ss = j->ss, E = j->E, J = j->J, c = (N - 1) / 2, shift = ss * (j->n / 2) - c;
  const size_t NN = (size_t)N * N, ENN = (size_t)E * NN;
  int rc;
  if (!j->nz_a) {
    if ((rc = dmalloc(j, &j->nz_a, ENN)) || (rc = dmalloc(j, &j->nz_b, ENN)) || (rc = dmalloc(j, &j->nz_c, ENN)) ||
        (rc = dmalloc(j, &j->nz_up, ENN)) || (rc = dmalloc(j, &j->nz_scene, ENN)) ||
        (rc = dmalloc(j, &j->St_alt, (size_t)E * j->KH * j->L)))
      return rc;
  }
  hipStream_t q = j->ctx->stream;
  const dim3 grid((unsigned)((NN + kNzThreads - 1) / kNzThreads), (unsigned)E), block(kNzThreads);
  hipLaunchKernelGGL(nz_up0_kernel, grid, block, 0, q, N, ss, j->wgt, j->nz_up);
  hipLaunchKernelGGL(nz_response_kernel, grid, block, 0, q, N, ss, j->psf_dev, j->nz_a);
  float *cur = j->nz_a, *nxt = j->nz_c;
  for (int s = 0; s <= J; ++s) {
    if (s < J) {
      hipLaunchKernelGGL(nz_pass_kernel, grid, block, 0, q, N, 1 << s, 1, cur, j->nz_b);
      hipLaunchKernelGGL(nz_pass_kernel, grid, block, 0, q, N, 1 << s, 0, j->nz_b, nxt);
    }
    hipLaunchKernelGGL(nz_kappa2_kernel, grid, block, 0, q, N, shift, cur, s < J ? nxt : (const float *)nullptr, j->nz_scene);
    LC_HIP(j->ctx, hipGetLastError());
    if ((rc = launch_aux(j, 4, j->nz_scene, nullptr, j->St_alt, nullptr))) return rc;       // spectra of kappa^2
    if ((rc = launch_aux(j, 3, j->nz_up, j->St_alt, nullptr, j->HG))) return rc;            // up0(w_e) (*) kappa_e^2
    if ((rc = launch_reduce(j, 1))) return rc;                                              // sum over the epochs
    hipLaunchKernelGGL(nz_sqrt_kernel, dim3((unsigned)((NN + kNzThreads - 1) / kNzThreads)), block, 0, q, (int)NN, j->shared,
                       j->W + (size_t)s * NN);
    LC_HIP(j->ctx, hipGetLastError());
    std::swap(cur, nxt);
  }
  return LC_OK;
}
}  // namespace

int lc_joint_propagate_noise(lc_joint *j, float *W_out) {
  if (!j) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_propagate_noise: not available on a batched star-photometry object");
  const int N = j->N, n = j->n, ss = j->ss, E = j->E, c = (N - 1) / 2;
  const size_t NN = (size_t)N * N, nn = (size_t)n * n;
  if (!std::getenv("LCMI_NOISE_HOST")) {
    int rc = propagate_noise_device(j);
    if (rc) return rc;
    j->have_W = true;
    if (W_out && (rc = d2h(j, W_out, j->W, (size_t)(j->J + 1) * NN * sizeof(float)))) return rc;
    return LC_OK;
  }
  // host path (double-precision FFT convolutions, threaded), kept as an independent cross-check: LCMI_NOISE_HOST=1
  // contributor = epoch; response of dL/dh to a unit of whitened noise in data pixel p* = (n/2, n/2):
  //   r_e[u'][v'] = sum_{(u,v) in block(p*)} s_e[u - u' + c][v - v' + c]      (adjoint of D_ss . conv_same(., s_e))
  std::vector<float> w((size_t)E * nn);
  for (size_t i = 0; i < w.size(); ++i) {
    const float sg = j->h_sigma2[i];
    w[i] = (std::isfinite(sg) && sg > 0.f) ? 1.0f / sg : 0.f;
  }
  const int T = noise_threads(E);
  std::vector<NoiseAccumulator> accs(T, NoiseAccumulator(N, ss));
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)
    pool.emplace_back([&, t]() {
      std::vector<double> r(NN);
      const int b0 = ss * (n / 2);
      for (int e = t; e < E; e += T) {
        const float *s = &j->h_psf[(size_t)e * NN];
        for (int up = 0; up < N; ++up)
          for (int vp = 0; vp < N; ++vp) {
            double acc = 0;
            for (int du = 0; du < ss; ++du)
              for (int dv = 0; dv < ss; ++dv) {
                const int a = b0 + du - up + c, b = b0 + dv - vp + c;
                if (a >= 0 && a < N && b >= 0 && b < N) acc += s[(size_t)a * N + b];
              }
            r[(size_t)up * N + vp] = acc;
          }
        accs[t].add(r, &w[(size_t)e * nn]);
      }
    });
  for (auto &th : pool) th.join();
  for (int t = 1; t < T; ++t) accs[0].merge(accs[t]);
  std::vector<float> W((size_t)(j->J + 1) * NN);
  accs[0].finalize(W.data());
  int rc = h2d(j, j->W, W.data(), W.size() * sizeof(float));
  if (rc) return rc;
  j->have_W = true;
  if (W_out) std::memcpy(W_out, W.data(), W.size() * sizeof(float));
  return LC_OK;
}

int lc_joint_step_local(lc_joint *j) {
  if (!j) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_step_local: not available on a batched star-photometry object");
  j->reg_pending = false;
  // inside lc_joint_run_adabelief (one GPU: the mean fluxes are all local) the point-source starlet term, which depends
  // on the current a, c_x, c_y only, is evaluated with the background regulariser on the second stream
  j->pts_pending = j->in_device_loop && j->cfg.lam_pts_source != 0.f && j->M > 0;
  j->reg_planes = j->reg_noflag = j->reg_counter = j->defer_event = false;
  j->epoch_wait_due = j->epoch_waited = false;
  {
    // will the fused reduction + update consume this iteration's chain (the conditions of fuse_full / fuse_stencil below, as far
    // as they are known before the epoch launch)?  Then the chain leaves its planes for that kernel to add; a wrong guess is
    // repaired in lc_joint_step_update (one more launch), never wrong numbers
    const bool gm_upd = !j->v->uk || (j->cfg.lam_pts_source == 0.f || j->pts_pending);
    const bool stencil = j->v->gspec && j->spec && j->tshift && !j->any_rotation && (j->N * j->N) % kStPix == 0 && j->N % 4 == 0 &&
                         std::getenv("LCMI_STENCIL_REDUCE");
    j->planes_pred = j->in_device_loop && j->free_mask[LC_P_H] && gm_upd && !std::getenv("LCMI_SPLIT_UPDATE") && !stencil;
  }
  if (reg_h_on(j) || j->pts_pending) {
    // starlet l1 + positivity of h depend on h alone: evaluate them on a second stream while the epoch
    // kernel (which leaves CUs idle whenever E < 256) runs; the update kernel joins the two
    // the chain reads what the previous update wrote: behind that update's completion counter (a one-wave gate kernel at the
    // head of this stream) when the update counted itself in, behind its event otherwise
    if (j->upd_gate_pending && !upd_gate_ok(j, j->ctx->stream)) {
      // (no event was recorded behind the last update because no chain was expected: record it now, behind that update)
      j->upd_gate_pending = false;
      LC_HIP(j->ctx, hipEventRecord(j->evUpd, j->ctx->stream));
    }
    if (j->upd_gate_pending) {
      j->upd_seq += 1;
      hipLaunchKernelGGL(mreg_gate_kernel, dim3(1), dim3(64), 0, j->streamB, j->upd_ctr, j->upd_seq, j->upd_ctr + 1);
      j->upd_gate_used = true;
      j->upd_signal_due = true;   // (the epoch launch below raises the word)
    } else {
      LC_HIP(j->ctx, hipStreamWaitEvent(j->streamB, j->evUpd, 0));
    }
    if (j->tl_events) (void)hipEventRecord(j->tl_events[1], j->streamB);
    int rc = launch_update(j, 0, 0, nullptr, false, false, 1, j->streamB);
    if (rc) return rc;
    if (j->tl_events) (void)hipEventRecord(j->tl_events[2], j->streamB);
    j->evreg_recorded = false;   // (recorded by the first consumer that waits for it: wait_for_chain_event)
    j->reg_pending = true;
    // Inside the library's loops, behind a chain whose last launch counts itself into reg_flag: the epoch launch carries the
    // wait for the chain (JointArgs::chain_flag) - where the chain is expected to be done well before the epochs are.  Measured
    // (us per iteration, with / without): C4 66.9 / 68.7, C5 shard 223.3 / 225.6, 1000 x 128 x 128 1541 / 1548; NOT beside the
    // cluster form, whose chain ends about when the epoch kernel does (the update's own poll overlaps its slab loads with
    // the chain's tail: 25 epochs 53.1 / 52.8), and not for few epochs of 128 x 128 (32 epochs: 114.5 / 112.2).
    j->epoch_wait_due = (j->in_device_loop || j->in_sharded_loop) && (j->reg_planes || j->reg_counter) && j->reg_flag &&
                        cluster_parts(j) == 0 && (j->N <= 128 || j->E >= 64) && !j->force_events &&
                        !std::getenv("LCMI_EVENT_SYNC") && !std::getenv("LCMI_EPOCH_WAIT_OFF");
  }
  const bool gm_update = !j->v->uk || ((j->cfg.lam_pts_source == 0.f || j->pts_pending) && (j->reg_pending || !reg_h_on(j)));
  // global-spectrum kernels, every epoch a translation, reduction and update fused: the reduction can apply T_e^T itself
  // (joint_stencil_update_kernel) - no phase D, no slabs (64 MB less traffic per iteration of the C5 shard).  Built, tested
  // (tests/test_joint_paths_gpu.py) and NOT the default: measured on MI355X it is no faster - C5 shard 261.7 against 260.6
  // us per iteration, 200 epochs 389.9 / 381.9, 1000 epochs 1621 / 1603 (blocks of 64 pixels with 4-byte taps: 260.8 /
  // 385.5 / 1617): the strided gather over the epochs' scratch costs what phase D and the slabs cost.  LCMI_STENCIL_REDUCE=1
  // selects it.
  j->fuse_stencil = j->in_device_loop && j->free_mask[LC_P_H] && gm_update && j->v->gspec && j->spec && j->tshift && !j->any_rotation &&
                    (j->N * j->N) % kStPix == 0 && j->N % 4 == 0 && !std::getenv("LCMI_SPLIT_UPDATE") && std::getenv("LCMI_STENCIL_REDUCE");
  int need = launch_epochs(j, 0, 0, false, nullptr);
  if (need < 0) return need;
  // inside lc_joint_run_adabelief, with the background fixed, only scalars are reduced: the multi-block update
  // kernel does that itself (one launch less per iteration)
  j->fuse_pending = j->in_device_loop && need == 0 && !j->free_mask[LC_P_H] && gm_update;
  // with the background free the same holds for the image part: reduction and update share one launch
  j->fuse_full = j->in_device_loop && need == 1 && j->free_mask[LC_P_H] && gm_update && !std::getenv("LCMI_SPLIT_UPDATE");
  if (j->fuse_pending || j->fuse_full) return LC_OK;
  return launch_reduce(j, need);
}
int lc_joint_shared_buffer_dev(lc_joint *j, void **dev_ptr, int *count) {
  if (!j || !dev_ptr || !count) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  *dev_ptr = j->shared;
  *count = j->shared_count;
  return LC_OK;
}
int lc_joint_shared_get(lc_joint *j, float *host, int count) {
  if (!j || !host || count != j->shared_count) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  return d2h(j, host, j->shared, (size_t)count * sizeof(float));
}
int lc_joint_shared_set(lc_joint *j, const float *host, int count) {
  if (!j || !host || count != j->shared_count) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  return h2d(j, j->shared, host, (size_t)count * sizeof(float));
}
// The chain of this iteration left planes for the fused reduction + update (planes_pred) and another consumer turned up: wait
// for the chain and let one more launch write greg / regs on the consumer's stream.
static int planes_repair(lc_joint *j, bool consumer_adds_planes) {
  if (!j->reg_planes || consumer_adds_planes) return LC_OK;
  const int NN = j->N * j->N, nb = (NN + kGmThreads - 1) / kGmThreads;
  {
    int rc = wait_for_chain_event(j, j->ctx->stream);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(mreg_finish3_kernel, dim3(nb + 1), dim3(kGmThreads), 0, j->ctx->stream, NN, nb, j->M, j->planes, j->greg, j->regs,
                     (unsigned int *)nullptr);
  LC_HIP(j->ctx, hipGetLastError());
  j->reg_planes = false;
  j->reg_noflag = true;
  return LC_OK;
}
int lc_joint_step_update(lc_joint *j, const lc_adabelief_cfg *cfg) {
  if (!j) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_step_update: not available on a batched star-photometry object");
  int rc = ensure_hist(j, j->iters_done + 2);
  if (rc) return rc;
  // the regulariser chain of this iteration was enqueued (second stream) before the epoch kernel.  The fused update checks
  // its completion flag in the kernel; every other form waits for the event
  // (only while the update's blocks cannot fill the machine: a block that had to wait for a late chain holds its wave
  //  slots and registers, and the chain needs some of its own to finish - N = 256 keeps the event)
  // two 16-pixel tiles per block: at most two resident 256-thread blocks per CU (8 of 32 wave slots, 192 of 512 registers
  // per lane and SIMD), next to which every kernel of the chain fits
  const bool few_blocks = ((j->N * j->N) / kRedPix) % 2 == 0 && (j->N * j->N) / kRedPix / 2 <= 2 * j->ctx->n_cu;
  if ((rc = planes_repair(j, j->fuse_full && !j->fuse_stencil))) return rc;
  // (LCMI_FLAG_SYNC_ALL=1: the flag form at 256 x 256 too, where the update has more blocks than fit the machine at once.
  //  Measured: a chain held back by 600 us is still scheduled beside the waiting blocks and the numbers are those of the event
  //  form, but the gain is small - C5 shard 222.6 -> 221.8 us, 32 epochs 111.5 -> 108.6 - and how the dispatcher treats a second
  //  queue while one kernel has hundreds of blocks pending is nothing this code can guarantee: opt-in.)
  const bool all_sizes = std::getenv("LCMI_FLAG_SYNC_ALL") != nullptr;
  j->flag_sync = j->reg_pending && !j->epoch_waited && j->fuse_full && j->mreg && j->reg_flag && (few_blocks || all_sizes) && !j->reg_noflag && !j->force_events &&
                 !std::getenv("LCMI_EVENT_SYNC");
  // (sharded drive behind a counting chain: launch_update decides - the multi-block update polls the counter in its kernel,
  //  every other consumer gets the event wait there)
  if (j->epoch_waited) j->gm_flag_used = true;   // (its error word is checked at the end of the loop)
  j->defer_event = j->reg_pending && !j->epoch_waited && !j->flag_sync && j->in_sharded_loop && j->reg_counter && !j->force_events &&
                   !std::getenv("LCMI_EVENT_SYNC");
  // (epoch_waited: the epoch launch of this iteration ended only when the chain had - nothing to wait for here)
  if (j->reg_pending && !j->epoch_waited && !j->flag_sync && !j->defer_event && (rc = wait_for_chain_event(j, j->ctx->stream))) return rc;
  rc = launch_update(j, 1, j->iters_done, cfg, true, false, j->reg_pending ? 2 : 0);
  j->defer_event = false;
  if (rc) return rc;
  // (inside the library's loops no event: the record would sit between this update and the next epoch kernel on this stream,
  //  4.7 - 6 us of every iteration; the next chain starts behind a gate kernel that the next epoch launch opens)
  // ... and no event at all where no chain will follow: a fit without background regulariser and point-source starlet term
  // (the reference's default star photometry) has nobody waiting for it, and the record cost 4 of its 18.8 us per iteration
  {
    const bool in_loop = (j->in_device_loop || j->in_sharded_loop) && !std::getenv("LCMI_UPD_EVENT");
    const bool will_chain = reg_h_on(j) || (j->in_device_loop && j->cfg.lam_pts_source != 0.f && j->M > 0);
    j->upd_gate_pending = will_chain ? upd_gate_ok(j, j->ctx->stream) : in_loop;
  }
  if (!j->upd_gate_pending) LC_HIP(j->ctx, hipEventRecord(j->evUpd, j->ctx->stream));
  j->reg_pending = false;
  j->fuse_pending = false;
  j->fuse_full = false;
  j->fuse_stencil = false;
  j->pts_pending = false;
  j->iters_done += 1;
  return LC_OK;
}

static int chain_check(lc_joint *j);
int lc_joint_step_grad(lc_joint *j, float *loss, float *const grads[LC_P_COUNT]) {
  if (!j) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_step_grad: not available on a batched star-photometry object");
  int rc = planes_repair(j, false);
  if (rc) return rc;
  if (j->reg_pending && (rc = wait_for_chain_event(j, j->ctx->stream))) return rc;
  rc = launch_update(j, 0, 0, nullptr, false, true, j->reg_pending ? 2 : 0);
  if (rc) return rc;
  LC_HIP(j->ctx, hipEventRecord(j->evUpd, j->ctx->stream));  // (the next lc_joint_step_local's chain waits for this one)
  j->reg_pending = false;
  j->fuse_pending = false;
  j->fuse_full = false;
  j->fuse_stencil = false;
  j->pts_pending = false;
  if (loss && (rc = d2h(j, loss, j->out_loss, sizeof(float)))) return rc;
  if (grads)
    for (int k = 0; k < LC_P_COUNT; ++k)
      if (grads[k] && k != LC_P_ALPHA && (rc = d2h(j, grads[k], j->gout[k], (size_t)j->psize[k] * sizeof(float)))) return rc;
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  return chain_check(j);
}

// Cluster launches (joint_kernels.h, PHASE = 7): did a barrier wait of the launches since the last check run out?  Reads the
// abort word (synchronises the stream); if it is set, clears it together with the arrival counters and takes the cluster
// form out of use for this object.  *aborted = 1 then: the numbers those launches produced are not to be used.
static int cluster_check(lc_joint *j, int *aborted) {
  *aborted = 0;
  if (!j->cl_ctr || j->cluster_off) return LC_OK;
  unsigned int word = 0;
  int rc = d2h(j, &word, j->cl_ctr + (size_t)j->E * kClStride, sizeof(word));
  if (rc) return rc;
  if (!word) return LC_OK;
  LC_HIP(j->ctx, hipMemsetAsync(j->cl_ctr, 0, (size_t)(j->E + 1) * kClStride * sizeof(unsigned int), j->ctx->stream));
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  j->cl_base = 0;
  j->cluster_off = true;
  j->cl_fallbacks += 1;
  *aborted = 1;
  return LC_OK;
}

// One-launch regulariser chains since the last check: did a sync of one give up (its workgroups were not resident together)?
// Then its outputs were never completed: the numbers of the run are invalid; the object keeps the launch form from here on.
static int chain_check(lc_joint *j) {
  if (!j->chain_flags || !j->chain_used) return LC_OK;
  j->chain_used = false;
  unsigned int word = 0;
  int rc = d2h(j, &word, j->chain_flags + kChainBlocks, sizeof(word));
  if (rc || !word) return rc;
  LC_HIP(j->ctx, hipStreamSynchronize(j->streamB));
  LC_HIP(j->ctx, hipMemsetAsync(j->chain_flags, 0, (kChainBlocks + 32) * sizeof(unsigned int), j->ctx->stream));
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  j->chain_base = 0;
  j->chain_off = true;
  LC_FAIL(j->ctx, LC_ERR_DEVICE, "joint fit: the one-launch regulariser chain gave up (its workgroups were not resident together); this run's numbers are invalid - the object has switched to the launch form, run again (LCMI_REG_CHAIN=0 selects it from the start)");
}

// Update launches that carried the point-source blocks since the last check: did block 0 give up waiting for them?
// End of a library loop: later calls order themselves behind the last update through its event again; a gate that gave up
// waiting means the chain of some iteration may have read a half-written background.
static int upd_gate_finish(lc_joint *j) {
  if (j->upd_gate_pending) {
    j->upd_gate_pending = false;
    LC_HIP(j->ctx, hipEventRecord(j->evUpd, j->ctx->stream));
  }
  if (!j->upd_gate_used || !j->upd_ctr) return LC_OK;
  j->upd_gate_used = false;
  LC_HIP(j->ctx, hipStreamSynchronize(j->streamB));
  unsigned int err = 0;
  int rc = d2h(j, &err, j->upd_ctr + 1, sizeof(err));
  if (rc || !err) return rc;
  const unsigned int zero = 0;
  (void)h2d(j, j->upd_ctr + 1, &zero, sizeof(zero));
  LC_FAIL(j->ctx, LC_ERR_DEVICE, "joint fit: the regulariser chain of an iteration gave up waiting for the previous update; this run's numbers are invalid (LCMI_UPD_EVENT=1 selects the event)");
}
static int pts_tail_check(lc_joint *j) {
  if (j->gm_flag_used && j->reg_flag) {  // updates that checked the chain's completion counter themselves: did a wait run out?
    j->gm_flag_used = false;
    unsigned int err = 0;
    int rc = d2h(j, &err, j->reg_flag + 1, sizeof(err));
    if (rc) return rc;
    if (err) {
      const unsigned int zero = 0;
      (void)h2d(j, j->reg_flag + 1, &zero, sizeof(zero));
      LC_FAIL(j->ctx, LC_ERR_DEVICE, "joint fit: the regulariser of an iteration did not complete in time on the second stream (sharded loop; set LCMI_EVENT_SYNC=1)");
    }
  }
  if (!j->pts_ctr || !j->pts_tail_used) return LC_OK;
  j->pts_tail_used = false;
  unsigned int err = 0;
  int rc = d2h(j, &err, j->pts_ctr + 1, sizeof(err));
  if (rc || !err) return rc;
  const unsigned int zero = 0;
  (void)h2d(j, j->pts_ctr + 1, &zero, sizeof(zero));
  LC_FAIL(j->ctx, LC_ERR_DEVICE, "joint fit: the point-source blocks of an update launch did not arrive in time; this run's numbers are invalid (LCMI_PTS_TAIL=0 selects the separate launches)");
}

int lc_joint_run_sharded(lc_joint *j, int n_iter, const lc_adabelief_cfg *cfg, lc_allreduce_fn allreduce, void *user) {
  if (!j || n_iter < 0 || !allreduce) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_run_sharded: not available on a batched star-photometry object");
  int rc = ensure_hist(j, j->iters_done + n_iter + 2);
  const bool may_cluster = j->cl_ctr && !j->cluster_off;
  j->in_sharded_loop = true;
  if (!rc) rc = probe_streams(j);
  // The library's own peer group as the transport: reduction over the epochs and exchange in one launch (LCMI_PEER_FUSED=1).
  // Only where every block of that launch is resident at once (its blocks wait for the peers' flags after publishing; see
  // joint_reduce_peer.h), the block is whole chunks of pixels plus the scalars, and the group is this context's.
  // OPT-IN: measured at world size 1 on MI355X it saves 0.5 us per iteration of a 25-epoch shard (69.8 against 70.3), 2.0 us
  // at 200 epochs (91.7 / 93.7), nothing at 125 x 128 x 128 (254.4 / 254.1) - the launch it removes is paid back by the longer
  // kernel - and with peers all of its N^2 / 64 + 1 blocks poll the peers' flags over xGMI (17 - 65 pollers in the two-launch
  // form), which a one-GPU box cannot price.  Same bits either way (tests/test_distributed_gpu.py runs both).
  j->peer_fuse = nullptr;
  if (allreduce == (lc_allreduce_fn)lc_peer_allreduce && user) {
    lc_peer_group *g = (lc_peer_group *)user;
    const int NN = j->N * j->N;
    const char *pf = std::getenv("LCMI_PEER_FUSED");
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, joint_reduce_peer_kernel, kRedThreads, 0) != hipSuccess) {
      (void)hipGetLastError();
      per_cu = 0;
    }
    if (g->ctx == j->ctx && g->count == j->shared_count && NN % lc_peer::kChunk == 0 && 4 * j->M + 2 <= lc_peer::kChunk &&
        NN / (kRedPix * kRpTiles) + 1 <= per_cu * j->ctx->n_cu && pf && std::atoi(pf) != 0)
      j->peer_fuse = g;
  }
  for (int it = 0; it < n_iter && !rc; ++it) {
    j->peer_fused_done = false;
    if ((rc = lc_joint_step_local(j))) break;
    if (j->peer_fused_done) {
    } else if (allreduce(user, j->shared, j->shared_count, (void *)j->ctx->stream)) {
      j->ctx->err = "lc_joint_run_sharded: the all-reduce callback failed";
      rc = LC_ERR_DEVICE;
      break;
    }
    rc = lc_joint_step_update(j, cfg);
  }
  j->in_sharded_loop = false;
  j->peer_fuse = nullptr;
  {
    const int rg = upd_gate_finish(j);
    if (!rc) rc = rg;
  }
  if (!rc) rc = chain_check(j);
  if (!rc) rc = pts_tail_check(j);
  if (!rc && may_cluster && j->cl_parts_last > 0) {
    // a sharded run cannot be redone by one rank alone (the others have moved on through the same all-reduces): report it;
    // the object runs the one-workgroup kernel from here on
    int aborted = 0;
    if ((rc = cluster_check(j, &aborted))) return rc;
    if (aborted) LC_FAIL(j->ctx, LC_ERR_DEVICE, "lc_joint_run_sharded: the workgroups of an epoch were not resident together (cluster launch gave up); this run's numbers are invalid - the object has switched to the one-workgroup kernel, run again");
  }
  return rc;
}

int lc_joint_loss_grad(lc_joint *j, float *loss, float *const grads[LC_P_COUNT]) {
  if (!j) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_loss_grad: not available on a batched star-photometry object");
  const bool want_h = grads && grads[LC_P_H];
  int need = launch_epochs(j, 0, 0, want_h, nullptr);
  if (need < 0) return need;
  int rc = launch_reduce(j, need);
  if (rc) return rc;
  rc = launch_update(j, 0, 0, nullptr, false, true);
  if (rc) return rc;
  if (loss && (rc = d2h(j, loss, j->out_loss, sizeof(float)))) return rc;
  if (grads)
    for (int k = 0; k < LC_P_COUNT; ++k)
      if (grads[k] && k != LC_P_ALPHA && (rc = d2h(j, grads[k], j->gout[k], (size_t)j->psize[k] * sizeof(float)))) return rc;
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  return LC_OK;
}

int lc_joint_model(lc_joint *j, float *model, float *chi2_per_epoch) {
  if (!j) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  int rc = launch_epochs(j, 1, 0, false, j->model);
  if (rc < 0) return rc;
  if (model && (rc = d2h(j, model, j->model, (size_t)j->E * j->n * j->n * sizeof(float)))) return rc;
  if (chi2_per_epoch && (rc = d2h(j, chi2_per_epoch, j->chi2_e, (size_t)j->E * sizeof(float)))) return rc;
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  return LC_OK;
}

int lc_joint_deconvolved(lc_joint *j, int epoch, float *scene, float *background) {
  if (!j || epoch < 0 || epoch >= j->E) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  const size_t NN = (size_t)j->N * j->N;
  // (a batched star-photometry object: `epoch` counts over the epochs of all stars; the positions are those of its star)
  size_t coff = 0;
  if (j->G > 0) {
    int g = 0;
    while (g + 1 < j->G && epoch >= j->gstart[g + 1]) ++g;
    coff = (size_t)g * j->M;
  }
  hipLaunchKernelGGL(joint_scene_kernel, dim3(64), dim3(256), 0, j->ctx->stream, j->N, j->ss, j->M, epoch, j->par[LC_P_A],
                     j->par[LC_P_CX] + coff, j->par[LC_P_CY] + coff, j->par[LC_P_DX], j->par[LC_P_DY], j->par[LC_P_ALPHA],
                     j->par[LC_P_H], j->scene2, j->scene2 + NN);
  LC_HIP(j->ctx, hipGetLastError());
  int rc;
  if (scene && (rc = d2h(j, scene, j->scene2, NN * sizeof(float)))) return rc;
  if (background && (rc = d2h(j, background, j->scene2 + NN, NN * sizeof(float)))) return rc;
  return LC_OK;
}

// Every free parameter belongs to one epoch and nothing couples the epochs (fluxes, shifts, sky levels free; the shared
// positions fixed, no background, no flux-uniformity / point-source / prior term - e.g. photometry at known positions;
// the reference's default star photometry frees c_x, c_y and does not qualify): the loop runs inside ONE launch of the
// point-source-only kernel (csrc/joint_ps.h, PERSIST), then the per-epoch losses are summed.
static int run_adabelief_persistent(lc_joint *j, int n_iter, const lc_adabelief_cfg *cfg, bool *done) {
  *done = false;
  const bool coupled = j->free_mask[LC_P_H] || j->h_nonzero || j->free_mask[LC_P_CX] || j->free_mask[LC_P_CY] ||
                       j->free_mask[LC_P_ALPHA] || j->cfg.lam_flux_uniformity != 0.f || j->cfg.lam_pts_source != 0.f ||
                       j->n_prior > 0;
  const bool any_free = j->free_mask[LC_P_A] || j->free_mask[LC_P_DX] || j->free_mask[LC_P_DY] || j->free_mask[LC_P_MEAN];
  if (coupled || !any_free || j->M <= 0 || j->M > kMaxSources || !j->psf_dev || std::getenv("LCMI_JOINT_FFT_ONLY") ||
      std::getenv("LCMI_PS_LOOP") || (j->phist && j->phist_rows + n_iter > j->phist_cap))
    return LC_OK;
  ps_fn pk = nullptr;
  int plds = 0;
  find_ps_kernel(j->N, j->ss, &pk, &plds, true);
  if (!pk) return LC_OK;
  int rc;
  if (!j->psF && (rc = dmalloc(j, &j->psF, (size_t)j->E * j->M * 3 * j->n * j->n))) return rc;
  hipStream_t q = j->ctx->stream;
  lc_adabelief_cfg ab;
  if (cfg) ab = *cfg; else lc_adabelief_defaults(&ab);
  std::vector<float> sched((size_t)n_iter * 3);
  for (int t = 0; t < n_iter; ++t) adabelief_schedule(ab, j->iters_done + t, sched[3 * t], sched[3 * t + 1], sched[3 * t + 2]);
  float *d_sched = nullptr, *d_hist_e = nullptr;
  LC_HIP(j->ctx, hipMalloc((void **)&d_sched, sched.size() * sizeof(float)));
  struct DevGuard {
    void *p;
    ~DevGuard() { (void)hipFree(p); }
  } g1{d_sched};
  LC_HIP(j->ctx, hipMalloc((void **)&d_hist_e, (size_t)j->E * n_iter * sizeof(float)));
  DevGuard g2{d_hist_e};
  LC_HIP(j->ctx, hipMemcpyAsync(d_sched, sched.data(), sched.size() * sizeof(float), hipMemcpyHostToDevice, q));
  JointPsArgs P;
  std::memset(&P, 0, sizeof(P));  // (fields a launch does not use are passed as zeros, not as stack contents)
  std::memset(&P, 0, sizeof(P));
  JointArgs &A = P.J;
  A.E = j->E;
  A.M = j->M;
  A.mode = 0;
  A.data = j->data;
  A.wgt = j->wgt;
  A.a = j->par[LC_P_A];
  A.cx = j->par[LC_P_CX];
  A.cy = j->par[LC_P_CY];
  A.dx = j->par[LC_P_DX];
  A.dy = j->par[LC_P_DY];
  A.alpha = j->par[LC_P_ALPHA];
  A.mean = j->par[LC_P_MEAN];
  P.psf = j->psf_dev;
  P.F = j->psF;
  P.T = n_iter;
  P.sched = d_sched;
  P.ab = ab;
  P.free_a = j->free_mask[LC_P_A];
  P.free_dx = j->free_mask[LC_P_DX];
  P.free_dy = j->free_mask[LC_P_DY];
  P.free_mean = j->free_mask[LC_P_MEAN];
  P.lam_pos_ps = j->cfg.lam_positivity_ps;
  P.par_a = j->par[LC_P_A];
  P.par_dx = j->par[LC_P_DX];
  P.par_dy = j->par[LC_P_DY];
  P.par_mean = j->par[LC_P_MEAN];
  P.pm_a = j->pm[LC_P_A];
  P.ps_a = j->ps[LC_P_A];
  P.pm_dx = j->pm[LC_P_DX];
  P.ps_dx = j->ps[LC_P_DX];
  P.pm_dy = j->pm[LC_P_DY];
  P.ps_dy = j->ps[LC_P_DY];
  P.pm_mean = j->pm[LC_P_MEAN];
  P.ps_mean = j->ps[LC_P_MEAN];
  P.hist_e = d_hist_e;
  if (j->phist) {  // return_param_history: the kernel writes the rows of its iterations itself
    P.phist = j->phist + (size_t)j->phist_rows * j->phist_P;
    P.phist_P = j->phist_P;
    P.poff_a = j->phist_off[LC_P_A];
    P.poff_dx = j->phist_off[LC_P_DX];
    P.poff_dy = j->phist_off[LC_P_DY];
    P.poff_mean = j->phist_off[LC_P_MEAN];
    j->phist_rows += n_iter;
  }
  LC_HIP(j->ctx, hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, plds));
  hipLaunchKernelGGL(pk, dim3(j->E), dim3(kPsThreads), plds, q, P);
  hipLaunchKernelGGL(joint_ps_hist_kernel, dim3(n_iter), dim3(64), 0, q, j->E, n_iter, d_hist_e, j->hist + j->iters_done);
  LC_HIP(j->ctx, hipGetLastError());
  LC_HIP(j->ctx, hipStreamSynchronize(q));  // the two temporaries go out of scope
  j->iters_done += n_iter;
  *done = true;
  return LC_OK;
}

// ---- batched star photometry: every star of the batch advances by one iteration per kernel pair ----------------------------
namespace {
int ensure_group_hist(lc_joint *j, int needed) {
  if (needed <= j->ghist_cap) return LC_OK;
  const int nc = std::max(needed, 2 * j->ghist_cap + 64);
  float *nh = nullptr;
  LC_HIP(j->ctx, hipMalloc((void **)&nh, (size_t)j->G * nc * sizeof(float)));
  LC_HIP(j->ctx, hipMemsetAsync(nh, 0, (size_t)j->G * nc * sizeof(float), j->ctx->stream));
  if (j->ghist) {
    LC_HIP(j->ctx, hipMemcpy2DAsync(nh, (size_t)nc * sizeof(float), j->ghist, (size_t)j->ghist_cap * sizeof(float),
                                    (size_t)j->ghist_cap * sizeof(float), j->G, hipMemcpyDeviceToDevice, j->ctx->stream));
    LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
    hipFree(j->ghist);
  }
  j->ghist = nh;
  j->ghist_cap = nc;
  return LC_OK;
}
// the per-star views of the update arguments (pointers offset to the star's epochs), uploaded before a run
int upload_group_views(lc_joint *j, const lc_adabelief_cfg *cfg) {
  const int M = j->M;
  std::vector<JointUpdArgs> views(j->G);
  for (int g = 0; g < j->G; ++g) {
    JointUpdArgs &A = views[g];
    std::memset(&A, 0, sizeof(A));
    const int e0 = j->gstart[g];
    A.E = j->gstart[g + 1] - e0;
    A.M = M;
    A.ss = j->ss;
    A.fuse_scalar_reduce = 1;
    const size_t off[LC_P_COUNT] = {(size_t)e0 * M, (size_t)g * M, (size_t)g * M, (size_t)e0, (size_t)e0, (size_t)e0, 0, (size_t)e0};
    for (int k = 0; k < LC_P_COUNT; ++k) {
      A.free_mask[k] = j->free_mask[k];
      A.par[k] = j->par[k] + off[k];
      A.pm[k] = j->pm[k] + off[k];
      A.ps[k] = j->ps[k] + off[k];
    }
    A.shared = A.shared_w = j->shared_g + (size_t)g * (4 * M + 2);
    A.a_ref = j->a_ref_g + (size_t)g * kMaxSources;
    A.g_a = j->g_a + (size_t)e0 * M;
    A.g_cx_e = j->g_cx_e + (size_t)e0 * M;
    A.g_cy_e = j->g_cy_e + (size_t)e0 * M;
    A.g_dx = j->g_dx + e0;
    A.g_dy = j->g_dy + e0;
    A.g_mean = j->g_mean + e0;
    A.chi2_e = j->chi2_e + e0;
    A.hist = j->ghist + (size_t)g * j->ghist_cap;
    A.out_loss = j->out_loss_g + g;
    A.lam_pos_ps = j->cfg.lam_positivity_ps;
    A.lam_fu = j->cfg.lam_flux_uniformity;
    if (cfg) A.ab = *cfg; else lc_adabelief_defaults(&A.ab);
  }
  return h2d(j, j->views_dev, views.data(), views.size() * sizeof(JointUpdArgs));
}
int group_update(lc_joint *j, int mode, int t, const lc_adabelief_cfg *cfg, int g0 = 0, int g1 = -1, hipStream_t stream = nullptr) {
  lc_adabelief_cfg ab;
  if (cfg) ab = *cfg; else lc_adabelief_defaults(&ab);
  float lr, bc1, bc2;
  adabelief_schedule(ab, t, lr, bc1, bc2);
  if (g1 < 0) g1 = j->G;
  hipLaunchKernelGGL(joint_update_groups_kernel, dim3(2 * (g1 - g0)), dim3(kGmThreads), 0, stream ? stream : j->ctx->stream,
                     j->views_dev + g0, mode, t, lr, bc1, bc2);
  LC_HIP(j->ctx, hipGetLastError());
  return LC_OK;
}
int run_adabelief_groups(lc_joint *j, int n_iter, const lc_adabelief_cfg *cfg) {
  int rc = ensure_group_hist(j, j->iters_done + n_iter + 2);
  if (rc || (rc = upload_group_views(j, cfg))) return rc;
  // The stars are independent fits, and the update of a batch is a launch of two short blocks per star (a chain of memory
  // round trips, ~15 us with the machine idle around it).  The batch therefore runs as K parts on K streams, each part its
  // own sequence of kernel pairs, the first launches staggered: while one part steps, the point-source kernels of the
  // others have the CUs.  Same kernels on the same operands per star: same bits.  (LCMI_GROUP_STREAMS=<K>, default 2: the
  // two streams the object has anyway - 49.3 -> 45.8 us per iteration for 30 stars x 100 epochs; four parts reach 44.6 us in
  // a process of their own and 64 us where earlier objects have used up the runtime's four hardware queues and two parts
  // share one.  1: every star in one kernel pair per iteration.)
  int K = 2;
  if (const char *ks = std::getenv("LCMI_GROUP_STREAMS")) K = std::atoi(ks);
  K = std::max(1, std::min(std::min(K, j->G), 8));
  if (K == 1) {
    for (int it = 0; it < n_iter; ++it) {
      int need = launch_epochs(j, 0, 0, false, nullptr);
      if (need < 0) return need;
      if ((rc = group_update(j, 1, j->iters_done, cfg))) return rc;
      j->iters_done += 1;
    }
    return LC_OK;
  }
  // contiguous parts of about E / K epochs each: part k = stars [gb[k], gb[k + 1])
  std::vector<int> gb(K + 1, j->G);
  gb[0] = 0;
  for (int k = 1, g = 0; k < K; ++k) {
    while (g < j->G - (K - k) && (long long)j->gstart[g + 1] * K <= (long long)j->E * k) ++g;
    g = std::max(g, gb[k - 1] + 1);
    gb[k] = g;
  }
  while ((int)j->gevents.size() < K - 1) {
    hipEvent_t ev = nullptr;
    LC_HIP(j->ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    j->gevents.push_back(ev);
  }
  while ((int)j->gstreams.size() < K - 2) {
    hipStream_t st = nullptr;
    LC_HIP(j->ctx, hipStreamCreate(&st));
    j->gstreams.push_back(st);
  }
  // (the second part takes the object's second stream, which exists anyway; further parts streams of their own)
  auto stream_of = [&](int k) { return k == 0 ? j->ctx->stream : (k == 1 ? j->streamB : j->gstreams[k - 2]); };
  // fork: every part starts behind what the caller has enqueued, and part k's first kernel behind part k - 1's first
  // point-source kernel (the stagger; afterwards the parts keep out of each other's way by themselves)
  LC_HIP(j->ctx, hipEventRecord(j->evUpd, j->ctx->stream));
  for (int k = 1; k < K; ++k) LC_HIP(j->ctx, hipStreamWaitEvent(stream_of(k), j->evUpd, 0));
  for (int it = 0; it < n_iter && !rc; ++it) {
    for (int k = 0; k < K && !rc; ++k) {
      hipStream_t q = stream_of(k);
      if (it == 0 && k > 0) rc = (hipStreamWaitEvent(q, j->gevents[k - 1], 0) == hipSuccess) ? LC_OK : LC_ERR_DEVICE;
      int need = rc ? -1 : launch_epochs(j, 0, 0, false, nullptr, j->gstart[gb[k]], j->gstart[gb[k + 1]], q);
      if (!rc && need < 0) rc = need;
      if (!rc && it == 0 && k < K - 1) rc = (hipEventRecord(j->gevents[k], q) == hipSuccess) ? LC_OK : LC_ERR_DEVICE;
      if (!rc) rc = group_update(j, 1, j->iters_done, cfg, gb[k], gb[k + 1], q);
    }
    if (!rc) j->iters_done += 1;
  }
  // join (also on an error: nothing may be left running on the other streams behind the caller's back)
  for (int k = 1; k < K; ++k) {
    (void)hipEventRecord(j->gevents[k - 1], stream_of(k));
    (void)hipStreamWaitEvent(j->ctx->stream, j->gevents[k - 1], 0);
  }
  return rc;
}
}  // namespace

int lc_joint_get_group_loss_history(lc_joint *j, float *history, int count_per_group) {
  if (!j || !history || j->G <= 0 || count_per_group < j->iters_done + 1) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  int rc = ensure_group_hist(j, j->iters_done + 2);
  if (rc || (rc = upload_group_views(j, nullptr))) return rc;
  int need = launch_epochs(j, 0, 0, false, nullptr);  // loss of the final parameters -> hist[g][T]
  if (need < 0) return need;
  if ((rc = group_update(j, 0, j->iters_done, nullptr))) return rc;
  LC_HIP(j->ctx, hipMemcpy2DAsync(history, (size_t)count_per_group * sizeof(float), j->ghist, (size_t)j->ghist_cap * sizeof(float),
                                  (size_t)(j->iters_done + 1) * sizeof(float), j->G, hipMemcpyDeviceToHost, j->ctx->stream));
  LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
  return LC_OK;
}

int lc_joint_run_adabelief(lc_joint *j, int n_iter, const lc_adabelief_cfg *cfg) {
  if (!j || n_iter <= 0) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) return run_adabelief_groups(j, n_iter, cfg);
  int rc = ensure_hist(j, j->iters_done + n_iter + 2);
  if (rc) return rc;
  bool done = false;
  if ((rc = run_adabelief_persistent(j, n_iter, cfg, &done)) || done) return rc;
  // Cluster launches (few epochs per GPU) rely on all workgroups of an epoch being resident together; every wait in them is
  // bounded, and a run in which one gave up is REDONE here with the one-workgroup kernel from a copy of the state taken now
  // (parameters, both moments: the epoch kernels only read them) - self-healing, like the two-workgroup PSF kernel.
  const bool may_cluster = j->cl_ctr && !j->cluster_off && j->v->ek_cluster;
  const int it0 = j->iters_done, ph0 = j->phist_rows;
  struct SnapGuard {
    float *p = nullptr;
    ~SnapGuard() { if (p) (void)hipFree(p); }
  } snap;
  auto snapshot = [&](bool restore) -> int {
    size_t total = 0;
    for (int k = 0; k < LC_P_COUNT; ++k) total += 3 * (size_t)std::max(j->psize[k], 1);
    if (!restore) LC_HIP(j->ctx, hipMalloc((void **)&snap.p, total * sizeof(float)));
    float *c = snap.p;
    for (int k = 0; k < LC_P_COUNT; ++k)
      for (float *blk : {j->par[k], j->pm[k], j->ps[k]}) {
        const size_t cnt = (size_t)std::max(j->psize[k], 1);
        LC_HIP(j->ctx, hipMemcpyAsync(restore ? blk : c, restore ? c : blk, cnt * sizeof(float), hipMemcpyDeviceToDevice, j->ctx->stream));
        c += cnt;
      }
    return LC_OK;
  };
  // (the same copy serves the in-kernel waits of the loop's stream synchronisation - completion flags, the gate kernel: bounded,
  //  ~1 s; a run in which one ran out is redone with events, and the object keeps to events afterwards)
  const bool may_wait = j->mreg != nullptr && !j->force_events;
  if ((may_cluster || may_wait) && (rc = snapshot(false))) return rc;
  if (may_cluster && std::getenv("LCMI_CLUSTER_TEST_ABORT")) {  // test hook: the first cluster launch finds the abort word set
    const unsigned int one = 1;
    if ((rc = h2d(j, j->cl_ctr + (size_t)j->E * kClStride, &one, sizeof(one)))) return rc;
  }
  if ((rc = probe_streams(j))) return rc;
  j->in_device_loop = true;
  bool flags_used = false;
 redo:
  // LCMI_TIMELINE=1 (diagnostic): HIP events around the parts of ONE iteration in the middle of the run - where the epoch
  // kernels, the regulariser chain on the second stream and the update start and end relative to each other
  hipEvent_t tl[5] = {};
  const int tl_it = std::getenv("LCMI_TIMELINE") ? n_iter / 2 : -1;
  for (int it = 0; it < n_iter && !rc; ++it) {
    if (it == tl_it) {
      for (auto &e : tl) (void)hipEventCreate(&e);
      j->tl_events = tl;
      (void)hipEventRecord(tl[0], j->ctx->stream);
    }
    if ((rc = lc_joint_step_local(j))) break;
    if (it == tl_it) (void)hipEventRecord(tl[3], j->ctx->stream);
    rc = lc_joint_step_update(j, cfg);
    if (it == tl_it) {
      (void)hipEventRecord(tl[4], j->ctx->stream);
      j->tl_events = nullptr;
    }
    flags_used = flags_used || j->flag_sync || j->epoch_waited;
  }
  if (tl_it >= 0 && !rc) {
    (void)hipStreamSynchronize(j->ctx->stream);
    (void)hipStreamSynchronize(j->streamB);
    float b0 = 0, b1 = 0, e1 = 0, e2 = 0;
    const bool chain = hipEventElapsedTime(&b0, tl[0], tl[1]) == hipSuccess && hipEventElapsedTime(&b1, tl[0], tl[2]) == hipSuccess;
    (void)hipEventElapsedTime(&e1, tl[0], tl[3]);
    (void)hipEventElapsedTime(&e2, tl[0], tl[4]);
    std::fprintf(stderr, "timeline (us, from the end of the previous update): epoch kernels end %.1f, update ends %.1f", e1 * 1e3f, e2 * 1e3f);
    if (chain) std::fprintf(stderr, ", regulariser chain %.1f .. %.1f", b0 * 1e3f, b1 * 1e3f);
    std::fprintf(stderr, "\n");
    for (auto &e : tl) (void)hipEventDestroy(e);
  }
  if (!rc && may_cluster && !j->cluster_off && j->cl_parts_last > 0) {
    int aborted = 0;
    if ((rc = cluster_check(j, &aborted))) return rc;
    if (aborted) {  // (cluster_off is set now: the same iterations again, one workgroup per epoch)
      if ((rc = snapshot(true))) return rc;
      // (the next regulariser chain, on the second stream, starts behind this event: it must see the restored background)
      LC_HIP(j->ctx, hipEventRecord(j->evUpd, j->ctx->stream));
      if (std::getenv("LCMI_CLUSTER_DEBUG") && j->reg_flag) {
        unsigned int w[2] = {0, 0};
        (void)d2h(j, w, j->reg_flag, sizeof(w));
        std::fprintf(stderr, "cluster fall-back: regulariser flag %u (host %u), wait error %u, flags_used %d\n", w[0], j->reg_seq, w[1], (int)flags_used);
      }
      j->iters_done = it0;
      j->phist_rows = ph0;
      j->reg_pending = j->fuse_pending = j->fuse_full = j->pts_pending = false;
      j->upd_gate_pending = false;   // (the event recorded above orders the next chain)
      goto redo;
    }
  }
  // in-kernel waits of this run that ran out (the update's wait for the chain or the epoch launch's extra block: reg_flag + 1;
  // the gate kernel: upd_ctr + 1): the numbers of the run cannot be trusted.  Once per object: restore the state copied at
  // the start, switch the object to events, and run the same iterations again.
  {
    unsigned int err_flag = 0, err_gate = 0;
    if (!rc && flags_used && j->reg_flag) {
      (void)hipStreamSynchronize(j->streamB);
      rc = d2h(j, &err_flag, j->reg_flag + 1, sizeof(err_flag));
    }
    if (!rc && j->upd_gate_used && j->upd_ctr) {
      (void)hipStreamSynchronize(j->streamB);
      rc = d2h(j, &err_gate, j->upd_ctr + 1, sizeof(err_gate));
    }
    if (!rc && (err_flag || err_gate)) {
      const unsigned int zero = 0;
      if (err_flag) (void)h2d(j, j->reg_flag + 1, &zero, sizeof(zero));
      if (err_gate) (void)h2d(j, j->upd_ctr + 1, &zero, sizeof(zero));
      if (may_wait && snap.p && !j->force_events) {
        j->force_events = true;
        j->wait_fallbacks += 1;
        if (std::getenv("LCMI_DEBUG_STREAMS")) std::fprintf(stderr, "lc_joint: an in-kernel wait ran out (flag %u, gate %u): run redone with events\n", err_flag, err_gate);
        LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
        LC_HIP(j->ctx, hipStreamSynchronize(j->streamB));
        if ((rc = snapshot(true))) return rc;
        LC_HIP(j->ctx, hipEventRecord(j->evUpd, j->ctx->stream));
        j->iters_done = it0;
        j->phist_rows = ph0;
        j->reg_pending = j->fuse_pending = j->fuse_full = j->pts_pending = false;
        j->upd_gate_pending = j->upd_gate_used = j->upd_signal_due = false;
        flags_used = false;
        goto redo;
      }
      unsigned int seen = 0;
      if (j->reg_flag) (void)d2h(j, &seen, j->reg_flag, sizeof(seen));
      static thread_local char msg[256];
      std::snprintf(msg, sizeof(msg), "joint fit: the regulariser of an iteration did not complete in time on the second stream (set LCMI_EVENT_SYNC=1); completion word %u, expected %u", seen, j->reg_seq);
      j->ctx->err = msg;
      rc = LC_ERR_DEVICE;
    }
  }
  j->in_device_loop = false;
  j->flag_sync = false;
  {
    const int rg = upd_gate_finish(j);
    if (!rc) rc = rg;
  }
  if (!rc) rc = chain_check(j);
  return rc;
}
int lc_joint_iterations_done(lc_joint *j) { return j ? j->iters_done : LC_ERR_INVALID; }
int lc_joint_cluster_info(lc_joint *j, int *parts_last, int *fallbacks) {
  if (!j) return LC_ERR_INVALID;
  if (parts_last) *parts_last = j->cl_parts_last;
  if (fallbacks) *fallbacks = j->cl_fallbacks;
  return LC_OK;
}

int lc_joint_param_history_end(lc_joint *j) {
  if (!j) return LC_ERR_INVALID;
  if (j->phist) {
    LC_ENTER(j->ctx);
    LC_HIP(j->ctx, hipStreamSynchronize(j->ctx->stream));
    (void)hipFree(j->phist);
  }
  j->phist = nullptr;
  j->phist_cap = j->phist_P = j->phist_rows = 0;
  return LC_OK;
}
int lc_joint_param_history_begin(lc_joint *j, int capacity, int *n_params) {
  if (!j || capacity <= 0) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_param_history_begin: not available on a batched star-photometry object");
  int rc = lc_joint_param_history_end(j);
  if (rc) return rc;
  int P = 0;
  for (int k = 0; k < LC_P_COUNT; ++k) {
    j->phist_off[k] = j->free_mask[k] ? P : -1;
    if (j->free_mask[k]) P += j->psize[k];
  }
  if (n_params) *n_params = P;
  if (P == 0) return LC_OK;
  LC_HIP(j->ctx, hipMalloc((void **)&j->phist, (size_t)capacity * P * sizeof(float)));
  j->phist_cap = capacity;
  j->phist_P = P;
  j->phist_rows = 0;
  return LC_OK;
}
int lc_joint_param_history_rows(lc_joint *j) { return j ? j->phist_rows : LC_ERR_INVALID; }
int lc_joint_param_history_get(lc_joint *j, int first, int count, float *out) {
  if (!j || !out || first < 0 || count < 0) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (first + count > j->phist_rows) LC_FAIL(j->ctx, LC_ERR_INVALID, "lc_joint_param_history_get: rows not recorded");
  if (count == 0) return LC_OK;
  return d2h(j, out, j->phist + (size_t)first * j->phist_P, (size_t)count * j->phist_P * sizeof(float));
}

// ---- bounded L-BFGS with the vectors on the device (csrc/joint_lbfgs.h) ------------------------------------------------
int lc_joint_run_lbfgs(lc_joint *j, int maxiter, const float *const lower[LC_P_COUNT], const float *const upper[LC_P_COUNT],
                       float *loss_history, int history_capacity, int *n_iterations, int *n_evaluations) {
  if (!j || maxiter < 0) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_run_lbfgs: not available on a batched star-photometry object");
  int D = 0, off[LC_P_COUNT] = {};
  for (int k = 0; k < LC_P_COUNT; ++k) {
    off[k] = D;
    if (j->free_mask[k]) D += j->psize[k];
  }
  if (n_iterations) *n_iterations = 0;
  if (n_evaluations) *n_evaluations = 0;
  if (D == 0) return LC_OK;
  hipStream_t q = j->ctx->stream;
  const float inf = std::numeric_limits<float>::infinity();
  std::vector<float> lo((size_t)D, -inf), hi((size_t)D, inf);
  for (int k = 0; k < LC_P_COUNT; ++k) {
    if (!j->free_mask[k]) continue;
    for (int i = 0; i < j->psize[k]; ++i) {
      if (lower && lower[k]) lo[off[k] + i] = lower[k][i];
      if (upper && upper[k]) hi[off[k] + i] = upper[k][i];
    }
  }
  // device work space (freed on every exit path)
  float *buf = nullptr;
  int *order_dev = nullptr;
  const size_t nvec = 7 + 2 * (size_t)kLbMem;
  LC_HIP(j->ctx, hipMalloc((void **)&buf, (nvec * D + kLbMem + 8) * sizeof(float)));
  struct Guard {
    float *b;
    int **o;
    ~Guard() {
      (void)hipFree(b);
      if (*o) (void)hipFree(*o);
    }
  } guard{buf, &order_dev};
  LC_HIP(j->ctx, hipMalloc((void **)&order_dev, kLbMem * sizeof(int)));
  // hipMalloc hands back recycled bytes: nothing below reads a vector before writing it, and the work space starts from
  // zeros all the same (LCMI_LBFGS_POISON, a test hook, fills it with NaN patterns instead: the results must not change)
  LC_HIP(j->ctx, hipMemsetAsync(buf, std::getenv("LCMI_LBFGS_POISON") ? 0xFF : 0, (nvec * D + kLbMem + 8) * sizeof(float), q));
  LC_HIP(j->ctx, hipMemsetAsync(order_dev, 0, kLbMem * sizeof(int), q));
  LbfgsDev L;
  L.D = D;
  L.x = buf;
  L.g = buf + (size_t)D;
  L.xt = buf + 2 * (size_t)D;
  L.gt = buf + 3 * (size_t)D;
  L.dir = buf + 4 * (size_t)D;
  float *dlo = buf + 5 * (size_t)D, *dhi = buf + 6 * (size_t)D;
  L.lo = dlo;
  L.hi = dhi;
  L.S = buf + 7 * (size_t)D;
  L.Y = L.S + (size_t)kLbMem * D;
  L.rho = L.Y + (size_t)kLbMem * D;
  L.scal = L.rho + kLbMem;
  LC_HIP(j->ctx, hipMemcpyAsync(dlo, lo.data(), (size_t)D * sizeof(float), hipMemcpyHostToDevice, q));
  LC_HIP(j->ctx, hipMemcpyAsync(dhi, hi.data(), (size_t)D * sizeof(float), hipMemcpyHostToDevice, q));
  auto pack = [&](float *const src[LC_P_COUNT], float *vec) -> int {   // parameter blocks -> flat vector
    for (int k = 0; k < LC_P_COUNT; ++k)
      if (j->free_mask[k])
        LC_HIP(j->ctx, hipMemcpyAsync(vec + off[k], src[k], (size_t)j->psize[k] * sizeof(float), hipMemcpyDeviceToDevice, q));
    return LC_OK;
  };
  auto unpack = [&](const float *vec) -> int {                          // flat vector -> parameter blocks
    for (int k = 0; k < LC_P_COUNT; ++k)
      if (j->free_mask[k])
        LC_HIP(j->ctx, hipMemcpyAsync(j->par[k], vec + off[k], (size_t)j->psize[k] * sizeof(float), hipMemcpyDeviceToDevice, q));
    return LC_OK;
  };
  const bool want_h = j->free_mask[LC_P_H] != 0;
  float host[8];
  int evals = 0;
  // loss and gradient at the parameters now on the device -> (out_loss, gout); gradient packed into `gvec`
  auto evaluate = [&](float *gvec, float &loss) -> int {
    int need = launch_epochs(j, 0, 0, want_h, nullptr);
    if (need < 0) return need;
    int rc = launch_reduce(j, need);
    if (rc) return rc;
    if ((rc = launch_update(j, 0, 0, nullptr, false, true))) return rc;
    if ((rc = pack(j->gout, gvec))) return rc;
    LC_HIP(j->ctx, hipMemcpyAsync(host, j->out_loss, sizeof(float), hipMemcpyDeviceToHost, q));
    LC_HIP(j->ctx, hipStreamSynchronize(q));
    loss = host[0];
    ++evals;
    return LC_OK;
  };
  int rc;
  if ((rc = pack(j->par, L.x))) return rc;
  // start inside the box
  hipLaunchKernelGGL(lb_clip_kernel, dim3(1), dim3(kLbThreads), 0, q, L);
  if ((rc = unpack(L.x))) return rc;
  float f = 0.f;
  if ((rc = evaluate(L.g, f))) return rc;
  int n_hist = 0;
  auto record = [&](float v) {
    if (loss_history && n_hist < history_capacity) loss_history[n_hist] = v;
    ++n_hist;
  };
  int m = 0, head = 0, iters = 0;  // m pairs stored, next slot = head
  const float c1 = 1e-4f, gtol = 1e-6f, ftol = 2.2e-9f;
  bool finite = std::isfinite(f);
  while (finite && iters < maxiter) {
    int order[kLbMem];
    for (int k = 0; k < m; ++k) order[k] = (head - m + k + 2 * kLbMem) % kLbMem;  // oldest .. newest
    LC_HIP(j->ctx, hipMemcpyAsync(order_dev, order, sizeof(order), hipMemcpyHostToDevice, q));
    hipLaunchKernelGGL(lb_direction_kernel, dim3(1), dim3(kLbThreads), 0, q, L, m, order_dev);
    LC_HIP(j->ctx, hipMemcpyAsync(host, L.scal, 3 * sizeof(float), hipMemcpyDeviceToHost, q));
    LC_HIP(j->ctx, hipStreamSynchronize(q));
    float gd = host[0], dd = host[1];
    const float pgmax = host[2];
    if (pgmax < gtol * std::max(1.f, std::fabs(f))) break;
    if (!(gd < 0.f)) {  // not a descent direction: restart from steepest descent
      m = 0;
      hipLaunchKernelGGL(lb_steepest_kernel, dim3(1), dim3(kLbThreads), 0, q, L);
      LC_HIP(j->ctx, hipMemcpyAsync(host, L.scal, 2 * sizeof(float), hipMemcpyDeviceToHost, q));
      LC_HIP(j->ctx, hipStreamSynchronize(q));
      gd = host[0];
      dd = host[1];
      if (!(gd < 0.f)) break;
    }
    float alpha = (m == 0) ? std::min(1.f, 1.f / std::sqrt(std::max(dd, 1e-30f))) : 1.f;
    bool accepted = false;
    float ft = f;
    for (int ls = 0; ls < 25; ++ls) {
      hipLaunchKernelGGL(lb_trial_kernel, dim3(1), dim3(kLbThreads), 0, q, L, alpha);
      if ((rc = unpack(L.xt))) return rc;
      if ((rc = evaluate(L.gt, ft))) return rc;
      LC_HIP(j->ctx, hipMemcpyAsync(host, L.scal + 3, sizeof(float), hipMemcpyDeviceToHost, q));
      LC_HIP(j->ctx, hipStreamSynchronize(q));
      const float dec = host[0];
      if (std::isfinite(ft) && ft <= f + c1 * dec) {
        accepted = true;
        break;
      }
      alpha *= 0.5f;
    }
    if (!accepted) {  // line search failed: back to the last accepted point
      if ((rc = unpack(L.x))) return rc;
      break;
    }
    hipLaunchKernelGGL(lb_accept_kernel, dim3(1), dim3(kLbThreads), 0, q, L, head);
    LC_HIP(j->ctx, hipMemcpyAsync(host, L.scal + 4, 3 * sizeof(float), hipMemcpyDeviceToHost, q));
    LC_HIP(j->ctx, hipStreamSynchronize(q));
    const float sy = host[0], ss = host[1], yy = host[2];
    if (sy > 1e-10f * std::sqrt(ss * yy) && sy > 0.f) {  // keep the pair (the slot was written by the kernel)
      head = (head + 1) % kLbMem;
      m = std::min(m + 1, kLbMem);
    }
    const float fold = f;
    f = ft;
    ++iters;
    record(f);
    if (std::fabs(fold - f) <= ftol * std::max({std::fabs(fold), std::fabs(f), 1.f})) break;
  }
  LC_HIP(j->ctx, hipGetLastError());
  LC_HIP(j->ctx, hipStreamSynchronize(q));
  if (n_iterations) *n_iterations = iters;
  if (n_evaluations) *n_evaluations = evals;
  if (n_hist == 0) record(f);
  return finite ? LC_OK : LC_ERR_NONFINITE;
}

int lc_joint_get_loss_history(lc_joint *j, float *history, int count) {
  if (!j || !history || count < j->iters_done + 1) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  if (j->G > 0) LC_FAIL(j->ctx, LC_ERR_UNSUPPORTED, "lc_joint_get_loss_history (use lc_joint_get_group_loss_history): not available on a batched star-photometry object");
  int rc = ensure_hist(j, j->iters_done + 2);
  if (rc) return rc;
  int need = launch_epochs(j, 0, 0, false, nullptr);  // loss of the final parameters -> hist[T]
  if (need < 0) return need;
  if ((rc = launch_reduce(j, need))) return rc;
  if ((rc = launch_update(j, 0, j->iters_done, nullptr, true, false))) return rc;
  return d2h(j, history, j->hist, (size_t)(j->iters_done + 1) * sizeof(float));
}

int lc_joint_fisher_flux_sigma(lc_joint *j, float *sigma_a) {
  if (!j || !sigma_a) return LC_ERR_INVALID;
  LC_ENTER(j->ctx);
  for (int i = 0; i < j->M; ++i) {
    int rc = launch_epochs(j, 2, i, false, nullptr);
    if (rc < 0) return rc;
  }
  return d2h(j, sigma_a, j->fisher, (size_t)j->E * j->M * sizeof(float));
}

#ifdef LC_STAMPS
int lc_debug_get_jstamps(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc::g_jstamps), 32 * sizeof(long long)) == hipSuccess ? 0 : -2;
}
int lc_debug_get_rstamps(long long *out) {  // (the row-block regulariser kernel: tools/rows_stamps.py)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc::g_rstamps), 16 * sizeof(long long)) == hipSuccess ? 0 : -2;
}
int lc_debug_get_ustamps(long long *out) {  // (the fused reduction + update launch: tools/update_stamps.py)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc::g_ustamps), 16 * sizeof(long long)) == hipSuccess ? 0 : -2;
}
#endif

}  // extern "C"
