// Joint multi-epoch forward model (STARRED "Deconv") device code, gfx950.
//
// Replaces Deconv.model / Loss / autodiff gradient as called from
// lightcurver/processes/star_photometry.py:66-137 and roi_modelling.py:213-334 (reference):
//   f_e = D_ss[ s_e (*) ( T_e[h] + sum_i a_{e,i} G(R_e c_i + d_e) ) ] + mean_e
// Kernel 1 (one workgroup per epoch): scene build, zero-padded FFT convolution with the epoch's
//   narrow PSF (spectrum precomputed), down-sampling, residuals, chi2, adjoint convolution, and
//   the analytic gradients w.r.t. a, c, dx, dy, mean plus T_e^T of the scene gradient (h slab).
// Kernel 2: deterministic reduction over epochs into the shared block [dh | dc_x | dc_y | ...].
// Kernel 3 (one workgroup): starlet l1 / positivity / flux terms, loss, fused AdaBelief update.
#pragma once
#include "fft_device.h"
#include "starlet_device.h"

namespace lc {

constexpr int kMaxSources = 8;

#ifdef LC_STAMPS
__device__ long long g_jstamps[32];
__device__ long long g_ustamps[16];
__device__ long long g_rstamps[16];
#define LC_JSTAMP(k)                                                \
  do {                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_jstamps[k] = clock64(); \
  } while (0)
// (the fused reduction + update launch: thread 0 of whichever block calls, 100 MHz wall clock - tools/update_stamps.py)
#define LC_USTAMP(k) do { if (threadIdx.x == 0) g_ustamps[k] = wall_clock64(); } while (0)
#else
#define LC_JSTAMP(k) do {} while (0)
#define LC_USTAMP(k) do {} while (0)
#endif

struct JointArgs {
  int E, M, mode, isrc;       // mode 0 = forward + backward, 1 = forward only, 2 = Fisher diagonal of source isrc
  int h_active, need_hgrad;   // h present in the scene; produce T^T slabs
  const float *data, *wgt;    // [E][n][n], wgt = 1 / sigma^2 (0 where invalid)
  const float2 *St;           // [E][L/2+1][L] PSF spectrum / L^2, transposed
  float2 *spec;               // [E][N][L/2+1] global spectrum scratch (GSPEC kernels only)
  const float2 *twid;         // [L] exp(-2 pi i m / L)
  const float *a, *cx, *cy, *dx, *dy, *alpha, *h, *mean;
  float *tabs;                // [E][4][M][N] gx, dgx, gy, dgy scratch
  float *HG;                  // [E][N*N] T_e^T (scene gradient)
  float *chi2_e, *g_a, *g_cx_e, *g_cy_e, *g_dx, *g_dy, *g_mean;
  float *model_out;           // [E][n][n] or null
  float *part;                // phased launches: [E][parts][4 + 3 kMaxSources] partial sums of the epoch's workgroups
  float *fisher_out;          // [E][M]
  // auxiliary instantiation (AUX = true; noise propagation on the device, csrc/joint_noise.h):
  //   mode 3: conv_same(scene_in[e], image whose spectrum is St[e]) -> conv_out[e] at full resolution
  //   mode 4: spectrum of scene_in[e] in the layout of St -> St_out[e]
  const float *scene_in;      // [E][N][N]
  float2 *St_out;             // [E][L/2+1][L]
  float *conv_out;            // [E][N][N]
  // batched star photometry (point-source-only kernel): epoch -> star; the shared positions are cx[group[e] * M + i]
  const int *group;           // [E] or null
  // the T_e^T step is applied by the reduction itself (joint_stencil_update_kernel): phase D leaves the scene-gradient rows
  // in the spectrum scratch and writes no slab
  int skip_D;
  float *tshift;              // [E][2] the shifts (dx, dy) this evaluation used: the reduction runs beside the update of dx, dy
  // cluster launch (PHASE = 7): the cl_parts workgroups of an epoch run in ONE launch, their phases separated by arrival
  // counters in global memory instead of launch boundaries (joint_epoch_kernel, cluster_sync)
  unsigned int *cl_ctr;       // [E][kClStride] one flag word per workgroup of the epoch: (sequence number << 4) | XCC id
  unsigned int cl_base;       // sequence number of the last sync before this launch (kept by the host: + kClBarriers per launch)
  unsigned int *cl_abort;     // [1] a wait ran out (sticky until the host clears it): every later wait gives up at once
  int cl_parts;
  // "the update in front of this launch is complete": the first thread of the launch stores upd_value here.  Inside the
  // library's loops the regulariser chain of the iteration starts behind a one-wave gate kernel that polls this word, instead
  // of behind an event the main stream had to record between the update and this launch (4.7 - 6 us of every iteration:
  // C4 71.7 -> 67, 25-epoch shard 56.6 -> 50.5, C5 shard 228 -> 222.6 us).  Stream order makes it true: this kernel starts
  // when the update has ended and its stores are out.  (The signal from inside the update - every block counting itself in
  // behind write-through stores - was built first and cost what it saved: +4.5 us in the update, +4.4 us for the gate's
  // polling against 514 atomic additions on the same word.)
  unsigned int *upd_signal;
  unsigned int upd_value;
  // ... and in the other direction: the update behind this launch needs the chain's results.  One EXTRA block of the launch
  // (blockIdx.x == wait_block) does nothing but poll the chain's completion word - so this launch is complete only when the
  // chain is, and the update needs neither a cross-stream event wait in front of it (6.2 us in the rocprofv3 timeline of the
  // C5 shard) nor a poll and cache-bypassing loads of its own.  The block holds one wave slot while it waits (bounded, ~1 s:
  // *chain_err); the chain is normally long done when the epoch's last phase starts.
  const unsigned int *chain_flag;
  unsigned int chain_seq;
  unsigned int *chain_err;
  int wait_block;
};
constexpr int kClStride = 32;    // flag words per epoch (one 128-byte line; kMaxParts <= 16 of them in use)
constexpr int kClBarriers = 6;   // start | A | B | C | B' | C' | D: every cluster launch passes exactly six syncs

// Data handed from one workgroup of an epoch to the others inside ONE launch (cluster form).  Two forms, chosen per launch
// by what the workgroups find out about each other at their first sync (MI355X_MICROARCH.md, inter-workgroup visibility):
//  * general: every store write-through and every load L1-bypassing (agent-scope relaxed atomics = global_store /
//    global_load ... sc1), the storing waves drain vmcnt, workgroup barrier, one lane raises an sc1 flag - valid wherever
//    the workgroups run, but an sc1 store drops its line from the XCD's L2 and every reader goes to the memory side;
//  * all workgroups of the epoch on ONE XCD (each reads HW_REG_XCC_ID; a workgroup never moves): plain stores, which stay in
//    the L2 they share, and the same L1-bypassing loads, now served from that L2; the flag likewise.  Speed only: a wrong
//    guess is impossible (the ids are exchanged, not assumed) and either form is correct under any placement.
// Relaxed atomics stay in flight like plain accesses (the compiler counts them in vmcnt; nothing waits here).
template <bool CL>
__device__ __forceinline__ float2 xwg_load(const float2 *p) {
  if constexpr (CL) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned int)v), __uint_as_float((unsigned int)(v >> 32)));
  } else {
    return *p;
  }
}
template <bool CL>
__device__ __forceinline__ void xwg_store(float2 *p, float2 v, bool same_xcd = false) {
  if constexpr (CL) {
    if (same_xcd) {
      *p = v;
    } else {
      const unsigned long long u = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
      __hip_atomic_store((unsigned long long *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    *p = v;
  }
}
template <bool CL>
__device__ __forceinline__ float xwg_loadf(const float *p) {
  if constexpr (CL) return __uint_as_float(__hip_atomic_load((const unsigned int *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
}
template <bool CL>
__device__ __forceinline__ void xwg_storef(float *p, float v, bool same_xcd = false) {
  if constexpr (CL) {
    if (same_xcd) *p = v;
    else __hip_atomic_store((unsigned int *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}
// Sync over the `nparts` workgroups of one epoch (every thread of the workgroup calls; st: two LDS words).  Every wave drains
// its stores, the workgroup meets, one lane raises the workgroup's flag - (sequence number << 4) | XCC id; the sequence
// number grows by one per sync over all launches, so nothing is ever reset - and lanes 0 .. nparts - 1 of the first wave
// poll one flag each (L1-bypassing, relaxed) until all have reached the number.  Bounded by the wall clock (s_memrealtime,
// 100 MHz: ~0.5 s): a workgroup whose partners are not resident - the one way this can fail - is reported through the abort
// word instead of waited for, and once the word is set every later wait gives up at once (the launches enqueued behind
// this one included: `first`).  Returns false when the wait was given up.  At the launch's first sync the lanes also
// compare the partners' XCC ids with their own: st[1] = 1 when all of them run on this XCD.
__device__ __forceinline__ bool cluster_sync(unsigned int *flags, int part, int nparts, unsigned int seq, unsigned int xcc,
                                             bool same_xcd, unsigned int *abort_word, int tid, int *st, bool first) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < 64) {
    const unsigned int word = (seq << 4) | xcc;
    if (tid == 0) {
      if (same_xcd) asm volatile("global_store_dword %0, %1, off" ::"v"(flags + part), "v"(word) : "memory");
      else __hip_atomic_store(flags + part, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int good = 1;
    if (first && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) good = 0;
    const unsigned int *mine = flags + (tid < nparts ? tid : 0);
    unsigned int got = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (28-bit sequence numbers compared modulo 2^28: the shifted words wrap with the 32-bit arithmetic)
    if (good && !__all((int)((got & ~15u) - (seq << 4)) >= 0)) {
      const long long t0 = wall_clock64();
      int spins = 0;
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        got = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((int)((got & ~15u) - (seq << 4)) >= 0)) break;
        if ((++spins & 63) == 0 &&
            (wall_clock64() - t0 > 50000000ll || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
          good = 0;
          break;
        }
      }
    }
    if (!good && tid == 0) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int same = __all((got & 15u) == xcc) ? 1 : 0;
    if (tid == 0) {
      st[0] = good;
      if (first) st[1] = same;
    }
  }
  __syncthreads();
  return st[0] != 0;
}

__device__ __forceinline__ void sample_coords(int u, int v, float c0, float ca, float sa, float sdx, float sdy,
                                              float &Xs, float &Ys) {
  const float qx = ((float)v - c0) - sdx;
  const float qy = ((float)u - c0) - sdy;
  Xs = c0 + (ca * qx + sa * qy);
  Ys = c0 + (ca * qy - sa * qx);
}

template <int N>
__device__ __forceinline__ float bilinear_h(const float *h, float Xs, float Ys, float &dHx, float &dHy) {
  const float x0f = floorf(Xs), y0f = floorf(Ys);
  const float fx = Xs - x0f, fy = Ys - y0f;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int xa = min(max(x0, 0), N - 1), xb = min(max(x0 + 1, 0), N - 1);
  const int ya = min(max(y0, 0), N - 1), yb = min(max(y0 + 1, 0), N - 1);
  const float h00 = h[ya * N + xa], h01 = h[ya * N + xb], h10 = h[yb * N + xa], h11 = h[yb * N + xb];
  const float top = h00 + fx * (h01 - h00), bot = h10 + fx * (h11 - h10);
  dHx = (1.f - fy) * (h01 - h00) + fy * (h11 - h10);
  dHy = bot - top;
  return top + fy * (bot - top);
}

template <int N_, int SS_, int L_, int NW_, bool GSPEC_ = false, int LPF_ = 16, bool TILECOLS_ = false>
struct JointCfg {
  static constexpr bool TILECOLS = TILECOLS_ && GSPEC_;  // column passes through an LDS tile (see OFF_TILE below)
  static constexpr int N = N_, SS = SS_, L = L_, n = N / SS;
  // lanes per transform (fft_device.h) and transforms side by side in a wave
  static constexpr int LPF = LPF_, GPW = 64 / LPF_;
  // GSPEC: the N x (L/2+1) half spectrum of the epoch lives in a global scratch instead of LDS (N = 256 does not fit)
  static constexpr bool GSPEC = GSPEC_;
  static constexpr int NW = NW_, NTHR = 64 * NW_;  // waves per epoch workgroup: as many as the LDS workspace allows
  static constexpr int KH = L / 2 + 1;           // stored spectrum columns
  // row stride of the spectrum: in global memory a multiple of 16 elements, so that rows and the 16-column tiles of the
  // column passes start on 128-byte lines
  static constexpr int KS = GSPEC_ ? (KH + 15) / 16 * 16 : KH;
  static constexpr int OFF_SPEC = 0;             // float2 units
  static constexpr int SZ_SPEC = GSPEC ? 0 : N * KH;
  static constexpr int OFF_WS = OFF_SPEC + SZ_SPEC;
  // linear row buffer(s) for the data-space step: one per quarter-wave when LDS allows (N <= 64), else one per
  // wave that the four quarters use in turn
  static constexpr bool WSQ = ((N <= 128) && (NW_ * GPW <= 32)) || GSPEC_;
  static constexpr int SZ_WS = NW * (WSQ ? GPW : 1) * L;
  static constexpr int OFF_TW = OFF_WS + SZ_WS;
  static constexpr int SZ_TW = L;
  static constexpr int OFF_RED = OFF_TW + SZ_TW;  // float2 units; reduction scratch as floats
  static constexpr int SZ_RED = ((NW + 1) * (4 + 3 * kMaxSources) + 1) / 2 + 8;
  // three rows of h per quarter-wave (translated epochs read the background from LDS): L float2 = 3 N floats, i.e. the
  // quarter's row buffer where there is one, a region of its own otherwise
  static constexpr int OFF_HROW = OFF_RED + SZ_RED;
  static constexpr int SZ_HROW = WSQ ? 0 : NW * GPW * ((3 * N + 1) / 2);
  // separable Gaussian factors of the point sources: GX[i][N], GY[i][N] floats, then their centres X_i, Y_i
  static constexpr int OFF_TAB = OFF_HROW + SZ_HROW;
  static constexpr int SZ_TAB = kMaxSources * N + kMaxSources + (kMaxSources + 1) / 2;  // + the fluxes of the epoch
  // binned rows (SS = 2): the inverse row transforms of the model and the forward ones of the residual run at the DATA
  // resolution, length L / 2 (joint_epoch_kernel, phase C): its twiddles, and phi[k] = exp(2 pi i k CREF / L) (1 +
  // exp(2 pi i k / L)), the transfer function of "shift by CREF, add neighbours" that precedes the decimation
  static constexpr bool FOLD = (SS == 2) && ((L / 2) % LPF == 0);
  // scene phase: next sweep's rows of h held in registers across a sweep (needs 24 registers: not at 4 waves per SIMD)
  static constexpr bool ROWPIPE = (NW_ <= 8);
  static constexpr int OFF_TWH = OFF_TAB + SZ_TAB;
  static constexpr int SZ_TWH = FOLD ? (L / 2 + L / 2 + 1) : 0;
  // GSPEC: the column passes can stage the NW * GPW spectrum columns of a sweep through an LDS tile [N][CPS + 1]: every
  // global access then moves whole 128-byte lines instead of 8 or 16 bytes of a line per lane.  Two workgroup barriers per
  // sweep: no gain while few workgroups run, faster once the column traffic of ~100 and more workgroups saturates L2 /
  // Infinity Cache: a kernel variant of its own (TILECOLS), chosen by the host from the number of workgroups.
  static constexpr int CPS = NW * GPW, TP = CPS + 1;
  static constexpr int OFF_TILE = OFF_TWH + SZ_TWH;
  static constexpr int SZ_TILE = TILECOLS ? N * TP : 0;
  static constexpr int LDS_BYTES = (OFF_TILE + 2 * SZ_TILE) * 8;  // (two tiles: see column_sweeps)
  // phased launches: the column phases and phase D use neither the row workspace nor the tables of the point sources; their
  // workgroups allocate [ twiddles | reduction scratch | tile ] only, so that several of them fit on a CU
  static constexpr int LITE_RED = L, LITE_TILE = L + SZ_RED, LDS_LITE = (LITE_TILE + 2 * SZ_TILE) * 8;  // (two tiles: see column_sweeps)
  static_assert(!WSQ || 2 * L >= 3 * N, "row buffer holds three rows of h");
  static constexpr int CREF = (N - 1) / 2;
  static_assert(N % 2 == 0, "row pairs");
  static_assert(L >= 2 * N - 1 - CREF, "FFT length too short for an alias-free 'same' window");
  static_assert(LDS_BYTES <= 163840, "LDS");
};


// Phased launches: the totals of an epoch are the sums of its workgroups' partial sums (in workgroup order); the outputs are
// those of the reductions at the end of the one-workgroup kernel.  Every thread of the workgroup calls (barrier inside); TOT: LDS.
template <bool SC1 = false>
__device__ __forceinline__ void joint_epoch_totals(const JointArgs &A, int e, int tid, int SS, int nparts, float *TOT) {
  constexpr int NQ = 4 + 3 * kMaxSources;
  const int M = A.M, nq = 4 + 3 * M;
  if (tid < nq) {
    float acc = 0.f;
    if constexpr (SC1) {
      // (L1-bypassing loads are not speculated by the compiler: all of them requested at once from clamped addresses, then
      //  added in workgroup order - one round trip instead of one per workgroup)
      constexpr int MAXP = 16;
      float v[MAXP];
#pragma unroll
      for (int p = 0; p < MAXP; ++p) v[p] = xwg_loadf<true>(&A.part[((size_t)e * nparts + min(p, nparts - 1)) * NQ + tid]);
#pragma unroll
      for (int p = 0; p < MAXP; ++p) acc += (p < nparts) ? v[p] : 0.f;
    } else {
      for (int p = 0; p < nparts; ++p) acc += A.part[((size_t)e * nparts + p) * NQ + tid];
    }
    TOT[tid] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    const float al = A.alpha[e] * 0.017453292519943295f;
    const float ca = cosf(al), sa = sinf(al);
    const float *t = TOT;
    A.chi2_e[e] = t[0];
    A.g_mean[e] = t[1];
    float gdx = t[2], gdy = t[3];
    for (int i = 0; i < M; ++i) {
      const float ai = A.a[e * M + i];
      const float gX = ai * t[5 + 3 * i], gY = ai * t[6 + 3 * i];
      A.g_a[e * M + i] = t[4 + 3 * i];
      gdx += SS * gX;
      gdy += SS * gY;
      A.g_cx_e[e * M + i] = SS * (ca * gX + sa * gY);
      A.g_cy_e[e * M + i] = SS * (ca * gY - sa * gX);
    }
    A.g_dx[e] = gdx;
    A.g_dy[e] = gdy;
  }
}

// PHASE = 0: the whole epoch in one workgroup (above).  PHASE = 1 .. 6 (global-spectrum kernels): ONE phase (A, B, C, B', C',
// D) per launch on a grid (E, parts): the `parts` workgroups of an epoch share the phase's rows / columns / pixels, the
// spectrum travels between the launches through global memory, and the launch boundaries are the synchronisation (no
// waiting inside a kernel).  An epoch that would occupy one CU out of two then uses the whole machine, and every launch
// holds the registers of one phase only.  The reductions of the epoch are finished by joint_epoch_finish_kernel.
template <class C, bool AUX = false, int PHASE = 0>
// (Column phases built for two workgroups per CU - 128 registers, 31 to 74 of them spilled - measured slower than one at every
//  split of the C5 shard: 367 - 405 us per iteration against 338.)
__global__ __launch_bounds__(C::NTHR) void joint_epoch_kernel(JointArgs A) {
  static_assert(PHASE == 0 || (C::GSPEC && !AUX), "one phase per launch: spectrum in global memory");
  if (A.upd_signal && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)   // (JointArgs: the update before this launch is complete)
    __hip_atomic_store(A.upd_signal, A.upd_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (A.chain_flag && (int)blockIdx.x == A.wait_block) {   // (JointArgs: the extra block that waits for the regulariser chain)
    if (blockIdx.y == 0 && threadIdx.x == 0) {
      int spins = 0;
      while ((int)(__hip_atomic_load(A.chain_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.chain_seq) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (++spins > (1 << 21)) {
          __hip_atomic_store(A.chain_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    return;
  }
  // PHASE = 7, the CLUSTER form: all six phases in ONE launch by `cl_parts` workgroups per epoch that share the phases' rows /
  // columns / pixels exactly as the phased launches do, with an arrival counter per epoch where those have a launch boundary
  // (cluster_sync) and the spectrum handed over through write-through stores and L1-bypassing loads (xwg_*).  Built for
  // fits that leave most CUs idle with one workgroup per epoch (a sharded C4: 25 epochs per GPU): every phase becomes ONE
  // sweep of one wave per SIMD.  Grid: 1-D, 8 * ceil(E / 8) * cl_parts blocks; block b serves epoch 8 (b / 8 / parts) + b % 8,
  // part (b / 8) % parts, so that the workgroups of an epoch have equal b % 8 - one XCD under the observed round-robin
  // placement (speed only: correctness does not depend on where a block lands).  All workgroups of an epoch must be
  // resident together: the host launches this form only when the whole grid fits on the device, every wait is bounded.
  constexpr bool CL = (PHASE == 7);
  constexpr bool ALLPH = (PHASE == 0 || PHASE == 7);
  static_assert(!CL || !C::TILECOLS, "cluster form: element-wise column access");
  // workgroup of the epoch, workgroups per epoch.  (0, 1 in the one-kernel form: compile-time constants for the LDS-spectrum
  // kernels; the global-spectrum build reads them from the grid there too - with run-time sweep strides the compiler keeps
  // fewer values in flight and its 256 registers hold everything, with constants it spilled 64 - 110 of them.)
  constexpr bool ONE = (PHASE == 0) && !C::GSPEC;
  const int part = CL ? (int)((blockIdx.x >> 3) % (unsigned)A.cl_parts) : (ONE ? 0 : (int)blockIdx.y);
  const int nparts = CL ? A.cl_parts : (ONE ? 1 : (int)gridDim.y);
  const int e = CL ? (int)(8 * ((blockIdx.x >> 3) / (unsigned)A.cl_parts) + (blockIdx.x & 7)) : (int)blockIdx.x;
  if constexpr (CL) {
    if (e >= A.E) return;  // (the grid is rounded up to whole groups of eight epochs: no such block takes part in anything)
  }
  constexpr int N = C::N, SS = C::SS, L = C::L, n = C::n, KH = C::KH, KS = C::KS, CREF = C::CREF;
  extern __shared__ __align__(16) float2 lds2[];
  float2 *SPEC = C::GSPEC ? (A.spec + (size_t)e * N * KS) : (lds2 + C::OFF_SPEC);
  constexpr bool LITE = (PHASE == 2 || PHASE == 4 || PHASE == 6);  // (JointCfg::LDS_LITE)
  float2 *TW = lds2 + (LITE ? 0 : C::OFF_TW);
  float *RED = (float *)(lds2 + (LITE ? C::LITE_RED : C::OFF_RED));
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // spectrum scratch accessors: plain, except where the workgroups of an epoch meet inside one launch
  bool same_xcd = false;  // cluster form: all workgroups of this epoch share an XCD (found out at the first sync)
  auto SLD = [&](int idx) { return xwg_load<CL>(SPEC + idx); };
  auto SST = [&](int idx, float2 v) { xwg_store<CL>(SPEC + idx, v, same_xcd); };
  int *CLST = (int *)(RED + (C::NW + 1) * (4 + 3 * kMaxSources));  // (two words inside SZ_RED's padding)
  unsigned int my_xcc = 0;
  if constexpr (CL) my_xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 15u;
  // end of a phase: workgroup barrier, or the sync over the epoch's workgroups (false: a wait was given up - leave);
  // k = 0: the launch's first sync, at the end of the prologue (nothing handed over yet: it exchanges the XCC ids)
  auto phase_end = [&](int k) -> bool {
    if constexpr (CL) {
      LC_JSTAMP(16 + k);
      const bool ok = cluster_sync(A.cl_ctr + (size_t)e * kClStride, part, nparts, (A.cl_base + (unsigned int)(k + 1)) & 0x0fffffffu,
                                   my_xcc, same_xcd, A.cl_abort, tid, CLST, k == 0);
      if (k == 0) same_xcd = CLST[1] != 0;
#ifdef LC_STAMPS
      if (k == 0 && blockIdx.x == 0 && threadIdx.x == 0) g_jstamps[30] = same_xcd ? 1 : 0;
#endif
      return ok;
    } else {
      __syncthreads();
      return true;
    }
  };
  const int pw = part * C::NW + wid, PWS = nparts * C::NW;  // this wave among the epoch's waves, their number
  const int M = A.M;
  const float c0 = (N - 1) * 0.5f;
  const float al = A.alpha[e] * 0.017453292519943295f;
  const float ca = cosf(al), sa = sinf(al);
  const float dxe = A.dx[e], dye = A.dy[e];
  const float sdx = SS * dxe, sdy = SS * dye;
  const float meane = A.mean[e];
  float *GX = (float *)(lds2 + C::OFF_TAB), *GY = GX + kMaxSources * N, *PSX = GY + kMaxSources * N, *PSY = PSX + kMaxSources;
  float *AMPL = PSY + kMaxSources;  // flux of source i in this epoch (mode 2: indicator of the source whose Fisher term is wanted)
  constexpr float inv_s2 = 1.0f / (kSigmaG * kSigmaG);

  LC_JSTAMP(0);
  // (measured and left out: s_setprio 3 here, so that the epoch kernel's waves win the issue arbitration against the
  //  regulariser chain's on a shared SIMD - C5 shard 251.1 against 251.4 us per iteration: the interference is not there)
  if constexpr (PHASE <= 1) {
    if (A.skip_D && tid == 0 && part == 0) {
      A.tshift[2 * e] = dxe;
      A.tshift[2 * e + 1] = dye;
    }
  }
  for (int k = tid; k < L; k += C::NTHR) TW[k] = A.twid[k];
  if constexpr (C::FOLD && !LITE) {
    float2 *TWH = lds2 + C::OFF_TWH, *PHI = TWH + L / 2;
    for (int k = tid; k <= L / 2; k += C::NTHR) {
      if (k < L / 2) TWH[k] = A.twid[2 * k];
      const float2 sh = A.twid[(k * CREF) % L], on = A.twid[k];  // exp(-2 pi i k CREF / L), exp(-2 pi i k / L)
      PHI[k] = cmul(make_float2(sh.x, -sh.y), make_float2(1.f + on.x, -on.y));
    }
  }
  // separable Gaussian factors of every point source (full grid, as the oracle evaluates them), in LDS: the scene and
  // gradient loops read them per pixel, and a global table put a load latency into every one of those reads
  if constexpr (!LITE) {
    const float nrm = 0.3989422804014327f / kSigmaG;
    for (int idx = tid; idx < M * N; idx += C::NTHR) {
      const int i = idx / N, p = idx % N;
      const float X = c0 + SS * (ca * A.cx[i] - sa * A.cy[i] + dxe);
      const float Y = c0 + SS * (sa * A.cx[i] + ca * A.cy[i] + dye);
      const float tx = (float)p - X, ty = (float)p - Y;
      GX[i * N + p] = nrm * expf(-0.5f * tx * tx * inv_s2);
      GY[i * N + p] = nrm * expf(-0.5f * ty * ty * inv_s2);
      if (p == 0) {
        PSX[i] = X;
        PSY[i] = Y;
      }
    }
  }
  if (!LITE && tid < kMaxSources) AMPL[tid] = (tid < M) ? ((A.mode == 2) ? ((tid == A.isrc) ? 1.f : 0.f) : A.a[e * M + tid]) : 0.f;
  {  // per-wave reduction slots: the gradient sums of the point sources are accumulated there sweep by sweep
    constexpr int NQ0 = 4 + 3 * kMaxSources;
    if (lane < NQ0) RED[wid * NQ0 + lane] = 0.f;
  }
  if constexpr (CL) {
    if (!phase_end(0)) return;
  } else {
    __syncthreads();
  }
  const bool use_h = A.h_active && A.mode != 2;

  LC_JSTAMP(1);
  // Every FFT below is a quarter-wave register transform (fft_device.h): each group of 16 lanes owns one row pair
  // or one spectrum column, so a wave works on four of them side by side.
  constexpr int LPF = C::LPF, GPW = C::GPW;
  constexpr int N2 = L / LPF;
  // l16: the lane's index inside its transform (fft_device.h: not the lane number, the exchanges are DPP involutions)
  // (l16, qid: index inside the transform and transform of the wave - 16 lanes / four transforms except on the longest grids)
  const int l16 = fft_index_n<LPF>(lane & (LPF - 1)), qid = lane / LPF, qbase = lane & (64 - LPF);
  const int kbase = N2 * bitrev_n<LPF>(l16);                    // first bin of this lane in the block layout
  const int lane_mirror = qbase | fft_index_n<LPF>(LPF - 1 - l16);  // holds bins L - k for k2 != 0
  const int lane_neg = qbase | fft_index_n<LPF>(bitrev_n<LPF>((LPF - bitrev_n<LPF>(l16)) & (LPF - 1)));  // bin (L - k) mod L, k2 == 0
  float2 *wsq = lds2 + C::OFF_WS + (C::WSQ ? (wid * GPW + qid) : wid) * L;  // linear workspace (L samples)
  // Spectrum rows in global memory: a lane of a row transform holds bins N2 b + k2 (b = its block index), so in bin order one
  // store instruction wrote 8-byte elements N2 apart - a different cache line per lane, and with 200+ workgroups the
  // request rate of L2 is what the row phases wait for.  The row is therefore kept in SLOT order: bin N2 b + k2 < L/2 sits
  // at slot k2 LPF/2 + b (the Nyquist bin stays at L/2), which makes the lanes of one instruction neighbours.  The column
  // phases walk slots and only need the bin of a slot to find its PSF spectrum.
  constexpr bool SLOTS = C::GSPEC && (N2 * (LPF / 2) == L / 2);
  const int bidx = bitrev_n<LPF>(l16);
  auto slot_of = [&](int k) { return SLOTS ? ((k < L / 2) ? (k % N2) * (LPF / 2) + k / N2 : L / 2) : k; };
  auto bin_of = [&](int s) { return SLOTS ? ((s < L / 2) ? (s % (LPF / 2)) * N2 + s / (LPF / 2) : L / 2) : s; };
  // slot of bin kbase + k2 <= L/2 of this lane, and of bin L - (kbase + k2) for kbase + k2 > L/2 (k2 is a compile-time index)
  auto slot_own = [&](int k2) { return SLOTS ? ((bidx == LPF / 2) ? L / 2 : k2 * (LPF / 2) + bidx) : kbase + k2; };
  auto slot_neg = [&](int k2) {
    return SLOTS ? ((k2 == 0) ? LPF - bidx : (N2 - k2) * (LPF / 2) + (LPF - 1 - bidx)) : L - (kbase + k2);
  };

  // Translated epochs (alpha = 0: every fit of the reference): scene pixel (u, v) samples h at (u + iyc + fyc, v + ixc + fxc)
  // with ONE integer offset and ONE pair of fractional weights for the whole epoch, so scene rows u0, u0 + 1 read the
  // three rows clamp(u0 + iyc + {0, 1, 2}) of h.  The quarter-wave copies those into LDS (coalesced 16-byte loads) and
  // interpolates from there: 6 LDS reads and 10 multiply-adds per pixel pair instead of the general path's two floor /
  // clamp / gather sequences (the scene phases are bound by instruction issue, not by memory).  Phase D's adjoint
  // stencil uses the same constants, which makes it the exact transpose of this interpolation.
  const bool translated = (sa == 0.f);
  const float t_nsx = -sdx, t_nsy = -sdy;
  const float t_ixf = floorf(t_nsx), t_iyf = floorf(t_nsy);
  const float fxc = t_nsx - t_ixf, fyc = t_nsy - t_iyf;
  const int ixc = (int)t_ixf, iyc = (int)t_iyf;
  float *hrow = C::WSQ ? (float *)wsq : (float *)(lds2 + C::OFF_HROW) + (wid * GPW + qid) * 3 * N;
  constexpr int HMASK = ((N & (N - 1)) == 0) ? N - 1 : -1;
  const int hskew = ((N & (N - 1)) == 0) ? 16 * qid : 0;  // the four quarters of a wave on different banks
  // The rows travel in two steps so that a sweep can request the rows of the NEXT one before its own transforms and
  // find them in registers afterwards (fetch_rows), then drop them into LDS once the readers of the previous contents
  // are done (put_rows): the load latency hides behind a whole sweep.
  constexpr bool HPIPE = (N % (4 * LPF) == 0);
  constexpr int NPRE = HPIPE ? 3 * (N / (4 * LPF)) : 1;
  auto fetch_rows = [&](int u0, bool active, float4 (&pre)[NPRE]) {
    if constexpr (HPIPE) {  // (the row index is clamped: an idle group loads valid rows it never uses)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float *src = A.h + (size_t)min(max(u0 + iyc + r, 0), N - 1) * N;
#pragma unroll
        for (int t = 0; t < N / (4 * LPF); ++t) pre[r * (N / (4 * LPF)) + t] = *(const float4 *)&src[4 * (l16 + LPF * t)];
      }
    }
  };
  auto put_rows = [&](int u0, bool active, const float4 (&pre)[NPRE]) {
    if (active) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if constexpr (HPIPE) {
#pragma unroll
          for (int t = 0; t < N / (4 * LPF); ++t) {
            const int x = 4 * (l16 + LPF * t);
            *(float4 *)&hrow[r * N + ((x + hskew) & HMASK)] = pre[r * (N / (4 * LPF)) + t];
          }
        } else {
          const float *src = A.h + (size_t)min(max(u0 + iyc + r, 0), N - 1) * N;
#pragma unroll
          for (int t = 0; t < N / LPF; ++t) {
            const int x = l16 + LPF * t;
            hrow[r * N + ((x + hskew) & HMASK)] = src[x];
          }
        }
      }
    }
  };
  auto stage_rows = [&](int u0, bool active) {
    if (active) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float *src = A.h + (size_t)min(max(u0 + iyc + r, 0), N - 1) * N;
        if constexpr (HPIPE) {
#pragma unroll
          for (int t = 0; t < N / (4 * LPF); ++t) {
            const int x = 4 * (l16 + LPF * t);
            *(float4 *)&hrow[r * N + ((x + hskew) & HMASK)] = *(const float4 *)&src[x];
          }
        } else {
#pragma unroll
          for (int t = 0; t < N / LPF; ++t) {
            const int x = l16 + LPF * t;
            hrow[r * N + ((x + hskew) & HMASK)] = src[x];
          }
        }
      }
    }
  };
  // x-interpolated samples of the three staged rows at scene column v, and their x-differences
  auto rows_at = [&](int v, float (&top)[3], float (&dif)[3]) {
    const int x0 = v + ixc;
    const int xa = (min(max(x0, 0), N - 1) + hskew) & HMASK, xb = (min(max(x0 + 1, 0), N - 1) + hskew) & HMASK;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float ha = hrow[r * N + xa], hb = hrow[r * N + xb];
      dif[r] = hb - ha;
      top[r] = fmaf(fxc, dif[r], ha);
    }
  };

  // two real rows -> two half spectra (2-for-1): Z = FFT(x1 + i x2), X1 = (Z[k] + conj Z[L-k]) / 2, X2 = (Z[k] - conj Z[L-k]) / 2i
  auto unpack_rows = [&](float2 (&x)[N2], int u0, bool active) {
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) {
      const float2 zk = x[k2];
      const float2 zc = (k2 == 0) ? shfl2(x[0], lane_neg) : shfl2(x[(N2 - k2) % N2], lane_mirror);
      const int k = kbase + k2;
      if (active && k <= L / 2) {
        const int sk = slot_own(k2);
        SST(u0 * KS + sk, make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y)));
        SST((u0 + 1) * KS + sk, make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x)));
      }
    }
  };
  // two half spectra -> Z[k] = X1[k] + i X2[k] over all L bins (Hermitian extension), block layout
  // (One address per lane and element, then both loads and a sign: with the two halves of the spectrum in the two arms of a
  //  branch the loads of an arm were consumed inside it, so every element cost the wave two dependent memory round trips -
  //  24 per row pair; from a single block all of them are in flight at once.  Multiplying by +-1 is exact: same values.)
  auto pack_rows = [&](float2 (&x)[N2], int u0, bool active) {
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) {
      const int k = kbase + k2;
      const bool lo = (k <= L / 2);
      const int sidx = lo ? slot_own(k2) : slot_neg(k2);
      const float sgn = lo ? -1.f : 1.f;
      float2 z = make_float2(0.f, 0.f);
      if constexpr (CL) {  // (the cluster form's L1-bypassing loads are not speculated: unconditional, from a valid row, then selected)
        const int uc = active ? u0 : 0;
        const float2 x1 = SLD(uc * KS + sidx), x2 = SLD((uc + 1) * KS + sidx);
        z = make_float2(active ? fmaf(sgn, x2.y, x1.x) : 0.f, active ? fmaf(-sgn, x1.y, x2.x) : 0.f);
      } else if (active) {
        const float2 x1 = SLD(u0 * KS + sidx), x2 = SLD((u0 + 1) * KS + sidx);
        z = make_float2(fmaf(sgn, x2.y, x1.x), fmaf(-sgn, x1.y, x2.x));
      }
      x[k2] = z;
    }
  };

  // Column passes: spectrum columns 0 (DC) and L/2 (Nyquist) of real rows are purely real, so the two travel
  // as ONE complex column (real part = column 0, imaginary part = column L/2) and are separated again in the
  // frequency domain by Hermitian symmetry.  That leaves L/2 column transforms instead of L/2 + 1: with 4 * NW
  // transforms per sweep (a power of two) the odd one out used to cost a whole extra sweep.
  constexpr int NCOL = L / 2;
  // Binned rows (FOLD): the residual spectra phase C hands to phase B' are the same for scene rows 2 I and 2 I + 1, so only
  // the even row is stored and the adjoint column loads (the ones with row_off = CREF) read row r & ~1 - half the stores of
  // phase C; the tile form also fetches half the rows.
  constexpr int RMASK_ADJ = C::FOLD ? ~1 : ~0;
  auto load_column = [&](float2 (&x)[N2], int kc, bool active, int row_off) {
    const int rmask = (row_off != 0) ? RMASK_ADJ : ~0;
    if constexpr (CL) {
      // cluster form: the loads bypass L1 and are not speculated by the compiler - with a test around each, every element was a
      // branch (and the packed DC / Nyquist column a dependent round trip per element): all loads of the column unconditional,
      // from a clamped row, the Nyquist column for the whole wave that holds the DC column (wave-uniform test)
      const bool dcwave = (kc - qid) == 0;
#pragma unroll
      for (int n2 = 0; n2 < N2; ++n2) {
        const int r = l16 + LPF * n2 - row_off;
        const bool ok = active && r >= 0 && r < N;
        const float2 v = SLD((ok ? (r & rmask) : 0) * KS + kc);
        x[n2] = make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
      }
      if (dcwave) {
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) {
          const int r = l16 + LPF * n2 - row_off;
          const bool ok = active && r >= 0 && r < N;
          const float2 w = SLD((ok ? (r & rmask) : 0) * KS + L / 2);
          if (ok && kc == 0) x[n2].y = w.x;
        }
      }
      return;
    }
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) {
      const int r = l16 + LPF * n2 - row_off;
      float2 v = make_float2(0.f, 0.f);
      if (active && r >= 0 && r < N) {
        v = SLD((r & rmask) * KS + kc);
        if (kc == 0) v.y = SLD((r & rmask) * KS + L / 2).x;
      }
      x[n2] = v;
    }
  };
  auto store_column = [&](const float2 (&x)[N2], int kc, bool active, int row_off) {
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) {
      const int r = l16 + LPF * n2 - row_off;
      if (active && r >= 0 && r < N) {
        if (kc == 0) {
          SST(r * KS, make_float2(x[n2].x, 0.f));
          SST(r * KS + L / 2, make_float2(x[n2].y, 0.f));
        } else {
          SST(r * KS + kc, x[n2]);
        }
      }
    }
  };
  // Global spectrum with two transforms per wave (N = 256): the wave's two columns are neighbours, so a lane moves 16 bytes
  // (both columns of one row) per access instead of 8: lane of group g takes the rows with n2 = 2 k + g, keeps its own
  // column's element and hands the other one to lane ^ 32 — a third of the memory instructions and cache-line requests
  // of the element-wise form (each lane of a column access touches a different line: rows are 1.5 KB apart).
  typedef float lc_f4a8 __attribute__((ext_vector_type(4), aligned(8)));
  constexpr bool PAIRCOL = C::GSPEC && (GPW == 2) && (N2 % 2 == 0);
  auto pair_load_columns = [&](float2 (&x)[N2], int c0, bool active, int row_off) {
    const int rmask = (row_off != 0) ? RMASK_ADJ : ~0;
#pragma unroll
    for (int k = 0; k < N2 / 2; ++k) {
      const int r = l16 + LPF * (2 * k + qid) - row_off;
      lc_f4a8 v = {0.f, 0.f, 0.f, 0.f};
      if (active && r >= 0 && r < N) v = *(const lc_f4a8 *)(SPEC + (r & rmask) * KS + c0);
      const float2 own = qid ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
      const float2 oth = qid ? make_float2(v.x, v.y) : make_float2(v.z, v.w);
      const float2 rec = make_float2(__shfl_xor(oth.x, 32, 64), __shfl_xor(oth.y, 32, 64));
      x[2 * k] = qid ? rec : own;
      x[2 * k + 1] = qid ? own : rec;
    }
  };
  auto pair_store_columns = [&](const float2 (&x)[N2], int c0, bool active, int row_off) {
#pragma unroll
    for (int k = 0; k < N2 / 2; ++k) {
      const int r = l16 + LPF * (2 * k + qid) - row_off;
      const float2 xo = qid ? x[2 * k + 1] : x[2 * k], snd = qid ? x[2 * k] : x[2 * k + 1];
      const float2 rec = make_float2(__shfl_xor(snd.x, 32, 64), __shfl_xor(snd.y, 32, 64));
      if (active && r >= 0 && r < N) {
        lc_f4a8 o;
        if (qid) o = (lc_f4a8){rec.x, rec.y, xo.x, xo.y};
        else o = (lc_f4a8){xo.x, xo.y, rec.x, rec.y};
        *(lc_f4a8 *)(SPEC + r * KS + c0) = o;
      }
    }
  };
  // LDS-tile form of the column access (GSPEC kernels built with TILECOLS; column_sweeps below): columns kt .. kt + CPS - 1 of
  // the global spectrum <-> tile, by every thread (the DC / Nyquist pair is packed into column 0 on the way in and separated
  // on the way out)
  float2 *TILE = lds2 + (LITE ? C::LITE_TILE : C::OFF_TILE);
  constexpr int CPS = C::CPS, TP = C::TP;
  constexpr int TPT = (N * CPS + C::NTHR - 1) / C::NTHR;  // tile elements per thread
  // x (block layout) *= PSF spectrum of column kc (conjugated for the adjoint); the packed column first splits
  // into its two Hermitian parts, each multiplied by its own spectrum column, and is packed again
  auto times_spectrum = [&](float2 (&x)[N2], const float2 (&sv)[N2], const float2 *Ste, int kc, bool conj) {
    // sv: spectrum column kc, requested before the forward transform so that its latency hides behind it
    if (kc == 0) {  // one quarter of one wave, once per phase
      // (the split below pairs lanes of opposite index parity: give the odd lanes their sign back for it, and take it away
      //  again afterwards - the transforms around this function run in their ODDNEG form)
      const float osg = (l16 & 1) ? -1.f : 1.f;
#pragma unroll
      for (int k2 = 0; k2 < N2; ++k2) x[k2] = make_float2(osg * x[k2].x, osg * x[k2].y);
      // bin k2 of this lane and bin N2 - k2 of its mirror lane are each other's conjugate partners: the two
      // registers of such a pair are read (own value + partner's value by shuffle) before either is overwritten,
      // so the split / multiply / re-pack runs in place
      auto one = [&](float2 zk, float2 zc, int k2) {
        const float2 p = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
        const float2 q = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
        const float2 sn = Ste[(size_t)(L / 2) * L + kbase + k2];
        const float2 pp = conj ? cmul_conj(p, sv[k2]) : cmul(p, sv[k2]);
        const float2 qq = conj ? cmul_conj(q, sn) : cmul(q, sn);
        return make_float2(pp.x - qq.y, pp.y + qq.x);
      };
      x[0] = one(x[0], shfl2(x[0], lane_neg), 0);
#pragma unroll
      for (int k2 = 1; 2 * k2 < N2; ++k2) {
        const float2 za = x[k2], zb = x[N2 - k2];
        const float2 ca = shfl2(zb, lane_mirror), cb = shfl2(za, lane_mirror);  // partners of za and of zb
        x[k2] = one(za, ca, k2);
        x[N2 - k2] = one(zb, cb, N2 - k2);
      }
      if constexpr (N2 % 2 == 0) x[N2 / 2] = one(x[N2 / 2], shfl2(x[N2 / 2], lane_mirror), N2 / 2);
#pragma unroll
      for (int k2 = 0; k2 < N2; ++k2) x[k2] = make_float2(osg * x[k2].x, osg * x[k2].y);
    } else {
#pragma unroll
      for (int k2 = 0; k2 < N2; ++k2) x[k2] = conj ? cmul_conj(x[k2], sv[k2]) : cmul(x[k2], sv[k2]);
    }
  };

  // Column phase of a phased launch in its LDS-tile build: the tile of the NEXT sweep is requested before this sweep's
  // transforms (16 registers per thread in flight) and dropped into the other of two LDS tiles afterwards, so the global
  // load latency hides behind the transforms and a sweep has two workgroup barriers instead of three.
  auto column_sweeps = [&](const float2 *Ste_, bool conj, int off_in, int off_out) {
    const int kt0 = part * C::NW * GPW, kstep = PWS * GPW;
    // (adjoint pass of the binned-row builds: only the even rows exist - half the tile is fetched, row r sits at r & ~1)
    const bool half_in = C::FOLD && off_in != 0;
    const int rstep = half_in ? 2 : 1, nfetch = half_in ? (N / 2) * CPS : N * CPS, rmask = half_in ? ~1 : ~0;
    // the tile travels as 16-byte elements (two neighbouring columns of a row per lane): 8-byte accesses run at 0.54 - 0.70 of
    // the 16-byte rate (MI355X_MICROARCH.md), and the column phases of a shard move 150 - 180 MB per launch
    constexpr bool WIDE = (TPT % 2 == 0) && (CPS % 2 == 0) && (NCOL % 2 == 0) && (KS % 2 == 0);
    constexpr int TPW = WIDE ? TPT / 2 : TPT, CPW = WIDE ? CPS / 2 : CPS;
    float4 pre4[WIDE ? TPW : 1];
    float2 pre[WIDE ? 1 : TPT];
    auto fetch = [&](int kt) {
      if constexpr (WIDE) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
          const int i = tid + q * C::NTHR, r = rstep * (i / CPW), c = 2 * (i % CPW), kc = kt + c;
          pre4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (2 * i < nfetch && kc < NCOL) pre4[q] = *(const float4 *)(SPEC + r * KS + kc);
        }
        if (kt == 0) {
#pragma unroll
          for (int q = 0; q < TPW; ++q) {
            const int i = tid + q * C::NTHR, r = rstep * (i / CPW), c = 2 * (i % CPW);
            if (2 * i < nfetch && c == 0) pre4[q].y = SPEC[r * KS + L / 2].x;
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
          const int i = tid + q * C::NTHR, r = rstep * (i / CPS), c = i % CPS, kc = kt + c;
          pre[q] = make_float2(0.f, 0.f);
          if (i < nfetch && kc < NCOL) pre[q] = SPEC[r * KS + kc];
        }
        if (kt == 0) {
#pragma unroll
          for (int q = 0; q < TPT; ++q) {
            const int i = tid + q * C::NTHR, r = rstep * (i / CPS), c = i % CPS;
            if (i < nfetch && c == 0) pre[q].y = SPEC[r * KS + L / 2].x;
          }
        }
      }
    };
    if (kt0 < NCOL) fetch(kt0);
    int buf = 0;
    for (int kt = kt0; kt < NCOL; kt += kstep, buf ^= 1) {
      float2 *T = TILE + buf * C::SZ_TILE;
      const int kc = kt + wid * GPW + qid;
      const bool active = kc < NCOL;
      const int kcs = active ? kc : 1;
      float2 x[N2], sv[N2];
#pragma unroll
      for (int k2 = 0; k2 < N2; ++k2) sv[k2] = Ste_[(size_t)bin_of(kcs) * L + kbase + k2];
      if constexpr (WIDE) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
          const int i = tid + q * C::NTHR, r = rstep * (i / CPW), c = 2 * (i % CPW);
          if (2 * i < nfetch) {
            T[r * TP + c] = make_float2(pre4[q].x, pre4[q].y);
            T[r * TP + c + 1] = make_float2(pre4[q].z, pre4[q].w);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
          const int i = tid + q * C::NTHR, r = rstep * (i / CPS), c = i % CPS;
          if (i < nfetch) T[r * TP + c] = pre[q];
        }
      }
      __syncthreads();
      if (kt + kstep < NCOL) fetch(kt + kstep);
#pragma unroll
      for (int n2 = 0; n2 < N2; ++n2) {
        const int r = l16 + LPF * n2 - off_in;
        x[n2] = (active && r >= 0 && r < N) ? T[(r & rmask) * TP + wid * GPW + qid] : make_float2(0.f, 0.f);
      }
      group_fft_fwd<L, LPF, true>(x, l16, TW);
      times_spectrum(x, sv, Ste_, kcs, conj);
      group_fft_inv<L, LPF, true>(x, l16, TW);
#pragma unroll
      for (int n2 = 0; n2 < N2; ++n2) {
        const int r = l16 + LPF * n2 - off_out;
        if (active && r >= 0 && r < N) T[r * TP + wid * GPW + qid] = x[n2];
      }
      __syncthreads();
      if constexpr (WIDE) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
          const int i = tid + q * C::NTHR, r = i / CPW, c = 2 * (i % CPW), kq = kt + c;
          if (2 * i < N * CPS && kq < NCOL) {
            const float2 v = T[r * TP + c], w = T[r * TP + c + 1];
            if (kq == 0) {  // (column 0 carries the DC column in its real and the Nyquist column in its imaginary part)
              *(float4 *)(SPEC + r * KS) = make_float4(v.x, 0.f, w.x, w.y);
              SPEC[r * KS + L / 2] = make_float2(v.y, 0.f);
            } else {
              *(float4 *)(SPEC + r * KS + kq) = make_float4(v.x, v.y, w.x, w.y);
            }
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
          const int i = tid + q * C::NTHR, r = i / CPS, c = i % CPS, kq = kt + c;
          if (i < N * CPS && kq < NCOL) {
            const float2 v = T[r * TP + c];
            if (kq == 0) {
              SPEC[r * KS] = make_float2(v.x, 0.f);
              SPEC[r * KS + L / 2] = make_float2(v.y, 0.f);
            } else {
              SPEC[r * KS + kq] = v;
            }
          }
        }
      }
    }
  };
  constexpr bool PIPED = C::TILECOLS && !AUX;

  // ---- phase A: scene rows, two real rows per complex FFT ---------------------------------------
  float4 hpre[NPRE];
#pragma unroll
  for (int q = 0; q < NPRE; ++q) hpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (declared here for every phase: the accumulators and pointers a later phase or the reductions use)
  float acc_chi = 0.f, acc_mean = 0.f, acc_fis = 0.f;
  float acc_dx = 0.f, acc_dy = 0.f, acc_hx = 0.f, acc_hy = 0.f;
  const float *de = A.data + (size_t)e * n * n, *we = A.wgt + (size_t)e * n * n;
  const float2 *Ste = A.St + (size_t)e * KH * L;
  // scene-gradient rows overwrite the spectrum rows they were computed from (row u: N floats inside the
  // KH float2 of spectrum row u), so the T^T gather below reads LDS, not global memory
  float *GSl = (float *)SPEC;
  constexpr int GST = 2 * KS;
  if constexpr (ALLPH || PHASE == 1) {
  if constexpr (!AUX && C::ROWPIPE) {
    if (use_h && translated) fetch_rows(2 * (pw * GPW + qid), pw * GPW + qid < N / 2, hpre);
  }
  for (int rp0 = pw * GPW, sw = 0; rp0 < N / 2; rp0 += PWS * GPW, ++sw) {
    // (of the two waves of a SIMD the one that is a sweep behind wins the issue arbitration: starlet_device.h)
    progress_prio_end(sw, (N / 2 + PWS * GPW - 1) / (PWS * GPW));
    const int rp = rp0 + qid, u0 = 2 * rp;
    const bool active = rp < N / 2;
    float2 x[N2];
    if constexpr (!AUX) {
      if (use_h && translated) {
        wave_lds_sync();  // the previous sweep's readers are done with the buffer
        if constexpr (C::ROWPIPE) {
          put_rows(u0, active, hpre);
          wave_lds_sync();
          const int rpn = rp + PWS * GPW;
          if (rp0 + PWS * GPW < N / 2) fetch_rows(2 * rpn, rpn < N / 2, hpre);
        } else {
          stage_rows(u0, active);
          wave_lds_sync();
        }
      }
    }
    // (translated epochs: the N / LPF pixels of a lane in ONE straight block - the staged rows are read for idle groups too,
    //  their results dropped by a select - so that the LDS reads of all pixels are in flight together; with a test per
    //  pixel every pixel waited for its own reads)
    static_assert(N % LPF == 0, "pixels per lane");
    const bool fast_rows = !AUX && use_h && translated;
    if (fast_rows) {
#pragma unroll
      for (int n2 = 0; n2 < N2; ++n2) {
        float2 z = make_float2(0.f, 0.f);
        if (n2 < N / LPF) {
          float top[3], dif[3];
          rows_at(l16 + LPF * n2, top, dif);
          const float s0 = fmaf(fyc, top[1] - top[0], top[0]), s1 = fmaf(fyc, top[2] - top[1], top[1]);
          z = make_float2(active ? s0 : 0.f, active ? s1 : 0.f);
        }
        x[n2] = z;
      }
    }
    if (!fast_rows) {
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) {
      const int v = l16 + LPF * n2;
      float2 z = make_float2(0.f, 0.f);
      if constexpr (AUX) {
        if (active && v < N) {
          const float *si = A.scene_in + (size_t)e * N * N;
          z = make_float2(si[u0 * N + v], si[(u0 + 1) * N + v]);
        }
      } else if (active && v < N) {
        float s0 = 0.f, s1 = 0.f;
        if (use_h) {
          float Xs, Ys, t0, t1;
          sample_coords(u0, v, c0, ca, sa, sdx, sdy, Xs, Ys);
          s0 += bilinear_h<N>(A.h, Xs, Ys, t0, t1);
          sample_coords(u0 + 1, v, c0, ca, sa, sdx, sdy, Xs, Ys);
          s1 += bilinear_h<N>(A.h, Xs, Ys, t0, t1);
        }
        z = make_float2(s0, s1);
      }
      x[n2] = z;
    }
    }
    if constexpr (!AUX) {
      // point sources: flux times the row factor of the quarter's two rows, then one LDS read per pixel pair
#pragma unroll
      for (int i = 0; i < kMaxSources; ++i) {
        if (i < M && active) {
          const float ai = AMPL[i];
          const float ay0 = ai * GY[i * N + u0], ay1 = ai * GY[i * N + u0 + 1];
#pragma unroll
          for (int n2 = 0; n2 < N / LPF; ++n2) {
            const float gx = GX[i * N + l16 + LPF * n2];
            x[n2].x = fmaf(ay0, gx, x[n2].x);
            x[n2].y = fmaf(ay1, gx, x[n2].y);
          }
        }
      }
    }
    group_fft_fwd<L, LPF>(x, l16, TW);
    unpack_rows(x, u0, active);
  }
  }  // phase A
  if (!phase_end(1)) return;
  LC_JSTAMP(2);
  if constexpr (AUX) {
    if (A.mode == 4) {  // forward column transforms only: the spectrum of scene_in[e], transposed, divided by L^2
      float2 *So = A.St_out + (size_t)e * KH * L;
      const float sc = 1.0f / ((float)L * (float)L);
      for (int kc0 = wid * GPW; kc0 < NCOL; kc0 += C::NW * GPW) {
        const int kc = kc0 + qid;
        const bool active = kc < NCOL;
        const int kcs = active ? kc : 1;
        float2 x[N2];
        load_column(x, kcs, active, 0);
        group_fft_fwd<L, LPF>(x, l16, TW);
        if (kcs == 0) {  // packed pair: column 0 = Hermitian part, column L/2 = anti-Hermitian part / i
#pragma unroll
          for (int k2 = 0; k2 < N2; ++k2) {
            const float2 zk = x[k2];
            const float2 zc = (k2 == 0) ? shfl2(x[0], lane_neg) : shfl2(x[(N2 - k2) % N2], lane_mirror);
            So[kbase + k2] = make_float2(0.5f * sc * (zk.x + zc.x), 0.5f * sc * (zk.y - zc.y));
            So[(size_t)(L / 2) * L + kbase + k2] = make_float2(0.5f * sc * (zk.y + zc.y), -0.5f * sc * (zk.x - zc.x));
          }
        } else if (active) {
#pragma unroll
          for (int k2 = 0; k2 < N2; ++k2) So[(size_t)bin_of(kcs) * L + kbase + k2] = make_float2(sc * x[k2].x, sc * x[k2].y);
        }
      }
      return;
    }
  }
  __builtin_amdgcn_s_setprio(0);
  // ---- phase B: columns: FFT, multiply by the PSF spectrum, inverse FFT, keep the 'same' window ----
  if constexpr (PIPED && (ALLPH || PHASE == 2)) {
    column_sweeps(Ste, false, 0, CREF);
  } else if constexpr (ALLPH || PHASE == 2) {
  for (int kt = part * C::NW * GPW, sw = 0; kt < NCOL; kt += PWS * GPW, ++sw) {  // one sweep: NW * GPW consecutive columns
    const int kc0 = kt + wid * GPW;
    if (kc0 >= NCOL) break;
    progress_prio_end(sw, (NCOL + PWS * GPW - 1) / (PWS * GPW));
    const int kc = kc0 + qid;
    const bool active = kc < NCOL;
    const int kcs = active ? kc : 1;
    float2 x[N2], sv[N2];
    LC_JSTAMP(9);
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) sv[k2] = Ste[(size_t)bin_of(kcs) * L + kbase + k2];
    const bool paired = PAIRCOL && kc0 != 0;  // (the packed DC / Nyquist column takes the element-wise form)
    if (paired) {
      pair_load_columns(x, kc0, active, 0);
    } else {
      load_column(x, kcs, active, 0);
    }
    LC_JSTAMP(10);
    group_fft_fwd<L, LPF, true>(x, l16, TW);
    LC_JSTAMP(11);
    times_spectrum(x, sv, Ste, kcs, false);
    LC_JSTAMP(12);
    group_fft_inv<L, LPF, true>(x, l16, TW);
    LC_JSTAMP(13);
    if (paired) {
      pair_store_columns(x, kc0, active, CREF);
    } else {
      store_column(x, kcs, active, CREF);
    }
    LC_JSTAMP(14);
  }
  }  // phase B
  __builtin_amdgcn_s_setprio(0);
  if (!phase_end(2)) return;
  LC_JSTAMP(3);
  if constexpr (AUX) {  // mode 3: inverse rows, 'same' window of the convolution at full resolution
    float *co = A.conv_out + (size_t)e * N * N;
    for (int rp0 = wid * GPW; rp0 < N / 2; rp0 += C::NW * GPW) {
      const int rp = rp0 + qid, u0 = 2 * rp;
      const bool active = rp < N / 2;
      float2 x[N2];
      pack_rows(x, u0, active);
      group_fft_inv<L, LPF>(x, l16, TW);
#pragma unroll
      for (int n2 = 0; n2 < N2; ++n2) {
        const int v = l16 + LPF * n2 - CREF;
        if (active && v >= 0 && v < N) {
          co[u0 * N + v] = x[n2].x;
          co[(u0 + 1) * N + v] = x[n2].y;
        }
      }
    }
    return;
  }
  // ---- phase C: inverse rows -> model, residuals; forward rows of the up-sampled weighted residual ----
  if constexpr (ALLPH || PHASE == 3) {
  if constexpr (C::FOLD) {
    // Binned rows.  Data row I is the sum of scene rows 2 I, 2 I + 1 (add their half spectra), shifted by CREF, added to
    // its right neighbour and decimated by two along x: in Fourier space a multiplication by phi and the fold
    // D[k'] = G[k'] + G[k' + L/2].  So ONE transform of length L / 2 per pair of DATA rows returns the model at the data
    // resolution (half as many transforms as scene row pairs, half as long), the residuals are formed in registers, and
    // their spectrum, periodically extended and multiplied by conj(phi), is the spectrum of the up-sampled residual rows.
    constexpr int LH = L / 2, N2H = LH / LPF, KQ = LH / 2;
    const float2 *TWH = lds2 + C::OFF_TWH, *PHI = TWH + LH;
    const int kbh = N2H * bitrev_n<LPF>(l16);
    auto folded = [&](int r0, int m) {  // D[m], 0 <= m <= KQ, of the data row made of scene rows r0, r0 + 1
      const int sa_ = slot_of(m), sb_ = slot_of(LH - m);
      const float2 a0 = SLD(r0 * KS + sa_), a1 = SLD((r0 + 1) * KS + sa_);
      const float2 b0 = SLD(r0 * KS + sb_), b1 = SLD((r0 + 1) * KS + sb_);
      const float2 ga = cmul(make_float2(a0.x + a1.x, a0.y + a1.y), PHI[m]);
      const float2 gb = cmul(make_float2(b0.x + b1.x, b0.y + b1.y), PHI[LH - m]);
      return make_float2(ga.x + gb.x, ga.y - gb.y);
    };
    for (int t0 = pw * GPW, sw = 0; t0 < n / 2; t0 += PWS * GPW, ++sw) {
      progress_prio_end(sw, (n / 2 + PWS * GPW - 1) / (PWS * GPW));
      const int t = t0 + qid, I0 = 2 * t, I1 = I0 + 1;
      const bool active = t < n / 2;
      float2 x[N2H];
#pragma unroll
      for (int k2 = 0; k2 < N2H; ++k2) {
        const int k = kbh + k2, m = (k <= KQ) ? k : LH - k;
        float2 z = make_float2(0.f, 0.f);
        if (CL || active) {  // (cluster form: unconditional L1-bypassing loads from valid rows, the result selected)
          const int J0 = (CL && !active) ? 0 : I0, J1 = (CL && !active) ? 1 : I1;
          float2 d0 = folded(2 * J0, m), d1 = folded(2 * J1, m);
          if (k > KQ) {  // Hermitian extension
            d0.y = -d0.y;
            d1.y = -d1.y;
          }
          z = make_float2(d0.x - d1.y, d0.y + d1.x);  // row I0 + i row I1
          if (CL && !active) z = make_float2(0.f, 0.f);
        }
        x[k2] = z;
      }
      // data and weights of the two rows: requested before the transform (used inside the per-pixel test below, each pixel
      // would wait for its own four loads)
      constexpr int NDH = (n + LPF - 1) / LPF;
      float pw0[NDH], pw1[NDH], pd0[NDH], pd1[NDH];
#pragma unroll
      for (int t = 0; t < NDH; ++t) {
        const int jd = l16 + LPF * t;
        const bool ok = active && jd < n;
        pw0[t] = ok ? we[I0 * n + jd] : 0.f;
        pw1[t] = ok ? we[I1 * n + jd] : 0.f;
        pd0[t] = ok ? de[I0 * n + jd] : 0.f;
        pd1[t] = ok ? de[I1 * n + jd] : 0.f;
      }
      group_fft_inv<LH, LPF>(x, l16, TWH);
#pragma unroll
      for (int n2 = 0; n2 < N2H; ++n2) {
        const int jd = l16 + LPF * n2;
        float2 rw = make_float2(0.f, 0.f);
        if (n2 < NDH && active && jd < n) {
          const float2 y = x[n2];
          const float w0 = pw0[n2 < NDH ? n2 : 0], w1 = pw1[n2 < NDH ? n2 : 0];
          if (A.mode == 2) {
            acc_fis = fmaf(w0 * y.x, y.x, acc_fis);
            acc_fis = fmaf(w1 * y.y, y.y, acc_fis);
          } else {
            const float m0 = y.x + meane, m1 = y.y + meane;
            const float r0 = m0 - pd0[n2 < NDH ? n2 : 0], r1 = m1 - pd1[n2 < NDH ? n2 : 0];
            rw = make_float2(w0 * r0, w1 * r1);
            acc_chi = fmaf(rw.x, r0, acc_chi);
            acc_chi = fmaf(rw.y, r1, acc_chi);
            acc_mean += rw.x + rw.y;
            if (A.model_out) {
              A.model_out[(size_t)e * n * n + I0 * n + jd] = m0;
              A.model_out[(size_t)e * n * n + I1 * n + jd] = m1;
            }
          }
        }
        x[n2] = rw;
      }
      if (A.mode == 0) {
        group_fft_fwd<LH, LPF>(x, l16, TWH);
#pragma unroll
        for (int k2 = 0; k2 < N2H; ++k2) {
          const float2 zk = x[k2];
          const float2 zc = (k2 == 0) ? shfl2(x[0], lane_neg) : shfl2(x[(N2H - k2) % N2H], lane_mirror);
          const int k = kbh + k2;
          if (active && k <= KQ) {
            const float2 R0 = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));   // spectrum of row I0's residual
            const float2 R1 = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));  // and of row I1's
            // scene rows 2 I, 2 I + 1 carry the same up-sampled row: columns k and LH - k of the periodic extension
            const float2 pa = PHI[k], pb = PHI[LH - k];
            const float2 u0a = cmul_conj(R0, pa), u1a = cmul_conj(R1, pa);
            const float2 u0b = cmul_conj(make_float2(R0.x, -R0.y), pb), u1b = cmul_conj(make_float2(R1.x, -R1.y), pb);
            const int ska = slot_of(k), skb = slot_of(LH - k);
            // (rows 2 I + 1 would hold the same values: the adjoint column loads read row r & ~1 instead)
            SST((2 * I0) * KS + ska, u0a);
            SST((2 * I1) * KS + ska, u1a);
            if (k != KQ) {
              SST((2 * I0) * KS + skb, u0b);
              SST((2 * I1) * KS + skb, u1b);
            }
          }
        }
      }
    }
  } else {
  for (int rp0 = pw * GPW; rp0 < N / 2; rp0 += PWS * GPW) {
    const int rp = rp0 + qid, u0 = 2 * rp;
    const bool active = rp < N / 2;
    float2 x[N2];
    pack_rows(x, u0, active);
    group_fft_inv<L, LPF>(x, l16, TW);
    constexpr int NTURN = C::WSQ ? 1 : GPW;
#pragma unroll
    for (int turn = 0; turn < NTURN; ++turn) {
    const bool mine = C::WSQ || (qid == turn);  // with a per-wave buffer the quarters take turns
    if (!C::WSQ) wave_lds_sync();
    if (mine) {
#pragma unroll
      for (int n2 = 0; n2 < N2; ++n2) wsq[l16 + LPF * n2] = x[n2];  // linear order for the data-space step
    }
    wave_lds_sync();
    constexpr int NDP = (n + LPF - 1) / LPF;  // data pixels of this quarter's row(s) per lane
    float rw0[NDP], rw1[NDP];
#pragma unroll
    for (int t = 0; t < NDP; ++t) {
      rw0[t] = rw1[t] = 0.f;
      const int jd = l16 + LPF * t;
      if (mine && active && jd < n) {
        if (SS == 2) {
          const int I = rp;
          const float2 y0 = wsq[2 * jd + CREF], y1 = wsq[2 * jd + 1 + CREF];
          const float conv = (y0.x + y1.x) + (y0.y + y1.y);
          const float w = we[I * n + jd];
          if (A.mode == 2) {
            acc_fis = fmaf(w * conv, conv, acc_fis);
          } else {
            const float model = conv + meane;
            const float res = model - de[I * n + jd];
            const float rw = w * res;
            acc_chi = fmaf(rw, res, acc_chi);
            acc_mean += rw;
            if (A.model_out) A.model_out[(size_t)e * n * n + I * n + jd] = model;
            rw0[t] = rw;
          }
        } else {
          const float2 y = wsq[jd + CREF];
          const float w0 = we[u0 * n + jd], w1 = we[(u0 + 1) * n + jd];
          if (A.mode == 2) {
            acc_fis = fmaf(w0 * y.x, y.x, acc_fis);
            acc_fis = fmaf(w1 * y.y, y.y, acc_fis);
          } else {
            const float m0 = y.x + meane, m1 = y.y + meane;
            const float r0 = m0 - de[u0 * n + jd], r1 = m1 - de[(u0 + 1) * n + jd];
            rw0[t] = w0 * r0;
            rw1[t] = w1 * r1;
            acc_chi = fmaf(rw0[t], r0, acc_chi);
            acc_chi = fmaf(rw1[t], r1, acc_chi);
            acc_mean += rw0[t] + rw1[t];
            if (A.model_out) {
              A.model_out[(size_t)e * n * n + u0 * n + jd] = m0;
              A.model_out[(size_t)e * n * n + (u0 + 1) * n + jd] = m1;
            }
          }
        }
      }
    }
    if (A.mode == 0) {
      // adjoint input rows: up-sampled weighted residuals at offset CREF, zero elsewhere
      wave_lds_sync();
      if (mine) {
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) wsq[l16 + LPF * n2] = make_float2(0.f, 0.f);
      }
      wave_lds_sync();
#pragma unroll
      for (int t = 0; t < NDP; ++t) {
        const int jd = l16 + LPF * t;
        if (mine && active && jd < n) {
          if (SS == 2) {
            wsq[2 * jd + CREF] = make_float2(rw0[t], rw0[t]);
            wsq[2 * jd + 1 + CREF] = make_float2(rw0[t], rw0[t]);
          } else {
            wsq[jd + CREF] = make_float2(rw0[t], rw1[t]);
          }
        }
      }
      wave_lds_sync();
      if (mine) {
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) x[n2] = wsq[l16 + LPF * n2];
      }
    }
    }  // turns
    if (A.mode == 0) {
      group_fft_fwd<L, LPF>(x, l16, TW);
      unpack_rows(x, u0, active);
    }
    wave_lds_sync();
  }
  }
  }  // phase C
  if constexpr (PHASE == 3 || CL) {  // this workgroup's share of chi2 and of the sky-level gradient
    constexpr int NQP = 4 + 3 * kMaxSources;
    const float s0 = wave_sum(acc_chi), s1 = wave_sum(acc_mean);
    if (lane == 0) {
      RED[wid * NQP + 0] = s0;
      RED[wid * NQP + 1] = s1;
    }
    __syncthreads();
    if (tid < 2) {
      float acc = 0.f;
      for (int w = 0; w < C::NW; ++w) acc += RED[w * NQP + tid];
      xwg_storef<CL>(&A.part[((size_t)e * nparts + part) * NQP + tid], acc, same_xcd);
    }
  }
  if (PHASE == 0 && A.mode != 0) {
    const float v0 = wave_sum((A.mode == 2) ? acc_fis : acc_chi);
    if (lane == 0) RED[wid] = v0;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int w = 0; w < C::NW; ++w) s += RED[w];
      if (A.mode == 2) A.fisher_out[e * M + A.isrc] = 1.0f / sqrtf(s);
      else A.chi2_e[e] = s;
    }
    return;
  }
  if (!phase_end(3)) return;
  LC_JSTAMP(4);
  __builtin_amdgcn_s_setprio(0);
  // ---- phase B': adjoint columns (rows sit at offset CREF), multiply by conj(spectrum) ------------
  if constexpr (PIPED && (ALLPH || PHASE == 4)) {
    column_sweeps(Ste, true, CREF, 0);
  } else if constexpr (ALLPH || PHASE == 4) {
  for (int kt = part * C::NW * GPW, sw = 0; kt < NCOL; kt += PWS * GPW, ++sw) {
    const int kc0 = kt + wid * GPW;
    if (kc0 >= NCOL) break;
    progress_prio_end(sw, (NCOL + PWS * GPW - 1) / (PWS * GPW));
    const int kc = kc0 + qid;
    const bool active = kc < NCOL;
    const int kcs = active ? kc : 1;
    float2 x[N2], sv[N2];
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) sv[k2] = Ste[(size_t)bin_of(kcs) * L + kbase + k2];
    const bool paired = PAIRCOL && kc0 != 0;
    if (paired) {
      pair_load_columns(x, kc0, active, CREF);
    } else {
      load_column(x, kcs, active, CREF);
    }
    group_fft_fwd<L, LPF, true>(x, l16, TW);
    times_spectrum(x, sv, Ste, kcs, true);
    group_fft_inv<L, LPF, true>(x, l16, TW);
    if (paired) {
      pair_store_columns(x, kc0, active, 0);
    } else {
      store_column(x, kcs, active, 0);
    }
  }
  }  // phase B'
  __builtin_amdgcn_s_setprio(0);
  if (!phase_end(4)) return;
  LC_JSTAMP(5);
  // ---- phase C': adjoint inverse rows -> scene gradient rows; parameter gradients ------------------
  if constexpr (ALLPH || PHASE == 5) {
  for (int rp0 = pw * GPW, sw = 0; rp0 < N / 2; rp0 += PWS * GPW, ++sw) {
    progress_prio_end(sw, (N / 2 + PWS * GPW - 1) / (PWS * GPW));
    const int rp = rp0 + qid, u0 = 2 * rp;
    const bool active = rp < N / 2;
    float2 x[N2];
    if (use_h && translated) {  // (no registers to spare here for rows in flight across a sweep: fetched and stored at once)
      wave_lds_sync();
      stage_rows(u0, active);
    }
    pack_rows(x, u0, active);
    group_fft_inv<L, LPF>(x, l16, TW);
    if (use_h && translated) wave_lds_sync();
    // point sources: per source the row factors of the quarter's two rows, then one LDS read and seven multiply-adds
    // per pixel pair
    // (their three sums per source leave the registers at the end of every sweep: lanes by shuffles, into the wave's slot)
#pragma unroll
    for (int i = 0; i < kMaxSources; ++i) {
      if (i < M) {
        float pa = 0.f, pX = 0.f, pY = 0.f;
        if (active) {
          const float Xi = PSX[i], Yi = PSY[i];
          const float gy0 = GY[i * N + u0], gy1 = GY[i * N + u0 + 1];
          const float dgy0 = gy0 * (((float)u0 - Yi) * inv_s2), dgy1 = gy1 * (((float)(u0 + 1) - Yi) * inv_s2);
#pragma unroll
          for (int n2 = 0; n2 < N / LPF; ++n2) {
            const int v = l16 + LPF * n2;
            const float2 g = x[n2];
            const float gx = GX[i * N + v], dgx = gx * (((float)v - Xi) * inv_s2);
            const float gg = g.x * gy0 + g.y * gy1;
            pa = fmaf(gg, gx, pa);
            pX = fmaf(gg, dgx, pX);
            pY = fmaf(g.x * dgy0 + g.y * dgy1, gx, pY);
          }
        }
        pa = wave_sum(pa);
        pX = wave_sum(pX);
        pY = wave_sum(pY);
        if (lane == 0) {
          float *slot = RED + wid * (4 + 3 * kMaxSources) + 4 + 3 * i;
          slot[0] += pa;
          slot[1] += pX;
          slot[2] += pY;
        }
      }
    }
    // (one straight block for the lane's pixels, as in phase A - in the global-spectrum builds only: the N = 128 kernel has
    //  no registers left for it, 29 spilled)
    if (C::GSPEC && use_h && translated) {
      float hx = 0.f, hy = 0.f;
#pragma unroll
      for (int n2 = 0; n2 < N / LPF; ++n2) {
        // d scene / d dx = -SS dH/dx, d scene / d dy = -SS dH/dy (the factor joins after the loop)
        const float2 g = x[n2];
        float top[3], dif[3];
        rows_at(l16 + LPF * n2, top, dif);
        hx = fmaf(g.x, fmaf(fyc, dif[1] - dif[0], dif[0]), hx);
        hx = fmaf(g.y, fmaf(fyc, dif[2] - dif[1], dif[1]), hx);
        hy = fmaf(g.x, top[1] - top[0], hy);
        hy = fmaf(g.y, top[2] - top[1], hy);
      }
      if (active) {  // (an idle group read stale rows: its sums are dropped whole)
        acc_hx += hx;
        acc_hy += hy;
        if (A.need_hgrad) {
#pragma unroll
          for (int n2 = 0; n2 < N / LPF; ++n2) {
            const int v = l16 + LPF * n2;
            xwg_storef<CL>(&GSl[u0 * GST + v], x[n2].x, same_xcd);
            xwg_storef<CL>(&GSl[(u0 + 1) * GST + v], x[n2].y, same_xcd);
          }
        }
      }
    } else {
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) {
      const int v = l16 + LPF * n2;
      if (active && v < N) {
        const float2 g = x[n2];
        if (use_h && translated) {
          float top[3], dif[3];
          rows_at(v, top, dif);
          acc_hx = fmaf(g.x, fmaf(fyc, dif[1] - dif[0], dif[0]), acc_hx);
          acc_hx = fmaf(g.y, fmaf(fyc, dif[2] - dif[1], dif[1]), acc_hx);
          acc_hy = fmaf(g.x, top[1] - top[0], acc_hy);
          acc_hy = fmaf(g.y, top[2] - top[1], acc_hy);
        } else if (use_h) {
          float Xs, Ys, hx, hy;
          sample_coords(u0, v, c0, ca, sa, sdx, sdy, Xs, Ys);
          bilinear_h<N>(A.h, Xs, Ys, hx, hy);
          acc_dx = fmaf(g.x, SS * (sa * hy - ca * hx), acc_dx);
          acc_dy = fmaf(g.x, -SS * (sa * hx + ca * hy), acc_dy);
          sample_coords(u0 + 1, v, c0, ca, sa, sdx, sdy, Xs, Ys);
          bilinear_h<N>(A.h, Xs, Ys, hx, hy);
          acc_dx = fmaf(g.y, SS * (sa * hy - ca * hx), acc_dx);
          acc_dy = fmaf(g.y, -SS * (sa * hx + ca * hy), acc_dy);
        }
        if (use_h) {
          if (A.need_hgrad) {
            xwg_storef<CL>(&GSl[u0 * GST + v], g.x, same_xcd);
            xwg_storef<CL>(&GSl[(u0 + 1) * GST + v], g.y, same_xcd);
          }
        }
      }
    }
    }
  }
  }  // phase C'
  __builtin_amdgcn_s_setprio(0);
  LC_JSTAMP(6);
  acc_dx = fmaf(-(float)SS, acc_hx, acc_dx);
  acc_dy = fmaf(-(float)SS, acc_hy, acc_dy);
  if constexpr (PHASE == 5 || CL) {  // this workgroup's share of the shift gradients and of the sums of the point sources
    constexpr int NQP = 4 + 3 * kMaxSources;
    const int nqp = 4 + 3 * M;
    const float s2 = wave_sum(acc_dx), s3 = wave_sum(acc_dy);
    if (lane == 0) {
      RED[wid * NQP + 2] = s2;
      RED[wid * NQP + 3] = s3;
    }
    __syncthreads();
    if (tid >= 2 && tid < nqp) {
      float acc = 0.f;
      for (int w = 0; w < C::NW; ++w) acc += RED[w * NQP + tid];
      xwg_storef<CL>(&A.part[((size_t)e * nparts + part) * NQP + tid], acc, same_xcd);
    }
  }
  if constexpr (CL) {
    // every workgroup's share of every sum is out, the scene gradient of the epoch complete: the first workgroup adds the
    // shares in workgroup order (the phased launches' order: same bits), all of them take their pixels of phase D
    if (!phase_end(5)) return;
    if (part == 0) joint_epoch_totals<true>(A, e, tid, SS, nparts, RED);
  }
  // reductions: lanes by shuffles, the four waves in fixed order
  if constexpr (PHASE == 0) {
    constexpr int NQ = 4 + 3 * kMaxSources;
    const int nq = 4 + 3 * M;  // quantities in use: the sums of the sources already sit in the waves' slots
    {
      const float s0 = wave_sum(acc_chi), s1 = wave_sum(acc_mean), s2 = wave_sum(acc_dx), s3 = wave_sum(acc_dy);
      if (lane == 0) {
        RED[wid * NQ + 0] = s0;
        RED[wid * NQ + 1] = s1;
        RED[wid * NQ + 2] = s2;
        RED[wid * NQ + 3] = s3;
      }
    }
    __syncthreads();
    float *TOT = RED + C::NW * NQ;
    if (tid < nq) {
      float acc = 0.f;
      for (int w = 0; w < C::NW; ++w) acc += RED[w * NQ + tid];
      TOT[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {
      const float *t = TOT;
      A.chi2_e[e] = t[0];
      A.g_mean[e] = t[1];
      float gdx = t[2], gdy = t[3];
      for (int i = 0; i < M; ++i) {
        const float ai = AMPL[i];  // = a[e][i] in this mode
        const float gX = ai * t[5 + 3 * i], gY = ai * t[6 + 3 * i];
        A.g_a[e * M + i] = t[4 + 3 * i];
        gdx += SS * gX;
        gdy += SS * gY;
        A.g_cx_e[e * M + i] = SS * (ca * gX + sa * gY);
        A.g_cy_e[e * M + i] = SS * (ca * gY - sa * gX);
      }
      A.g_dx[e] = gdx;
      A.g_dy[e] = gdy;
    }
  }
  LC_JSTAMP(7);
  // ---- phase D: T_e^T of the scene gradient by an exact, ordered gather -----------------------------
  if ((ALLPH || PHASE == 6) && use_h && A.need_hgrad && !A.skip_D) {
    if constexpr (PHASE == 6) {  // (launched with the workgroup count of the row phases, whose partial sums these are)
      if (part == 0) joint_epoch_totals(A, e, tid, SS, nparts, RED);
    }
    __syncthreads();  // scene gradient of this epoch complete in LDS
    float *HGe = A.HG + (size_t)e * N * N;
    const float asa = fabsf(sa);
    const int band = (int)ceilf(fmaxf(fabsf(sdx), fabsf(sdy)) + N * asa) + 3;
    const int mext = (int)ceilf(asa * (band + 2)) + 2;
    // pure translation: every scene pixel samples h at the same fractional offset, so an interior pixel k of
    // h collects exactly four scene pixels with constant weights (the adjoint of a 2 x 2 interpolation stencil).
    // Border pixels (and every pixel when the epoch is rotated) take the exact ordered gather instead; the two
    // sets are walked by separate loops so that no wave mixes the cheap and the expensive path.
    // Only the outermost ring of h collects clamped (edge-replicated) samples; every other pixel of a translated
    // epoch is served by the four-tap stencil, whose taps simply drop out where they leave the scene grid.
    const int marg = translated ? 1 : N / 2;
    const int NI = N - 2 * marg;
    if (translated) {
      // four consecutive pixels of a row per thread: two rows of five scene-gradient samples, one 16-byte store
      const float w00 = (1.f - fyc) * (1.f - fxc), w01 = (1.f - fyc) * fxc, w10 = fyc * (1.f - fxc), w11 = fyc * fxc;
      for (int c = part * C::NTHR + tid; c < N * (N / 4); c += nparts * C::NTHR) {
        const int ky = c / (N / 4), kx0 = 4 * (c % (N / 4));
        if (ky == 0 || ky == N - 1) continue;
        const int r0 = ky - iyc, q0 = kx0 - ixc;
        const bool ra = (r0 >= 0 && r0 < N), rb = (r0 >= 1 && r0 <= N);
        float ga[5], gb[5];  // rows r0 and r0 - 1 at columns q0 - 1 .. q0 + 3
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const int qq = q0 - 1 + t;
          const bool qin = (qq >= 0 && qq < N);
          if constexpr (CL) {  // (unconditional loads from clamped positions, selected afterwards: see load_column)
            const int qc = min(max(qq, 0), N - 1);
            const float va = xwg_loadf<true>(&GSl[min(max(r0, 0), N - 1) * GST + qc]);
            const float vb = xwg_loadf<true>(&GSl[min(max(r0 - 1, 0), N - 1) * GST + qc]);
            ga[t] = (ra && qin) ? va : 0.f;
            gb[t] = (rb && qin) ? vb : 0.f;
          } else {
            ga[t] = (ra && qin) ? GSl[r0 * GST + qq] : 0.f;
            gb[t] = (rb && qin) ? GSl[(r0 - 1) * GST + qq] : 0.f;
          }
        }
        float o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = fmaf(w00, ga[t + 1], fmaf(w01, ga[t], fmaf(w10, gb[t + 1], w11 * gb[t])));
        float *dst = HGe + ky * N + kx0;
        if (kx0 > 0 && kx0 + 4 < N) {
          *(float4 *)dst = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (kx0 + t > 0 && kx0 + t < N - 1) dst[t] = o[t];
        }
      }
    }
    const int n_rows = 2 * marg * N, n_cols = 2 * marg * NI;
    for (int q = part * C::NTHR + tid; q < n_rows + n_cols; q += nparts * C::NTHR) {
      int ky, kx;
      if (q < n_rows) {
        const int r = q / N;
        kx = q % N;
        ky = (r < marg) ? r : N - 2 * marg + r;
      } else {
        const int q2 = q - n_rows, r = q2 / (2 * marg), cidx = q2 % (2 * marg);
        ky = marg + r;
        kx = (cidx < marg) ? cidx : N - 2 * marg + cidx;
      }
      const float rx = (float)kx - c0, ry = (float)ky - c0;
      const float px = c0 + (ca * rx - sa * ry) + sdx, py = c0 + (sa * rx + ca * ry) + sdy;
      int vlo = (int)floorf(px) - 1, vhi = vlo + 3, ulo = (int)floorf(py) - 1, uhi = ulo + 3;
      if (kx == 0) { ulo -= mext; uhi += mext; }
      if (kx == N - 1) { ulo -= mext; uhi += mext; }
      if (ky == 0) { vlo -= mext; vhi += mext; }
      if (ky == N - 1) { vlo -= mext; vhi += mext; }
      if (kx == 0) vlo = 0;
      if (kx == N - 1) vhi = N - 1;
      if (ky == 0) ulo = 0;
      if (ky == N - 1) uhi = N - 1;
      if (translated) {
        // exact candidate ranges: scene pixel (u, v) touches rows clamp(u + iyc), clamp(u + iyc + 1) and the same in x
        ulo = (ky == 0) ? 0 : ky - iyc - 1;
        uhi = (ky == N - 1) ? N - 1 : ky - iyc;
        vlo = (kx == 0) ? 0 : kx - ixc - 1;
        vhi = (kx == N - 1) ? N - 1 : kx - ixc;
      }
      vlo = max(vlo, 0);
      vhi = min(vhi, N - 1);
      ulo = max(ulo, 0);
      uhi = min(uhi, N - 1);
      float acc = 0.f;
      for (int u = ulo; u <= uhi; ++u) {
        for (int v = vlo; v <= vhi; ++v) {
          float fx = fxc, fy = fyc;
          int x0 = v + ixc, y0 = u + iyc;
          if (!translated) {
            float Xs, Ys;
            sample_coords(u, v, c0, ca, sa, sdx, sdy, Xs, Ys);
            const float x0f = floorf(Xs), y0f = floorf(Ys);
            fx = Xs - x0f;
            fy = Ys - y0f;
            x0 = (int)x0f;
            y0 = (int)y0f;
          }
          const int xa = min(max(x0, 0), N - 1), xb = min(max(x0 + 1, 0), N - 1);
          const int ya = min(max(y0, 0), N - 1), yb = min(max(y0 + 1, 0), N - 1);
          const float wx = ((xa == kx) ? (1.f - fx) : 0.f) + ((xb == kx) ? fx : 0.f);
          const float wy = ((ya == ky) ? (1.f - fy) : 0.f) + ((yb == ky) ? fy : 0.f);
          acc = fmaf(wx * wy, xwg_loadf<CL>(&GSl[u * GST + v]), acc);
        }
      }
      HGe[ky * N + kx] = acc;
    }
  }
  LC_JSTAMP(8);
}

// the totals as a launch of their own (one block of 64 threads per epoch): when phase D, which computes them otherwise, does not run
__global__ void joint_epoch_finish_kernel(JointArgs A, int SS, int nparts) {
  __shared__ float TOT[4 + 3 * kMaxSources];
  joint_epoch_totals(A, blockIdx.x, threadIdx.x, SS, nparts, TOT);
}

// ---- kernel 2: ordered reduction over the epochs of this rank -------------------------------------
// shared = [ dL/dh (N*N) | dL/dc_x (M) | dL/dc_y (M) | sum_e (a - ref) (M) | sum_e (a - ref)^2 (M) | chi2 | n_epochs ]
// The flux moments are centred on a per-source reference flux a_ref (the same on every rank of a sharded fit): the
// flux-uniformity term needs var = <a^2> - <a>^2, which cancels catastrophically in fp32 when the relative scatter of
// a source is below ~1e-3 (all fluxes start equal in the reference's ROI fit); centred, the cancellation is relative
// to the scatter itself.
// Image part: a block owns 16 consecutive pixels (four 16-byte quads); its 64 groups of four lanes each sum the epochs
// e = g, g + 64, g + 128 ... (every load of a thread independent of the others: the 13 MB of slabs of a 200-epoch fit are
// in flight at once instead of trickling through a few waves per CU), then the 64 partials of a pixel are added in a
// fixed order, so the result does not depend on scheduling.  Scalars: last block, in double.
constexpr int kRedPix = 16, kRedParts = 64, kRedThreads = (kRedPix / 4) * kRedParts;
// sum over the epochs of HG[e][px0 .. px0 + 15] -> one value per pixel in threads tid < 16 (others: 0); in two steps, so that
// a block with several tiles can have the loads of all of them in flight before it combines the first
__device__ __forceinline__ float4 reduce_pixels16_partial(int E, int NN, const float *HG, int px0, int tid) {
  const int quad = tid & (kRedPix / 4 - 1), grp = tid / (kRedPix / 4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float *src = HG + px0 + 4 * quad;
  int e = grp;
  for (; e + 3 * kRedParts < E; e += 4 * kRedParts) {
    const float4 v0 = *(const float4 *)(src + (size_t)e * NN), v1 = *(const float4 *)(src + (size_t)(e + kRedParts) * NN);
    const float4 v2 = *(const float4 *)(src + (size_t)(e + 2 * kRedParts) * NN);
    const float4 v3 = *(const float4 *)(src + (size_t)(e + 3 * kRedParts) * NN);
    acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
    acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
    acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
    acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
  }
  for (; e < E; e += kRedParts) {
    const float4 v = *(const float4 *)(src + (size_t)e * NN);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  return acc;
}
// (every thread of the block calls; one barrier inside, and the caller puts one between two uses of `part`)
__device__ __forceinline__ float reduce_pixels16_combine(float4 acc, float4 (*part)[kRedPix / 4], int tid) {
  const int quad = tid & (kRedPix / 4 - 1), grp = tid / (kRedPix / 4);
  part[grp][quad] = acc;
  __syncthreads();
  float t = 0.f;
  if (tid < kRedPix) {
    const float *p = (const float *)part + tid;  // [grp][16 floats]
#pragma unroll 8
    for (int g = 0; g < kRedParts; ++g) t += p[g * kRedPix];
  }
  return t;
}
__device__ __forceinline__ float reduce_pixels16(int E, int NN, const float *HG, int px0, float4 (*part)[kRedPix / 4], int tid) {
  return reduce_pixels16_combine(reduce_pixels16_partial(E, NN, HG, px0, tid), part, tid);
}
// Scalars of the shared block: dc_x, dc_y, sum (a - ref), sum (a - ref)^2 per source, then chi2 and the epoch count.
// Every thread strides over the epochs with all 4 M + 1 running sums in double (the loads of an epoch are independent of
// each other: one latency for the whole block at E <= 256), lanes combine by shuffles, waves through LDS in order.
// lanes: [kRedThreads] doubles of LDS; scl (optional, LDS): receives a copy of the 4 M + 2 values.  256 threads.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
template <int MM>
__device__ __forceinline__ void wave_sums_f64(double (&acc)[4 * kMaxSources + 1], double *out, int lane) {
  constexpr int NQ = 4 * kMaxSources + 1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if ((q % kMaxSources) < MM || q == 4 * kMaxSources) acc[q] += __shfl_down(acc[q], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if ((q % kMaxSources) < MM || q == 4 * kMaxSources) out[q] = acc[q];
  }
}
__device__ __forceinline__ void reduce_scalars(int E, int M, int NN, const float *g_cx_e, const float *g_cy_e,
                                               const float *chi2_e, const float *a, const float *a_ref, float *shared,
                                               double *lanes, int tid, float *scl = nullptr) {
  constexpr int NQ = 4 * kMaxSources + 1;
  static_assert((kRedThreads / 64) * NQ <= kRedThreads, "LDS scratch");
  const int lane = tid & 63, wid = tid >> 6, nw = kRedThreads / 64;
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
  for (int e = tid; e < E; e += kRedThreads) {
#pragma unroll
    for (int i = 0; i < kMaxSources; ++i) {
      if (i < M) {
        const double ai = (double)a[e * M + i] - (double)a_ref[i];
        acc[i] += (double)g_cx_e[e * M + i];
        acc[kMaxSources + i] += (double)g_cy_e[e * M + i];
        acc[2 * kMaxSources + i] += ai;
        acc[3 * kMaxSources + i] += ai * ai;
      }
    }
    acc[4 * kMaxSources] += (double)chi2_e[e];
  }
  // The 4 M + 1 shuffle trees side by side (wave_sums_f64: per quantity the same additions in the same order as one tree
  // after the other, which is what the loop over the quantities compiled to with M known at run time only - every tree
  // six dependent LDS-crossbar round trips, 2.9 us of this block's 5.9 at M = 2 and E = 200).
  switch (M) {
    case 1: wave_sums_f64<1>(acc, lanes + wid * NQ, lane); break;
    case 2: wave_sums_f64<2>(acc, lanes + wid * NQ, lane); break;
    case 3: wave_sums_f64<3>(acc, lanes + wid * NQ, lane); break;
    case 4: wave_sums_f64<4>(acc, lanes + wid * NQ, lane); break;
    case 5: wave_sums_f64<5>(acc, lanes + wid * NQ, lane); break;
    case 6: wave_sums_f64<6>(acc, lanes + wid * NQ, lane); break;
    case 7: wave_sums_f64<7>(acc, lanes + wid * NQ, lane); break;
    default: wave_sums_f64<8>(acc, lanes + wid * NQ, lane); break;
  }
  __syncthreads();
  if (tid < NQ && ((tid % kMaxSources) < M || tid == 4 * kMaxSources)) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += lanes[w * NQ + tid];
    const int dst = (tid == 4 * kMaxSources) ? 4 * M : (tid / kMaxSources) * M + (tid % kMaxSources);
    shared[NN + dst] = (float)t;
    if (scl) scl[dst] = (float)t;
  }
  if (tid == 0) {
    shared[NN + 4 * M + 1] = (float)E;
    if (scl) scl[4 * M + 1] = (float)E;
  }
}
__global__ __launch_bounds__(kRedThreads) void joint_reduce_kernel(int E, int M, int NN, int need_h, const float *HG,
                                                                    const float *g_cx_e, const float *g_cy_e,
                                                                    const float *chi2_e, const float *a,
                                                                    const float *a_ref, float *shared) {
  __shared__ float4 part[kRedParts][kRedPix / 4];
  const int nimg = NN / kRedPix;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nimg) {
    const int px0 = blockIdx.x * kRedPix;
    const float t = need_h ? reduce_pixels16(E, NN, HG, px0, part, tid) : 0.f;
    if (tid < kRedPix) shared[px0 + tid] = t;
    return;
  }
  __shared__ double lanes[kRedThreads];
  reduce_scalars(E, M, NN, g_cx_e, g_cy_e, chi2_e, a, a_ref, shared, lanes, tid);
}

// ---- kernel 3: regularisers, loss, AdaBelief -------------------------------------------------------
// What the four-launch regulariser chain (joint_reg_fused.h) leaves for its consumer to add up: the sub-gradient as planes
// S_0 + sum_{s = 1 .. J} Z_s, the values of the terms and the inner products of the point-source term per 64 x 64 tile.
constexpr int kPtsBlocks = 64;                       // blocks of the separable point-source term (N^2 / 64 pixels each)
constexpr int kPtsStride = 3 * kMaxSources + 1;      // per block: 3 M inner products, [3 kMaxSources] the value of the term
struct RegPlanes {
  int on, J, ntile, npts;  // npts: kPtsBlocks when pts_part holds this iteration's point-source term, else 0
  const float *S0;        // [NN]
  const float *Z;         // [J + 2][NN]: planes 1 .. J in use
  const float *vals;      // [J + 1][ntile]: rows 0 .. J - 1 l1 per scale, row J positivity
  const float *pts_part;  // [kPtsBlocks][kPtsStride]
};
struct JointUpdArgs {
  int E, M, mode, t, hist_stride, ss;  // mode 1 = update, 0 = gradients only
  int reg_mode;                        // 0 = background regulariser inline, 1 = compute it only (-> greg, regs), 2 = use greg / regs
  float *greg, *regs;                  // [N*N] sub-gradient of the h regularisers, [2] = l1, positivity values
  int free_mask[LC_P_COUNT];
  const float *shared;             // reduced (and, multi-GPU, all-reduced) block
  float *h, *mh, *sh;              // [N*N]
  const float *W, *norms;          // [J][N*N] or null; [J]
  float *qscr;                     // [J][N*N]
  float *par[LC_P_COUNT], *pm[LC_P_COUNT], *ps[LC_P_COUNT];  // device parameter blocks and moments (P_H unused here)
  const float *g_a, *g_dx, *g_dy, *g_mean;                    // per-epoch gradients of kernel 1
  // fused scalar reduction (multi-block update kernel, background fixed): block 0 first sums the per-epoch scalars
  // into shared_w, sparing the separate reduction launch
  int fuse_scalar_reduce;
  const float *a_ref;              // [M] reference fluxes the moments of the shared block are centred on
  const float *g_cx_e, *g_cy_e, *chi2_e;
  float *shared_w;
  float *gout[LC_P_COUNT];         // gradient outputs (mode 0), nullable
  float *hist, *out_loss;
  float lam_sc, lam_hf, lam_pos, lam_pos_ps, lam_pts, lam_fu;
  int n_prior;
  const float *prior_cx_mean, *prior_cx_sigma, *prior_cy_mean, *prior_cy_sigma;
  lc_adabelief_cfg ab;
  float lr, bc1, bc2;  // learning rate and bias corrections of iteration t, evaluated on the host (adabelief_schedule)
  // point-source starlet term evaluated ahead of the update, next to the background regulariser (it depends on the
  // current a, c_x, c_y only): pts_early = 1 in the reg_mode 1 launch computes it (mean fluxes from par[A] itself) and
  // leaves regs[2] = value, regs[4 + 3 i ..] = d/d abar_i, d/dX_i / abar_i, d/dY_i / abar_i;  pts_early = 2 in the
  // update launch consumes them instead of evaluating the term
  int pts_early;
  // fused update of the device loop: the regulariser chain on the second stream publishes wait_seq in *wait_flag when greg /
  // regs are complete; the kernel checks the flag itself (the chain was enqueued first and is normally long done), which
  // spares the cross-stream event wait in front of it.  A wait that runs out sets *wait_err.
  const unsigned int *wait_flag;
  unsigned int wait_seq;
  unsigned int *wait_err;
  RegPlanes planes;   // planes.on: greg / regs are not materialised, the kernel adds the chain's planes and per-tile values itself
  // return_param_history (reference call sites star_photometry.py:115-122, roi_modelling.py:326-334): the row of the
  // device-resident history [iterations][P] this update fills - every free block at its offset poff[which] (-1: fixed),
  // written where the parameter itself is stored, so the loop stays on the device
  float *phist;
  int poff[LC_P_COUNT];
};
// (one extra store next to the store of an updated parameter)
__device__ __forceinline__ void phist_put(const JointUpdArgs &A, int which, int idx, float v) {
  if (A.phist && A.poff[which] >= 0) A.phist[A.poff[which] + idx] = v;
}

// All threads of a block: block until *flag has reached seq (thread 0 polls, bounded: ~0.2 s).  No acquire fence follows: an
// agent-scope acquire invalidates the L2 of the XCD, and a thousand blocks doing that cost 33 us; the few values the
// chain produced are read with ld_coherent (L2-bypassing loads) instead, and the chain wrote its L2 back before the flag.
// (seen: what flag_seen() returned to thread 0 earlier in the kernel - a look at the flag requested together with the kernel's
//  first loads, so that in the usual case, the chain long finished, the wait costs no memory round trip of its own)
__device__ __forceinline__ bool flag_seen(const unsigned int *flag, unsigned int seq) {
  return flag && threadIdx.x == 0 &&
         (int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) >= 0;
}
__device__ __forceinline__ void wait_for_flag(const unsigned int *flag, unsigned int seq, unsigned int *err, bool seen = false) {
  if (flag) {
    if (threadIdx.x == 0 && !seen) {
      int spins = 0;
      while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1 << 20)) {
          if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
  }
}
__device__ __forceinline__ float ld_coherent(const float *p, bool coherent) {
  return coherent ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
// greg of pixel k from the planes (mreg_finish2_kernel's order: S_0, then Z_1 .. Z_J); all loads requested before the first sum
constexpr int kPlanesMaxJ = 8;
__device__ __forceinline__ float planes_greg(const RegPlanes &P, int k, int NN, bool coherent) {
  float zv[kPlanesMaxJ];
  float g = ld_coherent(P.S0 + k, coherent);
#pragma unroll
  for (int s = 1; s <= kPlanesMaxJ; ++s) zv[s - 1] = ld_coherent(P.Z + (size_t)min(s, P.J) * NN + k, coherent);
#pragma unroll
  for (int s = 1; s <= kPlanesMaxJ; ++s) g += (s <= P.J) ? zv[s - 1] : 0.f;
  return g;
}
// (no contraction into fused multiply-adds here: several kernels apply this step - fused and split update, single-workgroup
//  and multi-block form - and their results are required to agree bit for bit, whatever the surrounding code looks like)
__device__ __forceinline__ float adabelief_step(float &p, float &m, float &s, float g, float lr, float bc1, float bc2,
                                                const lc_adabelief_cfg &ab) {
#pragma clang fp contract(off)
  const float mn = ab.b1 * m + (1.f - ab.b1) * g;
  const float dg = g - mn;
  const float sn = ab.b2 * s + (1.f - ab.b2) * dg * dg + ab.eps_root;
  m = mn;
  s = sn;
  p -= lr * (mn * bc1) / (sqrtf(sn * bc2) + ab.eps);
  return p;
}

template <int N, int PX>
__global__ __launch_bounds__(N *N / PX) void joint_update_kernel(JointUpdArgs A) {
  constexpr int NTHR = N * N / PX, NWV = (NTHR + 63) / 64, J = ilog2(N);
  extern __shared__ __align__(16) float lds[];
  __shared__ float red[NWV * 2 + 16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int pu = tid / (N / PX), pv = (tid % (N / PX)) * PX;
  const size_t pix = (size_t)pu * N + pv;
  const int E = A.E, M = A.M, NN = N * N;
  const float Etot = A.shared[NN + 4 * M + 1];
  const bool h_free = A.free_mask[LC_P_H] != 0;

  float hp[PX], g[PX];
#pragma unroll
  for (int q = 0; q < PX / 4; ++q) {
    const float4 v = *(const float4 *)(A.h + pix + 4 * q);
    const float4 gs = *(const float4 *)(A.shared + pix + 4 * q);
    hp[4 * q] = v.x; hp[4 * q + 1] = v.y; hp[4 * q + 2] = v.z; hp[4 * q + 3] = v.w;
    g[4 * q] = gs.x; g[4 * q + 1] = gs.y; g[4 * q + 2] = gs.z; g[4 * q + 3] = gs.w;
  }
  float l1 = 0.f, pos = 0.f;
  if (A.reg_mode == 2) {
    // the regulariser of h was evaluated by a concurrent launch (it only depends on h)
#pragma unroll
    for (int q = 0; q < PX / 4; ++q) {
      const float4 gr = *(const float4 *)(A.greg + pix + 4 * q);
      g[4 * q] += gr.x; g[4 * q + 1] += gr.y; g[4 * q + 2] += gr.z; g[4 * q + 3] += gr.w;
    }
    if (tid == 0) {
      l1 = A.regs[0];
      pos = A.regs[1];
    }
  }
  float greg_own[PX];
#pragma unroll
  for (int p = 0; p < PX; ++p) greg_own[p] = 0.f;
  if (A.reg_mode != 2 && (A.lam_sc != 0.f || A.lam_hf != 0.f)) {
    float z[PX];
    starlet_l1_grad<N, PX>(hp, A.W, A.norms, A.qscr, A.lam_sc, A.lam_hf, lds, tid, l1, z);
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      g[p] += z[p];
      greg_own[p] += z[p];
    }
  }
  if (A.reg_mode != 2 && A.lam_pos != 0.f) {
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      pos += (hp[p] < 0.f) ? -A.lam_pos * hp[p] : 0.f;
      g[p] += (hp[p] < 0.f) ? -A.lam_pos : 0.f;
      greg_own[p] += (hp[p] < 0.f) ? -A.lam_pos : 0.f;
    }
  }
  // ---- regularization_strength_pts_source: lam * sum W_0 |starlet_0(Pbar)|, Pbar = sum_i mean_e(a_i) G(c_i)
  //      (the point-source channel at the target resolution in the reference frame, DESIGN.md SPEC) ----
  __shared__ float pts_red[NWV * 3 * kMaxSources + 3 * kMaxSources];
  __shared__ float pts_abar[kMaxSources];
  __shared__ float pts_red3[NWV];
  float pts_l1_own = 0.f;
  const bool pts_here = A.lam_pts != 0.f && A.pts_early != 2 && (A.reg_mode != 1 || A.pts_early == 1);
  if (pts_here && A.reg_mode == 1) {  // mean fluxes straight from the parameters: one wave per source
    for (int i = wid; i < M; i += NWV) {
      float acc = 0.f;
      for (int e2 = lane; e2 < E; e2 += 64) acc += A.par[LC_P_A][e2 * M + i];
      acc = wave_sum(acc);
      if (lane == 0) pts_abar[i] = acc / (float)E;
    }
    __syncthreads();
  }
  if (pts_here) {
    const float c0 = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
    float pb[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) pb[p] = 0.f;
    for (int i = 0; i < M; ++i) {
      const float abar = (A.reg_mode == 1) ? pts_abar[i] : A.a_ref[i] + A.shared[NN + 2 * M + i] / Etot;
      const float X = c0 + A.ss * A.par[LC_P_CX][i], Y = c0 + A.ss * A.par[LC_P_CY][i];
      const float ty = (float)pu - Y;
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        const float tx = (float)(pv + p) - X;
        pb[p] = fmaf(abar * nrm2, expf(-0.5f * (tx * tx + ty * ty) * inv_s2), pb[p]);
      }
    }
    float zp[PX], l1p = 0.f;
    __syncthreads();  // the background starlet above is done with the LDS buffers
    starlet_l1_grad<N, PX, 1>(pb, A.W, A.norms, A.qscr, 0.f, A.lam_pts, lds, tid, l1p, zp);
    if (A.reg_mode != 1) pos += l1p;  // joins the loss through the same reduction
    pts_l1_own = l1p;
    for (int i = 0; i < M; ++i) {
      const float X = c0 + A.ss * A.par[LC_P_CX][i], Y = c0 + A.ss * A.par[LC_P_CY][i];
      const float ty = (float)pu - Y;
      float sa = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        const float tx = (float)(pv + p) - X;
        const float gq = zp[p] * nrm2 * expf(-0.5f * (tx * tx + ty * ty) * inv_s2);
        sa += gq;
        sx = fmaf(gq, tx * inv_s2, sx);
        sy = fmaf(gq, ty * inv_s2, sy);
      }
      sa = wave_sum(sa);
      sx = wave_sum(sx);
      sy = wave_sum(sy);
      if (lane == 0) {
        pts_red[(wid * kMaxSources + i) * 3 + 0] = sa;
        pts_red[(wid * kMaxSources + i) * 3 + 1] = sx;
        pts_red[(wid * kMaxSources + i) * 3 + 2] = sy;
      }
    }
    __syncthreads();
    if (tid < 3 * M) {
      const int i = tid / 3, q = tid % 3;
      float acc = 0.f;
      for (int w = 0; w < NWV; ++w) acc += pts_red[(w * kMaxSources + i) * 3 + q];
      pts_red[NWV * 3 * kMaxSources + i * 3 + q] = acc;  // d/d abar_i, d/dX_i / abar_i, d/dY_i / abar_i
    }
  }
  if (A.reg_mode == 1) {  // regulariser-only launch: publish and leave
#pragma unroll
    for (int q = 0; q < PX / 4; ++q)
      *(float4 *)(A.greg + pix + 4 * q) = make_float4(greg_own[4 * q], greg_own[4 * q + 1], greg_own[4 * q + 2], greg_own[4 * q + 3]);
    const float s1 = wave_sum(l1), s2 = wave_sum(pos), s3 = wave_sum(pts_l1_own);
    if (lane == 0) {
      red[wid * 2] = s1;
      red[wid * 2 + 1] = s2;
      pts_red3[wid] = s3;
    }
    __syncthreads();
    float pts_l1_total = 0.f;
    if (tid == 0) {
      float t1 = 0.f, t2 = 0.f;
      for (int w = 0; w < NWV; ++w) {
        t1 += red[w * 2];
        t2 += red[w * 2 + 1];
        pts_l1_total += pts_red3[w];
      }
      A.regs[0] = t1;
      A.regs[1] = t2;
      A.regs[2] = pts_l1_total;
    }
    if (pts_here && tid < 3 * M) A.regs[4 + tid] = pts_red[NWV * 3 * kMaxSources + tid];
    return;
  }
  const float *ptsg = (A.pts_early == 2) ? (A.regs + 4) : (pts_red + NWV * 3 * kMaxSources);
  if (A.pts_early == 2 && tid == 0) pos += A.regs[2];
  {
    const float s1 = wave_sum(l1), s2 = wave_sum(pos);
    if (lane == 0) {
      red[wid * 2] = s1;
      red[wid * 2 + 1] = s2;
    }
  }
  __syncthreads();
  // scalars: learning rate, bias corrections, flux statistics, loss
  float *sc = red + NWV * 2;  // [0] lr [1] bc1 [2] bc2
  if (tid == 0) {
    sc[0] = A.lr;
    sc[1] = A.bc1;
    sc[2] = A.bc2;
    float tl1 = 0.f, tpos = 0.f;
    for (int w = 0; w < NWV; ++w) {
      tl1 += red[w * 2];
      tpos += red[w * 2 + 1];
    }
    double loss = 0.5 * (double)A.shared[NN + 4 * M] + tl1 + tpos;
    // flux uniformity: lam * sum_i std_e(a_{e,i}) over ALL epochs of the fit
    if (A.lam_fu != 0.f && Etot > 1.f)
      for (int i = 0; i < M; ++i) {
        const double meanc = A.shared[NN + 2 * M + i] / Etot;  // centred on a_ref
        const double var = fmax((double)A.shared[NN + 3 * M + i] / Etot - meanc * meanc, 0.0);
        loss += A.lam_fu * sqrt(var);
      }
    if (A.n_prior > 0)
      for (int i = 0; i < M; ++i) {
        const double zx = (A.par[LC_P_CX][i] - A.prior_cx_mean[i]) / A.prior_cx_sigma[i];
        const double zy = (A.par[LC_P_CY][i] - A.prior_cy_mean[i]) / A.prior_cy_sigma[i];
        loss += 0.5 * (zx * zx + zy * zy);
      }
    sc[3] = (float)loss;  // positivity of the fluxes is added by the a-loop below through sc[4..]
  }
  __syncthreads();
  const float lr = sc[0], bc1 = sc[1], bc2 = sc[2];
  // ---- h ----
  if (A.mode == 0 && A.gout[LC_P_H]) {
#pragma unroll
    for (int p = 0; p < PX; ++p) A.gout[LC_P_H][pix + p] = g[p];
  }
  if (A.mode == 1 && h_free) {
#pragma unroll
    for (int q = 0; q < PX / 4; ++q) {
      float4 m = *(float4 *)(A.mh + pix + 4 * q), s = *(float4 *)(A.sh + pix + 4 * q);
      float *mm = &m.x, *ss_ = &s.x;
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) adabelief_step(hp[4 * q + e4], mm[e4], ss_[e4], g[4 * q + e4], lr, bc1, bc2, A.ab);
      *(float4 *)(A.h + pix + 4 * q) = make_float4(hp[4 * q], hp[4 * q + 1], hp[4 * q + 2], hp[4 * q + 3]);
      *(float4 *)(A.mh + pix + 4 * q) = m;
      *(float4 *)(A.sh + pix + 4 * q) = s;
      if (A.phist && A.poff[LC_P_H] >= 0)
        *(float4 *)(A.phist + A.poff[LC_P_H] + pix + 4 * q) = make_float4(hp[4 * q], hp[4 * q + 1], hp[4 * q + 2], hp[4 * q + 3]);
    }
  }
  // ---- per-epoch parameters: a (E*M), dx, dy, mean (E) ----
  float pos_ps = 0.f;
  for (int idx = tid; idx < E * M; idx += NTHR) {
    const int i = idx % M;
    float av = A.par[LC_P_A][idx];
    float ga = A.g_a[idx];
    if (A.lam_pos_ps != 0.f && av < 0.f) {
      pos_ps += -A.lam_pos_ps * av;
      ga -= A.lam_pos_ps;
    }
    if (A.lam_fu != 0.f && Etot > 1.f) {
      const float meanc = A.shared[NN + 2 * M + i] / Etot;
      const float var = fmaxf(A.shared[NN + 3 * M + i] / Etot - meanc * meanc, 0.f);
      const float sd = sqrtf(var);
      if (sd > 0.f) ga += A.lam_fu * ((av - A.a_ref[i]) - meanc) / (Etot * sd);
    }
    if (A.lam_pts != 0.f) ga += ptsg[i * 3] / Etot;
    if (A.mode == 0) {
      if (A.gout[LC_P_A]) A.gout[LC_P_A][idx] = ga;
    } else if (A.free_mask[LC_P_A]) {
      adabelief_step(av, A.pm[LC_P_A][idx], A.ps[LC_P_A][idx], ga, lr, bc1, bc2, A.ab);
      A.par[LC_P_A][idx] = av;
      phist_put(A, LC_P_A, idx, av);
    }
  }
  for (int idx = tid; idx < 3 * E; idx += NTHR) {
    const int which = (idx / E == 0) ? LC_P_DX : (idx / E == 1) ? LC_P_DY : LC_P_MEAN;
    const int e = idx % E;
    const float gv = (which == LC_P_DX) ? A.g_dx[e] : (which == LC_P_DY) ? A.g_dy[e] : A.g_mean[e];
    if (A.mode == 0) {
      if (A.gout[which]) A.gout[which][e] = gv;
    } else if (A.free_mask[which]) {
      float pv_ = A.par[which][e];
      adabelief_step(pv_, A.pm[which][e], A.ps[which][e], gv, lr, bc1, bc2, A.ab);
      A.par[which][e] = pv_;
      phist_put(A, which, e, pv_);
    }
  }
  // ---- shared point-source positions ----
  if (tid < 2 * M) {
    const int which = (tid < M) ? LC_P_CX : LC_P_CY, i = tid % M;
    float gv = A.shared[NN + (which == LC_P_CX ? 0 : M) + i];
    float cv = A.par[which][i];
    if (A.lam_pts != 0.f) gv += (A.a_ref[i] + A.shared[NN + 2 * M + i] / Etot) * A.ss * ptsg[i * 3 + (which == LC_P_CX ? 1 : 2)];
    if (A.n_prior > 0) {
      const float mu = (which == LC_P_CX) ? A.prior_cx_mean[i] : A.prior_cy_mean[i];
      const float sg = (which == LC_P_CX) ? A.prior_cx_sigma[i] : A.prior_cy_sigma[i];
      gv += (cv - mu) / (sg * sg);
    }
    if (A.mode == 0) {
      if (A.gout[which]) A.gout[which][i] = gv;
    } else if (A.free_mask[which]) {
      adabelief_step(cv, A.pm[which][i], A.ps[which][i], gv, lr, bc1, bc2, A.ab);
      A.par[which][i] = cv;
      phist_put(A, which, i, cv);
    }
  }
  // ---- loss (positivity of fluxes reduced here) ----
  {
    const float s = wave_sum(pos_ps);
    __syncthreads();
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (tid == 0) {
      float tp = 0.f;
      for (int w = 0; w < NWV; ++w) tp += red[w];
      const float loss = sc[3] + tp;
      if (A.hist) A.hist[A.t] = loss;
      if (A.out_loss) *A.out_loss = loss;
    }
  }
}

}  // namespace lc
