// PSF-fit device code: one persistent workgroup per frame (gfx950, wave64).
//
// Replaces the arithmetic of starred.procedures.psf_routines.build_psf as called at
// lightcurver/processes/psf_modelling.py:164-171 (reference).  Model of star i of a frame:
//     f_i = a_i * D_ss[ G(x0_i, y0_i) (*) (Moffat + B) ] + sky_i
// G is the separable FWHM-2 Gaussian, so the forward is a 1-D row pass fused with the column
// down-sampling followed by a 1-D column pass fused with the row down-sampling; the backward is
// the two transposed passes.  T = Moffat + B, the per-star intermediates and the weighted
// residuals live in LDS; B's chi2 gradient stays in registers through the starlet phase.
#pragma once
#include <utility>

#include "lc_common.h"
#include "starlet_device.h"

#ifndef LC_XPOLL_SLEEP
#define LC_XPOLL_SLEEP 2
#endif
#ifndef LC_PSF_WAVEFLAGS
#define LC_PSF_WAVEFLAGS 0
#endif
namespace lc {
constexpr int kXFlagStride = 16;   // flag words per (frame, role): one per wave in the per-wave form, word 0 in the workgroup form

#ifdef LC_STAMPS
__device__ long long g_stamps[128];  // [0, 64): block 0, [64, 128): block 8 (role 1 of frame 0 in the two-workgroup form)
#define LC_STAMP(k)                                                      \
  do {                                                                   \
    if ((blockIdx.x == 0 || blockIdx.x == 8) && threadIdx.x == 0 && it == A.n_iter - 1)  \
      g_stamps[(k) + (blockIdx.x == 8 ? 64 : 0)] = clock64();            \
  } while (0)
#else
#define LC_STAMP(k) do {} while (0)
#endif

// LC_LAUNDER (starlet_device.h) stops loop-invariant code motion from pre-computing (and spilling)
// per-thread LDS addresses.

struct PsfArgs {
  int F, S, n_iter, t0, hist_stride, mode;  // mode 0 = evaluate, 1 = AdaBelief loop
  const float *data, *wgt;                  // [F][S][n][n]
  const float *W;                           // [F][J][N*N] or null
  const float *norms;                       // [J] starlet scale norms (used when W == null)
  const float *Tm;                          // [F][N*N] unit-sum Moffat
  float *B, *mB, *sB;                       // [F][N*N] pixel grid and AdaBelief moments
  float *stars, *stars_m, *stars_s;         // [F][S][4] a, x0, y0, sky (+ moments)
  float *hist;                              // [F][hist_stride]
  float *qscratch;                          // [F][J][N*N] thread-private l1 sub-gradients
  float *out_loss, *out_chi2, *out_gstars, *out_ggrid, *out_gT, *out_model;  // eval outputs (nullable)
  float lam_sc, lam_hf;
  lc_adabelief_cfg ab;
  // two workgroups per frame (psf_fit_kernel<C, true>): exchange slabs [F][2 buffers][2 roles][N*N + 64],
  // per-(frame, role) iteration flags [F][2], one abort word
  float *xch;
  int *xflags, *xabort;
  // Abort word protocol: a split launch that gives up (a partner that does not show up) stores its own launch
  // sequence number there; the one-workgroup launch enqueued right behind it with heal = 1 compares the word with the
  // same number: equal -> it restores the pre-launch state of its frame from the copies below and redoes the whole
  // launch, different -> it returns at once.  The word is never cleared, so nothing can erase an abort.
  int launch_seq, heal, force_abort_it;  // force_abort_it >= 0: test hook, role 1 of frame 0 gives up at that iteration
  const float *bkB, *bkmB, *bksB, *bkstars, *bkstars_m, *bkstars_s;
  int *heal_count;
  const float *ext_grad;  // [F][N*N] or null: an additional d loss / d B, added to the chi2 part (distortion fit: sum over the
                          // stars of the adjoint resampling of their gradients, csrc/psf_distort.h)
  const float *sched;     // [>= t0 + n_iter][3]: learning rate and bias corrections by absolute iteration (host-made)
  float *B1, *mB1, *sB1;  // [F][N*N]: role 1's own copy of the pixel state when it does not fit in registers
  int xcd_fast;           // two-workgroup form: partners that find themselves on one XCD hand over through its L2 (below)
};

// Write-through (sc1) stores and L1-bypassing (sc1) loads for data handed from one workgroup to another
// inside a launch (MI355X_MICROARCH.md, inter-workgroup visibility: all stores and all loads of the handed-off
// bytes sc1, every storing wave drains vmcnt, workgroup barrier, one lane raises an sc1 flag).
typedef float lc_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1_x4(float *p, lc_v4f v) {
  // s_nop: a VMEM store of more than 64 bits followed by a VALU write of its data registers needs wait states;
  // the compiler's hazard recogniser cannot see inside the asm, so they are spelled out here
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_plain_x4(float *p, lc_v4f v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_sc1_f(float *p, float v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void load_sc1_2x4(const float *p, lc_v4f &a, lc_v4f &b) {
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b)
               : "v"(p)
               : "memory");
}
// the same with one more dword (the partner's scalar) in flight beside the two quads: one round trip instead of two
__device__ __forceinline__ void load_sc1_2x4_1(const float *p, lc_v4f &a, lc_v4f &b, const float *ps, float &s) {
  asm volatile("global_load_dword %2, %4, off sc1\n\tglobal_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b), "=&v"(s)
               : "v"(p), "v"(ps)
               : "memory");
}
__device__ __forceinline__ void load_sc1_x4(const float *p, lc_v4f &a) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory");
}
__device__ __forceinline__ float load_sc1_f(const float *p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

// (packed fp32 helpers lc_v2f / pk_fma / pk_bcast: starlet_device.h)

template <int N_, int SS_, int PX_, int SG_, bool WC_ = false, int JB_ = 8>
struct PsfCfg {
  static constexpr int N = N_, SS = SS_, PX = PX_, SG = SG_;
  // WC: the row pass, column pass and transposed column pass of one (star, block of JB down-sampled columns)
  // run inside ONE wave with wave-level synchronisation only (per-wave LDS scratch)
  static constexpr bool WC = WC_;
  static constexpr int n = N / SS;
  static constexpr int NTHR = N * N / PX;
  static constexpr int NW = (NTHR + kWave - 1) / kWave;
  static constexpr int NT = ntaps(SS);
  static constexpr int J = ilog2(N);
  static constexpr int LR = 8;  // row-pass strip (down-sampled columns per work item)
  // column-pass strip (down-sampled rows per work item) and transposed column pass strip (LA data rows = SS*LA
  // high-res rows): sized so that a wave task of JB columns keeps all 64 lanes busy
  static constexpr int LC = (JB_ * (N_ / SS_) / 4 >= 64) ? 4 : 2;
  static constexpr int LA = LC;
  static constexpr int TS = N + 1;  // padded LDS row stride of N-long rows
  static constexpr int RS = n + 1;  // padded LDS row stride of n-long rows
  // LDS carve-up (floats)
  static constexpr int OFF_T = 0;
  static constexpr int SZ_T = WC_ ? N * (N + 2 * (N / 4 + 10) + 1) : N * TS;
  static constexpr int OFF_R = OFF_T + SZ_T;
  static constexpr int SZ_R2 = SG * n * TS;  // R2t / R2xt: [SG][n][TS]
  // V: [SG][N][VS]; the WC layout keeps one zero column on either side of the n data columns so that the
  // transposed row pass reads out-of-stamp samples as zeros through a clamped index, without selects
  static constexpr int VS = WC_ ? n + 3 : RS, VO = WC_ ? 1 : 0;
  static constexpr int SZ_V = SG * N * VS;
  static constexpr int JB = JB_;                      // down-sampled columns per wave task (WC)
  // WC tiles carry zero aprons so that the filter windows are read without clamping or selects.  The star
  // offsets are limited to +-N/4 high-res pixels (tap_entry), which bounds every window by AP / APR.
  static constexpr int AP = N / 4 + 10;               // apron of N-long rows (high-res pixels)
  static constexpr int TSA = N + 2 * AP + 1;          // row stride of T, R2, R2x in the WC layout
  static constexpr int APR = AP / SS + 2;             // apron of the residual rows (data pixels)
  // padded row stride of the per-wave residual tile (at JB + 1 = 9 the reads of the transposed column pass are 2-way bank
  // conflicts, at 10 none are: measured no faster - 14.9 against 14.8 us per iteration - and 2 KB of LDS dearer: not kept)
  static constexpr int JBP = JB + 1;
  static constexpr int WSZ = JB * TSA + (n + 2 * APR) * JBP;  // per-wave scratch: R2 [JB][TSA], residuals [n+2APR][JBP]
  static constexpr int SZ_VR = (SZ_V > StarletLds<N>::FLOATS) ? SZ_V : StarletLds<N>::FLOATS;  // V, reused by the starlet
  static constexpr int OFF_WSC = OFF_R + SZ_VR;
  static constexpr int SZ_R = WC ? (SZ_VR + NW * WSZ) : ((2 * SZ_R2 > SZ_V) ? 2 * SZ_R2 : SZ_V);
  static constexpr int OFF_RES = OFF_R + SZ_R;
  static constexpr int SZ_RES = WC ? 0 : SG * n * n;
  static constexpr int OFF_TAPS = OFF_RES + SZ_RES;
  // tap rows are padded: [0] = 0, [1 + k] = tap k, [NT + 1] = [NT + 2] = 0, so that the paired-tap forms of the
  // passes read their out-of-range partner as a zero
  static constexpr int NTP = NT + 3;
  // plus the same taps interleaved as (value, derivative) pairs per axis, 8-byte aligned: [star][axis][NT] float2
  static constexpr int SZ_TAPS = (16 * 4 * NTP + 1) / 2 * 2 + 16 * 2 * NT * 2;  // every star of the frame
  static constexpr int OFF_TAPP = OFF_TAPS + (16 * 4 * NTP + 1) / 2 * 2;
  static constexpr int OFF_RED = OFF_TAPS + SZ_TAPS;
  static constexpr int IPS = n * n / LC;  // column-pass items per star
  static constexpr int IPS_PAD = (IPS + kWave - 1) / kWave * kWave;
  static constexpr int SLOTS = IPS_PAD / kWave;
  static constexpr int SZ_REDX = 16 * NW;  // per-(star, wave) partial sums of the x0 gradient (transposed row pass)
  static constexpr int SZ_RED = (WC ? 16 * (n / JB) * 5 : SG * SLOTS * 5) + SZ_REDX + NW + 8;
  static constexpr int OFF_REDX = OFF_RED + (WC ? 16 * (n / JB) * 5 : SG * SLOTS * 5);
  static constexpr int OFF_REDW = OFF_REDX + SZ_REDX;
  static constexpr int OFF_STAR = OFF_RED + SZ_RED;  // star params, grads, moments, ints
  static constexpr int MAXS = 16;
  static constexpr int SZ_STAR = MAXS * 20 + 16;
  static constexpr int LDS_FLOATS = OFF_STAR + SZ_STAR;
  static_assert(WC || StarletLds<N>::FLOATS <= SZ_T + SZ_R + SZ_RES, "starlet ping-pong buffers must fit over T + R + RES");
  static_assert(PX % SS == 0 && N % PX == 0 && n % LR == 0 && n % LC == 0 && n % LA == 0 && n % JB == 0, "tiling");
  static_assert(LDS_FLOATS * 4 <= 163840, "LDS");
  static_assert(OFF_TAPP % 2 == 0, "tap pairs are read as 64-bit words");
  static_assert(NTHR <= 1024 && NTHR % kWave == 0, "threads");
  static_assert(!WC_ || PX_ <= 8, "the WC layout keeps the pixel state in registers");
};


// Build the aligned tap tables of one (star, axis): taps[k] = Phi(base' + k), dtaps[k] = dPhi/ddelta.
// Phi(m) = sum_{dv<SS} phi(m + dv), phi(t) = N(t; delta, sigma) truncated to |t - round(delta)| <= kRg.
template <int SS, int NT>
__device__ inline void tap_entry(float delta, int k, float &tap, float &dtap, int &bq) {
  const int o = (int)nearbyintf(delta);  // |delta| is limited by the caller
  const int base = o - kRg - (SS - 1);
  int q = base / SS;
  if (q * SS > base) --q;  // floor division
  bq = q;
  const int m = q * SS + k;
  const float inv_s2 = 1.0f / (kSigmaG * kSigmaG);
  const float nrm = 0.3989422804014327f / kSigmaG;
  float a = 0.f, d = 0.f;
#pragma unroll
  for (int dv = 0; dv < SS; ++dv) {
    const int t = m + dv;
    if (t >= o - kRg && t <= o + kRg) {
      const float x = (float)t - delta;
      const float p = nrm * expf(-0.5f * x * x * inv_s2);
      a += p;
      d += p * x * inv_s2;
    }
  }
  tap = a;
  dtap = d;
}

// SPLIT: two workgroups per frame.  Role 0 evaluates the forward model and the chi2 gradient, role 1 the starlet
// l1 term of the same B; they swap their halves of dL/dB through L2 once per iteration and both apply the
// identical AdaBelief update to their own register copy of B (same operands, same order => same bits).
// Block b serves frame (b / 16) * 8 + b % 8 in role (b / 8) % 2: partners are 8 blocks apart, which the
// dispatcher is observed to place on one XCD (speed only).  The host launches this form only when the whole grid
// is resident at once (one workgroup per CU), so the partner a workgroup waits for is always running.
template <class C, bool SPLIT = false>
__global__ __launch_bounds__(C::NTHR) void psf_fit_kernel(PsfArgs A) {
  constexpr int N = C::N, SS = C::SS, PX = C::PX, SG = C::SG, n = C::n, NT = C::NT, J = C::J;
  constexpr int NTHR = C::NTHR, TS = C::TS, RS = C::RS, LR = C::LR, LC = C::LC, LA = C::LA;
  extern __shared__ __align__(16) float lds[];
  float *T = lds + C::OFF_T;
  float *R2t = lds + C::OFF_R;
  float *R2xt = R2t + C::SZ_R2;
  float *V = lds + C::OFF_R;
  float *RES = lds + C::OFF_RES;
  float *TAPS = lds + C::OFF_TAPS + 1;  // [S][4][NTP], entry k of a row at [k] (one zero in front, two behind): tx, dtx, ty, dty
  constexpr int NTP = C::NTP;
  lc_v2f *TAPP = (lc_v2f *)(lds + C::OFF_TAPP);  // [S][2 axes][NT] (tap, dtap)
  float *REDX = lds + C::OFF_REDX;
  float *RED = lds + C::OFF_RED;    // [SG][SLOTS][5] + [NW] + scalars
  float *REDW = lds + C::OFF_REDW;
  float *SCAL = REDW + C::NW;  // lr, bc1, bc2, l1, loss
  float *SP = lds + C::OFF_STAR;    // star params [MAXS][4]
  float *SGR = SP + C::MAXS * 4;    // star grads [MAXS][5]: chi2, ga, gx, gy, gsky
  float *SM = SGR + C::MAXS * 5;    // moments m [MAXS][4]
  float *SV = SM + C::MAXS * 4;     // moments s [MAXS][4]
  int *BQ = (int *)(SV + C::MAXS * 4);  // [S][2]

  int f = blockIdx.x, role = 0;
  if constexpr (SPLIT) {
    f = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7);
    role = ((int)blockIdx.x >> 3) & 1;
    if (f >= A.F) return;
  }
  const bool conv_role = !SPLIT || role == 0, starlet_role = !SPLIT || role == 1;
  // HW_REG_XCC_ID (id 20), bits [3:0]: the XCD this workgroup runs on (s_getreg immediate: id | offset << 6 | (width - 1) << 11)
  const int my_xcc = SPLIT ? (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u) : 0;
  int same_xcd = 0;
  const int tid0 = threadIdx.x;
  const int S = A.S;

  const float *dataf = A.data + (size_t)f * S * n * n;
  const float *wgtf = A.wgt + (size_t)f * S * n * n;

  if constexpr (!SPLIT) {
    if (A.heal) {  // fall-back launch behind a two-workgroup launch: does anything only if that launch gave up
      if (__hip_atomic_load(A.xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != A.launch_seq) return;
      const size_t o = (size_t)f * N * N;
      for (int i = tid0; i < N * N; i += NTHR) {
        A.B[o + i] = A.bkB[o + i];
        A.mB[o + i] = A.bkmB[o + i];
        A.sB[o + i] = A.bksB[o + i];
      }
      if (tid0 < S * 4) {
        A.stars[(size_t)f * S * 4 + tid0] = A.bkstars[(size_t)f * S * 4 + tid0];
        A.stars_m[(size_t)f * S * 4 + tid0] = A.bkstars_m[(size_t)f * S * 4 + tid0];
        A.stars_s[(size_t)f * S * 4 + tid0] = A.bkstars_s[(size_t)f * S * 4 + tid0];
      }
      if (f == 0 && tid0 == 0) atomicAdd(A.heal_count, 1);
      __threadfence_block();
      __syncthreads();
    }
  }

  if constexpr (C::WC) {  // aprons (and everything else) of T and of the per-wave tiles start at zero and stay zero
    for (int i = tid0; i < C::SZ_T; i += NTHR) lds[C::OFF_T + i] = 0.f;
    for (int i = tid0; i < C::NW * C::WSZ; i += NTHR) lds[C::OFF_WSC + i] = 0.f;
  }
  if (tid0 < S * 4) {
    SP[tid0] = A.stars[(size_t)f * S * 4 + tid0];
    SM[tid0] = A.stars_m[(size_t)f * S * 4 + tid0];
    SV[tid0] = A.stars_s[(size_t)f * S * 4 + tid0];
  }
  __syncthreads();

  // Pixel state of this thread.  With <= 8 pixels per thread it stays in registers for the whole launch
  // (B, both AdaBelief moments and the Moffat): HBM then sees it once per launch instead of per iteration.
  constexpr bool STATE_REGS = (PX <= 8);
  // pixel state in global memory (PX > 8): in the two-workgroup form role 1 works on its own copy, taken here
  // (every thread only ever touches its own pixels of it, so no barrier is needed)
  float *Bg = A.B, *mBg = A.mB, *sBg = A.sB;
  if constexpr (SPLIT && !STATE_REGS) {
    if (role == 1) {
      const size_t g0pix = (size_t)f * N * N + (size_t)(tid0 / (N / PX)) * N + (tid0 % (N / PX)) * PX;
#pragma unroll
      for (int q = 0; q < PX / 4; ++q) {
        ((float4 *)(A.B1 + g0pix))[q] = ((const float4 *)(A.B + g0pix))[q];
        ((float4 *)(A.mB1 + g0pix))[q] = ((const float4 *)(A.mB + g0pix))[q];
        ((float4 *)(A.sB1 + g0pix))[q] = ((const float4 *)(A.sB + g0pix))[q];
      }
      Bg = A.B1;
      mBg = A.mB1;
      sBg = A.sB1;
    }
  }
  float Bp[PX], Mp[PX], Sp_[PX], Tp[PX];
  {
    const size_t g0pix = (size_t)f * N * N + (size_t)(tid0 / (N / PX)) * N + (tid0 % (N / PX)) * PX;
#pragma unroll
    for (int q = 0; q < PX / 4; ++q) {
      if (STATE_REGS) {
        const float4 b = ((const float4 *)(A.B + g0pix))[q], t = ((const float4 *)(A.Tm + g0pix))[q];
        const float4 m = ((const float4 *)(A.mB + g0pix))[q], sv = ((const float4 *)(A.sB + g0pix))[q];
        Bp[4 * q] = b.x; Bp[4 * q + 1] = b.y; Bp[4 * q + 2] = b.z; Bp[4 * q + 3] = b.w;
        Tp[4 * q] = t.x; Tp[4 * q + 1] = t.y; Tp[4 * q + 2] = t.z; Tp[4 * q + 3] = t.w;
        Mp[4 * q] = m.x; Mp[4 * q + 1] = m.y; Mp[4 * q + 2] = m.z; Mp[4 * q + 3] = m.w;
        Sp_[4 * q] = sv.x; Sp_[4 * q + 1] = sv.y; Sp_[4 * q + 2] = sv.z; Sp_[4 * q + 3] = sv.w;
      }
    }
  }

  // ---- tap tables of every star of the frame: once before the first iteration, then right behind the star update of every
  //      iteration - in the two-workgroup form between the hand-off stores and their drain, whose latency they hide ----------
  auto compute_taps = [&](int tid) {
    for (int e = tid; e < S * 2 * NTP; e += NTHR) {
      const int s = e / (2 * NTP), ax = (e / NTP) & 1, k = e % NTP - 1;  // k = -1, NT, NT + 1: the zero padding
      float tap = 0.f, dtap = 0.f;
      int bq = 0;
      const float c_off = (N % 2 == 0) ? 0.5f : 0.0f;  // (N-1)/2 - (N-1)//2
      // a star further than N/4 high-res pixels from the stamp centre is a broken fit: pin the kernel there
      const float delta = fminf(fmaxf(SS * SP[s * 4 + 1 + ax], -(float)(N / 4)), (float)(N / 4)) + c_off;
      if (k >= 0 && k < NT) tap_entry<SS, NT>(delta, k, tap, dtap, bq);
      TAPS[(s * 4 + 2 * ax) * NTP + k] = tap;
      TAPS[(s * 4 + 2 * ax + 1) * NTP + k] = dtap;
      if (k >= 0 && k < NT) TAPP[(s * 2 + ax) * NT + k] = (lc_v2f){tap, dtap};
      if (k == 0) BQ[s * 2 + ax] = bq;
    }
  };
  // AdaBelief step of a, x0, y0 of every star (threads tid < 3 S) and the tap tables of the next iteration.  Needs the
  // complete star gradients (SGR) and this iteration's schedule (SCAL); every thread of the workgroup calls (barrier inside).
  auto update_stars_and_taps = [&](int tid) {
#pragma clang fp contract(off)
    if (tid < S * 3) {
      const float lr = SCAL[0], bc1 = SCAL[1], bc2 = SCAL[2];
      const float b1 = A.ab.b1, b2 = A.ab.b2, eps = A.ab.eps, eps_root = A.ab.eps_root;
      const int s = tid / 3, q = tid % 3;  // a, x0, y0
      const float g = SGR[s * 5 + 1 + q];
      const float mn = b1 * SM[s * 4 + q] + (1.f - b1) * g;
      const float dg = g - mn;
      const float sn = b2 * SV[s * 4 + q] + (1.f - b2) * dg * dg + eps_root;
      SM[s * 4 + q] = mn;
      SV[s * 4 + q] = sn;
      SP[s * 4 + q] -= lr * (mn * bc1) / (sqrtf(sn * bc2) + eps);
    }
    __syncthreads();
    compute_taps(tid);
  };
  if (conv_role) compute_taps(tid0);  // (visible behind the barrier that opens the first group of stars)
#if LC_PSF_WAVEFLAGS
  if constexpr (SPLIT) {
    if (tid0 == 0) ((int *)(SCAL + 5))[0] = 1;  // (per-wave hand-off: only ever cleared; read behind the iteration's barriers)
  }
#endif
  // (measured and left out: s_setprio 1 for the second-dispatched half of the workgroup - MI355X_MICROARCH.md, two waves per
  //  SIMD - C2 16.2 us per iteration with and without, C3 shard 86.8 / 86.9)
  for (int it = 0; it < A.n_iter; ++it) {
    const int tglob = A.t0 + it;
    if constexpr (SPLIT) {
      if (A.force_abort_it == it && f == 0 && role == 1) {  // test hook: this workgroup stops showing up
        if (tid0 == 0) __hip_atomic_store(A.xabort, A.launch_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    int tid = tid0;
    LC_LAUNDER(tid);  // everything derived from tid is recomputed per iteration, not kept live
    const int lane = tid & 63, wid = tid >> 6;
    const int pu = tid / (N / PX);         // owned row
    const int pv = (tid % (N / PX)) * PX;  // first owned column
    const size_t gpix = (size_t)f * N * N + (size_t)pu * N + pv;
    if (tid < 3 && A.mode == 1) SCAL[tid] = A.sched[3 * tglob + tid];  // lr, bc1, bc2
    LC_STAMP(0);
    float gB[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) gB[p] = 0.f;
    float tpix[PX];  // T at the thread's own pixels (WC layout)
    if (conv_role) {
    // ---- P1: T = Moffat + B into LDS -----------------------------------------------------
    if (C::WC && STATE_REGS) {
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        tpix[p] = Bp[p] + Tp[p];
        T[pu * C::TSA + C::AP + pv + p] = tpix[p];
      }
    } else if (STATE_REGS) {
#pragma unroll
      for (int p = 0; p < PX; ++p) T[pu * TS + pv + p] = Bp[p] + Tp[p];
    } else {
      const float4 *bp = (const float4 *)(Bg + gpix);
      const float4 *tp = (const float4 *)(A.Tm + gpix);
#pragma unroll
      for (int q = 0; q < PX / 4; ++q) {
        const float4 b = bp[q], t = tp[q];
        T[pu * TS + pv + 4 * q + 0] = b.x + t.x;
        T[pu * TS + pv + 4 * q + 1] = b.y + t.y;
        T[pu * TS + pv + 4 * q + 2] = b.z + t.z;
        T[pu * TS + pv + 4 * q + 3] = b.w + t.w;
      }
    }
    if (tid < S * 5) SGR[tid] = 0.f;
    // (the tap tables of this iteration were made at the end of the previous one, behind the star update: compute_taps)

    for (int g0 = 0; g0 < S; g0 += SG) {
      __syncthreads();  // T and taps visible; previous group's P5 (reads V) done before P2 rewrites R
      LC_STAMP(1 + 5 * (g0 / SG));
      if constexpr (C::WC) {
        // ---- wave tasks: (star, block of JB down-sampled columns), everything up to V inside the wave ----
        constexpr int JB = C::JB, NBLK = n / JB;
        float *wsc = lds + C::OFF_WSC + wid * C::WSZ;
        constexpr int TSA = C::TSA, AP = C::AP, APR = C::APR;
        float *R2w = wsc + AP, *RESw = wsc + JB * TSA + APR * C::JBP;
        // (Two waves share a SIMD and the arbiter favours the older one: waves 0-3 are done after ~14 k cycles, waves 4-7
        // after ~17.5 k, running alone - at 1 / 1.75 of the pair's rate - for the rest; tools/psf_stamps.py prints the times.
        // Measured and left out: handing the tasks out from an LDS counter.  With four tasks of 3.5 k cycles per wave the
        // older waves simply take a fifth one each and the phase ends at the same time: 16.2 against 15.9 us per iteration.)
        int kdone = 0;  // tasks this wave has finished: the wave that is behind gets the issue priority (progress_prio)
        for (int task = wid; task < SG * NBLK; task += C::NW, ++kdone) {
          progress_prio(kdone);
          const int sl = task / NBLK, blk = task % NBLK, s = g0 + sl;
          if (s >= S) continue;  // wave-uniform
          const int jd0 = blk * JB;
          const float *tx = TAPS + (s * 4 + 0) * NTP, *ty = tx + 2 * NTP, *dty = tx + 3 * NTP;
          const int bqx = BQ[s * 2 + 0], bqy = BQ[s * 2 + 1];
          const float amp = SP[s * 4 + 0], sky = SP[s * 4 + 3];
          // data / weights of the task's pixels: requested now, used after the row pass
          // (measured and left out: requesting them one task ahead - 8 more registers in flight, 5 spilled: 16.6 against 15.8 us)
          constexpr int NI3 = (JB * (n / LC) + 63) / 64;
          float dpre[NI3][LC], wpre[NI3][LC];
#pragma unroll
          for (int i3 = 0; i3 < NI3; ++i3) {
            const int item = lane + 64 * i3;
            const bool ok = item < JB * (n / LC);
            const int jl = item % JB, a0 = (item / JB) * LC;
#pragma unroll
            for (int j = 0; j < LC; ++j) {
              const size_t pix = ok ? ((size_t)s * n * n + (size_t)(a0 + j) * n + jd0 + jl) : 0;
              const float dv = dataf[pix], wv = wgtf[pix];
              dpre[i3][j] = ok ? dv : 0.f;
              wpre[i3][j] = ok ? wv : 0.f;
            }
          }
          LC_STAMP(10);
          // row pass (x taps) fused with the column down-sampling: lane = high-res row.  Only the value filter runs
          // here; the x0 gradient is taken from the transposed row pass (P5), which has the derivative taps for free
          // in the second half of its packed FMAs.
          {
            const int ws = SS * (jd0 - bqx) - (NT - 1);
            if constexpr (SS == 2) {
              // out_j = sum_m tr[m] win[2 j + m], tr[m] = tx[NT - 1 - m]: taps and window samples pair up as
              // (m, m + 1), m even, in aligned registers; the two halves of the packed accumulator are added at the end
              constexpr int NP = (NT + 1) / 2, WL2 = (JB - 1) + NP;
              lc_v2f tr2[NP];
#pragma unroll
              for (int h = 0; h < NP; ++h) tr2[h] = (lc_v2f){tx[NT - 1 - 2 * h], tx[NT - 2 - 2 * h]};  // tx[-1] = 0
              for (int u = lane; u < N; u += 64) {
                lc_v2f win2[WL2];
                const float *trow = T + u * TSA + AP + ws;
#pragma unroll
                for (int i = 0; i < WL2; ++i) win2[i] = (lc_v2f){trow[2 * i], trow[2 * i + 1]};
                // tap-major order: the JB accumulators are independent chains, so consecutive packed FMAs never wait
                // for each other (output-major order made one 7-long dependent chain per output)
                lc_v2f acc[JB];
#pragma unroll
                for (int j = 0; j < JB; ++j) acc[j] = (lc_v2f){0.f, 0.f};
#pragma unroll
                for (int h = 0; h < NP; ++h) {
#pragma unroll
                  for (int j = 0; j < JB; ++j) acc[j] = pk_fma(tr2[h], win2[j + h], acc[j]);
                }
#pragma unroll
                for (int j = 0; j < JB; ++j) R2w[j * TSA + u] = acc[j].x + acc[j].y;
              }
            } else {
              constexpr int WL = SS * (JB - 1) + NT;
              float tk[NT];
#pragma unroll
              for (int k = 0; k < NT; ++k) tk[k] = tx[k];
              for (int u = lane; u < N; u += 64) {
                float win[WL];
                const float *trow = T + u * TSA + AP + ws;
#pragma unroll
                for (int i = 0; i < WL; ++i) win[i] = trow[i];
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                  float acc = 0.f;
#pragma unroll
                  for (int k = 0; k < NT; ++k) acc = fmaf(tk[k], win[SS * j - k + NT - 1], acc);
                  R2w[j * TSA + u] = acc;
                }
              }
            }
          }
          wave_lds_sync();
          LC_STAMP(11);
          // column pass (y taps) fused with the row down-sampling, residuals, reductions: lane = (column, row strip);
          // value and y-derivative share the window sample: one packed FMA per tap
          float chi = 0.f, ga = 0.f, gy = 0.f, gs = 0.f;
          {
            constexpr int WL = SS * (LC - 1) + NT;
            lc_v2f tyd[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k) tyd[k] = TAPP[(s * 2 + 1) * NT + k];
#pragma unroll
            for (int i3 = 0; i3 < NI3; ++i3) {
              const int item = lane + 64 * i3;
              if (item < JB * (n / LC)) {
                const int jl = item % JB, a0 = (item / JB) * LC;
                const int ws = SS * (a0 - bqy) - (NT - 1);
                float win[WL];
                const float *r2 = R2w + jl * TSA + ws;
#pragma unroll
                for (int i = 0; i < WL; ++i) win[i] = r2[i];
                float lgy = 0.f;
                lc_v2f fvys[LC];  // tap-major order: LC independent chains
#pragma unroll
                for (int j = 0; j < LC; ++j) fvys[j] = (lc_v2f){0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NT; ++k) {
#pragma unroll
                  for (int j = 0; j < LC; ++j) fvys[j] = pk_fma(tyd[k], pk_bcast(win[SS * j - k + NT - 1]), fvys[j]);
                  // keeps the scheduler from folding the LC chains back into one (it does, to save registers)
                  if constexpr (LC == 4) asm volatile("" : "+v"(fvys[0]), "+v"(fvys[1]), "+v"(fvys[2]), "+v"(fvys[3]));
                  else asm volatile("" : "+v"(fvys[0]), "+v"(fvys[1]));
                }
#pragma unroll
                for (int j = 0; j < LC; ++j) {
                  const float fv = fvys[j].x, fy = fvys[j].y;
                  const float model = fmaf(amp, fv, sky);
                  const float res = model - dpre[i3][j];
                  const float rw = wpre[i3][j] * res;
                  chi = fmaf(rw, res, chi);
                  ga = fmaf(rw, fv, ga);
                  lgy = fmaf(rw, fy, lgy);
                  gs += rw;
                  RESw[(a0 + j) * C::JBP + jl] = rw;
                  if (A.out_model) A.out_model[(size_t)f * S * n * n + (size_t)s * n * n + (size_t)(a0 + j) * n + jd0 + jl] = model;
                }
                gy += lgy * amp * SS;
              }
            }
          }
          chi = wave_sum(chi);
          ga = wave_sum(ga);
          gy = wave_sum(gy);
          gs = wave_sum(gs);
          if (lane == 0) {
            float *r = RED + (s * NBLK + blk) * 5;
            r[0] = chi;
            r[1] = ga;
            r[3] = gy;
            r[4] = gs;
          }
          wave_lds_sync();
          LC_STAMP(12);
          // transposed column pass: V[u][jd] = sum_id PhiY(ss*id - u) r[id][jd]: lane = (column, strip of high-res rows)
          {
            constexpr int WI = (SS * LA - 1 + NT - 1) / SS + 1;
            constexpr int NI4 = (JB * (n / LA) + 63) / 64;
            if constexpr (SS == 2) {
              // outputs pair up as (2 h, 2 h + 1): residual i feeds them through taps (2 (i - h), 2 (i - h) - 1)
              constexpr int NP = (NT + 1) / 2;
              lc_v2f py[NP];
#pragma unroll
              for (int q = 0; q < NP; ++q) py[q] = (lc_v2f){ty[2 * q], ty[2 * q - 1]};  // ty[-1] = ty[NT] = 0
#pragma unroll
              for (int i4 = 0; i4 < NI4; ++i4) {
                const int item = lane + 64 * i4;
                if (item < JB * (n / LA)) {
                  const int jl = item % JB, a0 = (item / JB) * LA;
                  lc_v2f out2[LA];
#pragma unroll
                  for (int h = 0; h < LA; ++h) out2[h] = (lc_v2f){0.f, 0.f};
                  const float *rcol = RESw + (bqy + a0) * C::JBP + jl;
#pragma unroll
                  for (int i = 0; i < WI; ++i) {
                    const lc_v2f rv = pk_bcast(rcol[i * C::JBP]);
#pragma unroll
                    for (int h = 0; h < LA; ++h)
                      if (i - h >= 0 && i - h < NP) out2[h] = pk_fma(py[i - h], rv, out2[h]);
                  }
                  float *vrow = V + (sl * N + SS * a0) * C::VS + C::VO + jd0 + jl;
#pragma unroll
                  for (int h = 0; h < LA; ++h) {
                    vrow[(2 * h) * C::VS] = out2[h].x;
                    vrow[(2 * h + 1) * C::VS] = out2[h].y;
                  }
                  // the zero columns (the starlet phase reuses this region, so they are rewritten every iteration)
                  if (jd0 + jl == 0) {
#pragma unroll
                    for (int r = 0; r < SS * LA; ++r) V[(sl * N + SS * a0 + r) * C::VS] = 0.f;
                  }
                  if (jd0 + jl == n - 1) {
#pragma unroll
                    for (int r = 0; r < SS * LA; ++r) V[(sl * N + SS * a0 + r) * C::VS + n + 1] = 0.f;
                  }
                }
              }
            } else {
              float tk[NT];
#pragma unroll
              for (int k = 0; k < NT; ++k) tk[k] = ty[k];
#pragma unroll
              for (int i4 = 0; i4 < NI4; ++i4) {
                const int item = lane + 64 * i4;
                if (item < JB * (n / LA)) {
                  const int jl = item % JB, a0 = (item / JB) * LA;
                  float out[SS * LA];
#pragma unroll
                  for (int r = 0; r < SS * LA; ++r) out[r] = 0.f;
                  const float *rcol = RESw + (bqy + a0) * C::JBP + jl;
#pragma unroll
                  for (int i = 0; i < WI; ++i) {
                    const float rv = rcol[i * C::JBP];
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                      const int rel = SS * i - k;
                      if (rel >= 0 && rel < SS * LA) out[rel] = fmaf(tk[k], rv, out[rel]);
                    }
                  }
#pragma unroll
                  for (int r = 0; r < SS * LA; ++r) V[(sl * N + SS * a0 + r) * C::VS + C::VO + jd0 + jl] = out[r];
                  if (jd0 + jl == 0) {
#pragma unroll
                    for (int r = 0; r < SS * LA; ++r) V[(sl * N + SS * a0 + r) * C::VS] = 0.f;
                  }
                  if (jd0 + jl == n - 1) {
#pragma unroll
                    for (int r = 0; r < SS * LA; ++r) V[(sl * N + SS * a0 + r) * C::VS + n + 1] = 0.f;
                  }
                }
              }
            }
          }
          wave_lds_sync();  // the scratch is rewritten by the wave's next task
          LC_STAMP(13);
        }
        __builtin_amdgcn_s_setprio(0);
        LC_STAMP(14);
#ifdef LC_STAMPS
        if (blockIdx.x == 0 && lane == 0 && it == A.n_iter - 1) g_stamps[20 + wid] = clock64();  // when each wave is done with its tasks
#endif
        __syncthreads();
        LC_STAMP(15);
        if (tid < SG * 5) {
          const int sl = tid / 5, q = tid % 5;
          if (g0 + sl < S && q != 2) {  // the x0 gradient (slot 2) comes out of the transposed row pass below
            float acc = 0.f;
            for (int k = 0; k < NBLK; ++k) acc += RED[((g0 + sl) * NBLK + k) * 5 + q];
            SGR[(g0 + sl) * 5 + q] = acc;
          }
        }
      } else {
      // data / weights of this group's column-pass pixels: issued now, consumed in P3, so the L2/HBM
      // latency hides behind the row pass
      constexpr int NIT3 = (SG * C::IPS_PAD + NTHR - 1) / NTHR;
      float dpre[NIT3][LC], wpre[NIT3][LC];
#pragma unroll
      for (int i3 = 0; i3 < NIT3; ++i3) {
        const int item0 = wid * 64 + i3 * NTHR, item = item0 + lane;
        const int sl = item0 / C::IPS_PAD, within = item % C::IPS_PAD, s = g0 + sl;
        const bool ok = (item0 < SG * C::IPS_PAD) && (s < S) && (within < C::IPS);
        const int jd = within % n, a0 = (within / n) * LC;
#pragma unroll
        for (int j = 0; j < LC; ++j) {
          const size_t pix = ok ? ((size_t)s * n * n + (size_t)(a0 + j) * n + jd) : 0;
          const float dv = dataf[pix], wv = wgtf[pix];
          dpre[i3][j] = ok ? dv : 0.f;
          wpre[i3][j] = ok ? wv : 0.f;
        }
      }
      // ---- P2: row pass (x taps) fused with the column down-sampling ----------------------
      {
        constexpr int WL = SS * (LR - 1) + NT;
        constexpr int NSTRIP = n / LR;
        for (int item = tid; item < SG * N * NSTRIP; item += NTHR) {
          const int u = item % N, strip = (item / N) % NSTRIP, sl = item / (N * NSTRIP);
          if (g0 + sl >= S) continue;
          const float *tx = TAPS + ((g0 + sl) * 4 + 0) * NTP;
          const int bq = BQ[(g0 + sl) * 2 + 0];
          const int a0 = strip * LR;
          const int ws = SS * (a0 - bq) - (NT - 1);
          float win[WL];
#pragma unroll
          for (int i = 0; i < WL; ++i) {
            const int idx = ws + i;
            const int ci = min(max(idx, 0), N - 1);
            const float v = T[u * TS + ci];
            win[i] = (idx >= 0 && idx < N) ? v : 0.f;
          }
          float tk[NT];
#pragma unroll
          for (int k = 0; k < NT; ++k) tk[k] = tx[k];
#pragma unroll
          for (int j = 0; j < LR; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < NT; ++k) acc = fmaf(tk[k], win[SS * j - k + NT - 1], acc);
            R2t[(sl * n + a0 + j) * TS + u] = acc;
          }
        }
      }
      __syncthreads();
      LC_STAMP(2 + 5 * (g0 / SG));
      // ---- P3: column pass (y taps) fused with row down-sampling, residuals, reductions ----
      {
        constexpr int WL = SS * (LC - 1) + NT;
        constexpr int NSTRIP = n / LC;
#pragma unroll
        for (int i3 = 0; i3 < NIT3; ++i3) {
          const int item0 = wid * 64 + i3 * NTHR;
          if (item0 >= SG * C::IPS_PAD) break;
          const int item = item0 + lane;
          const int sl = item0 / C::IPS_PAD;  // wave-uniform
          const int within = item % C::IPS_PAD;
          const int s = g0 + sl;
          float chi = 0.f, ga = 0.f, gx = 0.f, gy = 0.f, gs = 0.f;
          if (s < S && within < C::IPS) {
            const int jd = within % n, strip = within / n;
            const float *ty = TAPS + (s * 4 + 2) * NTP, *dty = ty + NTP;
            const int bq = BQ[s * 2 + 1];
            const int a0 = strip * LC;
            const int ws = SS * (a0 - bq) - (NT - 1);
            const float amp = SP[s * 4 + 0], sky = SP[s * 4 + 3];
            float win[WL];
#pragma unroll
            for (int i = 0; i < WL; ++i) {
              const int idx = ws + i;
              const int ci = min(max(idx, 0), N - 1);
              const bool ok = (idx >= 0 && idx < N);
              const float v = R2t[(sl * n + jd) * TS + ci];
              win[i] = ok ? v : 0.f;
            }
            float tk[NT], dk[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k) {
              tk[k] = ty[k];
              dk[k] = dty[k];
            }
#pragma unroll
            for (int j = 0; j < LC; ++j) {
              float fv = 0.f, fy = 0.f;
#pragma unroll
              for (int k = 0; k < NT; ++k) {
                const float w = win[SS * j - k + NT - 1];
                fv = fmaf(tk[k], w, fv);
                fy = fmaf(dk[k], w, fy);
              }
              const int id = a0 + j;
              const size_t pix = (size_t)s * n * n + (size_t)id * n + jd;
              const float d = dpre[i3][j], w = wpre[i3][j];
              const float model = fmaf(amp, fv, sky);
              const float res = model - d;
              const float rw = w * res;
              chi = fmaf(rw, res, chi);
              ga = fmaf(rw, fv, ga);
              gy = fmaf(rw, fy, gy);
              gs += rw;
              RES[(sl * n + id) * n + jd] = rw;
              if (A.out_model) A.out_model[(size_t)f * S * n * n + pix] = model;
            }
            gy *= amp * SS;
          }
          chi = wave_sum(chi);
          ga = wave_sum(ga);
          gx = wave_sum(gx);
          gy = wave_sum(gy);
          gs = wave_sum(gs);
          if (lane == 0) {
            float *r = RED + (sl * C::SLOTS + (item0 % C::IPS_PAD) / 64) * 5;
            r[0] = chi;
            r[1] = ga;
            r[2] = gx;
            r[3] = gy;
            r[4] = gs;
          }
        }
      }
      __syncthreads();
      LC_STAMP(3 + 5 * (g0 / SG));
      // per-star sums in fixed slot order
      if (tid < SG * 5) {
        const int sl = tid / 5, q = tid % 5;
        if (g0 + sl < S && q != 2) {  // the x0 gradient (slot 2) comes out of the transposed row pass (P5)
          float acc = 0.f;
          for (int k = 0; k < C::SLOTS; ++k) acc += RED[(sl * C::SLOTS + k) * 5 + q];
          SGR[(g0 + sl) * 5 + q] = acc;
        }
      }
      // ---- P4: transposed column pass: V[u][jd] = sum_id PhiY(ss*id - u) r[id][jd] ---------
      {
        constexpr int WI = (SS * LA - 1 + NT - 1) / SS + 1;
        constexpr int NSTRIP = n / LA;
        for (int item = tid; item < SG * n * NSTRIP; item += NTHR) {
          const int jd = item % n, strip = (item / n) % NSTRIP, sl = item / (n * NSTRIP);
          if (g0 + sl >= S) continue;
          const float *ty = TAPS + ((g0 + sl) * 4 + 2) * NTP;
          const int bq = BQ[(g0 + sl) * 2 + 1];
          const int a0 = strip * LA;
          float tk[NT];
#pragma unroll
          for (int k = 0; k < NT; ++k) tk[k] = ty[k];
          float out[SS * LA];
#pragma unroll
          for (int r = 0; r < SS * LA; ++r) out[r] = 0.f;
#pragma unroll
          for (int i = 0; i < WI; ++i) {
            const int id = i + bq + a0;
            const int ci = min(max(id, 0), n - 1);
            float rv = RES[(sl * n + ci) * n + jd];
            rv = (id >= 0 && id < n) ? rv : 0.f;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
              const int rel = SS * i - k;
              if (rel >= 0 && rel < SS * LA) out[rel] = fmaf(tk[k], rv, out[rel]);
            }
          }
#pragma unroll
          for (int r = 0; r < SS * LA; ++r) V[(sl * N + SS * a0 + r) * RS + jd] = out[r];  // non-WC layout: VS == RS
        }
      }
      __syncthreads();
      LC_STAMP(4 + 5 * (g0 / SG));
      }
      // ---- P5: transposed row pass, summed over the stars of the group into registers ------
      LC_STAMP(16);
      if constexpr (C::WC) {
        // value and derivative taps ride in the two halves of one packed FMA: accp[p] = (sum_k tx V, sum_k dtx V).
        // The first half is this star's share of dchi2/dB at the thread's pixels; the second, multiplied by T at the
        // same pixels and summed over the frame, is dchi2/dx0 / (amp * SS) of the star (the adjoint form of the
        // x-derivative filter of the forward pass: same number, no second row pass, no second column pass).
        constexpr int WJ = (PX - 1 + NT - 1) / SS + 1;
        for (int sl = 0; sl < SG; ++sl) {
          const int s = g0 + sl;
          if (s >= S) break;
          progress_prio((4 * sl) / SG);  // (SG stars in four steps)
          const float *tx = TAPS + (s * 4 + 0) * NTP, *dtx = tx + NTP;
          const int bq = BQ[s * 2 + 0];
          const float amp = SP[s * 4 + 0];
          lc_v2f txd[NT];  // (value tap, derivative tap) pairs of this star, in registers for the whole window
#pragma unroll
          for (int k = 0; k < NT; ++k) txd[k] = TAPP[(s * 2 + 0) * NT + k];
          lc_v2f accp[PX];
#pragma unroll
          for (int p = 0; p < PX; ++p) accp[p] = (lc_v2f){0.f, 0.f};
          const float *vrow = V + (sl * N + pu) * C::VS + 1;
          const int jd0 = bq + pv / SS;
          float vwin[WJ];
#pragma unroll
          for (int i = 0; i < WJ; ++i) vwin[i] = vrow[min(max(jd0 + i, -1), n)];
#pragma unroll
          for (int i = 0; i < WJ; ++i) {
            const lc_v2f vv = pk_bcast(vwin[i]);
#pragma unroll
            for (int k = 0; k < NT; ++k) {
              const int rel = SS * i - k;
              if (rel >= 0 && rel < PX) accp[rel] = pk_fma(txd[k], vv, accp[rel]);
            }
          }
          float gxs = 0.f;
#pragma unroll
          for (int p = 0; p < PX; ++p) {
            gB[p] = fmaf(amp, accp[p].x, gB[p]);
            gxs = fmaf(accp[p].y, tpix[p], gxs);
          }
          gxs = wave_sum(gxs);
          if (lane == 0) REDX[s * C::NW + wid] = gxs;
        }
        __builtin_amdgcn_s_setprio(0);
      } else {
        constexpr int WJ = (PX - 1 + NT - 1) / SS + 1;
        for (int sl = 0; sl < SG; ++sl) {
          const int s = g0 + sl;
          if (s >= S) break;
          const float *tx = TAPS + (s * 4 + 0) * NTP, *dtx = tx + NTP;
          const int bq = BQ[s * 2 + 0];
          const float amp = SP[s * 4 + 0];
          float acc[PX], gxs = 0.f;
#pragma unroll
          for (int p = 0; p < PX; ++p) acc[p] = 0.f;
          // in halves of the thread's pixels: value taps into acc, derivative taps straight into the x0 gradient
          // (dotted with T, still in LDS), so that no second accumulator array is live
#pragma unroll
          for (int i = 0; i < WJ; ++i) {
            const int jd = i + bq + pv / SS;
            const int ci = min(max(jd, 0), n - 1);
            float vv = V[(sl * N + pu) * RS + ci];
            vv = (jd >= 0 && jd < n) ? vv : 0.f;
            float dsum = 0.f;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
              const int rel = SS * i - k;
              if (rel >= 0 && rel < PX) {
                acc[rel] = fmaf(tx[k], vv, acc[rel]);
                dsum = fmaf(dtx[k], T[pu * TS + pv + rel], dsum);
              }
            }
            gxs = fmaf(vv, dsum, gxs);
          }
#pragma unroll
          for (int p = 0; p < PX; ++p) gB[p] = fmaf(amp, acc[p], gB[p]);
          gxs = wave_sum(gxs);
          if (lane == 0) REDX[s * C::NW + wid] = gxs;
        }
      }
      LC_STAMP(17);
    }  // groups
    if (A.ext_grad) {
#pragma unroll
      for (int p = 0; p < PX; ++p) gB[p] += A.ext_grad[gpix + p];
    }
    }  // conv_role
    __syncthreads();
    LC_STAMP(40);
    {
      if (conv_role && tid < S) {  // dchi2/dx0 of every star: the waves' partial sums in fixed order
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < C::NW; ++w) acc += REDX[tid * C::NW + w];
        SGR[tid * 5 + 2] = acc * SP[tid * 4 + 0] * SS;
      }
    }

    if (A.out_gT) {
#pragma unroll
      for (int p = 0; p < PX; ++p) A.out_gT[gpix + p] = gB[p];
    }

    // ---- P6: starlet l1 on B: value + sub-gradient (starlet_device.h) ---------------------------
    float l1 = 0.f;
    float z[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) z[p] = 0.f;
    if (starlet_role && (A.lam_sc != 0.f || A.lam_hf != 0.f)) {
      float bpix[PX];
      if (STATE_REGS) {
#pragma unroll
        for (int p = 0; p < PX; ++p) bpix[p] = Bp[p];
      } else {
        const float4 *bp = (const float4 *)(Bg + gpix);
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
          const float4 b = bp[q];
          bpix[4 * q] = b.x;
          bpix[4 * q + 1] = b.y;
          bpix[4 * q + 2] = b.z;
          bpix[4 * q + 3] = b.w;
        }
      }
      starlet_l1_grad<N, PX>(bpix, A.W ? A.W + (size_t)f * J * N * N : nullptr, A.norms,
                             A.qscratch + (size_t)f * J * N * N, A.lam_sc, A.lam_hf, C::WC ? lds + C::OFF_R : lds, tid, l1, z);
    }
    LC_STAMP(42);
    // ---- loss ------------------------------------------------------------------------------
    {
      const float wl = wave_sum(l1);
      if (lane == 0) REDW[wid] = wl;
    }
    __syncthreads();
    float tl1 = 0.f;
    if (tid == 0 && starlet_role)
      for (int w = 0; w < C::NW; ++w) tl1 += REDW[w];
    LC_STAMP(44);
    if constexpr (SPLIT) {
      constexpr int XS = N * N + 64;
      float *mine = A.xch + (((size_t)f * 2 + (it & 1)) * 2 + role) * XS;
      const float *theirs = A.xch + (((size_t)f * 2 + (it & 1)) * 2 + (1 - role)) * XS;
      const int poff = pu * N + pv;
      // Partners on ONE XCD (each reads HW_REG_XCC_ID and tells the other in the first hand-off of the launch; a workgroup
      // never moves) share an L2: plain stores keep their lines there and the partner's L1-bypassing loads are served from
      // it, where an sc1 store drops the line and the partner reads at the memory-side rate (MI355X_MICROARCH.md, "stores
      // of each flavour").  Anything else - another XCD, the first iteration, xcd_fast off - takes the sc1 stores.
#pragma unroll
      for (int q = 0; q < PX / 4; ++q) {
        lc_v4f v;
        v.x = role ? z[4 * q] : gB[4 * q];
        v.y = role ? z[4 * q + 1] : gB[4 * q + 1];
        v.z = role ? z[4 * q + 2] : gB[4 * q + 2];
        v.w = role ? z[4 * q + 3] : gB[4 * q + 3];
        if (same_xcd) store_plain_x4(mine + poff + 4 * q, v);
        else store_sc1_x4(mine + poff + 4 * q, v);
      }
      if (tid == 0 && role == 1) {
        if (same_xcd) asm volatile("global_store_dword %0, %1, off" ::"v"(mine + N * N), "v"(tl1) : "memory");
        else store_sc1_f(mine + N * N, tl1);
      }
      if (tid == 0 && it == 0) store_sc1_f(mine + N * N + 1, __int_as_float(my_xcc + 1));
      // role 0: the stars' step and the next tap tables need nothing from the partner: done while the stores drain
      if (role == 0 && A.mode == 1) update_stars_and_taps(tid);
#if LC_PSF_WAVEFLAGS
      // (-DLC_PSF_WAVEFLAGS=1; built in round 4 and measured NOT faster: C2 14.8 against 14.8 - 14.9 us per iteration (54.0 - 54.2
      //  against 53.7 - 53.9 M cutouts/s), C3 shard 90.4 against 82.7 - sixteen polling lanes per workgroup at N = 128.  The one-lane
      //  form below stays the default.)
      // Per-WAVE flags: wave w of role 0 and wave w of role 1 own the same 64 PX pixels, so a wave hands over as soon as ITS
      // stores have drained and reads as soon as its partner wave has published - no workgroup barrier in front of the flag,
      // eight polling lanes instead of one, and a wave does not wait for the slowest wave of either workgroup.  One barrier
      // remains behind the reads: it makes the give-up decision uniform (a wave whose partner does not show up clears OK[0];
      // nobody leaves the loop before everybody has seen it) and publishes what the first hand-off found out about the XCDs.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      LC_STAMP(45);
      int *OK = (int *)(SCAL + 5);
      int wave_ok = 1;
      if (lane == 0) {
        int *myflag = A.xflags + ((size_t)f * 2 + role) * kXFlagStride + wid;
        const int *theirflag = A.xflags + ((size_t)f * 2 + (1 - role)) * kXFlagStride + wid;
        if (same_xcd) asm volatile("global_store_dword %0, %1, off" ::"v"(myflag), "v"(it + 1) : "memory");
        else __hip_atomic_store(myflag, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(theirflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it + 1) {
          __builtin_amdgcn_s_sleep(LC_XPOLL_SLEEP);
          ++spins;
          // exit condition every wave reaches: a partner that never shows up is reported, not waited for
          if (spins > (1 << 21) ||
              ((spins & 255) == 0 && __hip_atomic_load(A.xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.launch_seq)) {
            __hip_atomic_store(A.xabort, A.launch_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wave_ok = 0;
            OK[0] = 0;
            break;
          }
        }
        if (tid == 0 && it == 0 && wave_ok)
          OK[1] = (A.xcd_fast && __float_as_int(load_sc1_f(theirs + N * N + 1)) == my_xcc + 1) ? 1 : 0;
      }
      wave_ok = __builtin_amdgcn_readfirstlane(wave_ok);
#else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      LC_STAMP(45);
      int *OK = (int *)(SCAL + 5);
      if (tid == 0) {
        // (same XCD: the flag too stays in the shared L2, where the partner's L1-bypassing poll finds it)
        if (same_xcd) asm volatile("global_store_dword %0, %1, off" ::"v"(A.xflags + ((size_t)f * 2 + role) * kXFlagStride), "v"(it + 1) : "memory");
        else __hip_atomic_store(A.xflags + ((size_t)f * 2 + role) * kXFlagStride, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1, spins = 0;
        while (__hip_atomic_load(A.xflags + ((size_t)f * 2 + (1 - role)) * kXFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it + 1) {
          // (the interval between polls is no lever: s_sleep 0 / 2 / 6 / 14 measured at 14.8 - 15.0 us per iteration of C2 and
          //  82.2 - 83.0 of the C3 shard, all within the run-to-run spread; -DLC_XPOLL_SLEEP=<n> for an A/B build)
          __builtin_amdgcn_s_sleep(LC_XPOLL_SLEEP);
          ++spins;
          // exit condition every workgroup reaches: a partner that never shows up is reported, not waited for
          if (spins > (1 << 21) ||
              ((spins & 255) == 0 && __hip_atomic_load(A.xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.launch_seq)) {
            __hip_atomic_store(A.xabort, A.launch_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
        }
        if (it == 0 && ok) OK[1] = (A.xcd_fast && __float_as_int(load_sc1_f(theirs + N * N + 1)) == my_xcc + 1) ? 1 : 0;
        *OK = ok;
#ifdef LC_XCH_ACQUIRE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      }
#endif
#if !LC_PSF_WAVEFLAGS
      __syncthreads();
      LC_STAMP(46);
      if (*OK == 0) break;
      if (it == 0) same_xcd = OK[1];
#endif
      // (measured and left out: a first look at the partner's flag requested while the own stores drain, so that the role
      //  that arrives second skips its first poll - 15.2 against 15.0 us per iteration on one box, C3 shard 83.8 / 82.6)
      // (measured and left out: role 0, the longer of the two, asking for the partner's half BEFORE publishing its own - the
      //  partner has published long before - with loads the compiler tracks (agent-scope relaxed atomics, one dword each):
      //  17.0 against 14.8 us per iteration; the 16-byte form would need registers that are in flight across compiled code)
      float other[PX];
      float tl1_other = 0.f;  // (every thread reads the partner's scalar along with its pixels: one line, no second round trip)
      if constexpr (PX == 8) {
        lc_v4f a, b;
        load_sc1_2x4_1(theirs + poff, a, b, theirs + N * N, tl1_other);
        other[0] = a.x; other[1] = a.y; other[2] = a.z; other[3] = a.w;
        other[4] = b.x; other[5] = b.y; other[6] = b.z; other[7] = b.w;
      } else {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
          lc_v4f a;
          load_sc1_x4(theirs + poff + 4 * q, a);
          other[4 * q] = a.x; other[4 * q + 1] = a.y; other[4 * q + 2] = a.z; other[4 * q + 3] = a.w;
        }
      }
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        if (role) gB[p] = other[p]; else z[p] = other[p];
      }
      if (tid == 0 && role == 0) tl1 = (PX == 8) ? tl1_other : load_sc1_f(theirs + N * N);
#if LC_PSF_WAVEFLAGS
      __syncthreads();
      LC_STAMP(46);
      if (*OK == 0) break;
      if (it == 0) same_xcd = OK[1];
#endif
      LC_STAMP(47);
    }
    if (tid == 0 && conv_role) {
      float chi = 0.f;
      for (int s = 0; s < S; ++s) chi += SGR[s * 5];
      const float loss = 0.5f * chi + tl1;
      A.hist[(size_t)f * A.hist_stride + tglob] = loss;
      if (A.out_loss) A.out_loss[f] = loss;
      if (A.out_chi2) A.out_chi2[f] = chi;
    }
    if (A.out_ggrid) {
#pragma unroll
      for (int p = 0; p < PX; ++p) A.out_ggrid[gpix + p] = gB[p] + z[p];
    }
    if (A.out_gstars && tid < S * 4) {
      const int s = tid / 4, q = tid % 4;
      A.out_gstars[(size_t)f * S * 4 + tid] = SGR[s * 5 + 1 + q];
    }
    // ---- AdaBelief update ---------------------------------------------------------------------
    if (A.mode == 1) {
      // no contraction here: which product of b1 * m + (1 - b1) * g gets fused is the compiler's choice per
      // instantiation, and the one- and two-workgroup forms of this kernel must produce the same bits
#pragma clang fp contract(off)
      const float lr = SCAL[0], bc1 = SCAL[1], bc2 = SCAL[2];
      const float b1 = A.ab.b1, b2 = A.ab.b2, eps = A.ab.eps, eps_root = A.ab.eps_root;
      if (STATE_REGS) {
#pragma unroll
        for (int p = 0; p < PX; ++p) {
          const float g = gB[p] + z[p];
          const float mn = b1 * Mp[p] + (1.f - b1) * g;
          const float dg = g - mn;
          const float sn = b2 * Sp_[p] + (1.f - b2) * dg * dg + eps_root;
          Mp[p] = mn;
          Sp_[p] = sn;
          // hardware square root and reciprocal (1 ulp each): the IEEE sequences cost ~40 instructions per pixel
          Bp[p] -= lr * (mn * bc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(sn * bc2) + eps);
        }
      }
      // (state in global memory, PX > 8: every load of the thread's pixels before the first store - the three arrays may
      //  alias as far as the compiler knows, so a loop of load / step / store per 16 bytes ran as PX / 4 dependent memory
      //  round trips: C3 shard 85.2 -> 82.4 us per iteration.  What is left of the update's 27 k cycles is the traffic itself: the
      //  126 workgroups of a shard move 48 MB of state at the same moment.  Measured and left out: the next iteration's
      //  T = Moffat + B written to LDS from here instead of P1 re-reading B - 61 spilled registers instead of 36, 90.6 us)
      if constexpr (!STATE_REGS) {
        float4 *bp = (float4 *)(Bg + gpix);
        float4 *mp = (float4 *)(mBg + gpix);
        float4 *sp = (float4 *)(sBg + gpix);
        float4 bq[PX / 4], mq[PX / 4], sq[PX / 4];
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
          bq[q] = bp[q];
          mq[q] = mp[q];
          sq[q] = sp[q];
        }
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
          float *bb = &bq[q].x, *mm = &mq[q].x, *ss_ = &sq[q].x;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = gB[4 * q + e] + z[4 * q + e];
            const float mn = b1 * mm[e] + (1.f - b1) * g;
            const float dg = g - mn;
            const float sn = b2 * ss_[e] + (1.f - b2) * dg * dg + eps_root;
            mm[e] = mn;
            ss_[e] = sn;
            bb[e] -= lr * (mn * bc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(sn * bc2) + eps);
          }
        }
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
          bp[q] = bq[q];
          mp[q] = mq[q];
          sp[q] = sq[q];
        }
      }
      if constexpr (!SPLIT) update_stars_and_taps(tid);  // (the two-workgroup form did this beside its hand-off)
    }
    __syncthreads();
    LC_STAMP(43);
  }  // iterations

  if (A.mode == 1 && STATE_REGS && conv_role) {
    const size_t g0pix = (size_t)f * N * N + (size_t)(tid0 / (N / PX)) * N + (tid0 % (N / PX)) * PX;
#pragma unroll
    for (int q = 0; q < PX / 4; ++q) {
      ((float4 *)(A.B + g0pix))[q] = make_float4(Bp[4 * q], Bp[4 * q + 1], Bp[4 * q + 2], Bp[4 * q + 3]);
      ((float4 *)(A.mB + g0pix))[q] = make_float4(Mp[4 * q], Mp[4 * q + 1], Mp[4 * q + 2], Mp[4 * q + 3]);
      ((float4 *)(A.sB + g0pix))[q] = make_float4(Sp_[4 * q], Sp_[4 * q + 1], Sp_[4 * q + 2], Sp_[4 * q + 3]);
    }
  }
  if (A.mode == 1 && tid0 < S * 4 && conv_role) {
    A.stars[(size_t)f * S * 4 + tid0] = SP[tid0];
    A.stars_m[(size_t)f * S * 4 + tid0] = SM[tid0];
    A.stars_s[(size_t)f * S * 4 + tid0] = SV[tid0];
  }
}

}  // namespace lc
