// Fourth form of the regulariser chain (default from round 4 on): the batched tiled products of joint_reg_mfma.h with the
// element-wise launches folded into them and the point-source term taken out of the products altogether - FOUR launches per
// iteration instead of eight, seven products per launch instead of eight.
//
// The eight launches of the second form (Pbar, T = X AT, c = A T, S planes, T' = S A, Z = AT T', sums, values + flag) take
// 5 - 7.5 us each for 1 - 2 us of work and ~60 us per iteration at N = 128 (profiles/r04_cluster_e25_summary.txt): once the
// epoch kernel of a shard runs as a cluster launch (42 us) the chain IS the iteration, and one launch with in-kernel syncs
// was no shorter (59.4 us) - what shortens it is fewer stages.  The element-wise stages need no stage of their own:
//   f1   T = X AT_s                                        (plain product; the same launch carries the point-source blocks)
//   f2   c_s = A_s T                                       (plain product)
//   a1'  T' = S_s A_s with S_s = q_s - q_{s-1} formed in the operand fetch from c_{s-1}, c_s, c_{s+1} and the weights; the
//        tile that owns a slice's k range adds the values of the terms (l1 per scale, positivity) and writes the plane
//        that is a result itself: S_0 = q_0 - positivity sub-gradient
//   a2'  Z_s = AT_s T' stored write-through; every workgroup then adds one to the completion counter
// and the consumer adds up: the fused reduction + update of the device loop reads S_0 + sum_s Z_s per pixel (same order as
// mreg_finish2_kernel: the same bits) and its scalar block adds the per-tile values (planes_regs_block).  Other consumers
// (split update, gradient evaluations, the sharded drive) get greg / regs from ONE more launch, mreg_finish3_kernel.
//
// The point-source starlet term (scale 0 of Pbar = sum_i abar_i G_i only) was a ninth product in each of the four batches plus
// a launch for Pbar.  It needs no product: G_i = g^y_i (x) g^x_i is separable, so with R = A_1 (the 5-tap B3 smoothing,
// edge-replicating)
//     c_1(Pbar) = R Pbar R^T = sum_i abar_i (R g^y_i) (x) (R g^x_i)                         (1-D tables, an outer product per pixel)
//     <S - R^T S R, F> = <S, F - R F R^T>    for F = G_i, dG_i/dx, dG_i/dy (all separable)  (the adjoint moved onto the Gaussians)
// - one pass over the pixels with four 1-D tables per source (g, R g, g t / sigma^2, R (g t / sigma^2)), no dependence on h or
// on any product: pts_sep_block, 64 blocks that ride in the launch of f1 (or stand alone behind the all-reduce of a sharded
// fit).  Equal to the product form up to fp32 rounding of other summation orders (tests/test_joint_paths_gpu.py).
#pragma once
#include "joint_reg_mfma.h"

namespace lc {

struct MmxArgs {
  MmBatch mm;
  int scale[12];     // product b of the batch: scale s = 1 .. J of h
  int J, ntile;
  PtsSepArgs pts;    // f1: the blocks with blockIdx.z >= mm.nb evaluate the point-source term (none launched when it is off)
  // a1': S on the fly
  const float *X, *C, *W, *norms;   // h; [J + 2][NN] smoothed planes; [J][NN] weights or null; [J]
  float lam_sc, lam_hf, lam_pos;
  float *S;                         // [J + 2][NN]: plane 0 is written
  float *vals;                      // [J + 1][ntile]: row s < J: l1 of scale s; row J: positivity
  // a2': epilogue
  unsigned int *done;               // completion counter (every workgroup of the launch adds one) or null
};

// Registers: the fused update that waits for this chain's completion counter may have two of its blocks resident on every CU
// (N^2 / 32 <= 2 n_cu blocks, lc_joint_step_update); beside them a wave of every kernel of the chain must still fit, or a late
// chain could not be scheduled at all while the update waits for it: 2 R_update + R_chain <= 512 registers per lane and SIMD.
// The update takes 197 (allocated 200), so a kernel of the chain may take 112: a1', which reads five planes per element, is
// the largest with 92 + 16 accumulator registers (32-bit element offsets from scalar plane pointers instead of a 64-bit
// address per plane: 114 + 16 before) - check with -Rpass-analysis=kernel-resource-usage after touching either kernel.  What a
// chain that cannot be scheduled looks like: every update block waits out its bound and the run fails with "the regulariser of
// an iteration did not complete in time" - seen on the first fit of a process, where the chain's first launches are slow.
// MODE 0: f1 (+ the point-source blocks), 1: a1', 2: a2', 3: the plain product (f2).  Grid (N / 64, N / 64, nb [+ extra]) as
// mreg_mm_kernel, whose product loop this is.
template <int N, int MODE>
__global__ __launch_bounds__(kMmThreads) void mreg_mmx_kernel(MmxArgs Q) {
  constexpr int kAsF = 2 * 64 * (kMmKT + 1), kBsF = 2 * kMmKT * (64 + 4);
  constexpr int kTabF = (MODE == 0) ? 4 * kMaxSources * N : 0;
  __shared__ __align__(16) float smem[(kAsF + kBsF > kTabF) ? kAsF + kBsF : kTabF];
  __shared__ float red[kMmThreads / 64][kPtsStride];
  __shared__ float abar[kMaxSources];
  constexpr int NN = N * N;
  const MmBatch &G = Q.mm;
  const int bz = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, h = lane >> 5;
  if constexpr (MODE == 0) {
    if (bz >= G.nb) {  // the point-source term rides in this launch: it depends on nothing the chain computes
      const int pb = ((bz - G.nb) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
      if (pb < kPtsBlocks) pts_sep_block(Q.pts, pb, smem, red, abar);
      return;
    }
  }
  float (*As)[64][kMmKT + 1] = (float (*)[64][kMmKT + 1])smem;
  float (*Bs)[kMmKT][64 + 4] = (float (*)[kMmKT][64 + 4])(smem + kAsF);
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float *A = G.A[bz], *B = G.B[bz];
  const int wr = (wid >> 1) * 32, wc = (wid & 1) * 32;
  const int s = Q.scale[bz], J = Q.J;
  const int tile = blockIdx.y * (N / 64) + blockIdx.x;
  // a1': the planes S_s is formed from: c_{s-1}, c_s, c_{s+1}, weights of s - 1 and s
  const float *Pm = nullptr, *Pj = nullptr, *Pn = nullptr, *Wm = nullptr, *Wj = nullptr;
  float lam_m = 0.f, lam_j = 0.f, nrm_m = 0.f, nrm_j = 0.f;
  if constexpr (MODE == 1) {
    Pm = (s == 1) ? Q.X : Q.C + (size_t)(s - 1) * NN;
    Pj = Q.C + (size_t)s * NN;
    Pn = Q.C + (size_t)min(s + 1, J) * NN;
    lam_m = (s - 1 == 0) ? Q.lam_hf : Q.lam_sc;
    lam_j = (s < J) ? Q.lam_sc : 0.f;
    if (Q.W) {
      Wm = Q.W + (size_t)(s - 1) * NN;
      Wj = Q.W + (size_t)min(s, J - 1) * NN;
    } else {
      nrm_m = Q.norms[s - 1];
      nrm_j = Q.norms[min(s, J - 1)];
    }
  }
  float v_l1 = 0.f, v_l10 = 0.f, v_pos = 0.f;   // values of the terms over the elements this tile owns
  float4 pa[2], pb[2];
  float4 xm[2], xj[2], xn[2], wm[2], wj[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) xm[q] = xj[q] = xn[q] = wm[q] = wj[q] = pa[q] = pb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto sgn = [](float d, float lw) { return (d > 0.f) ? lw : ((d < 0.f) ? -lw : 0.f); };
  // (32-bit element offsets from workgroup-uniform plane pointers: scalar base + vector offset addressing, no 64-bit address
  //  per plane and lane - a1' reads five planes per element)
  unsigned int ia0[2], ib0[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + q * kMmThreads;
    ia0[q] = (unsigned int)((r0 + (e >> 3)) * N + (e & 7) * 4);       // 8 float4 per A row
    ib0[q] = (unsigned int)((e >> 4) * N + c0 + (e & 15) * 4);        // 16 float4 per B row
  }
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      pb[q] = *(const float4 *)(B + (ib0[q] + (unsigned int)(k0 * N)));
      const unsigned int ia = ia0[q] + (unsigned int)k0;
      if constexpr (MODE == 1) {
        xm[q] = *(const float4 *)(Pm + ia);
        xj[q] = *(const float4 *)(Pj + ia);
        xn[q] = *(const float4 *)(Pn + ia);
        if (Wm) {
          wm[q] = *(const float4 *)(Wm + ia);
          wj[q] = *(const float4 *)(Wj + ia);
        }
      } else {
        pa[q] = *(const float4 *)(A + ia);
      }
    }
  };
  // a1': what the fetched planes become: the A operand of the slice and, for the tile that owns the slice's k range, the values
  // of the terms and the plane that is a result itself
  auto form = [&](int k0) {
    if constexpr (MODE == 1) {
      const bool own = (k0 >= c0 && k0 < c0 + 64);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned int ia = ia0[q] + (unsigned int)k0;
        const float m4[4] = {xm[q].x, xm[q].y, xm[q].z, xm[q].w}, j4[4] = {xj[q].x, xj[q].y, xj[q].z, xj[q].w};
        const float n4[4] = {xn[q].x, xn[q].y, xn[q].z, xn[q].w};
        const float wm4[4] = {wm[q].x, wm[q].y, wm[q].z, wm[q].w}, wj4[4] = {wj[q].x, wj[q].y, wj[q].z, wj[q].w};
        float o[4], o0[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float lwm = Wm ? lam_m * wm4[c] : lam_m * nrm_m;
          const float dm = m4[c] - j4[c];
          const float qm = sgn(dm, lwm);
          float qj = 0.f;
          if (s < J) {
            const float lwj = Wm ? lam_j * wj4[c] : lam_j * nrm_j;
            const float dj = j4[c] - n4[c];
            qj = sgn(dj, lwj);
            if (own) v_l1 += lwj * fabsf(dj);
          }
          o[c] = qj - qm;
          o0[c] = 0.f;
          if (s == 1 && own) {   // scale 0 and the positivity term ride with scale 1: S_0 = q_0 - positivity sub-gradient
            v_l10 += lwm * fabsf(dm);
            float z = qm;
            const float hv = m4[c];
            if (Q.lam_pos != 0.f && hv < 0.f) {
              v_pos += -Q.lam_pos * hv;
              z -= Q.lam_pos;
            }
            o0[c] = z;
          }
        }
        pa[q] = make_float4(o[0], o[1], o[2], o[3]);
        if (own && s == 1) *(float4 *)(Q.S + ia) = make_float4(o0[0], o0[1], o0[2], o0[3]);
      }
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + q * kMmThreads;
      const int ar = e >> 3, ak = (e & 7) * 4;
      As[buf][ar][ak] = pa[q].x; As[buf][ar][ak + 1] = pa[q].y; As[buf][ar][ak + 2] = pa[q].z; As[buf][ar][ak + 3] = pa[q].w;
      const int bk = e >> 4, bc = (e & 15) * 4;
      *(float4 *)&Bs[buf][bk][bc] = pb[q];
    }
  };
  mr_acc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int kbeg = 0, kend = N;
  if (G.band[bz]) {
    const int lo = (G.band[bz] == 1) ? c0 : r0;
    kbeg = max(lo - G.hw[bz], 0) / kMmKT * kMmKT;
    kend = min((lo + 64 + G.hw[bz] + kMmKT - 1) / kMmKT * kMmKT, N);
  }
  if constexpr (MODE == 1) {  // the owner's slices are inside every band (the k range follows the tile's columns: band 1)
    kbeg = min(kbeg, c0);
    kend = max(kend, c0 + 64);
  }
  fetch(kbeg);
  form(kbeg);
  put(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kMmKT, buf ^= 1) {
    if (k0 + kMmKT < kend) fetch(k0 + kMmKT);
#pragma unroll
    for (int s2 = 0; s2 < kMmKT / 2; ++s2) {
      const int k = 2 * s2 + h;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][wr + i][k], Bs[buf][k][wc + i], acc, 0, 0, 0);
    }
    if (k0 + kMmKT < kend) {
      form(k0 + kMmKT);
      put(buf ^ 1);
    }
    __syncthreads();
  }
  float *C = G.C[bz];
  if constexpr (MODE != 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) C[(size_t)(r0 + wr + mr_row(r, h)) * N + c0 + wc + i] = acc[r];
  }
  if constexpr (MODE == 1) {
    // values of this tile's elements: lanes, then the four waves in order
    const float t0 = wave_sum_shfl(v_l1), t1 = wave_sum_shfl(v_l10), t2 = wave_sum_shfl(v_pos);
    if (lane == 0) {
      red[wid][0] = t0;
      red[wid][1] = t1;
      red[wid][2] = t2;
    }
    __syncthreads();
    if (tid < 3) {
      float t = 0.f;
      for (int w = 0; w < kMmThreads / 64; ++w) t += red[w][tid];
      if (tid == 0) {
        if (s < J) Q.vals[(size_t)s * Q.ntile + tile] = t;
      } else if (s == 1) {
        Q.vals[(size_t)(tid == 1 ? 0 : J) * Q.ntile + tile] = t;
      }
    }
  }
  if constexpr (MODE == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xwg_storef<true>(&C[(size_t)(r0 + wr + mr_row(r, h)) * N + c0 + wc + i], acc[r]);
    if (Q.done) {  // every store of this workgroup has left before its count does (cluster_sync's order)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(Q.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Head of the chain's stream inside the library's loops: one wave polls the word in which the first thread of the epoch launch
// announces that the update before it is complete (JointArgs::upd_signal), and ends; the chain's first launch follows it in
// stream order.  Replaces the cross-stream wait for an event that the main stream had to record between the update and the
// next epoch kernel.  Bounded (~1 s): a wait that runs out is reported.
__global__ void mreg_gate_kernel(const unsigned int *word, unsigned int target, unsigned int *err) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(16);
      if (++spins > (1 << 21)) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
}
// Can a kernel on the chain's stream run WHILE a later-enqueued kernel of the main stream starts?  Normally yes - two HIP
// streams, two hardware queues - but the runtime has only a few hardware queues (four by default) and maps further streams
// onto the same ones: a process that holds several objects can find the two streams of one of them behind each other in ONE
// queue, and there a gate kernel would wait for an epoch launch that cannot start before the gate has ended.  Probed once per
// object: this kernel waits (bounded: `ticks` of the 100 MHz counter) for a word that a kernel enqueued AFTERWARDS on the
// main stream sets, and reports whether it saw it.
__global__ void mreg_probe_kernel(const unsigned int *word, unsigned int *seen, long long ticks) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    unsigned int ok = 0;
    while (wall_clock64() - t0 < ticks) {
      if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        ok = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
    __hip_atomic_store(seen, ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// (where no epoch kernel follows - the point-source-only kernel - the signal is a launch of its own)
__global__ void mreg_signal_kernel(unsigned int *word, unsigned int value) {
  if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// For the consumers that take greg / regs: greg = S_0 + sum_s Z_s (blocks < nb), regs from the per-tile values (block nb).
// done: completion counter of the chain (every block adds one; the stores before it write-through) when the consumer checks
// it in its kernel - the fused update beside a one-workgroup epoch kernel, where the chain has time to spare and the update
// has none - or null behind an event.
__global__ __launch_bounds__(kGmThreads) void mreg_finish3_kernel(int NN, int nb, int M, RegPlanes P, float *greg, float *regs,
                                                                  unsigned int *done) {
  __shared__ float regl[4 + 3 * kMaxSources];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nb) {
    const int k = blockIdx.x * blockDim.x + tid;
    if (k < NN) {
      const float g = planes_greg(P, k, NN, false);
      if (done) xwg_storef<true>(&greg[k], g);
      else greg[k] = g;
    }
  } else {
    planes_regs_block(P, M, regl, false);
    __syncthreads();
    if (tid < 4 + 3 * M && tid != 3 && (tid < 2 || P.npts > 0)) {
      if (done) xwg_storef<true>(&regs[tid], regl[tid]);
      else regs[tid] = regl[tid];
    }
  }
  if (done) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace lc
