// Fourth form of the regulariser chain (default from round 4 on): the batched tiled products of joint_reg_mfma.h with the
// element-wise launches folded into them - FOUR launches per iteration instead of eight.
//
// The eight launches of the second form (Pbar, T = X AT, c = A T, S planes, T' = S A, Z = AT T', sums, values + flag) take
// 5 - 7.5 us each for 1 - 2 us of work and ~60 us per iteration at N = 128 (profiles/r04_cluster_e25_summary.txt): once the
// epoch kernel of a shard runs as a cluster launch (42 us) the chain IS the iteration, and one launch with in-kernel syncs
// was no shorter (59.4 us) - what shortens it is fewer stages.  The element-wise stages need no stage of their own:
//   f1'  T = X AT_s, and for the point-source channel T = Pbar AT_1 with Pbar EVALUATED in the operand fetch (mean fluxes
//        summed by the tile itself; the tile whose columns are the k range of a slice also writes those elements of Pbar out)
//   f2   c_s = A_s T                                       (the plain product: MODE 3)
//   a1'  T' = S_s A_s with S_s = q_s - q_{s-1} formed in the operand fetch from c_{s-1}, c_s, c_{s+1} and the weights; the
//        tile that owns a slice's k range adds the values of the terms (l1 per scale, positivity) and writes the two planes
//        that are results themselves: S_0 = q_0 - positivity sub-gradient and the point-source S
//   a2'  Z_s = AT_s T' stored write-through; the tiles of the point-source channel form z = S - Z and its inner products with
//        the sources' Gaussians in the epilogue (per-tile partials); every workgroup then adds one to the completion counter
// and the consumer adds up: the fused reduction + update of the device loop reads S_0 + sum_s Z_s per pixel (same order as
// mreg_finish2_kernel: the same bits) and its scalar block adds the per-tile values (planes_reg_value).  Other consumers
// (split update, gradient evaluations, the sharded drive) get greg / regs from ONE more launch, mreg_finish3_kernel.
// Same arithmetic as the second form element by element; the values and the inner products are added per 64 x 64 tile instead
// of per 256-pixel block, so they agree to fp32 rounding, not bit for bit (tests/test_joint_paths_gpu.py).
#pragma once
#include "joint_reg_mfma.h"

namespace lc {

struct MmxArgs {
  MmBatch mm;
  int scale[12];     // product b of the batch: scale s = 1 .. J of h, or -1: the point-source channel (scale 1 of Pbar)
  int J, ntile;
  // f1': Pbar
  int E, M, ss;
  const float *a, *cx, *cy;
  float *pbar;
  // a1': S on the fly
  const float *X, *C, *W, *norms;   // h; [J + 2][NN] smoothed planes (slot J + 1: c_1(Pbar)); [J][NN] weights or null; [J]
  float lam_sc, lam_hf, lam_pts, lam_pos;
  float *S;                         // [J + 2][NN]: planes 0 and J + 1 are written
  float *vals;                      // [J + 2][ntile]: row s < J: l1 of scale s; row J: positivity; row J + 1: point-source term
  // a2': epilogue
  float *pts_part;                  // [ntile][3 kMaxSources]
  unsigned int *done;               // completion counter (every workgroup of the launch adds one) or null
};

// MODE 0: f1', 1: a1', 2: a2', 3: the plain product (f2).  Grid (N / 64, N / 64, nb) as mreg_mm_kernel, whose product loop this is.
// Registers: the fused update that waits for this chain's completion counter may have two of its blocks resident on every CU
// (N^2 / 32 <= 2 n_cu blocks, lc_joint_step_update); beside them a wave of every kernel of the chain must still fit, or a late
// chain could not be scheduled at all while the update waits for it: 2 R_update + R_chain <= 512 registers per lane and SIMD.
// The update takes 197 (allocated 200), so a kernel of the chain may take 112: a1', which reads five planes per element, is
// the largest with 92 + 16 accumulator registers (32-bit element offsets from scalar plane pointers instead of a 64-bit
// address per plane: 114 + 16 before) - check with -Rpass-analysis=kernel-resource-usage after touching either kernel.  What a
// chain that cannot be scheduled looks like: every update block waits out its bound and the run fails with "the regulariser of
// an iteration did not complete in time" - seen on the first fit of a process, where the chain's first launches are slow.
template <int N, int MODE>
__global__ __launch_bounds__(kMmThreads) void mreg_mmx_kernel(MmxArgs Q) {
  __shared__ float As[2][64][kMmKT + 1];
  __shared__ float Bs[2][kMmKT][64 + 4];
  __shared__ float red[kMmThreads / 64][3 * kMaxSources + 1];
  __shared__ float abar[kMaxSources];
  constexpr int NN = N * N;
  const MmBatch &G = Q.mm;
  const int bz = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float *A = G.A[bz], *B = G.B[bz];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, h = lane >> 5;
  const int wr = (wid >> 1) * 32, wc = (wid & 1) * 32;
  const int s = Q.scale[bz], J = Q.J;
  const bool pts = (s < 0);
  const int tile = blockIdx.y * (N / 64) + blockIdx.x;
  const float cen = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
  if constexpr (MODE == 0) {
    if (pts) {  // mean fluxes, as mreg_pbar_kernel sums them
      for (int q = wid; q < Q.M; q += kMmThreads / 64) {
        float acc = 0.f;
        for (int e = lane; e < Q.E; e += 64) acc += Q.a[e * Q.M + q];
        acc = wave_sum_shfl(acc);
        if (lane == 0) abar[q] = acc / (float)Q.E;
      }
      __syncthreads();
    }
  }
  // a1': the planes S_s is formed from (h-scale s: c_{s-1}, c_s, c_{s+1}, weights of s - 1 and s; point sources: Pbar, c_1(Pbar),
  // weights of scale 0)
  const float *Pm = nullptr, *Pj = nullptr, *Pn = nullptr, *Wm = nullptr, *Wj = nullptr;
  float lam_m = 0.f, lam_j = 0.f, nrm_m = 0.f, nrm_j = 0.f;
  if constexpr (MODE == 1) {
    if (pts) {
      Pm = Q.pbar;                              // Pbar (f1' wrote it; the batch's A pointer of this product is the S plane)
      Pj = Q.C + (size_t)(J + 1) * NN;          // c_1(Pbar)
      Pn = Pj;
      lam_m = Q.lam_pts;
      if (Q.W) Wm = Q.W; else nrm_m = Q.norms[0];
      Wj = Wm;
    } else {
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
      } else if constexpr (MODE == 0) {
        if (!pts) pa[q] = *(const float4 *)(A + ia);
      } else {
        pa[q] = *(const float4 *)(A + ia);
      }
    }
  };
  // what the fetched planes become: the A operand of the slice (and, for the tile that owns the slice's k range, the values and
  // the planes that are results themselves)
  auto form = [&](int k0) {
    const bool own = (k0 >= c0 && k0 < c0 + 64);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + q * kMmThreads;
      const int ar = e >> 3, ak = (e & 7) * 4;
      const unsigned int ia = ia0[q] + (unsigned int)k0;
      if constexpr (MODE == 0) {
        if (pts) {
          const int u = r0 + ar;
          float o[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int v = k0 + ak + c;
            float acc = 0.f;
            for (int m = 0; m < Q.M; ++m) {
              const float tx = (float)v - (cen + Q.ss * Q.cx[m]), ty = (float)u - (cen + Q.ss * Q.cy[m]);
              acc = fmaf(abar[m] * nrm2, expf(-0.5f * (tx * tx + ty * ty) * inv_s2), acc);
            }
            o[c] = acc;
          }
          pa[q] = make_float4(o[0], o[1], o[2], o[3]);
          if (own) *(float4 *)(Q.pbar + ia) = pa[q];
        }
      }
      if constexpr (MODE == 1) {
        const float m4[4] = {xm[q].x, xm[q].y, xm[q].z, xm[q].w}, j4[4] = {xj[q].x, xj[q].y, xj[q].z, xj[q].w};
        const float n4[4] = {xn[q].x, xn[q].y, xn[q].z, xn[q].w};
        const float wm4[4] = {wm[q].x, wm[q].y, wm[q].z, wm[q].w}, wj4[4] = {wj[q].x, wj[q].y, wj[q].z, wj[q].w};
        float o[4], o0[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float lwm = Wm ? lam_m * wm4[c] : lam_m * nrm_m;
          const float dm = m4[c] - j4[c];
          const float qm = sgn(dm, lwm);
          if (pts) {
            o[c] = qm;
            if (own) v_l1 += lwm * fabsf(dm);
          } else {
            float qj = 0.f;
            if (s < J) {
              const float lwj = Wm ? lam_j * wj4[c] : lam_j * nrm_j;
              const float dj = j4[c] - n4[c];
              qj = sgn(dj, lwj);
              if (own) v_l1 += lwj * fabsf(dj);
            }
            o[c] = qj - qm;
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
        }
        pa[q] = make_float4(o[0], o[1], o[2], o[3]);
        if (own) {
          if (pts) *(float4 *)(Q.S + (size_t)(J + 1) * NN + ia) = pa[q];
          else if (s == 1) *(float4 *)(Q.S + ia) = make_float4(o0[0], o0[1], o0[2], o0[3]);
        }
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
  if constexpr (MODE == 0) {
    if (pts) {
      kbeg = min(kbeg, c0);
      kend = max(kend, c0 + 64);
    }
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
        if (pts) Q.vals[(size_t)(J + 1) * Q.ntile + tile] = t;
        else if (s < J) Q.vals[(size_t)s * Q.ntile + tile] = t;
      } else if (s == 1) {
        Q.vals[(size_t)(tid == 1 ? 0 : J) * Q.ntile + tile] = t;
      }
    }
  }
  if constexpr (MODE == 2) {
    if (!pts) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xwg_storef<true>(&C[(size_t)(r0 + wr + mr_row(r, h)) * N + c0 + wc + i], acc[r]);
    } else {
      // z = S - Z on the tile and its inner products with the sources' Gaussians and their position derivatives
      // (mreg_finish2_kernel's terms, added per tile)
      const float *Sp = Q.S + (size_t)(J + 1) * NN;
      float z[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = Sp[(size_t)(r0 + wr + mr_row(r, h)) * N + c0 + wc + i];
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] -= acc[r];
      const int v = c0 + wc + i;
      for (int m = 0; m < Q.M; ++m) {
        const float tx = (float)v - (cen + Q.ss * Q.cx[m]), yc = cen + Q.ss * Q.cy[m];
        float sa = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ty = (float)(r0 + wr + mr_row(r, h)) - yc;
          const float gq = z[r] * nrm2 * expf(-0.5f * (tx * tx + ty * ty) * inv_s2);
          sa += gq;
          sx += gq * tx * inv_s2;
          sy += gq * ty * inv_s2;
        }
        sa = wave_sum_shfl(sa);
        sx = wave_sum_shfl(sx);
        sy = wave_sum_shfl(sy);
        if (lane == 0) {
          red[wid][m * 3] = sa;
          red[wid][m * 3 + 1] = sx;
          red[wid][m * 3 + 2] = sy;
        }
      }
      __syncthreads();
      if (tid < 3 * Q.M) {
        float t = 0.f;
        for (int w = 0; w < kMmThreads / 64; ++w) t += red[w][tid];
        xwg_storef<true>(&Q.pts_part[(size_t)tile * 3 * kMaxSources + tid], t);
      }
    }
    if (Q.done) {  // every store of this workgroup has left before its count does (cluster_sync's order)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(Q.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// For the consumers that take greg / regs: greg = S_0 + sum_s Z_s (blocks < nb), regs from the per-tile values (block nb).
__global__ __launch_bounds__(kGmThreads) void mreg_finish3_kernel(int NN, int nb, int M, RegPlanes P, float *greg, float *regs) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nb) {
    const int k = blockIdx.x * blockDim.x + tid;
    if (k < NN) greg[k] = planes_greg(P, k, NN, false);
    return;
  }
  // (without the point-source channel its slots are left alone: the sharded drive fills them behind the all-reduce)
  if (tid < 2 || (P.has_pts && tid < 4 + 3 * M && tid != 3)) regs[tid] = planes_reg_value(P, tid, false);
}

}  // namespace lc
