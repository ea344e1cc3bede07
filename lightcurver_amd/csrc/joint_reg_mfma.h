// Starlet l1 regulariser of the background grid (and the point-source starlet term) as dense two-sided products on the
// fp32 matrix cores, for the grids whose cascade was the longest single-workgroup job of an iteration (N >= 128).
//
// The a-trous cascade c_{j+1} = Col_j Row_j c_j (STARRED's starlet as used by Loss: reference call sites
// lightcurver/processes/roi_modelling.py:313-322) is sequential in j and global in space, so one workgroup ran 2 J
// barrier-separated 5-tap sweeps (94 us at N = 128, longer than the epoch kernel it is meant to hide behind).  The same
// numbers follow from the cumulative 1-D operators A_j = R_{j-1} ... R_0 (R_s: edge-replicating B3 filter at dilation
// 2^s, an N x N matrix; A_0 = I):
//     c_j = A_j X A_j^T,   w_j = c_j - c_{j+1},   l1 = sum_j lam_j sum W_j |w_j|,   q_j = lam_j W_j sign(w_j)
//     d l1 / d X = sum_j ( A_j^T q_j A_j - A_{j+1}^T q_j A_{j+1} )
// Every scale is independent of the others, and so is every block of 32 columns of a product: a workgroup (column block b,
// scale j) computes  M ( S M[blk]^T )  with v_mfma_f32_32x32x2_f32 (exact fp32 multiply-adds), wave w owning output rows
// 32 w .. 32 w + 31.  Three launches replace the cascade: forward (c_j for all j), adjoint (q_j on the fly, the two
// products per scale), and a small kernel that sums the J partial sub-gradients, adds positivity and the inner products
// of the point-source term.  Fixed summation orders throughout: results do not depend on scheduling.
#pragma once
#include "joint_gm.h"

namespace lc {

typedef float mr_acc __attribute__((ext_vector_type(16)));

struct MregArgs {
  int J, has_pts;
  int s0;                // first slot of the launch (0, or J when only the point-source term is on)
  const float *A, *AT;  // [J + 1][N][N] cumulative smoothing operators and their transposes (row-major)
  const float *X;       // h
  const float *P;       // mean point-source channel Pbar (has_pts)
  float *C;             // [J + 2][N][N]: slot j = c_j(h) for j = 1 .. J; slot J + 1 = c_1(Pbar)
  const float *W;       // [J][N][N] or null (then norms[j])
  const float *norms;   // [J]
  float lam_sc, lam_hf, lam_pts;
  float *Z;             // [J + 1][N][N] partial sub-gradients per scale; slot J: d term / d Pbar
  float *l1p;           // [J + 1] value of the term per scale; slot J: the point-source term
};

// row of output element `reg` of a 32 x 32 accumulator tile held by lane half h (the column is lane & 31)
__device__ __forceinline__ int mr_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// acc[p] (rows 32 wid .., columns b0 ..) = M[p] ( S M[p][b0 .. b0 + 31, :]^T ),  p < NP products sharing S.
// srow(q) returns elements [32 wid + (lane & 31)][h HK + 4 q .. + 3] of S, h = lane >> 5, HK = N / 2: lane half h feeds
// the k range [h HK, (h + 1) HK) of every product (one k per half and MFMA step), so that all operands are contiguous
// per lane.  The intermediate S M^T passes through LDS once ([HK][64] image per product: step s reads 64 consecutive
// floats).  All 64 * N / 32 threads must call.
// The operands taken from M are read from its TRANSPOSE MT (the operators are kept in both forms): an MFMA operand wants the
// matrix row in the lane index and k in the step, so from M itself every lane of a load touched a line of its own (rows are
// N floats apart); from MT the 32 lanes of a half read 128 consecutive bytes.  Same values, same order of the sums.
#ifndef MR_UNROLL
#define MR_UNROLL 16
#endif
template <int N, int NP, class SRow>
__device__ __forceinline__ void mr_two_sided(SRow &&srow, const float *const (&MT)[NP], int b0, float *ylds, int lane, int wid,
                                             mr_acc (&acc)[NP]) {
  constexpr int HK = N / 2;
  const int i = lane & 31, h = lane >> 5;
  mr_acc y[NP];
  const float *m1[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int r = 0; r < 16; ++r) y[p][r] = 0.f;
    m1[p] = MT[p] + (size_t)(h * HK) * N + b0 + i;  // M[b0 + i][h HK + k] = MT[h HK + k][b0 + i]
  }
#pragma unroll MR_UNROLL
  for (int q = 0; q < HK / 4; ++q) {
    const float4 a4 = srow(q);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float *mq = m1[p] + (size_t)(4 * q) * N;
      const float4 b4 = make_float4(mq[0], mq[N], mq[2 * N], mq[3 * N]);
      y[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, y[p], 0, 0, 0);
      y[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, y[p], 0, 0, 0);
      y[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, y[p], 0, 0, 0);
      y[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, y[p], 0, 0, 0);
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = 32 * wid + mr_row(r, h);
      ylds[p * HK * 64 + (k % HK) * 64 + i + 32 * (k / HK)] = y[p][r];
    }
  __syncthreads();
  const float *m2[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    m2[p] = MT[p] + (size_t)(h * HK) * N + 32 * wid + i;
  }
#pragma unroll MR_UNROLL
  for (int q = 0; q < HK / 4; ++q) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float *mq = m2[p] + (size_t)(4 * q) * N;
      const float4 a4 = make_float4(mq[0], mq[N], mq[2 * N], mq[3 * N]);
      const float *yp = ylds + p * HK * 64 + (4 * q) * 64 + lane;
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, yp[0], acc[p], 0, 0, 0);
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, yp[64], acc[p], 0, 0, 0);
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, yp[128], acc[p], 0, 0, 0);
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, yp[192], acc[p], 0, 0, 0);
    }
  }
  __syncthreads();
}

template <int N>
struct MregCfg {
  static constexpr int NT = N / 32, NTHR = 64 * NT, HK = N / 2;
  static constexpr int LDS_FWD = HK * 64 * 4, LDS_ADJ = 2 * HK * 64 * 4;
  static_assert(N % 64 == 0 && LDS_ADJ <= 163840 - 1024, "grid size");
};

// forward: grid (N / 32, J + has_pts); block (b, s): c_{s+1}(h)[:, blk] -> C[s + 1], or (s == J) c_1(Pbar)[:, blk] -> C[J + 1]
template <int N>
__global__ __launch_bounds__(MregCfg<N>::NTHR) void mreg_forward_kernel(MregArgs A) {
  extern __shared__ __align__(16) float mr_lds[];
  constexpr int HK = N / 2;
  const int b0 = blockIdx.x * 32, s = blockIdx.y + A.s0, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const bool pts = (s == A.J);
  const int j = pts ? 1 : s + 1;
  const float *S = pts ? A.P : A.X;
  float *dst = A.C + (size_t)(pts ? A.J + 1 : j) * N * N;
  const float *const M[1] = {A.AT + (size_t)j * N * N};  // the product uses A_j; mr_two_sided reads it from its transpose
  const float4 *sp = (const float4 *)(S + (size_t)(32 * wid + i) * N + h * HK);
  mr_acc acc[1];
  mr_two_sided<N, 1>([&](int q) { return sp[q]; }, M, b0, mr_lds, lane, wid, acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[(size_t)(32 * wid + mr_row(r, h)) * N + b0 + i] = acc[0][r];
}

// adjoint: grid (N / 32, J + has_pts); block (b, s): Z[s][:, blk] = A_s^T q_s A_s[:, blk] - A_{s+1}^T q_s A_{s+1}[:, blk]
// (s == J: the point-source term, q from Pbar - c_1(Pbar) with the scale-0 weights);  l1p[s] from the blocks b == 0
template <int N>
__global__ __launch_bounds__(MregCfg<N>::NTHR) void mreg_adjoint_kernel(MregArgs A) {
  extern __shared__ __align__(16) float mr_lds[];
  __shared__ float red[MregCfg<N>::NT];
  constexpr int HK = N / 2, NN = N * N;
  const int b0 = blockIdx.x * 32, s = blockIdx.y + A.s0, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const bool pts = (s == A.J);
  const int j = pts ? 0 : s;
  const float *Cj = pts ? A.P : (j == 0 ? A.X : A.C + (size_t)j * NN);
  const float *Cj1 = A.C + (size_t)(pts ? A.J + 1 : j + 1) * NN;
  const float lam = pts ? A.lam_pts : (j == 0 ? A.lam_hf : A.lam_sc);
  const float *Wj = A.W ? A.W + (size_t)j * NN : nullptr;
  const float lwc = Wj ? 0.f : lam * A.norms[j];
  float l1 = 0.f;
  auto qval = [&](float c, float cn, float wgt) {
    const float w = c - cn, lw = Wj ? lam * wgt : lwc;
    l1 = fmaf(lw, fabsf(w), l1);
    return (w > 0.f) ? lw : ((w < 0.f) ? -lw : 0.f);
  };
  const size_t roff = (size_t)(32 * wid + i) * N + h * HK;
  const float4 *c4 = (const float4 *)(Cj + roff), *n4 = (const float4 *)(Cj1 + roff);
  const float4 *w4 = Wj ? (const float4 *)(Wj + roff) : nullptr;
  auto qrow = [&](int q) {
    const float4 c = c4[q], n = n4[q];
    const float4 w = Wj ? w4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    return make_float4(qval(c.x, n.x, w.x), qval(c.y, n.y, w.y), qval(c.z, n.z, w.z), qval(c.w, n.w, w.w));
  };
  float *Zs = A.Z + (size_t)s * NN;
  if (j == 0) {  // A_0 = I: the first product is q itself
    const float *const M[1] = {A.A + (size_t)NN};  // (products with A_1^T, read from A_1)
    mr_acc acc[1];
    mr_two_sided<N, 1>(qrow, M, b0, mr_lds, lane, wid, acc);
    const float l1_rows = l1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const size_t k = (size_t)(32 * wid + mr_row(r, h)) * N + b0 + i;
      Zs[k] = qval(Cj[k], Cj1[k], Wj ? Wj[k] : 0.f) - acc[0][r];
    }
    l1 = l1_rows;  // the second pass over this block's own pixels does not count twice
  } else {
    const float *const M[2] = {A.A + (size_t)j * NN, A.A + (size_t)(j + 1) * NN};
    mr_acc acc[2];
    mr_two_sided<N, 2>(qrow, M, b0, mr_lds, lane, wid, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) Zs[(size_t)(32 * wid + mr_row(r, h)) * N + b0 + i] = acc[0][r] - acc[1][r];
  }
  // the waves of a block cover every pixel of the scale once: value of the term, waves combined in order
  l1 = wave_sum_shfl(l1);
  if (lane == 0) red[wid] = l1;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < MregCfg<N>::NT; ++w) t += red[w];
    A.l1p[s] = t;
  }
}

// Pbar = sum_i abar_i G(c_i) on the grid of h (abar_i = mean over the epochs of a[e][i], summed in a fixed order by every block)
__global__ __launch_bounds__(kGmThreads) void mreg_pbar_kernel(int N, int ss, int E, int M, const float *a, const float *cx,
                                                               const float *cy, float *pbar) {
  __shared__ float abar[kMaxSources];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = wid; i < M; i += kGmThreads / 64) {
    float acc = 0.f;
    for (int e = lane; e < E; e += 64) acc += a[e * M + i];
    acc = wave_sum_shfl(acc);
    if (lane == 0) abar[i] = acc / (float)E;
  }
  __syncthreads();
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N * N) return;
  const float c0 = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
  const int u = k / N, v = k % N;
  float acc = 0.f;
  for (int i = 0; i < M; ++i) {
    const float tx = (float)v - (c0 + ss * cx[i]), ty = (float)u - (c0 + ss * cy[i]);
    acc = fmaf(abar[i] * nrm2, expf(-0.5f * (tx * tx + ty * ty) * inv_s2), acc);
  }
  pbar[k] = acc;
}

// greg = sum_j Z[j] + positivity sub-gradient; per-block positivity partials; per-block inner products of Z[J] with
// G_i and its position derivatives (gm_pts_inner_kernel's contract)
__global__ __launch_bounds__(kGmThreads) void mreg_finish_kernel(int N, int J, int l1_on, int has_pts, int ss, int M,
                                                                 const float *Z, const float *h, float lam_pos,
                                                                 const float *cx, const float *cy, float *greg,
                                                                 float *pos_part, float *pts_part) {
  __shared__ float red[kGmThreads / 64][kMaxSources * 3 + 1];
  const int NN = N * N, k = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool in = k < NN;
  float g = 0.f, pos = 0.f;
  if (in) {
    if (l1_on)
      for (int s = 0; s < J; ++s) g += Z[(size_t)s * NN + k];
    const float hv = h[k];
    if (lam_pos != 0.f && hv < 0.f) {
      pos = -lam_pos * hv;
      g -= lam_pos;
    }
    greg[k] = g;
  }
  pos = wave_sum_shfl(pos);
  if (lane == 0) red[wid][kMaxSources * 3] = pos;
  if (has_pts) {
    const float c0 = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
    const float z = in ? Z[(size_t)J * NN + k] : 0.f;
    const int u = in ? k / N : 0, v = in ? k % N : 0;
    for (int i = 0; i < M; ++i) {
      const float tx = (float)v - (c0 + ss * cx[i]), ty = (float)u - (c0 + ss * cy[i]);
      const float gq = z * nrm2 * expf(-0.5f * (tx * tx + ty * ty) * inv_s2);
      const float sa = wave_sum_shfl(gq), sx = wave_sum_shfl(gq * tx * inv_s2), sy = wave_sum_shfl(gq * ty * inv_s2);
      if (lane == 0) {
        red[wid][i * 3] = sa;
        red[wid][i * 3 + 1] = sx;
        red[wid][i * 3 + 2] = sy;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kGmThreads / 64; ++w) t += red[w][kMaxSources * 3];
    pos_part[blockIdx.x] = t;
  }
  if (has_pts && (int)threadIdx.x < 3 * M) {
    float acc = 0.f;
    for (int w = 0; w < kGmThreads / 64; ++w) acc += red[w][threadIdx.x];
    pts_part[(size_t)blockIdx.x * 3 * kMaxSources + threadIdx.x] = acc;
  }
}

// regs[0] = l1, regs[1] = positivity, regs[2] = point-source term, regs[4 + 3 i + q] = its inner products (one wave)
// done_flag (optional): receives done_seq once everything above is written — the fused update kernel of the device loop
// checks it instead of the host enqueueing a cross-stream event wait in front of it (joint_reduce_update_kernel)
__global__ void mreg_regs_kernel(int J, int l1_on, int has_pts, int nblocks, int M, const float *l1p, const float *pos_part,
                                 const float *pts_part, float *regs, unsigned int *done_flag, unsigned int done_seq) {
  const int lane = threadIdx.x;
  float b = 0.f;
  for (int i = lane; i < nblocks; i += 64) b += pos_part[i];
  b = wave_sum_shfl(b);
  if (lane == 0) {
    float a = 0.f;
    if (l1_on)
      for (int s = 0; s < J; ++s) a += l1p[s];
    regs[0] = a;
    regs[1] = b;
    if (has_pts) regs[2] = l1p[J];
  }
  if (has_pts)
    for (int t = 0; t < 3 * M; ++t) {
      float acc = 0.f;
      for (int blk = lane; blk < nblocks; blk += 64) acc += pts_part[(size_t)blk * 3 * kMaxSources + t];
      acc = wave_sum_shfl(acc);
      if (lane == 0) regs[4 + t] = acc;
    }
  if (done_flag) {  // (one wave; every write of this chain before the flag)
    __threadfence();
    if (lane == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- second form of the chain (default): batched tiled products, telescoped adjoint, one finishing launch -------------------
// The block-column kernels above keep 72 long-lived workgroups on the machine for ~130 us per iteration of a 256 x 256 grid,
// beside the epoch kernel's phases: every phase that shares CUs with them runs 40 - 60 % longer (rocprofv3 timelines,
// profiles/r03_c5_timeline_*.txt: C5 shard 264 us per iteration with the chain, 212 without).  The same numbers as plain
// batched products C = A B over the scales (row-major N x N operands, 64 x 64 output tiles, four waves of 32 x 32 MFMA
// accumulators, operands staged through LDS in K slices of 32 with 16-byte loads):
//     forward   T_j = X AT_j ,  c_j = A_j T_j                                (j = 1 .. J; the point-source channel with j = 1)
//     values    S_0 = q_0 - positivity ,  S_j = q_j - q_{j-1} ,  q_J = 0     (one element-wise launch, with the l1 / positivity
//                                                                             values per block)
//     adjoint   T'_j = S_j A_j ,  Z_j = AT_j T'_j                             (telescoped: d l1 / d X = S_0 + sum_{j >= 1} Z_j)
// Each launch is short and covers the machine (16 tiles x up to 9 scales), so the chain overlaps one or two phases instead
// of all of them.
struct MmBatch {
  int nb;                        // products of this launch
  const float *A[12], *B[12];    // row-major N x N
  float *C[12];
  // one operand is a cumulative smoothing operator, banded with half-width hw: band = 1: B[k][c] = 0 beyond |k - c| > hw (the
  // k range follows the tile's columns); band = 2: A[r][k] = 0 beyond |k - r| > hw (it follows the tile's rows); 0: dense.
  // The K loop then covers [lo - hw, lo + 64 + hw) only: 3 of 8 slices of 32 at the first scales of a 256 x 256 grid.
  int band[12], hw[12];
};
constexpr int kMmThreads = 256, kMmKT = 32;
// C[b] = A[b] B[b]; grid (N / 64, N / 64, nb)
template <int N>
__global__ __launch_bounds__(kMmThreads) void mreg_mm_kernel(MmBatch G) {
  __shared__ float As[2][64][kMmKT + 1];   // A tile: 64 rows x 32 k (padded: a lane reads one k of 32 consecutive rows)
  __shared__ float Bs[2][kMmKT][64 + 4];   // B tile: 32 k x 64 columns
  const int bz = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float *A = G.A[bz], *B = G.B[bz];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, h = lane >> 5;
  const int wr = (wid >> 1) * 32, wc = (wid & 1) * 32;  // the wave's 32 x 32 accumulator inside the tile
  // staging: A tile 64 x 32 = 512 float4 (two per thread), B tile 32 x 64 = 512 float4 (two per thread)
  float4 pa[2], pb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + q * kMmThreads;
      const int ar = e >> 3, ak = (e & 7) * 4;      // 8 float4 per A row
      pa[q] = *(const float4 *)(A + (size_t)(r0 + ar) * N + k0 + ak);
      const int bk = e >> 4, bc = (e & 15) * 4;     // 16 float4 per B row
      pb[q] = *(const float4 *)(B + (size_t)(k0 + bk) * N + c0 + bc);
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
  fetch(kbeg);
  put(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kMmKT, buf ^= 1) {
    if (k0 + kMmKT < kend) fetch(k0 + kMmKT);   // next slice in flight while this one is multiplied
    // lane half h feeds k = 2 s + h of every step: a = A[row][k], b = B[k][column]
#pragma unroll
    for (int s2 = 0; s2 < kMmKT / 2; ++s2) {
      const int k = 2 * s2 + h;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][wr + i][k], Bs[buf][k][wc + i], acc, 0, 0, 0);
    }
    if (k0 + kMmKT < kend) put(buf ^ 1);
    __syncthreads();
  }
  float *C = G.C[bz];
#pragma unroll
  for (int r = 0; r < 16; ++r) C[(size_t)(r0 + wr + mr_row(r, h)) * N + c0 + wc + i] = acc[r];
}

// S planes and values.  grid (NN / 256, slots): slot map as the adjoint: has_l1 ? 0 .. J : 0 only; then the point-source slot.
//   slot 0       S[0] = q_0 - positivity sub-gradient      (final: needs no product)
//   slot 1 .. J  S[j] = q_j - q_{j-1},  q_J = 0
//   pts          S[J + 1] = q^p = lam_pts W_0 sign(Pbar - c_1(Pbar))
// values per block: l1b[j][blk] (scale j < J, from slot j), posb[blk] (slot 0), l1b[J + 1][blk] (point-source slot)
struct MregSArgs {
  MregArgs B;
  float lam_pos;
  int has_l1;
  float *S;      // [J + 2][NN]
  float *l1b;    // [J + 2][nblk]
  float *posb;   // [nblk]
};
__global__ __launch_bounds__(kGmThreads) void mreg_splanes_kernel(MregSArgs G, int NN) {
  __shared__ float red[kGmThreads / 64][2];
  const MregArgs &A = G.B;
  const int J = A.J, nblk = gridDim.x, blk = blockIdx.x;
  const int k = blk * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nh = G.has_l1 ? J + 1 : 1;
  const bool pts = ((int)blockIdx.y >= nh);
  const int j = pts ? 0 : (int)blockIdx.y;
  auto plane = [&](int s) { return s == 0 ? A.X : A.C + (size_t)s * NN; };
  auto lw_of = [&](int s, float lam) { return A.W ? lam * A.W[(size_t)s * NN + k] : lam * A.norms[s]; };
  auto sgn = [](float d, float lw) { return (d > 0.f) ? lw : ((d < 0.f) ? -lw : 0.f); };
  float l1 = 0.f, pos = 0.f;
  if (k < NN) {
    if (pts) {
      const float d = A.P[k] - A.C[(size_t)(J + 1) * NN + k], lw = lw_of(0, A.lam_pts);
      G.S[(size_t)(J + 1) * NN + k] = sgn(d, lw);
      l1 = lw * fabsf(d);
    } else if (j == 0) {
      const float hv = A.X[k];
      float z = 0.f;
      if (G.has_l1) {
        const float d = hv - A.C[(size_t)NN + k], lw = lw_of(0, A.lam_hf);
        z = sgn(d, lw);
        l1 = lw * fabsf(d);
      }
      if (G.lam_pos != 0.f && hv < 0.f) {
        pos = -G.lam_pos * hv;
        z -= G.lam_pos;
      }
      G.S[k] = z;
    } else {
      const float cm = plane(j - 1)[k], cj = plane(j)[k];
      const float qm = sgn(cm - cj, lw_of(j - 1, j - 1 == 0 ? A.lam_hf : A.lam_sc));
      float qj = 0.f;
      if (j < J) {
        const float d = cj - plane(j + 1)[k], lw = lw_of(j, A.lam_sc);
        qj = sgn(d, lw);
        l1 = lw * fabsf(d);
      }
      G.S[(size_t)j * NN + k] = qj - qm;
    }
  }
  l1 = wave_sum_shfl(l1);
  pos = wave_sum_shfl(pos);
  if (lane == 0) {
    red[wid][0] = l1;
    red[wid][1] = pos;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t0 = 0.f, t1 = 0.f;
    for (int w = 0; w < kGmThreads / 64; ++w) {
      t0 += red[w][0];
      t1 += red[w][1];
    }
    if (pts) G.l1b[(size_t)(J + 1) * nblk + blk] = t0;
    else if (j < J) G.l1b[(size_t)j * nblk + blk] = t0;
    if (!pts && j == 0) G.posb[blk] = t1;
  }
}

// greg = S[0] + sum_{j = 1 .. J} Z[j]; the point-source slot: z = S[J + 1] - Z[J + 1] and its inner products with the
// Gaussians of the sources (per-block partials, gm_pts_inner_kernel's contract); a second, one-wave launch adds the per-block
// values and inner products into regs (regs[0] = l1, regs[1] = positivity, regs[2] = point-source term, regs[4 ..] the inner
// products; fixed order) and raises the completion flag (mreg_regs2_kernel).
__global__ __launch_bounds__(kGmThreads) void mreg_finish2_kernel(int N, int J, int has_l1, int has_pts, int ss, int M, const float *S,
                                                                  const float *Z, const float *cx, const float *cy, float *greg,
                                                                  float *pts_part) {
  __shared__ float red[kGmThreads / 64][kMaxSources * 3];
  const int NN = N * N;
  const int k = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool in = k < NN;
  if (in) {
    float g = S[k];
    if (has_l1)
      for (int s = 1; s <= J; ++s) g += Z[(size_t)s * NN + k];
    greg[k] = g;
  }
  if (has_pts) {
    const float c0 = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
    const float z = in ? S[(size_t)(J + 1) * NN + k] - Z[(size_t)(J + 1) * NN + k] : 0.f;
    const int u = in ? k / N : 0, v = in ? k % N : 0;
    for (int i = 0; i < M; ++i) {
      const float tx = (float)v - (c0 + ss * cx[i]), ty = (float)u - (c0 + ss * cy[i]);
      const float gq = z * nrm2 * expf(-0.5f * (tx * tx + ty * ty) * inv_s2);
      const float sa = wave_sum_shfl(gq), sx = wave_sum_shfl(gq * tx * inv_s2), sy = wave_sum_shfl(gq * ty * inv_s2);
      if (lane == 0) {
        red[wid][i * 3] = sa;
        red[wid][i * 3 + 1] = sx;
        red[wid][i * 3 + 2] = sy;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < 3 * M) {
      float acc = 0.f;
      for (int w = 0; w < kGmThreads / 64; ++w) acc += red[w][threadIdx.x];
      pts_part[(size_t)blockIdx.x * 3 * kMaxSources + threadIdx.x] = acc;
    }
  }
}
// inner products of the point-source term (lanes stride over the blocks, fixed combine order), then the completion flag
__global__ void mreg_regs2_kernel(int J, int has_l1, int has_pts, int nblocks, int M, const float *l1b, const float *posb,
                                  const float *pts_part, float *regs, unsigned int *done_flag, unsigned int done_seq) {
  const int lane = threadIdx.x;
  {  // values: l1 over the scales, positivity, point-source term (per-block partials; lanes stride, fixed combine order)
    float a = 0.f, b = 0.f, c = 0.f;
    if (has_l1) {  // (four running sums per lane: the loads of a lane do not wait for each other's additions)
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const int n = J * nblocks;
      int i = lane;
      for (; i + 192 < n; i += 256) {
        a0 += l1b[i];
        a1 += l1b[i + 64];
        a2 += l1b[i + 128];
        a3 += l1b[i + 192];
      }
      for (; i < n; i += 64) a0 += l1b[i];
      a = (a0 + a1) + (a2 + a3);
    }
    for (int i = lane; i < nblocks; i += 64) b += posb[i];
    if (has_pts)
      for (int i = lane; i < nblocks; i += 64) c += l1b[(size_t)(J + 1) * nblocks + i];
    a = wave_sum_shfl(a);
    b = wave_sum_shfl(b);
    c = wave_sum_shfl(c);
    if (lane == 0) {
      regs[0] = a;
      regs[1] = b;
      if (has_pts) regs[2] = c;
    }
  }
  if (has_pts)
    for (int t = 0; t < 3 * M; ++t) {
      float acc = 0.f;
      for (int blk = lane; blk < nblocks; blk += 64) acc += pts_part[(size_t)blk * 3 * kMaxSources + t];
      acc = wave_sum_shfl(acc);
      if (lane == 0) regs[4 + t] = acc;
    }
  if (done_flag) {  // (one wave; every write of this chain before the flag)
    __threadfence();
    if (lane == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- the second form of the chain as ONE launch --------------------------------------------------------------------------
// The eight launches above take 5 - 7.5 us each for 1 - 2 us of work (a kernel boundary, the write-back of the XCD's L2 at
// the end of every kernel, a cold start): ~60 us per iteration at N = 128, hidden behind a 53 us epoch kernel but the critical
// path once the epoch kernel runs as a cluster launch (42 us; profiles/r04_*).  mreg_chain_kernel runs the same stages -
// same arithmetic, same summation orders: bit-identical results - in one launch of kChainBlocks resident workgroups, the
// stages separated by cluster_sync (joint_kernels.h) over all of them; what one workgroup hands to another goes through
// write-through stores and L1-bypassing loads (xwg_*).  The element-wise stages walk the blocks of the launch form as
// virtual blocks (same per-block partial sums).  Only where all its workgroups fit beside the epoch kernel's (the host
// checks); every wait is bounded, and a chain that gave up never raises the completion flag, which the update's own bounded
// wait reports.
constexpr int kChainBlocks = 64;
struct MregChainArgs {
  MmBatch mm[4];               // f1, f2, a1, a2
  int xa[4], xb[4];            // operand A / B of product q is an intermediate of this launch (L1-bypassing loads); xa = 2: only in
                               // the last product of the batch (the point-source channel Pbar of stage 0)
  MregSArgs G;                 // S planes
  int sslots;
  int with_pts, N, ss, E, M, J, has_l1;
  const float *a, *cx, *cy;
  float *pbar;
  const float *S, *Z;
  float *greg, *pts_part;
  float *regs;
  unsigned int *done_flag;
  unsigned int done_seq;
  unsigned int *flags;         // [kChainBlocks] sync words, then the abort word
  unsigned int base;           // sequence number of the last sync before this launch
};
constexpr int kChainSyncs = 7;

template <int N>
__device__ __forceinline__ void chain_mm_tile(const MmBatch &G, int bz, int ty, int tx, bool xa, bool xb, float (*As)[64][kMmKT + 1],
                                              float (*Bs)[kMmKT][64 + 4]) {
  const int r0 = ty * 64, c0 = tx * 64;
  const float *A = G.A[bz], *B = G.B[bz];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, h = lane >> 5;
  const int wr = (wid >> 1) * 32, wc = (wid & 1) * 32;
  float4 pa[2], pb[2];
  auto ld4 = [&](const float *p, bool x) -> float4 {
    if (x) {
      const float2 lo = xwg_load<true>((const float2 *)p), hi = xwg_load<true>((const float2 *)p + 1);
      return make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    return *(const float4 *)p;
  };
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + q * kMmThreads;
      const int ar = e >> 3, ak = (e & 7) * 4;
      pa[q] = ld4(A + (size_t)(r0 + ar) * N + k0 + ak, xa);
      const int bk = e >> 4, bc = (e & 15) * 4;
      pb[q] = ld4(B + (size_t)(k0 + bk) * N + c0 + bc, xb);
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
  fetch(kbeg);
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
    if (k0 + kMmKT < kend) put(buf ^ 1);
    __syncthreads();
  }
  float *C = G.C[bz];
#pragma unroll
  for (int r = 0; r < 16; ++r) xwg_storef<true>(&C[(size_t)(r0 + wr + mr_row(r, h)) * N + c0 + wc + i], acc[r]);
}

template <int N>
__global__ __launch_bounds__(kMmThreads) void mreg_chain_kernel(MregChainArgs Q) {
  static_assert(kMmThreads == kGmThreads, "one block shape for the products and the element-wise stages");
  __shared__ float As[2][64][kMmKT + 1];
  __shared__ float Bs[2][kMmKT][64 + 4];
  __shared__ float red[kGmThreads / 64][kMaxSources * 3 + 2];
  __shared__ float abar[kMaxSources];
  __shared__ int st[2];
  constexpr int NN = N * N, NVB = NN / kGmThreads, TPB = (N / 64) * (N / 64);
  const int blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  unsigned int seq = Q.base;
  auto sync = [&](bool first) -> bool {
    seq = (seq + 1u) & 0x0fffffffu;
    return cluster_sync(Q.flags, blk, kChainBlocks, seq, 0u, false, Q.flags + kChainBlocks, tid, st, first);
  };
  auto LDF = [&](const float *p) { return xwg_loadf<true>(p); };
  const MregArgs &A = Q.G.B;
  const int J = Q.J;
  // ---- stage 0: Pbar ---------------------------------------------------------------------------------------------
  if (Q.with_pts) {
    for (int i = wid; i < Q.M; i += kGmThreads / 64) {
      float acc = 0.f;
      for (int e = lane; e < Q.E; e += 64) acc += Q.a[e * Q.M + i];
      acc = wave_sum_shfl(acc);
      if (lane == 0) abar[i] = acc / (float)Q.E;
    }
    __syncthreads();
    const float c0 = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
    for (int vb = blk; vb < NVB; vb += kChainBlocks) {
      const int k = vb * kGmThreads + tid, u = k / N, v = k % N;
      float acc = 0.f;
      for (int i = 0; i < Q.M; ++i) {
        const float tx = (float)v - (c0 + Q.ss * Q.cx[i]), ty = (float)u - (c0 + Q.ss * Q.cy[i]);
        acc = fmaf(abar[i] * nrm2, expf(-0.5f * (tx * tx + ty * ty) * inv_s2), acc);
      }
      xwg_storef<true>(&Q.pbar[k], acc);
    }
  }
  if (!sync(true)) return;
  // ---- stages 1, 2: forward products -----------------------------------------------------------------------------------
  for (int q = 0; q < 2; ++q) {
    for (int t = blk; t < TPB * Q.mm[q].nb; t += kChainBlocks) {
      const int bz = t / TPB;
      const bool xa = Q.xa[q] == 1 || (Q.xa[q] == 2 && bz == Q.mm[q].nb - 1);
      chain_mm_tile<N>(Q.mm[q], bz, (t % TPB) / (N / 64), (t % TPB) % (N / 64), xa, Q.xb[q] != 0, As, Bs);
    }
    if (!sync(false)) return;
  }
  // ---- stage 3: S planes and values (mreg_splanes_kernel's blocks as virtual blocks) -------------------------------------------
  {
    const int nh = Q.G.has_l1 ? J + 1 : 1;
    auto sgn = [](float d, float lw) { return (d > 0.f) ? lw : ((d < 0.f) ? -lw : 0.f); };
    for (int vbs = blk; vbs < NVB * Q.sslots; vbs += kChainBlocks) {
      const int vb = vbs % NVB, slot = vbs / NVB, k = vb * kGmThreads + tid;
      const bool pts = slot >= nh;
      const int j = pts ? 0 : slot;
      auto lw_of = [&](int s, float lam) { return A.W ? lam * A.W[(size_t)s * NN + k] : lam * A.norms[s]; };
      // (plane 0 is h itself, written before this launch; the others are products of this launch)
      auto plane_at = [&](int s) { return s == 0 ? A.X[k] : LDF(A.C + (size_t)s * NN + k); };
      float l1 = 0.f, pos = 0.f;
      if (pts) {
        const float d = LDF(A.P + k) - LDF(A.C + (size_t)(J + 1) * NN + k), lw = lw_of(0, A.lam_pts);
        xwg_storef<true>(&Q.G.S[(size_t)(J + 1) * NN + k], sgn(d, lw));
        l1 = lw * fabsf(d);
      } else if (j == 0) {
        const float hv = A.X[k];
        float z = 0.f;
        if (Q.G.has_l1) {
          const float d = hv - LDF(A.C + (size_t)NN + k), lw = lw_of(0, A.lam_hf);
          z = sgn(d, lw);
          l1 = lw * fabsf(d);
        }
        if (Q.G.lam_pos != 0.f && hv < 0.f) {
          pos = -Q.G.lam_pos * hv;
          z -= Q.G.lam_pos;
        }
        xwg_storef<true>(&Q.G.S[k], z);
      } else {
        const float cm = plane_at(j - 1), cj = plane_at(j);
        const float cn = LDF(A.C + (size_t)min(j + 1, J) * NN + k);   // (unconditional: see joint_kernels.h, load_column)
        const float qm = sgn(cm - cj, lw_of(j - 1, j - 1 == 0 ? A.lam_hf : A.lam_sc));
        float qj = 0.f;
        if (j < J) {
          const float d = cj - cn, lw = lw_of(j, A.lam_sc);
          qj = sgn(d, lw);
          l1 = lw * fabsf(d);
        }
        xwg_storef<true>(&Q.G.S[(size_t)j * NN + k], qj - qm);
      }
      l1 = wave_sum_shfl(l1);
      pos = wave_sum_shfl(pos);
      if (lane == 0) {
        red[wid][0] = l1;
        red[wid][1] = pos;
      }
      __syncthreads();
      if (tid == 0) {
        float t0 = 0.f, t1 = 0.f;
        for (int w = 0; w < kGmThreads / 64; ++w) {
          t0 += red[w][0];
          t1 += red[w][1];
        }
        if (pts) xwg_storef<true>(&Q.G.l1b[(size_t)(J + 1) * NVB + vb], t0);
        else if (j < J) xwg_storef<true>(&Q.G.l1b[(size_t)j * NVB + vb], t0);
        if (!pts && j == 0) xwg_storef<true>(&Q.G.posb[vb], t1);
      }
      __syncthreads();
    }
  }
  if (!sync(false)) return;
  // ---- stages 4, 5: adjoint products -----------------------------------------------------------------------------------
  for (int q = 2; q < 4; ++q) {
    for (int t = blk; t < TPB * Q.mm[q].nb; t += kChainBlocks)
      chain_mm_tile<N>(Q.mm[q], t / TPB, (t % TPB) / (N / 64), (t % TPB) % (N / 64), Q.xa[q] != 0, Q.xb[q] != 0, As, Bs);
    if (!sync(false)) return;
  }
  // ---- stage 6: greg and the inner products of the point-source term (mreg_finish2_kernel's blocks) ---------------------------
  {
    const float c0 = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
    for (int vb = blk; vb < NVB; vb += kChainBlocks) {
      const int k = vb * kGmThreads + tid;
      {
        constexpr int MAXJ = 8;
        float zv[MAXJ];
        float g = LDF(Q.S + k);
#pragma unroll
        for (int s = 1; s <= MAXJ; ++s) zv[s - 1] = LDF(Q.Z + (size_t)min(s, J) * NN + k);
        if (Q.has_l1) {
#pragma unroll
          for (int s = 1; s <= MAXJ; ++s) g += (s <= J) ? zv[s - 1] : 0.f;
        }
        xwg_storef<true>(&Q.greg[k], g);   // (write-through: the update kernel reads it behind the completion flag, L1-bypassing)
      }
      if (Q.with_pts) {
        const float z = LDF(Q.S + (size_t)(J + 1) * NN + k) - LDF(Q.Z + (size_t)(J + 1) * NN + k);
        const int u = k / N, v = k % N;
        for (int i = 0; i < Q.M; ++i) {
          const float tx = (float)v - (c0 + Q.ss * Q.cx[i]), ty = (float)u - (c0 + Q.ss * Q.cy[i]);
          const float gq = z * nrm2 * expf(-0.5f * (tx * tx + ty * ty) * inv_s2);
          const float sa = wave_sum_shfl(gq), sx = wave_sum_shfl(gq * tx * inv_s2), sy = wave_sum_shfl(gq * ty * inv_s2);
          if (lane == 0) {
            red[wid][i * 3] = sa;
            red[wid][i * 3 + 1] = sx;
            red[wid][i * 3 + 2] = sy;
          }
        }
        __syncthreads();
        if (tid < 3 * Q.M) {
          float acc = 0.f;
          for (int w = 0; w < kGmThreads / 64; ++w) acc += red[w][tid];
          xwg_storef<true>(&Q.pts_part[(size_t)vb * 3 * kMaxSources + tid], acc);
        }
        __syncthreads();
      }
    }
  }
  if (!sync(false)) return;
  // ---- stage 7: the values and inner products (mreg_regs2_kernel, one wave), then the completion flag ---------------------------
  if (blk == 0 && wid == 0) {
    const float *l1b = Q.G.l1b, *posb = Q.G.posb;
    float a = 0.f, b = 0.f, c = 0.f;
    if (Q.has_l1) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const int n = J * NVB;
      int i = lane;
      for (; i + 192 < n; i += 256) {
        const float v0 = LDF(l1b + i), v1 = LDF(l1b + i + 64), v2 = LDF(l1b + i + 128), v3 = LDF(l1b + i + 192);
        a0 += v0;
        a1 += v1;
        a2 += v2;
        a3 += v3;
      }
      for (; i < n; i += 64) a0 += LDF(l1b + i);
      a = (a0 + a1) + (a2 + a3);
    }
    for (int i = lane; i < NVB; i += 64) b += LDF(posb + i);
    if (Q.with_pts)
      for (int i = lane; i < NVB; i += 64) c += LDF(l1b + (size_t)(J + 1) * NVB + i);
    a = wave_sum_shfl(a);
    b = wave_sum_shfl(b);
    c = wave_sum_shfl(c);
    if (lane == 0) {
      Q.regs[0] = a;
      Q.regs[1] = b;
      if (Q.with_pts) Q.regs[2] = c;
    }
    if (Q.with_pts)
      for (int t = 0; t < 3 * Q.M; ++t) {
        float acc = 0.f;
        for (int vb = lane; vb < NVB; vb += 64) acc += LDF(Q.pts_part + (size_t)vb * 3 * kMaxSources + t);
        acc = wave_sum_shfl(acc);
        if (lane == 0) Q.regs[4 + t] = acc;
      }
    if (Q.done_flag) {
      __threadfence();
      if (lane == 0) __hip_atomic_store(Q.done_flag, Q.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// test hook (LCMI_REG_DELAY_US): holds the second stream back in front of the chain, so that the update kernel of the
// iteration really has to wait for the chain's completion flag
__global__ void mreg_delay_kernel(long long ticks) {  // ticks of the constant 100 MHz counter
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

}  // namespace lc
