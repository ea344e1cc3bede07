// Global-memory variant of the regularise + update step of the joint fit, for background grids whose
// starlet no longer fits one workgroup's LDS (N = 256: 128 x 128 ROIs, BASELINE.json configs[4]).
// Same arithmetic as joint_update_kernel / starlet_device.h, spread over plain multi-block kernels:
//   forward  : per scale j   tmp = Row_j c ; c' = Col_j tmp ; w = c - c' ; q_j = lam_j W_j sign(w) ; l1 partials
//   backward : per scale j   y = z - q_j ; tmp = Col_j^T y ; z = q_j + Row_j^T tmp      (exact edge-replicating adjoint)
// then one kernel applies AdaBelief to h (all blocks) and to the small parameter blocks (block 0).
// The launches go on the second stream and overlap the epoch kernel, which is long at this size.
#pragma once
#include "joint_kernels.h"

namespace lc {

constexpr int kGmThreads = 256;

__device__ __forceinline__ float gm_b3(int t) { return (t == 0) ? 0.375f : ((t == 1 || t == -1) ? 0.25f : 0.0625f); }

// out = pass along rows (axis 1) or columns (axis 0) of the edge-replicating 5-tap filter with dilation d
__global__ void gm_pass_kernel(int N, int d, int axis, const float *in, float *out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N * N) return;
  const int u = k / N, v = k % N;
  float acc = 0.f;
#pragma unroll
  for (int t = -2; t <= 2; ++t) {
    const int uu = axis == 0 ? min(max(u + t * d, 0), N - 1) : u;
    const int vv = axis == 1 ? min(max(v + t * d, 0), N - 1) : v;
    acc = fmaf(gm_b3(t), in[uu * N + vv], acc);
  }
  out[k] = acc;
}

// c holds c_j on entry and c_{j+1} = cn on exit; q_j = lam W_j sign(c_j - c_{j+1}); per-block l1 partial sums
__global__ void gm_coef_kernel(int N, const float *cn, float *c, const float *Wj, const float *norm, float lam, float *q,
                               float *l1_part) {
  __shared__ float red[kGmThreads / 64];
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  float l1 = 0.f;
  if (k < N * N) {
    const float w = c[k] - cn[k];
    const float lw = lam * (Wj ? Wj[k] : norm[0]);
    l1 = lw * fabsf(w);
    q[k] = (w > 0.f) ? lw : ((w < 0.f) ? -lw : 0.f);
    c[k] = cn[k];
  }
  l1 = wave_sum_shfl(l1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l1;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kGmThreads / 64; ++w) t += red[w];
    l1_part[blockIdx.x] = t;
  }
}

__global__ void gm_sub_kernel(int NN, const float *z, const float *q, float *y) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < NN) y[k] = z[k] - q[k];
}

// sums of the samples that clamp onto the two ends of every line (x <= d, x <= 2d, x >= N-1-d, x >= N-1-2d): one wave
// per line, lanes stride over the line.  edge[line][4]
__global__ void gm_edge_kernel(int N, int d, int axis, const float *in, float *edge) {
  const int line_id = blockIdx.x, lane = threadIdx.x;
  const int stride = axis == 0 ? N : 1;
  const float *line = in + (axis == 0 ? line_id : line_id * N);
  float s1 = 0.f, s2 = 0.f, e1 = 0.f, e2 = 0.f;
  for (int x = lane; x < N; x += 64) {
    const float g = line[x * stride];
    s1 += (x <= d) ? g : 0.f;
    s2 += (x <= 2 * d) ? g : 0.f;
    e1 += (x >= N - 1 - d) ? g : 0.f;
    e2 += (x >= N - 1 - 2 * d) ? g : 0.f;
  }
  s1 = wave_sum_shfl(s1);
  s2 = wave_sum_shfl(s2);
  e1 = wave_sum_shfl(e1);
  e2 = wave_sum_shfl(e2);
  if (lane == 0) {
    edge[line_id * 4 + 0] = s1;
    edge[line_id * 4 + 1] = s2;
    edge[line_id * 4 + 2] = e1;
    edge[line_id * 4 + 3] = e2;
  }
}
// adjoint of gm_pass_kernel along `axis` (end sums from gm_edge_kernel); when q != null the result is q + adjoint
__global__ void gm_pass_adjoint_kernel(int N, int d, int axis, const float *in, const float *edge, const float *q, float *out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N * N) return;
  const int u = k / N, v = k % N;
  const int x = axis == 0 ? u : v;               // position along the line
  const int line_id = axis == 0 ? v : u;
  const int stride = axis == 0 ? N : 1;
  const float *line = in + (axis == 0 ? v : u * N);
  float acc;
  if (x > 0 && x < N - 1) {
    acc = 0.f;
#pragma unroll
    for (int t = -2; t <= 2; ++t) {
      const int xx = x - t * d;
      if (xx >= 0 && xx <= N - 1) acc = fmaf(gm_b3(t), line[xx * stride], acc);
    }
  } else if (x == 0) {
    acc = 0.375f * line[0] + 0.25f * edge[line_id * 4 + 0] + 0.0625f * edge[line_id * 4 + 1];
  } else {
    acc = 0.375f * line[(N - 1) * stride] + 0.25f * edge[line_id * 4 + 2] + 0.0625f * edge[line_id * 4 + 3];
  }
  out[k] = (q ? q[k] : 0.f) + acc;
}

// greg += positivity sub-gradient; per-block positivity partial sums
__global__ void gm_positivity_kernel(int NN, const float *h, float lam_pos, float *greg, float *pos_part) {
  __shared__ float red[kGmThreads / 64];
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  float pos = 0.f;
  if (k < NN && lam_pos != 0.f && h[k] < 0.f) {
    pos = -lam_pos * h[k];
    greg[k] -= lam_pos;
  }
  pos = wave_sum_shfl(pos);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pos;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kGmThreads / 64; ++w) t += red[w];
    pos_part[blockIdx.x] = t;
  }
}

// sums of the partials -> regs[0] = l1, regs[1] = positivity: one wave, lanes stride over the partials, fixed combine order
__global__ void gm_regs_kernel(int nparts_l1, int nblocks, const float *l1_part, const float *pos_part, float *regs) {
  const int lane = threadIdx.x;
  float a = 0.f, b = 0.f;
  for (int i = lane; i < nparts_l1; i += 64) a += l1_part[i];
  for (int i = lane; i < nblocks; i += 64) b += pos_part[i];
  a = wave_sum_shfl(a);
  b = wave_sum_shfl(b);
  if (lane == 0) {
    regs[0] = a;
    regs[1] = b;
  }
}

// ---- regularization_strength_pts_source for the large grids: lam * sum W_0 |starlet_0(Pbar)|, Pbar = sum_i abar_i G(c_i) ----
// abar[i] = mean over the epochs of a[e][i]: from the parameters themselves (one GPU) or from the reduced block
__global__ void gm_abar_kernel(int E, int M, int NN, const float *a, const float *a_ref, const float *shared, int from_shared, float *abar) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = wid; i < M; i += nw) {
    if (from_shared) {
      if (lane == 0) abar[i] = a_ref[i] + shared[NN + 2 * M + i] / shared[NN + 4 * M + 1];
    } else {
      float acc = 0.f;
      for (int e = lane; e < E; e += 64) acc += a[e * M + i];
      acc = wave_sum_shfl(acc);
      if (lane == 0) abar[i] = acc / (float)E;
    }
  }
}
__global__ void gm_pbar_kernel(int N, int ss, int M, const float *abar, const float *cx, const float *cy, float *pbar) {
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
// z = q - adj (adjoint of the scale-0 smoothing applied to q); per-block partial inner products with G_i and its
// position derivatives: part[block][i][3] = sum z G_i, sum z G_i tx / sigma^2, sum z G_i ty / sigma^2
__global__ void gm_pts_inner_kernel(int N, int ss, int M, const float *q, const float *adj, const float *cx, const float *cy,
                                    float *part) {
  __shared__ float red[kGmThreads / 64][kMaxSources * 3];
  const int k = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float c0 = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
  const bool in = k < N * N;
  const float z = in ? q[k] - adj[k] : 0.f;
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
    part[(size_t)blockIdx.x * 3 * kMaxSources + threadIdx.x] = acc;
  }
}
// The same term in ONE launch (the sharded drive evaluates it behind the all-reduce, on the critical path: twelve short launches
// there cost 40 us).  Only scale 0 is involved - a 5 x 5 separable stencil - so a block can own a 16 x 16 tile: Pbar on the
// tile and a halo of 4 (evaluated at clamped coordinates, which IS the edge replication), row and column smoothing on
// shrinking halos, w and q on the tile + 2, the exact adjoint (the same end-of-line sums as gm_pass_adjoint_kernel: rows
// 0..2 / N-3..N-1 of q are inside the halo of the tiles that need them) and the inner products of gm_pts_inner_kernel.
// part / l1_part per tile in the layout gm_pts_final_kernel sums ((N / 16)^2 tiles = N^2 / 256 blocks).
constexpr int kPtT = 16;
// (shared == null: the mean fluxes are taken from the fluxes themselves, a [E][M] - one GPU, all epochs local; summed as
//  mreg_pbar_kernel sums them)
__global__ __launch_bounds__(kGmThreads) void gm_pts_direct_kernel(int N, int ss, int M, const float *a_ref, const float *shared,
                                                                    const float *cx, const float *cy, const float *W0,
                                                                    const float *norm, float lam, float *part, float *l1_part,
                                                                    const float *a = nullptr, int E = 0) {
  constexpr int T = kPtT, H4 = T + 8, H2 = T + 4;
  __shared__ float P[H4][H4 + 1], R1[H4][H2 + 1], Q[H2][H2 + 1], TA[T][H2 + 1];
  __shared__ float AB[kMaxSources], CX[kMaxSources], CY[kMaxSources];
  __shared__ float red[kGmThreads / 64][kMaxSources * 3 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, NN = N * N;
  const int tiles = N / T, tu0 = (blockIdx.x / tiles) * T, tv0 = (blockIdx.x % tiles) * T;
  const float c0 = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
  if (shared) {
    if (tid < M) AB[tid] = a_ref[tid] + shared[NN + 2 * M + tid] / shared[NN + 4 * M + 1];
  } else {
    for (int i = wid; i < M; i += kGmThreads / 64) {
      float acc = 0.f;
      for (int e = lane; e < E; e += 64) acc += a[e * M + i];
      acc = wave_sum_shfl(acc);
      if (lane == 0) AB[i] = acc / (float)E;
    }
  }
  if (tid < M) {
    CX[tid] = c0 + ss * cx[tid];
    CY[tid] = c0 + ss * cy[tid];
  }
  __syncthreads();
  auto gauss = [&](int i, int u, int v, float &tx, float &ty) {
    tx = (float)v - CX[i];
    ty = (float)u - CY[i];
    return nrm2 * expf(-0.5f * (tx * tx + ty * ty) * inv_s2);
  };
  for (int k = tid; k < H4 * H4; k += kGmThreads) {  // Pbar at the clamped coordinates of the tile + 4
    const int ru = k / H4, rv = k % H4;
    const int u = min(max(tu0 - 4 + ru, 0), N - 1), v = min(max(tv0 - 4 + rv, 0), N - 1);
    float acc = 0.f, tx, ty;
    for (int i = 0; i < M; ++i) acc = fmaf(AB[i], gauss(i, u, v, tx, ty), acc);
    P[ru][rv] = acc;
  }
  __syncthreads();
  for (int k = tid; k < H4 * H2; k += kGmThreads) {  // along the rows (axis 1), columns of the tile + 2
    const int ru = k / H2, rv = k % H2;
    float acc = 0.f;
#pragma unroll
    for (int t = -2; t <= 2; ++t) acc = fmaf(gm_b3(t), P[ru][rv + 2 + t], acc);
    R1[ru][rv] = acc;
  }
  __syncthreads();
  for (int k = tid; k < H2 * H2; k += kGmThreads) {  // along the columns (axis 0); w, q on the tile + 2 (zero off the grid)
    const int ru = k / H2, rv = k % H2;
    const int u = tu0 - 2 + ru, v = tv0 - 2 + rv;
    float q = 0.f;
    if (u >= 0 && u < N && v >= 0 && v < N) {
      float cn = 0.f;
#pragma unroll
      for (int t = -2; t <= 2; ++t) cn = fmaf(gm_b3(t), R1[ru + 2 + t][rv], cn);
      const float w = P[ru + 2][rv + 2] - cn;
      const float lw = lam * (W0 ? W0[u * N + v] : norm[0]);
      q = (w > 0.f) ? lw : ((w < 0.f) ? -lw : 0.f);
      if (ru >= 2 && ru < 2 + T && rv >= 2 && rv < 2 + T) P[ru + 2][rv + 2] = lw * fabsf(w);  // (own pixel: its share of the value)
    }
    Q[ru][rv] = q;
  }
  __syncthreads();
  for (int k = tid; k < T * H2; k += kGmThreads) {  // adjoint along axis 0 at the tile's rows, columns of the tile + 2
    const int r = k / H2, rv = k % H2, u = tu0 + r, ru = r + 2;
    float acc;
    if (u > 0 && u < N - 1) {
      acc = 0.f;
#pragma unroll
      for (int t = -2; t <= 2; ++t) acc = fmaf(gm_b3(t), Q[ru - t][rv], acc);  // (q is zero off the grid)
    } else if (u == 0) {
      const float s1 = Q[ru][rv] + Q[ru + 1][rv], s2 = s1 + Q[ru + 2][rv];
      acc = 0.375f * Q[ru][rv] + 0.25f * s1 + 0.0625f * s2;
    } else {
      const float e1 = Q[ru][rv] + Q[ru - 1][rv], e2 = e1 + Q[ru - 2][rv];
      acc = 0.375f * Q[ru][rv] + 0.25f * e1 + 0.0625f * e2;
    }
    TA[r][rv] = acc;
  }
  __syncthreads();
  float z = 0.f, l1 = 0.f;
  const int r = tid / T, cc = tid % T, u = tu0 + r, v = tv0 + cc;
  {  // adjoint along axis 1, z = q - adj at the thread's own pixel
    const int rv = cc + 2;
    float acc;
    if (v > 0 && v < N - 1) {
      acc = 0.f;
#pragma unroll
      for (int t = -2; t <= 2; ++t) acc = fmaf(gm_b3(t), TA[r][rv - t], acc);
    } else if (v == 0) {
      const float s1 = TA[r][rv] + TA[r][rv + 1], s2 = s1 + TA[r][rv + 2];
      acc = 0.375f * TA[r][rv] + 0.25f * s1 + 0.0625f * s2;
    } else {
      const float e1 = TA[r][rv] + TA[r][rv - 1], e2 = e1 + TA[r][rv - 2];
      acc = 0.375f * TA[r][rv] + 0.25f * e1 + 0.0625f * e2;
    }
    z = Q[r + 2][rv] - acc;
    l1 = P[r + 4][cc + 4];
  }
  for (int i = 0; i < M; ++i) {
    float tx, ty;
    const float gq = z * gauss(i, u, v, tx, ty);
    const float sa = wave_sum_shfl(gq), sx = wave_sum_shfl(gq * tx * inv_s2), sy = wave_sum_shfl(gq * ty * inv_s2);
    if (lane == 0) {
      red[wid][i * 3] = sa;
      red[wid][i * 3 + 1] = sx;
      red[wid][i * 3 + 2] = sy;
    }
  }
  l1 = wave_sum_shfl(l1);
  if (lane == 0) red[wid][kMaxSources * 3] = l1;
  __syncthreads();
  if (tid < 3 * M) {
    float acc = 0.f;
    for (int w = 0; w < kGmThreads / 64; ++w) acc += red[w][tid];
    part[(size_t)blockIdx.x * 3 * kMaxSources + tid] = acc;
  }
  if (tid == 64) {
    float acc = 0.f;
    for (int w = 0; w < kGmThreads / 64; ++w) acc += red[w][kMaxSources * 3];
    l1_part[blockIdx.x] = acc;
  }
}

// ordered final sums -> regs[2] = value of the term, regs[4 + 3 i + q] = the three inner products of source i
__global__ void gm_pts_final_kernel(int nblocks, int M, const float *part, const float *l1_part, float *regs) {
  // one wave per quantity (the waves of the block take the 3 M + 1 quantities in turn): the 64 lanes stride over the blocks
  // (4 independent loads in flight per lane), then across the lanes in a fixed order - the serial loop of one thread per
  // quantity cost 60 us at 256 blocks, one wave for all of them 11.6 us (it sits behind the all-reduce of a sharded fit)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int t = wid; t <= 3 * M; t += nw) {
    float acc = 0.f;
    for (int b = lane; b < nblocks; b += 64) acc += (t < 3 * M) ? part[(size_t)b * 3 * kMaxSources + t] : l1_part[b];
    acc = wave_sum_shfl(acc);
    if (lane == 0) regs[(t < 3 * M) ? 4 + t : 2] = acc;
  }
}

struct PtsSepArgs {
  int N, ss, E, M;
  const float *a, *cx, *cy;   // fluxes [E][M] (mean over the epochs taken here) and positions
  const float *abar_sum;      // or: sums over ALL epochs of (a - a_ref) per source (the all-reduced block of a sharded fit) ...
  const float *a_ref;         // ... with the references they are centred on, and
  const float *n_total;       // ... the total epoch count (one float)
  const float *W0;            // [NN] weights of scale 0, or null (then norms[0])
  const float *norms;
  float lam_pts;
  float *part;                // [kPtsBlocks][kPtsStride]
  int write_through;          // the consumer reads behind a completion counter, not behind a kernel boundary
};


// The point-source starlet term from separable tables (derivation: joint_reg_fused.h).  One block of it: pixels
// [pb NN / kPtsBlocks, (pb + 1) NN / kPtsBlocks).  tab: 4 M N floats of LDS.
// (N^2 a multiple of kPtsBlocks kGmThreads: N >= 128; all kGmThreads threads of the block call)
__device__ __forceinline__ void pts_sep_block(const PtsSepArgs &P, int pb, float *tab, float (*red)[kPtsStride], float *abar) {
  constexpr int NT = kGmThreads;
  const int N = P.N, NN = N * N, PX = NN / kPtsBlocks;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, M = P.M;
  const float cen = (N - 1) * 0.5f, inv_s2 = 1.0f / (kSigmaG * kSigmaG), nrm2 = 0.15915494309189535f * inv_s2;
  if (P.abar_sum) {
    if (tid < M) abar[tid] = P.a_ref[tid] + P.abar_sum[tid] / P.n_total[0];
  } else {
    for (int q = wid; q < M; q += NT / 64) {  // mean fluxes, as mreg_pbar_kernel sums them
      float acc = 0.f;
      for (int e = lane; e < P.E; e += 64) acc += P.a[e * M + q];
      acc = wave_sum_shfl(acc);
      if (lane == 0) abar[q] = acc / (float)P.E;
    }
  }
  float *gx = tab, *sgx = tab + M * N, *dgx = tab + 2 * M * N, *sdgx = tab + 3 * M * N;
  for (int t = tid; t < M * N; t += NT) {
    const int i = t / N, v = t % N;
    const float tx = (float)v - (cen + P.ss * P.cx[i]);
    const float g = expf(-0.5f * tx * tx * inv_s2);
    gx[t] = g;
    dgx[t] = g * tx * inv_s2;
  }
  __syncthreads();
  const float b0 = 0.0625f, b1 = 0.25f, b2 = 0.375f;
  for (int t = tid; t < M * N; t += NT) {
    const int i = t / N, v = t % N;
    const int m2 = max(v - 2, 0), m1 = max(v - 1, 0), p1 = min(v + 1, N - 1), p2 = min(v + 2, N - 1);
    const float *g = gx + i * N, *d = dgx + i * N;
    sgx[t] = b0 * g[m2] + b1 * g[m1] + b2 * g[v] + b1 * g[p1] + b0 * g[p2];
    sdgx[t] = b0 * d[m2] + b1 * d[m1] + b2 * d[v] + b1 * d[p1] + b0 * d[p2];
  }
  __syncthreads();
  // Two passes over the sources, few registers (this block rides in the first launch of the regulariser chain, whose kernels
  // must fit beside two waiting update blocks per CU: joint_reg_fused.h): first the sign pattern of the thread's pixels -
  // it needs all sources - then source by source the three inner products, the y-direction terms evaluated again.
  constexpr int MAXQ = 4;   // pixels per thread: N <= 256
  const int nq = PX / NT;
  auto yterms = [&](int u, int i, float &gy, float &sgy, float &dgy, float &sdgy) {
    const float yc = cen + P.ss * P.cy[i];
    float g5[5], d5[5];
#pragma unroll
    for (int t = -2; t <= 2; ++t) {
      const float ty = (float)min(max(u + t, 0), N - 1) - yc;
      g5[t + 2] = expf(-0.5f * ty * ty * inv_s2);
      d5[t + 2] = g5[t + 2] * ty * inv_s2;
    }
    gy = g5[2];
    dgy = d5[2];
    sgy = b0 * g5[0] + b1 * g5[1] + b2 * g5[2] + b1 * g5[3] + b0 * g5[4];
    sdgy = b0 * d5[0] + b1 * d5[1] + b2 * d5[2] + b1 * d5[3] + b0 * d5[4];
  };
  float sp[MAXQ], val = 0.f;
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    sp[q] = 0.f;
    if (q < nq) {
      const int k = pb * PX + q * NT + tid, u = k / N, v = k % N;
      float Pv = 0.f, Cv = 0.f;
      for (int i = 0; i < M; ++i) {
        float gy, sgy, dgy, sdgy;
        yterms(u, i, gy, sgy, dgy, sdgy);
        const float ab = abar[i] * nrm2;
        Pv = fmaf(ab, gy * gx[i * N + v], Pv);
        Cv = fmaf(ab, sgy * sgx[i * N + v], Cv);
      }
      const float d = Pv - Cv;
      const float lw = P.W0 ? P.lam_pts * P.W0[k] : P.lam_pts * P.norms[0];
      sp[q] = (d > 0.f) ? lw : ((d < 0.f) ? -lw : 0.f);
      val += lw * fabsf(d);
    }
  }
  val = wave_sum_shfl(val);
  if (lane == 0) red[wid][3 * kMaxSources] = val;
  for (int i = 0; i < M; ++i) {
    float sa = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      if (q < nq) {
        const int k = pb * PX + q * NT + tid, u = k / N, v = k % N;
        float gy, sgy, dgy, sdgy;
        yterms(u, i, gy, sgy, dgy, sdgy);
        const float x0 = gx[i * N + v], x1 = sgx[i * N + v], x2 = dgx[i * N + v], x3 = sdgx[i * N + v];
        sa += sp[q] * nrm2 * (gy * x0 - sgy * x1);
        sx += sp[q] * nrm2 * (gy * x2 - sgy * x3);
        sy += sp[q] * nrm2 * (dgy * x0 - sdgy * x1);
      }
    }
    sa = wave_sum_shfl(sa);
    sx = wave_sum_shfl(sx);
    sy = wave_sum_shfl(sy);
    if (lane == 0) {
      red[wid][3 * i] = sa;
      red[wid][3 * i + 1] = sx;
      red[wid][3 * i + 2] = sy;
    }
  }
  __syncthreads();
  if (tid < 3 * M || tid == 3 * kMaxSources) {
    float t = 0.f;
    for (int w = 0; w < NT / 64; ++w) t += red[w][tid];
    float *dst = P.part + (size_t)pb * kPtsStride + tid;
    if (P.write_through) xwg_storef<true>(dst, t);
    else *dst = t;
  }
}


// regs[t] from what the four-launch chain left (joint_reg_fused.h): t = 0 l1 (per-tile values of every scale), 1 positivity,
// 2 the point-source term, 4 + q its inner products (per-block partials of pts_sep_block).  All threads of the block call: the
// waves take the quantities in turn, the lanes of a wave stride over the partials (independent loads), one shuffle tree per
// quantity - fixed order.  Without the point-source term (npts == 0) its slots are left alone: the sharded drive fills them
// behind the all-reduce.
__device__ __forceinline__ void planes_regs_block(const RegPlanes &P, int M, float *out, bool coherent, int t_first = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  for (int t = t_first + wid; t < 4 + 3 * M; t += nw) {
    if (t == 3 || (t >= 2 && P.npts == 0)) continue;
    const float *base;
    int n, stride = 1;
    if (t == 0) { base = P.vals; n = P.J * P.ntile; }
    else if (t == 1) { base = P.vals + (size_t)P.J * P.ntile; n = P.ntile; }
    else if (t == 2) { base = P.pts_part + 3 * kMaxSources; n = P.npts; stride = kPtsStride; }
    else { base = P.pts_part + (t - 4); n = P.npts; stride = kPtsStride; }
    float acc = 0.f;
    for (int i = lane; i < n; i += 64) acc += ld_coherent(base + (size_t)i * stride, coherent);
    acc = wave_sum_shfl(acc);
    if (lane == 0) out[t] = acc;
  }
}

// AdaBelief on the small parameter blocks and the loss of the iteration (one block of kGmThreads threads; same rules as
// joint_update_kernel).  A.greg / A.regs hold the h regulariser when A.reg_mode == 2.
// sc: the scalar part of the reduced block (A.shared + N * N, or a copy in LDS); parts: bit 0 = fluxes, positions and the
// loss, bit 1 = dx, dy, mean (independent of the reduction: a block of their own in the fused launch)
// flux elements tid and tid + kGmThreads with their gradient and AdaBelief moments: requested by the fused kernel before
// the scalar reduction they have to wait for, so that one load latency serves both
struct FluxPre {
  float av0, ga0, pm0, ps0, av1, ga1, pm1, ps1;
};
__device__ __forceinline__ void gm_flux_preload(const JointUpdArgs &A, FluxPre &P) {
  const int tid = threadIdx.x, EM = A.E * A.M;
  P.av0 = P.ga0 = P.pm0 = P.ps0 = P.av1 = P.ga1 = P.pm1 = P.ps1 = 0.f;
  if (tid < EM) {
    P.av0 = A.par[LC_P_A][tid];
    P.ga0 = A.g_a[tid];
    P.pm0 = A.pm[LC_P_A][tid];
    P.ps0 = A.ps[LC_P_A][tid];
  }
  if (tid + kGmThreads < EM) {
    P.av1 = A.par[LC_P_A][tid + kGmThreads];
    P.ga1 = A.g_a[tid + kGmThreads];
    P.pm1 = A.pm[LC_P_A][tid + kGmThreads];
    P.ps1 = A.ps[LC_P_A][tid + kGmThreads];
  }
}
// rl: (optional, LDS) the 4 + 3 M values of A.regs, fetched by the caller in one round trip behind the flag; without it every
// use below is a coherent load of its own
// mode_ov / t_ov (>= 0): mode and iteration of this launch where A is a view kept in device memory (batched star photometry)
__device__ __forceinline__ void gm_small_blocks(const JointUpdArgs &A, int N, float lr, float bc1, float bc2, const float *sc,
                                                int parts, const FluxPre *pre = nullptr, const float *rl = nullptr,
                                                int mode_ov = -1, int t_ov = -1) {
  const int mode = mode_ov >= 0 ? mode_ov : A.mode, t_now = t_ov >= 0 ? t_ov : A.t;
  auto reg_at = [&](int k) { return rl ? rl[k] : ld_coherent(A.regs + k, A.wait_flag != nullptr); };
  __shared__ float red[kGmThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int E = A.E, M = A.M, NN = N * N;
  const float Etot = (parts & 1) ? sc[4 * M + 1] : 0.f;
  const bool use_reg = (A.reg_mode == 2);
  const bool pts = (A.lam_pts != 0.f && A.pts_early == 2);  // point-source starlet term evaluated by the reg launch
  float pos_ps = 0.f;
  auto flux = [&](int idx, float av, float ga, float pmv, float psv) {
    const int i = idx % M;
    if (A.lam_pos_ps != 0.f && av < 0.f) {
      pos_ps += -A.lam_pos_ps * av;
      ga -= A.lam_pos_ps;
    }
    if (A.lam_fu != 0.f && Etot > 1.f) {
      const float meanc = sc[2 * M + i] / Etot;  // centred on a_ref (joint_kernels.h, kernel 2)
      const float var = fmaxf(sc[3 * M + i] / Etot - meanc * meanc, 0.f);
      const float sd = sqrtf(var);
      if (sd > 0.f) ga += A.lam_fu * ((av - A.a_ref[i]) - meanc) / (Etot * sd);
    }
    if (pts) ga += reg_at(4 + i * 3) / Etot;
    if (mode == 0) {
      if (A.gout[LC_P_A]) A.gout[LC_P_A][idx] = ga;
    } else if (A.free_mask[LC_P_A]) {
      adabelief_step(av, pmv, psv, ga, lr, bc1, bc2, A.ab);
      A.par[LC_P_A][idx] = av;
      A.pm[LC_P_A][idx] = pmv;
      A.ps[LC_P_A][idx] = psv;
      phist_put(A, LC_P_A, idx, av);
    }
  };
  if (parts & 1) {
    int first = tid;
    if (pre) {
      if (tid < E * M) flux(tid, pre->av0, pre->ga0, pre->pm0, pre->ps0);
      if (tid + kGmThreads < E * M) flux(tid + kGmThreads, pre->av1, pre->ga1, pre->pm1, pre->ps1);
      first = tid + 2 * kGmThreads;
    }
    for (int idx = first; idx < E * M; idx += kGmThreads)
      flux(idx, A.par[LC_P_A][idx], A.g_a[idx], A.pm[LC_P_A][idx], A.ps[LC_P_A][idx]);
  }
  if (parts & 1) LC_USTAMP(5);
  if (parts & 2)
  for (int idx = tid; idx < 3 * E; idx += kGmThreads) {
    const int which = (idx / E == 0) ? LC_P_DX : (idx / E == 1) ? LC_P_DY : LC_P_MEAN;
    const int e = idx % E;
    const float gv = (which == LC_P_DX) ? A.g_dx[e] : (which == LC_P_DY) ? A.g_dy[e] : A.g_mean[e];
    if (mode == 0) {
      if (A.gout[which]) A.gout[which][e] = gv;
    } else if (A.free_mask[which]) {
      float pv_ = A.par[which][e];
      adabelief_step(pv_, A.pm[which][e], A.ps[which][e], gv, lr, bc1, bc2, A.ab);
      A.par[which][e] = pv_;
      phist_put(A, which, e, pv_);
    }
  }
  if (!(parts & 1)) return;
  double prior_loss = 0.0;
  if (tid < 2 * M) {
    const int which = (tid < M) ? LC_P_CX : LC_P_CY, i = tid % M;
    float gv = sc[(which == LC_P_CX ? 0 : M) + i];
    float cv = A.par[which][i];
    if (pts) gv += (A.a_ref[i] + sc[2 * M + i] / Etot) * A.ss * reg_at(4 + i * 3 + (which == LC_P_CX ? 1 : 2));
    if (A.n_prior > 0) {
      const float mu = (which == LC_P_CX) ? A.prior_cx_mean[i] : A.prior_cy_mean[i];
      const float sg = (which == LC_P_CX) ? A.prior_cx_sigma[i] : A.prior_cy_sigma[i];
      gv += (cv - mu) / (sg * sg);
      const double zz = ((double)cv - mu) / sg;
      prior_loss = 0.5 * zz * zz;
    }
    if (mode == 0) {
      if (A.gout[which]) A.gout[which][i] = gv;
    } else if (A.free_mask[which]) {
      adabelief_step(cv, A.pm[which][i], A.ps[which][i], gv, lr, bc1, bc2, A.ab);
      A.par[which][i] = cv;
      phist_put(A, which, i, cv);
    }
  }
  LC_USTAMP(6);
  // loss = 0.5 chi2 + l1 + positivity + positivity of fluxes + flux uniformity + prior
  float part = wave_sum_shfl(pos_ps + (float)prior_loss);
  if (lane == 0) red[wid] = part;
  __syncthreads();
  LC_USTAMP(7);
  if (tid == 0) {
    double loss = 0.5 * (double)sc[4 * M];
    for (int w = 0; w < kGmThreads / 64; ++w) loss += red[w];
    if (use_reg) loss += (double)reg_at(0) + (double)reg_at(1);
    if (pts) loss += (double)reg_at(2);
    if (A.lam_fu != 0.f && Etot > 1.f)
      for (int i = 0; i < M; ++i) {
        const double meanc = sc[2 * M + i] / Etot;
        const double var = fmax((double)sc[3 * M + i] / Etot - meanc * meanc, 0.0);
        loss += A.lam_fu * sqrt(var);
      }
    if (A.hist) A.hist[t_now] = (float)loss;
    if (A.out_loss) *A.out_loss = (float)loss;
  }
}

// The point-source starlet term INSIDE the update launch (sharded / step-by-step drive, where the term needs the all-reduced
// mean fluxes and used to be two launches of its own between the all-reduce and the update): the first kPtsBlocks blocks of
// the grid evaluate it (pts_sep_block) and count themselves in; only the scalar part of block 0 - fluxes, positions, loss -
// needs the result and waits for the count (bounded; all blocks of this small grid are resident together), the pixel blocks
// do not.  ctr grows by kPtsBlocks per launch (seq: its value once this launch's blocks are in).
struct PtsTail {
  int on;
  PtsSepArgs P;
  unsigned int *ctr;
  unsigned int seq;
  unsigned int *err;
};
// AdaBelief on h (every block) and on the small blocks + loss (block 0).  A.greg / A.regs hold the h regulariser.
__global__ __launch_bounds__(kGmThreads) void joint_update_gm_kernel(JointUpdArgs A, int N, PtsTail T) {
  extern __shared__ __align__(16) float gm_dyn[];   // T.on: 4 M N floats for the tables of the point-source blocks
  __shared__ double lanes[kGmThreads];
  __shared__ float pred[kGmThreads / 64][kPtsStride];
  __shared__ float pabar[kMaxSources];
  __shared__ float regl[4 + 3 * kMaxSources];
  const int tid = threadIdx.x;
  const int E = A.E, M = A.M, NN = N * N;
  int bx = blockIdx.x;
  if (T.on) {
    if (bx < kPtsBlocks) {
      pts_sep_block(T.P, bx, gm_dyn, pred, pabar);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(T.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    bx -= kPtsBlocks;
  }
  if (A.fuse_scalar_reduce) {  // grid of one block (the background is not updated)
    reduce_scalars(E, M, NN, A.g_cx_e, A.g_cy_e, A.chi2_e, A.par[LC_P_A], A.a_ref, A.shared_w, lanes, tid);
    __threadfence_block();
    __syncthreads();
  }
  const bool use_reg = (A.reg_mode == 2);
  const float lr = A.lr, bc1 = A.bc1, bc2 = A.bc2;
  const int k = bx * blockDim.x + tid;
  // (sharded drive: the chain's finishing launch counts its blocks into *wait_flag; checking that here spares the cross-stream
  //  event wait in front of this launch - 5 to 7 us on the tail behind the all-reduce.  This grid is at most N^2 / 256 + 64
  //  blocks: a chain that runs late always finds room beside it.)
  wait_for_flag(A.wait_flag, A.wait_seq, A.wait_err);
  if (k < NN) {
    const float g = A.shared[k] + (use_reg ? ld_coherent(A.greg + k, A.wait_flag != nullptr) : 0.f);
    if (A.mode == 0 && A.gout[LC_P_H]) A.gout[LC_P_H][k] = g;
    if (A.mode == 1 && A.free_mask[LC_P_H]) {
      float hv = A.h[k], m = A.mh[k], s = A.sh[k];
      adabelief_step(hv, m, s, g, lr, bc1, bc2, A.ab);
      A.h[k] = hv;
      A.mh[k] = m;
      A.sh[k] = s;
      phist_put(A, LC_P_H, k, hv);
    }
  }
  if (bx != 0) return;
  if (T.on) {
    // values of the background terms from the chain (regs[0 .. 1], behind the stream's event), the point-source term from the
    // blocks of this launch
    wait_for_flag(T.ctr, T.seq, T.err);
    RegPlanes R;
    R.on = 1; R.J = 0; R.ntile = 0; R.npts = kPtsBlocks;
    R.S0 = R.Z = R.vals = nullptr;
    R.pts_part = T.P.part;
    if (tid < 2) regl[tid] = use_reg ? ld_coherent(A.regs + tid, A.wait_flag != nullptr) : 0.f;
    planes_regs_block(R, M, regl, true, 2);
    __syncthreads();
    gm_small_blocks(A, N, lr, bc1, bc2, A.shared + NN, 3, nullptr, regl);
    return;
  }
  gm_small_blocks(A, N, lr, bc1, bc2, A.shared + NN, 3);
}

// Batched star photometry (lightcurver/processes/star_photometry.py:257: the reference fits its <= 30 stars one after the
// other): block g applies to star g - the epochs [e0, e1) of the batch, through the pointers of views[g] - exactly what the
// one-block launch of joint_update_gm_kernel applies to a fit of that star alone: the same scalar reduction over its
// epochs, the same gradient rules, the same AdaBelief step, its own loss history.  Same operands in the same order, so a
// star of the batch ends bit for bit where its separate fit ends (tests/test_star_batch_gpu.py).
__global__ __launch_bounds__(kGmThreads) void joint_update_groups_kernel(const JointUpdArgs *__restrict__ views, int mode, int t,
                                                                         float lr, float bc1, float bc2) {
  __shared__ double lanes[kGmThreads];
  // two blocks per star, as in the one-fit launch (joint_reduce_update_kernel): the shifts and sky levels need nothing from
  // the reduction over the epochs and step in a block of their own beside it
  // The star's view is READ IN PLACE (block-uniform scalar loads; mode and iteration of the launch go to the rules as
  // arguments): a private copy with the launch's fields patched in lived on the stack - 672 bytes of scratch per lane, every
  // field of it a memory access - and the kernel took 15.4 us for two short blocks per star.
  const JointUpdArgs &A = views[blockIdx.x >> 1];
  __shared__ float scl[4 * kMaxSources + 2];  // (the sums also go to LDS: the rules below read them without a trip through L2)
  // (the fluxes with their moments are requested first and arrive while the reduction's loads are in flight)
  if (blockIdx.x & 1) {
    gm_small_blocks(A, 0, lr, bc1, bc2, nullptr, 2, nullptr, nullptr, mode, t);
    return;
  }
  FluxPre pre;
  gm_flux_preload(A, pre);
  reduce_scalars(A.E, A.M, 0, A.g_cx_e, A.g_cy_e, A.chi2_e, A.par[LC_P_A], A.a_ref, A.shared_w, lanes, threadIdx.x, scl);
  __syncthreads();
  gm_small_blocks(A, 0, lr, bc1, bc2, scl, 1, &pre, nullptr, mode, t);
}

// The reduction over the epochs and the update in ONE launch (the device loop of a single GPU, where nothing has to
// happen between the two): block b < nimg sums its 16 pixels of the T^T slabs (joint_reduce_kernel's order) and applies
// AdaBelief to them right away; block nimg reduces the scalars, updates fluxes and positions and writes the loss; block
// nimg + 1 updates the per-epoch shifts and sky levels.
// `tiles` 16-pixel tiles per image block (2 where the regulariser flag is checked in the kernel: at most two 256-thread
// blocks per CU are then resident, so a chain that runs late always finds the wave slots and registers to finish).
// Registers: 197 per lane (the scalar block's side-by-side double-precision sums set the figure, whatever the tile count): two
// resident blocks per CU leave 112 of the 512 registers per lane and SIMD, which is what a wave of every kernel of the
// regulariser chain must fit in (joint_reg_fused.h) - a late chain must be schedulable while this kernel waits for it.
constexpr int kUpdMaxTiles = 4;  // tiles per block the kernel below holds side by side (the host never asks for more)
__global__ __launch_bounds__(kRedThreads) void joint_reduce_update_kernel(JointUpdArgs A, int N, const float *HG, int tiles) {
  static_assert(kRedThreads == kGmThreads, "one block size");
  __shared__ float4 part[kRedParts][kRedPix / 4];
  __shared__ double lanes[kRedThreads];
  const int E = A.E, M = A.M, NN = N * N;
  const int nimg = NN / kRedPix / tiles;  // image blocks
  const int tid = threadIdx.x;
  const float lr = A.lr, bc1 = A.bc1, bc2 = A.bc2;
  // The two single blocks come FIRST in the grid: the scalar block is the longest chain of the launch (tools/update_stamps.py)
  // and, dispatched behind the image blocks, it waited for one of them to leave a CU before it even started (4.5 us into the
  // launch at 512 image blocks, 10.9 us at the 1024 of a 256 x 256 grid).  bid: block among the image blocks, then the two.
  const int bid = ((int)blockIdx.x >= 2) ? (int)blockIdx.x - 2 : nimg + (int)blockIdx.x;
  if (bid < nimg) {
    // The tiles of a block side by side: state and slab loads of every tile requested first (one memory round trip for the
    // block instead of one per tile), the tiles combined one after the other through the LDS buffer, the regulariser's flag,
    // then its gradient for every tile in one more round trip.  Same sums in the same order as tile after tile.
    float hv[kUpdMaxTiles], m[kUpdMaxTiles], sv[kUpdMaxTiles], tsum[kUpdMaxTiles], gr[kUpdMaxTiles];
    float4 acc[kUpdMaxTiles];
    if (bid == 0) LC_USTAMP(0);
    const bool seen = flag_seen(A.wait_flag, A.wait_seq);
#pragma unroll
    for (int tl = 0; tl < kUpdMaxTiles; ++tl) {
      hv[tl] = m[tl] = sv[tl] = tsum[tl] = gr[tl] = 0.f;
      acc[tl] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tl < tiles) {
        const int px0 = (bid * tiles + tl) * kRedPix, px = px0 + tid;
        if (tid < kRedPix) {
          hv[tl] = A.h[px];
          m[tl] = A.mh[px];
          sv[tl] = A.sh[px];
        }
        acc[tl] = reduce_pixels16_partial(E, NN, HG, px0, tid);
      }
    }
#pragma unroll
    for (int tl = 0; tl < kUpdMaxTiles; ++tl) {
      if (tl < tiles) {
        if (tl > 0) __syncthreads();  // the partial sums in LDS are rewritten
        tsum[tl] = reduce_pixels16_combine(acc[tl], part, tid);
      }
    }
    wait_for_flag(A.wait_flag, A.wait_seq, A.wait_err, seen);  // the regulariser of this iteration (second stream)
    if (tid < kRedPix) {
#pragma unroll
      for (int tl = 0; tl < kUpdMaxTiles; ++tl)
        if (tl < tiles && A.reg_mode == 2) {
          const int px = (bid * tiles + tl) * kRedPix + tid;
          gr[tl] = A.planes.on ? planes_greg(A.planes, px, NN, A.wait_flag != nullptr) : ld_coherent(A.greg + px, A.wait_flag != nullptr);
        }
#pragma unroll
      for (int tl = 0; tl < kUpdMaxTiles; ++tl) {
        if (tl < tiles) {
          const int px = (bid * tiles + tl) * kRedPix + tid;
          A.shared_w[px] = tsum[tl];
          adabelief_step(hv[tl], m[tl], sv[tl], tsum[tl] + gr[tl], lr, bc1, bc2, A.ab);
          A.h[px] = hv[tl];
          A.mh[px] = m[tl];
          A.sh[px] = sv[tl];
          phist_put(A, LC_P_H, px, hv[tl]);
        }
      }
    }
    if (bid == nimg - 1) LC_USTAMP(1);
    return;
  }
  if (bid == nimg + 1) {  // shifts and sky levels: nothing to wait for
    gm_small_blocks(A, N, lr, bc1, bc2, nullptr, 2);
    return;
  }
  __shared__ float scl[4 * kMaxSources + 2];
  __shared__ float regl[4 + 3 * kMaxSources];
  FluxPre pre;
  LC_USTAMP(2);
  const bool seen = flag_seen(A.wait_flag, A.wait_seq);
  gm_flux_preload(A, pre);
  reduce_scalars(E, M, NN, A.g_cx_e, A.g_cy_e, A.chi2_e, A.par[LC_P_A], A.a_ref, A.shared_w, lanes, tid, scl);
  __syncthreads();
  LC_USTAMP(3);
  wait_for_flag(A.wait_flag, A.wait_seq, A.wait_err, seen);
  // what the chain left in regs (values of its terms, inner products of the point-source term): one round trip for the block
  const bool have_regs = (A.regs != nullptr) && (A.reg_mode == 2 || (A.lam_pts != 0.f && A.pts_early == 2));
  if (have_regs) {
    if (A.planes.on) {
      planes_regs_block(A.planes, M, regl, A.wait_flag != nullptr);
      // (a point-source term that was not part of the chain - none in the device loop - would sit in regs)
      if (A.planes.npts == 0 && tid >= 2 && tid < 4 + 3 * M) regl[tid] = ld_coherent(A.regs + tid, A.wait_flag != nullptr);
    } else if (tid < 4 + 3 * M) {
      regl[tid] = ld_coherent(A.regs + tid, A.wait_flag != nullptr);
    }
    __syncthreads();
  }
  LC_USTAMP(4);
  gm_small_blocks(A, N, lr, bc1, bc2, scl, 1, &pre, have_regs ? regl : nullptr);
  LC_USTAMP(8);
}

// The same launch with the T_e^T step folded into the reduction (global-spectrum kernels, every epoch a pure translation):
// phase D of the epoch kernel wrote one N x N slab per epoch (4-tap adjoint stencil of the scene gradient) that the reduction
// read back - 32 MB out and 32 MB in per iteration at 125 epochs of 256 x 256, a launch of its own in the phased form.  Here
// the reduction applies the stencil itself, reading the scene-gradient rows phase C' left in the spectrum scratch: a block
// owns 64 consecutive pixels of one row of h, wave w adds the epochs w, w + 4, ... (each lane the four taps of its pixel:
// coalesced 256-byte row segments, all loads of a wave independent), the four waves are combined in order, and the 64
// pixels take their AdaBelief step in the same block.  Same taps and weights as phase D (joint_kernels.h); the border ring
// of h (edge replication) takes phase D's exact ordered gather.
struct StencilSrc {
  const float *gs;      // scene-gradient rows of epoch e: gs + e * epoch_stride + u * row_stride + v
  size_t epoch_stride;  // floats
  int row_stride;       // floats
  const float *shifts;  // [E][2] (dx, dy) as the epoch kernel of this iteration used them (the blocks of this launch that
                        // update dx, dy run concurrently: the parameter arrays themselves are being rewritten)
  int ss;
};
constexpr int kStPix = 256, kStEpochs = 512;
__global__ __launch_bounds__(kRedThreads) void joint_stencil_update_kernel(JointUpdArgs A, int N, StencilSrc S) {
  __shared__ float part[kRedThreads / 64][kStPix];
  __shared__ double lanes[kRedThreads];
  const int E = A.E, M = A.M, NN = N * N;
  const int nimg = NN / kStPix;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float lr = A.lr, bc1 = A.bc1, bc2 = A.bc2;
  if ((int)blockIdx.x < nimg) {
    // a block owns 256 consecutive pixels of h; a lane four consecutive pixels of one row (N is a multiple of 4 ... 256):
    // per epoch two rows of five scene-gradient samples, as phase D takes them (16-byte + 4-byte loads)
    const int px0 = blockIdx.x * kStPix, pxl = px0 + 4 * lane, ky = pxl / N, kx0 = pxl % N;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // Per-epoch constants of the stencil (integer offsets, fractions) go through LDS in chunks of kStEpochs epochs: read per
    // epoch from global memory they put a load -> address -> load chain into every turn of the epoch loop; from LDS the taps
    // of several epochs are in flight at once.
    __shared__ int cI[kStEpochs][2];
    __shared__ float cF[kStEpochs][2];
    const bool row_border = (ky == 0 || ky == N - 1);
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    for (int e0 = 0; e0 < E; e0 += kStEpochs) {
      const int cnt = min(kStEpochs, E - e0);
      __syncthreads();
      for (int i = tid; i < cnt; i += kRedThreads) {
        const float t_nsx = -((float)S.ss * S.shifts[2 * (e0 + i)]), t_nsy = -((float)S.ss * S.shifts[2 * (e0 + i) + 1]);
        const float t_ixf = floorf(t_nsx), t_iyf = floorf(t_nsy);
        cI[i][0] = (int)t_ixf;
        cI[i][1] = (int)t_iyf;
        cF[i][0] = t_nsx - t_ixf;
        cF[i][1] = t_nsy - t_iyf;
      }
      __syncthreads();
#pragma unroll 2
      for (int k = wid; k < cnt; k += kRedThreads / 64) {
        const int ixc = cI[k][0], iyc = cI[k][1];
        const float fxc = cF[k][0], fyc = cF[k][1];
        const float *GS = S.gs + (size_t)(e0 + k) * S.epoch_stride;
        const float w00 = (1.f - fyc) * (1.f - fxc), w01 = (1.f - fyc) * fxc, w10 = fyc * (1.f - fxc), w11 = fyc * fxc;
        const int r0 = ky - iyc, q0 = kx0 - ixc;
        const bool ra = (r0 >= 0 && r0 < N), rb = (r0 >= 1 && r0 <= N);
        const int rca = min(max(r0, 0), N - 1), rcb = min(max(r0 - 1, 0), N - 1);
        float ga[5], gb[5];  // rows r0 and r0 - 1 at columns q0 - 1 .. q0 + 3
        if (q0 >= 1 && q0 + 3 < N) {  // the whole window inside the row: one 16-byte and one 4-byte load per row
          const float *pa = GS + (size_t)rca * S.row_stride + q0 - 1, *pb = GS + (size_t)rcb * S.row_stride + q0 - 1;
          const f4u va = *(const f4u *)pa, vb = *(const f4u *)pb;
          const float ea = pa[4], eb = pb[4];
          ga[0] = va.x; ga[1] = va.y; ga[2] = va.z; ga[3] = va.w; ga[4] = ea;
          gb[0] = vb.x; gb[1] = vb.y; gb[2] = vb.z; gb[3] = vb.w; gb[4] = eb;
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            ga[t] = ra ? ga[t] : 0.f;
            gb[t] = rb ? gb[t] : 0.f;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            const int qq = q0 - 1 + t;
            const bool qin = (qq >= 0 && qq < N);
            const int qc = min(max(qq, 0), N - 1);
            const float a = GS[(size_t)rca * S.row_stride + qc], b = GS[(size_t)rcb * S.row_stride + qc];
            ga[t] = (ra && qin) ? a : 0.f;
            gb[t] = (rb && qin) ? b : 0.f;
          }
        }
        float o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = fmaf(w00, ga[t + 1], fmaf(w01, ga[t], fmaf(w10, gb[t + 1], w11 * gb[t])));
        // the border ring of h collects clamped (edge-replicated) samples: phase D's exact ordered gather
        if (row_border || kx0 == 0 || kx0 + 4 == N) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int kx = kx0 + t;
            if (!(row_border || kx == 0 || kx == N - 1)) continue;
            const int ulo = max((ky == 0) ? 0 : ky - iyc - 1, 0), uhi = min((ky == N - 1) ? N - 1 : ky - iyc, N - 1);
            const int vlo = max((kx == 0) ? 0 : kx - ixc - 1, 0), vhi = min((kx == N - 1) ? N - 1 : kx - ixc, N - 1);
            float ob = 0.f;
            for (int u = ulo; u <= uhi; ++u)
              for (int v = vlo; v <= vhi; ++v) {
                const int x0 = v + ixc, y0 = u + iyc;
                const int xa = min(max(x0, 0), N - 1), xb = min(max(x0 + 1, 0), N - 1);
                const int ya = min(max(y0, 0), N - 1), yb = min(max(y0 + 1, 0), N - 1);
                const float wx = ((xa == kx) ? (1.f - fxc) : 0.f) + ((xb == kx) ? fxc : 0.f);
                const float wy = ((ya == ky) ? (1.f - fyc) : 0.f) + ((yb == ky) ? fyc : 0.f);
                ob = fmaf(wx * wy, GS[(size_t)u * S.row_stride + v], ob);
              }
            o[t] = ob;
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += o[t];
      }
    }
    *(float4 *)&part[wid][4 * lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    // state of this thread's pixel (one pixel per thread from here on): requested before the wait, used after it
    const int px = px0 + tid;
    float hv = A.h[px], m = A.mh[px], sv = A.sh[px];
    __syncthreads();
    wait_for_flag(A.wait_flag, A.wait_seq, A.wait_err);  // the regulariser of this iteration (second stream)
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kRedThreads / 64; ++w) t += part[w][tid];
    const float gr = (A.reg_mode == 2) ? ld_coherent(A.greg + px, A.wait_flag != nullptr) : 0.f;
    A.shared_w[px] = t;
    adabelief_step(hv, m, sv, t + gr, lr, bc1, bc2, A.ab);
    A.h[px] = hv;
    A.mh[px] = m;
    A.sh[px] = sv;
    phist_put(A, LC_P_H, px, hv);
    return;
  }
  if ((int)blockIdx.x == nimg + 1) {  // shifts and sky levels: nothing to wait for
    gm_small_blocks(A, N, lr, bc1, bc2, nullptr, 2);
    return;
  }
  __shared__ float scl[4 * kMaxSources + 2];
  FluxPre pre;
  gm_flux_preload(A, pre);
  reduce_scalars(E, M, NN, A.g_cx_e, A.g_cy_e, A.chi2_e, A.par[LC_P_A], A.a_ref, A.shared_w, lanes, tid, scl);
  __syncthreads();
  wait_for_flag(A.wait_flag, A.wait_seq, A.wait_err);
  gm_small_blocks(A, N, lr, bc1, bc2, scl, 1, &pre);
}

}  // namespace lc
