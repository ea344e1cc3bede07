// Starlet l1 regulariser of the 128 x 128 background grid in TWO stages (third form of the chain; opt-in, LCMI_REG_ROWS=1:
// built, tested, measured slower than the eight launches it replaces - see launch_reg_mfma in joint_fit.hip for the numbers).
//
// The batched-product form (joint_reg_mfma.h) takes eight dependent launches - pbar, T = X AT, c = A T, S planes, T' = S A,
// Z = AT T', sums, values - and every boundary between them is a hand-off between CUs: 5 - 7.5 us per stage whatever
// delivers it (launches: ~60 us per iteration; one launch with in-kernel syncs: 59 us, profiles/r04_cluster_*).  That is
// longer than the epoch kernel it is meant to hide behind (53 us at 64 x 64 stamps, 42 us as a cluster launch), so the
// update of every iteration waits for it.  What shortens it is fewer stages, not cheaper ones.
//
// Same mathematics (joint_reg_mfma.h: c_j = A_j X A_j^T with the cumulative operators A_j = R_{j-1} ... R_0; S_0 = q_0 -
// positivity, S_j = q_j - q_{j-1}, q_J = 0; d l1 / d X = S_0 + sum_{j >= 1} A_j^T S_j A_j), cut by ROWS instead of by product:
//   rows of c_j need rows of A_j only:   c_j[rb, :] = (A_j[rb, :] X) A_j^T     - no other workgroup's data
//   S_j[rb, :] is element-wise in c_{j-1}, c_j, c_{j+1}[rb, :]                - computed where the rows are
//   A_j^T S_j A_j = sum over row blocks rb of  A_j^T[:, rb] (S_j[rb, :] A_j)   - a rank-16 update per row block and scale
// so ONE workgroup (row block rb of 16 rows, scales [s_lo, s_hi]) runs forward products, S rows, and the adjoint products of
// its rows for its scales back to back, everything in LDS and registers, and writes one partial sub-gradient plane
// (accumulated over its scales in MFMA accumulators).  Stage two adds the planes of all workgroups per pixel and the values.
// The point-source starlet term (scale 0 only: a 5 x 5 stencil) takes its one-launch tile form (joint_gm.h,
// gm_pts_direct_kernel) beside it.  Three launches (tiles, rows, sums) instead of eight.
//
// Inside a workgroup the products of its scales run side by side, phase by phase (all T = A X, then all c = T A^T, the S
// rows, all U = S A, then Z): four workgroup barriers in all, and the operator loads of a phase are in flight together.
// v_mfma_f32_16x16x4_f32 (exact fp32 multiply-adds, 64 FLOP / clk / SIMD).  Operands from the operators travel as float4
// along k: lane (i = lane & 15, kq = lane >> 4) loads M[row][k0 + 4 kq .. + 3] once per four MFMA steps and step j sums
// over k = k0 + 4 kq' + j, kq' = 0 .. 3 - a permutation of the k order, which a sum does not mind (and the other operand
// follows it).  The operators are banded (half-width 2 (2^s - 1)): K ranges cover the band only, rounded to 16.
#pragma once
#include "joint_reg_mfma.h"

namespace lc {

typedef float rr_acc __attribute__((ext_vector_type(4)));

constexpr int kRrRows = 16;        // rows per row block (one MFMA tile)
constexpr int kRrThreads = 512;    // eight waves: wave w owns column tile w (two waves per SIMD hide each other's load latency)
constexpr int kRrMaxParts = 3;     // scale groups per row block
constexpr int kRrBatch = 4;        // scales whose products run side by side (LDS: one T / S / U plane each)

struct MregRowsArgs {
  int J, nparts;
  int s_lo[kRrMaxParts], s_hi[kRrMaxParts];   // scales of part p (1-based, inclusive; at most kRrBatch of them)
  const float *A, *AT;    // [J + 1][N][N] cumulative smoothing operators and their transposes (row-major)
  const float *X;         // h
  const float *W;         // [J][N][N] or null (then norms[j])
  const float *norms;     // [J]
  float lam_sc, lam_hf, lam_pos;
  float *Zp;              // [N / 16 * nparts][N][N] partial sub-gradient planes
  float *S0;              // [N][N] q_0 - positivity sub-gradient (needs no product)
  float *vals;            // [N / 16 * nparts][2] l1 value, positivity value of the workgroup
  int dbg;                // diagnostic (LCMI_REG_ROWS_DBG, wrong numbers): 1 = no operator loads, 2 = no LDS operand loads, 4 = no products
};

template <int N>
struct RrCfg {
  static constexpr int XS = N + 16;   // row stride of X and U in LDS (B operands: k in the row index; 16 mod 32 banks)
  static constexpr int TS = N + 2;    // row stride of T1, c and S in LDS (A operands: k in the column index; 2 mod 32 banks)
  static constexpr int NCP = kRrBatch + 2;                 // planes of c rows: the part's scales and their two neighbours
  static constexpr int OFF_X = 0;                          // X [N][XS]; after the forward products: U [kRrBatch][16][XS]
  static constexpr int OFF_C = OFF_X + N * XS;             // c rows [NCP][16][TS]
  static constexpr int OFF_T = OFF_C + NCP * kRrRows * TS; // T1 [kRrBatch][16][TS]; afterwards S [kRrBatch][16][TS]
  static constexpr int OFF_RED = OFF_T + kRrBatch * kRrRows * TS;
  static constexpr int LDS_FLOATS = OFF_RED + 32;
  static constexpr int LDS_BYTES = LDS_FLOATS * 4;
  static_assert(kRrBatch * kRrRows * XS <= N * XS, "U planes fit where X was");
  static_assert(LDS_BYTES <= 163840, "LDS");
  static_assert(N / 16 == kRrThreads / 64, "one column tile per wave");
};

// k range [lo, hi) (multiples of 16 inside [0, N)) of a product whose band of half-width hw is centred on the 16 indices
// starting at c
template <int N>
__device__ __forceinline__ void rr_band(int c, int hw, int &lo, int &hi) {
  lo = max(c - hw, 0) / 16 * 16;
  hi = min((c + 16 + hw + 15) / 16 * 16, N);
}

// One 16 x 16 tile of a banded product: acc = sum over k in [klo, khi) of A(k) x B(k), one operand from the operator (the
// lane's row `op_row`, float4 along k: group g + 1 is requested before group g is multiplied), the other from LDS
// (`oth[k * ostride]`).  OPA: the operator is the A operand.  Deliberately a ROLLED loop in a function of its own (noinline):
// every product of every scale runs the same two hundred instructions.  The unrolled form (all groups, all scales inline)
// was 30 - 50 KB of straight-line code that each workgroup walked through once - 50 us per launch for 10 us of arithmetic and
// latency, whatever was prefetched: instruction fetch from a cold cache, not the operands, is what it waited for.
// (A function of its own sees generic pointers and would load through the FLAT path - measured: 28 flat_load_dword + 7
//  flat_load_dwordx4 per product, ~5 k cycles each.  So the LDS operand arrives as an offset into the kernel's dynamic LDS
//  block and the operator as an address-space-1 pointer: ds_read / global_load again.)
extern __shared__ __align__(16) float rr_lds[];
typedef const float __attribute__((address_space(1))) *rr_gptr;
template <int N, bool OPA>
__device__ __noinline__ rr_acc rr_product(const float *op_row_generic, int klo, int khi, int kq, int oth_off, int ostride, int dbg) {
  constexpr int MAXG = N / 16;
  const rr_gptr op_row = (rr_gptr)op_row_generic;
  const float *oth = rr_lds + oth_off;
  // every group of the operator's row requested at once (clamped past the band: valid bytes, never used), the LDS operand
  // likewise: one memory round trip per product, then the products in two chains (a step waits for its own chain only)
  float4 o[MAXG];
#pragma unroll
  for (int g = 0; g < MAXG; ++g) {
    const rr_gptr q = op_row + min(klo + 16 * g, N - 16) + 4 * kq;
    o[g] = (dbg & 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(q[0], q[1], q[2], q[3]);
  }
  float t[MAXG][4];
#pragma unroll
  for (int g = 0; g < MAXG; ++g) {
    const float *ok = oth + (size_t)(min(klo + 16 * g, N - 16) + 4 * kq) * ostride;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) t[g][jj] = (dbg & 2) ? 1.f : ok[jj * ostride];
  }
  rr_acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < MAXG; ++g) {
    if (klo + 16 * g < khi && !(dbg & 4)) {
      if (OPA) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o[g].x, t[g][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o[g].y, t[g][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o[g].z, t[g][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o[g].w, t[g][3], acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t[g][0], o[g].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t[g][1], o[g].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(t[g][2], o[g].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t[g][3], o[g].w, acc1, 0, 0, 0);
      }
    }
  }
  return acc0 + acc1;
}

template <int N>
__global__ __launch_bounds__(kRrThreads) void mreg_rows_kernel(MregRowsArgs Q) {
  typedef RrCfg<N> C;
  constexpr int XS = C::XS, TS = C::TS, NN = N * N, NT = N / 16, PL = kRrRows * TS, UL = kRrRows * XS;
  float *Xs = rr_lds + C::OFF_X, *Cp = rr_lds + C::OFF_C, *Tp = rr_lds + C::OFF_T, *RED = rr_lds + C::OFF_RED;
  float *Up = Xs;   // (X is dead once the forward products are done)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 15, kq = lane >> 4;
  const int rb = blockIdx.x / Q.nparts, part = blockIdx.x % Q.nparts, r0 = rb * kRrRows, J = Q.J, c0 = wid * 16;
  const int s_lo = Q.s_lo[part], s_hi = Q.s_hi[part];
  const int f_lo = max(s_lo - 1, 1), f_hi = min(s_hi + 1, J);     // scales whose rows of c this part needs (c_0 = X itself)
  auto cplane = [&](int s) { return Cp + (s - (s_lo - 1)) * PL; };  // rows of c_s, s_lo - 1 <= s <= s_hi + 1
  auto hw_of = [](int s) { return 2 * ((1 << s) - 1); };
#ifdef LC_STAMPS
#define RR_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_rstamps[k] = clock64(); } while (0)
#else
#define RR_STAMP(k) do {} while (0)
#endif
  RR_STAMP(0);
  // X into LDS (coalesced 16-byte loads)
  for (int e = tid; e < NN / 4; e += kRrThreads) {
    const int r = e / (N / 4), c4 = (e % (N / 4)) * 4;
    const float4 v = *(const float4 *)(Q.X + (size_t)r * N + c4);
    float *d = Xs + r * XS + c4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  RR_STAMP(1);
  if (s_lo == 1) {  // c_0 rows = X rows
    float *c0p = cplane(0);
    for (int e = tid; e < kRrRows * N; e += kRrThreads) c0p[(e / N) * TS + e % N] = Xs[(r0 + e / N) * XS + e % N];
  }
  // ---- forward: rows rb of c_s = (A_s[rb, :] X) A_s^T for s = f_lo .. f_hi, kRrBatch scales per pair of barriers -------------
  for (int b0 = f_lo; b0 <= f_hi; b0 += kRrBatch) {
    const int nb = min(kRrBatch, f_hi - b0 + 1);
#pragma unroll 1
    for (int q = 0; q < nb; ++q) {  // T1[m][n] = sum_k A_s[r0 + m][k] X[k][n], k in the band of the rows: the operator is the A operand
      const int s = b0 + q;
      int klo, khi;
      rr_band<N>(r0, hw_of(s), klo, khi);
      const rr_acc acc = rr_product<N, true>(Q.A + (size_t)s * NN + (size_t)(r0 + i) * N, klo, khi, kq, C::OFF_X + c0 + i, XS, Q.dbg);
      float *T1 = Tp + q * PL;
#pragma unroll
      for (int r = 0; r < 4; ++r) T1[(4 * kq + r) * TS + c0 + i] = acc[r];
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < nb; ++q) {  // c[m][n] = sum_k T1[m][k] A_s[c0 + n][k], k in the band of the tile's columns: B operand
      const int s = b0 + q;
      int klo, khi;
      rr_band<N>(c0, hw_of(s), klo, khi);
      const rr_acc acc = rr_product<N, false>(Q.A + (size_t)s * NN + (size_t)(c0 + i) * N, klo, khi, kq, C::OFF_T + q * PL + i * TS, 1, Q.dbg);
      float *cs = cplane(s);
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[(4 * kq + r) * TS + c0 + i] = acc[r];
    }
    __syncthreads();
  }
  RR_STAMP(2);
  // ---- S rows and the values ----------------------------------------------------------------------------------------------
  float l1 = 0.f, pos = 0.f;
  auto lw_of = [&](int s, float lam, int k) { return Q.W ? lam * Q.W[(size_t)s * NN + k] : lam * Q.norms[s]; };
  auto sgn = [](float d, float lw) { return (d > 0.f) ? lw : ((d < 0.f) ? -lw : 0.f); };
  if (part == 0) {  // S_0 = q_0 - positivity sub-gradient: needs c_0 (= X) and c_1, no product; its rows go out as they are
    const float *c0p = cplane(0), *c1p = cplane(1);
    for (int e = tid; e < kRrRows * N; e += kRrThreads) {
      const int r = e / N, c = e % N, k = (r0 + r) * N + c;
      const float hv = c0p[r * TS + c], d = hv - c1p[r * TS + c], lw = lw_of(0, Q.lam_hf, k);
      float z = sgn(d, lw);
      l1 += lw * fabsf(d);
      if (Q.lam_pos != 0.f && hv < 0.f) {
        pos += -Q.lam_pos * hv;
        z -= Q.lam_pos;
      }
      Q.S0[k] = z;
    }
  }
  for (int s = s_lo; s <= s_hi; ++s) {   // S_s rows = q_s - q_{s-1} (q_J = 0) -> where T1 was
    const float *cm = cplane(s - 1), *cj = cplane(s), *cn = cplane(min(s + 1, J));
    float *Ss = Tp + (s - s_lo) * PL;
    for (int e = tid; e < kRrRows * N; e += kRrThreads) {
      const int r = e / N, c = e % N, k = (r0 + r) * N + c;
      const float vj = cj[r * TS + c];
      const float qm = sgn(cm[r * TS + c] - vj, lw_of(s - 1, s - 1 == 0 ? Q.lam_hf : Q.lam_sc, k));
      float qj = 0.f;
      if (s < J) {
        const float d = vj - cn[r * TS + c], lw = lw_of(s, Q.lam_sc, k);
        qj = sgn(d, lw);
        l1 += lw * fabsf(d);
      }
      Ss[r * TS + c] = qj - qm;
    }
  }
  __syncthreads();
  RR_STAMP(3);
  // ---- adjoint: U_s = S_s[rb rows, :] A_s for the wave's column tile, then Z += A_s[rb rows, :]^T U_s ---------------------------
  const int ns = s_hi - s_lo + 1;
#pragma unroll 1
  for (int q = 0; q < ns; ++q) {  // U[m][n] = sum_k S[m][k] A_s[k][c0 + n] = sum_k S[m][k] AT_s[c0 + n][k]: B operand from the transpose
    const int s = s_lo + q;
    int klo, khi;
    rr_band<N>(c0, hw_of(s), klo, khi);
    const rr_acc acc = rr_product<N, false>(Q.AT + (size_t)s * NN + (size_t)(c0 + i) * N, klo, khi, kq, C::OFF_T + q * PL + i * TS, 1, Q.dbg);
    float *Us = Up + q * UL;
#pragma unroll
    for (int r = 0; r < 4; ++r) Us[(4 * kq + r) * XS + c0 + i] = acc[r];
  }
  wave_lds_sync();   // (a wave reads back its own column tile of U only)
  RR_STAMP(4);
  rr_acc zacc[NT];
#pragma unroll
  for (int m = 0; m < NT; ++m) zacc[m] = (rr_acc){0.f, 0.f, 0.f, 0.f};
  {  // Z[m][n] += sum_{k < 16} A_s[r0 + k][m] U[k][n] = AT_s[m][r0 + k] U[k][n]: row tiles m inside the band of the rows
    float4 cur[NT], nxt[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) cur[m] = *(const float4 *)(Q.AT + (size_t)s_lo * NN + (size_t)(16 * m + i) * N + r0 + 4 * kq);
#pragma unroll 1
    for (int q = 0; q < ns; ++q) {
      const int s = s_lo + q, sn = min(s + 1, s_hi);
#pragma unroll
      for (int m = 0; m < NT; ++m) nxt[m] = *(const float4 *)(Q.AT + (size_t)sn * NN + (size_t)(16 * m + i) * N + r0 + 4 * kq);
      int mlo, mhi;
      rr_band<N>(r0, hw_of(s), mlo, mhi);
      const float *Us = Up + q * UL;
      float ub[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) ub[jj] = Us[(4 * kq + jj) * XS + c0 + i];
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        if (16 * m >= mlo && 16 * m < mhi) {
          const float av[4] = {cur[m].x, cur[m].y, cur[m].z, cur[m].w};
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) zacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], ub[jj], zacc[m], 0, 0, 0);
        }
      }
#pragma unroll
      for (int m = 0; m < NT; ++m) cur[m] = nxt[m];
    }
  }
  RR_STAMP(5);
  // the partial plane of this workgroup
  float *Zp = Q.Zp + (size_t)blockIdx.x * NN;
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) Zp[(size_t)(16 * m + 4 * kq + r) * N + c0 + i] = zacc[m][r];
  // values: lanes by shuffles, waves in order
  l1 = wave_sum_shfl(l1);
  pos = wave_sum_shfl(pos);
  if (lane == 0) {
    RED[wid] = l1;
    RED[8 + wid] = pos;
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < kRrThreads / 64; ++w) {
      a += RED[w];
      b += RED[8 + w];
    }
    Q.vals[blockIdx.x * 2] = a;
    Q.vals[blockIdx.x * 2 + 1] = b;
  }
  RR_STAMP(6);
}

// Stage two: greg = S_0 + sum of the partial planes (workgroup order); one more block adds the values and the tile partials of
// the point-source term (gm_pts_final_kernel's sums) into regs.  Every block ends by adding one to the completion counter
// behind a release fence: the fused update waits until the counter has reached its number for this iteration
// (done_target) instead of for one flag store - no third launch, no cross-block step inside this one.
__global__ __launch_bounds__(kGmThreads) void mreg_rows_finish_kernel(int NN, int nplanes, const float *Zp, const float *S0, float *greg,
                                                                      const float *vals, int has_pts, int pts_blocks, int M,
                                                                      const float *pts_part, const float *pts_l1, float *regs,
                                                                      unsigned int *done_counter) {
  const int nimg = NN / kGmThreads, tid = threadIdx.x, lane = tid & 63;
  if ((int)blockIdx.x < nimg) {
    const int k = blockIdx.x * kGmThreads + tid;
    // (all planes of the pixel requested before the first addition: one round trip, not one per plane; added in plane order)
    constexpr int MAXP = (128 / kRrRows) * kRrMaxParts;
    float v[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) v[p] = Zp[(size_t)min(p, nplanes - 1) * NN + k];
    float g = S0[k];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) g += (p < nplanes) ? v[p] : 0.f;
    greg[k] = g;
  } else if (tid < 64) {
    float a = 0.f, b = 0.f;
    for (int p = lane; p < nplanes; p += 64) {
      a += vals[2 * p];
      b += vals[2 * p + 1];
    }
    a = wave_sum_shfl(a);
    b = wave_sum_shfl(b);
    if (lane == 0) {
      regs[0] = a;
      regs[1] = b;
    }
    if (has_pts) {
      float c = 0.f;
      for (int p = lane; p < pts_blocks; p += 64) c += pts_l1[p];
      c = wave_sum_shfl(c);
      if (lane == 0) regs[2] = c;
      for (int t = 0; t < 3 * M; ++t) {
        float acc = 0.f;
        for (int blk = lane; blk < pts_blocks; blk += 64) acc += pts_part[(size_t)blk * 3 * kMaxSources + t];
        acc = wave_sum_shfl(acc);
        if (lane == 0) regs[4 + t] = acc;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's stores have left before the block signals for them
  __syncthreads();
  if (tid == 0 && done_counter) {
    __threadfence();
    __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace lc
