// Noise propagation into starlet space, shared by the PSF fit and the joint fit:
//   W_j = sqrt( conv_same(V, psi_j^2) ),   psi_j = 2-D starlet atom of scale j (dirac at the zero-lag index)
// psi_j = p_j (x) p_j - p_{j+1} (x) p_{j+1} with p_j the 1-D cascade of the dirac, so
// psi_j^2 = p_j^2 (x) p_j^2 - 2 (p_j p_{j+1}) (x) (p_j p_{j+1}) + p_{j+1}^2 (x) p_{j+1}^2: three separable kernels.
// Restates starred.utils.noise_utils.propagate_noise(method='SLIT') as frozen in DESIGN.md "SPEC"
// (reference call sites: lightcurver/processes/star_photometry.py:108-110, roi_modelling.py:299-301).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "lc_common.h"

namespace lc {

// W_j = sqrt( sum_{term} c_term * (k_term (x) k_term) (*) V ), 'same' with zero lag at (N-1)//2.
// atoms: [J+1][3][N] 1-D factors; coefficients {+1, -2, +1}.  One block per (frame, scale).
static __global__ void starlet_noise_w_kernel(int N, int J, const float *V, const float *atoms, float *W, float *tmp) {
  const int f = blockIdx.x, j = blockIdx.y;
  const int c = (N - 1) / 2;
  const float *Vf = V + (size_t)f * N * N;
  float *t = tmp + ((size_t)f * J + j) * N * N;
  float *Wf = W + ((size_t)f * J + j) * N * N;
  const float coef[3] = {1.f, -2.f, 1.f};
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) Wf[i] = 0.f;
  for (int term = 0; term < 3; ++term) {
    const float *k = atoms + ((size_t)j * 3 + term) * N;
    __syncthreads();
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) {  // rows
      const int u = i / N, v = i % N;
      double acc = 0;
      for (int vp = 0; vp < N; ++vp) {
        const int kk = v - vp + c;
        if (kk >= 0 && kk < N) acc += (double)Vf[u * N + vp] * k[kk];
      }
      t[i] = (float)acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) {  // columns
      const int u = i / N, v = i % N;
      double acc = 0;
      for (int upp = 0; upp < N; ++upp) {
        const int kk = u - upp + c;
        if (kk >= 0 && kk < N) acc += (double)t[upp * N + v] * k[kk];
      }
      Wf[i] += coef[term] * (float)acc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) Wf[i] = sqrtf(fmaxf(Wf[i], 0.f));
}

// 1-D starlet cascade of a dirac at the zero-lag index with edge replication: p_j = A_j delta_c.
static inline void starlet_atoms_1d(int N, int J, std::vector<std::vector<double>> &p) {
  const double b3[5] = {1. / 16, 4. / 16, 6. / 16, 4. / 16, 1. / 16};
  p.assign(J + 1, std::vector<double>(N, 0.0));
  p[0][(N - 1) / 2] = 1.0;
  for (int j = 0; j < J; ++j) {
    const int d = 1 << j;
    for (int i = 0; i < N; ++i) {
      double acc = 0;
      for (int t = -2; t <= 2; ++t) acc += b3[t + 2] * p[j][std::min(std::max(i + t * d, 0), N - 1)];
      p[j + 1][i] = acc;
    }
  }
}


// norms[j] = ||psi_j||_2 (j <= J, last = coarse atom), atoms[j][3][N] = p_j^2, p_j p_{j+1}, p_{j+1}^2.
static inline void starlet_noise_tables(int N, int J, std::vector<float> &norms, std::vector<float> &atoms) {
  std::vector<std::vector<double>> p;
  starlet_atoms_1d(N, J, p);
  norms.assign(J + 1, 0.f);
  atoms.assign((size_t)(J + 1) * 3 * N, 0.f);
  for (int j = 0; j <= J; ++j) {
    double s_pp = 0, s_qq = 0, s_pq = 0;
    for (int i = 0; i < N; ++i) {
      const double pj = p[j][i], qj = (j < J) ? p[j + 1][i] : 0.0;
      s_pp += pj * pj;
      s_qq += qj * qj;
      s_pq += pj * qj;
      atoms[((size_t)j * 3 + 0) * N + i] = (float)(pj * pj);
      atoms[((size_t)j * 3 + 1) * N + i] = (float)(pj * qj);
      atoms[((size_t)j * 3 + 2) * N + i] = (float)(qj * qj);
    }
    norms[j] = (float)std::sqrt(std::max(s_pp * s_pp - 2.0 * s_pq * s_pq + s_qq * s_qq, 0.0));
  }
}

}  // namespace lc
