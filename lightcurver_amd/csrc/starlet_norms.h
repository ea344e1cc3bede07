// Starlet scale norms: the l1 weights STARRED falls back to when no noise-propagated weight map is given
// ("lambda is not normalized", docs/example_starred_notebooks/example_roi_modelling.ipynb:302).
// psi_j = p_j (x) p_j - p_{j+1} (x) p_{j+1} with p_j the 1-D cascade of a dirac at the zero-lag index.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "lc_common.h"

namespace lc {

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


// norms[j] = ||psi_j||_2 (j <= J, last = the coarse atom)
static inline void starlet_scale_norms(int N, int J, std::vector<float> &norms) {
  std::vector<std::vector<double>> p;
  starlet_atoms_1d(N, J, p);
  norms.assign(J + 1, 0.f);
  for (int j = 0; j <= J; ++j) {
    double s_pp = 0, s_qq = 0, s_pq = 0;
    for (int i = 0; i < N; ++i) {
      const double pj = p[j][i], qj = (j < J) ? p[j + 1][i] : 0.0;
      s_pp += pj * pj;
      s_qq += qj * qj;
      s_pq += pj * qj;
    }
    norms[j] = (float)std::sqrt(std::max(s_pp * s_pp - 2.0 * s_pq * s_pq + s_qq * s_qq, 0.0));
  }
}

}  // namespace lc
