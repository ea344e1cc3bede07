// Host-side bounded L-BFGS over a batch of independent small problems whose loss/gradient are
// evaluated on the device in ONE launch per trial point set.
//
// The reference runs scipy's L-BFGS-B on the host with one host<->device round trip per function
// evaluation and one problem at a time (STARRED Optimizer(method='l-bfgs-b'): reference call sites
// lightcurver/processes/psf_modelling.py:164-171 stage A, roi_modelling.py:278-280,
// utilities/starred_utilities.py:33-34).  Here every problem of the batch advances in lock step, so
// F frames cost the same number of round trips as one.  Iterate-level parity with scipy is not a
// goal (SURVEY section 7 "hard parts"); the optimum is.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <vector>

namespace lc {

struct LbfgsResult {
  std::vector<double> f;    // final loss per problem
  std::vector<int> iters;   // accepted iterations per problem
  int evaluations = 0;      // batched evaluations performed
};

// eval(X [nb*D], F [nb], G [nb*D]) evaluates all problems at X.
using BatchEval = std::function<int(const std::vector<double> &, std::vector<double> &, std::vector<double> &)>;

inline int batched_lbfgs(int nb, int D, std::vector<double> &x, const std::vector<double> &lo,
                         const std::vector<double> &hi, int maxiter, const BatchEval &eval,
                         LbfgsResult &out, int mem = 10, double gtol = 1e-7, double ftol = 1e-12) {
  const double c1 = 1e-4;
  const int max_ls = 25;
  auto clip = [&](int i, double v) { return std::min(std::max(v, lo[i]), hi[i]); };
  std::vector<double> f(nb), g((size_t)nb * D), xt((size_t)nb * D), ft(nb), gt((size_t)nb * D);
  std::vector<double> d((size_t)nb * D), alpha(nb, 1.0), gd(nb, 0.0);
  std::vector<int> state(nb, 0);  // 0 = need direction, 1 = in line search, 2 = done
  std::vector<int> ls_count(nb, 0), iters(nb, 0);
  std::vector<std::vector<std::vector<double>>> S(nb), Y(nb);
  std::vector<std::vector<double>> RHO(nb);
  for (size_t i = 0; i < x.size(); ++i) x[i] = clip((int)i, x[i]);
  int rc = eval(x, f, g);
  if (rc) return rc;
  out.evaluations = 1;
  for (int b = 0; b < nb; ++b)
    if (!std::isfinite(f[b])) state[b] = 2;

  auto active = [&](int b, int i) {
    const int k = b * D + i;
    return (x[k] <= lo[k] && g[k] > 0) || (x[k] >= hi[k] && g[k] < 0);
  };

  while (true) {
    bool any = false;
    for (int b = 0; b < nb; ++b) {
      if (state[b] == 2) continue;
      if (state[b] == 0) {
        if (iters[b] >= maxiter) {
          state[b] = 2;
          continue;
        }
        // projected gradient and convergence test
        double pgmax = 0;
        std::vector<double> q(D);
        for (int i = 0; i < D; ++i) {
          q[i] = active(b, i) ? 0.0 : g[b * D + i];
          pgmax = std::max(pgmax, std::fabs(q[i]));
        }
        if (pgmax < gtol * std::max(1.0, std::fabs(f[b]))) {
          state[b] = 2;
          continue;
        }
        // two-loop recursion on the free variables
        const int m = (int)S[b].size();
        std::vector<double> a(m);
        for (int k = m - 1; k >= 0; --k) {
          double dot = 0;
          for (int i = 0; i < D; ++i) dot += S[b][k][i] * q[i];
          a[k] = RHO[b][k] * dot;
          for (int i = 0; i < D; ++i) q[i] -= a[k] * Y[b][k][i];
        }
        double gamma = 1.0;
        if (m > 0) {
          double sy = 0, yy = 0;
          for (int i = 0; i < D; ++i) {
            sy += S[b][m - 1][i] * Y[b][m - 1][i];
            yy += Y[b][m - 1][i] * Y[b][m - 1][i];
          }
          gamma = sy / yy;
        }
        for (int i = 0; i < D; ++i) q[i] *= gamma;
        for (int k = 0; k < m; ++k) {
          double dot = 0;
          for (int i = 0; i < D; ++i) dot += Y[b][k][i] * q[i];
          const double beta = RHO[b][k] * dot;
          for (int i = 0; i < D; ++i) q[i] += S[b][k][i] * (a[k] - beta);
        }
        double gdot = 0;
        for (int i = 0; i < D; ++i) {
          d[b * D + i] = active(b, i) ? 0.0 : -q[i];
          gdot += d[b * D + i] * g[b * D + i];
        }
        if (!(gdot < 0)) {  // not a descent direction: restart from steepest descent
          S[b].clear();
          Y[b].clear();
          RHO[b].clear();
          gdot = 0;
          for (int i = 0; i < D; ++i) {
            d[b * D + i] = active(b, i) ? 0.0 : -g[b * D + i];
            gdot += d[b * D + i] * g[b * D + i];
          }
        }
        gd[b] = gdot;
        if (S[b].empty()) {
          double nrm = 0;
          for (int i = 0; i < D; ++i) nrm += d[b * D + i] * d[b * D + i];
          alpha[b] = std::min(1.0, 1.0 / std::sqrt(std::max(nrm, 1e-300)));
        } else {
          alpha[b] = 1.0;
        }
        ls_count[b] = 0;
        state[b] = 1;
      }
      for (int i = 0; i < D; ++i) xt[b * D + i] = clip(b * D + i, x[b * D + i] + alpha[b] * d[b * D + i]);
      any = true;
    }
    if (!any) break;
    for (int b = 0; b < nb; ++b)
      if (state[b] == 2)
        for (int i = 0; i < D; ++i) xt[b * D + i] = x[b * D + i];
    rc = eval(xt, ft, gt);
    if (rc) return rc;
    ++out.evaluations;
    for (int b = 0; b < nb; ++b) {
      if (state[b] != 1) continue;
      double dec = 0;  // g . (x_new - x): projected step
      for (int i = 0; i < D; ++i) dec += g[b * D + i] * (xt[b * D + i] - x[b * D + i]);
      if (std::isfinite(ft[b]) && ft[b] <= f[b] + c1 * dec) {
        std::vector<double> s(D), y(D);
        double sy = 0, ss = 0, yy = 0;
        for (int i = 0; i < D; ++i) {
          s[i] = xt[b * D + i] - x[b * D + i];
          y[i] = gt[b * D + i] - g[b * D + i];
          sy += s[i] * y[i];
          ss += s[i] * s[i];
          yy += y[i] * y[i];
        }
        const double fold = f[b];
        for (int i = 0; i < D; ++i) {
          x[b * D + i] = xt[b * D + i];
          g[b * D + i] = gt[b * D + i];
        }
        f[b] = ft[b];
        ++iters[b];
        if (sy > 1e-10 * std::sqrt(ss * yy) && sy > 0) {
          if ((int)S[b].size() == mem) {
            S[b].erase(S[b].begin());
            Y[b].erase(Y[b].begin());
            RHO[b].erase(RHO[b].begin());
          }
          S[b].push_back(s);
          Y[b].push_back(y);
          RHO[b].push_back(1.0 / sy);
        }
        state[b] = 0;
        if (std::fabs(fold - f[b]) <= ftol * std::max({std::fabs(fold), std::fabs(f[b]), 1.0})) state[b] = 2;
      } else {
        alpha[b] *= 0.5;
        if (++ls_count[b] >= max_ls) state[b] = 2;  // line search failed: keep the last accepted point
      }
    }
  }
  out.f = f;
  out.iters = iters;
  return 0;
}

}  // namespace lc
