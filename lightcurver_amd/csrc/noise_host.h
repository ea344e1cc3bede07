// One-time noise propagation into starlet space (host, double precision, threaded):
//   W_j(x)^2 = sum_k ( up0(w_k) (*) kappa_{k,j}^2 )(x),   kappa_{k,j} = starlet scale j of the response r_k of the
//   gradient image to a unit of whitened noise in the central data pixel of contributor k (epoch / star).
// Restates starred.utils.noise_utils.propagate_noise(method='SLIT', likelihood_type='chi2') as frozen in
// DESIGN.md "SPEC" (reference call sites: lightcurver/processes/star_photometry.py:108-110,
// roi_modelling.py:299-301, and the propagate_noise call inside build_psf, psf_modelling.py:164-171).
#pragma once
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace lc {

// plain complex pair: std::complex's operator* goes through the NaN-recovering runtime routine on the host
// compiler, several times the cost of the four multiplications below
struct cd {
  double re, im;
  cd() : re(0), im(0) {}
  cd(double r, double i = 0.0) : re(r), im(i) {}
  double real() const { return re; }
  double imag() const { return im; }
};
inline cd operator+(cd a, cd b) { return cd(a.re + b.re, a.im + b.im); }
inline cd operator-(cd a, cd b) { return cd(a.re - b.re, a.im - b.im); }
inline cd operator*(cd a, cd b) { return cd(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
inline cd &operator+=(cd &a, cd b) {
  a.re += b.re;
  a.im += b.im;
  return a;
}
inline cd conj(cd a) { return cd(a.re, -a.im); }

// twiddles exp(-2 pi i k / L), k < L / 2, computed once per length and thread
inline const std::vector<cd> &host_twiddles(int L) {
  thread_local std::vector<cd> tw;
  thread_local int twL = 0;
  if (twL != L) {
    tw.resize(L / 2);
    for (int k = 0; k < L / 2; ++k) tw[k] = cd(std::cos(-2.0 * M_PI * k / L), std::sin(-2.0 * M_PI * k / L));
    twL = L;
  }
  return tw;
}

inline void host_fft1d(cd *x, int L, bool inv) {
  if (L & (L - 1)) {  // not a power of two (the device uses 3 * 2^m lengths too): plain DFT, cross-check paths only
    std::vector<cd> y(L);
    for (int k = 0; k < L; ++k) {
      cd acc(0, 0);
      for (int n = 0; n < L; ++n) {
        const double ang = (inv ? 2.0 : -2.0) * M_PI * (double)(((long long)k * n) % L) / L;
        acc += x[n] * cd(std::cos(ang), std::sin(ang));
      }
      y[k] = acc;
    }
    for (int k = 0; k < L; ++k) x[k] = y[k];
    return;
  }
  for (int i = 1, j = 0; i < L; ++i) {
    int bit = L >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(x[i], x[j]);
  }
  const std::vector<cd> &tw = host_twiddles(L);
  for (int len = 2; len <= L; len <<= 1) {
    const int step = L / len, half = len / 2;
    for (int i = 0; i < L; i += len) {
      for (int k = 0; k < half; ++k) {
        cd w = tw[k * step];
        if (inv) w.im = -w.im;
        const cd u = x[i + k], v = x[i + k + half] * w;
        x[i + k] = u + v;
        x[i + k + half] = u - v;
      }
    }
  }
}
// 2-D FFT (unnormalised) of an L x L array whose non-zero rows are [r0, r0 + nr)
inline void host_fft2d(std::vector<cd> &a, int L, int r0, int nr, bool inv) {
  if (!inv)
    for (int r = r0; r < r0 + nr; ++r) host_fft1d(&a[(size_t)r * L], L, false);
  // columns, eight at a time through a transposed tile so that the strided accesses stay in cache lines
  constexpr int TB = 8;
  std::vector<cd> col((size_t)TB * L);
  for (int c0 = 0; c0 < L; c0 += TB) {
    const int nb = std::min(TB, L - c0);
    for (int r = 0; r < L; ++r)
      for (int b = 0; b < nb; ++b) col[(size_t)b * L + r] = a[(size_t)r * L + c0 + b];
    for (int b = 0; b < nb; ++b) host_fft1d(&col[(size_t)b * L], L, inv);
    for (int r = 0; r < L; ++r)
      for (int b = 0; b < nb; ++b) a[(size_t)r * L + c0 + b] = col[(size_t)b * L + r];
  }
  if (inv)
    for (int r = 0; r < L; ++r) host_fft1d(&a[(size_t)r * L], L, true);
}
inline int host_fft_length(int N) {  // smallest power of two that keeps the 'same' window alias free
  const int c = (N - 1) / 2;
  int L = 1;
  while (L < 2 * N - 1 - c) L <<= 1;
  return L;
}

// 2-D first-generation starlet, edge replicating: out[j] (j < J) detail scales, out[J] coarse.
inline void host_starlet(const std::vector<double> &img, int N, int J, std::vector<std::vector<double>> &out) {
  const double b3[5] = {1. / 16, 4. / 16, 6. / 16, 4. / 16, 1. / 16};
  out.assign(J + 1, std::vector<double>((size_t)N * N));
  std::vector<double> c = img, r((size_t)N * N), cn((size_t)N * N);
  for (int j = 0; j < J; ++j) {
    const int d = 1 << j;
    for (int u = 0; u < N; ++u)
      for (int v = 0; v < N; ++v) {
        double acc = 0;
        for (int t = -2; t <= 2; ++t) acc += b3[t + 2] * c[(size_t)u * N + std::min(std::max(v + t * d, 0), N - 1)];
        r[(size_t)u * N + v] = acc;
      }
    for (int u = 0; u < N; ++u)
      for (int v = 0; v < N; ++v) {
        double acc = 0;
        for (int t = -2; t <= 2; ++t) acc += b3[t + 2] * r[(size_t)std::min(std::max(u + t * d, 0), N - 1) * N + v];
        cn[(size_t)u * N + v] = acc;
      }
    for (size_t i = 0; i < c.size(); ++i) out[j][i] = c[i] - cn[i];
    c = cn;
  }
  out[J] = c;
}

struct NoiseAccumulator {
  int N, n, ss, J, L, c, shift;
  std::vector<std::vector<cd>> acc;  // [J+1][L*L], Fourier space
  NoiseAccumulator(int N_, int ss_) : N(N_), n(N_ / ss_), ss(ss_) {
    J = 0;
    for (int m = N; m > 1; m >>= 1) ++J;
    L = host_fft_length(N);
    c = (N - 1) / 2;
    shift = ss * (n / 2) - c;
    acc.assign(J + 1, std::vector<cd>((size_t)L * L, cd(0, 0)));
  }
  // r: N*N impulse response, w: n*n inverse variances of this contributor
  void add(const std::vector<double> &r, const float *w) {
    std::vector<std::vector<double>> kap;
    host_starlet(r, N, J, kap);
    std::vector<cd> A((size_t)L * L, cd(0, 0)), B((size_t)L * L);
    for (int i = 0; i < n; ++i)
      for (int jx = 0; jx < n; ++jx) {
        const float wv = w[(size_t)i * n + jx];
        A[(size_t)(ss * i) * L + ss * jx] = (std::isfinite(wv) && wv > 0.f) ? (double)wv : 0.0;
      }
    host_fft2d(A, L, 0, N, false);
    // two real images per complex transform: Z = FFT(k_j^2 + i k_{j+1}^2);  by Hermitian symmetry
    //   FFT(k_j^2)[q] = (Z[q] + conj Z[-q]) / 2,   FFT(k_{j+1}^2)[q] = (Z[q] - conj Z[-q]) / (2 i)
    for (int j = 0; j <= J; j += 2) {
      const bool two = (j + 1 <= J);
      std::fill(B.begin(), B.end(), cd(0, 0));
      for (int u = 0; u < N; ++u)
        for (int v = 0; v < N; ++v) {
          const int su = u + shift, sv = v + shift;
          if (su < 0 || su >= N || sv < 0 || sv >= N) continue;
          const double k0 = kap[j][(size_t)su * N + sv];
          const double k1 = two ? kap[j + 1][(size_t)su * N + sv] : 0.0;
          B[(size_t)u * L + v] = cd(k0 * k0, k1 * k1);
        }
      host_fft2d(B, L, 0, N, false);
      for (int r = 0; r < L; ++r) {
        const int rm = (L - r) & (L - 1);
        for (int q = 0; q < L; ++q) {
          const int qm = (L - q) & (L - 1);
          const cd z = B[(size_t)r * L + q], zc = conj(B[(size_t)rm * L + qm]);
          const cd a = A[(size_t)r * L + q];
          acc[j][(size_t)r * L + q] += a * cd(0.5 * (z.re + zc.re), 0.5 * (z.im + zc.im));
          if (two) acc[j + 1][(size_t)r * L + q] += a * cd(0.5 * (z.im - zc.im), -0.5 * (z.re - zc.re));
        }
      }
    }
  }
  void merge(const NoiseAccumulator &o) {
    for (int j = 0; j <= J; ++j)
      for (size_t i = 0; i < acc[j].size(); ++i) acc[j][i] += o.acc[j][i];
  }
  // W: [(J+1)][N*N]
  void finalize(float *W) {
    const double sc = 1.0 / ((double)L * L);
    // the results are real, so two scales share one inverse transform (real part / imaginary part)
    for (int j = 0; j <= J; j += 2) {
      const bool two = (j + 1 <= J);
      std::vector<cd> &z = acc[j];
      if (two)
        for (size_t i = 0; i < z.size(); ++i) z[i] = cd(z[i].re - acc[j + 1][i].im, z[i].im + acc[j + 1][i].re);
      host_fft2d(z, L, 0, L, true);
      for (int u = 0; u < N; ++u)
        for (int v = 0; v < N; ++v) {
          const cd val = z[(size_t)(u + c) * L + (v + c)];
          W[((size_t)j * N + u) * N + v] = (float)std::sqrt(std::max(val.re * sc, 0.0));
          if (two) W[((size_t)(j + 1) * N + u) * N + v] = (float)std::sqrt(std::max(val.im * sc, 0.0));
        }
    }
  }
};

inline int noise_threads(int work_items) {
  int t = (int)std::thread::hardware_concurrency();
  if (t <= 0) t = 4;
  return std::max(1, std::min({t, 16, work_items}));
}

}  // namespace lc
