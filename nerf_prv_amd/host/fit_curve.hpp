// fit_curve.hpp -- the stopping criterion of the planner: PSNR-vs-#views curve fit and the
// "gap k%" / "gradient g" view-count labels.  Restates Origin_scripts/NeRF_fit_curve.cpp
// (Fit_ShapeNet :56-212): model = Origin's built-in LognormalCDF
//     y = y0 + A * Phi((ln x - xc) / w),
// fitted there by OriginPro's proprietary ODR solver (NLFitSession, :119-140).  Here: a plain
// Levenberg-Marquardt least-squares fit (x are exact integers, so ODR and OLS share the
// minimiser up to the x-error weighting).  Parameter values are therefore NOT comparable with
// an Origin run; the labels derived from the fitted curve are.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

namespace prvhost {

struct LognormalCDF {
  double y0 = 0, A = 1, xc = 1, w = 1;
  double operator()(double x) const { return y0 + A * 0.5 * std::erfc(-((std::log(x) - xc) / w) / std::sqrt(2.0)); }
};

struct FitResult {
  LognormalCDF f;
  bool fit_converged = false; // the solver's own outcome
  bool converged = false;     // label.txt's "Converged": solver ok AND no data point above max_psnr (:143-151)
  int iterations = 0;
  double rss = 0;
};

inline FitResult fit_lognormal_cdf(const std::vector<double>& x, const std::vector<double>& y, double max_psnr) {
  FitResult r;
  const size_t n = x.size();
  if (n < 4 || y.size() != n) return r;
  const double ymin = *std::min_element(y.begin(), y.end()), ymax = *std::max_element(y.begin(), y.end());
  // p = (y0, A, xc, u) with w = exp(u) > 0
  double p[4] = {ymin - 0.05 * (ymax - ymin + 1e-9), 1.1 * (ymax - ymin) + 1e-6, std::log(x[n / 4] > 0 ? x[n / 4] : 1.0), 0.0};
  auto eval = [&](const double* q, std::vector<double>* res, std::vector<double>* J) {
    const double w = std::exp(q[3]);
    double rss = 0;
    for (size_t i = 0; i < n; i++) {
      const double z = (std::log(x[i]) - q[2]) / w;
      const double Phi = 0.5 * std::erfc(-z / std::sqrt(2.0));
      const double phi = std::exp(-0.5 * z * z) / std::sqrt(2.0 * std::acos(-1.0));
      const double ri = y[i] - (q[0] + q[1] * Phi);
      rss += ri * ri;
      if (res) (*res)[i] = ri;
      if (J) { // d model / d p
        (*J)[i * 4 + 0] = 1.0;
        (*J)[i * 4 + 1] = Phi;
        (*J)[i * 4 + 2] = -q[1] * phi / w;
        (*J)[i * 4 + 3] = -q[1] * phi * z; // d/du with w = e^u: (-A phi z / w) * w
      }
    }
    return rss;
  };
  std::vector<double> res(n), J(n * 4);
  double lambda = 1e-3, rss = eval(p, &res, &J);
  for (r.iterations = 0; r.iterations < 200; r.iterations++) {
    double JtJ[16] = {0}, Jtr[4] = {0};
    for (size_t i = 0; i < n; i++)
      for (int a = 0; a < 4; a++) {
        Jtr[a] += J[i * 4 + a] * res[i];
        for (int b = 0; b < 4; b++) JtJ[a * 4 + b] += J[i * 4 + a] * J[i * 4 + b];
      }
    bool improved = false;
    for (int tries = 0; tries < 30 && !improved; tries++) {
      double M[4][5];
      for (int a = 0; a < 4; a++) {
        for (int b = 0; b < 4; b++) M[a][b] = JtJ[a * 4 + b] + (a == b ? lambda * (JtJ[a * 4 + a] + 1e-12) : 0.0);
        M[a][4] = Jtr[a];
      }
      bool singular = false;
      for (int c = 0; c < 4 && !singular; c++) { // Gauss-Jordan with partial pivoting
        int piv = c;
        for (int q = c + 1; q < 4; q++)
          if (std::fabs(M[q][c]) > std::fabs(M[piv][c])) piv = q;
        if (std::fabs(M[piv][c]) < 1e-300) { singular = true; break; }
        for (int k = 0; k < 5; k++) std::swap(M[c][k], M[piv][k]);
        for (int q = 0; q < 4; q++)
          if (q != c) {
            const double f = M[q][c] / M[c][c];
            for (int k = c; k < 5; k++) M[q][k] -= f * M[c][k];
          }
      }
      if (singular) { lambda *= 10; continue; }
      double cand[4];
      for (int a = 0; a < 4; a++) cand[a] = p[a] + M[a][4] / M[a][a];
      cand[3] = std::min(std::max(cand[3], -8.0), 8.0);
      const double rss_new = eval(cand, nullptr, nullptr);
      if (std::isfinite(rss_new) && rss_new < rss) {
        const double rel = (rss - rss_new) / (rss + 1e-300);
        for (int a = 0; a < 4; a++) p[a] = cand[a];
        rss = eval(p, &res, &J);
        lambda = std::max(lambda * 0.3, 1e-12);
        improved = true;
        if (rel < 1e-12) r.fit_converged = true;
      } else {
        lambda *= 10;
      }
    }
    if (!improved) { r.fit_converged = true; break; } // no descent direction left: at a minimum
    if (r.fit_converged) break;
  }
  r.f.y0 = p[0]; r.f.A = p[1]; r.f.xc = p[2]; r.f.w = std::exp(p[3]);
  r.rss = rss;
  // Origin counts "converged" and "max iterations reached" both as usable (:143-146)
  r.converged = std::isfinite(rss);
  for (size_t i = 0; i < n; i++)
    if (max_psnr < y[i]) r.converged = false; // :149-151
  return r;
}

struct Labels {
  std::vector<double> fit_y; // x = 3..100 (:176-183)
  int gap[11];               // gap k%: first x with FitY/max_psnr >= 1 - 0.01k, else -1 (:186-195)
  int gradient[20];          // gradient g = 0.01..0.20: first x (from the 2nd point) with FitY[j]-FitY[j-1] <= g (:197-206)
};

inline Labels make_labels(const LognormalCDF& f, double max_psnr) {
  Labels L;
  for (int j = 3; j <= 100; j++) L.fit_y.push_back(f((double)j));
  const int n = (int)L.fit_y.size();
  for (int gap = 0; gap <= 10; gap++) {
    int j = 0;
    for (; j < n; j++)
      if (L.fit_y[j] / max_psnr >= (1.0 - 0.01 * gap)) break;
    L.gap[gap] = j == n ? -1 : j + 3;
  }
  int gi = 0;
  for (double gradient = 0.01; gradient <= 0.20 + 1e-6; gradient += 0.01, gi++) { // same float loop as the reference
    int j = 1;
    for (; j < n; j++)
      if (L.fit_y[j] - L.fit_y[j - 1] <= gradient) break;
    if (gi < 20) L.gradient[gi] = j == n ? -1 : j + 3;
  }
  return L;
}

// label.txt in the reference's format (:162-206)
inline bool write_label_file(const std::string& path, const FitResult& r, double max_psnr) {
  FILE* fp = fopen(path.c_str(), "w+");
  if (!fp) return false;
  const Labels L = make_labels(r.f, max_psnr);
  fprintf(fp, r.converged ? "Converged 1\n" : "Converged 0\n");
  for (size_t j = 0; j < L.fit_y.size(); j++) fprintf(fp, "%d %f\n", (int)j + 3, L.fit_y[j]);
  for (int gap = 0; gap <= 10; gap++) fprintf(fp, "gap %d%% %d\n", gap, L.gap[gap]);
  int gi = 0;
  for (double gradient = 0.01; gradient <= 0.20 + 1e-6; gradient += 0.01, gi++) fprintf(fp, "gradient %.2f %d\n", gradient, L.gradient[std::min(gi, 19)]);
  fclose(fp);
  return true;
}

} // namespace prvhost
