// path_planner.hpp -- movement cost between views and the visiting order of a view set.
// Restates get_local_path (View_Space.hpp:206-305) literally and replaces Global_Path_Planner
// (main.cpp:398-594, a Gurobi MILP with lazy sub-tour cuts) by an exact Held-Karp dynamic
// programme for up to 20 views and nearest-neighbour + 2-opt/Or-opt beyond (flagged inexact).
// Same problem statement: shortest Hamiltonian PATH from now_view_id (free or fixed end): the
// reference models it as a tour through a zero-cost copy node (main.cpp:428-437, 488-490).
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <utility>
#include <vector>

#include "View_Space.hpp"

namespace prvhost {

enum { ErrorPath = -2, WrongPath = -1, LinePath = 0, CirclePath = 1 }; // View_Space.hpp:201-204

inline double pow2(double x) { return x * x; }

// straight segment M->N, or M->P + arc PQ on the sphere (O, r) + Q->N when the segment crosses it
inline std::pair<int, double> get_local_path(const Vec3& M, const Vec3& N, const Vec3& O, double r) {
  const double x1 = M.x, y1 = M.y, z1 = M.z, x2 = N.x, y2 = N.y, z2 = N.z, x0 = O.x, y0 = O.y, z0 = O.z;
  const double a = pow2(x2 - x1) + pow2(y2 - y1) + pow2(z2 - z1);
  const double b = 2.0 * ((x2 - x1) * (x1 - x0) + (y2 - y1) * (y1 - y0) + (z2 - z1) * (z1 - z0));
  const double c = pow2(x1 - x0) + pow2(y1 - y0) + pow2(z1 - z0) - pow2(r);
  const double delta = pow2(b) - 4.0 * a * c;
  if (delta <= 0) return {LinePath, (N - M).norm()}; // :218-223
  double t3 = (-b - std::sqrt(delta)) / (2.0 * a), t4 = (-b + std::sqrt(delta)) / (2.0 * a);
  if ((t3 < 0 || t3 > 1) && (t4 < 0 || t4 > 1)) return {LinePath, (N - M).norm()}; // :228-233
  if ((t3 < 0 || t3 > 1) || (t4 < 0 || t4 > 1)) return {WrongPath, 1e10};         // start or end inside: :234-237
  if (t3 > t4) std::swap(t3, t4);
  const double x3 = (x2 - x1) * t3 + x1, y3 = (y2 - y1) * t3 + y1, z3 = (z2 - z1) * t3 + z1;
  const double x4 = (x2 - x1) * t4 + x1, y4 = (y2 - y1) * t4 + y1, z4 = (z2 - z1) * t4 + z1;
  const Vec3 P(x3, y3, z3), Q(x4, y4, z4);
  const double X1 = x3 - x0, X2 = x4 - x0, Y1 = y3 - y0, Y2 = y4 - y0, Z1 = z3 - z0, Z2 = z4 - z0;
  const double A = Y1 * Z2 - Y2 * Z1, B = Z1 * X2 - Z2 * X1, C = X1 * Y2 - X2 * Y1; // plane MON :255-257
  const double two_pi = 2.0 * std::acos(-1.0);
  auto angle = [&](double x, double y, double z) { // :262-276, same for P and Q
    const double sin_t = -(z - z0) / r * std::sqrt(pow2(A) + pow2(B) + pow2(C)) / std::sqrt(pow2(A) + pow2(B));
    double t = std::asin(sin_t);
    if (t < 0) t += two_pi;
    if (t >= two_pi) t -= two_pi;
    const double xt = x0 + r * B / std::sqrt(pow2(A) + pow2(B)) * std::cos(t) +
                      r * A * C / std::sqrt(pow2(A) + pow2(B)) / std::sqrt(pow2(A) + pow2(B) + pow2(C)) * std::sin(t);
    const double yt = y0 - r * A / std::sqrt(pow2(A) + pow2(B)) * std::cos(t) +
                      r * B * C / std::sqrt(pow2(A) + pow2(B)) / std::sqrt(pow2(A) + pow2(B) + pow2(C)) * std::sin(t);
    if (std::fabs(x - xt) > 1e-6 || std::fabs(y - yt) > 1e-6) {
      t = std::acos(-1.0) - t;
      if (t < 0) t += two_pi;
      if (t >= two_pi) t -= two_pi;
    }
    return t;
  };
  const double theta3 = angle(x3, y3, z3), theta4 = angle(x4, y4, z4);
  const double L = std::fabs(theta3 - theta4) * r; // :299 (as written: not the shorter arc when > pi)
  return {CirclePath, (M - P).norm() + L + (Q - N).norm()};
}

class Global_Path_Planner {
public:
  int now_view_id, end_view_id;
  bool solved = false, exact = false;
  int n = 0; // number of views on the path
  std::vector<int> labels;
  std::vector<std::vector<double>> graph;
  double total_shortest = -1;
  std::vector<int> global_path; // indices into labels

  // views / view_set_label / now_view_id / end_view_id as in main.cpp:415
  Global_Path_Planner(const std::vector<View>& views, const std::vector<int>& view_set_label, int _now_view_id,
                      const Vec3& obstacle_center, double obstacle_radius, int _end_view_id = -1)
      : now_view_id(_now_view_id), end_view_id(_end_view_id), labels(view_set_label) {
    n = (int)labels.size();
    graph.assign(n, std::vector<double>(n, 0.0));
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        if (i == j) continue;
        const auto lp = get_local_path(views[labels[i]].init_pos, views[labels[j]].init_pos, obstacle_center, obstacle_radius);
        graph[i][j] = lp.first < 0 ? 1e10 : lp.second; // :446-450
      }
  }

  double solve() {
    int s = -1, e = -1;
    for (int i = 0; i < n; i++) {
      if (labels[i] == now_view_id) s = i;
      if (labels[i] == end_view_id) e = i;
    }
    if (s < 0 || n == 0) return total_shortest;
    if (n <= 20) held_karp(s, e);
    else heuristic(s, e);
    solved = true;
    return total_shortest;
  }

  // visiting order in view ids, starting at now_view_id (main.cpp:558-593)
  std::vector<int> get_path_id_set() const {
    std::vector<int> ans;
    for (int i : global_path) ans.push_back(labels[i]);
    return ans;
  }

private:
  double path_len(const std::vector<int>& p) const {
    double d = 0;
    for (size_t i = 0; i + 1 < p.size(); i++) d += graph[p[i]][p[i + 1]];
    return d;
  }
  void held_karp(int s, int e) {
    const size_t FULL = (size_t)1 << n;
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> dp(FULL * n, INF);
    std::vector<int8_t> par(FULL * n, -1);
    dp[((size_t)1 << s) * n + s] = 0;
    for (size_t mask = 0; mask < FULL; mask++) {
      if (!((mask >> s) & 1)) continue;
      for (int j = 0; j < n; j++) {
        const double cur = dp[mask * n + j];
        if (cur == INF) continue;
        for (int k = 0; k < n; k++) {
          if ((mask >> k) & 1) continue;
          const size_t nm = mask | ((size_t)1 << k);
          const double v = cur + graph[j][k];
          if (v < dp[nm * n + k]) {
            dp[nm * n + k] = v;
            par[nm * n + k] = (int8_t)j;
          }
        }
      }
    }
    int end = e;
    if (end < 0) {
      end = 0;
      for (int j = 1; j < n; j++)
        if (dp[(FULL - 1) * n + j] < dp[(FULL - 1) * n + end]) end = j;
    }
    total_shortest = dp[(FULL - 1) * n + end];
    global_path.clear();
    size_t mask = FULL - 1;
    for (int cur = end; cur >= 0;) {
      global_path.push_back(cur);
      const int p = par[mask * n + cur];
      mask ^= (size_t)1 << cur;
      cur = p;
    }
    std::reverse(global_path.begin(), global_path.end());
    exact = true;
  }
  void heuristic(int s, int e) { // nearest neighbour, then 2-opt and Or-opt on the open path
    std::vector<int> p{s};
    std::vector<char> used(n, 0);
    used[s] = 1;
    if (e >= 0) used[e] = 1;
    while ((int)p.size() < n - (e >= 0 ? 1 : 0)) {
      int best = -1;
      for (int k = 0; k < n; k++)
        if (!used[k] && (best < 0 || graph[p.back()][k] < graph[p.back()][best])) best = k;
      used[best] = 1;
      p.push_back(best);
    }
    if (e >= 0) p.push_back(e);
    const int last_free = (int)p.size() - (e >= 0 ? 1 : 0); // positions [1, last_free) may move
    bool improved = true;
    for (int round = 0; improved && round < 200; round++) {
      improved = false;
      for (int i = 1; i < last_free; i++)
        for (int j = i + 1; j < last_free; j++) { // reverse p[i..j]
          const double before = graph[p[i - 1]][p[i]] + (j + 1 < (int)p.size() ? graph[p[j]][p[j + 1]] : 0.0);
          const double after = graph[p[i - 1]][p[j]] + (j + 1 < (int)p.size() ? graph[p[i]][p[j + 1]] : 0.0);
          if (after + 1e-12 < before) {
            std::reverse(p.begin() + i, p.begin() + j + 1);
            improved = true;
          }
        }
      for (int i = 1; i < last_free; i++) { // Or-opt: move one vertex elsewhere
        const int v = p[i];
        const double gain = graph[p[i - 1]][v] + (i + 1 < (int)p.size() ? graph[v][p[i + 1]] - graph[p[i - 1]][p[i + 1]] : 0.0);
        for (int j = 0; j + 1 <= (int)p.size() - 1 && !improved; j++) {
          if (j == i || j == i - 1) continue;
          if (e >= 0 && j >= (int)p.size() - 1) continue;
          const double cost = graph[p[j]][v] + (j + 1 < (int)p.size() ? graph[v][p[j + 1]] - graph[p[j]][p[j + 1]] : 0.0);
          if (cost + 1e-12 < gain) {
            p.erase(p.begin() + i);
            const int at = j < i ? j + 1 : j;
            p.insert(p.begin() + at, v);
            improved = true;
          }
        }
        if (improved) break;
      }
    }
    global_path = p;
    total_shortest = path_len(p);
    exact = false;
  }
};

} // namespace prvhost
