// path_planner.hpp -- movement cost between views and the visiting order of a view set.
// get_local_path (View_Space.hpp:206-305) in vector form (same doubles as the reference formula: the stored tours and
// movement/<it>.txt depend on them) and, in place of Global_Path_Planner
// (main.cpp:398-594, a Gurobi MILP with lazy sub-tour cuts) by an exact Held-Karp dynamic
// programme for up to 20 views and an iterated local search (2-opt + Or-opt + segment swaps) beyond -- flagged
// inexact, though it reaches the stored Gurobi tour length on every view set the reference ships (N = 3..100).
// Same problem statement: shortest Hamiltonian PATH from now_view_id (free or fixed end): the
// reference models it as a tour through a zero-cost copy node (main.cpp:428-437, 488-490).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>

#include "View_Space.hpp"

namespace prvhost {

enum { ErrorPath = -2, WrongPath = -1, LinePath = 0, CirclePath = 1 }; // View_Space.hpp:201-204

inline double dot3(const Vec3& u, const Vec3& v) { return u.x * v.x + u.y * v.y + u.z * v.z; }

// The great circle through two points of the sphere (O, r), parametrised as the reference does (View_Space.hpp:255-297):
// with n = p x q the circle's normal, a point of the circle is O + e1 cos(t) + e2 sin(t), where e1 lies in the xy plane
// (perpendicular to n's xy part) and e2 = the in-plane direction that climbs in -z.  Only x and y of the basis are ever
// compared, so only those are kept.
struct ArcFrame {
  double n_xy, n_len; // |(n.x, n.y)| and |n|
  double e1x, e1y, e2x, e2y;
  double r;
  ArcFrame(const Vec3& p, const Vec3& q, double radius) : r(radius) {
    const Vec3 n = p.cross(q);
    n_xy = std::sqrt(n.x * n.x + n.y * n.y);
    n_len = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
    e1x = r * n.y / n_xy;
    e1y = r * n.x / n_xy; // enters with a minus sign below
    e2x = r * n.x * n.z / n_xy / n_len;
    e2y = r * n.y * n.z / n_xy / n_len;
  }
  static double wrap(double t) { // one turn up or down into [0, 2 pi)
    const double two_pi = 2.0 * std::acos(-1.0);
    if (t < 0) t += two_pi;
    if (t >= two_pi) t -= two_pi;
    return t;
  }
  // the circle parameter of a point of the circle: asin of its height above the centre fixes t up to the mirror pi - t;
  // the branch whose x and y reproduce the point (to 1e-6) is taken (:262-276, 279-293: the same rule for both ends)
  double parameter(const Vec3& point, const Vec3& centre) const {
    double t = wrap(std::asin(-(point.z - centre.z) / r * n_len / n_xy));
    const double xt = centre.x + e1x * std::cos(t) + e2x * std::sin(t);
    const double yt = centre.y - e1y * std::cos(t) + e2y * std::sin(t);
    if (std::fabs(point.x - xt) > 1e-6 || std::fabs(point.y - yt) > 1e-6) t = wrap(std::acos(-1.0) - t);
    return t;
  }
};

// straight segment M->N, or M->P + arc PQ on the sphere (O, r) + Q->N when the segment crosses it (View_Space.hpp:206-305).
// P, Q = the roots of |M + t (N - M) - O|^2 = r^2 in t; no real pair, or both outside [0, 1]: the straight line;
// exactly one inside: an end point lies inside the obstacle (WrongPath).
inline std::pair<int, double> get_local_path(const Vec3& M, const Vec3& N, const Vec3& O, double r) {
  const Vec3 step = N - M, from_centre = M - O;
  const double qa = dot3(step, step), qb = 2.0 * dot3(step, from_centre), qc = dot3(from_centre, from_centre) - r * r;
  const double disc = qb * qb - 4.0 * qa * qc;
  if (disc <= 0) return {LinePath, step.norm()}; // :218-223
  double t_in = (-qb - std::sqrt(disc)) / (2.0 * qa), t_out = (-qb + std::sqrt(disc)) / (2.0 * qa);
  const bool in_off = t_in < 0 || t_in > 1, out_off = t_out < 0 || t_out > 1;
  if (in_off && out_off) return {LinePath, step.norm()}; // :228-233
  if (in_off || out_off) return {WrongPath, 1e10};       // :234-237
  if (t_in > t_out) std::swap(t_in, t_out);
  const Vec3 P = step * t_in + M, Q = step * t_out + M;
  const Vec3 p = P - O, q = Q - O;
  const ArcFrame circle(p, q, r);
  const double arc = std::fabs(circle.parameter(P, O) - circle.parameter(Q, O)) * r; // :299 (as written: not the shorter arc when > pi)
  return {CirclePath, (M - P).norm() + arc + (Q - N).norm()};
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
  // ---- beyond the dynamic programme: iterated local search.  The open path is a TOUR through one extra node that
  // costs nothing to reach from anywhere (the reference's own modelling trick, main.cpp:428-437) and is welded to
  // the start view -- and to the end view when that is fixed -- by a hugely negative edge.  Local search = 2-opt and
  // Or-opt (segments of 1..3 views, either way round) over neighbour lists with don't-look bits; perturbation =
  // a double bridge that spares the welded edges, accepted when shorter (or, early in the run, barely longer);
  // restart from the best tour when the search stalls; at the end every window of 12 consecutive views is re-ordered
  // exactly.
  // Deterministic (own LCG, an iteration budget, no clock).  Not a proof of optimality -- `exact` stays false --
  // but on the reference's own view sets (Hemisphere/N.txt, N = 21..100) it reaches the length of the stored
  // Gurobi tours or a shorter one (tests/test_host.py).
  struct Ils {
    int m = 0;                         // nodes incl. the extra one (index m - 1)
    std::vector<std::vector<double>> D; // m x m
    std::vector<std::vector<int>> near; // candidate neighbours, nearest first
    std::vector<int> t, pos;           // tour and inverse
    std::vector<char> look;            // 1 = examine this city
    static constexpr double kWeld = -1e6;

    int next(int c) const { return t[pos[c] + 1 == m ? 0 : pos[c] + 1]; }
    int prev(int c) const { return t[pos[c] == 0 ? m - 1 : pos[c] - 1]; }
    void set_tour(const std::vector<int>& tour) {
      t = tour;
      pos.assign(m, 0);
      for (int i = 0; i < m; i++) pos[t[i]] = i;
    }
    double length() const {
      double d = 0;
      for (int i = 0; i < m; i++) d += D[t[i]][t[i + 1 == m ? 0 : i + 1]];
      return d;
    }
    // reverse the tour between cities b .. c (inclusive, walking forward from b)
    void reverse(int b, int c) {
      int i = pos[b], j = pos[c];
      int len = j - i;
      if (len < 0) len += m;
      len += 1;
      for (int k = 0; k < len / 2; k++) {
        const int x = t[i], y = t[j];
        t[i] = y, pos[y] = i;
        t[j] = x, pos[x] = j;
        i = i + 1 == m ? 0 : i + 1;
        j = j == 0 ? m - 1 : j - 1;
      }
    }
    bool try_2opt(int a) {
      for (int dir = 0; dir < 2; dir++) {
        const int b = dir == 0 ? next(a) : prev(a);
        const double dab = D[a][b];
        for (int c : near[a]) {
          if (D[a][c] >= dab) break;
          const int d = dir == 0 ? next(c) : prev(c);
          if (c == b || d == a) continue;
          if (D[a][c] + D[b][d] - dab - D[c][d] < -1e-12) {
            if (dir == 0) reverse(b, c); // a b ... c d  ->  a c ... b d
            else reverse(c, b);          // d c ... b a  ->  d b ... c a
            look[a] = look[b] = look[c] = look[d] = 1;
            return true;
          }
        }
      }
      return false;
    }
    bool try_oropt(int a) {
      for (int L = 1; L <= 3 && L < m - 2; L++) {
        // the segment a .. e walking forward
        int e = a;
        for (int k = 1; k < L; k++) e = next(e);
        const int p = prev(a), q = next(e);
        if (p == e || q == a || p == q) continue;
        const double gain = D[p][a] + D[e][q] - D[p][q];
        if (gain <= 1e-12) continue;
        auto inside = [&](int c) {
          int x = a;
          for (int k = 0; k < L; k++) {
            if (x == c) return true;
            x = next(x);
          }
          return false;
        };
        for (int end = 0; end < 2; end++) {
          const int x = end == 0 ? a : e, y = end == 0 ? e : a; // x lands next to c, y on the far side
          for (int c : near[x]) {
            if (D[x][c] >= gain) break;
            if (inside(c)) continue;
            // neighbours of c once the segment is out
            const int cn = c == p ? q : next(c), cp = c == q ? p : prev(c);
            for (int side = 0; side < 2; side++) {
              const int o = side == 0 ? cn : cp; // the segment goes between c and o
              if (inside(o)) continue;
              const double add = D[c][x] + D[y][o] - D[c][o];
              if (add - gain < -1e-12) {
                // rebuild: tour without the segment, then the segment between c and o with x next to c
                std::vector<int> seg, rest;
                int w = a;
                for (int k = 0; k < L; k++) seg.push_back(w), w = next(w);
                for (int i = 0; i < m; i++)
                  if (!inside(t[i])) rest.push_back(t[i]);
                // orientation in the array: ... c seg o ... needs seg to start with x when o follows c
                std::vector<int> ins = seg; // a .. e
                const bool c_before_o = side == 0;
                if (c_before_o ? x != ins.front() : x != ins.back()) std::reverse(ins.begin(), ins.end());
                int at = 0;
                for (int i = 0; i < (int)rest.size(); i++)
                  if (rest[i] == (c_before_o ? c : o)) at = i + 1;
                rest.insert(rest.begin() + at, ins.begin(), ins.end());
                set_tour(rest);
                look[p] = look[q] = look[c] = look[o] = 1;
                for (int v : seg) look[v] = 1;
                return true;
              }
            }
          }
        }
      }
      return false;
    }
    void local_search() {
      for (bool any = true; any;) {
        any = false;
        for (int c = 0; c < m; c++) {
          if (!look[c]) continue;
          if (try_2opt(c) || try_oropt(c)) any = true;
          else look[c] = 0;
        }
      }
    }
  };
  // exact re-ordering of every window of 12 consecutive views (first and last of the window stay; the path's free
  // end is free in the last window) by the same dynamic programme, slid along the path until nothing improves:
  // catches the local rearrangements no 2-opt / Or-opt move reaches
  void polish_windows(std::vector<int>& path, bool end_fixed) const {
    const int W = 12, np = (int)path.size();
    if (np <= W) return;
    std::vector<int> starts;
    for (int i = 0; i < np - W; i += W / 2) starts.push_back(i);
    starts.push_back(np - W);
    for (bool improved = true; improved;) {
      improved = false;
      for (int i : starts) {
        const bool free_tail = !end_fixed && i + W == np;
        // Held-Karp over the window: node 0 fixed first, node W-1 fixed last unless free_tail
        const int* v = &path[i];
        const int FULL = 1 << W;
        static thread_local std::vector<double> dp;
        static thread_local std::vector<int8_t> par;
        dp.assign((size_t)FULL * W, std::numeric_limits<double>::infinity());
        par.assign((size_t)FULL * W, -1);
        dp[(size_t)1 * W + 0] = 0.0;
        for (int mask = 1; mask < FULL; mask += 2) // node 0 always in
          for (int j = 0; j < W; j++) {
            const double cur = dp[(size_t)mask * W + j];
            if (!(cur < std::numeric_limits<double>::infinity())) continue;
            for (int k = 1; k < W; k++) {
              if ((mask >> k) & 1) continue;
              const int nm = mask | (1 << k);
              const double val = cur + graph[v[j]][v[k]];
              if (val < dp[(size_t)nm * W + k]) dp[(size_t)nm * W + k] = val, par[(size_t)nm * W + k] = (int8_t)j;
            }
          }
        int end = W - 1;
        if (free_tail)
          for (int j = 1; j < W; j++)
            if (dp[(size_t)(FULL - 1) * W + j] < dp[(size_t)(FULL - 1) * W + end]) end = j;
        double old = 0;
        for (int k = 0; k + 1 < W; k++) old += graph[v[k]][v[k + 1]];
        if (dp[(size_t)(FULL - 1) * W + end] < old - 1e-12) {
          std::vector<int> ord;
          int mask = FULL - 1;
          for (int cur = end; cur >= 0;) {
            ord.push_back(v[cur]);
            const int pr = par[(size_t)mask * W + cur];
            mask ^= 1 << cur;
            cur = pr;
          }
          std::reverse(ord.begin(), ord.end());
          std::copy(ord.begin(), ord.end(), path.begin() + i);
          improved = true;
        }
      }
    }
  }
  void heuristic(int s, int e) {
    Ils S;
    const int m = S.m = n + 1, X = n; // X = the extra node
    S.D.assign(m, std::vector<double>(m, 0.0));
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) S.D[i][j] = graph[i][j];
    S.D[X][s] = S.D[s][X] = Ils::kWeld;
    if (e >= 0) S.D[X][e] = S.D[e][X] = Ils::kWeld;
    const int K = std::min(m - 1, 12);
    S.near.assign(m, {});
    for (int i = 0; i < m; i++) {
      std::vector<int> ord;
      for (int j = 0; j < m; j++)
        if (j != i) ord.push_back(j);
      std::stable_sort(ord.begin(), ord.end(), [&](int u, int v) { return S.D[i][u] < S.D[i][v]; });
      ord.resize(K);
      S.near[i] = ord;
    }
    // nearest-neighbour start: X, s, ..., (e)
    std::vector<int> p{X, s};
    std::vector<char> used(m, 0);
    used[X] = used[s] = 1;
    if (e >= 0) used[e] = 1;
    while ((int)p.size() < m - (e >= 0 ? 1 : 0)) {
      int best = -1;
      for (int k = 0; k < n; k++)
        if (!used[k] && (best < 0 || graph[p.back()][k] < graph[p.back()][best])) best = k;
      used[best] = 1;
      p.push_back(best);
    }
    if (e >= 0) p.push_back(e);
    S.set_tour(p);
    S.look.assign(m, 1);
    S.local_search();
    std::vector<int> best = S.t, cur = S.t;
    double best_len = S.length(), cur_len = best_len;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](int k) { // uniform in [0, k)
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      return (int)((rng >> 33) % (uint64_t)k);
    };
    const int iterations = ils_iterations > 0 ? ils_iterations : 2000 * n;
    const int stall_max = 100;
    // a path up to `accept` longer than the current one is taken now and then; the allowance shrinks to zero over
    // the run (measured on the reference's 80 view sets of 21..100 views: without it 2 sets end 5e-5..2e-4 above
    // the stored tour, with it none)
    const double accept = 0.08;
    int stall = 0;
    std::vector<int> rot(m), cand;
    for (int it = 0; it < iterations && m >= 8; it++) {
      // rotate so that the extra node leads: [X, s, ......, (e)]; cuts 2 <= a < b < c <= m (m - 1 with a fixed end)
      const int px = (int)(std::find(cur.begin(), cur.end(), X) - cur.begin());
      for (int i = 0; i < m; i++) rot[i] = cur[(px + i) % m];
      if (rot[1] != s) { // the tour runs the other way round: X, (e), ..., s
        std::reverse(rot.begin() + 1, rot.end());
      }
      const int hi = e >= 0 ? m - 1 : m;
      int a = 2 + rnd(hi - 1), b = 2 + rnd(hi - 1), c = 2 + rnd(hi - 1);
      int lo = std::min(a, std::min(b, c)), up = std::max(a, std::max(b, c)), mid = a + b + c - lo - up;
      if (lo == mid || mid == up) continue;
      cand.assign(rot.begin(), rot.begin() + lo);
      cand.insert(cand.end(), rot.begin() + mid, rot.begin() + up);
      cand.insert(cand.end(), rot.begin() + lo, rot.begin() + mid);
      cand.insert(cand.end(), rot.begin() + up, rot.end());
      S.set_tour(cand);
      S.look.assign(m, 0);
      for (int cut : {lo, mid, up}) {
        S.look[rot[cut - 1]] = 1;
        S.look[rot[cut == m ? 0 : cut]] = 1;
      }
      S.local_search();
      const double len = S.length();
      if (len < cur_len - 1e-12 || (accept > 0 && len < cur_len + accept * (1.0 - (double)it / iterations) && rnd(4) == 0)) {
        if (len < cur_len - 1e-12) stall = 0;
        cur = S.t;
        cur_len = len;
        if (len < best_len - 1e-12) best = S.t, best_len = len;
      } else if (++stall > stall_max) {
        cur = best;
        cur_len = best_len;
        stall = 0;
      }
    }
    // cut the tour at the extra node, start view first
    const int px = (int)(std::find(best.begin(), best.end(), X) - best.begin());
    std::vector<int> path;
    for (int i = 1; i < m; i++) path.push_back(best[(px + i) % m]);
    if (path.front() != s) std::reverse(path.begin(), path.end());
    polish_windows(path, e >= 0);
    global_path = path;
    total_shortest = path_len(path);
    exact = false;
  }

public:
  int ils_iterations = 0; // 0: 2000 x the number of views
};

} // namespace prvhost
