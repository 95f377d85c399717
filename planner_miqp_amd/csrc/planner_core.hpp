// planner_core.hpp - bark-free restatement of the host logic that sits directly on top of the solve path:
//   * region tables and region bookkeeping   common/parameter/parameter_preparer.cpp:37-143, regions.cpp:16-127
//   * receding-horizon warm start (shift by one step)                src/miqp_planner.cpp:787-1051
//   * the region-combination retry loop of MiqpPlanner::Plan         src/miqp_planner.cpp:633-766
//   * ReferenceTrajectoryGenerator on a polyline, MiqpPlanner::UpdateCar   common/reference/reference_trajectory_generator.cpp:51-148, src/miqp_planner.cpp:284-390
// Plain arrays in, plain arrays out; float where the reference computes in float (the rotated limits, the region test).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/miqp_types.h"

namespace miqp {

// fraction_parameters: sector k is bounded by the rays through (F[k][0], F[k][1]) and (F[k][2], F[k][3])
// (ParameterPreparer::CalculateFractionParameters, parameter_preparer.cpp:37-52; maxVelocityFitting is a float there)
inline void fraction_parameters(int R, float vmax_fit, double* out) {
  for (int k = 0; k < R; ++k) {
    const double step = (2.0 * M_PI - 0.0) / R;   // Eigen setLinSpaced(0, 2 pi) over R + 1 points: low + k * step, head(R)
    double a0 = 0.0 + k * step;
    out[k * 4 + 0] = vmax_fit * std::cos(a0); out[k * 4 + 1] = vmax_fit * std::sin(a0);
  }
  for (int k = 0; k < R; ++k) { int n = (k + 1) % R; out[k * 4 + 2] = out[n * 4 + 0]; out[k * 4 + 3] = out[n * 4 + 1]; }
}

// FittingPolynomialParameters (common/parameter/fitting_polynomial_parameters.hpp:28-92, 97-168): the six fitted R x 3
// tables of one (nr_regions, max_velocity_fitting, min_velocity_fitting) variant, row-major [region][3], in the order
// SINT_UB, SINT_LB, COSS_UB, COSS_LB, KAPPA_AX_MAX, KAPPA_AX_MIN; false for a combination the reference rejects with
// "Invalid number of regions or velocity!"
#include "fitting_tables.inc"
inline bool fitting_polynomial_parameters(int R, float vmax_fit, float vmin_fit, double* out /* [6][R*3] */) {
  for (const auto& v : FIT_VARIANTS)
    if (v.R == R && (float)v.vmax == vmax_fit && (float)v.vmin == vmin_fit) {
      const double* t = &v.t[0][0];
      for (int k = 0; k < 6 * 3 * R; ++k) out[k] = t[k];
      return true;
    }
  return false;
}

inline double wrap_2pi(double a) { a = std::fmod(a, 2.0 * M_PI); if (a < 0) a += 2.0 * M_PI; return a; }

// ParameterPreparer::CalculateMeanAngleVector (parameter_preparer.cpp:96-113)
inline void mean_angles(const double* F, int R, double* out) {
  for (int k = 0; k < R; ++k) {
    double a1 = wrap_2pi(std::atan2(F[k * 4 + 1], F[k * 4 + 0])), a2 = wrap_2pi(std::atan2(F[k * 4 + 3], F[k * 4 + 2]));
    if (k + 1 == R) a2 += M_PI * 2.0;
    out[k] = 0.5 * (a1 + a2);
  }
}

// RotateLimitVectors (parameter_preparer.cpp:115-143): the box of the four rotated corner sums, in float
inline void rotate_limits(float long_min, float long_max, float lat_min, float lat_max, float angle, float& min_x, float& max_x, float& min_y, float& max_y) {
  auto rot = [&](float x, float y, float& ox, float& oy) { ox = x * std::cos(angle) - y * std::sin(angle); oy = x * std::sin(angle) + y * std::cos(angle); };
  float x[4], y[4];
  rot(long_max, lat_max, x[0], y[0]); rot(long_max, lat_min, x[1], y[1]); rot(long_min, lat_max, x[2], y[2]); rot(long_min, lat_min, x[3], y[3]);
  max_x = std::max(std::max(x[0], x[1]), std::max(x[2], x[3])); min_x = std::min(std::min(x[0], x[1]), std::min(x[2], x[3]));
  max_y = std::max(std::max(y[0], y[2]), std::max(y[3], y[1])); min_y = std::min(std::min(y[0], y[2]), std::min(y[3], y[1]));
  if (max_x < min_x) std::swap(max_x, min_x);
  if (max_y < min_y) std::swap(max_y, min_y);
}

// CalculateAccLimitsPerCar / CalculateJerkLimitsPerCar (parameter_preparer.cpp:54-94): out[4][R] = min_x, max_x, min_y, max_y
inline void limits_per_region(const double* F, int R, float long_min, float long_max, float lat_min, float lat_max, double* min_x, double* max_x, double* min_y, double* max_y) {
  std::vector<double> ang(R); mean_angles(F, R, ang.data());
  for (int k = 0; k < R; ++k) {
    float a, b, c, d; rotate_limits(long_min, long_max, lat_min, lat_max, (float)ang[k], a, b, c, d);
    min_x[k] = a; max_x[k] = b; min_y[k] = c; max_y[k] = d;
  }
}

// CalculateRegionIdx (regions.cpp:16-33): every sector whose two half-planes hold within eps = 1e-3 (float vx, vy);
// returns the count, indices ascending (the multimap iterates the non-initial entries in insertion order)
inline int calculate_region_idx(const double* F, int R, float vx, float vy, int* out) {
  const float eps = 1e-3f; int n = 0;
  for (int k = 0; k < R; ++k) {
    bool below_ub = F[k * 4 + 2] * vy <= F[k * 4 + 3] * vx + eps;
    bool above_lb = F[k * 4 + 0] * vy >= F[k * 4 + 1] * vx - eps;
    if (below_ub && above_lb) out[n++] = k;
  }
  return n;
}

// ReserveNeighborRegions (regions.cpp:75-112) on one row of R flags, including its quirk: `first`/`last` start as
// numeric_limits<int>::quiet_NaN() == 0 and isnan(int) is never true, so a row without a 0/1 border sets region 0
inline bool reserve_neighbor_regions(int* row, int R, int expansions) {
  for (int e = 0; e < (expansions > 1 ? expansions : 1); ++e) {
    int first = 0, last = 0;
    for (int i = 0; i < R; ++i)
      if (row[i] == 1) {
        if (i >= 1 && row[i - 1] == 0) first = i - 1;
        if (i + 1 < R && row[i + 1] == 0) last = i + 1;
        if (i == 0 && row[R - 1] == 0) first = R - 1;
        if (i == R - 1 && row[0] == 0) last = 0;
      }
    row[first] = 1; row[last] = 1;
  }
  return true;
}

// CalculatePossibleRegions (regions.cpp:114-127): union of the regions of the headings theta[k]
inline void calculate_possible_regions(const double* F, int R, const double* theta, int n, int* flags) {
  std::vector<int> idx(R);
  for (int k = 0; k < R; ++k) flags[k] = 0;
  for (int k = 0; k < n; ++k) { int m = calculate_region_idx(F, R, (float)std::cos(theta[k]), (float)std::sin(theta[k]), idx.data()); for (int q = 0; q < m; ++q) flags[idx[q]] = 1; }
}

// CalculateRegionCombinations (regions.cpp:45-62): cartesian product, car 0 slowest; per car the candidate marked as
// initial region (SetInitialRegion) comes first, the others ascending
inline void region_combinations(const std::vector<std::vector<int>>& per_car, std::vector<std::vector<int>>& out) {
  out.clear(); std::vector<int> cur;
  struct Rec { static void go(const std::vector<std::vector<int>>& pc, size_t c, std::vector<int>& cur, std::vector<std::vector<int>>& out) {
    if (c == pc.size()) { out.push_back(cur); return; }
    for (int r : pc[c]) { cur.push_back(r); go(pc, c + 1, cur, out); cur.pop_back(); } } };
  Rec::go(per_car, 0, cur, out);
}

// ---------------------------------------------------------------- ReferenceTrajectoryGenerator on a polyline
// common/reference/reference_trajectory_generator.cpp:51-148 (GenerateTrajectoryInternal).  The reference walks a
// bark::geometry::Line (third-party: bark-simulator/bark @ 53562ac, util/deps.bzl:7-10, absent from the checkout) that it
// first passes through bark's SmoothLine (spline resampling at line_interp_inc).  Restated here on the POLYLINE itself -
// nearest point by projection onto the segments, point and tangent at arc length s by linear interpolation inside the
// segment, curvature by finite differences of the resampled vertices: identical to the reference on straight reference
// lines (the case its tests and the C-API test K8 pin numerically), an approximation of the spline on curved ones.
struct PolyLine {
  std::vector<double> x, y, s;
  void build(const double* xy, int n) {
    x.clear(); y.clear(); s.clear();
    for (int k = 0; k < n; ++k) {
      if (k && xy[2 * k] == x.back() && xy[2 * k + 1] == y.back()) continue;
      x.push_back(xy[2 * k]); y.push_back(xy[2 * k + 1]);
      s.push_back(k && !s.empty() ? s.back() + std::hypot(x.back() - x[x.size() - 2], y.back() - y[y.size() - 2]) : 0.0);
    }
  }
  // bark's SmoothLine (bark/geometry/line.hpp @ 53562ac, absent from the checkout) passes a cubic spline through the vertices,
  // parameterised by the arc length of the polyline, and samples it every `inc`.  Restated as the natural cubic spline (second
  // derivative zero at both ends) per coordinate: on collinear vertices it reproduces the polyline exactly, on a bent line it
  // spreads the curvature of a kink over the neighbouring spans - which is what makes the reference's curvature-dependent
  // velocity drop ahead of a bend (common/tests/reference_trajectory_generator_test.cc:96-136).
  PolyLine smoothed(double inc) const {
    const int n = (int)x.size();
    if (n < 3 || !(inc > 0)) return resampled(inc);
    auto second = [&](const std::vector<double>& f) {   // Thomas algorithm for the knot second derivatives
      std::vector<double> M(n, 0.0), cp(n, 0.0), dp(n, 0.0);
      for (int i = 1; i + 1 < n; ++i) {
        const double h0 = s[i] - s[i - 1], h1 = s[i + 1] - s[i];
        const double a = h0, b = 2.0 * (h0 + h1), c = h1, d = 6.0 * ((f[i + 1] - f[i]) / h1 - (f[i] - f[i - 1]) / h0);
        const double m = b - a * cp[i - 1];
        cp[i] = c / m; dp[i] = (d - a * dp[i - 1]) / m;
      }
      for (int i = n - 2; i >= 1; --i) M[i] = dp[i] - cp[i] * M[i + 1];
      return M;
    };
    const std::vector<double> Mx = second(x), My = second(y);
    auto eval = [&](double t, double& px, double& py) {
      t = std::min(std::max(t, 0.0), s.back());
      const int k = segment_of(t); const double h = s[k + 1] - s[k], A = (s[k + 1] - t) / h, B = 1.0 - A;
      const double ca = (A * A * A - A) * h * h / 6.0, cb = (B * B * B - B) * h * h / 6.0;
      px = A * x[k] + B * x[k + 1] + ca * Mx[k] + cb * Mx[k + 1]; py = A * y[k] + B * y[k + 1] + ca * My[k] + cb * My[k + 1];
    };
    PolyLine r;
    auto push = [&](double px, double py) { r.s.push_back(r.x.empty() ? 0.0 : r.s.back() + std::hypot(px - r.x.back(), py - r.y.back())); r.x.push_back(px); r.y.push_back(py); };
    for (double t = 0.0; t < s.back(); t += inc) { double px, py; eval(t, px, py); push(px, py); }
    push(x.back(), y.back());
    return r;
  }
  PolyLine resampled(double inc) const {   // vertices every `inc` of arc length (and the end point)
    PolyLine r; if (x.size() < 2 || !(inc > 0)) return *this;
    for (double t = 0.0; t < s.back(); t += inc) { double px, py; point_at(t, px, py); r.x.push_back(px); r.y.push_back(py); r.s.push_back(t); }
    r.x.push_back(x.back()); r.y.push_back(y.back()); r.s.push_back(s.back());
    return r;
  }
  int segment_of(double t) const { int k = 0; while (k + 2 < (int)s.size() && t >= s[k + 1]) ++k; return k; }
  void point_at(double t, double& px, double& py) const {
    if (x.size() < 2) { px = x.empty() ? 0 : x[0]; py = y.empty() ? 0 : y[0]; return; }
    t = std::min(std::max(t, 0.0), s.back());
    const int k = segment_of(t); const double L = s[k + 1] - s[k], a = L > 0 ? (t - s[k]) / L : 0.0;
    px = x[k] + a * (x[k + 1] - x[k]); py = y[k] + a * (y[k + 1] - y[k]);
  }
  double tangent_at(double t) const {
    if (x.size() < 2) return 0.0;
    t = std::min(std::max(t, 0.0), s.back());
    const int k = segment_of(t); return std::atan2(y[k + 1] - y[k], x[k + 1] - x[k]);
  }
  double nearest_s(double px, double py) const {
    double best = 1e300, bs = 0.0;
    for (size_t k = 0; k + 1 < x.size(); ++k) {
      const double dx = x[k + 1] - x[k], dy = y[k + 1] - y[k], L2 = dx * dx + dy * dy;
      double a = L2 > 0 ? ((px - x[k]) * dx + (py - y[k]) * dy) / L2 : 0.0; a = std::min(std::max(a, 0.0), 1.0);
      const double d = std::hypot(px - (x[k] + a * dx), py - (y[k] + a * dy));
      if (d < best) { best = d; bs = s[k] + a * std::sqrt(L2); }
    }
    return bs;
  }
  int nearest_idx(double px, double py) const {
    int b = 0; double best = 1e300;
    for (size_t k = 0; k < x.size(); ++k) { const double d = std::hypot(px - x[k], py - y[k]); if (d < best) { best = d; b = (int)k; } }
    return b;
  }
  double curvature(int k) const {   // finite differences at vertex k (one-sided at the ends)
    const int n = (int)x.size(); if (n < 3) return 0.0;
    const int a = std::max(0, std::min(k - 1, n - 3));
    const double x1 = x[a + 1] - x[a], y1 = y[a + 1] - y[a], x2 = x[a + 2] - x[a + 1], y2 = y[a + 2] - y[a + 1];
    const double cr = x1 * y2 - y1 * x2, l1 = std::hypot(x1, y1), l2 = std::hypot(x2, y2), l3 = std::hypot(x[a + 2] - x[a], y[a + 2] - y[a]);
    const double den = l1 * l2 * l3; return den > 0 ? 2.0 * cr / den : 0.0;
  }
};

inline double interpolate_with_bounds(double x0, double y0, double x1, double y1, double xi) {   // common/math/math.hpp:34-46
  if (xi > x1) return y1;
  if (xi < x0) return y0;
  return (y1 - y0) / (x1 - x0) * (xi - x0) + y0;
}

// out[num_points][5] = (time, x, y, theta, v) (bark's StateDefinition order); state = the same five of the start
inline void reference_trajectory(const PolyLine& center, const double* state, double dt, int num_points, double line_interp_inc, double vel_desired,
                                 double delta_s_desired, double acc_lat_max, bool vel_curve_dep, double* out) {
  const PolyLine line = center.smoothed(line_interp_inc);
  for (int q = 0; q < 5; ++q) out[q] = state[q];   // the state at t = 0
  const double s_start = line.nearest_s(state[1], state[2]);
  const double s_end = line.s.empty() ? 0.0 : line.s.back();
  double s_desired_vel = std::min(s_end, s_start + delta_s_desired);
  const double start_time = state[0];
  double s_i = s_start;
  const double vel_0 = delta_s_desired <= 0.0 ? vel_desired : state[4];
  double vel_i = vel_0, vel_end = vel_desired;
  if (vel_i * num_points * dt + s_i > s_end) {
    if (vel_end > 0 || s_desired_vel > s_end) { vel_end = 0; s_desired_vel = s_end; }   // the line ends inside the horizon
  }
  for (int i = 1; i < num_points; ++i) {
    s_i += vel_i * dt;
    double px, py; line.point_at(s_i, px, py);
    const double ang = line.tangent_at(s_i);
    if ((s_desired_vel - s_start) < 1e-2) vel_i = 0;
    else vel_i = interpolate_with_bounds(s_start, vel_0, s_desired_vel, vel_end, s_i);
    if (vel_curve_dep) {
      const double kap = std::fabs(line.curvature(line.nearest_idx(px, py)));
      if (kap > 0) vel_i = std::min(vel_i, std::sqrt(acc_lat_max / kap));
    }
    double* o = out + i * 5;
    o[0] = (double)i * dt + start_time; o[1] = px; o[2] = py; o[3] = ang; o[4] = vel_i;
  }
}

// MiqpPlanner::UpdateCar (src/miqp_planner.cpp:284-390) for one car, without bark types: reference rows of the model from the
// generated reference trajectory, possible regions from the headings of the longer-horizon reference plus their
// neighbours, and the weights; `initial` = (x, vx, ax, y, vy, ay).  Outputs: ref[4][N] = x_ref, y_ref, vx_ref, vy_ref;
// possible[R]; weights[8] = POS_X, VEL_X, ACC_X, POS_Y, VEL_Y, ACC_Y, JERK_X, JERK_Y.  Returns false when the region
// expansion fails (the reference only logs that).
struct CarUpdateSettings {   // the fields of MiqpPlannerSettings this function reads (src/miqp_planner_settings.h)
  int nr_regions, nr_steps, nr_neighbouring_possible_regions, additional_steps_longer_horizon;
  double ts, ref_line_interp_inc, acc_lat_max, lambda, position_weight, velocity_weight, acceleration_weight, jerk_weight;
};
inline bool update_car(const CarUpdateSettings& S, const double* F, const double* initial, const double* ref_xy, int n_ref, double desired_velocity,
                       double delta_s_desired, double timestep, bool track_reference_positions, bool is_ego, int num_cars, double* ref, int* possible, double* weights) {
  PolyLine line; line.build(ref_xy, n_ref);
  const double st[5] = {timestep, initial[0], initial[3], std::atan2(initial[4], initial[1]), std::sqrt(initial[1] * initial[1] + initial[4] * initial[4])};
  const int N = S.nr_steps, NL = S.nr_steps + S.additional_steps_longer_horizon;
  std::vector<double> tr((size_t)NL * 5);
  reference_trajectory(line, st, S.ts, N, S.ref_line_interp_inc, desired_velocity, delta_s_desired, S.acc_lat_max, false, tr.data());
  for (int i = 0; i < N; ++i) {
    ref[0 * N + i] = tr[i * 5 + 1]; ref[1 * N + i] = tr[i * 5 + 2];
    ref[2 * N + i] = tr[i * 5 + 4] * std::cos(tr[i * 5 + 3]); ref[3 * N + i] = tr[i * 5 + 4] * std::sin(tr[i * 5 + 3]);
  }
  reference_trajectory(line, st, S.ts, NL, S.ref_line_interp_inc, desired_velocity, delta_s_desired, S.acc_lat_max, false, tr.data());
  std::vector<double> th(NL); for (int i = 0; i < NL; ++i) th[i] = tr[i * 5 + 3];
  calculate_possible_regions(F, S.nr_regions, th.data(), NL, possible);
  const bool ok = reserve_neighbor_regions(possible, S.nr_regions, S.nr_neighbouring_possible_regions);
  const double scale = is_ego ? S.lambda : (1.0 - S.lambda) / (num_cars - 1);
  if (track_reference_positions) { weights[0] = weights[3] = scale * S.position_weight; weights[1] = weights[4] = scale * S.velocity_weight; }
  else { weights[0] = weights[3] = 0.0; weights[1] = weights[4] = 2.0; }
  weights[2] = weights[5] = scale * S.acceleration_weight; weights[6] = weights[7] = scale * S.jerk_weight;
  return ok;
}

// ---------------------------------------------------------------- environment and obstacles of MiqpPlanner (convex pieces given)
// The reference convexifies a bark map polygon (common/map/convexified_map.cpp, bark + boost::geometry: out of scope); everything
// after that works on convex counter-clockwise pieces and is restated here on plain vertex arrays (x0, y0, x1, y1, ...).

// boost::geometry::within(point, polygon): strictly inside (a point on the boundary is not within)
inline bool point_within_convex(const double* v, int n, double px, double py) {
  for (int k = 0; k < n; ++k) {
    const int k2 = (k + 1) % n;
    const double cr = (v[2 * k2] - v[2 * k]) * (py - v[2 * k + 1]) - (px - v[2 * k]) * (v[2 * k2 + 1] - v[2 * k + 1]);
    if (!(cr > 0.0)) return false;
  }
  return n >= 3;
}
inline bool point_in_or_on_convex(const double* v, int n, double px, double py) {
  for (int k = 0; k < n; ++k) {
    const int k2 = (k + 1) % n;
    const double cr = (v[2 * k2] - v[2 * k]) * (py - v[2 * k + 1]) - (px - v[2 * k]) * (v[2 * k2 + 1] - v[2 * k + 1]);
    if (cr < 0.0) return false;
  }
  return n >= 3;
}
// boost::geometry::intersects of two convex polygons (touching counts): separating axis test over the edge normals of both
inline bool convex_polygons_intersect(const double* a, int na, const double* b, int nb) {
  for (int pass = 0; pass < 2; ++pass) {
    const double* p = pass ? b : a; const int np = pass ? nb : na;
    for (int k = 0; k < np; ++k) {
      const int k2 = (k + 1) % np;
      const double nx = p[2 * k2 + 1] - p[2 * k + 1], ny = -(p[2 * k2] - p[2 * k]);   // a normal of edge k
      double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
      for (int q = 0; q < na; ++q) { const double d = nx * a[2 * q] + ny * a[2 * q + 1]; amin = std::min(amin, d); amax = std::max(amax, d); }
      for (int q = 0; q < nb; ++q) { const double d = nx * b[2 * q] + ny * b[2 * q + 1]; bmin = std::min(bmin, d); bmax = std::max(bmax, d); }
      if (amax < bmin || bmax < amin) return false;
    }
  }
  return na >= 3 && nb >= 3;
}
inline bool segment_hits_convex(const double* v, int n, double x0, double y0, double x1, double y1) {
  if (point_in_or_on_convex(v, n, x0, y0) || point_in_or_on_convex(v, n, x1, y1)) return true;
  const double seg[4] = {x0, y0, x1, y1};   // degenerate two-vertex "polygon": the separating axis test needs its normal too
  for (int k = 0; k < n; ++k) {   // proper crossing with an edge
    const int k2 = (k + 1) % n;
    auto orient = [](double ax, double ay, double bx, double by, double cx, double cy) { return (bx - ax) * (cy - ay) - (cx - ax) * (by - ay); };
    const double d1 = orient(v[2 * k], v[2 * k + 1], v[2 * k2], v[2 * k2 + 1], seg[0], seg[1]), d2 = orient(v[2 * k], v[2 * k + 1], v[2 * k2], v[2 * k2 + 1], seg[2], seg[3]);
    const double d3 = orient(seg[0], seg[1], seg[2], seg[3], v[2 * k], v[2 * k + 1]), d4 = orient(seg[0], seg[1], seg[2], seg[3], v[2 * k2], v[2 * k2 + 1]);
    if (((d1 > 0) != (d2 > 0)) && ((d3 > 0) != (d4 > 0))) return true;
  }
  return false;
}
// ConvexifiedMap::GetIntersectingConvexPolygons for one reference trajectory (its x, y as a polyline): does it touch the piece?
inline bool polyline_hits_convex(const double* pts, int npts, const double* v, int n) {
  if (npts == 1) return point_in_or_on_convex(v, n, pts[0], pts[1]);
  for (int k = 0; k + 1 < npts; ++k) if (segment_hits_convex(v, n, pts[2 * k], pts[2 * k + 1], pts[2 * k + 2], pts[2 * k + 3])) return true;
  return false;
}

// The initial-pose check of MiqpPlanner::Plan (src/miqp_planner.cpp:654-685): the rear axle point and the front axle point
// (wheel base ahead along atan2(vy, vx)) of every car must each lie within one of the environment pieces.  Returns -1 when
// every car passes, else 2 * car + (1: the front point failed, 0: the rear point failed).
inline int initial_pose_check(const miqp_model_params_c& p) {
  for (int c = 0; c < p.NumCars; ++c) {
    const double x = p.IntitialState[c * 6 + 0], y = p.IntitialState[c * 6 + 3];
    const double th = std::atan2(p.IntitialState[c * 6 + 4], p.IntitialState[c * 6 + 1]);
    const double fx = x + std::cos(th) * p.WheelBase[c], fy = y + std::sin(th) * p.WheelBase[c];
    bool rear = false, front = false;
    for (int e = 0; e < p.nr_environments; ++e) {
      const double* v = p.env_vertices + 2 * p.env_offsets[e]; const int n = p.env_offsets[e + 1] - p.env_offsets[e];
      rear = rear || point_within_convex(v, n, x, y); front = front || point_within_convex(v, n, fx, fy);
    }
    if (!rear) return 2 * c;
    if (!front) return 2 * c + 1;
  }
  return -1;
}

// MiqpPlanner::UpdateObstaclesROI (src/miqp_planner.cpp:1308-1335): the region of interest around the ego car, four vertices
// (front upper, front lower, rear lower, rear upper).  As in the reference the side offset is +-(sin theta, cos theta) * side - a
// parallelogram whose sides are only perpendicular to the heading at theta = 0 or pi / 2; restated as written, not "corrected".
inline void obstacles_roi(double x, double y, double theta, double behind, double front, double side, double* roi8) {
  const double c = std::cos(theta), s = std::sin(theta);
  const double fx = x + c * front, fy = y + s * front;
  const double rx = x + std::cos(theta + M_PI) * behind, ry = y + std::sin(theta + M_PI) * behind;
  roi8[0] = fx + s * side; roi8[1] = fy + c * side;
  roi8[2] = fx - s * side; roi8[3] = fy - c * side;
  roi8[4] = rx - s * side; roi8[5] = ry - c * side;
  roi8[6] = rx + s * side; roi8[7] = ry + c * side;
}

// MiqpPlanner::ObstacleIntersectsEnvironment (src/miqp_planner.cpp:1248-1306): does the obstacle (T time steps of 4 vertices)
// intersect one of the pieces - at step 0 only when it is static.  With a region of interest (roi: 4 vertices, or nullptr for the
// reference's empty polygon) a step at which the obstacle lies outside it does not count: a static obstacle is then irrelevant,
// a moving one is looked at again at its next step (:1278-1288).
inline bool obstacle_intersects_environment(const double* pieces, const int* off, int n_pieces, const double* obstacle, int T, bool is_static, const double* roi = nullptr) {
  if (n_pieces == 0) return true;   // empty environment: every obstacle is added
  for (int t = 0; t < T; ++t) {
    if (roi && !convex_polygons_intersect(roi, 4, obstacle + (size_t)t * 8, 4)) { if (is_static) return false; continue; }
    for (int e = 0; e < n_pieces; ++e)
      if (convex_polygons_intersect(pieces + 2 * off[e], off[e + 1] - off[e], obstacle + (size_t)t * 8, 4)) return true;
    if (is_static) return false;
  }
  return false;
}

// MiqpPlanner::GetBarkTrajectory (src/miqp_planner.cpp:1132-1170) on plain arrays: rows (time, x, y, theta, v) - bark's StateDefinition
// order TIME, X, Y, THETA, VEL - of car `car`, cut off at the first step whose velocity components are both at most `min_speed` in
// magnitude (IsVxVyValid :1184-1187; the planner's minimum_valid_speed_vx_vy_ is 0.7, :53-54: the heading atan2(vy, vx) means nothing there).
// Returns the number of rows written.
inline int bark_trajectory(const miqp_raw_results_c& r, int car, double start_time, double ts, double min_speed, double* out5) {
  const int N = r.N; int n = 0;
  for (int i = 0; i < N; ++i) {
    const double vx = r.vel_x[car * N + i], vy = r.vel_y[car * N + i];
    if (!(std::fabs(vx) > min_speed || std::fabs(vy) > min_speed)) break;
    double* o = out5 + 5 * n++;
    o[0] = start_time + i * ts; o[1] = r.pos_x[car * N + i]; o[2] = r.pos_y[car * N + i]; o[3] = std::atan2(vy, vx); o[4] = std::sqrt(vx * vx + vy * vy);
  }
  return n;
}

// MiqpPlanner::EnvironmentWarmstart (src/miqp_planner.cpp:1053-1115): the environment binaries of the warm start follow the
// piece ids - pieces that stay keep their columns (steps 0 .. N-2: the reference copies an extent of NumSteps - 1), new pieces
// and the last step start as 1 ("not within").  in / out are the five [C][E][N] arrays with E_old / E_new pieces.
inline void environment_warmstart(const int* const in[5], int* const out[5], int C, int N, const int* ids_old, int n_old, const int* ids_new, int n_new) {
  for (int f = 0; f < 5; ++f) {
    for (int q = 0; q < C * n_new * N; ++q) out[f][q] = 1;
    for (int c = 0; c < C; ++c)
      for (int e = 0; e < n_new; ++e) {
        int from = -1; for (int k = 0; k < n_old; ++k) if (ids_old[k] == ids_new[e]) { from = k; break; }
        if (from < 0) continue;
        for (int i = 0; i + 1 < N; ++i) out[f][(c * n_new + e) * N + i] = in[f][(c * n_old + from) * N + i];
      }
  }
}

// ---------------------------------------------------------------- MiqpPlanner::CalculateWarmstart (miqp_planner.cpp:787-1051)
// `w` (same sizes as `rr`, caller allocated) receives the last solution shifted by one step.  Quirks kept:
// the last step of every binary family except the region-change flags is copied UNSHIFTED from the last step of `rr`
// (:951-963, 997-1001, 1029-1046); the last-step active_region row stays all zero because `:982` is an expression
// without effect; slackvarsObstacle* are not touched (they keep whatever `w` held); u of the last step is zero and the
// last state is one explicit Euler step of the shifted step N-2 (:884-918).
inline void calculate_warmstart(const miqp_raw_results_c& rr, miqp_raw_results_c& w, double ts, double min_region_change_speed) {
  const int C = rr.NrCars, N = rr.N, R = rr.NrRegions, E = rr.NrEnvironments, O = rr.NrObstacles, L = rr.MaxLinesObstacles, K = rr.NrCarToCarCollisions;
  auto shift2d = [&](double* dst, const double* src) { for (int c = 0; c < C; ++c) for (int i = 0; i + 1 < N; ++i) dst[c * N + i] = src[c * N + i + 1]; };
  auto shift2i = [&](int* dst, const int* src) { for (int c = 0; c < C; ++c) for (int i = 0; i + 1 < N; ++i) dst[c * N + i] = src[c * N + i + 1]; };
  shift2d(w.u_x, rr.u_x); shift2d(w.u_y, rr.u_y); shift2d(w.pos_x, rr.pos_x); shift2d(w.vel_x, rr.vel_x); shift2d(w.acc_x, rr.acc_x);
  shift2d(w.pos_y, rr.pos_y); shift2d(w.vel_y, rr.vel_y); shift2d(w.acc_y, rr.acc_y);
  shift2d(w.pos_x_front_UB, rr.pos_x_front_UB); shift2d(w.pos_x_front_LB, rr.pos_x_front_LB);
  shift2d(w.pos_y_front_UB, rr.pos_y_front_UB); shift2d(w.pos_y_front_LB, rr.pos_y_front_LB);
  shift2i(w.region_change_not_allowed_combined, rr.region_change_not_allowed_combined);
  shift2i(w.region_change_not_allowed_x_negative, rr.region_change_not_allowed_x_negative);
  shift2i(w.region_change_not_allowed_x_positive, rr.region_change_not_allowed_x_positive);
  shift2i(w.region_change_not_allowed_y_negative, rr.region_change_not_allowed_y_negative);
  shift2i(w.region_change_not_allowed_y_positive, rr.region_change_not_allowed_y_positive);
  for (int c = 0; c < C; ++c) {
    const int a = c * N + N - 1, b = c * N + N - 2;
    w.u_x[a] = 0; w.u_y[a] = 0;
    w.pos_x[a] = w.pos_x[b] + ts * w.vel_x[b]; w.pos_y[a] = w.pos_y[b] + ts * w.vel_y[b];
    w.vel_x[a] = w.vel_x[b] + ts * w.acc_x[b]; w.vel_y[a] = w.vel_y[b] + ts * w.acc_y[b];
    w.acc_x[a] = w.acc_x[b] + ts * w.u_x[b]; w.acc_y[a] = w.acc_y[b] + ts * w.u_y[b];
    w.pos_x_front_UB[a] = w.pos_x_front_UB[b] + ts * w.vel_x[b]; w.pos_x_front_LB[a] = w.pos_x_front_LB[b] + ts * w.vel_x[b];
    w.pos_y_front_UB[a] = w.pos_y_front_UB[b] + ts * w.vel_y[b]; w.pos_y_front_LB[a] = w.pos_y_front_LB[b] + ts * w.vel_y[b];
    const int xp = w.vel_x[a] <= min_region_change_speed, yp = w.vel_y[a] <= min_region_change_speed;
    const int xn = w.vel_x[a] >= -min_region_change_speed, yn = w.vel_y[a] >= -min_region_change_speed;
    w.region_change_not_allowed_x_positive[a] = xp; w.region_change_not_allowed_y_positive[a] = yp;
    w.region_change_not_allowed_x_negative[a] = xn; w.region_change_not_allowed_y_negative[a] = yn;
    w.region_change_not_allowed_combined[a] = (xp + yp + xn + yn) > 3;
  }
  if (E > 0) {
    int* dst[5] = {w.notWithinEnvironmentRear, w.notWithinEnvironmentFrontUbUb, w.notWithinEnvironmentFrontUbLb, w.notWithinEnvironmentFrontLbUb, w.notWithinEnvironmentFrontLbLb};
    const int* src[5] = {rr.notWithinEnvironmentRear, rr.notWithinEnvironmentFrontUbUb, rr.notWithinEnvironmentFrontUbLb, rr.notWithinEnvironmentFrontLbUb, rr.notWithinEnvironmentFrontLbLb};
    for (int f = 0; f < 5; ++f)
      for (int q = 0; q < C * E; ++q) { for (int i = 0; i + 1 < N; ++i) dst[f][q * N + i] = src[f][q * N + i + 1]; dst[f][q * N + N - 1] = src[f][q * N + N - 1]; }
  }
  for (int c = 0; c < C; ++c) {
    for (int i = 0; i + 1 < N; ++i) for (int j = 0; j < R; ++j) w.active_region[(c * N + i) * R + j] = rr.active_region[(c * N + i + 1) * R + j];
    for (int j = 0; j < R; ++j) w.active_region[(c * N + N - 1) * R + j] = 0;
  }
  if (K > 0) {
    for (int q = 0; q < K * K; ++q) {
      for (int i = 0; i + 1 < N; ++i) for (int s = 0; s < 16; ++s) w.car2car_collision[(q * N + i) * 16 + s] = rr.car2car_collision[(q * N + i + 1) * 16 + s];
      for (int s = 0; s < 16; ++s) w.car2car_collision[(q * N + N - 1) * 16 + s] = rr.car2car_collision[(q * N + N - 1) * 16 + s];
      for (int i = 0; i + 1 < N; ++i) for (int s = 0; s < 4; ++s) {
        w.slackvars[(q * N + i) * 4 + s] = rr.slackvars[(q * N + i + 1) * 4 + s];
        if (w.slackvars_real && rr.slackvars_real) w.slackvars_real[(q * N + i) * 4 + s] = rr.slackvars_real[(q * N + i + 1) * 4 + s];
      }
      for (int s = 0; s < 4; ++s) {
        w.slackvars[(q * N + N - 1) * 4 + s] = rr.slackvars[(q * N + N - 1) * 4 + s];
        if (w.slackvars_real && rr.slackvars_real) w.slackvars_real[(q * N + N - 1) * 4 + s] = rr.slackvars_real[(q * N + N - 1) * 4 + s];
      }
    }
  }
  if (O > 0) {
    for (int q = 0; q < C * O; ++q) {
      for (int i = 0; i + 1 < N; ++i) for (int k = 0; k < L; ++k) {
        w.deltacc[(q * N + i) * L + k] = rr.deltacc[(q * N + i + 1) * L + k];
        for (int s = 0; s < 4; ++s) w.deltacc_front[((q * N + i) * L + k) * 4 + s] = rr.deltacc_front[((q * N + i + 1) * L + k) * 4 + s];
      }
      for (int k = 0; k < L; ++k) {
        w.deltacc[(q * N + N - 1) * L + k] = rr.deltacc[(q * N + N - 1) * L + k];
        for (int s = 0; s < 4; ++s) w.deltacc_front[((q * N + N - 1) * L + k) * 4 + s] = rr.deltacc_front[((q * N + N - 1) * L + k) * 4 + s];
      }
    }
  }
}

}  // namespace miqp
