// lp_export.hpp - CPLEX LP-format dump of the raw big-M model of one instance (debug / interoperability format,
// replaces cplex.exportModel, src/cplex_wrapper.cpp:150-154).  Row by row after cplexmodel/*.mod so that anyone with a
// CPLEX licence can re-solve the instance; duplicate rows and `==` fixings are kept as rows.
#pragma once
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "host_inst.hpp"

namespace miqp {

struct LpWriter {
  const HostInst& I; FILE* f; long rows = 0;
  struct Term { std::string v; double c; };
  static std::string nm(const char* base, std::initializer_list<int> idx) {
    std::string s = base; for (int k : idx) { s += "#"; s += std::to_string(k + 1); } return s;
  }
  void row(std::vector<Term> lhs, const char* sense, double rhs) {  // terms may repeat; constants already folded into rhs
    std::vector<Term> t;
    for (auto& a : lhs) { bool hit = false; for (auto& b : t) if (b.v == a.v) { b.c += a.c; hit = true; } if (!hit) t.push_back(a); }
    std::fprintf(f, " c%ld:", ++rows);
    bool any = false;
    for (auto& a : t) if (a.c != 0.0) { std::fprintf(f, " %+.17g %s", a.c, a.v.c_str()); any = true; }
    if (!any) std::fprintf(f, " 0 %s", t.empty() ? "dummy" : t[0].v.c_str());
    std::fprintf(f, " %s %.17g\n", sense, rhs);
  }
  void cross(std::vector<Term>& t, double& k, const double* e, const std::string& X, const std::string& Y) {
    double dx = e[2] - e[0], dy = e[3] - e[1];
    t.push_back({Y, dx}); t.push_back({X, -dy}); k = -dx * e[1] + e[0] * dy;   // cross = sum + k
  }
  bool write() {
    const int C = I.C, N = I.N, R = I.R, E = I.E, O = I.O, L = I.L, K = C - 1; const double ts = I.ts;
    auto V = [&](const char* b, int c, int i) { return nm(b, {c, i}); };
    std::fprintf(f, "\\ planner-miqp MIQP (cplexmodel/*.mod), written by libmiqp_gpu\nMinimize\n obj:");
    // linear part of sum W (v - ref)^2, quadratic part below; the constant sum W ref^2 is omitted (objective_function.mod:7-19)
    const char* sv[6] = {"pos_x", "vel_x", "acc_x", "pos_y", "vel_y", "acc_y"};
    for (int c = 0; c < C; ++c) for (int i = 0; i < N; ++i) for (int k = 0; k < 6; ++k) {
      double w = I.W[c * 8 + k], r = I.ref[((size_t)c * N + i) * 6 + k];
      if (w != 0.0 && r != 0.0) std::fprintf(f, " %+.17g %s", -2.0 * w * r, V(sv[k], c, i).c_str());
    }
    std::fprintf(f, " + [");
    for (int c = 0; c < C; ++c) for (int i = 0; i < N; ++i) {
      for (int k = 0; k < 6; ++k) if (I.W[c * 8 + k] != 0.0) std::fprintf(f, " %+.17g %s ^2", 2.0 * I.W[c * 8 + k], V(sv[k], c, i).c_str());
      if (I.W[c * 8 + 6] != 0.0) std::fprintf(f, " %+.17g %s ^2", 2.0 * I.W[c * 8 + 6], V("u_x", c, i).c_str());
      if (I.W[c * 8 + 7] != 0.0) std::fprintf(f, " %+.17g %s ^2", 2.0 * I.W[c * 8 + 7], V("u_y", c, i).c_str());
      for (int o = 0; o < O; ++o) {
        std::fprintf(f, " %+.17g %s ^2", 2.0 * I.w_slack_obs, nm("slackvarsObstacle", {c, o, i}).c_str());
        for (int q = 0; q < 4; ++q) std::fprintf(f, " %+.17g %s ^2", 2.0 * I.w_slack_obs, nm("slackvarsObstacle_front", {c, o, i, q}).c_str());
      }
    }
    for (int a = 0; a < K; ++a) for (int b = 0; b < K; ++b) for (int i = 0; i < N; ++i) for (int q = 0; q < 4; ++q)
      std::fprintf(f, " %+.17g %s ^2", 2.0 * I.w_slack, nm("slackvars", {a, b, i, q}).c_str());
    std::fprintf(f, " ] / 2\nSubject To\n");
    // A1 initial_conditions.mod:11-61
    for (int c = 0; c < C; ++c) {
      for (int k = 0; k < 6; ++k) row({{V(sv[k], c, 0), 1}}, "=", I.x0[c * 6 + k]);
      double th = std::atan2(I.x0[c * 6 + 4], I.x0[c * 6 + 1]);
      double fx = I.x0[c * 6] + std::cos(th) * I.wb[c], fy = I.x0[c * 6 + 3] + std::sin(th) * I.wb[c];
      row({{V("pos_x_front_UB", c, 0), 1}}, "=", fx); row({{V("pos_x_front_LB", c, 0), 1}}, "=", fx);
      row({{V("pos_y_front_UB", c, 0), 1}}, "=", fy); row({{V("pos_y_front_LB", c, 0), 1}}, "=", fy);
      row({{V("u_x", c, N - 1), 1}}, "=", 0); row({{V("u_y", c, N - 1), 1}}, "=", 0);
    }
    for (int j = 0; j < R; ++j) for (int c = 0; c < C; ++c) {
      std::string ar = nm("active_region", {c, 0, j});
      row({{ar, 1}}, "=", (j + 1 == I.init_region[c]) ? 1 : 0);
      const double* jl = &I.jerk_lim[((size_t)c * R + j) * 4];
      for (int ax = 0; ax < 2; ++ax) {
        std::string u = V(ax ? "u_y" : "u_x", c, 0);
        row({{u, 1}, {ar, 10.0}}, "<=", jl[2 * ax + 1] + 10.0);
        row({{u, 1}, {ar, -10.0}}, ">=", jl[2 * ax] - 10.0);
      }
    }
    const char* rc[5] = {"region_change_not_allowed_x_positive", "region_change_not_allowed_y_positive", "region_change_not_allowed_x_negative",
                         "region_change_not_allowed_y_negative", "region_change_not_allowed_combined"};
    for (int c = 0; c < C; ++c) for (int k = 0; k < 5; ++k) row({{V(rc[k], c, 0), 1}}, "=", 0);
    // A2 dynamics model_region_constraints.mod:11-19
    for (int i = 1; i < N; ++i) for (int c = 0; c < C; ++c) for (int ax = 0; ax < 2; ++ax) {
      const char *P_ = ax ? "pos_y" : "pos_x", *V_ = ax ? "vel_y" : "vel_x", *A_ = ax ? "acc_y" : "acc_x", *U_ = ax ? "u_y" : "u_x";
      row({{V(P_, c, i), 1}, {V(P_, c, i - 1), -1}, {V(V_, c, i - 1), -ts}, {V(A_, c, i - 1), -0.5 * ts * ts}, {V(U_, c, i - 1), -ts * ts * ts / 6.0}}, "=", 0);
      row({{V(V_, c, i), 1}, {V(V_, c, i - 1), -1}, {V(A_, c, i - 1), -ts}, {V(U_, c, i - 1), -0.5 * ts * ts}}, "=", 0);
      row({{V(A_, c, i), 1}, {V(A_, c, i - 1), -1}, {V(U_, c, i - 1), -ts}}, "=", 0);
    }
    // A3 :22-39 (the vel_x upper row appears twice, vel_y has none: kept)
    for (int i = 0; i < N; ++i) for (int c = 0; c < C; ++c) {
      row({{V("vel_x", c, i), 1}}, ">=", I.vmin); row({{V("vel_y", c, i), 1}}, ">=", I.vmin);
      row({{V("vel_x", c, i), 1}}, "<=", I.vmax); row({{V("vel_x", c, i), 1}}, "<=", I.vmax);
      row({{V("acc_x", c, i), 1}}, "<=", I.amax); row({{V("acc_x", c, i), 1}}, ">=", I.amin);
      row({{V("acc_y", c, i), 1}}, "<=", I.amax); row({{V("acc_y", c, i), 1}}, ">=", I.amin);
      row({{V("u_x", c, i), 1}}, "<=", I.jmax); row({{V("u_x", c, i), 1}}, ">=", I.jmin);
      row({{V("u_y", c, i), 1}}, "<=", I.jmax); row({{V("u_y", c, i), 1}}, ">=", I.jmin);
    }
    // A4 :43-117
    for (int i = 1; i < N; ++i) for (int c = 0; c < C; ++c) {
      std::string vx = V("vel_x", c, i), vy = V("vel_y", c, i), ax_ = V("acc_x", c, i), ay_ = V("acc_y", c, i), q = V(rc[4], c, i);
      std::vector<Term> sum;
      for (int j = 0; j < R; ++j) {
        std::string ar = nm("active_region", {c, i, j});
        sum.push_back({ar, 1});
        if (I.possible[c * R + j] != 1) { row({{ar, 1}}, "=", 0); continue; }
        const double* F = &I.frac[j * 4];
        row({{vy, F[0]}, {vx, -F[1]}, {ar, -1000.0}, {q, 1000.0}}, ">=", -1000.0);
        row({{vy, F[2]}, {vx, -F[3]}, {ar, 1000.0}, {q, -1000.0}}, "<=", 1000.0);
        const char* fv[4] = {"pos_x_front_UB", "pos_x_front_LB", "pos_y_front_UB", "pos_y_front_LB"};
        const char* pv[4] = {"pos_x", "pos_x", "pos_y", "pos_y"}; const int pt[4] = {2, 3, 0, 1};
        for (int k = 0; k < 4; ++k) {
          const double* p = &I.poly[pt[k]][j * 3]; double wb = I.wb[c];
          row({{V(fv[k], c, i), 1}, {V(pv[k], c, i), -1}, {vx, -wb * p[1]}, {vy, -wb * p[2]}, {ar, -100.0}}, ">=", wb * p[0] - 100.0);
          row({{V(fv[k], c, i), 1}, {V(pv[k], c, i), -1}, {vx, -wb * p[1]}, {vy, -wb * p[2]}, {ar, 100.0}}, "<=", wb * p[0] + 100.0);
        }
        const double* jl = &I.jerk_lim[((size_t)c * R + j) * 4]; const double* al = &I.acc_lim[((size_t)c * R + j) * 4];
        for (int a = 0; a < 2; ++a) { std::string u = V(a ? "u_y" : "u_x", c, i); row({{u, 1}, {ar, 10.0}}, "<=", jl[2 * a + 1] + 10.0); row({{u, 1}, {ar, -10.0}}, ">=", jl[2 * a] - 10.0); }
        for (int a = 0; a < 2; ++a) { std::string av = a ? ay_ : ax_; row({{av, 1}, {ar, 10.0}}, "<=", al[2 * a + 1] + 10.0); row({{av, 1}, {ar, -10.0}}, ">=", al[2 * a] - 10.0); }
        double rho = (F[1] + F[3]) / (F[0] + F[2]); const double* kx = &I.poly[4][j * 3]; const double* kn = &I.poly[5][j * 3];
        row({{ay_, 1}, {vx, -kx[1]}, {vy, -kx[2]}, {ax_, -rho}, {ar, 1000.0}, {q, -1000.0}}, "<=", kx[0] + 1000.0);
        row({{ay_, 1}, {vx, -kn[1]}, {vy, -kn[2]}, {ax_, -rho}, {ar, -1000.0}, {q, 1000.0}}, ">=", kn[0] - 1000.0);
      }
      row(sum, "=", 1);
    }
    // A5 minimum_speed_constraints.mod:9-49 (emitted once per region, as OPL does)
    for (int i = 1; i < N; ++i) for (int c = 0; c < C; ++c) {
      std::string vx = V("vel_x", c, i), vy = V("vel_y", c, i), xp = V(rc[0], c, i), yp = V(rc[1], c, i), xn = V(rc[2], c, i), yn = V(rc[3], c, i), cb = V(rc[4], c, i);
      for (int j = 0; j < R; ++j) {
        for (int a = 0; a < 2; ++a) {
          std::string v = a ? vy : vx, p = a ? yp : xp, n_ = a ? yn : xn;
          row({{v, 1}, {p, 100.0}}, ">=", I.vm); row({{v, 1}, {p, 100.0}}, "<=", I.vm + 100.0);
          row({{v, -1}, {n_, 100.0}}, "<=", I.vm + 100.0); row({{v, -1}, {n_, 100.0}}, ">=", I.vm);
        }
        row({{nm("active_region", {c, i, j}), 1}, {nm("active_region", {c, i - 1, j}), -1}, {cb, 1}}, "<=", 1);
        row({{nm("active_region", {c, i, j}), 1}, {nm("active_region", {c, i - 1, j}), -1}, {cb, -1}}, ">=", -1);
        row({{cb, 1}, {xp, -1}}, "<=", 0); row({{cb, 1}, {yp, -1}}, "<=", 0); row({{cb, 1}, {xn, -1}}, "<=", 0); row({{cb, 1}, {yn, -1}}, "<=", 0);
        row({{cb, 1}, {xp, -1}, {yp, -1}, {xn, -1}, {yn, -1}}, ">=", -3);
      }
    }
    // A6 obstacle_environment_constraints.mod:6-47
    const char* nwn[5] = {"notWithinEnvironmentRear", "notWithinEnvironmentFrontUbUb", "notWithinEnvironmentFrontLbUb", "notWithinEnvironmentFrontUbLb", "notWithinEnvironmentFrontLbLb"};
    const char* ex[5] = {"pos_x", "pos_x_front_UB", "pos_x_front_LB", "pos_x_front_UB", "pos_x_front_LB"};
    const char* ey[5] = {"pos_y", "pos_y_front_UB", "pos_y_front_UB", "pos_y_front_LB", "pos_y_front_LB"};
    if (E > 0)
      for (int i = 0; i < N; ++i) for (int c = 0; c < C; ++c) {
        for (int e = 0; e < E; ++e) for (int k = I.env_off[e]; k < I.env_off[e + 1]; ++k) for (int p = 0; p < 5; ++p) {
          std::vector<Term> t; double k0; cross(t, k0, &I.env_edges[(size_t)k * 4], V(ex[p], c, i), V(ey[p], c, i));
          t.push_back({nm(nwn[p], {c, e, i}), 10000.0}); row(t, ">=", -k0);
        }
        for (int p = 0; p < 5; ++p) { std::vector<Term> t; for (int e = 0; e < E; ++e) t.push_back({nm(nwn[p], {c, e, i}), 1}); row(t, "<=", E - 1); }
      }
    // A7 :52-109
    const char* ox[5] = {"pos_x", "pos_x_front_LB", "pos_x_front_UB", "pos_x_front_LB", "pos_x_front_UB"};
    const char* oy[5] = {"pos_y", "pos_y_front_LB", "pos_y_front_LB", "pos_y_front_UB", "pos_y_front_UB"};
    for (int i = 0; i < N && O > 0; ++i) for (int c = 0; c < C; ++c) for (int o = 0; o < O; ++o) {
      auto dv = [&](int p, int k) { return p == 0 ? nm("deltacc", {c, o, i, k}) : nm("deltacc_front", {c, o, i, k, p - 1}); };
      for (int k = 0; k < L; ++k) for (int p = 0; p < 5; ++p) {
        std::vector<Term> t; double k0; cross(t, k0, &I.obs_edges[((size_t)(o * N + i) * L + k) * 4], V(ox[p], c, i), V(oy[p], c, i));
        t.push_back({dv(p, k), -10000.0}); row(t, "<=", -k0);
      }
      for (int p = 0; p < 5; ++p) {
        std::vector<Term> t; for (int k = 0; k < L; ++k) t.push_back({dv(p, k), 1});
        if (I.obs_soft[o] == 1) t.push_back({p == 0 ? nm("slackvarsObstacle", {c, o, i}) : nm("slackvarsObstacle_front", {c, o, i, p - 1}), -1});
        row(t, "<=", L - 1);
      }
    }
    // A8 agent_collision_constraints.mod:10-73
    if (C > 1) {
      for (int i = 0; i < N; ++i) for (int c1 = 1; c1 < K; ++c1) for (int c2 = 0; c2 < c1; ++c2) {
        for (int s = 0; s < 4; ++s) row({{nm("slackvars", {c1, c2, i, s}), 1}}, "=", 0);
        for (int s = 0; s < 16; ++s) row({{nm("car2car_collision", {c1, c2, i, s}), 1}}, "=", 0);
      }
      for (int i = 0; i < N; ++i) for (int c1 = 0; c1 < C - 1; ++c1) for (int c2 = c1 + 1; c2 < C; ++c2) {
        double D = I.rad[c1] + I.rad[c2] + I.safety[i], S = I.safety_slack[i]; int q2 = c2 - 1;
        auto cc = [&](int s) { return nm("car2car_collision", {c1, q2, i, s}); };
        auto sl = [&](int s) { return nm("slackvars", {c1, q2, i, s}); };
        struct R_ { std::string lv, rv; int sense, b, sl; };
        std::vector<R_> rw = {
            {V("pos_x", c1, i), V("pos_x", c2, i), -1, 0, 0}, {V("pos_x", c1, i), V("pos_x", c2, i), 1, 1, 0},
            {V("pos_y", c1, i), V("pos_y", c2, i), -1, 2, 1}, {V("pos_y", c1, i), V("pos_y", c2, i), 1, 3, 1},
            {V("pos_x", c1, i), V("pos_x_front_LB", c2, i), -1, 4, -1}, {V("pos_x", c1, i), V("pos_x_front_UB", c2, i), 1, 5, -1},
            {V("pos_y", c1, i), V("pos_y_front_LB", c2, i), -1, 6, -1}, {V("pos_y", c1, i), V("pos_y_front_UB", c2, i), 1, 7, -1},
            {V("pos_x", c2, i), V("pos_x_front_LB", c1, i), -1, 8, -1}, {V("pos_x", c2, i), V("pos_x_front_UB", c1, i), 1, 9, -1},
            {V("pos_y", c2, i), V("pos_y_front_LB", c1, i), -1, 10, -1}, {V("pos_y", c2, i), V("pos_y_front_UB", c1, i), 1, 11, -1},
            {V("pos_x_front_UB", c2, i), V("pos_x_front_LB", c1, i), -1, 12, 2}, {V("pos_x_front_LB", c2, i), V("pos_x_front_UB", c1, i), 1, 13, 2},
            {V("pos_y_front_UB", c2, i), V("pos_y_front_LB", c1, i), -1, 14, 3}, {V("pos_y_front_LB", c2, i), V("pos_y_front_UB", c1, i), 1, 15, 3}};
        for (int g = 0; g < 4; ++g) {
          for (int q = 0; q < 4; ++q) {
            const R_& r = rw[4 * g + q]; double sg = r.sense < 0 ? -1.0 : 1.0;
            std::vector<Term> t = {{r.lv, 1}, {r.rv, -1}, {cc(r.b), sg * 1000.0}};
            if (r.sl >= 0) t.push_back({sl(r.sl), sg});
            row(t, r.sense < 0 ? "<=" : ">=", sg * (D + (r.sl >= 0 ? S : 0.0)));
          }
          row({{cc(4 * g), 1}, {cc(4 * g + 1), 1}, {cc(4 * g + 2), 1}, {cc(4 * g + 3), 1}}, "<=", 3);
          if (g == 0 || g == 3) for (int q = 0; q < 2; ++q) row({{sl((g == 0 ? 0 : 2) + q), 1}}, "<=", S);
        }
      }
    }
    // bounds: OPL dvar float is free; slack ranges of decision_variables.mod:43-53
    std::fprintf(f, "Bounds\n");
    const char* fr[12] = {"u_x", "u_y", "pos_x", "vel_x", "acc_x", "pos_y", "vel_y", "acc_y", "pos_x_front_UB", "pos_x_front_LB", "pos_y_front_UB", "pos_y_front_LB"};
    for (int k = 0; k < 12; ++k) for (int c = 0; c < C; ++c) for (int i = 0; i < N; ++i) std::fprintf(f, " %s free\n", V(fr[k], c, i).c_str());
    for (int c = 0; c < C; ++c) for (int o = 0; o < O; ++o) for (int i = 0; i < N; ++i) {
      std::fprintf(f, " 0 <= %s <= 1\n", nm("slackvarsObstacle", {c, o, i}).c_str());
      for (int q = 0; q < 4; ++q) std::fprintf(f, " 0 <= %s <= 1\n", nm("slackvarsObstacle_front", {c, o, i, q}).c_str());
    }
    for (int a = 0; a < K; ++a) for (int b = 0; b < K; ++b) for (int i = 0; i < N; ++i) for (int q = 0; q < 4; ++q)
      std::fprintf(f, " 0 <= %s <= %.17g\n", nm("slackvars", {a, b, i, q}).c_str(), I.max_slack);
    std::fprintf(f, "Binaries\n");
    long nb = 0;
    auto B = [&](const std::string& s) { std::fprintf(f, " %s\n", s.c_str()); ++nb; };
    for (int p = 0; p < 5; ++p) for (int c = 0; c < C; ++c) for (int e = 0; e < E; ++e) for (int i = 0; i < N; ++i) B(nm(nwn[p], {c, e, i}));
    for (int c = 0; c < C; ++c) for (int i = 0; i < N; ++i) for (int j = 0; j < R; ++j) B(nm("active_region", {c, i, j}));
    for (int k = 0; k < 5; ++k) for (int c = 0; c < C; ++c) for (int i = 0; i < N; ++i) B(V(rc[k], c, i));
    for (int c = 0; c < C; ++c) for (int o = 0; o < O; ++o) for (int i = 0; i < N; ++i) for (int k = 0; k < L; ++k) {
      B(nm("deltacc", {c, o, i, k})); for (int q = 0; q < 4; ++q) B(nm("deltacc_front", {c, o, i, k, q}));
    }
    for (int a = 0; a < K; ++a) for (int b = 0; b < K; ++b) for (int i = 0; i < N; ++i) for (int s = 0; s < 16; ++s) B(nm("car2car_collision", {a, b, i, s}));
    std::fprintf(f, "End\n");
    return true;
  }
};

inline int export_lp(const HostInst& I, const char* path) {
  FILE* f = std::fopen(path, "w");
  if (!f) return -2;
  LpWriter w{I, f};
  w.write();
  std::fclose(f);
  return 0;
}

}  // namespace miqp
