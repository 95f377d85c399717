"""Offline lab for valid inequalities on the region disjunction (no GPU): how much of the gap between the relaxation of a node
with undecided regions and the optimum do candidate cuts close?  Uses the dense numpy model of tools/miqp_py and the optimal
solutions dumped by tools/dump_hard.py.

    python tools/cut_lab.py gpurun_out/r05a/hard.json cfg3 seed [seed ...]

For each seed: the node "every car/car and environment disjunction as in the optimum, every region undecided" is solved
 (a) as the device solves it (hull boxes of the static region sets),
 (b) with candidate cuts added,
 (c) with the regions fixed as in the optimum (= the optimum).
"""
import json, math, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, HERE)
from miqp_py.bnb import DModel, solve_qp, BIGM
from miqp_py.model import PX, VX, AX, PY, VY, AY
from ipm_lab import instance


def reach_boxes(I):
    """per (car, step): velocity box reachable with the global acceleration limits (as host_inst.hpp's reachability presolve)"""
    out = np.zeros((I.C, I.N, 4))
    for c in range(I.C):
        alo = min(I.amin, I.x0[c, AX], I.x0[c, AY]); ahi = max(I.amax, I.x0[c, AX], I.x0[c, AY])
        for i in range(I.N):
            t = i * I.ts
            out[c, i] = [max(I.vmin, I.x0[c, VX] + t * alo), min(I.vmax, I.x0[c, VX] + t * ahi), max(I.vmin, I.x0[c, VY] + t * alo), I.x0[c, VY] + t * ahi]
    return out


def sector_polygon(I, j, box, slow=False):
    """vertices of sector j (or the slow square) inside the velocity box"""
    rows = [(-1, 0, -box[0]), (1, 0, box[1]), (0, -1, -box[2]), (0, 1, box[3])]
    if slow:
        rows += [(1, 0, I.vm), (-1, 0, I.vm), (0, 1, I.vm), (0, -1, I.vm)]
    else:
        F = I.frac[j]
        rows += [(F[1], -F[0], 0.0), (-F[3], F[2], 0.0)]
    V = []
    for a in range(len(rows)):
        for b in range(a + 1, len(rows)):
            det = rows[a][0] * rows[b][1] - rows[a][1] * rows[b][0]
            if abs(det) < 1e-14: continue
            x = (rows[a][2] * rows[b][1] - rows[a][1] * rows[b][2]) / det
            y = (rows[a][0] * rows[b][2] - rows[a][2] * rows[b][0]) / det
            if all(r[0] * x + r[1] * y <= r[2] + 1e-9 * (1 + abs(r[2])) for r in rows):
                V.append((x, y))
    return V


def node_rows(M, sol, regions=None, sets=None, cuts=None, extra=()):
    """rows of the node: c2c (rear/rear only unless regions are given) + env rear as in `sol`; regions: None = undecided (hull of sets[c][i]),
    True = all fixed as in the optimum, an int k = the regions of steps 1..k fixed, the later ones undecided (round 6: nodes deeper in the tree)"""
    I = M.I; rows = []
    reg = sol["region"]
    fixed = lambda i: regions is True or (regions is not None and regions is not False and not isinstance(regions, bool) and i <= regions)
    for c in range(I.C):
        for i in range(I.N):
            rows += M.global_rows(c, i)
    # car/car
    c2c = np.array(sol["c2c"])
    for p, (c1, c2) in enumerate(M.pairs):
        for i in range(1, I.N):
            for g in range(4):
                if not fixed(i) and g != 0 and 'c2cfront' not in extra: continue
                alt = next((a for a in range(4) if c2c[c1, c2 - 1, i, 4 * g + a] == 0), None)
                if alt is None: continue
                rows += M.c2c_rows(p, i, g, alt, reg[c1][i], reg[c2][i])
    env = np.array(sol["env"])   # [5][C][E][N]
    for c in range(I.C):
        for i in range(1, I.N):
            for pt in range(5):
                if not fixed(i) and pt != 0 and 'envfront' not in extra: continue
                e = next((e for e in range(I.E) if env[pt][c][e][i] == 0), None)
                if e is None: continue
                rows += M.env_rows(c, i, pt, e, reg[c][i])
    rc = np.array(sol["rc"])
    for c in range(I.C):
        o = 6 * c; u = 6 * I.C + 2 * c
        for i in range(1, I.N):
            if fixed(i):
                j = reg[c][i]
                if rc[4][c][i] == 1: h = -1
                else:
                    h = 0
                    for k, (ax, sg) in enumerate(M._nonslow[j]):
                        b = (rc[0] if sg > 0 else rc[2]) if ax == 0 else (rc[1] if sg > 0 else rc[3])
                        if b[c][i] == 0: h = k; break
                rows += M.region_rows(c, i, (j, h))
            else:
                if 'sector' in extra or 'curv' in extra:
                    j = reg[c][i]
                    if rc[4][c][i] == 1: h = -1
                    else:
                        h = 0
                        for k, (ax, sg) in enumerate(M._nonslow[j]):
                            b = (rc[0] if sg > 0 else rc[2]) if ax == 0 else (rc[1] if sg > 0 else rc[3])
                            if b[c][i] == 0: h = k; break
                    rr = M.region_rows(c, i, (j, h))
                    if h >= 0:
                        if 'sector' in extra: rows += rr[0:3]
                        if 'curv' in extra: rows += rr[3:5]
                    elif 'sector' in extra: rows += rr[0:4]
                S = sets[c][i]
                for s, st in ((0, AX), (1, AY)):
                    hi = max(I.acc_lim[c, j, 2 * s + 1] for j in S); lo = min(I.acc_lim[c, j, 2 * s] for j in S)
                    rows += [M.lin_row(i, [(o + st, 1)], min(hi, I.amax)), M.lin_row(i, [(o + st, -1)], -max(lo, I.amin))]
                if i <= I.N - 2:
                    for s in (0, 1):
                        hi = max(I.jerk_lim[c, j, 2 * s + 1] for j in S); lo = min(I.jerk_lim[c, j, 2 * s] for j in S)
                        rows += [M.lin_row(i, [(u + s, 1)], min(hi, I.jmax)), M.lin_row(i, [(u + s, -1)], -max(lo, I.jmin))]
    if cuts:
        rows += cuts
    return rows


def support_cuts(M, sets, boxes, dirs_v, kinds=("ax+", "ax-", "ay+", "ay-", "ux+", "ux-", "uy+", "uy-"), slow_ok=True):
    """cuts  alpha.v + beta.(a|u) <= max_j [ h_{V_j}(alpha) + h_{B_j}(beta) ]  for beta = a unit direction of the acceleration / jerk box and alpha from
    `dirs_v` scaled so that the cut is tight at two regions (enumerated: for every pair of regions (j1, j2) of the set and every beta the
    alpha that equalises the two)"""
    I = M.I; cuts = []
    for c in range(I.C):
        o = 6 * c; u = 6 * I.C + 2 * c
        for i in range(1, I.N):
            S = sets[c][i]
            if len(S) < 2: continue
            polys = {j: sector_polygon(I, j, boxes[c, i]) for j in S}
            slowp = sector_polygon(I, 0, boxes[c, i], slow=True) if slow_ok else []
            S2 = [j for j in S if polys[j]]
            for kind in kinds:
                var = kind[:2]; sg = 1.0 if kind[2] == "+" else -1.0
                if var[0] == "u" and i > I.N - 2: continue
                col = {"ax": o + AX, "ay": o + AY, "ux": u, "uy": u + 1}[var]
                def hB(j):
                    lim = I.acc_lim if var[0] == "a" else I.jerk_lim
                    s = 0 if var[1] == "x" else 1
                    return lim[c, j, 2 * s + 1] if sg > 0 else -lim[c, j, 2 * s]
                hmax = max(hB(j) for j in S2)
                if all(abs(hB(j) - hmax) < 1e-9 for j in S2): continue
                for d in dirs_v:
                    # alpha = t * d, t >= 0: rhs(t) = max_j [ t * h_Vj(d) + hB(j) ]; (the slow square keeps every region's box: its support joins every j)
                    hv = {j: max(d[0] * x + d[1] * y for (x, y) in polys[j]) for j in S2}
                    if slowp:
                        hs = max(d[0] * x + d[1] * y for (x, y) in slowp)
                        hv = {j: max(hv[j], hs) for j in S2}
                    # candidate t: breakpoints where two regions' lines cross
                    ts_ = set()
                    for a in S2:
                        for b in S2:
                            if a < b and abs(hv[a] - hv[b]) > 1e-12:
                                t = (hB(b) - hB(a)) / (hv[a] - hv[b])
                                if t > 1e-9: ts_.add(t)
                    for t in ts_:
                        rhs = max(t * hv[j] + hB(j) for j in S2)
                        cuts.append(M.lin_row(i, [(o + VX, t * d[0]), (o + VY, t * d[1]), (col, sg)], rhs))
    return cuts


def main():
    sols = json.load(open(sys.argv[1])); cfg = sys.argv[2]
    for s in sys.argv[3:]:
        if s not in sols: print("seed", s, "not in dump"); continue
        sol = sols[s]
        I = instance(cfg, int(s)); M = DModel(I)
        boxes = reach_boxes(I)
        poss = [[j for j in range(I.R) if I.possible[c, j]] for c in range(I.C)]
        sets = [[[j for j in poss[c] if sector_polygon(I, j, boxes[c, i]) or sector_polygon(I, j, boxes[c, i], slow=True)] for i in range(I.N)] for c in range(I.C)]
        r_opt = solve_qp(M, node_rows(M, sol, regions=True), tol=1e-9, maxit=100)
        r_hull = solve_qp(M, node_rows(M, sol, sets=sets), tol=1e-9, maxit=100)
        print("seed %s: device optimum %.4f | all fixed: %.4f (viol %.1e) | regions undecided, hull boxes: %.4f  -> gap to close %.3f %%" % (s, sol["obj"], r_opt["obj"], r_opt["viol"], r_hull["obj"], 100 * (r_opt["obj"] - r_hull["obj"]) / r_opt["obj"]))
        print("   regions of the optimum:", sol["region"])
        sets0 = [[[sol["region"][c][i]] for i in range(I.N)] for c in range(I.C)]
        for ex in ((), ("sector",), ("curv",), ("c2cfront",), ("envfront",), ("sector", "curv", "c2cfront", "envfront")):
            ra = solve_qp(M, node_rows(M, sol, sets=sets, extra=ex), tol=1e-9, maxit=150)
            rb = solve_qp(M, node_rows(M, sol, sets=sets0, extra=ex), tol=1e-9, maxit=150)
            print("   undecided + %-40s: set hull boxes %.4f (gap %.3f %%) | exact region's boxes %.4f (gap %.3f %%)" % ("+".join(ex) or "nothing", ra["obj"], 100 * (r_opt["obj"] - ra["obj"]) / r_opt["obj"], rb["obj"], 100 * (r_opt["obj"] - rb["obj"]) / r_opt["obj"]))
        if os.environ.get("CUTS", "0") != "1": continue
        nd = 16
        dirs = [(math.cos(2 * math.pi * k / nd), math.sin(2 * math.pi * k / nd)) for k in range(nd)]
        # sector-border normals as directions
        bd = []
        for j in range(I.R):
            F = I.frac[j]; n1 = math.hypot(F[0], F[1]); bd += [(F[1] / n1, -F[0] / n1), (-F[1] / n1, F[0] / n1)]
        for name, dv in (("16 directions", dirs), ("border normals", bd)):
            cuts = support_cuts(M, sets, boxes, dv)
            r_c = solve_qp(M, node_rows(M, sol, sets=sets, cuts=cuts), tol=1e-9, maxit=150)
            print("   + support cuts (%s, %d rows): %.4f  closes %.1f %% of the gap (ok %s)" % (name, len(cuts), r_c["obj"], 100 * (r_c["obj"] - r_hull["obj"]) / max(1e-12, r_opt["obj"] - r_hull["obj"]), r_c["ok"]))
        # the same with the sets restricted to the optimum's region and its neighbours in the set (what the tree's set tightening reaches)
        for w in (1, 0):
            sets2 = [[[j for j in sets[c][i] if min((j - sol["region"][c][i]) % I.R, (sol["region"][c][i] - j) % I.R) <= w] or sets[c][i] for i in range(I.N)] for c in range(I.C)]
            r_h2 = solve_qp(M, node_rows(M, sol, sets=sets2), tol=1e-9, maxit=100)
            cuts = support_cuts(M, sets2, boxes, bd)
            r_c2 = solve_qp(M, node_rows(M, sol, sets=sets2, cuts=cuts), tol=1e-9, maxit=150)
            print("   sets within %d of the optimum's region: hull %.4f (gap %.3f %%), with cuts %.4f (gap %.3f %%)" % (w, r_h2["obj"], 100 * (r_opt["obj"] - r_h2["obj"]) / r_opt["obj"], r_c2["obj"], 100 * (r_opt["obj"] - r_c2["obj"]) / r_opt["obj"]))




# ---------------------------------------------------------------- relaxed front-point rows (regions undecided)
def relaxed_c2c_rows(M, p, i, grp, alt, S1, S2, ref1, ref2, boxes, mode):
    """the rows of (pair, step, group, alternative) with the front-point offsets of cars whose region is undecided replaced by an envelope that
    holds for every region of the car's set S on that region's velocity set: mode 'const' = constant bounds, 'ref' = the polynomial of region `ref`
    shifted by its largest deviation from the other regions' polynomials"""
    I = M.I; c1, c2 = M.pairs[p]
    D = I.rad[c1] + I.rad[c2] + I.safety[i]; Ssl = I.safety_slack[i]
    isx = alt < 2
    R, U, L = 0, 1, 2
    if grp == 0: A_, B_ = ((c1, R), (c2, R)) if alt in (0, 2) else ((c2, R), (c1, R)); soft = True
    elif grp == 1: (A_, B_) = ((c1, R), (c2, L)) if alt in (0, 2) else ((c2, U), (c1, R)); soft = False
    elif grp == 2: (A_, B_) = ((c2, R), (c1, L)) if alt in (0, 2) else ((c1, U), (c2, R)); soft = False
    else: (A_, B_) = ((c2, U), (c1, L)) if alt in (0, 2) else ((c1, U), (c2, L)); soft = True
    co = np.zeros(M.nz); rhs_shift = 0.0
    for (c, t), sg in ((A_, 1.0), (B_, -1.0)):
        o = 6 * c
        co[o + (PX if isx else PY)] += sg
        if t == R: continue
        S = S1 if c == c1 else S2; ref = ref1 if c == c1 else ref2
        key = ("COSS_" if isx else "SINT_") + ("UB" if t == U else "LB")
        def off(j, v): return I.wb[c] * (I.poly[key][j] @ np.array([1.0, v[0], v[1]]))
        polys = {j: (sector_polygon(I, j, boxes[c, i]) + sector_polygon(I, j, boxes[c, i], slow=True)) for j in S}
        if mode == "const":
            vals = [off(j, v) for j in S for v in polys[j]]
            bound = min(vals) if sg > 0 else max(vals)      # row is lhs <= rhs: a positive coefficient needs the lower envelope
            rhs_shift += sg * bound
        else:
            devs = [off(j, v) - off(ref, v) for j in S for v in polys[j]]
            d = min(devs) if sg > 0 else max(devs)
            pr = I.wb[c] * I.poly[key][ref]
            rhs_shift += sg * (pr[0] + d); co[o + VX] += sg * pr[1]; co[o + VY] += sg * pr[2]
    if not soft:
        return [(i, co, -D - rhs_shift, 0.0)]
    smax = min(Ssl, I.max_slack)
    rows = [(i, co, -(D + Ssl) + smax - rhs_shift, 0.0)]
    if smax > 0 and I.w_slack > 0: rows.append((i, co, -(D + Ssl) - rhs_shift, 2 * I.w_slack))
    return rows


def lab_front(sols, cfg, seeds):
    for s in seeds:
        sol = sols[s]; I = instance(cfg, int(s)); M = DModel(I); boxes = reach_boxes(I)
        poss = [[j for j in range(I.R) if I.possible[c, j]] for c in range(I.C)]
        sets = [[[j for j in poss[c] if sector_polygon(I, j, boxes[c, i]) or sector_polygon(I, j, boxes[c, i], slow=True)] for i in range(I.N)] for c in range(I.C)]
        base = node_rows(M, sol, sets=sets)
        r_opt = solve_qp(M, node_rows(M, sol, regions=True), tol=1e-9, maxit=100)
        r0 = solve_qp(M, base, tol=1e-9, maxit=100)
        c2c = np.array(sol["c2c"]); reg = sol["region"]
        out = ["seed %s: optimum %.3f, undecided %.3f (gap %.2f %%)" % (s, r_opt["obj"], r0["obj"], 100 * (r_opt["obj"] - r0["obj"]) / r_opt["obj"])]
        for mode in ("const", "ref"):
            for near in (None, 1):
                extra = []
                for p, (c1, c2) in enumerate(M.pairs):
                    for i in range(1, I.N):
                        S1, S2 = sets[c1][i], sets[c2][i]
                        if near is not None:
                            S1 = [j for j in S1 if min((j - reg[c1][i]) % I.R, (reg[c1][i] - j) % I.R) <= near] or S1
                            S2 = [j for j in S2 if min((j - reg[c2][i]) % I.R, (reg[c2][i] - j) % I.R) <= near] or S2
                        for g in (1, 2, 3):
                            alt = next((a for a in range(4) if c2c[c1, c2 - 1, i, 4 * g + a] == 0), None)
                            if alt is None: continue
                            extra += relaxed_c2c_rows(M, p, i, g, alt, S1, S2, reg[c1][i], reg[c2][i], boxes, mode)
                r = solve_qp(M, base + extra, tol=1e-9, maxit=150)
                out.append("   relaxed front rows (%s, sets %s): %.3f (gap %.2f %%)" % (mode, "static" if near is None else "within %d of the optimum" % near, r["obj"], 100 * (r_opt["obj"] - r["obj"]) / r_opt["obj"]))
        print("\n".join(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("FRONT") == "1": lab_front(json.load(open(sys.argv[1])), sys.argv[2], sys.argv[3:])
    else: main()
