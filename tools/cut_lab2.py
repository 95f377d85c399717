"""Round 6, review item 2: the cut lab again with a velocity set that EXCLUDES the apex of the sectors - the incumbent-cutoff ellipsoid (no GPU).

With the multipliers of a solved node fixed its Lagrangian on the trajectories of the dynamics is L(z) = D + 1/2 (z - z*)' H (z - z*) (H: the objective's
Hessian); every descendant that can still improve the incumbent has L(z) <= cutoff, hence per (car, step) |v - v*| <= sqrt(2 (cutoff - D) Sigma_vv) with
Sigma = Z H^-1 Z' the response of that velocity component - a box around the node's own velocity, rigorous and free (the lifting tables hold Sigma).  With it
the per-sector sets (sector ^ box) x (acceleration / jerk box) need not share the apex v = 0 any more, and the support cuts of tools/cut_lab.py
(alpha.v + beta.(a|u) <= max_j [h_Vj(alpha) + h_Bj(beta)]) can bite.  Nodes: every car/car and environment disjunction as in the optimum, the regions of
steps 1..k fixed as in the optimum, the later ones undecided (k = 0: the node of round 5's lab; larger k: deeper in the tree, a smaller gap to the optimum);
budget cutoff - D = (optimum - D) x (1 + 1e-3): the smallest budget that keeps the optimum.

    python tools/cut_lab2.py profiles/r05_hard_solutions.json cfg3 3314 1913 307 243
"""
import json, math, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, HERE)
from miqp_py.bnb import DModel, solve_qp
from miqp_py.model import VX, VY
from ipm_lab import instance
from cut_lab import reach_boxes, sector_polygon, node_rows, support_cuts


def main():
    sols = json.load(open(sys.argv[1])); cfg = sys.argv[2]
    for s in sys.argv[3:]:
        if s not in sols: print("seed", s, "not in dump"); continue
        sol = sols[s]
        I = instance(cfg, int(s)); M = DModel(I)
        Hi = np.linalg.inv(M.H)
        rboxes = reach_boxes(I)
        poss = [[j for j in range(I.R) if I.possible[c, j]] for c in range(I.C)]
        sets = [[[j for j in poss[c] if sector_polygon(I, j, rboxes[c, i]) or sector_polygon(I, j, rboxes[c, i], slow=True)] for i in range(I.N)] for c in range(I.C)]
        r_opt = solve_qp(M, node_rows(M, sol, regions=True), tol=1e-9, maxit=100)
        opt = r_opt["obj"]
        bd = []
        for j in range(I.R):
            F = I.frac[j]; n1 = math.hypot(F[0], F[1]); bd += [(F[1] / n1, -F[0] / n1), (-F[1] / n1, F[0] / n1)]
        print("seed %s: optimum %.4f" % (s, opt), flush=True)
        for k in (0, 4, 8, 12, 15):
            r0 = solve_qp(M, node_rows(M, sol, regions=(k if k > 0 else None), sets=sets), tol=1e-9, maxit=120)
            D = r0["obj"]; gap = (opt - D) / opt
            if gap < 1e-4:
                print("   k = %2d: node gap %.3f %% - nothing left to close" % (k, 100 * gap)); continue
            line = "   k = %2d: node value %.4f, gap to the optimum %.2f %%;" % (k, D, 100 * gap)
            for mult, name in ((1.001, "smallest valid budget"), (1.5, "1.5 x that")):
                budget = (opt - D) * mult
                boxes = rboxes.copy(); widths = []
                for c in range(I.C):
                    for i in range(1, I.N):
                        z = r0["Z"][i]
                        for ax, col in ((0, 6 * c + VX), (1, 6 * c + VY)):
                            g = M.ZU[i][col]; sig = float(g @ Hi @ g)
                            w = math.sqrt(2.0 * budget * sig)
                            boxes[c, i, 2 * ax] = max(rboxes[c, i, 2 * ax], z[col] - w); boxes[c, i, 2 * ax + 1] = min(rboxes[c, i, 2 * ax + 1], z[col] + w)
                            widths.append(w)
                # the sets the interval leaves (undecided steps only): sectors that still meet the box
                sets2 = [[[j for j in sets[c][i] if (i <= k and j == sol["region"][c][i]) or (i > k and (sector_polygon(I, j, boxes[c, i]) or sector_polygon(I, j, boxes[c, i], slow=True)))] or sets[c][i] for i in range(I.N)] for c in range(I.C)]
                nsets = np.mean([len(sets2[c][i]) for c in range(I.C) for i in range(k + 1, I.N)])
                r_s = solve_qp(M, node_rows(M, sol, regions=(k if k > 0 else None), sets=sets2), tol=1e-9, maxit=120)
                cuts = [q for q in support_cuts(M, sets2, boxes, bd, slow_ok=False) if q[0] > k]
                r_c = solve_qp(M, node_rows(M, sol, regions=(k if k > 0 else None), sets=sets2, cuts=cuts), tol=1e-9, maxit=200)
                line += " [%s: half-widths %.2f..%.2f m/s, %.1f regions left per undecided site: sets alone close %.1f %%, sets + %d support cuts %.1f %% of the node gap%s]" % (
                    name, min(widths), max(widths), nsets, 100 * (r_s["obj"] - D) / (opt - D), len(cuts), 100 * (r_c["obj"] - D) / (opt - D), "" if r_c["ok"] else " (not converged)")
            print(line, flush=True)


if __name__ == "__main__":
    main()
