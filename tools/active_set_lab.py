"""Offline lab (no GPU, no oracle): a DUAL ACTIVE-SET (Goldfarb-Idnani) solve of a child's relaxation started from its parent's
optimum, against the interior point the device runs (tools/ipm_lab.py's replica) - the go / no-go asked for in round 6.

The node QP in condensed form (U = all inputs):  min 1/2 U'HU + g'U  s.t.  G U <= h  (elastic rows taken as hard: a node whose rows
cannot hold together is reported infeasible; quadratic-soft rows - the car / car slack - carry their own slack column, cost 1/2 a s^2).
H is the Hessian of the OBJECTIVE ALONE, the same for every node of an instance: on the device H^-1 applied to a sparse row is one
backward + one forward substitution with the constant gains of the unconstrained regulator (no factorisation), and an entry
g_a H^-1 g_b' of the Schur complement S is a sum over <= 6 x 6 entries of the response tables the bound lifting already holds.

A child starts from the parent's active set A0 (multiplier above the slack at the parent's interior point solution: lambda > s) restricted to the rows that the child still has with the
same right-hand side; S_A0 is factored once (Cholesky of |A0|), multipliers that come out negative are dropped (the parent's rows that
left took their share), and from that dual-feasible point Goldfarb-Idnani adds the most violated row, ratio-tests the multipliers and
drops the blocking rows.  Counted per child: |A0|, rows added, rows dropped, and the work in units one interior point iteration of
the device costs (F_iter of DESIGN.md 6 = 217 kflop + 96 flop per row for two cars x 20 steps):
    set-up        |A0|^3 / 3 + 2 |A0|^2 + nnz-products of S_A0 (<= 36 per pair) + one H^-1 application (2 N n_s n_c flop: constant gains)
    per added row one H^-1 application for the new iterate + the row pass that finds the next violated row (2 nnz per row)
                  + 4 |A|^2 for the bordered update and the two triangular solves
    per drop      3 |A|^2 (Givens down-date + re-solve)
and compared with the iterations the interior point replica needs from the parent's solution (the device's start).

    python tools/active_set_lab.py [cfg] [seed] [nodes]
"""
import heapq
import itertools
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)

from ipm_lab import instance, pack, ipm          # noqa: E402
from miqp_py.bnb import BnB, DModel, FEAS_TOL    # noqa: E402

DEP = 1e-8           # a new row whose curvature g P g' falls below this share of g H^-1 g' depends on the active rows
VTOL = 1e-7          # a row counts as violated above this (rows are normalised: metres, m/s, ...)


class Condensed:
    """H^-1 of the objective once per instance (the device: constant regulator gains + response tables)"""

    def __init__(self, M):
        self.M = M
        self.Hi = np.linalg.inv(M.H)
        self.Uunc = -self.Hi @ M.g


def gi_solve(Cd, keys, G, h, av, A0, stats):
    """Goldfarb-Idnani from the active set A0 (indices into the rows).  Returns dict(U, lam, obj, infeasible)."""
    M = Cd.M; Hi = Cd.Hi
    m = len(h)
    soft = av > 0
    dreg = np.where(soft, 1.0 / np.where(soft, av, 1.0), 0.0)      # S_kk gains 1 / a for a quadratic-soft row (its own slack column)
    HiGt = Hi @ G.T                                                  # lab only: the device forms the columns it needs
    Sfull = G @ HiGt + np.diag(dreg)
    c_unc = G @ Cd.Uunc - h                                          # violation of every row at the unconstrained optimum

    def eqp(A):
        if not A:
            return np.zeros(0)
        return np.linalg.solve(Sfull[np.ix_(A, A)] + 1e-13 * np.eye(len(A)), c_unc[A])

    A = list(A0)
    lam = eqp(A)
    stats["chol"] += len(A) ** 3 / 3 + 2 * len(A) ** 2
    stats["setup_pairs"] += len(A) * (len(A) + 1) // 2
    nfix = 0
    while len(A) and lam.min() < -1e-10:                             # rows of the parent that left: drop what went negative, re-solve
        keep = [a for a, l in zip(A, lam) if l >= -1e-10]
        nfix += len(A) - len(keep)
        A = keep; lam = eqp(A)
        stats["flops"] += 3 * len(A) ** 2 * max(1, nfix)
    lam = np.maximum(lam, 0.0)
    stats["fix_drops"] += nfix
    lamv = np.zeros(m); lamv[A] = lam
    adds = drops = 0
    applies = 1
    infeasible = False
    for _ in range(400):
        viol = c_unc - Sfull @ lamv                                  # = G U - h - s at U = Uunc - Hi G' lam (soft rows: s = lam / a)
        viol[A] = -np.inf
        p = int(np.argmax(viol))
        if viol[p] <= VTOL:
            break
        adds += 1
        vp = viol[p]
        while True:
            if A:
                SAA = Sfull[np.ix_(A, A)] + 1e-13 * np.eye(len(A))
                r = np.linalg.solve(SAA, Sfull[A, p])
                curv = Sfull[p, p] - Sfull[p, A] @ r
            else:
                r = np.zeros(0); curv = Sfull[p, p]
            stats["flops"] += 4 * len(A) ** 2
            t_d = math.inf; kd = -1
            for k, (a, rk) in enumerate(zip(A, r)):
                if rk > 1e-12 and lamv[a] / rk < t_d:
                    t_d = lamv[a] / rk; kd = k
            if curv <= DEP * Sfull[p, p]:               # the new row depends on the active ones: dual step only
                if kd < 0:
                    infeasible = True
                    break
                t = t_d
            else:
                t = min(vp / curv, t_d)
            for a, rk in zip(A, r):
                lamv[a] -= t * rk
            lamv[p] += t
            if lamv[p] > 1.0e5:                                      # the exact penalty of the elastic rows: beyond rho the row gives way = infeasible
                infeasible = True
                break
            if curv > DEP * Sfull[p, p] and t == vp / curv:
                A.append(p)
                break
            vp -= t * max(curv, 0.0)
            lamv[A[kd]] = 0.0
            A.pop(kd); drops += 1
            stats["flops"] += 3 * len(A) ** 2
        if infeasible:
            break
        applies += 1
    lamv = np.maximum(lamv, 0.0)
    U = Cd.Uunc - HiGt @ lamv
    obj = float(0.5 * U @ M.H @ U + M.g @ U + M.k0) + float((0.5 * dreg * lamv ** 2).sum())
    stats["adds"].append(adds); stats["drops"].append(drops + nfix); stats["A0"].append(len(A0)); stats["Aend"].append(len(A))
    stats["applies"].append(applies)
    return dict(U=U, lam=lamv, obj=obj, infeasible=infeasible, A=A)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    I = instance(cfg, seed)
    M = DModel(I)
    B = BnB(M, gap=0.01)
    Cd = Condensed(M)
    N, C = I.N, I.C
    ns, nc = 8 * C, 6 * C
    F_iter = N * (ns ** 3 / 3 + 2 * ns * ns * nc + ns * nc * nc + 4 * ns * ns)
    F_apply = 2 * N * ns * nc * 2                                                    # backward + forward substitution with constant gains
    heap = [(-math.inf, 0, {}, None)]
    cnt = itertools.count(1)
    inc = math.inf; nodes = 0
    st = dict(chol=0.0, setup_pairs=0, flops=0.0, fix_drops=0, adds=[], drops=[], A0=[], Aend=[], applies=[])
    ipm_its = []; cost_as = []; cost_ipm = []; mismatch = 0; infeas_both = 0; infeas_mis = 0; nrows = []; feas = []; newrows = []
    while heap and nodes < nmax:
        bnd, _, fix, par = heapq.heappop(heap)
        if inc < math.inf and inc - bnd <= 0.01 * abs(inc):
            break
        nodes += 1
        keys, G, h, av = pack(M, B.node_rows(fix))
        ref = ipm(M, keys, G, h, av, start=par, mode="primal")
        ref_inf = ref["viol"] > FEAS_TOL
        if par is not None:
            pk = par["rows"]                                        # key -> (rhs, multiplier) of the parent
            cpar = h - G @ par["U"]                                 # slack of the child's rows at the parent's solution
            A0 = [k for k, key in enumerate(keys) if key in pk and abs(pk[key][0] - h[k]) < 1e-12 and pk[key][1] > max(cpar[k], 0.0)]
            f0 = st["flops"]; c0 = st["chol"]; p0 = st["setup_pairs"]
            if os.environ.get("AS_COLD"): A0 = []
            r = gi_solve(Cd, keys, G, h, av, A0, st)
            work = (st["flops"] - f0) + (st["chol"] - c0) + 36 * (st["setup_pairs"] - p0) + st["applies"][-1] * (F_apply + 2 * 3 * len(h))
            cost_as.append(work / (F_iter + 96 * len(h)))
            ipm_its.append(ref["it"]); cost_ipm.append(ref["it"]); nrows.append(len(h)); feas.append(not (r["infeasible"] or ref_inf))
            newrows.append(sum(1 for k, key in enumerate(keys) if key not in pk or abs(pk[key][0] - h[k]) >= 1e-12))
            if r["infeasible"] or ref_inf:
                if r["infeasible"] and ref_inf:
                    infeas_both += 1
                else:
                    infeas_mis += 1
            elif abs(r["obj"] - ref["obj"]) > 2e-4 * max(1.0, abs(ref["obj"])):
                mismatch += 1                                       # (the interior point stops at a complementarity of 1e-6 |obj|)
        obj = ref["obj"] + B.const_cost(fix)
        if ref_inf or not ref["ok"] or obj >= inc:
            continue
        viol, comp = B.complete(fix, ref["Z"])
        if not viol:
            inc = obj
            continue
        _, _, key, alts = viol[0]
        rowinfo = {k: (float(hh), float(ref["lam"][k])) for k, hh in zip(keys, h)}
        for alt in alts:
            f2 = dict(fix); f2[key] = alt
            heapq.heappush(heap, (obj, next(cnt), f2, dict(U=ref["U"], lam=ref["lam"], rows=rowinfo)))
    a = lambda x: np.array(x, dtype=float)
    q = lambda x: "median %5.1f  mean %5.1f  p95 %5.1f  max %5.0f" % (np.median(a(x)), a(x).mean(), np.percentile(a(x), 95), a(x).max())
    print("%s seed %d: %d nodes (%d children), incumbent %s, rows per node %d" % (cfg, seed, nodes, len(cost_as), inc, int(np.mean(nrows))))
    print("  interior point from the parent's solution, iterations        %s" % q(ipm_its))
    print("  active set: rows active at the parent and kept |A0|          %s" % q(st["A0"]))
    print("              rows added (Goldfarb-Idnani outer steps)          %s" % q(st["adds"]))
    print("              rows dropped (incl. negative multipliers at A0)   %s" % q(st["drops"]))
    print("              H^-1 applications (iterate refreshes)             %s" % q(st["applies"]))
    print("              work in interior-point-iteration equivalents      %s" % q(cost_as))
    f = np.array(feas, dtype=bool)
    for nm, msk in (("feasible children", f), ("infeasible children", ~f)):
        if msk.any():
            print("  %-20s %4d: interior point iterations %s" % (nm, msk.sum(), q(a(ipm_its)[msk])))
            print("  %-20s       rows added                %s" % ("", q(a(st["adds"])[msk])))
            print("  %-20s       rows dropped              %s" % ("", q(a(st["drops"])[msk])))
            print("  %-20s       active at the end         %s" % ("", q(a(st["Aend"])[msk])))
            print("  %-20s       rows new or tightened     %s" % ("", q(a(newrows)[msk])))
    print("  children infeasible in both %d, verdicts differ %d, optimum differs by more than the interior point's own tolerance %d" % (infeas_both, infeas_mis, mismatch))


if __name__ == "__main__":
    main()
