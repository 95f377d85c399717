"""Offline lab for the start of the node relaxations' interior point (no GPU, no oracle): a dense condensed numpy replica of the
device iteration (same row model - elastic rows with the exact penalty rho, quadratic-soft rows -, same centring rule, step
fraction, tolerances; kernels.hip: init_elastic, row_step, ipm_onchip.hip) on the rows of tools/miqp_py's plain disjunctive
model.  Walks a best-first tree of a synthetic instance and solves every child from several starting points - cold, the parent's
solution, the parent's solution and multipliers, ... - counting the iterations each needs.  Used to choose the defaults of
MIQP_WS_* (DESIGN.md 3.2); nothing here is on the product path.

    python tools/ipm_lab.py [cfg] [seed] [nodes]
"""
import ctypes as C
import heapq
import itertools
import math
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)

from miqp_py.bnb import BnB, DModel, FEAS_TOL   # noqa: E402
from miqp_py.dat import load_dat                # noqa: E402
from miqp_py.model import Inst                  # noqa: E402

RHO = 1.0e5
T0, S0, LAM0 = 1.0e-3, 100.0, 2000.0
SIG0, SIG_LO, SIG_HI = 0.3, 0.02, 0.5
STEPFRAC = 0.995


def instance(cfg, seed):
    import planner_miqp_amd as P
    from planner_miqp_amd import synthetic
    p = synthetic.generate(cfg, seed, gap=0.01, max_time=10)
    w = P.CplexWrapper(); w.resetParameters(p)
    assert w._push_inputs() == 0
    f = os.path.join(tempfile.mkdtemp(), "inst.dat")
    assert P.load_library().miqp_solver_write_dat(w._h, f.encode()) == 0
    return Inst.from_dat(load_dat(f))


def pack(M, rows):
    """rows -> (keys, G in U space, h, aq) with the rows of one key (stage, coefficients) collapsed to the tightest"""
    best = {}
    for (i, co, rhs, a) in rows:
        k = (i, co.tobytes(), a)
        if k not in best or rhs < best[k][2]:
            best[k] = (i, co, rhs, a)
    keys = list(best)
    G = np.stack([best[k][1] @ M.ZU[best[k][0]] for k in keys])
    h = np.array([best[k][2] - best[k][1] @ M.Z0[best[k][0]] for k in keys])
    av = np.array([best[k][3] for k in keys])
    return keys, G, h, av


def ipm(M, keys, G, h, av, start=None, tol=1e-6, maxit=80, theta=1.0, trace=False, mu0=1.0, delta=1e-3, mode="primal", lamfloor=None, sviol=0.0, svmin=0.0, muv=None, pc=0, sig_hi=None):
    """start: None (cold) or dict(U=, lam={key: lambda}); mode: how the parent's multipliers are used;
    pc: 0 the device's centring rule (sigma = 1 - alpha of the previous step), 1 sigma from an affine predictor step ((mu_aff / mu)^3, Mehrotra's
    heuristic) without the second-order term, 2 with it (the full predictor-corrector); both cost a second solve with the same factorisation.
    Returns it = iterations and solves = linear solves (1 or 2 per iteration)"""
    H, g = M.H, M.g
    el = av == 0
    m = len(h); ncomp = int(el.sum()) * 2 + int((~el).sum())
    U = np.zeros(M.nU) if start is None else start["U"].copy()
    c = h - G @ U
    s = np.empty(m); lam = np.empty(m); t = np.zeros(m)
    warm = start is not None
    lp = np.zeros(m)
    if warm and start.get("lam") is not None and mode != "primal":
        lp = np.array([start["lam"].get(k, 0.0) for k in keys])
    for k in range(m):
        if el[k]:
            if not warm:
                if c[k] > T0:
                    t[k] = T0; s[k] = c[k] + T0
                else:
                    s[k] = S0 * T0; t[k] = s[k] - c[k]
                lam[k] = LAM0
            else:
                sk = max(c[k], delta)
                if c[k] < delta and (sviol > 0 or svmin > 0):   # violated (or not yet satisfied with room) at the parent's solution: a slack the first step can shrink
                    sk = max(delta, sviol * abs(c[k]), svmin)
                lk = (muv if (muv is not None and c[k] < delta) else mu0) / sk
                if mode == "max":
                    lk = max(lk, lp[k])
                elif mode == "parent":          # the parent's multiplier wherever the row existed there
                    lk = lp[k] if lp[k] > 0 else lk
                    if lamfloor is not None:
                        lk = max(lk, lamfloor)
                elif mode == "parent_s":        # ... and the slack that goes with it at mu0 (active rows sit at mu0 / lambda)
                    if lp[k] > 0:
                        lk = max(lp[k], lamfloor or 0.0); sk = max(c[k], mu0 / lk)
                lk = min(lk, 0.5 * RHO)
                tt = mu0 / (RHO - lk)
                if sk - c[k] > tt:
                    tt = sk - c[k]
                t[k] = tt; s[k] = c[k] + tt; lam[k] = lk
        else:
            lam[k] = max(1.0, -2 * c[k] * av[k] + 1.0, lp[k]); s[k] = c[k] + lam[k] / av[k]
    sigma = SIG0; resid_fac = 1.0; R0 = None; ok = False; nsolves = 0
    wmix = theta if warm else 1.0
    it = 0
    for it in range(1, maxit + 1):
        mu = np.where(el, RHO - lam, 1.0)
        obj = 0.5 * U @ H @ U + g @ U + M.k0
        comp = (s @ lam + (t * mu)[el].sum()) / max(1, ncomp)
        rd = H @ U + g + G.T @ lam
        if R0 is None:
            R0 = np.abs(rd).max()
        if trace:
            print("   it %2d comp %.3e obj %.5f resid %.2e (true %.2e)" % (it, comp, obj, resid_fac * R0, np.abs(rd).max()))
        if comp < tol * max(1.0, abs(obj)) and resid_fac * R0 < 1e-7:
            ok = True
            break
        zz = np.where(el, t / mu, 1.0 / np.where(el, 1.0, av))
        w = 1.0 / (s / lam + zz)
        K = H + G.T @ (w[:, None] * G)
        so1 = 0.0; so2 = 0.0
        if pc:
            r1a = -s * lam; r2a = np.where(el, -t * mu, 0.0)
            kapa = (r1a / lam - np.where(el, r2a / mu, 0.0)) * w
            dUa = np.linalg.solve(K, -rd - G.T @ kapa); nsolves += 1
            dla = w * (G @ dUa) + kapa; dsa = (r1a - s * dla) / lam; dta = np.where(el, (r2a + t * dla) / mu, 0.0)
            ra = [np.where(dsa < 0, -s / np.where(dsa < 0, dsa, -1), np.inf), np.where(dla < 0, -lam / np.where(dla < 0, dla, -1), np.inf),
                  np.where(el & (dta < 0), -t / np.where(dta < 0, dta, -1), np.inf), np.where(el & (dla > 0), mu / np.where(dla > 0, dla, 1), np.inf)]
            aa = min(1.0, min(float(r.min()) for r in ra))
            mua = ((s + aa * dsa) @ (lam + aa * dla) + ((t + aa * dta) * (mu - aa * dla))[el].sum()) / max(1, ncomp)
            sigma = min(sig_hi if sig_hi is not None else SIG_HI, max(SIG_LO if pc < 3 else 1e-4, (mua / comp) ** 3))
            if pc == 2:
                so1 = dsa * dla; so2 = np.where(el, -dta * dla, 0.0)
        tau = sigma * comp * wmix; k1 = 1.0 - sigma * (1.0 - wmix)
        r1 = tau - k1 * s * lam - so1
        r2 = np.where(el, tau - k1 * t * mu - so2, 0.0)
        kap = (r1 / lam - np.where(el, r2 / mu, 0.0)) * w
        dU = np.linalg.solve(K, -rd - G.T @ kap); nsolves += 1
        gd = G @ dU
        dl = w * gd + kap
        ds = (r1 - s * dl) / lam
        dt = np.where(el, (r2 + t * dl) / mu, 0.0)
        ratios = [np.where(ds < 0, -s / np.where(ds < 0, ds, -1), np.inf), np.where(dl < 0, -lam / np.where(dl < 0, dl, -1), np.inf),
                  np.where(el & (dt < 0), -t / np.where(dt < 0, dt, -1), np.inf), np.where(el & (dl > 0), mu / np.where(dl > 0, dl, 1), np.inf)]
        amax = min(float(r.min()) for r in ratios)
        if trace:
            bi = int(np.argmin([r.min() for r in ratios])); kk = int(np.argmin(ratios[bi]))
            print("        blocked by %s of row %d (stage %d, new %s): c %.3e s %.3e lam %.3e t %.3e  ds %.3e dl %.3e dt %.3e gd %.3e" % (
                ["s", "lam", "t", "mu"][bi], kk, keys[kk][0], lp[kk] == 0.0 if warm else None, (h - G @ U)[kk], s[kk], lam[kk], t[kk], ds[kk], dl[kk], dt[kk], gd[kk]))
        alpha = min(1.0, STEPFRAC * amax)
        U = U + alpha * dU; s = s + alpha * ds; lam = lam + alpha * dl; t = t + alpha * dt
        resid_fac *= (1.0 - alpha)
        if not pc:
            sigma = min(SIG_HI, max(SIG_LO, 1.0 - alpha))
        if trace:
            print("        alpha %.4f sigma_next %.3f" % (alpha, sigma))
        if alpha < 1e-12:
            break
    c = h - G @ U
    viol = float(np.maximum(-c, 0)[el].max()) if el.any() else 0.0
    sc = float((0.5 * av[~el] * (lam[~el] / av[~el]) ** 2).sum()) if (~el).any() else 0.0
    obj = float(0.5 * U @ H @ U + g @ U + M.k0) + sc
    return dict(U=U, Z=M.Z0 + M.ZU @ U, obj=obj, viol=viol, it=it, ok=ok, solves=nsolves, lam={k: float(v) for k, v in zip(keys, lam)})


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    I = instance(cfg, seed)
    M = DModel(I)
    B = BnB(M, gap=0.01)
    variants = {
        "cold": dict(start=False),
        "primal sv1": dict(mode="primal", sviol=1.0),
        "primal svmin.1": dict(mode="primal", svmin=0.1),
        "primal svmin.03": dict(mode="primal", svmin=0.03),
        "parent svmin.1": dict(mode="parent", svmin=0.1),
        "parent svmin.1 mu.1": dict(mode="parent", svmin=0.1, mu0=0.1),
        "parent svmin.1 mu.1 muv1": dict(mode="parent", svmin=0.1, mu0=0.1, muv=1.0),
        "parent svmin.1 mu.01 muv1": dict(mode="parent", svmin=0.1, mu0=0.01, muv=1.0),
        "parent svmin.1 muv10": dict(mode="parent", svmin=0.1, muv=10.0),
        "parent svmin.3 muv10": dict(mode="parent", svmin=0.3, muv=10.0),
        "parent sv1 svmin.03": dict(mode="parent", sviol=1.0, svmin=0.03),
        "parent svmin.1 th.3": dict(mode="parent", svmin=0.1, theta=0.3),
        "primal mu1 d1e-3": dict(mode="primal"),
        "primal mu1 th.3": dict(mode="primal", theta=0.3),
        "max mu1 d1e-3": dict(mode="max"),
        "max mu.01": dict(mode="max", mu0=0.01),
        "parent mu1": dict(mode="parent"),
        "parent mu.1": dict(mode="parent", mu0=0.1),
        "parent mu.01": dict(mode="parent", mu0=0.01),
        "parent mu.1 th.3": dict(mode="parent", mu0=0.1, theta=0.3),
        "parent_s mu.1": dict(mode="parent_s", mu0=0.1),
        "parent_s mu.01": dict(mode="parent_s", mu0=0.01),
        "parent_s mu.01 th.3": dict(mode="parent_s", mu0=0.01, theta=0.3),
        "parent_s mu.001": dict(mode="parent_s", mu0=0.001),
    }
    if len(sys.argv) > 4 and sys.argv[4] == "proj":   # round 5: the parent's solution moved onto the rows the branching violates before the interior point starts
        # (the minimiser of the parent's Lagrangian model 1/2 dU' H dU - H: the objective alone, as in the bound lifting - subject to the violated rows holding with a margin)
        variants = {
            "cold": dict(start=False),
            "primal (device rule)": dict(mode="primal"),
            "proj margin 0": dict(mode="primal", proj=0.0),
            "proj margin 1e-3": dict(mode="primal", proj=1e-3),
            "proj margin 1e-2": dict(mode="primal", proj=1e-2),
            "proj margin 1e-1": dict(mode="primal", proj=1e-1),
            "proj 1e-2 mu.1": dict(mode="primal", proj=1e-2, mu0=0.1),
            "proj 1e-2 mu.01": dict(mode="primal", proj=1e-2, mu0=0.01),
            "proj 1e-1 mu.1": dict(mode="primal", proj=1e-1, mu0=0.1),
        }
    if len(sys.argv) > 4 and sys.argv[4] == "pc":   # the centring rules against each other, product start (parent's solution, mu0 = 1, delta = 1e-3)
        variants = {
            "cold": dict(start=False),
            "cold pc1": dict(start=False, pc=1),
            "cold pc2": dict(start=False, pc=2),
            "primal (device rule)": dict(mode="primal"),
            "primal pc1": dict(mode="primal", pc=1),
            "primal pc1 hi1": dict(mode="primal", pc=1, sig_hi=1.0),
            "primal pc1 lo1e-4": dict(mode="primal", pc=3),
            "primal pc2": dict(mode="primal", pc=2),
            "primal pc2 hi1": dict(mode="primal", pc=2, sig_hi=1.0),
        }
    its = {k: [] for k in variants}; fails = {k: 0 for k in variants}; sol = {k: [] for k in variants}
    heap = [(-math.inf, 0, {}, None)]
    cnt = itertools.count(1)
    inc = math.inf; nodes = 0
    while heap and nodes < nmax:
        bnd, _, fix, par = heapq.heappop(heap)
        if inc < math.inf and inc - bnd <= 0.01 * abs(inc):
            break
        nodes += 1
        keys, G, h, av = pack(M, B.node_rows(fix))
        ref = ipm(M, keys, G, h, av, start=par, mode="primal")
        if par is not None:
            for name, kw in variants.items():
                kw = dict(kw); st = par if kw.pop("start", True) else None
                pm = kw.pop("proj", None)
                if pm is not None and st is not None:
                    U0 = st["U"]; cres = h - G @ U0
                    V = np.where((av == 0) & (cres < pm))[0]            # elastic rows that do not hold with the margin at the parent's solution
                    if len(V):
                        Hi_Gt = np.linalg.solve(M.H, G[V].T)             # H^-1 G_V'
                        S_ = G[V] @ Hi_Gt + 1e-12 * np.eye(len(V))
                        dU = -Hi_Gt @ np.linalg.solve(S_, (pm - cres[V]))
                        st = dict(U=U0 + dU, lam=st.get("lam"))
                r = ipm(M, keys, G, h, av, start=st, **kw)
                its[name].append(r["it"]); sol[name].append(r["solves"])
                if not r["ok"] or abs(r["obj"] - ref["obj"]) > 1e-4 * max(1.0, abs(ref["obj"])):
                    fails[name] += 1
        obj = ref["obj"] + B.const_cost(fix)
        if ref["viol"] > FEAS_TOL or not ref["ok"] or obj >= inc:
            continue
        viol, comp = B.complete(fix, ref["Z"])
        if not viol:
            inc = obj
            continue
        _, _, key, alts = viol[0]
        for alt in alts:
            f2 = dict(fix); f2[key] = alt
            heapq.heappush(heap, (obj, next(cnt), f2, dict(U=ref["U"], lam=ref["lam"])))
    print("%s seed %d: %d nodes, incumbent %s, rows of the last node %d" % (cfg, seed, nodes, inc, len(keys)))
    for name in variants:
        a = np.array(its[name])
        print("  %-22s mean %5.2f  median %4.1f  p90 %4.1f  max %3d  solves %5.2f  (%d children, %d not converged / other optimum)" % (name, a.mean(), np.median(a), np.percentile(a, 90), a.max(), np.mean(sol[name]), len(a), fails[name]))


if __name__ == "__main__":
    main()


def trace_one(cfg="cfg3", seed=0, depth=6, **kw):
    """trace of the solve of one child `depth` levels down the first dive"""
    I = instance(cfg, seed); M = DModel(I); B = BnB(M, gap=0.01)
    fix = {}; par = None
    for d in range(depth + 1):
        keys, G, h, av = pack(M, B.node_rows(fix))
        last = d == depth
        if last:
            print("child at depth %d: %d rows; cold:" % (d, len(keys)))
            ipm(M, keys, G, h, av, start=None, trace=True)
            print("warm (%s):" % kw)
        r = ipm(M, keys, G, h, av, start=par, trace=last, **(kw if last else dict(mode="primal")))
        viol, comp = B.complete(fix, r["Z"])
        if not viol:
            break
        _, _, key, alts = viol[0]
        fix = dict(fix); fix[key] = alts[0]
        par = dict(U=r["U"], lam=r["lam"])
