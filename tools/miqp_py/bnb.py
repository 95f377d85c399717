"""Disjunctive model, dense condensed IPM and a sequential best-first B&B (numpy).

Checker only (tests/, golden generation).  See model.py for the reference
citations; SURVEY.md App. A is the line-by-line restatement this follows.
"""
import heapq
import itertools
import math
import numpy as np

from .model import (BIGM, PX, VX, AX, PY, VY, AY, PT_R, PT_U, PT_L, ENV_PTS, OBS_PTS, Inst)

RHO = 1.0e5     # exact-penalty weight of the elastic rows
FEAS_TOL = 1e-6


class DModel:
    def __init__(self, I: Inst):
        self.I = I
        C, N = I.C, I.N
        self.nz = 8 * C
        self.nU = 2 * C * (N - 1)
        A, B = I.A_B()
        self.A, self.B = A, B
        # condensed maps z_i = Z0[i] + ZU[i] @ U
        Z0 = np.zeros((N, self.nz)); ZU = np.zeros((N, self.nz, self.nU))
        x0 = I.x0.reshape(-1).copy()
        X0 = x0; XU = np.zeros((6 * C, self.nU))
        for i in range(N):
            Z0[i, :6 * C] = X0; ZU[i, :6 * C, :] = XU
            if i < N - 1:
                sel = np.zeros((2 * C, self.nU)); sel[:, 2 * C * i:2 * C * (i + 1)] = np.eye(2 * C)
                ZU[i, 6 * C:, :] = sel
                X0 = A @ X0; XU = A @ XU + B @ sel
        self.Z0, self.ZU = Z0, ZU
        # objective: sum_i (z_i - r_i)' W (z_i - r_i)
        Wd = np.zeros(self.nz); R = np.zeros((N, self.nz))
        for c in range(C):
            Wd[6 * c:6 * c + 6] = I.W[c, :6]; Wd[6 * C + 2 * c:6 * C + 2 * c + 2] = I.W[c, 6:8]
            R[:, 6 * c:6 * c + 6] = I.ref[c]
        self.Wd, self.Rref = Wd, R
        H = np.zeros((self.nU, self.nU)); g = np.zeros(self.nU); k = 0.0
        for i in range(N):
            d = Z0[i] - R[i]
            H += 2 * ZU[i].T @ (Wd[:, None] * ZU[i]); g += 2 * ZU[i].T @ (Wd * d); k += float(d @ (Wd * d))
        self.H, self.g, self.k0 = H, g, k
        self.pairs = [(a, b) for a in range(C) for b in range(a + 1, C)]
        self._nonslow = [self._nonslow_halfspaces(j) for j in range(I.R)]

    # ---------------- geometry helpers --------------------------------
    def _nonslow_halfspaces(self, j):
        """Halfspaces h (axis, sign) such that sector_j \\ slow-square = U_h sector_j ^ {sign*v_axis >= vm},
        dominated ones removed (exact for any sector narrower than pi)."""
        F = self.I.frac[j]
        t1 = math.atan2(F[1], F[0]); t2 = math.atan2(F[3], F[2])
        if t2 < t1:
            t2 += 2 * math.pi
        cands = []
        for ax, sg in ((0, 1), (0, -1), (1, 1), (1, -1)):
            n = np.zeros(2); n[ax] = sg
            ths = np.linspace(t1, t2, 65)
            vals = n[0] * np.cos(ths) + n[1] * np.sin(ths)
            if vals.max() > 1e-12:
                cands.append((ax, sg, ths, vals))
        keep = []
        for a in cands:
            dominated = False
            for b in cands:
                if a is b:
                    continue
                m = a[3] > 1e-12
                # on every ray where a is reachable, b is reached no later than a
                if np.all(b[3][m] >= a[3][m] - 1e-12) and (np.any(b[3][m] > a[3][m] + 1e-12) or cands.index(b) < cands.index(a)):
                    dominated = True
                    break
            if not dominated:
                keep.append((a[0], a[1]))
        return keep

    def point_affine(self, c, i, j, tx, ty):
        """(cx, kx, cy, ky): X = cx . z_i + kx, Y = cy . z_i + ky for the point with x/y types tx/ty."""
        I = self.I
        cx = np.zeros(self.nz); cy = np.zeros(self.nz)
        o = 6 * c
        if i == 0:
            th = I.theta0(c)
            X = I.x0[c, PX] + (math.cos(th) * I.wb[c] if tx != PT_R else 0.0)
            Y = I.x0[c, PY] + (math.sin(th) * I.wb[c] if ty != PT_R else 0.0)
            return cx, X, cy, Y
        cx[o + PX] = 1.0; kx = 0.0
        if tx != PT_R:
            p = I.poly["COSS_UB" if tx == PT_U else "COSS_LB"][j]
            kx = I.wb[c] * p[0]; cx[o + VX] += I.wb[c] * p[1]; cx[o + VY] += I.wb[c] * p[2]
        cy[o + PY] = 1.0; ky = 0.0
        if ty != PT_R:
            p = I.poly["SINT_UB" if ty == PT_U else "SINT_LB"][j]
            ky = I.wb[c] * p[0]; cy[o + VX] += I.wb[c] * p[1]; cy[o + VY] += I.wb[c] * p[2]
        return cx, kx, cy, ky

    # rows: (i, coef[nz], rhs, a)  meaning coef.z_i - slack <= rhs ; a=0 elastic, a>0 quadratic-soft
    def lin_row(self, i, terms, rhs, a=0.0):
        co = np.zeros(self.nz)
        for idx, v in terms:
            co[idx] += v
        return (i, co, rhs, a)

    def poly_row(self, i, parts, gamma, a=0.0, normalize=True):
        """sum_k alpha_k*X_k + beta_k*Y_k <= gamma ; parts = [(c, j, tx, ty, alpha, beta)]"""
        co = np.zeros(self.nz); rhs = gamma
        nrm = 0.0
        for (c, j, tx, ty, al, be) in parts:
            cx, kx, cy, ky = self.point_affine(c, i, j, tx, ty)
            co += al * cx + be * cy
            rhs -= al * kx + be * ky
            nrm = max(nrm, math.hypot(al, be))
        if normalize and nrm > 0:
            co = co / nrm; rhs = rhs / nrm; a = a * nrm * nrm if a else a
        return (i, co, rhs, a)

    # ---------------- alternatives --------------------------------------
    def global_rows(self, c, i):
        I = self.I; o = 6 * c; u = 6 * I.C + 2 * c
        rows = []
        if i >= 1:
            rows += [self.lin_row(i, [(o + VX, -1)], -I.vmin), self.lin_row(i, [(o + VY, -1)], -I.vmin),
                     self.lin_row(i, [(o + VX, 1)], I.vmax)]
            for s in (AX, AY):
                rows += [self.lin_row(i, [(o + s, 1)], I.amax), self.lin_row(i, [(o + s, -1)], -I.amin)]
        if i <= I.N - 2:
            for s in (0, 1):
                hi, lo = I.jmax, I.jmin
                if i == 0:  # A1: initial region jerk box + residual rows of all other regions
                    j0 = I.init_region[c] - 1
                    for j in range(I.R):
                        M = 0.0 if j == j0 else BIGM["jerk"]
                        hi = min(hi, I.jerk_lim[c, j, 2 * s + 1] + M); lo = max(lo, I.jerk_lim[c, j, 2 * s] - M)
                rows += [self.lin_row(i, [(u + s, 1)], hi), self.lin_row(i, [(u + s, -1)], -lo)]
        return rows

    def region_alts(self, c, i):
        """alternatives (j, h): h = -1 slow (region inherited, must equal region of step i-1), else index of
        the non-slow halfspace of sector j."""
        I = self.I
        alts = []
        for j in range(I.R):
            if not I.possible[c, j]:
                continue
            for h in range(len(self._nonslow[j])):
                alts.append((j, h))
            alts.append((j, -1))
        return alts

    def region_rows(self, c, i, alt):
        I = self.I; j, h = alt; o = 6 * c; u = 6 * I.C + 2 * c
        F = I.frac[j]; rows = []
        poss = [jj for jj in range(I.R) if I.possible[c, jj]]
        if h >= 0:
            n1 = math.hypot(F[0], F[1]); n3 = math.hypot(F[2], F[3])
            rows.append(self.lin_row(i, [(o + VY, -F[0] / n1), (o + VX, F[1] / n1)], 0.0))
            rows.append(self.lin_row(i, [(o + VY, F[2] / n3), (o + VX, -F[3] / n3)], 0.0))
            ax, sg = self._nonslow[j][h]
            rows.append(self.lin_row(i, [(o + (VX if ax == 0 else VY), -sg)], -I.vm))
            rho = (F[1] + F[3]) / (F[0] + F[2])
            Kx = I.poly["KAPPA_AX_MAX"][j]; Kn = I.poly["KAPPA_AX_MIN"][j]
            rows.append(self.lin_row(i, [(o + AY, 1), (o + AX, -rho), (o + VX, -Kx[1]), (o + VY, -Kx[2])], Kx[0]))
            rows.append(self.lin_row(i, [(o + AY, -1), (o + AX, rho), (o + VX, Kn[1]), (o + VY, Kn[2])], -Kn[0]))
        else:
            for s in (VX, VY):
                rows += [self.lin_row(i, [(o + s, 1)], I.vm), self.lin_row(i, [(o + s, -1)], I.vm)]
        # boxes incl. residual big-M rows of the other possible regions (a_j' = 0)
        for s, st in ((0, AX), (1, AY)):
            hi = min([I.acc_lim[c, jj, 2 * s + 1] + (0 if jj == j else BIGM["acc"]) for jj in poss])
            lo = max([I.acc_lim[c, jj, 2 * s] - (0 if jj == j else BIGM["acc"]) for jj in poss])
            rows += [self.lin_row(i, [(o + st, 1)], hi), self.lin_row(i, [(o + st, -1)], -lo)]
        if i <= I.N - 2:
            for s in (0, 1):
                hi = min([I.jerk_lim[c, jj, 2 * s + 1] + (0 if jj == j else BIGM["jerk"]) for jj in poss])
                lo = max([I.jerk_lim[c, jj, 2 * s] - (0 if jj == j else BIGM["jerk"]) for jj in poss])
                rows += [self.lin_row(i, [(u + s, 1)], hi), self.lin_row(i, [(u + s, -1)], -lo)]
        return rows

    def env_rows(self, c, i, pt, e, j):
        tx, ty = ENV_PTS[pt]; rows = []
        for (x1, y1, x2, y2) in self.I.env[e]:
            al = (y2 - y1); be = -(x2 - x1)
            rows.append(self.poly_row(i, [(c, j, tx, ty, al, be)], al * x1 + be * y1))
        return rows

    def obs_rows(self, c, o, i, pt, k, j):
        tx, ty = OBS_PTS[pt]
        x1, y1, x2, y2 = self.I.obs[o][i][k]
        al = -(y2 - y1); be = (x2 - x1)
        return [self.poly_row(i, [(c, j, tx, ty, al, be)], al * x1 + be * y1)]

    def c2c_rows(self, p, i, grp, alt, j1, j2):
        I = self.I; c1, c2 = self.pairs[p]
        D = I.rad[c1] + I.rad[c2] + I.safety[i]; S = I.safety_slack[i]
        isx = alt < 2
        R, U, L = PT_R, PT_U, PT_L
        if grp == 0:
            A_, B_ = ((c1, j1, R, R), (c2, j2, R, R)) if alt in (0, 2) else ((c2, j2, R, R), (c1, j1, R, R))
            soft = True
        elif grp == 1:  # rear c1 vs front c2
            if alt in (0, 2):
                A_, B_ = (c1, j1, R, R), (c2, j2, L, L)
            else:
                A_, B_ = (c2, j2, U, U), (c1, j1, R, R)
            soft = False
        elif grp == 2:  # rear c2 vs front c1
            if alt in (0, 2):
                A_, B_ = (c2, j2, R, R), (c1, j1, L, L)
            else:
                A_, B_ = (c1, j1, U, U), (c2, j2, R, R)
            soft = False
        else:  # front/front worst case
            if alt in (0, 2):
                A_, B_ = (c2, j2, U, U), (c1, j1, L, L)
            else:
                A_, B_ = (c1, j1, U, U), (c2, j2, L, L)
            soft = True
        al, be = (1.0, 0.0) if isx else (0.0, 1.0)
        parts = [A_ + (al, be), B_ + (-al, -be)]
        if not soft:
            return [self.poly_row(i, parts, -D)]
        smax = min(S, I.max_slack)
        rows = [self.poly_row(i, parts, -(D + S) + smax)]
        if smax > 0:
            if I.w_slack > 0:
                rows.append(self.poly_row(i, parts, -(D + S), a=2 * I.w_slack))
        return rows


def row_value(row, Z):
    i, co, rhs, a = row
    return float(co @ Z[i] - rhs)


def solve_qp(M: DModel, rows, tol=1e-9, maxit=60, verbose=False, U0=None, tau0=1.0, corr=0.0, fixed_sigma=None):
    """Dense condensed primal-dual IPM; every row elastic (a=0) or quadratic-soft (a>0).
    Returns dict(U, Z, obj, viol, it, ok)."""
    H, g = M.H, M.g
    n = M.nU
    if rows:
        G = np.stack([co @ M.ZU[i] for (i, co, rhs, a) in rows])
        h = np.array([rhs - co @ M.Z0[i] for (i, co, rhs, a) in rows])
        av = np.array([a for (i, co, rhs, a) in rows])
    else:
        G = np.zeros((0, n)); h = np.zeros(0); av = np.zeros(0)
    el = av == 0
    m = len(h)
    U = np.zeros(n) if U0 is None else U0.copy()
    c = h - G @ U
    avs = np.where(el, 1.0, av)
    rho = np.where(el, RHO, 0.0)
    # crude but robust start: lam = tau0, elastic slack t >= 1 so that s = c + t >= 1
    lam = np.where(el, tau0, np.maximum(tau0, -2 * c * avs + tau0))
    s = np.where(el, np.maximum(c, 0) + 1.0, c + lam / avs)
    it = 0
    ok = False
    for it in range(1, maxit + 1):
        c = h - G @ U
        t = np.where(el, s - c, 0.0)
        mu = np.where(el, rho - lam, 0.0)
        z = np.where(el, t / np.where(el, mu, 1), 1.0 / np.where(el, 1, av))
        D = s / lam + z
        w = 1.0 / D
        rd = H @ U + g + G.T @ lam
        comp = (s @ lam + (t * mu)[el].sum()) / max(1, m + el.sum())
        if verbose: print(it, 'comp %.3e rd %.3e'%(comp, np.abs(rd).max()))
        if comp < tol * max(1.0, abs(0.5 * U @ H @ U + g @ U + M.k0)) and np.abs(rd).max() < 1e-6:
            ok = True
            break
        K = H + G.T @ (w[:, None] * G)
        if not np.all(np.isfinite(K)):
            break
        try:
            Lc = np.linalg.cholesky(K + 1e-13 * np.eye(n))
        except np.linalg.LinAlgError:
            try:
                Lc = np.linalg.cholesky(K + 1e-8 * np.eye(n))
            except np.linalg.LinAlgError:
                break

        def solve(r1, r2):
            kap = (r1 / lam - np.where(el, r2 / np.where(el, mu, 1), 0)) / D
            rhs = -rd - G.T @ kap
            dU = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs))
            gd = G @ dU
            dlam = w * gd + kap
            ds = (r1 - s * dlam) / lam
            dt = np.where(el, ds + gd, 0); dmu = -dlam
            return dU, dlam, ds, dt, dmu

        def steplen(dlam, ds, dt, dmu):
            a_ = 1.0
            for v, dv, msk in ((s, ds, None), (lam, dlam, None), (t, dt, el), (mu, dmu, el)):
                if msk is not None:
                    v = v[msk]; dv = dv[msk]
                neg = dv < 0
                if neg.any():
                    a_ = min(a_, float((-v[neg] / dv[neg]).min()))
            return a_
        if fixed_sigma is None:
            dU, dlam, ds, dt, dmu = solve(-s * lam, -t * mu)
            aa = steplen(dlam, ds, dt, dmu)
            comp_aff = ((s + aa * ds) @ (lam + aa * dlam) + ((t + aa * dt) * (mu + aa * dmu))[el].sum()) / max(1, m + el.sum())
            sig = (comp_aff / comp) ** 3 if comp > 0 else 0.0
        else:
            sig = fixed_sigma; ds = dlam = dt = dmu = 0.0
        tau = sig * comp
        dU, dlam, ds, dt, dmu = solve(tau - s * lam - corr * ds * dlam, tau - t * mu - corr * dt * dmu)
        aa = min(1.0, 0.995 * steplen(dlam, ds, dt, dmu))
        if verbose and it >= 20 and it < 24:
            for nm, v, dv in (("s", s, ds), ("lam", lam, dlam), ("t", t, dt), ("mu", mu, dmu)):
                with np.errstate(divide="ignore", invalid="ignore"):
                    r_ = np.where(dv < 0, -v / dv, np.inf)
                if nm in ("t", "mu"):
                    r_ = np.where(el, r_, np.inf)
                k_ = int(np.argmin(r_))
                print("     block", nm, "row", k_, "ratio %.3e" % r_[k_], "s %.3e lam %.3e t %.3e mu %.3e c %.3e" % (s[k_], lam[k_], t[k_], mu[k_], c[k_]), "stage", rows[k_][0], "a", rows[k_][3])
        if verbose: print('   sig %.3e alpha %.4f'%(sig,aa))
        U = U + aa * dU; lam = lam + aa * dlam; s = s + aa * ds
        if aa < 1e-10:
            ok = comp < 1e-6
            break
    Z = M.Z0 + M.ZU @ U
    c = h - G @ U
    viol = float(np.maximum(-c, 0)[el].max()) if el.any() else 0.0
    slack_cost = float((0.5 * av[~el] * (lam[~el] / av[~el]) ** 2).sum()) if (~el).any() else 0.0
    obj = float(0.5 * U @ H @ U + g @ U + M.k0) + slack_cost
    return dict(U=U, Z=Z, obj=obj, viol=viol, it=it, ok=ok, lam=lam)


# ======================================================================
#  Branch and bound over the disjunctions
# ======================================================================
class Node:
    __slots__ = ("fix", "bound", "U", "depth")

    def __init__(self, fix, bound, U, depth):
        self.fix = fix; self.bound = bound; self.U = U; self.depth = depth


class BnB:
    """Best-first B&B.  A node fixes a subset of disjunctions:
       ('r', c, i)        -> (j, h)     region / slow flag
       ('e', c, i, pt)    -> e          environment piece of point pt
       ('o', c, o, i, pt) -> k          separating obstacle edge (k == L: soft obstacle ignored at cost)
       ('a', p, i, grp)   -> alt        agent/agent separation alternative
    Rows of alternatives that involve a front point are only added once the region of that
    (car, step) is fixed (dropping rows is a valid relaxation)."""

    def __init__(self, M: DModel, gap=0.01, verbose=False, max_nodes=200000):
        self.M = M; self.I = M.I; self.gap = gap; self.verbose = verbose; self.max_nodes = max_nodes
        self.inc = math.inf; self.inc_sol = None
        self.nodes = 0; self.qp_iters = 0; self.qp_args = {}; self.warm = True

    # ----- relaxation rows of a node
    def node_rows(self, fix):
        M, I = self.M, self.I
        rows = []
        reg = {}
        for c in range(I.C):
            for i in range(I.N):
                rows += M.global_rows(c, i)
                if i >= 1 and ('r', c, i) in fix:
                    alt = fix[('r', c, i)]
                    reg[(c, i)] = alt[0]
                    rows += M.region_rows(c, i, alt)
        for c in range(I.C):
            for i in range(1, I.N):
                j = reg.get((c, i))
                for pt in range(5):
                    if I.E == 0:
                        break
                    e = 0 if I.E == 1 else fix.get(('e', c, i, pt))
                    if e is None or (pt > 0 and j is None):
                        continue
                    rows += M.env_rows(c, i, pt, e, j if j is not None else 0)
                for o in range(I.O):
                    for pt in range(5):
                        k = fix.get(('o', c, o, i, pt))
                        if k is None or k >= len(I.obs[o][i]) or (pt > 0 and j is None):
                            continue
                        rows += M.obs_rows(c, o, i, pt, k, j if j is not None else 0)
        for p, (c1, c2) in enumerate(M.pairs):
            for i in range(1, I.N):
                for grp in range(4):
                    alt = fix.get(('a', p, i, grp))
                    if alt is None:
                        continue
                    j1, j2 = reg.get((c1, i)), reg.get((c2, i))
                    need1 = grp in (2, 3); need2 = grp in (1, 3)
                    if (need1 and j1 is None) or (need2 and j2 is None):
                        continue
                    rows += M.c2c_rows(p, i, grp, alt, j1 or 0, j2 or 0)
        return rows

    def const_cost(self, fix):
        k = 0.0
        for key, v in fix.items():
            if key[0] == 'o' and v >= len(self.I.obs[key[2]][key[3]]):
                k += self.I.w_slack_obs
        return k

    # ----- completion / violated disjunction search
    def complete(self, fix, Z, tol=FEAS_TOL):
        """Returns (violated list, completion dict).  violated = [(key, alts)] earliest step first."""
        M, I = self.M, self.I
        comp = dict(fix); viol = []
        # regions, in time order so that 'slow' can inherit
        for c in range(I.C):
            prev = I.init_region[c] - 1
            for i in range(1, I.N):
                key = ('r', c, i)
                if key in fix:
                    prev = fix[key][0]
                    continue
                cands = self.region_candidates(fix, c, i)
                best = None; bestv = math.inf
                for alt in cands:
                    if alt[1] < 0 and key not in fix and alt[0] != prev:
                        continue
                    v = max([row_value(r, Z) for r in M.region_rows(c, i, alt)] + [0.0])
                    if v < bestv:
                        best, bestv = alt, v
                if best is None or bestv > tol:
                    viol.append((i, 0, key, cands))
                    # continue with the least violated one so that later checks have a region
                    if best is None:
                        best = cands[0] if cands else (prev, 0)
                comp[key] = best
                prev = best[0]
        reg = {(c, i): comp[('r', c, i)][0] for c in range(I.C) for i in range(1, I.N)}
        for c in range(I.C):
            for i in range(1, I.N):
                j = reg[(c, i)]
                runfixed = ('r', c, i) not in fix
                if I.E >= 1:
                    for pt in range(5):
                        key = ('e', c, i, pt)
                        if I.E == 1:
                            if pt > 0 and runfixed:
                                v = max(row_value(r, Z) for r in M.env_rows(c, i, pt, 0, j))
                                if v > tol:
                                    viol.append((i, 0, ('r', c, i), self.region_candidates(fix, c, i)))
                            continue
                        if key in fix and not (pt > 0 and runfixed):
                            continue
                        vals = [max(row_value(r, Z) for r in M.env_rows(c, i, pt, e, j)) for e in range(I.E)]
                        if key in fix:
                            ok = vals[fix[key]] <= tol
                        else:
                            ok = min(vals) <= tol
                            comp[key] = int(np.argmin(vals))
                        if not ok:
                            if pt > 0 and runfixed:
                                viol.append((i, 0, ('r', c, i), self.region_candidates(fix, c, i)))
                            else:
                                viol.append((i, 1, key, list(np.argsort(vals))))
                for o in range(I.O):
                    L = len(I.obs[o][i])
                    for pt in range(5):
                        key = ('o', c, o, i, pt)
                        if key in fix and (fix[key] >= L or not (pt > 0 and runfixed)):
                            continue
                        vals = [row_value(M.obs_rows(c, o, i, pt, k, j)[0], Z) for k in range(L)]
                        if key in fix:
                            ok = vals[fix[key]] <= tol
                        else:
                            ok = min(vals) <= tol
                            comp[key] = int(np.argmin(vals))
                        if not ok:
                            if pt > 0 and runfixed:
                                viol.append((i, 0, ('r', c, i), self.region_candidates(fix, c, i)))
                            else:
                                alts = [int(k) for k in np.argsort(vals)]
                                if I.obs_soft[o]:
                                    alts.append(L)
                                viol.append((i, 2, key, alts))
        for p, (c1, c2) in enumerate(M.pairs):
            for i in range(1, I.N):
                j1, j2 = reg[(c1, i)], reg[(c2, i)]
                for grp in range(4):
                    key = ('a', p, i, grp)
                    need1 = grp in (2, 3); need2 = grp in (1, 3)
                    unf = [c for c, need in ((c1, need1), (c2, need2)) if need and ('r', c, i) not in fix]
                    if key in fix and not unf:
                        continue
                    # satisfied with zero slack <=> last row (the strict one) holds
                    vals = [row_value(M.c2c_rows(p, i, grp, alt, j1, j2)[-1], Z) if len(M.c2c_rows(p, i, grp, alt, j1, j2)) == 1
                            or True else 0 for alt in range(4)]
                    # rows[-1] is the zero-slack row when a soft row exists, else the hard one
                    if key in fix:
                        ok = vals[fix[key]] <= tol
                    else:
                        ok = min(vals) <= tol
                        comp[key] = int(np.argmin(vals))
                    if not ok:
                        if unf:
                            viol.append((i, 0, ('r', unf[0], i), self.region_candidates(fix, unf[0], i)))
                        else:
                            viol.append((i, 3, key, [int(k) for k in np.argsort(vals)]))
        viol.sort(key=lambda v: (v[0], v[1]))
        return viol, comp

    def region_candidates(self, fix, c, i):
        """alternatives of ('r',c,i) compatible with fixed neighbours (A5 freeze: slow => same region as i-1)."""
        M, I = self.M, self.I
        prev = (I.init_region[c] - 1, 0) if i == 1 else fix.get(('r', c, i - 1))
        nxt = fix.get(('r', c, i + 1)) if i + 1 < I.N else None
        out = []
        for alt in M.region_alts(c, i):
            if alt[1] < 0 and prev is not None and prev[0] != alt[0]:
                continue
            if nxt is not None and nxt[1] < 0 and nxt[0] != alt[0]:
                continue
            out.append(alt)
        return out

    # ----- main loop
    def solve(self):
        root = Node({}, -math.inf, None, 0)
        heap = [(-math.inf, 0, root)]
        cnt = itertools.count(1)
        best_bound = -math.inf
        while heap:
            bnd, _, nd = heapq.heappop(heap)
            best_bound = bnd if self.inc == math.inf else min(bnd, self.inc)
            if self.inc < math.inf and (self.inc - bnd) <= self.gap * (1e-10 + abs(self.inc)):
                best_bound = bnd
                break
            if self.nodes >= self.max_nodes:
                break
            self.nodes += 1
            rows = self.node_rows(nd.fix)
            r = solve_qp(self.M, rows, U0=nd.U if self.warm else None, **self.qp_args)
            if not r["ok"] and self.verbose:
                print("  NOT CONVERGED", nd.fix)
            self.qp_iters += r["it"]
            obj = r["obj"] + self.const_cost(nd.fix)
            if r["viol"] > FEAS_TOL or not r["ok"]:
                if self.verbose:
                    print("  node %d infeasible viol %.2e ok %s" % (self.nodes, r["viol"], r["ok"]))
                continue
            if obj >= self.inc * (1 - 1e-12) - 1e-12:
                continue
            viol, comp = self.complete(nd.fix, r["Z"])
            if not viol:
                self.inc = obj; self.inc_sol = (comp, r["Z"].copy())
                if self.verbose:
                    print("  node %d incumbent %.6f depth %d open %d" % (self.nodes, obj, nd.depth, len(heap)))
                continue
            _, _, key, alts = viol[0]
            for alt in alts:
                f2 = dict(nd.fix); f2[key] = alt
                heapq.heappush(heap, (obj, next(cnt), Node(f2, obj, r["U"], nd.depth + 1)))
        else:
            best_bound = self.inc
        if self.inc < math.inf and (not heap or best_bound > self.inc):
            best_bound = self.inc
        return dict(obj=self.inc, bound=best_bound, nodes=self.nodes, iters=self.qp_iters, sol=self.inc_sol)
