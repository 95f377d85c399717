"""Independent optimum / bound pins for small multi-car and obstacle instances (run in the build container only).

The RAW big-M model of an instance - the statement-by-statement LP dump of cplexmodel/*.mod that
miqp_solver_export_lp writes (the dump whose row / non-zero / binary counts equal the reference's 12361 / 29834 / 1240
for the testcase) - is read back by HiGHS (scipy's bundled HiGHS reads CPLEX LP files incl. the quadratic objective) and
solved by the plain branch and bound below: binaries relaxed to [0, 1], HiGHS' QP solver on every node, depth-first dives
with best-bound backtracking, branching on the most fractional binary.  Nothing of the solvers under test is involved:
no disjunctive reformulation, no presolve of ours, no interior point of ours.  What is committed is data:
tests/golden/highs_fixtures.json = generator spec (config tuple, seed), optimum, proven bound, node count.

    python tests/golden/make_highs_fixtures.py            # ~ minutes on 8 cores
"""
import heapq
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

INSTANCES = [  # (config tuple (cars, steps, regions, env pieces, obstacles), seed, modifier)
    # sizes the plain B&B below finishes (the big-M relaxations are weak: every extra step multiplies the tree); each worker
    # process is stopped after LIMIT_S and its instance left out.  Instances already in highs_fixtures.json are kept, not re-solved.
    ((2, 4, 16, 1, 0), 0, None), ((2, 3, 16, 1, 0), 1, "close"), ((2, 3, 16, 1, 0), 2, "cross"), ((2, 4, 16, 1, 0), 3, "cross"),
    ((1, 7, 32, 1, 1), 1, "obstacle"),
    # round 3: three and four cars, 32 regions with a two-piece environment, a soft obstacle, cars that have to accelerate hard
    # (the hull of the region boxes binds), an obstacle between two cars
    ((3, 2, 16, 1, 0), 1, "close"),
    ((2, 3, 32, 2, 0), 4, "close"), ((2, 3, 32, 2, 0), 5, "cross"), ((1, 5, 16, 1, 1), 2, "soft"), ((1, 4, 32, 1, 1), 3, "soft"),
    ((1, 6, 16, 1, 0), 7, "accel"), ((2, 4, 16, 1, 0), 6, "accel"), ((2, 3, 16, 2, 0), 8, "accel"),
    # (three cars x 3 steps, four cars x 2 / 3 steps with the cars 4.5 m apart - modifier "close" - and two cars with an obstacle
    # did not finish within an hour each: the same sizes with the generator's own spacing)
    ((4, 2, 16, 1, 0), 0, None), ((4, 2, 16, 1, 0), 1, "accel"), ((3, 3, 16, 1, 0), 0, None), ((3, 3, 16, 1, 0), 2, "accel"), ((4, 3, 16, 1, 0), 3, None),
]
LIMIT_S = int(os.environ.get('HIGHS_LIMIT_S', '3000'))   # per instance; the three / four cars x 3 steps cases were given 3 hours (HIGHS_LIMIT_S=10800)


def build(cfg, seed, mod):
    """generator instance; the modifiers put the cars / the obstacle where the disjunctions actually bind"""
    from planner_miqp_amd import synthetic
    p = synthetic.generate(cfg, seed, gap=1e-6, max_time=600)
    N = p.NumSteps
    if mod == "close":        # cars 4.5 m apart, the rear one faster: the merge is contested
        for c in range(p.NumCars):
            p.IntitialState[c, 0] = 4.5 * c
            p.x_ref[c] = p.IntitialState[c, 0] + (9.0 - 2.0 * c) * p.ts * np.arange(N); p.vx_ref[c] = 9.0 - 2.0 * c
            p.IntitialState[c, 1] = 8.0 - 1.5 * c
    if mod == "cross":        # car 1 starts behind car 0 in the target lane and wants to go faster
        p.IntitialState[1, 0] = p.IntitialState[0, 0] - 5.0; p.IntitialState[1, 1] = 9.0; p.IntitialState[0, 1] = 5.0
        p.x_ref[1] = p.IntitialState[1, 0] + 10.0 * p.ts * np.arange(N); p.vx_ref[1] = 10.0
        p.x_ref[0] = p.IntitialState[0, 0] + 5.0 * p.ts * np.arange(N); p.vx_ref[0] = 5.0
    if mod == "accel":        # slow start, fast reference: the acceleration boxes of the regions bind along the whole horizon
        for c in range(p.NumCars):
            p.IntitialState[c, 1] = 4.0 + 0.5 * c
            p.x_ref[c] = p.IntitialState[c, 0] + 10.0 * p.ts * np.arange(N); p.vx_ref[c] = 10.0
    if mod == "soft":         # the obstacle of "obstacle", declared soft (obstacle_environment_constraints.mod:85-91)
        p.obstacle_is_soft = [1] * p.nr_obstacles
    if mod in ("obstacle", "soft"):     # static box on the reference path a few metres ahead
        cx, cy, hl, hw = float(p.IntitialState[0, 0]) + 7.0, -1.75, 3.4, 1.9
        box = np.array([[cx - hl, cy - hw], [cx + hl, cy - hw], [cx + hl, cy + hw], [cx - hl, cy + hw]])
        p.ObstacleConvexPolygon = [[box.copy() for _ in range(N)] for _ in range(p.nr_obstacles)]
    from planner_miqp_amd import planner_core as K   # initial region consistent with the modified initial velocity
    F = np.asarray(p.fraction_parameters, float).reshape(-1, 4)
    for c in range(p.NumCars):
        j = K.calculate_region_idx(F, p.IntitialState[c, 1], p.IntitialState[c, 4])[0]
        p.initial_region[c] = j + 1; p.possible_region[c, j] = 1
    return p


def objective_constant(p):
    """sum W ref^2: the LP dump omits the constant of objective_function.mod:7-19 (inputs rounded to 10 decimals like the solver does)"""
    r = lambda a: np.round(np.asarray(a, float), 10)
    c = 0.0
    for W, ref in ((p.WEIGHTS_POS_X, p.x_ref), (p.WEIGHTS_VEL_X, p.vx_ref), (p.WEIGHTS_POS_Y, p.y_ref), (p.WEIGHTS_VEL_Y, p.vy_ref)):
        c += float((r(W)[:, None] * r(ref) ** 2).sum())
    return c


def solve_raw_miqp(lp_path, gap=1e-6, time_limit=2400.0, log=None):
    """plain B&B; a bound is only ever taken from a node HiGHS reports OPTIMAL (or from its parent), infeasibility only
    from a node it reports INFEASIBLE; where its QP solver gives up (degenerate big-M vertices: status "solve error")
    the node keeps its parent's bound and is branched further"""
    import scipy.sparse as sp
    from scipy.optimize._highspy import _core as hs
    h0 = hs._Highs()
    h0.setOptionValue("output_flag", False)
    assert h0.readModel(lp_path) == hs.HighsStatus.kOk
    model = h0.getModel(); lp = model.lp_
    n = lp.num_col_
    isbin = np.array([lp.integrality_[k] == hs.HighsVarType.kInteger for k in range(n)])
    bins = np.nonzero(isbin)[0].astype(np.int32)
    names = [lp.col_names_[k] for k in range(n)]
    # exact trivial presolve of the dump: singleton rows become column bounds, duplicated rows (OPL emits the A5 block
    # once per region) are kept once
    A = lp.a_matrix_
    M = sp.csc_matrix((np.array(A.value_), np.array(A.index_), np.array(A.start_)), shape=(lp.num_row_, n)).tocsr()
    rlo, rhi = np.array(lp.row_lower_), np.array(lp.row_upper_)
    lo0, hi0 = np.array(lp.col_lower_, float), np.array(lp.col_upper_, float)
    lo0[bins] = np.maximum(lo0[bins], 0.0); hi0[bins] = np.minimum(hi0[bins], 1.0)
    keep, seen = [], set()
    for r in range(M.shape[0]):
        s_, e_ = M.indptr[r], M.indptr[r + 1]
        idx, v = M.indices[s_:e_], M.data[s_:e_]
        nzm = v != 0; idx, v = idx[nzm], v[nzm]
        if len(idx) == 0:
            assert rlo[r] <= 1e-12 and rhi[r] >= -1e-12
            continue
        if len(idx) == 1:
            k, a = idx[0], v[0]
            l, u = (rlo[r] / a, rhi[r] / a) if a > 0 else (rhi[r] / a, rlo[r] / a)
            lo0[k] = max(lo0[k], l); hi0[k] = min(hi0[k], u)
            continue
        key = (tuple(idx), tuple(v), rlo[r], rhi[r])
        if key in seen:
            continue
        seen.add(key); keep.append(r)
    M2 = M[keep].tocsc()
    lp2 = hs.HighsLp()
    lp2.num_col_ = n; lp2.num_row_ = len(keep); lp2.col_cost_ = np.array(lp.col_cost_); lp2.col_lower_ = lo0; lp2.col_upper_ = hi0
    lp2.row_lower_ = rlo[keep]; lp2.row_upper_ = rhi[keep]; lp2.offset_ = lp.offset_
    lp2.a_matrix_.format_ = hs.MatrixFormat.kColwise; lp2.a_matrix_.num_col_ = n; lp2.a_matrix_.num_row_ = len(keep)
    lp2.a_matrix_.start_ = M2.indptr.astype(np.int32); lp2.a_matrix_.index_ = M2.indices.astype(np.int32); lp2.a_matrix_.value_ = M2.data
    m2 = hs.HighsModel(); m2.lp_ = lp2; m2.hessian_ = model.hessian_
    h = hs._Highs(); h.setOptionValue("output_flag", False)
    assert h.passModel(m2) == hs.HighsStatus.kOk
    free = np.array([k for k in bins if lo0[k] < hi0[k]], dtype=np.int32)
    ITOL = 1e-6
    stats = dict(err=0)

    def relax(fix):
        lo, hi = lo0.copy(), hi0.copy()
        for k, v in fix.items():
            lo[k] = hi[k] = v
        h.changeColsBounds(len(free), free, lo[free], hi[free])
        h.run()
        ms = h.getModelStatus()
        if ms == hs.HighsModelStatus.kOptimal:
            return "opt", h.getObjectiveValue(), np.array(h.getSolution().col_value)
        if ms == hs.HighsModelStatus.kInfeasible:
            return "inf", None, None
        stats["err"] += 1
        x = np.array(h.getSolution().col_value)
        return "err", None, (x if len(x) == n else None)

    def pick(x, fix):
        """most fractional free binary; region and car/car decisions first (they shape the rest)"""
        best, bk = -1.0, -1
        for k in free:
            if k in fix:
                continue
            f = min(x[k], 1.0 - x[k])
            if f <= ITOL:
                continue
            pri = 2.0 if names[k].startswith("active_region") else (1.5 if names[k].startswith("car2car") else 1.0)
            if f * pri > best:
                best, bk = f * pri, k
        return bk

    t0 = time.time()
    inc = np.inf
    heap = [(-np.inf, 0, {})]; cnt = 1; nodes = 0
    while heap:
        lb = heap[0][0]
        if inc < np.inf and inc - lb <= gap * (1e-10 + abs(inc)):
            break
        if time.time() - t0 > time_limit:
            return dict(status="time", objective=inc, bound=lb, nodes=nodes)
        b, _, fix = heapq.heappop(heap)
        while True:   # dive
            if inc < np.inf and b >= inc - gap * (1e-10 + abs(inc)):
                break
            st, obj, x = relax(fix); nodes += 1
            if st == "inf":
                break
            if st == "opt":
                if obj >= inc - gap * (1e-10 + abs(inc)):
                    break
                b = max(b, obj)
                k = pick(x, fix)
                if k < 0:
                    inc = obj
                    if log:
                        log("  incumbent %.9f after %d nodes (%.0f s)" % (inc, nodes, time.time() - t0))
                    break
            else:     # no verdict from the QP solver: keep the parent's bound, branch on any free binary
                k = pick(x, fix) if x is not None else -1
                if k < 0:
                    rest = [q for q in free if q not in fix]
                    if not rest:
                        return dict(status="error", detail="QP solver gave no verdict on a leaf")
                    k = rest[0]
            up = x is not None and x[k] >= 0.5
            other = dict(fix); other[k] = 0.0 if up else 1.0
            cnt += 1; heapq.heappush(heap, (b, cnt, other))
            fix = dict(fix); fix[k] = 1.0 if up else 0.0
    lb = min([inc] + [e[0] for e in heap]) if heap else inc
    return dict(status="optimal", objective=inc, bound=lb, nodes=nodes, seconds=time.time() - t0, qp_no_verdict=stats["err"])


def one(args):
    cfg, seed, mod = args
    import planner_miqp_amd as P
    p = build(cfg, seed, mod)
    w = P.CplexWrapper(); w.resetParameters(p)
    assert w._push_inputs() == 0
    with tempfile.TemporaryDirectory() as d:
        lp = os.path.join(d, "m.lp")
        assert P.load_library().miqp_solver_export_lp(w._h, lp.encode()) == 0
        r = solve_raw_miqp(lp, time_limit=max(600.0, LIMIT_S - 600.0))
    c0 = objective_constant(p)
    out = dict(config=list(cfg), seed=seed, modifier=mod, gap=1e-6, raw_sizes=w.rawSizes(), status=r["status"])
    if r["status"] == "optimal":
        out.update(objective=r["objective"] + c0, bound=r["bound"] + c0, nodes=r["nodes"], seconds=round(r["seconds"], 1), qp_no_verdict=r["qp_no_verdict"])
    print(out, flush=True)
    return out


def _worker(args, q):
    q.put(one(args))


def main():
    import multiprocessing as mp
    import planner_miqp_amd as P
    P.build_library()
    ctx = mp.get_context("fork")
    have = {}
    if os.path.exists(os.path.join(HERE, "highs_fixtures.json")):
        for r in json.load(open(os.path.join(HERE, "highs_fixtures.json")))["instances"]:
            have[(tuple(r["config"]), r["seed"], r["modifier"])] = r
    res, running, todo = [have[k] for k in INSTANCES if k in have], [], [k for k in INSTANCES if k not in have]
    while todo or running:
        while todo and len(running) < int(os.environ.get('WORKERS', '6')):
            q = ctx.Queue(); pr = ctx.Process(target=_worker, args=(todo.pop(0), q)); pr.start(); running.append((pr, q, time.time()))
        time.sleep(2)
        for item in list(running):
            pr, q, t0 = item
            if not q.empty():
                res.append(q.get()); pr.join(); running.remove(item)
            elif not pr.is_alive():
                running.remove(item)
            elif time.time() - t0 > LIMIT_S:
                pr.kill(); pr.join(); running.remove(item); print("stopped after the limit", flush=True)
    res = [r for r in res if r["status"] in ("optimal", "infeasible")]
    json.dump(dict(source="tests/golden/make_highs_fixtures.py: raw big-M LP dump solved by a plain B&B over HiGHS-QP (scipy %s)" % __import__("scipy").__version__,
                   instances=res), open(os.path.join(HERE, "highs_fixtures.json"), "w"), indent=1)
    print("written", len(res))


if __name__ == "__main__":
    main()
