"""Optimum / bound pins that do not share anything with the solvers under test: tests/golden/highs_fixtures.json holds, for
small multi-car and obstacle instances, the optimum a plain branch and bound over HiGHS' QP solver proved on the RAW big-M
model (the LP dump of cplexmodel/*.mod; generator tests/golden/make_highs_fixtures.py, run in the build container).  The CPU
oracle and the device solver must both bracket it: best_bound <= optimum <= objective, and at a tight gap agree with it."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
FIX = os.path.join(HERE, "golden", "highs_fixtures.json")


def fixtures():
    if not os.path.exists(FIX):
        return []
    return [f for f in json.load(open(FIX))["instances"] if f["status"] == "optimal"]


def build(f):
    import make_highs_fixtures as M
    return M.build(tuple(f["config"]), f["seed"], f["modifier"])


def test_fixture_file_is_there_and_covers_multi_car_cases():
    fx = fixtures()
    assert len(fx) >= 10 and sum(f["config"][0] >= 2 for f in fx) >= 3 and any(f["config"][4] >= 1 for f in fx)
    # config = (cars, steps past the first, regions, environment pieces, obstacles): three and four cars, the bench's 32
    # regions over a two-piece environment, a soft obstacle (the slack branch of obstacle_environment_constraints.mod:85-91)
    assert any(f["config"][0] == 3 for f in fx) and any(f["config"][0] == 4 for f in fx)
    assert any(f["config"][2] == 32 and f["config"][3] == 2 for f in fx)
    assert any(f["modifier"] == "soft" and f["config"][4] >= 1 for f in fx)
    assert all(f["bound"] <= f["objective"] + 1e-9 and f["objective"] - f["bound"] <= 2e-6 * max(1.0, abs(f["objective"])) for f in fx)


@pytest.mark.parametrize("k", range(16))
def test_oracle_brackets_the_independent_optimum(oracle, k):
    fx = fixtures()
    if k >= len(fx):
        pytest.skip("fewer fixtures")
    f = fx[k]; p = build(f)
    h = oracle.from_params(p, 10)
    assert oracle.sizes(h) == f["raw_sizes"]          # the same raw model the independent solver read
    st, res, pr = oracle.solve(h, oracle.dims(p), gap=1e-7, time_limit=120)
    assert st == 0
    tol = 2e-6 * max(1.0, abs(f["objective"]))
    assert pr.best_bound <= f["objective"] + tol and f["objective"] <= pr.objective + tol, (f["config"], pr.best_bound, f["objective"], pr.objective)
    assert abs(pr.objective - f["objective"]) <= tol
    v, obj, worst = oracle.raw_eval(h, res)
    assert v < 1e-5 and abs(obj - pr.objective) <= tol, worst
    oracle.free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("gap", [1e-7, 0.01])
def test_device_brackets_the_independent_optimum(oracle, gap):
    """device best_bound <= independent optimum <= device objective (and objective <= optimum (1 + gap)): a shared modelling
    error in the disjunctive reformulation that cut off feasible multi-car points would show here"""
    import planner_miqp_amd as P
    fx = fixtures()
    assert fx
    for f in fx:
        p = build(f); p.relative_mip_gap_tolerance = gap; p.max_solution_time = 60
        w = P.CplexWrapper(); w.resetParameters(p)
        assert int(w.callCplex()) == 0, f["config"]
        pr = w.getSolutionProperties()
        tol = 2e-6 * max(1.0, abs(f["objective"]))
        assert pr.best_bound <= f["objective"] + tol, (f["config"], f["seed"], pr.best_bound, f["objective"])
        assert f["objective"] <= pr.objective + tol and pr.objective <= f["objective"] * (1 + gap) + tol, (f["config"], f["seed"], pr.objective, f["objective"])
        h = oracle.from_params(p, 10)
        v, obj, worst = oracle.raw_eval(h, w.getRawResults())
        oracle.free(h)
        assert v < 1e-5 and abs(obj - pr.objective) <= tol, worst
