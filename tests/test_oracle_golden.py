"""Pins the CPU oracle against the reference's own known answers."""
import re

import numpy as np
import pytest

from helpers import CONT_FIELDS, dat_path, k3, k3_results, load_params
from planner_miqp_amd.ctypes_types import RawResults


def test_k1_raw_model_sizes(oracle):
    """test/cplex_wrapper_test.cc:866-871: 12361 rows, 1240 binaries, 340 continuous, 29834 non-zeros"""
    h = oracle.from_dat(dat_path("cplexmodel_testcase.dat"))
    s = oracle.sizes(h)
    g = k3()["sizes"]
    assert s == dict(rows=g["NrConstraints"], bin=g["NrBinaryVariables"], cont=g["NrFloatVariables"], nnz=g["NonZeroCoefficients"])
    oracle.free(h)


def test_k3_vector_is_feasible_for_raw_model(oracle):
    """the reference's solution vector satisfies every raw big-M row to its print precision and evaluates to K2"""
    h = oracle.from_dat(dat_path("cplexmodel_testcase.dat"))
    r, g = k3_results()
    v, obj, worst = oracle.raw_eval(h, r, use_real_slack=False)
    assert v < 2e-3, worst          # 5 significant digits on values up to 22
    assert abs(obj - g["objective"]) < 5e-4
    oracle.free(h)


def test_k2_k3_fixed_binaries_reproduce_reference_solution(oracle):
    """with K3's binaries asserted the continuous optimum is unique: objective 9.57603 +- 1e-5 (cc:874) and the
    states of cc:283-456 to their printed digits"""
    h = oracle.from_dat(dat_path("cplexmodel_testcase.dat"))
    fx, g = k3_results()
    st, res, obj, it = oracle.solve_fixed(h, (1, 20, 32, 1, 1, 4), fx)
    assert st == 0
    assert abs(obj - g["objective"]) <= g["objective_tol"]
    for n in CONT_FIELDS:
        ref = np.array(g[n]).reshape(1, 20)
        tol = 1e-4 if not n.startswith("pos_x_front") else 6e-4   # 5 printed digits on values ~20
        assert np.abs(getattr(res, n) - ref).max() <= tol + 5e-5 * np.abs(ref).max(), n
    oracle.free(h)


def test_testcase_branch_and_bound_optimum(oracle):
    """B&B to 1e-6: optimum equals the reference incumbent 9.57603 (so CPLEX's 10%-gap answer was optimal),
    the returned vector is feasible for the raw model with its binaries, regions equal K3's"""
    h = oracle.from_dat(dat_path("cplexmodel_testcase.dat"))
    st, res, p = oracle.solve(h, (1, 20, 32, 1, 1, 4), gap=1e-6)
    g = k3()
    assert st == 0 and p.status in (101, 102)
    assert abs(p.objective - g["objective"]) <= g["objective_tol"]
    assert p.best_bound <= p.objective + 1e-9 and p.gap <= 1e-6
    v, obj, worst = oracle.raw_eval(h, res)
    assert v < 1e-5, worst
    assert abs(obj - p.objective) < 1e-6
    assert np.array_equal(res.active_region.reshape(-1), np.array(g["active_region"], dtype=np.int32))
    assert p.NrConstraints == 12361 and p.NrBinaryVariables == 1240 and p.NrFloatVariables == 340 and p.NonZeroCoefficients == 29834
    oracle.free(h)


def test_datfile_vs_cppinputs_same_objective(oracle):
    """test_hardcoded_data_versus_datfile (cc:474-505): both parameter sources give the same objective"""
    h1 = oracle.from_dat(dat_path("cplexmodel_testcase.dat"))
    p = load_params("cplexmodel_testcase.dat")
    h2 = oracle.from_params(p, 10)
    s1 = oracle.solve(h1, oracle.dims(p), gap=1e-6)
    s2 = oracle.solve(h2, oracle.dims(p), gap=1e-6)
    assert s1[0] == 0 and s2[0] == 0
    assert abs(s1[2].objective - s2[2].objective) < 1e-9
    oracle.free(h1); oracle.free(h2)


def test_sos_fixture_solves(oracle):
    """test_compare_sos (cc:604-635): test_sos.dat solves; SOS on/off cannot change the optimum of an exact solver"""
    h = oracle.from_dat(dat_path("test_sos.dat"))
    st, res, p = oracle.solve(h, (1, 20, 16, 0, 0, 0), gap=1e-4)
    assert st == 0
    v, obj, worst = oracle.raw_eval(h, res)
    assert v < 1e-5, worst
    oracle.free(h)


def test_shorter_budget_gives_worse_or_equal_answer(oracle):
    """test_overwrite_parameters (cc:821-842): a tighter budget cannot give a better objective"""
    h = oracle.from_dat(dat_path("cplexmodel_testcase.dat"))
    full = oracle.solve(h, (1, 20, 32, 1, 1, 4), gap=1e-6)
    short = oracle.solve(h, (1, 20, 32, 1, 1, 4), gap=1e-6, max_nodes=20)
    assert short[0] in (0, 3)
    if short[0] == 0:
        assert short[2].objective >= full[2].objective - 1e-9 and short[2].gap >= full[2].gap
    oracle.free(h)


def _parse_model_run(path):
    from miqp_py.dat import parse_dat
    txt = open(path).read()
    txt = re.sub(r"//[^\n]*", "", txt)
    return parse_dat(txt)


def test_k5_three_car_vector_is_feasible(oracle):
    """cplexmodel.dat + modelRun.txt: the only multi-car vector of the reference (3 cars, 8 steps, 11 pieces)
    satisfies the raw model to its 5-digit print precision: pins the A5/A6/A8 reading incl. the [c1, c2-1] indexing"""
    h = oracle.from_dat(dat_path("cplexmodel.dat"))
    d = _parse_model_run(dat_path("modelRun.txt"))
    Cn, N, R, E = 3, 8, 32, 11
    r = RawResults(Cn, N, R, E, 0, 0)
    for n in CONT_FIELDS:
        getattr(r, n)[...] = np.array(d[n], float).reshape(Cn, N)
    for n in ["notWithinEnvironmentRear", "notWithinEnvironmentFrontUbUb", "notWithinEnvironmentFrontLbUb",
              "notWithinEnvironmentFrontUbLb", "notWithinEnvironmentFrontLbLb"]:
        getattr(r, n)[...] = np.array(d[n], dtype=np.int32).reshape(Cn, E, N)
    r.active_region[...] = np.array(d["active_region"], dtype=np.int32).reshape(Cn, N, R)
    for n in ["region_change_not_allowed_x_positive", "region_change_not_allowed_y_positive", "region_change_not_allowed_x_negative",
              "region_change_not_allowed_y_negative", "region_change_not_allowed_combined"]:
        getattr(r, n)[...] = np.array(d[n], dtype=np.int32).reshape(Cn, N)
    r.car2car_collision[...] = np.array(d["car2car_collision"], dtype=np.int32).reshape(2, 2, N, 16)
    r.slackvars_real[...] = np.array(d["slackvars"], float).reshape(2, 2, N, 4)
    r.slackvars[...] = 0
    v, obj, worst = oracle.raw_eval(h, r, use_real_slack=True)
    assert v < 0.05, worst
    oracle.free(h)
