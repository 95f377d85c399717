"""include/cplex_wrapper.hpp - the class a maintainer drops into the reference tree in place of src/cplex_wrapper.{hpp,cpp} -
compiled and driven here against minimal stand-ins of Eigen and of the reference's data contract (tests/standin/, test
infrastructure only).  The CPU test checks that the header parses, links against libmiqp_gpu.so and fails loudly without a
device; the GPU test compares the adapter (column-major Eigen tensors) with the Python mirror on the same parameters."""
import os
import subprocess

import numpy as np
import pytest

import planner_miqp_amd as P
from planner_miqp_amd import synthetic
from helpers import dat_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SD = os.path.join(ROOT, "tests", "standin")


@pytest.fixture(scope="module")
def adapter():
    P.build_library()
    out = os.path.join(SD, "_build", "adapter_check")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    libdir = os.path.join(ROOT, "planner_miqp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + SD, "-I" + os.path.join(ROOT, "include"), "-o", out,
                           os.path.join(SD, "adapter_check.cpp"), "-L" + libdir, "-lmiqp_gpu", "-Wl,-rpath," + libdir])
    return out


def dump_params(p, path):
    """flat text form of a ModelParameters record for tests/standin/adapter_check.cpp"""
    with open(path, "w") as f:
        def w(name, a):
            a = np.atleast_2d(np.asarray(a, float))
            f.write("%s %d %d %s\n" % (name, a.shape[0], a.shape[1], " ".join(repr(float(x)) for x in a.ravel())))
        for n in ["max_solution_time", "relative_mip_gap_tolerance", "NumSteps", "ts", "nr_regions", "NumCars", "min_vel_x_y", "max_vel_x_y", "total_min_acc",
                  "total_max_acc", "total_min_jerk", "total_max_jerk", "maximum_slack", "WEIGHTS_SLACK", "WEIGHTS_SLACK_OBSTACLE", "minimum_region_change_speed",
                  "nr_obstacles", "max_lines_obstacles", "nr_environments"]:
            w(n, [[getattr(p, n)]])
        for n in ["agent_safety_distance", "agent_safety_distance_slack", "WEIGHTS_POS_X", "WEIGHTS_VEL_X", "WEIGHTS_ACC_X", "WEIGHTS_POS_Y", "WEIGHTS_VEL_Y",
                  "WEIGHTS_ACC_Y", "WEIGHTS_JERK_X", "WEIGHTS_JERK_Y", "WheelBase", "CollisionRadius", "initial_region"]:
            w(n, np.asarray(getattr(p, n), float).reshape(-1, 1))
        for n in ["IntitialState", "x_ref", "vx_ref", "y_ref", "vy_ref", "min_acc_x", "max_acc_x", "min_acc_y", "max_acc_y", "min_jerk_x", "max_jerk_x",
                  "min_jerk_y", "max_jerk_y", "possible_region", "fraction_parameters", "POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB",
                  "POLY_KAPPA_AX_MAX", "POLY_KAPPA_AX_MIN"]:
            w(n, getattr(p, n))
        w("obstacle_is_soft", np.asarray(list(p.obstacle_is_soft) + [0], float).reshape(-1, 1))
        for e in p.MultiEnvironmentConvexPolygon:
            w("env", e)
        for o in p.ObstacleConvexPolygon:
            f.write("obs_new 0 0\n")
            for t in o:
                w("obs", t)


def run(adapter, *args):
    r = subprocess.run([adapter] + list(args), capture_output=True, text=True, timeout=300)
    out = {}
    for l in r.stdout.splitlines():
        k = l.split()
        if k[0] == "status":
            out.update(status=int(k[1]), objective=float(k[3]), gap=float(k[5]), nnz=int(k[7]))
        elif k[0] == "copy":
            out.update(copy_before=int(k[2]), copy_after=int(k[4]), copy_objective=float(k[6]), copy_tmpfile_equal=int(k[8]))
        elif k[0] == "warm":
            out.update(warm_status=int(k[2]), warm_objective=float(k[4]))
        elif k[0] in ("pos_x", "region"):
            out[k[0]] = [float(x) for x in k[1:]]
    return out, r


def test_adapter_header_compiles_and_fails_loudly_without_a_device(adapter, tmp_path):
    import torch
    o, r = run(adapter, "dat", dat_path("cplexmodel_testcase.dat"))
    if torch.cuda.is_available():
        assert o["status"] == 0
    else:
        assert o["status"] == 2 and np.isnan(o["objective"]) and np.isnan(o["gap"]) and "no HIP device" in r.stderr   # FAILED_SEG_FAULT, NaN (cplex_wrapper.cpp:231-248)
    p = synthetic.generate("mini", 0)
    dump_params(p, str(tmp_path / "p.txt"))
    o, r = run(adapter, "cpp", str(tmp_path / "p.txt"))
    assert o["status"] == (0 if torch.cuda.is_available() else 2), r.stderr
    # copy constructor (src/cplex_wrapper.hpp:116-151): the configuration travels, the parameters do not
    assert o["copy_before"] == 2 and o["copy_tmpfile_equal"] == 1 and o["copy_after"] == o["status"], o


@pytest.mark.gpu
def test_adapter_matches_the_python_mirror(adapter, tmp_path):
    """the Eigen-typed adapter and the ctypes mirror are two marshalling layers over one C ABI: identical objective, states
    and regions (column-major <-> row-major conversion of every array), K2 through the DATFILE source, MIP start accepted"""
    o, r = run(adapter, "dat", dat_path("cplexmodel_testcase.dat"))
    assert o["status"] == 0 and abs(o["objective"] - 9.57603) <= 1e-5 and o["nnz"] == 29834, r.stdout + r.stderr   # cc:866-874
    for cfg, seed in (("mini", 2), ("mini1", 1), ((2, 6, 16, 2, 1), 0)):
        p = synthetic.generate(cfg, seed, gap=1e-6, max_time=60)
        dump_params(p, str(tmp_path / "p.txt"))
        o, r = run(adapter, "cpp", str(tmp_path / "p.txt"))
        w = P.CplexWrapper(); w.resetParameters(p)
        assert int(w.callCplex()) == o["status"] == 0, r.stderr
        pr = w.getSolutionProperties(); res = w.getRawResults()
        assert abs(o["objective"] - pr.objective) <= 1e-9 * max(1.0, abs(pr.objective))
        assert np.abs(np.array(o["pos_x"]) - res.pos_x.ravel()).max() <= 1e-7
        assert np.array_equal(np.array(o["region"], int), res.active_region.argmax(-1).ravel())
        assert o["warm_status"] == 0 and o["warm_objective"] <= o["objective"] * (1 + 1e-9)
        # the copy rounds its inputs to 12 instead of 10 decimals (cp2.precision_, src/cplex_wrapper.hpp:129): same optimum to ~1e-9
        assert o["copy_before"] == 2 and o["copy_after"] == 0 and abs(o["copy_objective"] - o["objective"]) <= 1e-6 * max(1.0, abs(o["objective"]))
