"""Planner core against the expectations of the reference's own CPLEX-free unit tests:
common/tests/parameter_preparer_test.cc, common/tests/regions_test.cc.  No GPU needed."""
import numpy as np
import pytest

import planner_miqp_amd as P
from planner_miqp_amd import planner_core as K
from planner_miqp_amd.ctypes_types import RawResults


@pytest.fixture(scope="module", autouse=True)
def _built():
    P.build_library()


def preparer(R=32, vfit=20, vmin=2):
    # parameter_preparer_test.cc:19-27: accLonMax 2, accLonMin -4, jerkLonMax 3, accLat 1.6, jerkLat 1.4
    return K.ParameterPreparer(R, vfit, vmin, 2, -4, 3, 1.6, 1.4)


def test_mean_angle_vector():
    """parameter_preparer_test.cc:17-40"""
    exp = [0.0982, 0.2945, 0.4909, 0.6872, 0.8836, 1.0799, 1.2763, 1.4726, 1.6690, 1.8653, 2.0617, 2.2580, 2.4544, 2.6507, 2.8471, 3.0434,
           3.2398, 3.4361, 3.6325, 3.8288, 4.0252, 4.2215, 4.4179, 4.6142, 4.8106, 5.0069, 5.2033, 5.3996, 5.5960, 5.7923, 5.9887, 6.1850]
    assert np.abs(preparer().GetMeanAngleVector() - np.array(exp)).max() < 1e-3


JX = [3.1228, 3.2772, 3.3057, 3.2072, 2.9854, 2.6489, 2.2106, 1.6873, 1.6873, 2.2106, 2.6489, 2.9854, 3.2072, 3.3057, 3.2772, 3.1228]
JY = [1.6873, 2.2106, 2.6489, 2.9854, 3.2072, 3.3057, 3.2772, 3.1228, 3.1228, 3.2772, 3.3057, 3.2072, 2.9854, 2.6489, 2.2106, 1.6873]


def test_jerk_limits():
    """parameter_preparer_test.cc:42-89 (the 32 values are the 16 above repeated)"""
    lim = preparer().CalculateJerkLimitsPerCar()
    assert np.abs(lim["max_x"] - np.array(JX + JX)).max() < 1e-3 and np.abs(lim["min_x"] + np.array(JX + JX)).max() < 1e-3
    assert np.abs(lim["max_y"] - np.array(JY + JY)).max() < 1e-3 and np.abs(lim["min_y"] + np.array(JY + JY)).max() < 1e-3


def test_acc_limits():
    """parameter_preparer_test.cc:91-135; the same numbers are the min/max_acc tables of cplexmodel_testcase.dat"""
    max_x = [2.1472, 2.3783, 2.5181, 2.5611, 2.5056, 2.3539, 2.1117, 1.7883, 1.9844, 2.6922, 3.2967, 3.7744, 4.1071, 4.2819, 4.2922, 4.1376,
             4.1376, 4.2922, 4.2819, 4.1071, 3.7744, 3.2967, 2.6922, 1.9844, 1.7883, 2.1117, 2.3539, 2.5056, 2.5611, 2.5181, 2.3783, 2.1472]
    min_x = [-4.1376, -4.2922, -4.2819, -4.1071, -3.7744, -3.2967, -2.6922, -1.9844, -1.7883, -2.1117, -2.3539, -2.5056, -2.5611, -2.5181,
             -2.3783, -2.1472, -2.1472, -2.3783, -2.5181, -2.5611, -2.5056, -2.3539, -2.1117, -1.7883, -1.9844, -2.6922, -3.2967, -3.7744,
             -4.1071, -4.2819, -4.2922, -4.1376]
    max_y = [1.7883, 2.1117, 2.3539, 2.5056, 2.5611, 2.5181, 2.3783, 2.1472, 2.1472, 2.3783, 2.5181, 2.5611, 2.5056, 2.3539, 2.1117, 1.7883,
             1.9844, 2.6922, 3.2967, 3.7744, 4.1071, 4.2819, 4.2922, 4.1376, 4.1376, 4.2922, 4.2819, 4.1071, 3.7744, 3.2967, 2.6922, 1.9844]
    min_y = [-1.9844, -2.6922, -3.2967, -3.7744, -4.1071, -4.2819, -4.2922, -4.1376, -4.1376, -4.2922, -4.2819, -4.1071, -3.7744, -3.2967,
             -2.6922, -1.9844, -1.7883, -2.1117, -2.3539, -2.5056, -2.5611, -2.5181, -2.3783, -2.1472, -2.1472, -2.3783, -2.5181, -2.5611,
             -2.5056, -2.3539, -2.1117, -1.7883]
    lim = preparer().CalculateAccLimitsPerCar()
    for k, e in (("max_x", max_x), ("min_x", min_x), ("max_y", max_y), ("min_y", min_y)):
        assert np.abs(lim[k] - np.array(e)).max() < 1e-3, k
    from helpers import load_params
    p = load_params("cplexmodel_testcase.dat")
    assert np.abs(np.asarray(p.max_acc_x).ravel() - lim["max_x"]).max() < 1e-4 and np.abs(np.asarray(p.min_acc_y).ravel() - lim["min_y"]).max() < 1e-4


def test_fraction_parameters_16_exact():
    """parameter_preparer_test.cc:184-222 requires a Frobenius distance below 1e-30, i.e. identical doubles"""
    exp = [20.0, 0.0, 18.477590650225736, 7.653668647301796, 18.477590650225736, 7.653668647301796, 14.142135623730951, 14.14213562373095,
           14.142135623730951, 14.14213562373095, 7.653668647301797, 18.477590650225736, 7.653668647301797, 18.477590650225736,
           1.2246467991473533e-15, 20.0, 1.2246467991473533e-15, 20.0, -7.653668647301794, 18.477590650225736, -7.653668647301794,
           18.477590650225736, -14.14213562373095, 14.142135623730951, -14.14213562373095, 14.142135623730951, -18.477590650225736,
           7.653668647301798, -18.477590650225736, 7.653668647301798, -20.0, 2.4492935982947065e-15, -20.0, 2.4492935982947065e-15,
           -18.477590650225736, -7.653668647301793, -18.477590650225736, -7.653668647301793, -14.142135623730955, -14.14213562373095,
           -14.142135623730955, -14.14213562373095, -7.6536686473018065, -18.47759065022573, -7.6536686473018065, -18.47759065022573,
           -3.673940397442059e-15, -20.0, -3.673940397442059e-15, -20.0, 7.6536686473018, -18.477590650225732, 7.6536686473018,
           -18.477590650225732, 14.142135623730947, -14.142135623730955, 14.142135623730947, -14.142135623730955, 18.47759065022573,
           -7.653668647301808, 18.47759065022573, -7.653668647301808, 20.0, 0.0]
    fp = preparer(16).GetFractionParameters()
    assert np.linalg.norm(fp - np.array(exp).reshape(16, 4)) < 1e-30


def test_fraction_parameters_32_match_the_fixture():
    """parameter_preparer_test.cc:137-182 (tolerance 1e-3 on the norm); the .dat fixture carries the same table"""
    from helpers import load_params
    p = load_params("cplexmodel_testcase.dat")
    assert np.linalg.norm(preparer(32).GetFractionParameters() - np.asarray(p.fraction_parameters).reshape(32, 4)) < 1e-3


def test_calculate_region_idx():
    """regions_test.cc:19-61, 187-203"""
    F = preparer().GetFractionParameters()
    assert K.calculate_region_idx(F, 0.1, 0.01) == [0]
    assert K.calculate_region_idx(F, 0.1951, 0.9808) == [6, 7]
    assert K.calculate_region_idx(F, 0.1, 0.9) == [7]
    assert K.calculate_region_idx(F, -0.1, -0.9) == [23]
    assert K.calculate_region_idx(F, 0.1, -0.01) == [31]


def test_calculate_possible_regions():
    """regions_test.cc:97-185: heading 0.1 rad lies in region 0; headings 0.3 / 0.5 / 0.7 give regions 1, 2, 3 and not 0"""
    F = preparer().GetFractionParameters()
    assert 0 in K.calculate_possible_regions(F, [0.1, 0.1, 0.1])
    assert 31 in K.calculate_possible_regions(F, [-0.1, -0.1, -0.1])
    r = K.calculate_possible_regions(F, [0.3, 0.5, 0.7])
    assert 0 not in r and {1, 2, 3} <= r


def test_reserve_neighbor_regions():
    """regions_test.cc:205-288"""
    r = np.zeros((2, 32), dtype=np.int32); r[0, 2] = r[0, 3] = 1; r[1, 4] = r[1, 5] = 1
    assert K.reserve_neighbor_regions(r, 0, 1) and r[0, :6].tolist() == [0, 1, 1, 1, 1, 0]
    assert K.reserve_neighbor_regions(r, 1, 1) and r[1, 2:8].tolist() == [0, 1, 1, 1, 1, 0]
    r = np.zeros((2, 32), dtype=np.int32); r[0, 0] = r[0, 1] = 1; r[1, 30] = r[1, 31] = 1
    assert K.reserve_neighbor_regions(r, 0, 1) and r[0, 30:].tolist() == [0, 1] and r[0, :4].tolist() == [1, 1, 1, 0]
    assert K.reserve_neighbor_regions(r, 1, 1) and r[1, 28:].tolist() == [0, 1, 1, 1] and r[1, :2].tolist() == [1, 0]
    r = np.zeros((1, 32), dtype=np.int32); r[0, 0] = r[0, 1] = 1
    assert K.reserve_neighbor_regions(r, 0, 2) and r[0, 29:].tolist() == [0, 1, 1] and r[0, :5].tolist() == [1, 1, 1, 1, 0]


def test_calculate_warmstart_shift_and_quirks():
    """MiqpPlanner::CalculateWarmstart (src/miqp_planner.cpp:787-1051): shift by one step, Euler step for the last
    state, region-change flags recomputed, last-step binaries copied unshifted, last active_region row all zero"""
    rng = np.random.default_rng(5)
    Cn, N, R, E, O, L = 2, 6, 8, 2, 1, 4
    a = RawResults(Cn, N, R, E, O, L)
    for n in ["u_x", "u_y", "pos_x", "vel_x", "acc_x", "pos_y", "vel_y", "acc_y", "pos_x_front_UB", "pos_x_front_LB", "pos_y_front_UB", "pos_y_front_LB"]:
        getattr(a, n)[...] = rng.normal(size=(Cn, N))
    for n in ["notWithinEnvironmentRear", "notWithinEnvironmentFrontUbUb", "notWithinEnvironmentFrontLbUb", "notWithinEnvironmentFrontUbLb",
              "notWithinEnvironmentFrontLbLb", "active_region", "region_change_not_allowed_x_positive", "region_change_not_allowed_combined",
              "deltacc", "deltacc_front", "car2car_collision", "slackvars"]:
        arr = getattr(a, n); arr[...] = rng.integers(0, 2, size=arr.shape)
    ts, vm = 0.25, 2.0
    w = K.calculate_warmstart(a, ts, vm)
    for n in ["u_x", "pos_x", "vel_y", "acc_y", "pos_x_front_UB", "pos_y_front_LB"]:
        assert np.array_equal(getattr(w, n)[:, :N - 1], getattr(a, n)[:, 1:]), n
    assert np.all(w.u_x[:, N - 1] == 0) and np.all(w.u_y[:, N - 1] == 0)
    assert np.allclose(w.pos_x[:, N - 1], w.pos_x[:, N - 2] + ts * w.vel_x[:, N - 2]) and np.allclose(w.vel_y[:, N - 1], w.vel_y[:, N - 2] + ts * w.acc_y[:, N - 2])
    assert np.allclose(w.acc_x[:, N - 1], w.acc_x[:, N - 2] + ts * w.u_x[:, N - 2]) and np.allclose(w.pos_y_front_UB[:, N - 1], w.pos_y_front_UB[:, N - 2] + ts * w.vel_y[:, N - 2])
    assert np.array_equal(w.region_change_not_allowed_x_positive[:, N - 1], (w.vel_x[:, N - 1] <= vm).astype(np.int32))
    assert np.array_equal(w.notWithinEnvironmentRear[:, :, :N - 1], a.notWithinEnvironmentRear[:, :, 1:])
    assert np.array_equal(w.notWithinEnvironmentRear[:, :, N - 1], a.notWithinEnvironmentRear[:, :, N - 1])      # copied unshifted (:951-963)
    assert np.array_equal(w.active_region[:, :N - 1], a.active_region[:, 1:]) and not w.active_region[:, N - 1].any()  # :982 has no effect
    assert np.array_equal(w.car2car_collision[:, :, :N - 1], a.car2car_collision[:, :, 1:]) and np.array_equal(w.car2car_collision[:, :, N - 1], a.car2car_collision[:, :, N - 1])
    assert np.array_equal(w.deltacc[:, :, :N - 1], a.deltacc[:, :, 1:]) and np.array_equal(w.deltacc_front[:, :, N - 1], a.deltacc_front[:, :, N - 1])


def test_fitting_polynomial_parameters_match_the_reference_unit_test():
    """common/tests/fitting_polynomial_parameters_test.cc:17-167 (EXPECT_EQ: identical doubles; EXPECT_NEAR where the
    reference uses it): spot values of the (32, 20, 2), (16, 20, 2) and (16, 10, 2) fits, the exact first row block of
    POLY_SINT_UB_32, valid / invalid parameter pairs"""
    f = K.FittingPolynomialParameters(32, 20, 2)
    m = f.GetPOLY_SINT_UB(); assert m.shape == (32, 3) and m[0, 0] == 0.18825 and m[31, 2] == 0.049029
    should = np.array([0.18825, -0.0091624, 0.05868, 0.38235, -0.019062, 0.050001, 0.55699, -0.023663, 0.036849, 0.71125, -0.023834, 0.024695,
                       0.83632, -0.019361, 0.013151, 0.93174, -0.019049, 0.0079673, 0.98244, -0.015382, 0.0037045, 1.0047, -0.0055247, 0.00032035]).reshape(8, 3)
    assert np.allclose(m[:8], should, rtol=1e-12, atol=0)
    m = f.GetPOLY_SINT_LB(); assert m[0, 0] == -0.005 and m[31, 2] == 0.056807
    m = f.GetPOLY_COSS_UB(); assert m[0, 0] == 1.005 and m[31, 2] == 0.0055247
    m = f.GetPOLY_COSS_LB(); assert m[0, 0] == 0.97651 and m[31, 2] == 0.0055499
    m = f.GetPOLY_KAPPA_AX_MAX(); assert m[0, 0] == -1.0225 and m[31, 2] == -0.1005
    m = f.GetPOLY_KAPPA_AX_MIN(); assert m[0, 0] == 1.7056 and m[31, 2] == 0.12325
    m16 = K.FittingPolynomialParameters(16, 20, 2).GetPOLY_SINT_UB()
    assert m16[0, 0] == 0.348508875688441 and m16[15, 0] == 0.005000000000002289 and m16[0, 1] == -0.017175443784421922 and m16[15, 2] == 0.04559607525876267
    g = K.FittingPolynomialParameters(16, 10, 2)
    m = g.GetPOLY_KAPPA_AX_MIN(); assert abs(m[0, 0] - 1.72762) < 1e-4 and abs(m[15, 2] - 0.249466) < 1e-4
    m = g.GetPOLY_COSS_LB(); assert m[0, 0] == 0.9317320781172977 and m[15, 2] == 0.024320547171933854
    for bad in ((32, 10, 3), (64, 20, 2), (48, 10, 1)):
        with pytest.raises(ValueError, match="Invalid number of regions or velocity!"):
            K.FittingPolynomialParameters(*bad)
    for ok in ((16, 10, 1), (16, 10, 2), (16, 20, 2), (32, 10, 1), (32, 10, 2), (32, 20, 2), (64, 10, 1)):
        assert K.FittingPolynomialParameters(*ok).GetPOLY_COSS_UB().shape == (ok[0], 3)


def test_fixture_tables_are_the_fitted_variants():
    """the region tables of the reference's .dat fixtures are the (32, 20, 2) and (16, 20, 2) fits (to the 6 digits OPL
    prints), and the generator's 64-region table is the (64, 10, 1) fit"""
    import json, os
    data = os.path.join(os.path.dirname(K.__file__), "data")
    names = ["POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX", "POLY_KAPPA_AX_MIN"]
    for R, combo, tol in ((32, (32, 20, 2), 1e-5), (16, (16, 20, 2), 1e-3), (64, (64, 10, 1), 0.0)):   # test_sos.dat prints 3-4 digits
        t = json.load(open(os.path.join(data, "region_tables_%d.json" % R)))
        f = K.FittingPolynomialParameters(*combo)
        for n in names:
            a = np.array(t[n], float).reshape(R, 3); b = getattr(f, "Get" + n)()
            assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (R, n, np.abs(a - b).max())


# ---- ReferenceTrajectoryGenerator on a polyline: the expectations of common/tests/reference_trajectory_generator_test.cc
def test_reference_generator_straight_ref_line():
    """:22-59 - start at 5 m/s, desired 10 m/s reached 0.1 m ahead: velocity and integrated positions"""
    from planner_miqp_amd import planner_core as K
    dt = 0.2
    t = K.reference_trajectory([[0, 0], [50, 0], [100, 0]], [0, 0, 0, 0, 5.0], dt, 20, 0.2, 10.0, 0.1, 1.0, True)
    assert abs(t[0, 4] - 5.0) < 1e-3 and abs(t[1, 4] - 10.0) < 1e-3 and abs(t[-1, 4] - 10.0) < 1e-3
    assert abs(t[1, 1] - 5.0 * dt) < 1e-3 and abs(t[2, 1] - (5.0 * dt + 10.0 * dt)) < 1e-3
    assert np.allclose(t[:, 0], dt * np.arange(20)) and np.allclose(t[:, 2], 0) and np.allclose(t[:, 3], 0)


def test_reference_generator_starting():
    """:61-92 - start at 0.1 m/s: the velocity rises at once and ends at the desired one"""
    from planner_miqp_amd import planner_core as K
    t = K.reference_trajectory([[0, 0], [50, 0], [100, 0]], [0, 0, 0, 0, 0.1], 0.2, 20, 0.2, 10.0, 0.1, 1.0, True)
    assert abs(t[0, 4] - 0.1) < 1e-3 and t[1, 4] > 0.1 and abs(t[-1, 4] - 10.0) < 1e-3


def test_reference_generator_stops_on_a_short_line():
    """:135-166 (onept) - desired velocity 0 on a 2 m line: one step of travel at the current speed, then standstill"""
    from planner_miqp_amd import planner_core as K
    dt = 0.2
    t = K.reference_trajectory([[1, 0], [2, 0], [3, 0]], [0, 1.0, 1.0, 0, 5.0], dt, 20, 0.2, 0.0, 0.1, 1.0, True)
    assert abs(t[0, 4] - 5.0) < 1e-3
    assert abs(t[1, 1] - (1.0 + 5.0 * dt)) < 1e-3 and abs(t[19, 1] - (1.0 + 5.0 * dt)) < 1e-3


def test_reference_generator_follows_a_bend():
    """curved reference line of :94-136.  bark smooths the line with a spline before walking it (SmoothLine); the restatement
    passes the natural cubic spline through the vertices: the curvature of the bend at x = 5 .. 10 reaches back over the first
    span, so that with the curvature-dependent limit switched on the velocity one step ahead (x = 2) is already BELOW the desired
    one - the reference's own expectation `traj(1).v < vel_desired` (:135), not weakened"""
    from planner_miqp_amd import planner_core as K
    line = [[0, 0], [5, 0], [10, 1.5], [20, 1.5], [30, 1.5], [40, 1.5], [50, 1.5], [60, 1.5]]
    u = K.reference_trajectory(line, [0, 0, 0, 0, 10.0], 0.2, 20, 0.2, 10.0, 0.1, 1.8, False)
    assert abs(u[0, 4] - 10.0) < 1e-3 and np.allclose(u[1:, 4], 10.0)               # :131 and the unlimited profile
    assert np.all(np.diff(u[:, 1]) > 0) and abs(u[-1, 2] - 1.5) < 0.05              # walks forward, ends on the upper straight
    assert 0.1 < u[4, 3] < 0.45 and abs(u[-1, 3]) < 0.02                             # heading on the ramp (atan(1.5 / 5) = 0.29), level again at the end
    t = K.reference_trajectory(line, [0, 0, 0, 0, 10.0], 0.2, 20, 0.2, 10.0, 0.1, 1.8, True)
    assert abs(t[0, 4] - 10.0) < 1e-3
    assert t[1, 4] < 10.0                                                            # :135, strict
    assert np.all(t[:, 4] <= 10.0 + 1e-12) and t[:, 4].min() > 1.0
    # a straight line is reproduced exactly by the spline: nothing changes for the straight-line tests above and for K8
    v = K.reference_trajectory([[0, 0], [5, 0], [30, 0], [300, 0]], [0, 0, 0, 0, 10.0], 0.2, 20, 0.2, 10.0, 0.1, 1.8, True)
    assert np.allclose(v[1:, 4], 10.0) and np.allclose(v[:, 2], 0.0) and np.allclose(v[:, 3], 0.0)


def test_environment_pieces_pose_check_and_obstacles_of_the_planner():
    """MiqpPlanner's environment and obstacle bookkeeping on convex pieces (src/miqp_planner.cpp:405-488, 490-537, 617-629,
    654-685, 1053-1115, 1248-1306): pieces are chosen by the reference trajectories, the initial pose must lie within one,
    obstacles outside the environment are refused with id -1, environment binaries of a warm start follow the piece ids"""
    from planner_miqp_amd import planner_core as K
    from planner_miqp_amd.ctypes_types import RawResults
    A = [[-10, -4], [40, -4], [40, 4], [-10, 4]]; B = [[30, -4], [80, -4], [80, 4], [30, 4]]; Cc = [[200, -4], [300, -4], [300, 4], [200, 4]]
    assert K.select_environment([A, B, Cc], [np.array([[0.0, 0.0], [20.0, 0.0]])]) == [0]
    assert K.select_environment([A, B, Cc], [np.array([[0.0, 0.0], [35.0, 0.0]]), np.array([[70.0, 1.0], [75.0, 1.0]])]) == [0, 1]
    assert K.select_environment([A, B, Cc], [np.array([[100.0, 0.0], [120.0, 0.0]])]) == []
    ob_in = [np.array([[10, -0.5], [11, -0.5], [11, 0.5], [10, 0.5]], float)] * 20
    ob_out = [np.array([[100, -0.5], [101, -0.5], [101, 0.5], [100, 0.5]], float)] * 20
    ob_later = [np.array([[100 - 4.0 * i, -0.5], [101 - 4.0 * i, -0.5], [101 - 4.0 * i, 0.5], [100 - 4.0 * i, 0.5]], float) for i in range(20)]
    assert K.obstacle_intersects_environment([A], ob_in, True) and not K.obstacle_intersects_environment([A], ob_out, True)
    assert K.obstacle_intersects_environment([A], ob_later, False) and not K.obstacle_intersects_environment([A], ob_later, True)   # static: step 0 only
    assert K.obstacle_intersects_environment([], ob_out, True)                                                        # empty environment admits everything

    pl = K.MiqpPlanner(mapPieces=[A, B, Cc])
    idx = pl.AddCar([0, 5, 0, 0, 0.01, 0], [[0, 0], [100, 0]], 5, 1, 0.0, True)
    pl.ResetEnvironment(pl.CalculateReferenceTrajectoriesLongerHorizon())
    p = pl.GetParameters()
    assert p.nr_environments == 1 and pl._env_ids == [0]                       # 24 steps at 5 m/s stay inside the first piece
    assert K.initial_pose_check(p) is None
    assert pl.AddObstacle(ob_out, False, True) == -1 and p.nr_obstacles == 0   # outside the environment of the last reset
    assert pl.AddObstacle(ob_in, False, False) == 0 and p.nr_obstacles == 1 and p.max_lines_obstacles == 4
    assert pl.AddObstacle(ob_later, True, False) == 1 and p.obstacle_is_soft == [0, 1]
    pl.UpdateObstacle(0, ob_later); assert np.allclose(p.ObstacleConvexPolygon[0][3], ob_later[3])
    with pytest.raises(NotImplementedError):
        pl.RemoveObstacle(0)
    pl.RemoveAllObstacles(); assert p.nr_obstacles == 0 and p.max_lines_obstacles == 0 and p.ObstacleConvexPolygon == []
    # the rear point inside, the front point (2.8 m ahead) outside; then the rear point on the boundary (within is strict)
    p.IntitialState[0] = [38.5, 5, 0, 0, 0.0, 0]; assert K.initial_pose_check(p) == (0, "front")
    p.IntitialState[0] = [-10.0, 5, 0, 0, 0.0, 0]; assert K.initial_pose_check(p) == (0, "rear")
    # inflated rectangle of CreateMiqpObstacle: 1 x 1 box at (20, -1.5), collision radius 1 -> 3 x 3, counter-clockwise
    ob = pl.CreateMiqpObstacle(np.tile([0.0, 20.0, -1.5, 0.0, 0.0], (20, 1)), [[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]])
    q = ob[0]; assert np.allclose(sorted(q[:, 0]), [18.5, 18.5, 21.5, 21.5]) and np.allclose(sorted(q[:, 1]), [-3.0, -3.0, 0.0, 0.0])
    assert 0.5 * np.sum(q[:, 0] * np.roll(q[:, 1], -1) - np.roll(q[:, 0], -1) * q[:, 1]) > 0
    # EnvironmentWarmstart: piece ids (0, 1) -> (1, 2): piece 1 keeps its column for the steps 0 .. N-2, the rest is 1
    last = RawResults(1, 20, 16, 2, 0, 0)
    for nm in ("notWithinEnvironmentRear", "notWithinEnvironmentFrontUbUb", "notWithinEnvironmentFrontLbUb", "notWithinEnvironmentFrontUbLb", "notWithinEnvironmentFrontLbLb"):
        getattr(last, nm)[...] = 1; getattr(last, nm)[0, 1, :] = 0
    out = K.environment_warmstart(last, [0, 1], [1, 2])
    assert out.dims[3] == 2 and np.all(out.notWithinEnvironmentRear[0, 0, :19] == 0) and out.notWithinEnvironmentRear[0, 0, 19] == 1
    assert np.all(out.notWithinEnvironmentRear[0, 1] == 1) and np.all(out.notWithinEnvironmentFrontLbLb[0, 0, :19] == 0)


def test_add_car_fills_the_model_like_update_car():
    """MiqpPlanner::AddCar / UpdateCar (src/miqp_planner.cpp:180-390) on the start of the C-API test
    (test/miqp_planner_c_api_test.cc:164-172): references, possible regions, weights, limits"""
    from planner_miqp_amd import planner_core as K
    pl = K.MiqpPlanner()
    idx = pl.AddCar([0, 0, 0, 1, 0.01, 0], [[0, 0], [5, 0], [30, 0]], 5, 1, 0.0, True)
    p = pl.GetParameters()
    assert idx == 0 and p.NumCars == 1 and p.NumSteps == 20 and p.nr_regions == 16 and p.nr_environments == 0
    assert p.x_ref.shape == (1, 20) and abs(p.x_ref[0, 0]) < 1e-12 and np.all(np.diff(p.x_ref[0]) > 0) and np.allclose(p.y_ref[0, 1:], 0)
    assert abs(p.vx_ref[0, -1] - 5.0) < 1e-9 and np.allclose(p.vy_ref[0, 1:], 0)        # desired velocity reached 1 m ahead
    assert p.possible_region[0, 0] == 1 and p.possible_region[0, 4] == 1                   # heading of the reference and of the start (vy > 0)
    assert p.WEIGHTS_POS_X[0] == 1.0 and p.WEIGHTS_JERK_X[0] == 0.5 and p.WEIGHTS_VEL_X[0] == 0.0      # lambda 0.5 x (2, 1, 0)
    assert abs(p.total_max_jerk - (max(p.max_jerk_x.max(), p.max_jerk_y.max()) + 1e-6)) < 1e-12
    assert p.WheelBase[0] == 2.8 and p.CollisionRadius[0] == 1.0


def test_obstacle_region_of_interest_filter():
    """the obstacle_roi_filter of MiqpPlanner (src/miqp_planner.cpp:380-387 UpdateCar, :1278-1288 ObstacleIntersectsEnvironment,
    :1308-1335 UpdateObstaclesROI; settings src/miqp_planner_settings.h:74-77, the values of test/miqp_planner_test.cc:65-68):
    the region around the ego car exactly as the reference computes it, and what it filters"""
    # the vertices as written in the reference: side offsets +-(sin theta, cos theta) * side
    x, y, th, behind, front, side = 3.0, -2.0, 0.3, 10.0, 100.0, 15.0
    roi = K.obstacles_roi(x, y, th, behind, front, side)
    fx, fy = x + np.cos(th) * front, y + np.sin(th) * front
    rx, ry = x + np.cos(th + np.pi) * behind, y + np.sin(th + np.pi) * behind
    want = [[fx + np.sin(th) * side, fy + np.cos(th) * side], [fx - np.sin(th) * side, fy - np.cos(th) * side],
            [rx - np.sin(th) * side, ry - np.cos(th) * side], [rx + np.sin(th) * side, ry + np.cos(th) * side]]
    np.testing.assert_allclose(roi, want, rtol=0, atol=1e-12)
    # heading along x: the rectangle [x - behind, x + front] x [y - side, y + side]
    roi0 = K.obstacles_roi(0.0, 0.0, 0.0, 10.0, 100.0, 15.0)
    np.testing.assert_allclose(roi0, [[100, 15], [100, -15], [-10, -15], [-10, 15]], atol=1e-9)
    road = [[-200, -4], [400, -4], [400, 4], [-200, 4]]
    box = lambda cx: np.array([[cx, -0.5], [cx + 1, -0.5], [cx + 1, 0.5], [cx, 0.5]], float)
    near, behind_far, ahead_far = [box(50.0)] * 20, [box(-60.0)] * 20, [box(150.0)] * 20
    approaching = [box(150.0 - 5.0 * i) for i in range(20)]                     # enters the region at step 10
    assert K.obstacle_intersects_environment([road], near, True, roi0)
    assert not K.obstacle_intersects_environment([road], behind_far, True, roi0) and K.obstacle_intersects_environment([road], behind_far, True)
    assert not K.obstacle_intersects_environment([road], ahead_far, False, roi0)   # never inside the region
    assert K.obstacle_intersects_environment([road], approaching, False, roi0)     # a moving obstacle is looked at step by step
    assert not K.obstacle_intersects_environment([road], approaching, True, roi0)  # a static one at step 0 only
    assert K.obstacle_intersects_environment([], behind_far, True, roi0)           # the empty environment admits everything, region or not
    touching = [box(100.0)] * 20                                                  # shares the edge x = 100 with the region: intersects (boost: touching counts)
    assert K.obstacle_intersects_environment([road], touching, True, roi0)

    # the planner: the region follows the ego car's update; other cars do not move it; without the setting nothing is filtered
    S = dict(obstacle_roi_filter=True, obstacle_roi_behind_distance=10.0, obstacle_roi_front_distance=100.0, obstacle_roi_side_distance=15.0)
    pl = K.MiqpPlanner(settings=S, mapPieces=[road])
    assert pl._obstacles_roi is None
    pl.AddCar([0, 5, 0, 0, 0.0, 0], [[0, 0], [300, 0]], 5, 1, 0.0, True)
    np.testing.assert_allclose(pl._obstacles_roi, roi0, atol=1e-9)
    assert pl.AddObstacle(behind_far, False, True) == -1 and pl.AddObstacle(ahead_far, False, False) == -1
    assert pl.AddObstacle(near, False, True) == 0 and pl.AddObstacle(approaching, False, False) == 1
    pl.AddCar([40, 5, 0, 0, 0.0, 0], [[0, 0], [300, 0]], 5, 1, 0.0, True)          # a second car: not the ego car
    np.testing.assert_allclose(pl._obstacles_roi, roi0, atol=1e-9)
    pl.UpdateCar(0, [100, 5, 0, 0, 0.0, 0], [[0, 0], [300, 0]], 0.0, True)         # the ego car has moved on: the far obstacle is now relevant
    assert pl.AddObstacle(ahead_far, False, True) == 2
    pl2 = K.MiqpPlanner(mapPieces=[road]); pl2.AddCar([0, 5, 0, 0, 0.0, 0], [[0, 0], [300, 0]], 5, 1, 0.0, True)
    assert pl2._obstacles_roi is None and pl2.AddObstacle(behind_far, False, True) == 0


def test_bark_trajectory_read_out_is_cut_at_the_first_invalid_velocity():
    """miqp_bark_trajectory = MiqpPlanner::GetBarkTrajectory (src/miqp_planner.cpp:1132-1170) on a hand-made result record: rows
    (time, x, y, atan2(vy, vx), |v|), cut off at the first step with |vx| <= 0.7 and |vy| <= 0.7 (IsVxVyValid :1184-1187)"""
    import ctypes as C
    from planner_miqp_amd.ctypes_types import RawResults
    r = RawResults(2, 6, 4, 0, 0, 0)
    r.pos_x[1] = [0, 1, 2, 3, 4, 5]; r.pos_y[1] = [0, .1, .2, .3, .4, .5]
    r.vel_x[1] = [4, 3, 0.5, 0.7, 0.2, 5]; r.vel_y[1] = [0, -1, 0.71, -0.7, 0.1, 5]
    out = np.zeros((6, 5)); c = r.to_c(); L = K._lib()
    K.reference_trajectory([[0, 0], [1, 0]], [0, 0, 0, 0, 1], 1.0, 2, 1.0, 1.0, 0.0, 1.0)   # (declares the prototypes)
    n = L.miqp_bark_trajectory(C.byref(c), 1, 2.0, 0.25, 0.7, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert n == 3                                                   # step 2 is valid through vy = 0.71, step 3 (0.7, -0.7) is not: strict inequality
    np.testing.assert_allclose(out[:3, 0], [2.0, 2.25, 2.5]); np.testing.assert_allclose(out[:3, 1], [0, 1, 2]); np.testing.assert_allclose(out[:3, 2], [0, .1, .2])
    np.testing.assert_allclose(out[:3, 3], np.arctan2([0, -1, 0.71], [4, 3, 0.5])); np.testing.assert_allclose(out[:3, 4], np.hypot([4, 3, 0.5], [0, -1, 0.71]))
    assert L.miqp_bark_trajectory(C.byref(c), 2, 0.0, 0.25, 0.7, out.ctypes.data_as(C.POINTER(C.c_double))) == -1    # no such car
    r.vel_x[0] = 0.1; r.vel_y[0] = 0.0
    assert L.miqp_bark_trajectory(C.byref(c), 0, 0.0, 0.25, 0.7, out.ctypes.data_as(C.POINTER(C.c_double))) == 0     # standing from the start: empty
    s = K.MiqpPlanner.CarStateToMiqpState(1.0, 2.0, 0.5, 3.0, -1.0)
    np.testing.assert_allclose(s, [[1.0, np.cos(0.5) * 3, np.cos(0.5) * -1, 2.0, np.sin(0.5) * 3, np.sin(0.5) * -1]], rtol=1e-7)
