"""Python face of the planner core in libmiqp_gpu.so (csrc/planner_core.hpp): region tables and bookkeeping
(common/parameter/parameter_preparer.cpp, regions.cpp), the receding-horizon warm start and the region-combination loop
of MiqpPlanner::Plan (src/miqp_planner.cpp:633-766, 787-1051).  Everything except ``plan`` runs without a GPU."""
import ctypes as C

import numpy as np

from .ctypes_types import RawResults, c_double_p, c_int_p
from .wrapper import OptimizationStatus, WarmstartType, load_library

_PROTO = False


def _lib():
    global _PROTO
    L = load_library()
    if not _PROTO:
        dp, ip, f = c_double_p, c_int_p, C.c_float
        L.miqp_fraction_parameters.restype = C.c_int; L.miqp_fraction_parameters.argtypes = [C.c_int, f, dp]
        L.miqp_fitting_polynomial_parameters.restype = C.c_int; L.miqp_fitting_polynomial_parameters.argtypes = [C.c_int, f, f, dp]
        L.miqp_mean_angles.restype = C.c_int; L.miqp_mean_angles.argtypes = [dp, C.c_int, dp]
        L.miqp_limits_per_region.restype = C.c_int; L.miqp_limits_per_region.argtypes = [dp, C.c_int, f, f, f, f, dp, dp, dp, dp]
        L.miqp_calculate_region_idx.restype = C.c_int; L.miqp_calculate_region_idx.argtypes = [dp, C.c_int, f, f, ip]
        L.miqp_reserve_neighbor_regions.restype = C.c_int; L.miqp_reserve_neighbor_regions.argtypes = [ip, C.c_int, C.c_int]
        L.miqp_calculate_possible_regions.restype = C.c_int; L.miqp_calculate_possible_regions.argtypes = [dp, C.c_int, dp, C.c_int, ip]
        from .ctypes_types import ModelParamsC, RawResultsC
        L.miqp_calculate_warmstart.restype = C.c_int
        L.miqp_calculate_warmstart.argtypes = [C.POINTER(RawResultsC), C.POINTER(RawResultsC), C.c_double, C.c_double]
        L.miqp_plan.restype = C.c_int
        L.miqp_plan.argtypes = [C.c_void_p, C.POINTER(ModelParamsC), ip, ip, C.POINTER(RawResultsC), C.c_int, C.c_double, C.POINTER(C.c_int)]
        L.miqp_initial_pose_check.restype = C.c_int; L.miqp_initial_pose_check.argtypes = [C.POINTER(ModelParamsC)]
        L.miqp_select_environment.restype = C.c_int; L.miqp_select_environment.argtypes = [dp, ip, C.c_int, dp, ip, C.c_int, ip]
        L.miqp_obstacle_intersects_environment.restype = C.c_int; L.miqp_obstacle_intersects_environment.argtypes = [dp, ip, C.c_int, dp, C.c_int, C.c_int]
        L.miqp_obstacle_intersects_environment_roi.restype = C.c_int; L.miqp_obstacle_intersects_environment_roi.argtypes = [dp, ip, C.c_int, dp, C.c_int, C.c_int, dp]
        L.miqp_bark_trajectory.restype = C.c_int; L.miqp_bark_trajectory.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, dp]
        L.miqp_obstacles_roi.restype = C.c_int; L.miqp_obstacles_roi.argtypes = [C.c_double] * 6 + [dp]
        L.miqp_environment_warmstart.restype = C.c_int
        L.miqp_environment_warmstart.argtypes = [C.POINTER(RawResultsC), C.POINTER(RawResultsC), ip, C.c_int, ip, C.c_int]
        _PROTO = True
    return L


def _d(a):
    return a.ctypes.data_as(c_double_p)


def _i(a):
    return a.ctypes.data_as(c_int_p)


class FittingPolynomialParameters:
    """common/parameter/fitting_polynomial_parameters.hpp:28-168: the fitted front-axle and curvature polynomials of one
    (nr_regions, max_velocity_fitting, min_velocity_fitting) variant as R x 3 matrices; an unknown combination raises
    ValueError("Invalid number of regions or velocity!") like the reference's std::invalid_argument"""
    _ORDER = ["POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX", "POLY_KAPPA_AX_MIN"]

    def __init__(self, nr_regions, max_velocity_fitting, min_velocity_fitting):
        self.nr_regions = int(nr_regions)
        t = np.zeros((6, self.nr_regions, 3))
        if _lib().miqp_fitting_polynomial_parameters(self.nr_regions, float(max_velocity_fitting), float(min_velocity_fitting), _d(t)) != 0:
            raise ValueError("Invalid number of regions or velocity!")
        self._t = t

    def GetPolynomialDimension(self):
        return 3


for _k, _n in enumerate(FittingPolynomialParameters._ORDER):
    setattr(FittingPolynomialParameters, "Get" + _n, (lambda k: lambda self: self._t[k].copy())(_k))


class ParameterPreparer:
    """common/parameter/parameter_preparer.hpp: fraction parameters, mean angles and the per-region acceleration / jerk
    boxes for (nrRegions, maxVelocityFitting, straight-line limits)"""

    def __init__(self, nrRegions, maxVelocityFitting, minVelocityFitting, accLonMaxLimit, accLonMinLimit, jerkLonMaxLimit,
                 accLatMinMaxLimit, jerkLatMinMaxLimit):
        self.nrRegions = int(nrRegions)
        self.limits = (accLonMaxLimit, accLonMinLimit, jerkLonMaxLimit, accLatMinMaxLimit, jerkLatMinMaxLimit)
        self.fraction_parameters = np.zeros((self.nrRegions, 4))
        _lib().miqp_fraction_parameters(self.nrRegions, float(maxVelocityFitting), _d(self.fraction_parameters))

    def GetFractionParameters(self):
        return self.fraction_parameters

    def GetMeanAngleVector(self):
        out = np.zeros(self.nrRegions)
        _lib().miqp_mean_angles(_d(self.fraction_parameters), self.nrRegions, _d(out))
        return out

    def _limits(self, lo, hi, lat):
        o = [np.zeros(self.nrRegions) for _ in range(4)]
        _lib().miqp_limits_per_region(_d(self.fraction_parameters), self.nrRegions, float(lo), float(hi), float(-lat), float(lat), *[_d(a) for a in o])
        return dict(min_x=o[0], max_x=o[1], min_y=o[2], max_y=o[3])

    def CalculateAccLimitsPerCar(self):
        amax, amin, _, alat, _ = self.limits
        return self._limits(amin, amax, alat)

    def CalculateJerkLimitsPerCar(self):
        _, _, jmax, _, jlat = self.limits
        return self._limits(-jmax, jmax, jlat)


def calculate_region_idx(fraction_parameters, vx, vy):
    F = np.ascontiguousarray(fraction_parameters, dtype=np.float64)
    out = np.zeros(F.shape[0], dtype=np.int32)
    n = _lib().miqp_calculate_region_idx(_d(F), F.shape[0], float(vx), float(vy), _i(out))
    return out[:n].tolist()


def reserve_neighbor_regions(regions, row, expansions):
    """in place on ``regions[row]`` (int32 matrix), as ReserveNeighborRegions"""
    r = np.ascontiguousarray(regions[row], dtype=np.int32)
    ok = _lib().miqp_reserve_neighbor_regions(_i(r), r.shape[0], int(expansions))
    regions[row] = r
    return bool(ok)


def calculate_possible_regions(fraction_parameters, theta_ref):
    F = np.ascontiguousarray(fraction_parameters, dtype=np.float64)
    th = np.ascontiguousarray(theta_ref, dtype=np.float64)
    flags = np.zeros(F.shape[0], dtype=np.int32)
    _lib().miqp_calculate_possible_regions(_d(F), F.shape[0], _d(th), th.shape[0], _i(flags))
    return set(np.nonzero(flags)[0].tolist())


def calculate_warmstart(last: RawResults, ts, minimum_region_change_speed, out: RawResults = None):
    """MiqpPlanner::CalculateWarmstart: ``last`` shifted by one step into ``out`` (a fresh record by default)"""
    if out is None:
        out = RawResults(*last.dims)
        out.slackvarsObstacle[...] = 0; out.slackvarsObstacle_front[...] = 0
    a, b = last.to_c(), out.to_c()
    rc = _lib().miqp_calculate_warmstart(C.byref(a), C.byref(b), float(ts), float(minimum_region_change_speed))
    if rc != 0:
        raise ValueError("sizes of the two records differ")
    return out


def _flat_polys(polys):
    """list of [n, 2] vertex arrays -> (x0, y0, ... of all, vertex offsets)"""
    off = np.zeros(len(polys) + 1, dtype=np.int32)
    for k, q in enumerate(polys):
        off[k + 1] = off[k] + len(q)
    xy = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1, 2) for q in polys]).reshape(-1)) if polys else np.zeros(0)
    return xy, off


def initial_pose_check(params):
    """the check of MiqpPlanner::Plan (src/miqp_planner.cpp:654-685): None when the rear and the front axle point of every car lie
    within a piece of the environment, else (car, "rear" | "front")"""
    s, keep = params.to_c()
    r = _lib().miqp_initial_pose_check(C.byref(s))
    del keep
    if r == -2:
        raise ValueError("invalid parameters")
    return None if r < 0 else (r >> 1, "front" if (r & 1) else "rear")


def select_environment(pieces, trajectories):
    """MiqpPlanner::ResetEnvironment's choice (src/miqp_planner.cpp:490-505): indices of the convex pieces that one of the reference
    trajectories (arrays of x, y points) touches"""
    if not pieces:
        return []
    pxy, poff = _flat_polys(pieces); txy, toff = _flat_polys(trajectories)
    sel = np.zeros(len(pieces), dtype=np.int32)
    _lib().miqp_select_environment(_d(pxy), _i(poff), len(pieces), _d(txy), _i(toff), len(trajectories), _i(sel))
    return np.nonzero(sel)[0].tolist()


def obstacles_roi(x, y, theta, behind_distance, front_distance, side_distance):
    """MiqpPlanner::UpdateObstaclesROI (src/miqp_planner.cpp:1308-1335): the four vertices of the region of interest around the ego car"""
    roi = np.zeros((4, 2))
    if _lib().miqp_obstacles_roi(float(x), float(y), float(theta), float(behind_distance), float(front_distance), float(side_distance), _d(roi)) != 0:
        raise ValueError("invalid region of interest")
    return roi


def obstacle_intersects_environment(pieces, dynamic_obstacle, is_static, roi=None):
    """MiqpPlanner::ObstacleIntersectsEnvironment (src/miqp_planner.cpp:1248-1306); roi: the region of interest (4 vertices, see
    obstacles_roi) of the reference's obstacle_roi_filter, None = the empty polygon the planner starts with"""
    pxy, poff = _flat_polys(pieces)
    ob = np.ascontiguousarray(np.asarray(dynamic_obstacle, dtype=np.float64).reshape(-1, 4, 2))
    if roi is None:
        return _lib().miqp_obstacle_intersects_environment(_d(pxy), _i(poff), len(pieces), _d(ob), ob.shape[0], int(bool(is_static))) == 1
    r = np.ascontiguousarray(np.asarray(roi, dtype=np.float64).reshape(4, 2))
    return _lib().miqp_obstacle_intersects_environment_roi(_d(pxy), _i(poff), len(pieces), _d(ob), ob.shape[0], int(bool(is_static)), _d(r)) == 1


def environment_warmstart(last: RawResults, ids_old, ids_new):
    """MiqpPlanner::EnvironmentWarmstart (src/miqp_planner.cpp:1053-1115): a copy of ``last`` whose five environment arrays follow
    the new list of piece ids (kept pieces keep their columns for the steps 0 .. N-2, everything else is 1)"""
    Cn, N, R, E, O, Lo = last.dims
    out = RawResults(Cn, N, R, len(ids_new), O, Lo)
    for nm in out.__dict__:
        a, b = getattr(out, nm), getattr(last, nm, None)
        if isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.shape == b.shape:
            a[...] = b
    io, in_ = np.ascontiguousarray(ids_old, dtype=np.int32), np.ascontiguousarray(ids_new, dtype=np.int32)
    a, b = last.to_c(), out.to_c()
    if _lib().miqp_environment_warmstart(C.byref(a), C.byref(b), _i(io), len(io), _i(in_), len(in_)) != 0:
        raise ValueError("records and id lists do not fit")
    return out


def plan(wrapper, params, warmstart: RawResults = None, warmstart_type=WarmstartType.NO_WARMSTART, timestamp=0.0):
    """MiqpPlanner::Plan's region-combination loop on the solver of ``wrapper``; ``params.initial_region`` and
    ``params.possible_region`` are updated like the reference updates its ModelParameters.  Returns (ok, status)."""
    L = _lib()
    s, keep = params.to_c()
    Cn, R = params.NumCars, params.nr_regions
    ir = np.ascontiguousarray(np.asarray(params.initial_region, dtype=np.int32).reshape(Cn))
    pr = np.ascontiguousarray(np.asarray(params.possible_region, dtype=np.int32).reshape(Cn, R))
    st = C.c_int(int(OptimizationStatus.FAILED_NO_SOLUT))
    wc = warmstart.to_c() if warmstart is not None else None
    ok = L.miqp_plan(wrapper._h, C.byref(s), _i(ir), _i(pr), C.byref(wc) if wc is not None else None, int(warmstart_type), float(timestamp), C.byref(st))
    params.initial_region = ir.copy(); params.possible_region = pr.copy()
    wrapper._params = params
    status = wrapper._collect(st.value)
    del keep
    return bool(ok), status


def reference_trajectory(ref_xy, state, dt, num_points, line_interp_inc, vel_desired, delta_s_desired, acc_lat_max, vel_curve_dep=False):
    """ReferenceTrajectoryGenerator::GenerateTrajectory on a polyline (common/reference/reference_trajectory_generator.cpp:
    51-148); ``state`` = (time, x, y, theta, v); returns [num_points, 5] in the same order.  bark's spline smoothing of the
    centre line is replaced by the polyline itself: identical on straight reference lines."""
    L = _lib()
    if not hasattr(L, "_ref_proto"):
        L.miqp_reference_trajectory.restype = C.c_int
        L.miqp_reference_trajectory.argtypes = [c_double_p, C.c_int, c_double_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, c_double_p]
        L.miqp_update_car.restype = C.c_int
        L.miqp_update_car.argtypes = [c_double_p, c_double_p, c_double_p, c_double_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, c_double_p, c_int_p, c_double_p]
        L._ref_proto = True
    xy = np.ascontiguousarray(np.asarray(ref_xy, dtype=np.float64).reshape(-1, 2))
    st = np.ascontiguousarray(np.asarray(state, dtype=np.float64).reshape(5))
    out = np.zeros((int(num_points), 5))
    rc = L.miqp_reference_trajectory(_d(xy), xy.shape[0], _d(st), float(dt), int(num_points), float(line_interp_inc), float(vel_desired),
                                     float(delta_s_desired), float(acc_lat_max), int(bool(vel_curve_dep)), _d(out))
    if rc != 0:
        raise ValueError("invalid reference line or arguments (%d)" % rc)
    return out


def DefaultSettings():
    """miqp::planner::DefaultSettings() (src/miqp_planner_data.hpp:190-242), the fields this mirror reads"""
    return dict(nr_regions=16, nr_steps=20, nr_neighbouring_possible_regions=1, ts=0.25, max_solution_time=10.0, relative_mip_gap_tolerance=0.1,
                mipdisplay=2, mipemphasis=0, relobjdif=0.0, cutpass=0, probe=0, repairtries=0, rinsheur=0, varsel=0, mircuts=0, precision=12,
                constant_agent_safety_distance_slack=3.0, minimum_region_change_speed=2.0, lambda_=0.5, wheelBase=2.8, collisionRadius=1.0,
                slackWeight=30.0, slackWeightObstacle=2000.0, jerkWeight=1.0, positionWeight=2.0, velocityWeight=0.0, acclerationWeight=0.0,
                accLonMaxLimit=2.0, accLonMinLimit=-4.0, jerkLonMaxLimit=3.0, accLatMinMaxLimit=1.6, jerkLatMinMaxLimit=1.4, refLineInterpInc=0.2,
                additionalStepsForReferenceLongerHorizon=4, max_velocity_fitting=20.0, parallelMode=1, warmstartType=WarmstartType.NO_WARMSTART,
                obstacle_roi_filter=False, obstacle_roi_behind_distance=5.0, obstacle_roi_front_distance=30.0, obstacle_roi_side_distance=15.0)


class MiqpPlanner:
    """Bark-free mirror of the part of miqp::planner::MiqpPlanner that feeds the solve path (src/miqp_planner.cpp: constructor
    :27-147, AddCar :180-281, UpdateCar :284-390, Plan :633-766, and the raw trajectory read-out of the C API,
    src/miqp_planner_c_api.cpp:101-136) for a planner WITHOUT map and obstacles (empty environment polygon, as the reference's
    C-API test uses it).  Reference lines are polylines (see reference_trajectory)."""
    EPS = 0.000001   # MiqpPlanner::eps_ (src/miqp_planner.hpp:415)

    def __init__(self, settings=None, mapPieces=None, **wrapper_args):
        from .ctypes_types import ModelParameters
        from .wrapper import CplexWrapper
        S = dict(DefaultSettings()); S.update(settings or {})
        self.settings = S
        self.pp = ParameterPreparer(S["nr_regions"], S["max_velocity_fitting"], S["minimum_region_change_speed"], S["accLonMaxLimit"], S["accLonMinLimit"],
                                    S["jerkLonMaxLimit"], S["accLatMinMaxLimit"], S["jerkLatMinMaxLimit"])
        fit = FittingPolynomialParameters(S["nr_regions"], S["max_velocity_fitting"], S["minimum_region_change_speed"])
        p = ModelParameters()
        R, N = S["nr_regions"], S["nr_steps"]
        p.nr_regions, p.NumSteps, p.NumCars, p.nr_obstacles, p.nr_environments, p.max_lines_obstacles = R, N, 0, 0, 0, 0
        for k in FittingPolynomialParameters._ORDER:
            setattr(p, k, getattr(fit, "Get" + k)())
        for k in ("max_solution_time", "relative_mip_gap_tolerance", "mipdisplay", "mipemphasis", "relobjdif", "cutpass", "probe", "repairtries", "rinsheur",
                  "varsel", "mircuts", "ts", "minimum_region_change_speed"):
            setattr(p, k, S[k])
        p.parallelmode = int(S["parallelMode"])
        p.agent_safety_distance = np.zeros(N); p.agent_safety_distance_slack = np.full(N, float(S["constant_agent_safety_distance_slack"]))
        p.maximum_slack = S["constant_agent_safety_distance_slack"]; p.WEIGHTS_SLACK = S["slackWeight"]; p.WEIGHTS_SLACK_OBSTACLE = S["slackWeightObstacle"]
        p.min_vel_x_y = -S["max_velocity_fitting"] - self.EPS; p.max_vel_x_y = S["max_velocity_fitting"] + self.EPS
        p.fraction_parameters = self.pp.GetFractionParameters()
        for k in ModelParameters.VEC_C:
            setattr(p, k, np.zeros(0))
        p.IntitialState = np.zeros((0, 6))
        for k in ModelParameters.MAT_CN:
            setattr(p, k, np.zeros((0, N)))
        for k in ModelParameters.MAT_CR:
            setattr(p, k, np.zeros((0, R)))
        p.initial_region = np.zeros(0, dtype=np.int32); p.possible_region = np.zeros((0, R), dtype=np.int32)
        self.parameters = p
        self.egoCarIdx = 0
        self._obstacles_roi = None   # MiqpPlanner::obstacles_roi_: empty until the ego car is updated with obstacle_roi_filter set
        self._refs = []          # per car: (reference line, desired velocity, delta s)
        self.wrapper = CplexWrapper("cplexmodel.mod", precision=S["precision"], **wrapper_args)
        self.status = None
        self._ws = None          # start for the next plan (receding horizon)
        self._car_ref = {}       # per car: what its reference trajectories are generated from
        self._map = []           # convex pieces of the map (ids = their index), see UpdateConvexifiedMap
        self._env_ids = []       # ids of the pieces in the parameters / in the warm start
        self._ws_env_ids = []
        if mapPieces:
            self.UpdateConvexifiedMap(mapPieces)

    def GetN(self):
        return self.settings["nr_steps"]

    def GetTs(self):
        return self.settings["ts"]

    def GetParameters(self):
        return self.parameters

    def AddCar(self, initialState, referencePath, desiredVelocity, deltaSForDesiredVel, timestep=0.0, track_reference_positions=True):
        p, S = self.parameters, self.settings
        idx = p.NumCars; p.NumCars = idx + 1
        R, N = p.nr_regions, p.NumSteps

        def grow(a, shape):
            b = np.zeros(shape, dtype=np.asarray(a).dtype); b[tuple(slice(0, n) for n in np.asarray(a).shape)] = a
            return b
        p.CollisionRadius = grow(p.CollisionRadius, (idx + 1,)); p.CollisionRadius[idx] = S["collisionRadius"]
        p.WheelBase = grow(p.WheelBase, (idx + 1,)); p.WheelBase[idx] = S["wheelBase"]
        p.IntitialState = grow(p.IntitialState, (idx + 1, 6))
        acc, jerk = self.pp.CalculateAccLimitsPerCar(), self.pp.CalculateJerkLimitsPerCar()
        for nm, d in (("acc", acc), ("jerk", jerk)):
            for k in ("min_x", "max_x", "min_y", "max_y"):
                key = "%s_%s_%s" % (k[:3], nm, k[-1])
                m = grow(getattr(p, key), (idx + 1, R)); m[idx] = d[k]; setattr(p, key, m)
        p.total_max_acc = max(p.max_acc_x.max(), p.max_acc_y.max()) + self.EPS; p.total_min_acc = min(p.min_acc_x.min(), p.min_acc_y.min()) - self.EPS
        p.total_max_jerk = max(p.max_jerk_x.max(), p.max_jerk_y.max()) + self.EPS; p.total_min_jerk = min(p.min_jerk_x.min(), p.min_jerk_y.min()) - self.EPS
        for k in ModelParameters_MAT_CN:
            setattr(p, k, grow(getattr(p, k), (idx + 1, N)))
        p.possible_region = grow(p.possible_region, (idx + 1, R)); p.initial_region = grow(p.initial_region, (idx + 1,))
        for k in ("WEIGHTS_POS_X", "WEIGHTS_VEL_X", "WEIGHTS_ACC_X", "WEIGHTS_POS_Y", "WEIGHTS_VEL_Y", "WEIGHTS_ACC_Y", "WEIGHTS_JERK_X", "WEIGHTS_JERK_Y"):
            setattr(p, k, grow(getattr(p, k), (idx + 1,)))
        self._refs.append((desiredVelocity, deltaSForDesiredVel))
        self.UpdateCar(idx, initialState, referencePath, timestep, track_reference_positions)
        return idx

    def UpdateCar(self, idx, initialState, referencePath, timestep=0.0, track_reference_positions=True):
        p, S = self.parameters, self.settings
        reference_trajectory([[0, 0], [1, 0]], [0, 0, 0, 0, 1], 1.0, 2, 1.0, 1.0, 0.0, 1.0)   # (declares the prototypes)
        L = _lib()
        st = np.ascontiguousarray(np.asarray(initialState, dtype=np.float64).reshape(6))
        p.IntitialState[idx] = st
        xy = np.ascontiguousarray(np.asarray(referencePath, dtype=np.float64).reshape(-1, 2))
        self._car_ref[idx] = (xy.copy(), st.copy(), float(timestep))
        s12 = np.array([S["nr_regions"], S["nr_steps"], S["nr_neighbouring_possible_regions"], S["additionalStepsForReferenceLongerHorizon"], S["ts"],
                        S["refLineInterpInc"], S["accLatMinMaxLimit"], S["lambda_"], S["positionWeight"], S["velocityWeight"], S["acclerationWeight"], S["jerkWeight"]], dtype=np.float64)
        F = np.ascontiguousarray(p.fraction_parameters, dtype=np.float64)
        ref = np.zeros((4, p.NumSteps)); poss = np.zeros(p.nr_regions, dtype=np.int32); w8 = np.zeros(8)
        vdes, ds = self._refs[idx]
        rc = L.miqp_update_car(_d(s12), _d(F), _d(st), _d(xy), xy.shape[0], float(vdes), float(ds), float(timestep), int(bool(track_reference_positions)),
                               int(idx == self.egoCarIdx), int(p.NumCars), _d(ref), _i(poss), _d(w8))
        if rc < 0:
            raise ValueError("invalid car update (%d)" % rc)
        p.x_ref[idx], p.y_ref[idx], p.vx_ref[idx], p.vy_ref[idx] = ref
        p.possible_region[idx] = poss
        for k, nm in enumerate(("WEIGHTS_POS_X", "WEIGHTS_VEL_X", "WEIGHTS_ACC_X", "WEIGHTS_POS_Y", "WEIGHTS_VEL_Y", "WEIGHTS_ACC_Y", "WEIGHTS_JERK_X", "WEIGHTS_JERK_Y")):
            getattr(p, nm)[idx] = w8[k]
        if S["obstacle_roi_filter"] and idx == self.egoCarIdx:   # far-away obstacles are filtered out around the ego car (src/miqp_planner.cpp:380-387)
            self._obstacles_roi = obstacles_roi(st[0], st[3], np.arctan2(st[4], st[1]), S["obstacle_roi_behind_distance"], S["obstacle_roi_front_distance"], S["obstacle_roi_side_distance"])
        return rc == 0

    def Plan(self, timestamp=0.0):
        """MiqpPlanner::Plan (src/miqp_planner.cpp:633-785): the region-combination loop with the start derived from the last
        solution when the settings ask for a warm start; a successful plan prepares the start of the next one (:768-779)"""
        wt = self.settings["warmstartType"]
        if self._map:   # convexifiedMap_.HasValidPolygon(): the environment follows the reference trajectories (:648-652)
            self.ResetEnvironment(self.CalculateReferenceTrajectoriesLongerHorizon())
        ws = self._ws if (wt != WarmstartType.NO_WARMSTART and self._ws is not None and self._ws.dims == self._dims()) else None
        ok, self.status = plan(self.wrapper, self.parameters, ws, wt if ws is not None else WarmstartType.NO_WARMSTART, timestamp)
        self._ws = None
        if ok and wt != WarmstartType.NO_WARMSTART:
            self._ws = calculate_warmstart(self.GetSolution(), self.parameters.ts, self.parameters.minimum_region_change_speed)
            self._ws_env_ids = list(self._env_ids)
        return ok

    # ---- environment (src/miqp_planner.cpp:490-537, 1053-1115, 1207-1221) on convex pieces
    def UpdateConvexifiedMap(self, pieces):
        """the reference convexifies a bark map polygon here (ConvexifiedMap::Convert, out of scope); this mirror takes the convex
        counter-clockwise pieces themselves - e.g. the map polygon itself when it is convex, as in the reference's planner tests"""
        self._map = [np.asarray(q, dtype=np.float64).reshape(-1, 2).copy() for q in pieces]
        return all(len(q) >= 3 for q in self._map)

    def CalculateReferenceTrajectoriesLongerHorizon(self):
        S = self.settings; out = []
        for idx in range(self.parameters.NumCars):
            xy, st, t0 = self._car_ref[idx]; vdes, ds = self._refs[idx]
            st5 = [t0, st[0], st[3], np.arctan2(st[4], st[1]), np.hypot(st[1], st[4])]
            tr = reference_trajectory(xy, st5, S["ts"], S["nr_steps"] + S["additionalStepsForReferenceLongerHorizon"], S["refLineInterpInc"], vdes, ds, S["accLatMinMaxLimit"])
            out.append(tr[:, 1:3].copy())
        return out

    def ResetEnvironment(self, referenceTrajectories):
        p = self.parameters
        sel = select_environment(self._map, referenceTrajectories)
        p.MultiEnvironmentConvexPolygon = [self._map[k] for k in sel]
        p.nr_environments = len(sel)
        self._env_ids = list(sel)
        if self._ws is not None and self._ws_env_ids != self._env_ids:   # EnvironmentWarmstart
            self._ws = environment_warmstart(self._ws, self._ws_env_ids, self._env_ids)
            self._ws_env_ids = list(self._env_ids)

    # ---- obstacles (src/miqp_planner.cpp:392-488, 617-629)
    def CreateMiqpObstacle(self, predicted_traj, shape_xy):
        """CreateMiqpObstacle for a rectangular shape (vertices in the obstacle's frame): placed at every (x, y, theta) of the predicted
        trajectory (rows time, x, y, theta, v) and inflated by the collision radius - bark's BufferPolygon + simplification to four
        edges, restated as the outward offset of the four edges (exact for a rectangle with mitred corners); counter-clockwise"""
        sh = np.asarray(shape_xy, dtype=np.float64).reshape(-1, 2)[:4]
        c = sh.mean(0); r = self.settings["collisionRadius"]
        d = sh - c
        infl = c + d + r * np.sign(d)          # axis-aligned rectangle in its own frame: every side moves out by r
        area = 0.5 * np.sum(infl[:, 0] * np.roll(infl[:, 1], -1) - np.roll(infl[:, 0], -1) * infl[:, 1])
        if area < 0:
            infl = infl[::-1]
        out = []
        for row in np.asarray(predicted_traj, dtype=np.float64).reshape(-1, 5):
            cs, sn = np.cos(row[3]), np.sin(row[3])
            out.append(np.stack([row[1] + cs * infl[:, 0] - sn * infl[:, 1], row[2] + sn * infl[:, 0] + cs * infl[:, 1]], 1))
        return out

    def AddObstacle(self, dynamic_obstacle, is_soft=False, is_static=False):
        """dynamic_obstacle: N arrays of 4 counter-clockwise vertices; returns the obstacle id, -1 when it does not touch the environment"""
        p = self.parameters
        ob = [np.asarray(q, dtype=np.float64).reshape(4, 2).copy() for q in dynamic_obstacle]
        assert len(ob) == p.NumSteps
        env = list(p.MultiEnvironmentConvexPolygon) if p.nr_environments > 0 else list(self._map)
        if not obstacle_intersects_environment(env, ob, is_static, getattr(self, "_obstacles_roi", None)):
            return -1
        p.ObstacleConvexPolygon = list(p.ObstacleConvexPolygon) + [ob]
        p.obstacle_is_soft = list(p.obstacle_is_soft) + [int(bool(is_soft))]
        p.nr_obstacles = len(p.ObstacleConvexPolygon); p.max_lines_obstacles = 4
        self._ws = None          # the sizes have changed: the previous warm start cannot be used
        return p.nr_obstacles - 1

    def AddStaticObstacle(self, shape_xy, pose=(0.0, 0.0, 0.0)):
        tr = np.tile(np.array([0.0, pose[0], pose[1], pose[2], 0.0]), (self.parameters.NumSteps, 1))
        return self.AddObstacle(self.CreateMiqpObstacle(tr, shape_xy), False, True)

    def UpdateObstacle(self, id, dynamic_obstacle):
        ob = [np.asarray(q, dtype=np.float64).reshape(4, 2).copy() for q in dynamic_obstacle]
        self.parameters.ObstacleConvexPolygon[id] = ob

    def RemoveObstacle(self, id):
        raise NotImplementedError("RemoveObstacle")   # NotImplementedException (src/miqp_planner.cpp:617)

    def RemoveAllObstacles(self):
        p = self.parameters
        p.ObstacleConvexPolygon = []; p.nr_obstacles = 0; p.obstacle_is_soft = []; p.max_lines_obstacles = 0
        self._ws = None

    def _dims(self):
        p = self.parameters
        return (p.NumCars, p.NumSteps, p.nr_regions, p.nr_environments, p.nr_obstacles, p.max_lines_obstacles)

    def GetSolution(self):
        return self.wrapper.getRawResults()

    def GetRawCMiqpTrajectory(self, carIdx, start_time=0.0):
        """rows (time, x, y, vx, vy, ax, ay, ux, uy) as GetRawCMiqpTrajectoryCMiqpPlanner (src/miqp_planner_c_api.cpp:101-136)"""
        r = self.GetSolution()
        N = r.N
        out = np.zeros((N, 9))
        out[:, 0] = start_time + self.parameters.ts * np.arange(N)
        for k, nm in enumerate(("pos_x", "pos_y", "vel_x", "vel_y", "acc_x", "acc_y", "u_x", "u_y")):
            out[:, 1 + k] = getattr(r, nm)[carIdx]
        return out


    MINIMUM_VALID_SPEED_VX_VY = 0.7   # MiqpPlanner::minimum_valid_speed_vx_vy_ (src/miqp_planner.cpp:53-54)

    def GetBarkTrajectory(self, carIdx, start_time=0.0):
        """MiqpPlanner::GetBarkTrajectory (src/miqp_planner.cpp:1132-1170) as a plain array: rows (time, x, y, theta, v) in bark's
        StateDefinition order, cut off at the first step whose velocity components are both below the valid speed"""
        r = self.GetSolution()
        if not 0 <= carIdx < r.NrCars:
            raise IndexError("car %d of %d" % (carIdx, r.NrCars))
        out = np.zeros((r.N, 5)); c = r.to_c()
        n = _lib().miqp_bark_trajectory(C.byref(c), int(carIdx), float(start_time), float(self.parameters.ts), self.MINIMUM_VALID_SPEED_VX_VY, _d(out))
        if n < 0:
            raise ValueError("invalid trajectory request")
        return out[:n].copy()

    def Get2ndOrderStateFromSolution(self, timeIdx, carIdx):
        """(x, vx, ax, y, vy, ay) of a car at a step of the last solution (src/miqp_planner.cpp:1117-1130)"""
        r = self.GetSolution()
        return np.array([[getattr(r, nm)[carIdx, timeIdx] for nm in ("pos_x", "vel_x", "acc_x", "pos_y", "vel_y", "acc_y")]])

    @staticmethod
    def CarStateToMiqpState(x, y, theta, v, a):
        """(x, vx, ax, y, vy, ay) of a car state given as pose, speed and acceleration along the heading (src/miqp_planner.cpp:1189-1200;
        the reference takes its arguments as float: they are rounded to single precision first)"""
        x, y, theta, v, a = (float(np.float32(q)) for q in (x, y, theta, v, a))
        return np.array([[x, np.cos(theta) * v, np.cos(theta) * a, y, np.sin(theta) * v, np.sin(theta) * a]])


ModelParameters_MAT_CN = ["x_ref", "vx_ref", "y_ref", "vy_ref"]
