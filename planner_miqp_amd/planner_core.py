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
