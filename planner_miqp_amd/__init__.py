"""planner_miqp_amd - MI355X-native MIQP solve path behind planner-miqp's CplexWrapper boundary.

The product is libmiqp_gpu.so (hand-written HIP for gfx950, C ABI in include/miqp_gpu.h); this package is the
host-side mirror of the reference's operator interface for that path (CplexWrapper, ModelParameters,
RawResults, SolutionProperties) over that C ABI.  There is no CPU fallback: importing works anywhere,
solving requires the built library and a HIP device.
"""
from .ctypes_types import ModelParameters, RawResults  # noqa: F401
from .wrapper import (CplexWrapper, OptimizationStatus, SolutionProperties, WarmstartType, ParameterSource,  # noqa: F401
                      solve_batch, prepare_batch, materialize_results, load_library, library_path, build_library)
