#!/bin/bash
# tuning / profiling / ablation variants of the library, kept out of the package (tools/_build/, selected with MIQP_GPU_LIB): libmiqp_gpu_prof.so (per-phase clock64 counters,
# printed per solve) and libmiqp_gpu_abl.so (MIQP_REPLAY replays with parts of the on-chip kernel removed, see miqp_gpu.hip)
cd "$(dirname "$0")/.."
mkdir -p tools/_build
F="--offload-arch=gfx950 -O3 -fno-math-errno -fno-trapping-math -Xarch_device -freciprocal-math -Xarch_device -fno-signed-zeros -fPIC -shared -std=c++17"
# (every variant is a tuning build: all MIQP_* experiment switches of host_inst.hpp's KNOB_T are live; the product library reads only the KNOB_P list of INTEGRATION.md)
/opt/rocm/bin/hipcc $F -DMIQP_TUNING=1 -DMIQP_PROFILE -o tools/_build/libmiqp_gpu_prof.so planner_miqp_amd/csrc/miqp_gpu.hip &
/opt/rocm/bin/hipcc $F -DMIQP_TUNING=1 -DMIQP_ABLATE -o tools/_build/libmiqp_gpu_abl.so planner_miqp_amd/csrc/miqp_gpu.hip &
/opt/rocm/bin/hipcc $F -DMIQP_TUNING=1 -o tools/_build/libmiqp_gpu_tune.so planner_miqp_amd/csrc/miqp_gpu.hip &
wait
