#!/bin/bash
# profiling / ablation variants of the library, kept out of the package (tools/_build/, selected with MIQP_GPU_LIB): libmiqp_gpu_prof.so (per-phase clock64 counters,
# printed per solve) and libmiqp_gpu_abl.so (MIQP_REPLAY replays with parts of the on-chip kernel removed, see miqp_gpu.hip)
cd "$(dirname "$0")/.."
mkdir -p tools/_build
F="--offload-arch=gfx950 -O3 -fno-math-errno -fno-trapping-math -Xarch_device -freciprocal-math -Xarch_device -fno-signed-zeros -fPIC -shared -std=c++17"
/opt/rocm/bin/hipcc $F -DMIQP_PROFILE -o tools/_build/libmiqp_gpu_prof.so planner_miqp_amd/csrc/miqp_gpu.hip &
/opt/rocm/bin/hipcc $F -DMIQP_ABLATE -o tools/_build/libmiqp_gpu_abl.so planner_miqp_amd/csrc/miqp_gpu.hip &
wait
