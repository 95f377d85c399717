#!/bin/bash
# profiling / ablation variants of the library next to the product build: libmiqp_gpu_prof.so (per-phase clock64 counters,
# printed per solve) and libmiqp_gpu_abl.so (MIQP_REPLAY replays with parts of the on-chip kernel removed, see miqp_gpu.hip)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -fno-math-errno -freciprocal-math -fno-signed-zeros -fno-trapping-math -fPIC -shared -std=c++17"
/opt/rocm/bin/hipcc $F -DMIQP_PROFILE -o planner_miqp_amd/libmiqp_gpu_prof.so planner_miqp_amd/csrc/miqp_gpu.hip &
/opt/rocm/bin/hipcc $F -DMIQP_ABLATE -o planner_miqp_amd/libmiqp_gpu_abl.so planner_miqp_amd/csrc/miqp_gpu.hip &
wait
