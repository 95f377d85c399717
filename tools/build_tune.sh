#!/bin/bash
# interior point constants as compile-time variants: tools/build_tune.sh name "-DMIQP_LAM0=500" ... -> planner_miqp_amd/libmiqp_gpu_<name>.so
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -fno-math-errno -freciprocal-math -fno-signed-zeros -fno-trapping-math -fPIC -shared -std=c++17"
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc $F $2 -o planner_miqp_amd/libmiqp_gpu_$1.so planner_miqp_amd/csrc/miqp_gpu.hip 2>&1 | grep -i " error" &
  shift 2
done
wait
