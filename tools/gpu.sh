#!/bin/bash
# rebuilds the in-tree library when a source is newer, then runs the command on a GPU box: tools/gpu.sh <timeout> '<command>'
cd "$(dirname "$0")/.."
python -c "import planner_miqp_amd as P; P.build_library()" || exit 1
make -s -C oracle || exit 1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
