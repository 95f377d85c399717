#!/bin/bash
# branching-order sweep on one 256-instance bench step (seeds of step $1): solved count, nodes, finish-time quantiles
for m in "$@"; do sk=$((m << 8)); echo "== prio mode $m (MIQP_SEQ_KINDS=$sk)"; MIQP_SEQ_KINDS=$sk BP_TOP=0 python tools/batch_profile.py 2 2>&1 | grep -v amdgpu.ids | grep "batch\|quantiles" | cut -c1-200; done
