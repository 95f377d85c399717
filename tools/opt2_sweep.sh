#!/bin/bash
# experiment switches of eval_kernel (MIQP_OPT2) on 256-instance bench steps: ./opt2_sweep.sh STEP value...
st=$1; shift
for o in "$@"; do echo "== step $st MIQP_OPT2=$o"; MIQP_OPT2=$o BP_TOP=0 python tools/batch_profile.py $st 2>&1 | grep -v amdgpu.ids | grep "batch\|quantiles" | cut -c1-160; done
