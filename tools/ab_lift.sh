#!/bin/bash
# A/B of the bound lifting (seq_kinds bit 17 switches it off): hard single instances and one bench batch
for sk in 1280 132352; do
  echo "== MIQP_SEQ_KINDS=$sk"
  MIQP_SEQ_KINDS=$sk MIQP_STATS=1 python tools/trace_one.py cfg3 662 518 664 20 2>&1 | grep "seed\|outcomes"
  MIQP_SEQ_KINDS=$sk BP_TOP=6 python tools/batch_profile.py 2 2>&1 | grep -v amdgpu.ids | head -8
done
