for sk in 1280 525568 263424 787712; do echo "== seq_kinds $sk"; MIQP_SEQ_KINDS=$sk BP_TOP=8 python tools/batch_profile.py 2 2>&1 | grep -v amdgpu.ids | head -9; done
