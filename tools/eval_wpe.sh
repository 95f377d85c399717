#!/bin/bash
# eval_kernel register-allocated for 4 wavefronts per SIMD (128 VGPRs) against the compiler's choice (141 VGPRs, 3 per SIMD): average duration of the kernel
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in default ev4; do
  if [ $v != default ]; then export MIQP_GPU_LIB=$R/tools/_build/libmiqp_gpu_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/evw_$v -o e -- python3 $R/tools/stream_check.py 2048 1024 0 10 > /dev/null 2>&1
  echo "== $v"; find $R/gpurun_out/evw_$v -name "*kernel_stats.csv" -exec grep "eval_kernel\|select_kernel" {} \; | cut -c1-120
  find $R/gpurun_out/evw_$v -name "*kernel_trace.csv" -delete; find $R/gpurun_out/evw_$v -name "*.db" -delete
done
