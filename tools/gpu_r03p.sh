#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_build.py tests/test_gpu_sharded.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-900
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > /dev/null 2>&1
f=$(find /tmp/prof_p -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r03p_kernel_stats.csv; head -16 $f | cut -d, -f1-4 | cut -c1-150
