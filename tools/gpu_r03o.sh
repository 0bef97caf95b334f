#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_integration.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
bash tools/collect_profiles.sh r03_b 2>&1 | tail -n 3
for w in 1 8; do timeout 400 python tools/rank_critical_path.py --world $w --n 10000000 --trees 12 > gpurun_out/r03_b_critpath_w$w.log 2>&1; grep "^{" gpurun_out/r03_b_critpath_w$w.log | cut -c1-400; done
timeout 400 python tools/bench_configs.py c3 c5 c4_one_gpu k30 k60 > gpurun_out/r03_b_other_configs.jsonl 2>&1; tail -n 4 gpurun_out/r03_b_other_configs.jsonl | cut -c1-600
