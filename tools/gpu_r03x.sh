#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && PYTHONPATH=$R timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $R/tools/rank_critical_path.py --world 8 --n 10000000 --trees 12 --builds 1 2>&1 | grep "^{" | cut -c1-300
python $R/tools/prof_summary.py /tmp/prof_x $R/gpurun_out/r03_e_sharded8_10m_kernel_stats.txt; head -24 $R/gpurun_out/r03_e_sharded8_10m_kernel_stats.txt | cut -c1-150
