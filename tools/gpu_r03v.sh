#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -- python $R/tools/prof_one_tree_10m.py 10000000 1 2>&1 | grep "ms_forest"
python $R/tools/prof_summary.py /tmp/prof_v $R/gpurun_out/r03v_one_tree_10m_kernel_stats.txt; head -28 $R/gpurun_out/r03v_one_tree_10m_kernel_stats.txt | cut -c1-170
