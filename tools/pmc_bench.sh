#!/bin/bash
# usage: tools/pmc_bench.sh <tag> "<counters>" -- per-kernel PMC sums of one bench build into gpurun_out/<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_$tag
timeout 400 rocprofv3 --pmc $1 --output-format csv -d /tmp/pb_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pb_$tag.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pb_$tag $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
