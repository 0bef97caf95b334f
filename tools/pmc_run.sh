#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counter list>" -- per-kernel PMC sums of one ab_stage run, condensed into gpurun_out/<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $1 --output-format csv -d /tmp/pmc_$tag -- python $GRAFT_REPO_ROOT/tools/ab/ab_stage.py 8 > /tmp/pmc_$tag.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pmc_$tag $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
