#!/bin/bash
# usage: tools/trace_run.sh <tag> -- kernel-trace summary of one tools/ab/ab_stage.py run into gpurun_out/<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -- python $GRAFT_REPO_ROOT/tools/ab/ab_stage.py 8 > /tmp/tr_$tag.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/tr_$tag $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
