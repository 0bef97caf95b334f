#!/bin/bash
# crash hunt: repeat the multi-rank tests, native backtrace on abort (tools/dbg/abrt_bt.c)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ -f tools/dbg/abrt_bt.so ] || gcc -O1 -g -shared -fPIC -o tools/dbg/abrt_bt.so tools/dbg/abrt_bt.c
N=${1:-16}
fails=0
for i in $(seq 1 $N); do
  LD_PRELOAD="$GRAFT_REPO_ROOT/tools/dbg/abrt_bt.so" timeout 250 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_integration.py -m gpu -q -x -s -p no:cacheprovider -k "not at_scale and not rccl and not bench" > gpurun_out/heap_$i.log 2>&1
  if ! grep -q " passed" gpurun_out/heap_$i.log || grep -q "failed\|native backtrace" gpurun_out/heap_$i.log; then echo "run $i FAILED"; fails=$((fails+1)); grep -n -A40 "native backtrace" gpurun_out/heap_$i.log | head -60 | cut -c1-200; fi
done
echo "done $N runs, $fails failures"; grep -h " passed" gpurun_out/heap_1.log
