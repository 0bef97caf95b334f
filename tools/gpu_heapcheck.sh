#!/bin/bash
# heap-corruption hunt: repeat the sharded tests with glibc's malloc checking and a native backtrace on abort
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
N=${1:-16}
for i in $(seq 1 $N); do
  LD_PRELOAD="/lib/x86_64-linux-gnu/libc_malloc_debug.so.0 $GRAFT_REPO_ROOT/tools/dbg/abrt_bt.so" GLIBC_TUNABLES=glibc.malloc.check=3 MALLOC_PERTURB_=165 \
    timeout 250 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x -s -p no:cacheprovider -k "not at_scale and not rccl" > gpurun_out/heap_$i.log 2>&1
  if ! grep -q "5 passed" gpurun_out/heap_$i.log; then echo "run $i FAILED"; grep -n -A40 "native backtrace" gpurun_out/heap_$i.log | head -60 | cut -c1-200; break; fi
done
echo "done $i runs"
