#!/bin/bash
# A/B of environment knobs on a KNOBS library variant: tools/ab/ab_env.sh <tag> <variant> "<ENV=.. ENV=..>" ["<...>" ...]
tag=$1; v=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
lib=$R/pynndescent_amd/_exp/lib_$v.so
for e in "$@"; do
  echo "== $e" >> $O/${tag}_ab_env.log
  ( cd $R && env $e PYNND_AMD_LIB=$lib timeout 300 python tools/ab/ab_forest.py 1000000 8 4 2>&1 | grep route | head -1 >> $O/${tag}_ab_env.log )
  ( cd $R && env $e PYNND_AMD_LIB=$lib timeout 300 python tools/ab/ab_forest.py 10000000 2 2 2>&1 | grep route | head -1 >> $O/${tag}_ab_env.log )
done
cat $O/${tag}_ab_env.log
