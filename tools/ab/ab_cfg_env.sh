#!/bin/bash
# recall / time of tools/bench_configs.py configurations under environment knobs (KNOBS variant of the library):
#   tools/ab/ab_cfg_env.sh <tag> <variant> "<configs>" "<ENV=..>" ["<ENV=..>" ...]
tag=$1; v=$2; cfgs=$3; shift; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
lib=$R/pynndescent_amd/_exp/lib_$v.so
for e in "$@"; do
  echo "== $e" >> $O/${tag}_ab_cfg.log
  ( cd $R && env $e PYNND_AMD_LIB=$lib timeout 400 python tools/bench_configs.py $cfgs 2>&1 | grep '^{' | cut -c1-420 >> $O/${tag}_ab_cfg.log )
done
cat $O/${tag}_ab_cfg.log
