#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on the forest stage: tools/ab/ab_route.sh <tag> <variant> ...   ("base" = the product library)
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for v in "$@"; do
  lib=$R/pynndescent_amd/_exp/lib_$v.so; [ "$v" == "base" ] && lib=$R/pynndescent_amd/libpynnd_amd.so
  echo "== $v" >> $O/${tag}_ab_route.log
  ( cd $R && PYNND_AMD_LIB=$lib timeout 300 python tools/ab/ab_forest.py 1000000 8 4 2>&1 | grep route >> $O/${tag}_ab_route.log )
  ( cd $R && PYNND_AMD_LIB=$lib timeout 300 python tools/ab/ab_forest.py 10000000 2 2 2>&1 | grep route >> $O/${tag}_ab_route.log )
done
cat $O/${tag}_ab_route.log
