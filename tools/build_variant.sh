#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> "<extra flags>"  -> pynndescent_amd/_exp/lib_<name>.so (A/B experiments)
# the variant object is compiled with -DNND_EXPERIMENT_KNOBS as well (the NND_* environment knobs of state.h nnd_knob)
set -e
cd "$(dirname "$0")/../pynndescent_amd/csrc"
name=$1; file=$2; flags=$3
mkdir -p ../_exp _obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DNND_EXPERIMENT_KNOBS $flags -c $file -o _obj/${file%.hip}_$name.o
objs=""
for f in prep rpforest leaf_join join sample merge finalize prune searchgraph hubtree query capi comm shard; do
  if [ "$f.hip" == "$file" ]; then objs="$objs _obj/${f}_$name.o"; else objs="$objs _obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_exp/lib_$name.so $objs -ldl -lpthread
echo built ../_exp/lib_$name.so
