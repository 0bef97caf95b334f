#!/bin/bash
# sample stride / cell size of the routing forest, sharded critical path + per-tree cost on one GPU:
#   tools/ab/ab_stride.sh <tag> <variant> "<ENV=.. ENV=..>" ["<...>" ...]     (variant: a KNOBS build of capi.hip)
tag=$1; v=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
lib=$R/pynndescent_amd/_exp/lib_$v.so
log=$O/${tag}_ab_stride.log
for e in "$@"; do
  echo "== $e" >> $log
  ( cd $R && env $e PYNND_AMD_LIB=$lib timeout 400 python tools/rank_critical_path.py --world 8 --n 10000000 --trees 12 ${ONE_GPU:+--one-gpu} 2>&1 | grep '^{' >> $log )
  ( cd $R && env $e PYNND_AMD_LIB=$lib timeout 300 python tools/ab/ab_forest.py 10000000 2 2 2>&1 | grep route | head -1 >> $log )
  ( cd $R && env $e PYNND_AMD_LIB=$lib timeout 300 python tools/ab/ab_forest.py 1000000 8 4 2>&1 | grep route | head -1 >> $log )
done
python - <<PY
import json
for line in open("$log"):
    if line.startswith("=="): print(line.strip())
    elif line.startswith("{"):
        r = json.loads(line)
        print("  crit %.1f compute %.1f exch %.1f ag_exposed %.1f recall %.4f one_gpu %s rank0 %s" % (r["critical_path_ms"], r["compute_critical_path_ms"], r["modelled_exchange_ms"], r["allgather_exposed_ms"], r["recall_at_10"], r.get("one_gpu_same_set_ms"), r["rank0_stage_ms"]))
        print("  per-rank", r["per_rank_compute_ms"])
        print("  sections", r["sections_max_ms"][:9])
    else: print("  " + line.strip()[:300])
PY
