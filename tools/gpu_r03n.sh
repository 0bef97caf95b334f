#!/bin/bash
# round 3: full -m gpu suite, then the profile set (bench line, kernel trace, FETCH/WRITE/SQ PMC passes) as r03_a
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --durations=8 -p no:cacheprovider > gpurun_out/r03n_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03n_pytest.log )
grep -v "^  File\|Extension modules" gpurun_out/r03n_pytest.log | tail -n 25
bash tools/collect_profiles.sh r03_a 2>&1 | tail -n 12
tail -c 600 gpurun_out/r03_a_bench_1m.json.log
