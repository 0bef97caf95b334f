#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_build.py \
    tests/test_gpu_integration.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz_parity.py tests/test_gpu_fullsize.py tests/test_handover_reference.py \
    -m gpu -q --maxfail=10 --durations=12 -p no:cacheprovider > gpurun_out/r03b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03b_pytest.log )
grep -v "^  File\|Extension modules" gpurun_out/r03b_pytest.log | tail -n 60
( timeout 300 python tools/host_breakdown.py > gpurun_out/r03b_host_breakdown.log 2>&1 ); cat gpurun_out/r03b_host_breakdown.log
( timeout 500 python tools/rank_critical_path.py --world 8 --n 10000000 --trees 12 > gpurun_out/r03b_critpath.log 2>&1; echo "critpath rc=$?" >> gpurun_out/r03b_critpath.log )
tail -c 2500 gpurun_out/r03b_critpath.log
