#!/bin/bash
# round 3, GPU call A: full -m gpu suite (new tests first), bench line, sharded critical path at configs[3] size
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_search.py tests/test_gpu_kernels.py tests/test_gpu_build.py \
    tests/test_gpu_integration.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz_parity.py tests/test_gpu_fullsize.py tests/test_handover_reference.py \
    -m gpu -q --maxfail=10 --durations=12 -p no:cacheprovider > gpurun_out/r03a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a_pytest.log )
tail -n 40 gpurun_out/r03a_pytest.log
( timeout 500 python bench.py --steps 5 --warmup 2 > gpurun_out/r03a_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/r03a_bench.log )
tail -c 1500 gpurun_out/r03a_bench.log
( timeout 500 python tools/rank_critical_path.py --world 8 --n 10000000 --trees 12 > gpurun_out/r03a_critpath.log 2>&1; echo "critpath rc=$?" >> gpurun_out/r03a_critpath.log )
tail -c 3000 gpurun_out/r03a_critpath.log
