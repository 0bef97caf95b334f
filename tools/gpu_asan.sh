#!/bin/bash
# ASan build of the HOST side of comm.hip / shard.hip / capi.hip (-Xarch_host: the device pass stays as it is) and the
# repetition loop of tools/asan_loop.py under it.  Build part runs anywhere (hipcc cross-compiles); run part on the GPU box:
#   tools/gpu_asan.sh build [nolock]   tools/gpu_asan.sh run [reps]        (nolock: without the process-wide lifecycle lock)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/pynndescent_amd/csrc
# runtime: gcc's libasan (the ROCm compiler-rt's ASan runtime intercepts HSA allocations and wants the instrumented ROCm
# stack: "out of memory" in hsa_amd_memory_pool_allocate at the first hipMalloc) + a shim for three helpers it lacks
rt="$(gcc -print-file-name=libasan.so.6) /usr/lib/x86_64-linux-gnu/libstdc++.so.6 $R/tools/dbg/asan_shim.so"  # libstdc++ first: the __cxa_throw interceptor needs the real one at start-up
case ${1:-build} in
  build)
    mkdir -p _obj_asan ../_exp
    for f in capi comm shard; do
      /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer ${2:+-DNND_TEST_NO_LIFECYCLE_LOCK} -c $f.hip -o _obj_asan/$f.o
    done
    objs=""; for f in prep rpforest leaf_join join sample merge finalize prune hubtree query; do objs="$objs _obj/$f.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_exp/lib_asan.so $objs _obj_asan/capi.o _obj_asan/comm.o _obj_asan/shard.o -ldl -lpthread
    gcc -O1 -shared -fPIC -o $R/tools/dbg/asan_shim.so $R/tools/dbg/asan_shim.c
    echo built ../_exp/lib_asan.so ;;
  run)
    mkdir -p $R/gpurun_out
    cd $R
    # (an intercepted dlopen loses the caller's RUNPATH: torch's own libraries have to be on the search path)
    tl=$(python -c "import importlib.util as u; print(u.find_spec('torch').submodule_search_locations[0] + '/lib')")
    LD_LIBRARY_PATH=$tl:$LD_LIBRARY_PATH LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:alloc_dealloc_mismatch=0 PYNND_AMD_LIB=$R/pynndescent_amd/_exp/lib_asan.so \
      timeout ${TMO:-900} python tools/asan_loop.py ${2:-200} > gpurun_out/asan_loop.log 2>&1; echo "rc=$?" >> gpurun_out/asan_loop.log
    grep -n "ERROR: AddressSanitizer" -A40 gpurun_out/asan_loop.log | head -80 | cut -c1-220
    tail -n 8 gpurun_out/asan_loop.log | cut -c1-220 ;;
esac
