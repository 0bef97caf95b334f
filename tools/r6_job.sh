#!/bin/bash
# round-6 GPU-box jobs: tools/r6_job.sh <tag> <job> ...   (outputs under gpurun_out/<tag>_*)
#   leafab           : leaf-kernel A/B -- default library against pynndescent_amd/_exp/lib_<variant>.so for every variant
#                      named in $VARIANTS (qbench lines + per-launch leaf kernel times from a kernel trace)
#   ktests:<expr>    : pytest -m gpu -k <expr>
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for job in "$@"; do
  name=${job%%:*}; arg=""; [[ "$job" == *:* ]] && arg=${job#*:}
  case $name in
    ktests)
      ( cd $R && timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$arg" > $O/${tag}_ktests.log 2>&1; echo "pytest rc=$?" >> $O/${tag}_ktests.log )
      tail -n 8 $O/${tag}_ktests.log | cut -c1-220 ;;
    leafab)
      for v in default $VARIANTS; do
        lib=$R/pynndescent_amd/libpynnd_amd.so; [ $v != default ] && lib=$R/pynndescent_amd/_exp/lib_$v.so
        for kk in ${KS:-15}; do
          echo "== $v k=$kk" >> $O/${tag}_leafab.log
          ( cd $R && PYNND_AMD_LIB=$lib timeout 300 python tools/qbench.py --k $kk --reps 3 >> $O/${tag}_leafab.log 2>&1 )
          rm -rf /tmp/p_lab; ( cd /tmp && PYNND_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_lab -- python $R/tools/qbench.py --k $kk --reps 2 --no-recall > /tmp/p_lab.log 2>&1 )
          python $R/tools/per_launch.py /tmp/p_lab k_leaf_join 16 >> $O/${tag}_leafab.log 2>&1
        done
      done
      cat $O/${tag}_leafab.log | cut -c1-400 ;;
    sqpmc)  # SQ counters of every kernel of one qbench build (arg: qbench arguments, ':' separated); library: $LIB (default: the product library)
      lib=${LIB:-$R/pynndescent_amd/libpynnd_amd.so}
      rm -rf /tmp/p_sq; ( cd /tmp && PYNND_AMD_LIB=$lib timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU --output-format csv -d /tmp/p_sq -- python $R/tools/qbench.py --reps 1 --no-recall ${arg//:/ } > /tmp/p_sq.log 2>&1 )
      python $R/tools/prof_summary.py /tmp/p_sq $O/${tag}_pmc_sq.txt; grep -A9 -E "${KGREP:-k_leaf_join}" $O/${tag}_pmc_sq.txt | head -60 | cut -c1-150 ;;
    *) bash $R/tools/gpu_job.sh $tag "$job" ;;
  esac
done
