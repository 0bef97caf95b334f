#!/bin/bash
# One parameterised GPU-box script (replaces the per-call gpu_r03?.sh files):  tools/gpu_job.sh <tag> <job> [<job> ...]
# Every job writes gpurun_out/<tag>_<job>.log (and summaries next to it).  Jobs:
#   tests[:expr]   pytest -m gpu (optionally -k expr)         bench[:args]   bench.py with args (default --steps 5 --warmup 2)
#   trace          rocprofv3 kernel trace of a short bench    pmc            FETCH / WRITE / SQ counter passes (tools/collect_profiles.sh)
#   critpath:G:N:T rank critical path, G ranks, N points, T trees          k30trace        kernel trace of the k = 30 workload
#   onetree        kernel trace of one tree over 10 M points  py:<file>[:args]  python <file> args
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O; export TMPDIR=/tmp
for job in "$@"; do
  name=${job%%:*}; arg=""; [[ "$job" == *:* ]] && arg=${job#*:}
  log=$O/${tag}_${name}.log
  case $name in
    tests)
      ( cd $R && timeout 1500 python -m pytest tests -m gpu -q -rP --maxfail=10 --durations=10 -p no:cacheprovider ${arg:+-k "$arg"} > $log 2>&1; echo "pytest rc=$?" >> $log )
      grep -c PASSED $log; tail -n 25 $log | cut -c1-200 ;;
    bench)
      ( cd $R && timeout 900 python bench.py ${arg:---steps 5 --warmup 2} > $log 2>&1; echo "bench rc=$?" >> $log )
      tail -c 2500 $log ;;
    trace)
      rm -rf /tmp/p_tr; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $log 2>&1 )
      python $R/tools/prof_summary.py /tmp/p_tr $O/${tag}_kernel_stats_bench1m.txt; head -45 $O/${tag}_kernel_stats_bench1m.txt | cut -c1-150 ;;
    pmc)
      bash $R/tools/collect_profiles.sh $tag > $log 2>&1; tail -n 12 $log ;;
    critpath)
      IFS=: read -r G N T <<< "$arg"
      ( cd $R && timeout 900 python tools/rank_critical_path.py --world ${G:-8} --n ${N:-10000000} --trees ${T:-12} > $O/${tag}_critpath_w${G:-8}.log 2>&1; echo "rc=$?" >> $O/${tag}_critpath_w${G:-8}.log )
      tail -c 3500 $O/${tag}_critpath_w${G:-8}.log ;;
    k30trace)
      rm -rf /tmp/p_k30; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k30 -- python $R/tools/prof_k30.py > $log 2>&1 )
      python $R/tools/prof_summary.py /tmp/p_k30 $O/${tag}_k30_kernel_stats.txt; head -24 $O/${tag}_k30_kernel_stats.txt | cut -c1-150 ;;
    onetree)
      rm -rf /tmp/p_ot; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ot -- python $R/tools/prof_one_tree_10m.py 10000000 ${arg:-1} > $log 2>&1 )
      python $R/tools/prof_summary.py /tmp/p_ot $O/${tag}_one_tree_10m_kernel_stats.txt; head -28 $O/${tag}_one_tree_10m_kernel_stats.txt | cut -c1-150 ;;
    pytrace)  # kernel trace of python <file> args
      f=${arg%%:*}; a=""; [[ "$arg" == *:* ]] && a=${arg#*:}
      rm -rf /tmp/p_py; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_py -- python $R/$f ${a//:/ } > $O/${tag}_$(basename $f .py)_trace.log 2>&1 )
      python $R/tools/prof_summary.py /tmp/p_py $O/${tag}_$(basename $f .py)_kernel_stats.txt; head -${NHEAD:-30} $O/${tag}_$(basename $f .py)_kernel_stats.txt | cut -c1-150 ;;
    pypmc)  # counter pass of python <file> args; counters in $PMC
      f=${arg%%:*}; a=""; [[ "$arg" == *:* ]] && a=${arg#*:}
      rm -rf /tmp/p_pmc; ( cd /tmp && timeout 600 rocprofv3 --pmc $PMC --output-format csv -d /tmp/p_pmc -- python $R/$f ${a//:/ } > $O/${tag}_$(basename $f .py)_pmc.log 2>&1 )
      python $R/tools/prof_summary.py /tmp/p_pmc $O/${tag}_$(basename $f .py)_pmc.txt; grep -A${NHEAD:-10} -E "${KGREP:-k_}" $O/${tag}_$(basename $f .py)_pmc.txt | head -80 | cut -c1-150 ;;
    py)
      f=${arg%%:*}; a=""; [[ "$arg" == *:* ]] && a=${arg#*:}
      ( cd $R && timeout 900 python $f ${a//:/ } > $O/${tag}_$(basename $f .py).log 2>&1; echo "rc=$?" >> $O/${tag}_$(basename $f .py).log )
      tail -c 3000 $O/${tag}_$(basename $f .py).log ;;
    *) echo "unknown job $job" ;;
  esac
done
