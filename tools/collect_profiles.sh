#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh <tag>
# bench line (with cpu baseline), kernel trace + stats, FETCH_SIZE calibration, HBM traffic PMC passes, SQ counters of the
# final kernels -> gpurun_out/<tag>_*
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python bench.py --steps 5 --warmup 2 > $O/${tag}_bench_1m.json.log 2>&1 )
rm -rf /tmp/p_tr /tmp/p_fe /tmp/p_wr /tmp/p_sq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/p_tr.log 2>&1
python $R/tools/prof_summary.py /tmp/p_tr $O/${tag}_kernel_stats_bench1m.txt
timeout 200 python $R/tools/fetch_calib.py $O/${tag}_fetch_calibration.json > /tmp/p_cal.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fe -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /tmp/p_fe.log 2>&1
python $R/tools/prof_summary.py /tmp/p_fe $O/${tag}_pmc_fetch.txt
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_wr -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /tmp/p_wr.log 2>&1
python $R/tools/prof_summary.py /tmp/p_wr $O/${tag}_pmc_write.txt
python $R/tools/pmc_traffic.py /tmp/p_fe /tmp/p_wr $O/${tag}_traffic.json $O/${tag}_fetch_calibration.json
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/p_sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /tmp/p_sq.log 2>&1
python $R/tools/prof_summary.py /tmp/p_sq $O/${tag}_pmc_sq.txt
tail -n 2 /tmp/p_cal.log /tmp/p_fe.log /tmp/p_wr.log /tmp/p_sq.log
# the k = 30 workload (the reference's default n_neighbors): kernel trace + SQ counters of its kernels
rm -rf /tmp/p_k30t /tmp/p_k30s
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k30t -- python $R/tools/prof_k30.py > /tmp/p_k30t.log 2>&1
python $R/tools/prof_summary.py /tmp/p_k30t $O/${tag}_k30_kernel_stats.txt
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/p_k30s -- python $R/tools/prof_k30.py > /tmp/p_k30s.log 2>&1
python $R/tools/prof_summary.py /tmp/p_k30s $O/${tag}_k30_pmc_sq.txt
