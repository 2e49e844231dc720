#!/bin/bash
# Round evidence, run on the GPU box from the repo root (bash bench/profile_round.sh [G]): writes everything under gpurun_out/round/.
#   1. the default bench line (with the CPU baseline)
#   2. rocprofv3 --kernel-trace --stats of ONE unit in flight (a group of G instances, one lane): per-kernel averages are uncontended
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil, LDS bank conflicts / active cycles) of the same command, summarised by bench/pmc_summary.py
G=${1:-12}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --batch $G --group $G --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --no-single"
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/bench_one_unit_under_rocprof.json 2> $O/stats.err < /dev/null
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; fi
for c in FETCH_SIZE WRITE_SIZE MfmaUtil SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- $CMD --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1 < /dev/null
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); if [ -n "$f" ]; then grep -E "Correlation_Id|k_schur|k_ldl_trailing|k_gemv_t" "$f" | head -120 > $O/pmc_$c.csv; fi
done
if [ -s $O/pmc_FETCH_SIZE.csv ] && [ -s $O/pmc_WRITE_SIZE.csv ]; then python $R/bench/pmc_summary.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $G $O/pmc_summary.json MfmaUtil:$O/pmc_MfmaUtil.csv SQ_LDS_BANK_CONFLICT:$O/pmc_SQ_LDS_BANK_CONFLICT.csv SQ_LDS_IDX_ACTIVE:$O/pmc_SQ_LDS_IDX_ACTIVE.csv; fi
if [ -x /opt/rocm/bin/hipcc ]; then bash $R/bench/mfma_peak_counters.sh > $O/mfma_f64_peak_counters.txt 2>&1; fi
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MfmaUtil $O/pmc_SQ_LDS_BANK_CONFLICT $O/pmc_SQ_LDS_IDX_ACTIVE
tail -c 600 $O/bench_default.json; echo; head -6 $O/kernel_stats.csv | cut -c1-160
