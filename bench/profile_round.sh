#!/bin/bash
# Round evidence, run on the GPU box from the repo root (bash bench/profile_round.sh [G]): writes everything under gpurun_out/round/.
#   1. the default bench line (with the CPU baseline)
#   2. rocprofv3 --kernel-trace --stats of (a) the single-system region alone (--batch 0: the headline) and (b) ONE group of G instances in
#      flight (--no-single, one lane): per-kernel averages are uncontended in both
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil, LDS bank conflicts / active cycles) of the same two commands, summarised by
#      bench/pmc_summary.py into pmc_summary.json = {"single": {...}, "group": {...}}  (bench.py reads it from profiles/r05_pmc_summary.json)
G=${1:-12}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SINGLE="python $R/bench.py --batch 0 --steps 10 --warmup 2 --no-cpu-baseline --no-c4 --no-c2-c5"
GROUP="python $R/bench.py --batch $G --group $G --lanes 1 --steps 10 --warmup 2 --batched-passes 10 --no-cpu-baseline --no-single --no-c4 --no-c2-c5"
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null
for mode in single group; do
  if [ $mode = single ]; then CMD=$SINGLE; Z=1; else CMD=$GROUP; Z=$G; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -- $CMD > $O/bench_${mode}_under_rocprof.json 2> $O/stats_$mode.err < /dev/null
  f=$(find $O/stats_$mode -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $O/kernel_stats_$mode.csv; fi
  rm -rf $O/stats_$mode
  for c in FETCH_SIZE WRITE_SIZE MfmaUtil SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${mode}_$c -- $CMD --steps 2 --warmup 1 --batched-passes 2 > $O/pmc_${mode}_$c.log 2>&1 < /dev/null
    f=$(find $O/pmc_${mode}_$c -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then grep -E "Correlation_Id|k_lfac|k_schur|k_ldl_step|k_ldl_scale|k_tinv_merge|k_gemv_t|k_gemv_n_partial|k_trsv|k_solve_tail|k_refine_x_fused|k_wform_product" "$f" | head -400 > $O/pmc_${mode}_$c.csv; fi
    rm -rf $O/pmc_${mode}_$c
  done
  if [ -s $O/pmc_${mode}_FETCH_SIZE.csv ] && [ -s $O/pmc_${mode}_WRITE_SIZE.csv ]; then
    python $R/bench/pmc_summary.py $O/pmc_${mode}_FETCH_SIZE.csv $O/pmc_${mode}_WRITE_SIZE.csv $Z $O/pmc_summary_$mode.json MfmaUtil:$O/pmc_${mode}_MfmaUtil.csv SQ_LDS_BANK_CONFLICT:$O/pmc_${mode}_SQ_LDS_BANK_CONFLICT.csv SQ_LDS_IDX_ACTIVE:$O/pmc_${mode}_SQ_LDS_IDX_ACTIVE.csv > /dev/null
  fi
done
python - <<PY
import json, os
out = {}
for mode in ("single", "group"):
    p = "$O/pmc_summary_%s.json" % mode
    if os.path.exists(p):
        out[mode] = json.load(open(p))
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1)
PY
tail -c 800 $O/bench_default.json; echo; head -8 $O/kernel_stats_single.csv | cut -c1-150; head -6 $O/kernel_stats_group.csv | cut -c1-150
