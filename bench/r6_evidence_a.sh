#!/bin/bash
# round 6, evidence pass A (GPU box): no-concurrency probe, W-form for groups A/B, counters of k_mf_factor, kernel stats of one C4T group (final binary)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 bench/overlap_probe.hip -o /tmp/overlap_probe 2>/dev/null && timeout 120 /tmp/overlap_probe > $O/overlap_probe.txt 2>&1
for w in 0 1 0 1; do
  CALIPSO_BENCH_GROUP_SOLVE_WFORM=$w timeout 300 python bench.py --no-single --no-c4 --no-c2-c5 --no-cpu-baseline --batched-passes 10 > $O/group_wform_$w.json 2> /dev/null
  python - <<PY
import json
d=json.loads([l for l in open("$O/group_wform_$w.json") if l.startswith("{")][-1])
b=d["config"]["batched"]; p=d["config"]["roofline_phases"]["one_group_alone"]
print("group solve_wform=$w: batched %.1f steps/s, one group alone %.1f; group step %.3f ms: factor %.3f (schur %.3f ldl %.3f) solve_and_refine %.3f" % (b["newton_steps_per_s"], b["one_group_alone_steps_per_s"], p["whole_step_ms"], p["factor"]["ms"], p["factor"]["schur_ms"], p["factor"]["ldl_ms"], p["solve_and_refine"]["ms"]))
PY
done > $O/group_wform_ab.txt 2>&1
bash bench/pmc_mf_factor.sh > $O/pmc_mf_factor.txt 2>&1
bash bench/c4t_group_stats.sh > $O/c4t_group_stats.txt 2>&1; cp gpurun_out/c4t_stats.csv $O/kernel_stats_c4t_group.csv
cat $O/group_wform_ab.txt; tail -20 $O/pmc_mf_factor.txt
