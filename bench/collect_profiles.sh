#!/bin/bash
# Copy the summaries of bench/evidence_round.sh (gpurun_out/round/, scratch) into profiles/ under the round's prefix: bash bench/collect_profiles.sh r05
P=${1:-r05}; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/round; D=$R/profiles
cpy() { [ -s "$O/$1" ] && cp "$O/$1" "$D/${P}_$2"; }
cpy bench_default.json bench_default.json
cpy bench_single_under_rocprof.json bench_single_under_rocprof.json
cpy bench_group_under_rocprof.json bench_group_under_rocprof.json
cpy kernel_stats_single.csv kernel_stats_single.csv
cpy kernel_stats_group.csv kernel_stats_group.csv
cpy kernel_stats_c4t_group.csv kernel_stats_c4t_group.csv
[ -s "$O/kernel_stats_c4t_group.csv" ] && python3 "$R/bench/steps_only_stats.py" "$O/kernel_stats_c4t_group.csv" "$D/${P}_kernel_stats_c4t_group_steps_only.csv"
cpy pmc_summary.json pmc_summary.json
for m in single group; do for c in FETCH_SIZE WRITE_SIZE MfmaUtil SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do cpy pmc_${m}_$c.csv pmc_${m}_$c.csv; done; done
cpy bench_C4.json bench_C4.json
cpy bench_C4T.json bench_C4T.json
cpy bench_C4T_192.json bench_C4T_192.json
cpy ldl_steps_single.txt ldl_steps_single_right_looking.txt
cpy ldl_steps_group12_pairs.txt ldl_steps_group12_pairs.txt
cpy lfac_timeline.txt lfac_timeline.txt
cpy lfac_items.txt lfac_items.txt
cpy lfac_sizes.txt lfac_sizes.txt
cpy ldl_chain_timeline.txt ldl_chain_timeline_right_looking.txt
cpy mf_trace.txt mf_front_timeline.txt
cpy mf_trace_group.txt mf_front_timeline_group.txt
cpy mf_solve_trace.txt mf_solve_timeline.txt
cpy c4t_group_stats.txt c4t_group_stats.txt
cpy pmc_mf_factor.txt pmc_mf_factor.txt
cpy ldl_bulk_trace.txt ldl_bulk_trace.txt
cpy step_gaps_under_rocprof.txt step_gaps_under_rocprof.txt
cpy diag_bench3.txt diag_bench3.txt
cpy wide_fronts.txt wide_fronts.txt
cpy kernel_stats_wide_fronts.csv kernel_stats_wide_fronts.csv
for f in small_newton_rate_49_40_0_4096_20.json small_newton_rate_49_40_20_4096_10.json small_newton_rate_24_12_24_8192_20.json small_newton_phases.txt small_newton_threads.txt schur64_probe.txt; do cpy $f $f; done
cpy overlap_probe.txt overlap_probe_1.txt; cpy overlap_probe2.txt overlap_probe_2_short_kernels.txt; cpy overlap_probe3.txt overlap_probe_3_which_property.txt; cpy overlap_probe4.txt overlap_probe_4_saturating_mfma.txt
[ -s "$R/gpurun_out/sparse_ldl_rate.json" ] && cp "$R/gpurun_out/sparse_ldl_rate.json" "$D/${P}_sparse_ldl_rate.json"
ls $D | grep "^${P}_" | wc -l
