#!/bin/bash
# Everything profiles/ cites for a round, in one call on the GPU box (bash bench/evidence_round.sh): output under gpurun_out/round/.
#   profile_round.sh (default bench line, kernel stats and PMC passes of the single system and of one group), the per-launch durations of the
#   LDL^T chain (single / group), the C4 and C4T bench lines and the kernel stats of one C4T group, the chain timeline, the block harnesses, the wide fronts of the multifrontal path.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/round; mkdir -p $O
cd $R
make -C calipso.jl_amd/csrc trace > /dev/null 2>&1      # the stamped build (bench/ldl_trace.py, mf_trace.py, ldl_bulk_trace.py): a stale one lacks the ABI of the round
bash bench/profile_round.sh 12 > $O/profile_round.log 2>&1
bash bench/ldl_step_times.sh 0 0 > /dev/null 2>&1; cp gpurun_out/ldlsteps_0_0/steps.txt $O/ldl_steps_single.txt
bash bench/ldl_step_times.sh 12 1 > /dev/null 2>&1; cp gpurun_out/ldlsteps_12_1/steps.txt $O/ldl_steps_group12_pairs.txt
for c in C4 C4T; do
  timeout 600 python bench.py --config $c --batch 32 --group 16 --lanes 2 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err < /dev/null
done
timeout 600 python bench.py --config C4T --batch 192 --group 64 --lanes 3 --no-cpu-baseline --no-single --no-c4 --no-c2-c5 > $O/bench_C4T_192.json 2> /dev/null < /dev/null
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4t -- python $R/bench.py --config C4T --batch 64 --group 64 --lanes 1 --steps 10 --warmup 2 --batched-passes 10 --no-cpu-baseline --no-single > $O/bench_c4t_group_under_rocprof.json 2> /dev/null < /dev/null)
f=$(find $O/stats_c4t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_c4t_group.csv; rm -rf $O/stats_c4t
bash bench/lfac_trace.sh round > $O/lfac_timeline.txt 2>&1      # per-launch timeline of the left-looking factorisation (one dense system)
timeout 300 python bench/lfac_items.py > $O/lfac_items.txt 2>&1
timeout 600 python bench/lfac_sizes.py > $O/lfac_sizes.txt 2>&1          # left-looking against right-looking at other sizes than C3
CALIPSO_HIP_LFAC=0 timeout 300 python bench/ldl_trace.py > $O/ldl_chain_timeline.txt 2>&1      # the right-looking chain (what a group's members take), stamped build
timeout 300 python bench/mf_trace.py > $O/mf_trace.txt 2>&1
for g in 1 16 64; do timeout 200 python bench/mf_trace_group.py $g 2> /dev/null | head -9; done > $O/mf_trace_group.txt      # the same front while a group's other fronts run beside it
timeout 200 python bench/mf_solve_trace.py 16 2> /dev/null > $O/mf_solve_trace.txt      # the phases of the sweeps' launches (forward / backward) in a group of 16
bash bench/c4t_group_stats.sh > $O/c4t_group_stats.txt 2>&1                                 # kernel time per group step of 16 C4T members
bash bench/pmc_mf_factor.sh > $O/pmc_mf_factor.txt 2>&1                                    # counters of k_mf_factor (a group of 16)
timeout 300 python bench/ldl_bulk_trace.py 12 > $O/ldl_bulk_trace.txt 2>&1
bash bench/step_gaps.sh > /dev/null 2>&1; cp gpurun_out/step_gaps.txt $O/step_gaps_under_rocprof.txt
timeout 300 python bench/wide_fronts.py 1500 5 > $O/wide_fronts.txt 2>&1      # fronts beyond the LDS: many workgroups per front against one (sparse_wide.hpp)
timeout 200 python bench/wide_fronts.py 4000 2 1 >> $O/wide_fronts.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wf -- python $R/bench/wide_fronts.py 1500 5 1 > /dev/null 2>&1 < /dev/null)
f=$(find $O/stats_wf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_wide_fronts.csv; rm -rf $O/stats_wf
hipcc -O3 --offload-arch=gfx950 bench/diag_bench3.hip -o /tmp/d3 2>/dev/null && timeout 60 /tmp/d3 > $O/diag_bench3.txt
ls -la $O | head -60
# round 6: the small-problem kernel (rates, phase clocks), what overlaps on this chip (probes), the 256-thread Schur viability probe
for a in "49 40 0 4096 20" "49 40 20 4096 10" "24 12 24 8192 20"; do timeout 200 python bench/small_newton_rate.py $a 2>/dev/null | tail -1 > $O/small_newton_rate_$(echo $a | tr ' ' '_').json; done
bash bench/small_newton_phases.sh 2>&1 | tail -14 > $O/small_newton_phases.txt
bash bench/small_newton_threads.sh > $O/small_newton_threads.txt 2>&1
for p in overlap_probe overlap_probe2 overlap_probe3 overlap_probe4 schur64_probe; do hipcc --offload-arch=gfx950 -O3 bench/$p.hip -o /tmp/$p 2>/dev/null && timeout 120 /tmp/$p > $O/$p.txt 2>&1; done
ls -la $O | wc -l
