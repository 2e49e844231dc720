cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_g8 -- python $R/bench.py --batch 8 --group 8 --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_g8.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/prof_g8 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -30 "$f" | cut -c1-200; fi
tail -1 $R/gpurun_out/prof_g8.log | cut -c1-300
