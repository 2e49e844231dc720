#!/bin/bash
# One Newton step of the single C3 system under rocprofv3 --kernel-trace: every launch of the last step in order, with the idle time in front of it
# (gaps above 3 us are what the host or a read-back put there).  bash bench/step_gaps.sh  ->  gpurun_out/step_gaps.txt
# BENCH_ARGS="--config C4T --batch 32 --group 32 --lanes 1 --no-single --batched-passes 4" bash bench/step_gaps.sh: one step of a group instead
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gaps; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py ${BENCH_ARGS:---batch 0 --steps 4 --warmup 2} --no-cpu-baseline --no-c4 --no-c2-c5 > $O/bench.json 2> $O/err.log < /dev/null
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - "$f" "$R/gpurun_out/step_gaps.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
diag = [i for i, n in enumerate(names) if "k_cone_weights" in n and "wide" not in n]      # one per factorisation = per Newton step here
a, b = diag[-2], diag[-1]
out = []; busy = 0; gaps = 0; big = 0
prev_end = int(rows[a - 1]["End_Timestamp"])
mainq = rows[a]["Queue_Id"]              # the queue of the handle's stream; its second stream (the finish of completed solve blocks beside the pivot chain) is another
side = [r for r in rows[a:b] if r["Queue_Id"] != mainq]
for r in rows[a:b]:
    if r["Queue_Id"] != mainq: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - prev_end) / 1e3; d = (e - s) / 1e3
    busy += d; gaps += max(g, 0.0); big += g if g > 3.0 else 0.0
    short = r["Kernel_Name"].split("(")[0].replace("calipso::", "").replace("void ", "")
    out.append("%8.1f us idle  %8.1f us  %s" % (g, d, short))
    prev_end = max(prev_end, e)
out.append("one step: %d launches on the handle's stream, busy %.1f us, idle %.1f us (of which gaps > 3 us: %.1f us); %d more on its second stream beside the pivot chain (%.1f us of kernel time)"
           % (b - a - len(side), busy, gaps, big, len(side), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in side) / 1e3))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print(out[-1])
PY
rm -rf $O/tr
