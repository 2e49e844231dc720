#!/bin/bash
# per-launch durations of the LDL^T kernels of ONE group factorisation (rocprofv3 kernel trace), in launch order: bash bench/ldl_step_times.sh [G] [pairs]
G=${1:-12}; P=${2:-1}   # G = 0: the single-system region instead of a group
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ldlsteps_${G}_$P; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$G" = 0 ]; then ARGS="--batch 0 --steps 2 --warmup 1"; else ARGS="--batch $G --group $G --lanes 1 --steps 2 --warmup 1 --batched-passes 2 --no-single"; fi
CALIPSO_HIP_LDL_PAIRS=$P timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py $ARGS --no-cpu-baseline --no-c4 > $O/bench.json 2> $O/err.log < /dev/null
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - "$f" "$O/steps.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last factorisation: from the last k_ldl_diag to the first k_tinv_merge after it
last = max(i for i, n in enumerate(names) if "k_ldl_diag" in n)
out = []
for r in rows[last:]:
    n = r["Kernel_Name"]
    if "k_trsv" in n or "k_gemv" in n or "k_residual" in n: break
    short = n.split("(")[0].replace("calipso::", "")
    out.append("%-28s %8.1f us" % (short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
t0 = int(rows[last]["Start_Timestamp"]); 
end = [int(r["End_Timestamp"]) for r in rows[last:last + len(out)]][-1]
out.append("total %.1f us over %d launches" % ((end - t0) / 1e3, len(out)))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
PY
rm -rf $O/tr
tail -1 $O/steps.txt
