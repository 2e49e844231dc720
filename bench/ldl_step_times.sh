#!/bin/bash
# per-launch durations of the LDL^T kernels of ONE group factorisation (rocprofv3 kernel trace), in launch order: bash bench/ldl_step_times.sh [G] [pairs]
G=${1:-12}; P=${2:-1}   # G = 0: the single-system region instead of a group
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ldlsteps_${G}_$P; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$G" = 0 ]; then ARGS="--batch 0 --steps 2 --warmup 1"; else ARGS="--batch $G --group $G --lanes 1 --steps 2 --warmup 1 --batched-passes 2 --no-single"; fi
# (the right-looking schedule — CALIPSO_HIP_LFAC=0: what a group takes; one dense system alone takes the left-looking one, bench/lfac_trace.sh —
# (the listing is taken with the finish AFTER the chain — CALIPSO_HIP_LDL_OVERLAP=0 —: under the tracer the host feeds the second stream late and the join
# waits for it; the product's default, the finish of completed solve blocks beside the chain, is timed below without the tracer)
CALIPSO_HIP_LFAC=0 CALIPSO_HIP_LDL_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py $ARGS --no-cpu-baseline --no-c4 --no-c2-c5 > $O/bench.json 2> $O/err.log < /dev/null
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
if [ "$G" = 0 ]; then
  cd $R && timeout 300 python bench.py --batch 0 --no-cpu-baseline --no-c4 --no-c2-c5 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{\"metric'):
        d = json.loads(l); p = d['config']['roofline_phases']['single_system'] if 'roofline_phases' in d['config'] else d['roofline_phases']['single_system']
        print('without the tracer, HIP events, mean of the timed steps (the default: finish of the completed solve blocks on the second stream beside the chain): whole factorisation %.1f us, of which the panel steps %.1f us' % (1e3 * p['factor']['ldl_ms'], 1e3 * d['roofline']['ms_per_step']))
" >> $O/steps.txt
fi
tail -2 $O/steps.txt
