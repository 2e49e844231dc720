#!/bin/bash
# Quick look at the headline step (GPU box, repo root): bash bench/quick_single.sh [TAG] — two bench processes (single system only) and one rocprofv3
# --kernel-trace --stats pass of the same command; kernel stats -> gpurun_out/quick_TAG_kernel_stats.csv, summary lines on stdout.
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
for i in 1 2; do
  python bench.py --batch 0 --no-c4 --no-c2-c5 --no-cpu-baseline --steps 30 > $O/quick_${TAG}_$i.json 2> $O/quick_${TAG}_$i.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/quick_${TAG}_$i.json") if l.startswith("{")][-1])
p=d["config"]["roofline_phases"]["single_system"]
r=d["roofline"]; ch=r if ("k_ldl" in r["kernel"] or "k_lfac" in r["kernel"]) else r["secondary"][0]
print("$TAG run $i: value %.1f  ms/step %.3f  solve_and_refine %.3f  schur %.3f  ldl %.3f  chain %.3f (%d launches)" % (d["value"], d["ms_per_step"], p["solve_and_refine"]["ms"], p["factor"]["schur_ms"], p["factor"]["ldl_ms"], ch["ms_per_step"], ch["launches_per_step"]))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/quick_stats_$TAG -- python $R/bench.py --batch 0 --steps 10 --warmup 2 --no-cpu-baseline --no-c4 --no-c2-c5 > /dev/null 2> $O/quick_stats_$TAG.err < /dev/null
f=$(find $O/quick_stats_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/quick_${TAG}_kernel_stats.csv
rm -rf $O/quick_stats_$TAG
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/quick_${TAG}_kernel_stats.csv")))
rows=[r for r in rows if "mfma_f64_peak" not in r["Name"]]
steps=float([r for r in rows if "k_cone_search" in r["Name"]][0]["Calls"])      # one cone search per Newton step
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:32]:
    print("%-70s calls/step %6.1f avg %7.2f us  per step %7.1f us" % (r["Name"][:70], int(r["Calls"])/steps, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/steps/1e3))
PY
