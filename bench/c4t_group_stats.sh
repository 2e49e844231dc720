#!/bin/bash
# kernel stats of ONE C4T group of 16 in flight (rocprofv3 --kernel-trace --stats): bash bench/c4t_group_stats.sh -> gpurun_out/c4t_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/c4t; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --config C4T --batch 16 --group 16 --lanes 1 --steps 10 --warmup 2 --batched-passes 10 --no-cpu-baseline --no-single > $O/bench.json 2> /dev/null < /dev/null)
f=$(find $O/st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/c4t_stats.csv; rm -rf $O/st
python - "$R/gpurun_out/c4t_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    if "mfma_f64_peak" in r["Name"] or "rocclr" in r["Name"]: continue
    tot += int(r["TotalDurationNs"])
print("kernel time per pass: %.3f ms" % (tot / 12e6))
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:14]:
    if "mfma_f64_peak" in r["Name"]: continue
    print("%7.1f us/pass %5.1f calls %6.1f us  %s" % (int(r["TotalDurationNs"]) / 12e3, int(r["Calls"]) / 12, float(r["AverageNs"]) / 1e3, r["Name"].replace("calipso::", "").replace("(anonymous namespace)::", "")[:60]))
PY
