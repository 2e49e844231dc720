#!/bin/bash
# Per-launch timeline of the left-looking factorisation (csrc/lfac.hip) on the headline step (GPU box, repo root): bash bench/lfac_trace.sh [TAG]
# rocprofv3 --kernel-trace of a short single-system run; prints, for the last Newton step, every k_lfac dispatch (duration, gap to its predecessor) and what ran beside it.
TAG=${1:-t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/lfac_trace_$TAG -- python $R/bench.py --batch 0 --steps 4 --warmup 2 --no-cpu-baseline --no-c4 --no-c2-c5 > /dev/null 2> $O/lfac_trace_$TAG.err < /dev/null
f=$(find $O/lfac_trace_$TAG -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
lf=[i for i,r in enumerate(rows) if "k_lfac" in r["Kernel_Name"]]
# the last complete factorisation: the last run of 41+ consecutive k_lfac dispatches (other kernels may interleave from the second stream)
n=0
for NP_launches in (41,):
    last=lf[-NP_launches:]
t0=int(rows[last[0]]["Start_Timestamp"])
prev_end=None
tot=0
out=[]
for idx,i in enumerate(last):
    r=rows[i]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    gap=(s-prev_end)/1e3 if prev_end else 0.0
    beside=[q["Kernel_Name"].split("(")[0].replace("calipso::","")[:28] for q in rows[last[0]:last[-1]+1] if "k_lfac" not in q["Kernel_Name"] and int(q["Start_Timestamp"])<e and int(q["End_Timestamp"])>s]
    out.append("launch %3d  start %8.1f us  dur %7.2f  gap %5.2f  beside: %s" % (idx-2,(s-t0)/1e3,(e-s)/1e3,gap," ".join(sorted(set(beside)))))
    prev_end=e
print("\n".join(out))
print("head .. last end: %.1f us" % ((prev_end-t0)/1e3))
PY
rm -rf $O/lfac_trace_$TAG
