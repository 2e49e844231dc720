#!/bin/bash
# usage (GPU box, from the repo root): bash bench/pmc_kernel.sh OUTDIR KERNEL_REGEX COUNTER [COUNTER ...] -- <bench.py args>
# One rocprofv3 --pmc pass per counter (with --kernel-trace only), rows of the kernels matching KERNEL_REGEX kept.
O=$(realpath -m $1); shift; KR=$1; shift
CS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do CS+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "${CS[@]}"; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/raw_$c -- python $R/bench.py "$@" > $O/pmc_$c.log 2>&1 < /dev/null
  f=$(find $O/raw_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 "$f" > $O/pmc_$c.csv; grep -E "$KR" "$f" | head -200 >> $O/pmc_$c.csv; fi
  rm -rf $O/raw_$c
  python3 - "$O/pmc_$c.csv" "$c" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        rows[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])].append(float(r["Counter_Value"]))
except Exception as e:
    print("no data", e)
for k, v in sorted(rows.items()):
    print("%-14s %-40s grid %-10s n=%3d  mean %.6g" % (sys.argv[2], k[0][-40:], k[1], len(v), sum(v) / len(v)))
PY
done
