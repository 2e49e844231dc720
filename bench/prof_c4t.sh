cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc4t -- python $R/bench.py --config C4T --batch 12 --group 12 --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --no-single > /tmp/pc4t.log 2>&1 < /dev/null
f=$(find /tmp/pc4t -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=22
for r in rows[:16]:
    print("%-24s calls/step %5.1f  per step %7.3f ms  avg %8.1f us" % (r['Name'].split('(')[0].replace('calipso::',''), int(r['Calls'])/n, int(r['TotalDurationNs'])/1e6/n, float(r['AverageNs'])/1e3))
PY
fi
tail -1 /tmp/pc4t.log | cut -c1-100
