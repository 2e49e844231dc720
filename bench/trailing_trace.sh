# per-launch durations of k_ldl_trailing within one group step (grid size tells the panel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -- python $R/bench.py --batch ${1:-12} --group ${1:-12} --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-single > /tmp/tt.log 2>&1 < /dev/null
f=$(find /tmp/tt -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python3 - "$f" ${1:-12} <<'PY'
import csv,sys
G=int(sys.argv[2])
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_ldl_trailing" in r["Kernel_Name"]]
last=rows[-39:]
print("tiles/inst  duration_us  TFLOP/s  (S r+w GB/s)")
for idx, r in enumerate(last):
    ntr = 39 - idx                                   # C3: 39 panel steps, ntr 64-row blocks left after panel idx
    wg = ntr * (ntr + 1) // 2                        # tiles per instance (the launch itself uses persistent workgroups)
    dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    fl=wg*G*2*64**3
    print("%6d %10.1f %8.1f %10.0f" % (wg, dur, fl/dur*1e-6, wg*G*2*64*64*8/dur*1e-3))
PY
fi
