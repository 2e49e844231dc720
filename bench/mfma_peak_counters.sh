cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/bench/mfma_f64_peak.hip -o /tmp/mfma_peak || exit 1
for c in MfmaUtil GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/mp_$c -- /tmp/mfma_peak > /tmp/mp_$c.log 2>&1 < /dev/null
f=$(find /tmp/mp_$c -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then echo "== $c"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Kernel_Name"][:40], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"], int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
PY
fi
done
