R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmcmf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$n -- python $R/bench.py --config C4T --batch 16 --group 16 --lanes 1 --steps 2 --warmup 1 --batched-passes 2 --no-cpu-baseline --no-single > /dev/null 2>&1 < /dev/null
  f=$(find $O/$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "k_mf_factor" in k: agg["k_mf_factor"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in d.items(): print(k, c, "mean per launch %.0f over %d" % (sum(v) / len(v), len(v)))
PY
  else echo "no csv for $c"; fi
  rm -rf $O/$n
done
