#!/bin/bash
# A/B of an environment switch of the library on the headline step (single system only): bash bench/ab_env.sh NAME v1 v2 ...; prints value and the phases per process
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
NAME=$1; shift
for v in "$@"; do
  env $NAME=$v python bench.py --batch 0 --no-c4 --no-c2-c5 --no-cpu-baseline --steps 30 > gpurun_out/ab_env.json 2>gpurun_out/ab_env.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab_env.json") if l.startswith("{")][-1])
p=d["config"]["roofline_phases"]["single_system"]
r=d["roofline"]; ch=r if ("k_ldl" in r["kernel"] or "k_lfac" in r["kernel"]) else r["secondary"][0]
print("$NAME=$v value %.1f ms/step %.3f  solve_and_refine %.3f  factor %.3f (schur %.3f ldl %.3f) chain %.3f rounds %s" % (d["value"], d["ms_per_step"], p["solve_and_refine"]["ms"], p["factor"]["ms"], p["factor"]["schur_ms"], p["factor"]["ldl_ms"], ch["ms_per_step"], d["config"].get("refinement_rounds")))
PY
done
