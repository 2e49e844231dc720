#!/bin/bash
# A/B of "opt.solve_block" on the headline step (single system only, W-form solves): bash bench/ab_solve_block.sh [widths...]; prints value, solve_and_refine.ms, ldl_ms, chain
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for sb in ${@:-1024 2048 1024 2048}; do
  CALIPSO_BENCH_SOLVE_BLOCK=$sb python bench.py --batch 0 --no-c4 --no-c2-c5 --no-cpu-baseline --steps 30 > gpurun_out/ab_sb_$sb.json 2>gpurun_out/ab_sb_$sb.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab_sb_$sb.json") if l.startswith("{")][-1])
p=d["config"]["roofline_phases"]["single_system"]
r=d["roofline"]; ch=r if ("k_ldl" in r["kernel"] or "k_lfac" in r["kernel"]) else r["secondary"][0]
print("solve_block=$sb value %.1f ms/step %.3f  solve_and_refine %.3f  factor %.3f (schur %.3f ldl %.3f) chain %.3f rounds %s" % (d["value"], d["ms_per_step"], p["solve_and_refine"]["ms"], p["factor"]["ms"], p["factor"]["schur_ms"], p["factor"]["ldl_ms"], ch["ms_per_step"], d["config"].get("refinement_rounds")))
PY
done
