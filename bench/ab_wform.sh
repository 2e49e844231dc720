#!/bin/bash
# A/B of "opt.solve_wform" on the headline step (single system only), two processes each; prints value, solve_and_refine.ms, factor ldl_ms
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for w in 1 0 1 0; do
  CALIPSO_BENCH_SOLVE_WFORM=$w python bench.py --batch 0 --no-c4 --no-c2-c5 --no-cpu-baseline --steps 30 > gpurun_out/ab_wform_$w.json 2>gpurun_out/ab_wform_$w.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_wform_$w.json"))
p=d["config"]["roofline_phases"]["single_system"]
print("wform=$w value %.1f ms/step %.3f  solve_and_refine %.3f  factor %.3f (schur %.3f ldl %.3f) chain %.3f" % (d["value"], d["ms_per_step"], p["solve_and_refine"]["ms"], p["factor"]["ms"], p["factor"]["schur_ms"], p["factor"]["ldl_ms"], d["roofline"]["ms_per_step"] if "k_ldl" in d["roofline"]["kernel"] else d["roofline"]["secondary"][0]["ms_per_step"]))
PY
done
