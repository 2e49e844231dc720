#!/bin/bash
# the headline step with and without the other instances of the default bench in the process: bash bench/headline_context.sh
cd "$(dirname "$0")/.."
for args in "--batch 0 --no-c4 --no-c2-c5" "--batch 36 --no-c4 --no-c2-c5 --batched-passes 2" "--batch 0 --no-c4 --no-c2-c5" "--batch 36 --no-c4 --no-c2-c5 --batched-passes 2"; do
  for q in "" 8; do
    GPU_MAX_HW_QUEUES=$q python bench.py $args --no-cpu-baseline --steps 30 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['config']['roofline_phases']['single_system']
print('args [$args] GPU_MAX_HW_QUEUES=[$q]: value %.1f  wall %.3f ms  events: step %.3f factor %.3f (ldl %.3f) solve %.3f' % (d['value'], d['ms_per_step'], p['whole_step_ms'], p['factor']['ms'], p['factor']['ldl_ms'], p['solve_and_refine']['ms']))"
  done
done
