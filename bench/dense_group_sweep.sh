#!/bin/bash
# dense C3 / C4 batched rate against (instances, group size, groups in flight): bash bench/dense_group_sweep.sh [C3|C4] -> gpurun_out/dense_group_sweep_CFG.txt
cd "$(dirname "$0")/.."
CFG=${1:-C3}
O=gpurun_out/dense_group_sweep_$CFG.txt; : > $O
for cfg in "36 12 3" "36 18 2" "36 36 1" "48 24 2" "48 16 3" "64 32 2" "72 24 3" "96 48 2"; do
  set -- $cfg
  timeout 900 python bench.py --config $CFG --batch $1 --group $2 --lanes $3 --no-cpu-baseline --no-c4 --no-c2-c5 --no-single --batched-passes 6 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['config']['batched']; p=d['config']['roofline_phases'].get('one_group_alone',{})
print('$CFG instances %4d  group %4d  lanes %d : %8.0f steps/s  (%.2f ms per pass; one group alone %8.0f steps/s: factor %.2f solve %.2f ms)' % (b['instances_per_gpu'], b['instances_per_group'], b['groups_in_flight'], b['newton_steps_per_s'], b['ms_per_pass'], b['one_group_alone_steps_per_s'], p.get('factor',{}).get('ms',0), p.get('solve_and_refine',{}).get('ms',0)))" | tee -a $O
done
