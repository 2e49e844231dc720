#!/bin/bash
# C4T (structured handles) batched rate against (instances, group size, groups in flight): bash bench/c4t_group_sweep.sh  -> gpurun_out/c4t_group_sweep.txt
cd "$(dirname "$0")/.."
O=gpurun_out/c4t_group_sweep.txt; : > $O
for cfg in "32 16 2" "32 32 1" "64 32 2" "96 32 3" "128 64 2" "192 64 3" "256 128 2" "256 64 4" "384 128 3"; do
  set -- $cfg
  timeout 600 python bench.py --config C4T --batch $1 --group $2 --lanes $3 --no-cpu-baseline --no-c4 --no-c2-c5 --no-single --batched-passes 10 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['config']['batched']
print('C4T instances %4d  group %4d  lanes %d : %8.0f steps/s  (%.3f ms per pass; one group alone %8.0f steps/s)' % (b['instances_per_gpu'], b['instances_per_group'], b['groups_in_flight'], b['newton_steps_per_s'], b['ms_per_pass'], b['one_group_alone_steps_per_s']))" | tee -a $O
done
