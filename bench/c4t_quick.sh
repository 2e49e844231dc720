#!/bin/bash
# C4T at the two group sizes the record quotes (32 = config 4's per-GPU share, 256 = all of it on one GPU) + the phases of a front: bash bench/c4t_quick.sh -> gpurun_out/c4t_quick.txt
cd "$(dirname "$0")/.."
O=gpurun_out/c4t_quick.txt; : > $O
for cfg in "32 16 2" "256 64 4"; do
  set -- $cfg
  for rep in 1 2; do
  timeout 600 python bench.py --config C4T --batch $1 --group $2 --lanes $3 --no-cpu-baseline --no-c4 --no-c2-c5 --no-single --batched-passes 10 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['config']['batched']
print('C4T instances %4d  group %4d  lanes %d : %8.0f steps/s  (%.3f ms per pass; one group alone %8.0f steps/s)' % (b['instances_per_gpu'], b['instances_per_group'], b['groups_in_flight'], b['newton_steps_per_s'], b['ms_per_pass'], b['one_group_alone_steps_per_s']))" | tee -a $O
  done
done
[ -f calipso.jl_amd/libcalipso_hip_trace.so ] && for g in 1 16; do timeout 300 python bench/mf_trace_group.py $g 2>/dev/null | tee -a $O; done
