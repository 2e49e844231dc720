#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: bash bench/ab_lib.sh A.so B.so [reps] [bench.py arguments ...] -> the headline and the phases per run
A=$1; B=$2; N=${3:-3}; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
ARGS=${@:---batch 0 --no-c4 --no-c2-c5 --no-cpu-baseline --steps 100}
for i in $(seq $N); do for L in $A $B; do
  CALIPSO_HIP_LIB=$R/$L python bench.py $ARGS 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=(d['config'].get('roofline_phases') or {}).get('single_system') or {}; b=d['config'].get('batched') or {}
print('$L: value %.1f  ms/step %.3f  solve_and_refine %s  batched %s' % (d['value'], d['ms_per_step'], p.get('solve_and_refine',{}).get('ms'), b.get('newton_steps_per_s')))"
done; done
