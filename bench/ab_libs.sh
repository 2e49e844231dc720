#!/bin/bash
# A/B/... of several builds of the library on ONE box, alternating: bash bench/ab_libs.sh REPS "bench.py arguments" A.so B.so ... -> the headline, solve + refinement and the batched rate per run
N=$1; ARGS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in $(seq $N); do for L in "$@"; do
  CALIPSO_HIP_LIB=$R/$L python bench.py $ARGS 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=(d['config'].get('roofline_phases') or {}).get('single_system') or {}; b=d['config'].get('batched') or {}
print('$L: value %.1f  ms/step %.3f  solve_and_refine %s  batched %s  one group alone %s' % (d['value'], d['ms_per_step'], p.get('solve_and_refine',{}).get('ms'), b.get('newton_steps_per_s'), b.get('one_group_alone_steps_per_s')))"
done; done
