#!/bin/bash
# what bounds k_solve_tail (vectors.hip)?  Rebuilds it with -DTAIL_EXP=0..3 (0: as shipped, 1: no constraint code, 2: no constraint code + two workgroups per row group,
# 3: the constraint code without the mat-vec's loads) and reads its average duration from a kernel trace of the single-system bench region.  Variants 1..3 compute
# garbage: timings only.  GPU box; the plain build is restored.
R=${GRAFT_REPO_ROOT:-$(pwd)}
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result"
cd /tmp && export TMPDIR=/tmp
for e in ${1:-0 1 2 3}; do
  (cd $R/calipso.jl_amd/csrc && hipcc $FL -DTAIL_EXP=$e -c vectors.hip -o vectors.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcalipso_hip.so *.o -ldl)
  rm -rf /tmp/tp_$e
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp_$e -- python $R/bench.py --batch 0 --steps 10 --warmup 2 --no-cpu-baseline --no-c4 --no-c2-c5 > /tmp/tp_$e.json 2> /tmp/tp_$e.err < /dev/null
  f=$(find /tmp/tp_$e -name "*kernel_stats.csv" | head -1)
  echo "TAIL_EXP=$e: $(grep k_solve_tail $f | awk -F, '{gsub(/"/,""); print "calls " $(NF-6) " avg ns " $(NF-4) " min " $(NF-2) " max " $(NF-1)}')"
done
(cd $R/calipso.jl_amd/csrc && hipcc $FL -c vectors.hip -o vectors.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcalipso_hip.so *.o -ldl)
