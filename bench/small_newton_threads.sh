#!/bin/bash
# the batched small-problem kernel by threads per instance (options.threads = 64 / 128 / 256, and 0 = the library's choice: csrc/smallnewton.hip, sn_threads) over a range of
# shapes: one line per (shape, threads) with the residency the runtime reports.   bash bench/small_newton_threads.sh > profiles/r0N_small_newton_threads.txt   (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for sh in "10 4 6 32768" "20 8 12 16384" "30 10 20 16384" "49 40 0 8192" "49 40 20 4096" "60 40 30 4096" "100 60 40 2048"; do
  for nt in 64 128 256 0; do
    SN_THREADS=$nt timeout 300 python bench/small_newton_rate.py $sh 20 2>/dev/null | tail -1 | python3 -c "
import sys, json
o = json.loads(sys.stdin.readline()); k = o['kernel']
print('shape (%3d, %2d, %2d) x %5d  threads %-4s -> %3d threads, %6d B LDS, %d instances per compute unit: %9.0f solve!s/s  %10.0f Newton steps/s (benchmark steps)  %6.1f us per step of a resident instance  converged %d' % (
    *o['shape'], o['batch'], o['threads'], k['threads_per_instance'], k['lds_bytes_per_instance'], k['instances_per_compute_unit'], o['solve']['solves_per_s'], o['steps']['newton_steps_per_s'],
    o['steps']['us_per_step_of_a_resident_instance'], o['solve']['converged']))"
  done
done
