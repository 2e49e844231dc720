# throughput vs stream-priority classes / lanes / HW queues with groups of 8 (C3)
for cfg in "3 24 3 4" "1 24 3 4" "1 32 4 4" "1 32 4 8" "2 16 2 4" "3 48 6 8" "1 48 6 8"; do
  set -- $cfg
  echo "== classes=$1 batch=$2 lanes=$3 hwq=$4"
  GPU_MAX_HW_QUEUES=$4 timeout 300 python bench.py --batch $2 --group 8 --lanes $3 --steps 8 --warmup 2 --no-cpu-baseline --no-single 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.1f  ms/round %.2f' % (d['value'], d['ms_per_step']))
"
done
