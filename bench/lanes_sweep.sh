for cfg in "3 6 3 4" "3 6 3 8" "1 8 4 8" "1 8 8 8" "3 12 6 8" "2 8 4 8" "1 6 3 4"; do
  set -- $cfg
  echo "== classes=$1 batch=$2 lanes=$3 hwq=$4"
  CALIPSO_HIP_PRIORITY_CLASSES=$1 GPU_MAX_HW_QUEUES=$4 timeout 200 python bench.py --batch $2 --lanes $3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.1f single %.1f' % (d['value'], d['config']['single_instance_steps_per_s']))
"
done
