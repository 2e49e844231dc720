# throughput vs group size / lanes (C3): python bench.py --batch B --group G --lanes L
for cfg in "6 1 3" "2 2 1" "4 4 1" "8 8 1" "16 16 1" "12 4 3" "24 8 3" "12 6 2" "16 8 2" "48 16 3"; do
  set -- $cfg
  echo "== batch=$1 group=$2 lanes=$3"
  timeout 600 python bench.py --batch $1 --group $2 --lanes $3 --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --config ${CONFIG:-C3} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('value %.1f steps/s  ms/round %.2f  single %.1f  schur_ms(concurrent) %.3f' % (d['value'], d['ms_per_step'], c['single_instance_steps_per_s'], list(d['roofline'].values())[-1]))
    elif 'Error' in l or 'error' in l: print(l.strip()[:300])
"
done
