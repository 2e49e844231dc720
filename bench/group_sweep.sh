# throughput vs group size / lanes (C3):  bash bench/group_sweep.sh [B:G:L ...]   (python bench.py --batch B --group G --lanes L)
[ $# -eq 0 ] && set -- 24:8:3 30:10:3 36:12:3 48:16:3 16:8:2 32:16:2
for cfg in "$@"; do
  IFS=: read B G L <<< "$cfg"
  echo "== batch=$B group=$G lanes=$L"
  timeout 600 python bench.py --batch $B --group $G --lanes $L --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-single --config ${CONFIG:-C3} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('value %.1f steps/s  ms/round %.2f  unit alone %.1f' % (d['value'], d['ms_per_step'], c['one_unit_alone_steps_per_s']))
    elif 'Error' in l or 'error' in l: print(l.strip()[:300])
"
done
