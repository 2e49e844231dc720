# reads bench.py JSON lines from stdin: the config.c4 block (single / batched rates per configuration)
import json, sys
for l in sys.stdin:
    if l.startswith('{"metric'):
        d = json.loads(l)
        print("single", round(d["value"], 1))
        for k, v in (d["config"].get("c4") or {}).items():
            if isinstance(v, dict):
                print(k, round(v["single_system_steps_per_s"], 1), round(v["batched_newton_steps_per_s"], 1), v.get("ms_per_pass"))
            else:
                print(k, str(v)[:200])
