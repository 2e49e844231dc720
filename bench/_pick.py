# reads bench.py JSON lines from stdin and prints the headline, the roofline entry and the batched figures on one line (used by the sweeps in DESIGN 5.0)
import json, sys
for line in sys.stdin:
    if line.startswith('{"metric'):
        d = json.loads(line); r = d["roofline"]; c = d["config"]
        print("single steps/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "| roofline", (r.get("kernel") or "")[:12], r.get("ms_per_step"), "frac", round(r.get("frac", 0), 4),
              "| batched", (c.get("batched") or {}).get("newton_steps_per_s"), "alone", (c.get("batched") or {}).get("one_group_alone_steps_per_s"), "| group ldl", r.get("group_launch"))
