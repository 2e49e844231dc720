"""One line per bench.py run for the config.c4 block: python bench.py ... | grep '^{' | tail -1 | python bench/c4line.py TAG
-> TAG C4 batched <steps/s> <lane_streams> | C4T batched ... (the A/Bs of profiles/r06_ab_closing.txt, items 7-10)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read())
c = d["config"]["c4"]
print(tag, " | ".join("%s batched %.1f %s" % (k, c[k]["batched_newton_steps_per_s"], c[k]["lane_streams"]) for k in ("C4", "C4T") if k in c))
