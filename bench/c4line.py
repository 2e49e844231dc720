import json,sys
tag=sys.argv[1]
d=json.loads(sys.stdin.read()); c=d["config"]["c4"]
print(tag, " | ".join("%s batched %.1f %s" % (k, c[k]["batched_newton_steps_per_s"], c[k]["lane_streams"]) for k in ("C4","C4T") if k in c))
