# reads bench.py JSON lines from stdin: ms per step and the phase split of the single system (LDL^T, Schur complement, solve + refinement)
import json, sys
for line in sys.stdin:
    if line.startswith('{"metric'):
        d = json.loads(line)
        def find(o, key):
            if isinstance(o, dict):
                if key in o: return o[key]
                for v in o.values():
                    r = find(v, key)
                    if r is not None: return r
            return None
        ss = find(d, "single_system")
        print(round(d["ms_per_step"], 3), "ldl", round(ss["factor"]["ldl_ms"], 3), "schur", round(ss["factor"]["schur_ms"], 3), "solve+refine", round(ss["solve_and_refine"]["ms"], 3), "launches", ss.get("launches_per_step"))
