#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass, counter_collection.csv) into per-kernel HBM bytes per launch.

usage: pmc_summary.py FETCH_SIZE.csv WRITE_SIZE.csv GRID_Z out.json [COUNTER:file.csv ...]
(extra COUNTER passes, e.g. MfmaUtil, SQ_LDS_BANK_CONFLICT, are averaged per kernel and added under that name)
Only dispatches whose Grid_Size matches the group launch (k_schur: grid.z = GRID_Z instances) are averaged for k_schur; the other
kernels are averaged over all their dispatches (k_ldl_step: over the NP / 64 - 1 panel steps of a factorisation, early and late ones alike —
the same average bench.py's roofline uses).  FETCH_SIZE is doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950
rocprofv3 reports half of a wide streaming read); both counters are in KiB."""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            if name.startswith("void "):          # template instantiations are printed with their return type
                name = name[5:]
            rows[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return rows


def main():
    fetch, write, gz, out = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("calipso::"):
            continue
        f, w = fetch.get(k, []), write.get(k, [])
        if k == "calipso::k_schur":            # keep the launches that carry gz instances (largest grid)
            gmax = max(g for g, _ in f + w)
            f = [x for x in f if x[0] == gmax]
            w = [x for x in w if x[0] == gmax]
        e = {"launches_sampled": len(f),
             "FETCH_SIZE_KiB_per_launch": sum(v for _, v in f) / max(1, len(f)),
             "WRITE_SIZE_KiB_per_launch": sum(v for _, v in w) / max(1, len(w))}
        e["instances_per_launch"] = gz
        e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KiB_per_launch"] + e["WRITE_SIZE_KiB_per_launch"]) * 1024.0
        if k == "calipso::k_schur":
            e["note"] = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 rocprofv3 reports half of a wide streaming read; 8 B/lane "
                         "loads are uncalibrated, so this is an upper bound); WRITE_SIZE as reported")
        res[k] = e
    for spec in sys.argv[5:]:
        name, path = spec.split(":", 1)
        try:
            rows = load(path)
        except OSError:
            continue
        for k, v in rows.items():
            if k in res and v:
                if k == "calipso::k_schur":
                    gmax = max(g for g, _ in v)
                    v = [x for x in v if x[0] == gmax]
                res[k][name + "_per_launch"] = sum(x for _, x in v) / len(v)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("calipso::k_schur", {})))


if __name__ == "__main__":
    main()
