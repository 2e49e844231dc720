#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv without what is not a Newton step: the runtime's copy / fill kernels of the handles' set-up (uploads of the problem data, memsets at
create) and bench.py's own fp64 matrix-peak probe; percentages recomputed.  python bench/steps_only_stats.py in.csv out.csv"""
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))
head, body = rows[0], rows[1:]
iname, itot, ipct = head.index("Name"), head.index("TotalDurationNs"), head.index("Percentage")
drop = ("__amd_rocclr_", "k_mfma_f64_peak")
kept = [r for r in body if not any(d in r[iname] for d in drop)]
gone = [r for r in body if any(d in r[iname] for d in drop)]
total = sum(float(r[itot]) for r in kept) or 1.0
for r in kept:
    r[ipct] = "%.4f" % (100.0 * float(r[itot]) / total)
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(head)
    w.writerows(kept)
print("kept %d kernels (%.3f ms), dropped %d set-up / probe kernels (%.3f ms)" % (len(kept), total * 1e-6, len(gone), sum(float(r[itot]) for r in gone) * 1e-6))
