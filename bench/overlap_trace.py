"""How much do the kernels of different groups overlap on the device?  python bench/overlap_trace.py <kernel_trace.csv>: reads rocprofv3's kernel trace of a batched
bench run (several groups on several host lanes), classifies the kernels (matrix-core bound: k_schur, k_ldl_step, merges, products; the rest: bandwidth / latency bound)
and prints, over the busy part of the trace: time with >= 1 kernel running, time with kernels of >= 2 queues running, time with a matrix-core kernel AND another
kernel running, and the per-class busy times."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
for r in rows:
    n = r[name_key]
    if "mfma_f64_peak" in n or "rocclr" in n:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "0")
    mf = any(k in n for k in ("k_schur", "k_ldl_step", "k_tinv_merge", "k_wform_product", "k_ldl_diag", "k_ldl_scale"))
    ev.append((s, e, q, mf, n))
ev.sort()
# keep the last 60 % of the trace (the timed passes, not creation / warm-up)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + 0.4 * (t1 - t0)
ev = [e for e in ev if e[0] >= cut]
pts = []
for s, e, q, mf, n in ev:
    pts.append((s, 1, q, mf)); pts.append((e, -1, q, mf))
pts.sort()
active = {}          # (queue, mf) -> count
busy = two = mix = mfb = otb = 0
last = pts[0][0]
for t, d, q, mf in pts:
    dt = t - last
    if dt > 0:
        qs = {k[0] for k, v in active.items() if v > 0}
        anymf = any(v > 0 and k[1] for k, v in active.items())
        anyot = any(v > 0 and not k[1] for k, v in active.items())
        if qs:
            busy += dt
        if len(qs) >= 2:
            two += dt
        if anymf and anyot:
            mix += dt
        if anymf:
            mfb += dt
        if anyot:
            otb += dt
    active[(q, mf)] = active.get((q, mf), 0) + d
    last = t
span = pts[-1][0] - pts[0][0]
print("span %.1f ms; >= 1 kernel running %.1f ms; kernels of >= 2 queues running %.1f ms; a matrix-core kernel AND another kind running %.1f ms" % (span / 1e6, busy / 1e6, two / 1e6, mix / 1e6))
print("matrix-core kernels running %.1f ms, other kernels running %.1f ms (sum %.1f ms)" % (mfb / 1e6, otb / 1e6, (mfb + otb) / 1e6))
sums = {}
for s, e, q, mf, n in ev:
    k = n.split("(")[0][:60]
    sums[k] = sums.get(k, 0) + (e - s)
for k, v in sorted(sums.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-62s %.1f ms (sum of durations)" % (k, v / 1e6))
