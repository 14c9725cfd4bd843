#!/usr/bin/env python3
"""overlap of the kernels of different queues in a rocprofv3 kernel trace CSV"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void gs::") or "gs::" in r["Kernel_Name"]]
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
print("queues:", {q: len(v) for q, v in byq.items()})
qs = sorted(byq, key=lambda q: -len(byq[q]))[:2]
if len(qs) == 2:
    a, b = byq[qs[0]], byq[qs[1]]
    busy_a = sum(e - s for s, e, _ in a)
    busy_b = sum(e - s for s, e, _ in b)
    ov = 0
    for s, e, _ in a:
        for s2, e2, _ in b:
            lo, hi = max(s, s2), min(e, e2)
            if hi > lo:
                ov += hi - lo
    t0 = min(s for s, _, _ in a + b)
    t1 = max(e for _, e, _ in a + b)
    print("span %.3f ms  busy A %.3f  busy B %.3f  overlap %.3f ms" % ((t1 - t0) / 1e6, busy_a / 1e6, busy_b / 1e6, ov / 1e6))
    ev = sorted([(s, "A", n, e - s) for s, e, n in a] + [(s, "B", n, e - s) for s, e, n in b])
    for s, q, n, d in ev[-60:-20]:
        print("%10.1f us %s %-42s %7.1f us" % ((s - t0) / 1e3, q, n, d / 1e3))
