#!/usr/bin/env python3
"""Timeline of gs_espnet_segment_host from a rocprofv3 --kernel-trace --memory-copy-trace run of host_pipe_trace.py:
per batch, the upload, the compute span (stem .. dec_tail) and the download; gaps between consecutive computes."""
import csv, glob, sys
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
kt.sort(key=lambda r: int(r["Start_Timestamp"]))
stems = [r for r in kt if "stem_kernel" in r["Kernel_Name"]]
tails = [r for r in kt if "dec_tail_kernel<5, 8" in r["Kernel_Name"]]
t0 = int(stems[0]["Start_Timestamp"])
us = lambda t: (int(t) - t0) / 1e3
print("batches", len(stems), len(tails))
h2d = sorted([r for r in mc if "HOST_TO_DEVICE" in r["Direction"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000], key=lambda r: int(r["Start_Timestamp"]))
d2h = sorted([r for r in mc if "DEVICE_TO_HOST" in r["Direction"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000], key=lambda r: int(r["Start_Timestamp"]))
print("big h2d", len(h2d), "big d2h", len(d2h))
n = min(len(stems), len(tails))
prev_end = None
for i in range(n):
    s, e = us(stems[i]["Start_Timestamp"]), us(tails[i]["End_Timestamp"])
    line = "batch %2d compute %9.1f .. %9.1f (%7.1f)" % (i, s, e, e - s)
    if i < len(h2d):
        line += "  h2d %9.1f .. %9.1f (%6.1f)" % (us(h2d[i]["Start_Timestamp"]), us(h2d[i]["End_Timestamp"]), us(h2d[i]["End_Timestamp"]) - us(h2d[i]["Start_Timestamp"]))
    if i < len(d2h):
        line += "  d2h %9.1f .. %9.1f (%6.1f)" % (us(d2h[i]["Start_Timestamp"]), us(d2h[i]["End_Timestamp"]), us(d2h[i]["End_Timestamp"]) - us(d2h[i]["Start_Timestamp"]))
    print(line)
# busy time of the GPU's compute: union of kernel intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kt if int(r["Start_Timestamp"]) >= t0)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
busy += cur_e - cur_s
span = max(b for a, b in iv) - t0
print("kernel-busy %.1f ms of %.1f ms span (%.1f %%)" % (busy / 1e6, span / 1e6, 100.0 * busy / span))
print("sum of kernel durations %.1f ms" % (sum(b - a for a, b in iv) / 1e6))
