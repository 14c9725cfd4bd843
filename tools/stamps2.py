"""per-chunk stamps (GS_VARIANT=140/141, gpurun_out/stamps2.txt): median duration of every chunk's MFMA steps
and of every epilogue, over the waves' first tasks"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.float64)
cpd = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nch = 5 * cpd
print("waves", len(a))
prev = None
tot_m = tot_e = 0.0
for c in range(nch):
    m = a[:, 2 + 2 * c]
    start = a[:, 3 + 2 * (c - 1)] if (c > 0 and c % cpd == 0) else (a[:, 2 + 2 * (c - 1)] if c > 0 else None)
    if start is not None:
        d = (m - start) / 100.0
        tot_m += np.median(d)
        line = "chunk %2d steps  med %6.2f p90 %6.2f us" % (c, np.median(d), np.percentile(d, 90))
    else:
        line = "chunk %2d steps  (first)" % c
    if (c + 1) % cpd == 0:
        e = (a[:, 3 + 2 * c] - m) / 100.0
        tot_e += np.median(e)
        line += "   epilogue med %6.2f p90 %6.2f us" % (np.median(e), np.percentile(e, 90))
    print(line)
print("sum of medians: steps %.2f us, epilogues %.2f us" % (tot_m, tot_e))

if a.shape[1] >= 54:
    print("inside the epilogues (median us): entry->r1, r1->r2, r2->r3, r3->end")
    for di in range(5):
        c = di * cpd + cpd - 1
        t = [a[:, 34 + di * 4 + r] for r in range(4)] + [a[:, 3 + 2 * c]]
        print("  d%-2d" % (1 << di), " ".join("%5.2f" % np.median((t[k + 1] - t[k]) / 100.0) for k in range(4)),
              "  (steps end -> entry %5.2f)" % np.median((t[0] - a[:, 2 + 2 * c]) / 100.0))
