"""From gpurun_out/stamps2.txt (GS_VARIANT=140): how much of a wave's epilogue time its SIMD partner (wave + WAVES/2 of the
same workgroup) spends in an epilogue too.  usage: stamps_overlap.py file [cpd] [waves_per_wg]"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.float64)
cpd = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n = (len(a) // W) * W
a = a[:n]
tot = ov = 0.0
lag = []
for b in range(0, n, W):
    for w in range(W // 2):
        x, y = a[b + w], a[b + w + W // 2]
        ex = [(x[2 + 2 * c], x[3 + 2 * c]) for c in range(cpd - 1, 5 * cpd, cpd)]
        ey = [(y[2 + 2 * c], y[3 + 2 * c]) for c in range(cpd - 1, 5 * cpd, cpd)]
        for (s0, e0), (s1, e1) in zip(ex, ey):
            lag.append((s1 - s0) / 100.0)
        for s0, e0 in ex:
            tot += e0 - s0
            for s1, e1 in ey:
                ov += max(0.0, min(e0, e1) - max(s0, s1))
print("pairs", n // 2, "epilogue time overlapped with the partner's epilogue: %.1f %%" % (100 * ov / tot))
lag = np.array(lag)
print("partner epilogue entry lag (us): median %.2f, p10 %.2f, p90 %.2f, mean |lag| %.2f" % (np.median(lag), np.percentile(lag, 10), np.percentile(lag, 90), np.abs(lag).mean()))
