import numpy as np, sys
a = np.loadtxt(sys.argv[1], dtype=np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
a = (a - t0) / 100.0   # 100 MHz -> microseconds
names = ["start", "staged", "d1", "d2", "d4", "d8", "d16(end)"]
print("waves", len(a))
for k, n in enumerate(names):
    c = a[:, k]
    print("%-9s min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f us" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(a, axis=1)
print("segment durations (median / p90 / max) us:")
for k in range(6):
    print("  %-8s -> %-8s %7.2f %7.2f %7.2f" % (names[k], names[k + 1], np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max()))
