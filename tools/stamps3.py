#!/usr/bin/env python3
"""Where one launch of a conv_mfma_kernel spends its time, from the per-chunk stamps of EVERY task of every wave
(-DGS_DIAG build, GS_VARIANT=160..165 -> gpurun_out/stamps_*.txt; F_X_STAMP2 in csrc/conv_mfma.h).

    python tools/stamps3.py gpurun_out/stamps_l3esp.txt NCHUNK CPD [mfma_cycles_per_chunk] [clock_ghz]

A wave alternates between k-steps (matrix instructions of a chunk) and epilogues (after the last chunk of a dilation).  The two
waves that share a SIMD are (block, wid) and (block, wid + 4).  For every SIMD the launch interval [first start, last end] is cut
into the states below; the table is the average over SIMDs and sums to the launch time."""
import sys

import numpy as np


def main():
    path, nchunk, cpd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    ideal_cycles = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0   # matrix-pipe cycles of one chunk's k-steps of ONE wave
    ghz = float(sys.argv[5]) if len(sys.argv) > 5 else 2.4
    raw = np.loadtxt(path, dtype=np.float64)
    wave_id = raw[:, 0].astype(int)
    st = raw[:, 1:]
    t0 = st[:, 0].min()
    us = lambda x: (x - t0) / 100.0   # 100 MHz -> microseconds after the first wave's start
    per_task = 2 * nchunk
    waves = {}
    for w, row in zip(wave_id, st):
        segs = []   # (t_begin, t_end, kind) kind 'k' = k-steps, 'e' = epilogue
        t = row[1]   # staged
        ntask = 0
        for ti in range((len(row) - 2) // per_task):
            base = 2 + ti * per_task
            if row[base] == 0 and row[base + 2 * (nchunk - 1)] == 0:
                break
            ntask += 1
            for c in range(nchunk):
                tk = row[base + 2 * c]
                if tk == 0:   # (cannot happen for a stamped task)
                    continue
                segs.append((us(t), us(tk), 'k', c))
                t = tk
                if (c + 1) % cpd == 0:
                    te = row[base + 2 * c + 1]
                    if te:
                        segs.append((us(t), us(te), 'e', c))
                        t = te
        waves[w] = dict(start=us(row[0]), staged=us(row[1]), end=us(t), segs=segs, ntask=ntask)
    L = max(v['end'] for v in waves.values())
    print("waves %d   tasks per wave %s   launch (first start -> last end) %.2f us" % (
        len(waves), sorted(set(v['ntask'] for v in waves.values())), L))
    a = np.array([[v['start'], v['staged'], v['end']] for v in waves.values()])
    for k, n in enumerate(["start", "staged", "end"]):
        c = a[:, k]
        print("  %-7s min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f us" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))

    # chunk / epilogue durations
    kd = np.array([s[1] - s[0] for v in waves.values() for s in v['segs'] if s[2] == 'k' and s[1] > s[0]])
    ed = np.array([s[1] - s[0] for v in waves.values() for s in v['segs'] if s[2] == 'e'])
    print("k-steps of a chunk: median %.2f us (p10 %.2f, p90 %.2f), sum per wave %.1f us;  epilogue: median %.2f us (p90 %.2f), sum per wave %.1f us" % (
        np.median(kd), np.percentile(kd, 10), np.percentile(kd, 90), kd.sum() / len(waves), np.median(ed), np.percentile(ed, 90), ed.sum() / len(waves)))

    # per-SIMD state partition
    names = ["ramp (launch, staging, first operands)", "both waves in k-steps", "one in k-steps, partner in its epilogue",
             "one in k-steps, partner finished / not started", "no wave in k-steps (epilogues only)", "SIMD finished, launch still running"]
    tot = np.zeros(len(names))
    nsimd = 0
    lone_k, pair_k = [], []   # chunk durations by partner state at the chunk's midpoint
    for w, v in waves.items():
        blk, wid = divmod(w, 8)
        if wid >= 4:
            continue
        p = waves.get(blk * 8 + wid + 4)
        pair = [v] + ([p] if p else [])
        nsimd += 1
        ev = set([0.0, L])
        for x in pair:
            ev.add(x['staged'])
            ev.add(x['end'])
            for s in x['segs']:
                ev.add(s[0])
                ev.add(s[1])
        ev = sorted(ev)

        def state(x, t):
            if t < x['staged']:
                return 'n'
            if t >= x['end']:
                return 'f'
            for s in x['segs']:
                if s[0] <= t < s[1]:
                    return s[2]
            return 'e'
        first_staged = min(x['staged'] for x in pair)
        last_end = max(x['end'] for x in pair)
        for t1, t2 in zip(ev[:-1], ev[1:]):
            mid = 0.5 * (t1 + t2)
            d = t2 - t1
            if mid < first_staged:
                tot[0] += d
                continue
            if mid >= last_end:
                tot[5] += d
                continue
            ss = [state(x, mid) for x in pair]
            nk = ss.count('k')
            if nk == 2:
                tot[1] += d
            elif nk == 1:
                other = [q for q in ss if q != 'k']
                if not other or other[0] in 'nf':
                    tot[3] += d
                else:
                    tot[2] += d
            else:
                tot[4] += d
        if p:
            for x, y in ((v, p), (p, v)):
                for s in x['segs']:
                    if s[2] != 'k' or s[1] <= s[0]:
                        continue
                    q = state(y, 0.5 * (s[0] + s[1]))
                    (pair_k if q == 'k' else lone_k).append(s[1] - s[0])
    tot /= nsimd
    print("\nstate of a SIMD over the launch (average of %d SIMDs):" % nsimd)
    for n, t in zip(names, tot):
        print("  %-52s %7.2f us  %5.1f %%" % (n, t, 100.0 * t / L))
    print("  %-52s %7.2f us" % ("sum", tot.sum()))
    if pair_k and lone_k:
        print("\nchunk k-steps with the partner also in k-steps: median %.2f us (%d chunks); partner elsewhere: %.2f us (%d chunks)" % (
            np.median(pair_k), len(pair_k), np.median(lone_k), len(lone_k)))
    if ideal_cycles:
        ideal = ideal_cycles / (ghz * 1000.0)
        print("matrix-pipe time of one chunk of one wave: %.2f us -> pipe share of a chunk run in a pair %.2f (two waves: %.2f), of a lone chunk %.2f" % (
            ideal, ideal / np.median(pair_k), 2 * ideal / np.median(pair_k), ideal / np.median(lone_k)))


if __name__ == "__main__":
    main()
