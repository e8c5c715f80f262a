"""Summarise an LCB_TRACE_LAUNCHES (+ LCB_TRACE_SEEDS=1) file: where the kernel time of the phase loop goes."""
import collections
import gzip
import sys

import numpy as np

path = sys.argv[1]
op = gzip.open if path.endswith(".gz") else open
launch, seeds, grid = {}, collections.defaultdict(list), {}
for line in op(path, "rt"):
    f = line.rstrip("\n").split("\t")
    if f[0] == "#seed":
        d = dict(x.split("=") for x in f[4:])
        d = {k: int(v) for k, v in d.items()}
        d["vid"] = int(f[3])
        seeds[int(f[1])].append(d)
    else:
        launch[int(f[0])] = (int(f[1]), f[3], float(f[4])); grid[int(f[0])] = int(f[2])
ms = np.array([v[2] for v in launch.values()])
n = np.array([v[0] for v in launch.values()])
mode = np.array([v[1] for v in launch.values()])
print("launches %d  kernel ms %.1f" % (len(ms), ms.sum()))
for md in ("compact", "wide", "big"):
    sel = mode == md
    print("  %-6s launches %5d  ms %8.1f   single-seed launches %5d ms %8.1f" % (md, sel.sum(), ms[sel].sum(), (sel & (n == 1)).sum(), ms[sel & (n == 1)].sum()))
ph = ms[(mode == "compact") & (n > 1)]
if len(ph):
    print("compact launches: %d sum %.1f ms  percentiles 10/50/90/99 = %s" % (len(ph), ph.sum(), np.round(np.percentile(ph, [10, 50, 90, 99]), 3)))
allp = [d for v in seeds.values() for d in v]
if allp:
    tk = np.array([d["ticks"] for d in allp]) / 100.0
    pu = np.array([d["push"] for d in allp])
    vo = np.array([d["vote"] for d in allp])
    print("seeds > 20us: %d   total wave-seconds %.2f   us/push (seeds with >100 pushes): median %.2f" % (len(allp), tk.sum() / 1e6, np.median(tk[pu > 100] / pu[pu > 100])))
    if "tv" in allp[0]:
        sel = [d for d in allp if d["push"] > 50 and d["inst"] < 40]
        tv = sum(d["tv"] for d in sel); tp = sum(d["tp"] for d in sel); ts = sum(d["ts"] for d in sel); tt = sum(d["ticks"] for d in sel)
        npu = sum(d["push"] for d in sel); nvo = sum(d["vote"] for d in sel)
        print("section split over %d mid-size seeds: vote %.1f%% push %.1f%% score+snapshot %.1f%% other %.1f%%  | us/vote %.2f us/push %.2f us/score %.2f" % (
            len(sel), 100.0 * tv / tt, 100.0 * tp / tt, 100.0 * ts / tt, 100.0 * (tt - tv - tp - ts) / tt, tv / 100.0 / nvo, tp / 100.0 / npu, ts / 100.0 / npu))
    big = sorted(allp, key=lambda d: -d["ticks"])[:5]
    for d in big:
        print("   slow seed vid=%d st=%d ticks=%.1f us push=%d vote=%d inst=%d  -> %.1f us/push" % (d["vid"], d["st"], d["ticks"] / 100.0, d["push"], d["vote"], d["inst"], d["ticks"] / 100.0 / max(1, d["push"])))

# ---- per-launch critical path: how much of each launch is its longest seed vs the work of all its seeds spread over the workgroups of its grid
if allp:
    rows = []
    for lid, (nn, md, t) in launch.items():
        sd = seeds.get(lid, [])
        tks = np.array([d["ticks"] for d in sd]) / 1e5 if sd else np.zeros(1)          # ms
        rows.append((nn, t, tks.max(), tks.sum() / max(1, grid.get(lid, 1)), len(sd), md))
    R = np.array([r[:5] for r in rows]); M = np.array([r[5] for r in rows])
    tot = R[:, 1].sum()
    print("critical path: sum of launch ms %.1f | sum of longest-seed ms %.1f (%.0f%%) | sum of (work of the traced seeds / workgroups of the grid) ms %.1f (%.0f%%)" % (
        tot, R[:, 2].sum(), 100 * R[:, 2].sum() / tot, R[:, 3].sum(), 100 * R[:, 3].sum() / tot))
    for md in ("compact", "wide", "big"):
        for lo, hi in ((1, 1), (2, 16), (17, 512), (513, 4096), (4097, 1 << 30)):
            sel = (M == md) & (R[:, 0] >= lo) & (R[:, 0] <= hi)
            if sel.any():
                print("  %-7s launches of %6d..%-10d seeds: %5d launches %9.1f ms | longest seed %9.1f ms | work / grid %9.1f ms | traced (> 20 us) seeds per launch %.0f, mean launch %.3f ms" % (
                    md, lo, hi, sel.sum(), R[sel, 1].sum(), R[sel, 2].sum(), R[sel, 3].sum(), R[sel, 4].mean(), R[sel, 1].mean()))
    pu = np.array([d["push"] for d in allp]); tk = np.array([d["ticks"] for d in allp]) / 100.0
    for lo, hi in ((0, 10), (10, 100), (100, 1000), (1000, 10000), (10000, 1 << 40)):
        s = (pu >= lo) & (pu < hi)
        if s.any():
            print("  seeds with %6d..%-8d pushes: %8d seeds, %9.2f wave-seconds, us/push %.2f" % (lo, hi, s.sum(), tk[s].sum() / 1e6, tk[s].sum() / max(1, pu[s].sum())))
    ppu = np.array([d["ticks"] / 100.0 / d["push"] for d in allp if d["push"] >= 50])
    ins = np.array([d["inst"] for d in allp if d["push"] >= 50])
    for lo, hi in ((0, 8), (8, 32), (32, 64), (64, 128), (128, 256), (256, 100000)):
        s = (ins >= lo) & (ins < hi)
        if s.any():
            print("  us/push for seeds with max inst in [%d,%d): n=%d median %.2f" % (lo, hi, s.sum(), np.median(ppu[s])))
