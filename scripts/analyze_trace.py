"""Summarise an LCB_TRACE_LAUNCHES (+ LCB_TRACE_SEEDS=1) file: where the kernel time of the phase loop goes."""
import collections
import gzip
import sys

import numpy as np

path = sys.argv[1]
op = gzip.open if path.endswith(".gz") else open
launch, seeds = {}, collections.defaultdict(list)
for line in op(path, "rt"):
    f = line.rstrip("\n").split("\t")
    if f[0] == "#seed":
        d = dict(x.split("=") for x in f[4:])
        d = {k: int(v) for k, v in d.items()}
        d["vid"] = int(f[3])
        seeds[int(f[1])].append(d)
    else:
        launch[int(f[0])] = (int(f[1]), f[3], float(f[4]))
ms = np.array([v[2] for v in launch.values()])
n = np.array([v[0] for v in launch.values()])
mode = np.array([v[1] for v in launch.values()])
print("launches %d  kernel ms %.1f" % (len(ms), ms.sum()))
for md in ("small", "medium", "big"):
    sel = mode == md
    print("  %-6s launches %5d  ms %8.1f   single-seed launches %5d ms %8.1f" % (md, sel.sum(), ms[sel].sum(), (sel & (n == 1)).sum(), ms[sel & (n == 1)].sum()))
ph = ms[(mode == "small") & (n > 1)]
print("phase launches: %d sum %.1f ms  percentiles 10/50/90/99 = %s" % (len(ph), ph.sum(), np.round(np.percentile(ph, [10, 50, 90, 99]), 3)))
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
