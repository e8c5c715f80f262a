#!/usr/bin/env python3
"""Per-vote / per-push section times from an LCB_TRACE_LAUNCHES + LCB_TRACE_SEEDS=1 trace (the instrumented kernel instantiation):

    LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=trace.tsv python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline
    python scripts/vote_sections.py trace.tsv <tag> [push]

`push`: the trace is of a library built with -DLCB_PROF_PUSH=1 (the vote-section slots carry the sections of a push)."""
import sys

path, tag = sys.argv[1], sys.argv[2]
push = len(sys.argv) > 3 and sys.argv[3] == "push"
rows, mode = [], {}
for line in open(path):
    f = line.rstrip("\n").split("\t")
    if f[0] != "#seed":
        mode[int(f[0])] = f[3]
        continue
    d = dict(x.split("=") for x in f[4:])
    d = {k: int(v) for k, v in d.items()}
    d["launch"] = int(f[1])
    rows.append(d)
for md in ("compact", "wide", "big"):
    for lo, hi in ((50, 500), (500, 10**9)):
        sel = [d for d in rows if mode.get(d["launch"]) == md and lo <= d["vote"] < hi]
        if not sel:
            continue
        S = lambda k: sum(d[k] for d in sel)
        nv, npu = S("vote"), max(1, S("push"))
        if push:
            print("%s %s votes [%d,%d): %d seeds | per push %.2f us = until in path set %.2f + search/classify %.2f + cross-lane/apply %.2f + index merge %.2f + rest %.2f | score %.2f us | per vote %.2f us" % (
                tag, md, lo, hi, len(sel), S("tp") / 100.0 / npu, S("cwalk") / 100.0 / npu, S("cwaitb") / 100.0 / npu, S("creduce") / 100.0 / npu, S("cscan") / 100.0 / npu,
                (S("tp") - S("cwalk") - S("cwaitb") - S("creduce") - S("cscan")) / 100.0 / npu, S("ts") / 100.0 / npu, S("tv") / 100.0 / nv))
        else:
            print("%s %s votes [%d,%d): %d seeds %.1f s | per vote: total %.2f us = walk %.2f + waitB %.2f + reduce %.2f | touch/vote %.1f, wave-0 voters/vote %.2f chunks/vote %.2f | per push %.2f us, score %.2f us, pushes/vote %.2f" % (
                tag, md, lo, hi, len(sel), S("ticks") / 1e8, S("tv") / 100.0 / nv, S("cwalk") / 100.0 / nv, S("cwaitb") / 100.0 / nv, S("creduce") / 100.0 / nv,
                S("touch") / nv, S("voters") / nv, S("chunks") / nv, S("tp") / 100.0 / npu, S("ts") / 100.0 / npu, S("push") / nv))
