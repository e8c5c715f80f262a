# Round 3, GPU call 3: section timers inside the vote of the heavy seeds (instrumented variant, LCB_TRACE_SEEDS=1)
mkdir -p gpurun_out/r3e3
O=gpurun_out/r3e3
export LCB_WATCHDOG_S=300
LCB_VERBOSE=1 LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/prof.json 2> $O/prof.err
python - <<'PY'
import collections
rows = []
mode = {}
for line in open("gpurun_out/r3e3/trace.tsv"):
    f = line.rstrip("\n").split("\t")
    if f[0] != "#seed":
        mode[int(f[0])] = f[3]; continue
    d = dict(x.split("=") for x in f[4:]); d = {k: int(v) for k, v in d.items()}; d["launch"] = int(f[1]); rows.append(d)
for md in ("compact", "wide", "big"):
    sel = [d for d in rows if mode.get(d["launch"]) == md and d["vote"] >= 500]
    if not sel: continue
    S = lambda k: sum(d[k] for d in sel)
    nv = S("vote")
    print("%s: %d seeds with >= 500 votes | per vote: total %.2f us = walk %.2f + waitB %.2f + reduce %.2f (scan inside walk %.2f) | touch/vote %.1f, wave-0 voters/vote %.2f chunks/vote %.2f | per push %.2f us, score %.2f us" % (
        md, len(sel), S("tv") / 100.0 / nv, S("cwalk") / 100.0 / nv, S("cwaitb") / 100.0 / nv, S("creduce") / 100.0 / nv, S("cscan") / 100.0 / nv,
        S("touch") / nv, S("voters") / nv, S("chunks") / nv, S("tp") / 100.0 / S("push"), S("ts") / 100.0 / S("push")))
PY
gzip -f $O/trace.tsv
