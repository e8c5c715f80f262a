# Round 4, GPU call 3: voter tickets (dynamic deal of voters to the wavefronts of the wide / big variants), coarse page bitmaps in front
# of the engine's range sets (validation, dry runs), LDS page summary in the commit kernel; push section timers (diagnostic build).
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
git rev-parse HEAD > $O/head.txt 2>/dev/null
export LCB_WATCHDOG_S=300
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "resident or variant or footprints or side_lanes or early or screened or overflow" > $O/pytest_gpu_subset.log 2>&1; tail -2 $O/pytest_gpu_subset.log
V="base hostc:host_commit=1 base_again"
for w in ecoli62 primates8_test mice16_test ecoli10; do
  LCB_VERBOSE=1 timeout 900 python scripts/ab_engine.py --workload $w $V > $O/ab_$w.txt 2> $O/ab_$w.err; cat $O/ab_$w.txt; grep -E "lcb engine" $O/ab_$w.err | head -4
done
run() {
  local v=$1; shift
  timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel busy %.1f ms (sum %.1f, side %.1f), launches %s stops %s jobs %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["kernel_ms_sum_over_streams_per_step"], d["roofline"]["kernel_ms_on_side_lanes_per_step"], d["roofline"]["launches_per_step"], c["job_launches"], c["jobs"], c["host_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
for w in ecoli62 primates8_test; do LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_notickets.so run notickets_$w --workload $w; run stock_$w --workload $w; done
# section timers: votes (stock instrumented variant) and pushes (diagnostic build)
prof() {
  local tag=$1; shift
  LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace_$tag.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/prof_$tag.json 2> $O/prof_$tag.err
  python - <<PY
rows = []; mode = {}
for line in open("$O/trace_$tag.tsv"):
    f = line.rstrip("\n").split("\t")
    if f[0] != "#seed":
        mode[int(f[0])] = f[3]; continue
    d = dict(x.split("=") for x in f[4:]); d = {k: int(v) for k, v in d.items()}; d["launch"] = int(f[1]); rows.append(d)
push = "$tag".startswith("push")
for md in ("compact", "wide", "big"):
    for lo, hi in ((50, 500), (500, 10**9)):
        sel = [d for d in rows if mode.get(d["launch"]) == md and lo <= d["vote"] < hi]
        if not sel: continue
        S = lambda k: sum(d[k] for d in sel)
        nv, npu = S("vote"), max(1, S("push"))
        if push:
            print("$tag %s votes [%d,%d): %d seeds | per push %.2f us = until in path set %.2f + search/classify %.2f + cross-lane/apply %.2f + index merge %.2f + rest %.2f | score %.2f us | per vote %.2f us" % (
                md, lo, hi, len(sel), S("tp") / 100.0 / npu, S("cwalk") / 100.0 / npu, S("cwaitb") / 100.0 / npu, S("creduce") / 100.0 / npu, S("cscan") / 100.0 / npu,
                (S("tp") - S("cwalk") - S("cwaitb") - S("creduce") - S("cscan")) / 100.0 / npu, S("ts") / 100.0 / npu, S("tv") / 100.0 / nv))
        else:
            print("$tag %s votes [%d,%d): %d seeds %.1f s | per vote: total %.2f us = walk %.2f + waitB %.2f + reduce %.2f | touch/vote %.1f, wave-0 voters/vote %.2f chunks/vote %.2f | per push %.2f us, score %.2f us, pushes/vote %.2f" % (
                md, lo, hi, len(sel), S("ticks") / 1e8, S("tv") / 100.0 / nv, S("cwalk") / 100.0 / nv, S("cwaitb") / 100.0 / nv, S("creduce") / 100.0 / nv,
                S("touch") / nv, S("voters") / nv, S("chunks") / nv, S("tp") / 100.0 / npu, S("ts") / 100.0 / npu, S("push") / nv))
PY
  grep -v "^#seed" $O/trace_$tag.tsv > $O/launch_trace_$tag.tsv; rm -f $O/trace_$tag.tsv
}
prof vote_ecoli62 --workload ecoli62
LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_profpush.so prof push_ecoli62 --workload ecoli62
LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_profpush.so prof push_mice16 --workload mice16_test
