# First GPU call of the next round (prepared at the end of round 3, when no GPU time was left):
#  1. the early critical launch (lcb_hooks.early_critical: a stop's own jobs computed while the host plans the rest; expected: the dry-run
#     time, ~1.8 s of a config-3 pass, leaves the critical path) and the device-side commit kernel on the MI355X for the first time: their
#     parity tests, then A/B on config 3 / config 2 / the k = 25 shapes;
#  2. the reference at -t 32 on the k = 25 test-size shapes beside the whole sibeliaz-lcb process (<= 0.5 x);
#  3. the bench line of the default build with the bounded CPU-baseline protocol.
# Before the call (build container): `python sibeliaz_amd/build.py variant nwc4 -DLCB_NW_COMPACT=4` (the compact variant with four wavefronts:
# round 2 measured it on the 62-strain workload only; the k = 25 shapes vote with ~15 voters per vote, 7-8 chunks per wavefront at two)
# and `python sibeliaz_amd/build.py variant ahead -DLCB_PUSH_AHEAD=1` (the compact variant's pushes as a software pipeline: a push waits for no
# global load of its own; exact under the emulator, 155 instead of 143 VGPRs, same occupancy).
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
export LCB_WATCHDOG_S=300
LCB_TEST_EARLY_CRITICAL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "early_critical" > $O/pytest_early_critical.log 2>&1; grep -E "passed|failed" $O/pytest_early_critical.log | tail -2
LCB_TEST_DEVICE_COMMIT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "device_side_commit" > $O/pytest_device_commit.log 2>&1; grep -E "passed|failed" $O/pytest_device_commit.log | tail -2
run() {
  local v=$1; shift
  LCB_VERBOSE=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, launches %s stops %s jobs %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"], c["job_launches"], c["jobs"], c["host_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
# one load of the workload, every variant one pass on the same box (scripts/ab_engine.py); the blocks of all variants must be equal
V="base early:early_critical=1 dc:device_commit=1 both:early_critical=1,device_commit=1 early_prio:early_critical=1,dev.stream_priority=1 \
early_jobs512:early_critical=1,max_jobs=512 early_round1024:early_critical=1,round_phases=1024,dev.batch=262144 base_again"
for w in ecoli62 ecoli10 primates8_test mice16_test; do
  timeout 900 python scripts/ab_engine.py --workload $w $V > $O/ab_$w.txt 2> $O/ab_$w.err; cat $O/ab_$w.txt; tail -3 $O/ab_$w.err
done
if [ -f sibeliaz_amd/libsibeliaz_amd_ahead.so ]; then
for w in ecoli62 primates8_test mice16_test ecoli10; do LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_ahead.so run ahead_$w --workload $w; run stock_$w --workload $w; done
fi
if [ -f sibeliaz_amd/libsibeliaz_amd_nwc4.so ]; then
for w in primates8_test mice16_test ecoli10; do LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_nwc4.so run nwc4_$w --workload $w; done
fi
python - <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import bench
for wl in ("primates8_test", "mice16_test"):
    w = bench.ensure_workload(wl)
    r = bench.run_reference(w, 32, "t32", 600)
    t = time.time()
    p = subprocess.run([os.path.join(bench.BIN, "sibeliaz-lcb"), "--graph", w["graph"], w["fasta"], "-k", str(w["k"]), "-b", str(w["b"]), "-m", str(w["m"]), "-a", str(w["a"]), "-t", "32", "-o", os.path.join(w["dir"], "cli"), "--noseq"], capture_output=True, text=True)
    ours = time.time() - t
    same = isinstance(r, tuple) and bench.md5(r[2]) == bench.md5(os.path.join(w["dir"], "cli", "blocks_coords.gff"))
    print("%s: reference -t 32 whole process %s s (analyze %s s) | sibeliaz-lcb on the MI355X whole process %.1f s rc %d | gff equal %s" % (wl, "%.1f" % r[1] if isinstance(r, tuple) else r, "%.1f" % r[0] if isinstance(r, tuple) else "-", ours, p.returncode, same))
PY
# section timers of the heavy seeds on the current build (instrumented variant): repeated votes, voters and chunks per vote
LCB_VERBOSE=1 LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/prof.json 2> $O/prof.err
python - <<'PY'
rows = []
mode = {}
for line in open("gpurun_out/r4a/trace.tsv"):
    f = line.rstrip("\n").split("\t")
    if f[0] != "#seed":
        mode[int(f[0])] = f[3]; continue
    d = dict(x.split("=") for x in f[4:]); d = {k: int(v) for k, v in d.items()}; d["launch"] = int(f[1]); rows.append(d)
for md in ("compact", "wide", "big"):
    for lo, hi in ((50, 500), (500, 10**9)):
        sel = [d for d in rows if mode.get(d["launch"]) == md and lo <= d["vote"] < hi]
        if not sel: continue
        S = lambda k: sum(d[k] for d in sel)
        nv = S("vote")
        print("%s votes [%d,%d): %d seeds %.1f s | per vote: total %.2f us = walk %.2f + waitB %.2f + reduce %.2f | touch/vote %.1f, wave-0 voters/vote %.2f chunks/vote %.2f, repeated votes (probe) %.3f | per push %.2f us, score %.2f us, pushes/vote %.2f inst %.0f" % (
            md, lo, hi, len(sel), S("ticks") / 1e8, S("tv") / 100.0 / nv, S("cwalk") / 100.0 / nv, S("cwaitb") / 100.0 / nv, S("creduce") / 100.0 / nv,
            S("touch") / nv, S("voters") / nv, S("chunks") / nv, S("probe") / nv, S("tp") / 100.0 / S("push"), S("ts") / 100.0 / S("push"), S("push") / nv, S("inst") / len(sel)))
PY
python scripts/analyze_trace.py $O/trace.tsv > $O/trace_summary.txt 2>&1; head -30 $O/trace_summary.txt
grep -v "^#seed" $O/trace.tsv > $O/launch_trace.tsv; rm -f $O/trace.tsv
