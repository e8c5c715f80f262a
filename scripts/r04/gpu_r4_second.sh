# Round 4, GPU call 2: the build without scratch memory (LcbOcc shifts, -simplifycfg-sink-common=false), early critical launch + stream
# priorities on by default, device-resident commit (lcb_commit_kernel chained behind every launch of a round) on by default.
#  1. the whole parity suite (every test runs: no opt-in tests left);
#  2. one-load A/Bs: default vs host_commit vs sync_jobs on config 3 and the k = 25 shapes, with the engine's section timers (LCB_VERBOSE);
#  3. what the overflow ladder costs: seeds that run in one variant, overflow and run again in the next (instrumented variant).
mkdir -p gpurun_out/r4b
O=gpurun_out/r4b
git rev-parse HEAD > $O/head.txt 2>/dev/null
export LCB_WATCHDOG_S=300
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x > $O/pytest_gpu_parity.log 2>&1; tail -3 $O/pytest_gpu_parity.log
V="base hostc:host_commit=1 sync:sync_jobs=1 base_again"
for w in ecoli62 primates8_test mice16_test ecoli10; do
  LCB_VERBOSE=1 timeout 900 python scripts/ab_engine.py --workload $w $V > $O/ab_$w.txt 2> $O/ab_$w.err; cat $O/ab_$w.txt; grep -E "lcb engine|side lanes|overflows out|seeds per variant" $O/ab_$w.err | head -24
done
LCB_VERBOSE=1 LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/prof.json 2> $O/prof.err
python - <<'PY'
import collections
launch = {}; seeds = collections.defaultdict(list)
for line in open("gpurun_out/r4b/trace.tsv"):
    f = line.rstrip("\n").split("\t")
    if f[0] != "#seed":
        launch[int(f[0])] = (int(f[1]), f[3], float(f[4])); continue
    d = dict(x.split("=") for x in f[4:]); d = {k: int(v) for k, v in d.items()}; seeds[int(f[1])].append(d)
tot = sum(v[2] for v in launch.values())
print("launches %d, kernel ms %.1f" % (len(launch), tot))
waste = collections.defaultdict(float); cnt = collections.Counter(); crit = collections.defaultdict(float)
for lid, (n, mode, ms) in launch.items():
    sd = seeds.get(lid, [])
    if not sd: continue
    ovf = [d for d in sd if d["st"] != 0]
    ok = [d for d in sd if d["st"] == 0]
    for d in ovf: waste[mode] += d["ticks"] / 1e5; cnt[mode] += 1
    longest_ok = max([d["ticks"] for d in ok], default=0) / 1e5
    longest_ovf = max([d["ticks"] for d in ovf], default=0) / 1e5
    if longest_ovf > longest_ok: crit[mode] += longest_ovf - longest_ok
for m in ("compact", "wide", "big"):
    print("%s: %d seeds left the variant with an overflow after %.1f ms of work in total; launches whose longest seed was one of them: %.1f ms of critical path beyond the longest seed that finished" % (m, cnt[m], waste[m], crit[m]))
# the slowest 12 seeds of the pass
allp = sorted((d for v in seeds.values() for d in v), key=lambda d: -d["ticks"])[:12]
for d in allp: print("   seed st=%d ticks=%.1f ms push=%d vote=%d inst=%d" % (d["st"], d["ticks"] / 1e5, d["push"], d["vote"], d["inst"]))
PY
grep -v "^#seed" $O/trace.tsv > $O/launch_trace.tsv; rm -f $O/trace.tsv
# the flight-recorder sites (LCB_MARK) compiled out: what they cost
run() {
  local v=$1; shift
  timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, launches %s stops %s jobs %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"], c["job_launches"], c["jobs"], c["host_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
for w in ecoli62 mice16_test; do LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_nomark.so run nomark_$w --workload $w; run stock_$w --workload $w; done
