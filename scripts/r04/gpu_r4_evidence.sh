# Round 4 evidence on ONE box for the final build (HEAD written first): the whole `pytest -m gpu` suite + smoke, rocprofv3 kernel stats and
# the two PMC traffic passes of the bench command (no reference legs there), the bench lines of config 3 and of both k = 25 shapes with the
# bounded CPU-baseline protocol (<= 6 minutes of reference legs each), then the Gbp-scale parity cases. Outputs under gpurun_out/r4ev
# (copied into profiles/r04 afterwards). Every step has its own time limit.
MODE=${1:-all}
mkdir -p gpurun_out/r4ev
R=$PWD; O=$R/gpurun_out/r4ev
export LCB_WATCHDOG_S=600
# (the snapshot on the GPU box has no .git: the commit is written into .evidence_head right before the call)
git -C $R rev-parse HEAD > $O/evidence_head.txt 2>/dev/null || cp $R/.evidence_head $O/evidence_head.txt
cat $O/evidence_head.txt
if [ "$MODE" = all ] || [ "$MODE" = tests ]; then
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x -s > $O/pytest_gpu.log 2>&1; grep -E "seeds,|passed|failed|skipped" $O/pytest_gpu.log | tail -8
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
fi
if [ "$MODE" = all ] || [ "$MODE" = prof ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/prof.log 2>&1
cat $O/prof/*kernel_stats.csv 2>/dev/null | head -12; find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats.csv \;
find $O/prof -name "*agent_info.csv" -exec cp {} $O/rocprofv3_agent_info.csv \;
find $O/prof -name "*kernel_trace.csv" -delete
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/pmc_write.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os, json
O = "gpurun_out/r4ev"
tot = {}; launches = 0
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob("%s/%s/**/*counter_collection.csv" % (O, d), recursive=True)
    if not files: print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][-44:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k] += 1
    with open("%s/%s_summary.txt" % (O, d), "w") as f:
        for k, v in agg.items():
            line = k + "  dispatches=%d  " % calls[k] + "  ".join("%s=%.6g" % kv for kv in sorted(v.items()))
            print(line); f.write(line + "\n")
    ship = lambda k: "lcb_process_kernel" in k
    tot[name] = sum(v[name] for k, v in agg.items() if ship(k))
    launches = sum(c for k, c in calls.items() if ship(k))
    for fn in glob.glob("%s/%s/**/*.csv" % (O, d), recursive=True): os.remove(fn)
if len(tot) == 2:
    b = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/gpu_r4_evidence.sh) over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline` (config 3 at full size), summed over the lcb_process_kernel instantiations; KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction, an upper bound for narrow gathers)",
               "fetch_kb_raw": tot["FETCH_SIZE"], "write_kb_raw": tot["WRITE_SIZE"], "launches": launches, "hbm_bytes_per_pass": b,
               "hbm_bytes_per_launch": b / max(1, launches)}, open("%s/pmc_traffic.json" % O, "w"), indent=1)
    print(open("%s/pmc_traffic.json" % O).read())
    os.makedirs("profiles/r04", exist_ok=True)
    json.dump(json.load(open("%s/pmc_traffic.json" % O)), open("profiles/r04/pmc_traffic.json", "w"), indent=1)   # the bench line below quotes it
PY
fi
if [ "$MODE" = all ] || [ "$MODE" = bench ]; then
LCB_VERBOSE=1 timeout 1200 python bench.py --steps 2 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
tail -6 $O/bench_n1.err | cut -c1-400; cut -c1-1500 $O/bench_n1.json
for w in primates8_test mice16_test; do
  LCB_VERBOSE=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --cpu-baseline-budget 240 > $O/bench_n1_$w.json 2> $O/bench_n1_$w.err; cut -c1-400 $O/bench_n1_$w.json
  python - <<PY
import json
d = json.load(open("$O/bench_n1_$w.json")); cb = d.get("cpu_baseline", {}); wc = d.get("wall_clock", {})
print("$w: %.0f seeds/s (%.0f ms per pass) | reference -t %s: %.0f seeds/s (%s; gff equal %s) | whole sibeliaz-lcb process %.1f s" % (d["value"], d["ms_per_step"], cb.get("cores"), cb.get("value", 0), "whole workload" if "WHOLE" in cb.get("sample", "") else "sample", cb.get("gff_md5_equal"), wc.get("sibeliaz_lcb_process_s", 0)))
PY
done
fi
if [ "$MODE" = all ] || [ "$MODE" = scaled ]; then
timeout 900 python scripts/check_fullsize_scaled.py > $O/fullsize_scaled.log 2>&1; tail -4 $O/fullsize_scaled.log
fi
