# Round 5 evidence on ONE box for the final build (commit in head.txt): the whole `pytest -m gpu` suite + smoke, rocprofv3 kernel stats and the
# two PMC traffic passes of the bench command, the bench line of config 3 with the bounded CPU-baseline protocol, the lines of the other
# BASELINE shapes (config 3 with a = 868, configs 4 / 5 at test size) for bench.py's `secondary` array, and config 4's shape at 4.2 Gbp against
# the reference's hash. Outputs under gpurun_out/r5ev (copied into profiles/r05 afterwards). Every step has its own time limit.
MODE=${1:-all}
mkdir -p gpurun_out/r5ev
R=$PWD; O=$R/gpurun_out/r5ev
export LCB_WATCHDOG_S=600
cp $R/.evidence_head $O/evidence_head.txt 2>/dev/null; cat $O/evidence_head.txt
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
if [ "$MODE" = all ] || [ "$MODE" = tests ]; then
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|skipped" $O/pytest_gpu.log | tail -3
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
fi
if [ "$MODE" = all ] || [ "$MODE" = prof ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats.csv \; ; head -8 $O/rocprofv3_kernel_stats.csv | cut -c1-200
find $O/prof -name "*agent_info.csv" -exec cp {} $O/rocprofv3_agent_info.csv \;
find $O/prof -name "*kernel_trace.csv" -delete
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/pmc_write.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os, json
O = "gpurun_out/r5ev"
tot = {}; launches = 0
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob("%s/%s/**/*counter_collection.csv" % (O, d), recursive=True)
    if not files: print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][-52:]
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
    js = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/r05/gpu_r5_evidence.sh) over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline` (config 3 at full size), summed over the lcb_process_kernel instantiations; KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction, an upper bound for narrow gathers)",
          "kernel_source_hash": open("%s/kernel_source_hash.txt" % O).read().strip(), "commit": open("%s/evidence_head.txt" % O).read().strip() if os.path.exists("%s/evidence_head.txt" % O) else None,
          "fetch_kb_raw": tot["FETCH_SIZE"], "write_kb_raw": tot["WRITE_SIZE"], "launches": launches, "hbm_bytes_per_pass": b, "hbm_bytes_per_launch": b / max(1, launches)}
    json.dump(js, open("%s/pmc_traffic.json" % O, "w"), indent=1)
    print(open("%s/pmc_traffic.json" % O).read())
    os.makedirs("profiles/r05", exist_ok=True)
    json.dump(js, open("profiles/r05/pmc_traffic.json", "w"), indent=1)   # the bench line below quotes it
PY
fi
if [ "$MODE" = all ] || [ "$MODE" = bench ]; then
LCB_VERBOSE=1 timeout 1100 python bench.py --steps 2 --warmup 1 --cpu-baseline-budget 330 > $O/bench_n1.json 2> $O/bench_n1.err
grep "lcb engine" $O/bench_n1.err | tail -2 | cut -c1-400; cut -c1-1200 $O/bench_n1.json
for w in primates8_test mice16_test; do
  LCB_VERBOSE=1 timeout 500 python bench.py --workload $w --steps 3 --warmup 1 --cpu-baseline-budget 150 > $O/bench_n1_$w.json 2> $O/bench_n1_$w.err; cut -c1-300 $O/bench_n1_$w.json
done
LCB_VERBOSE=1 timeout 400 python bench.py --workload ecoli62_a868 --steps 2 --warmup 1 --no-roofline > $O/bench_n1_ecoli62_a868.json 2> $O/bench_n1_ecoli62_a868.err; cut -c1-300 $O/bench_n1_ecoli62_a868.json
fi
if [ "$MODE" = all ] || [ "$MODE" = big ]; then
timeout 1500 python scripts/check_fullsize_scaled.py config4_primates8_4g_scaled > $O/fullsize_4g.log 2>&1; tail -4 $O/fullsize_4g.log
fi
