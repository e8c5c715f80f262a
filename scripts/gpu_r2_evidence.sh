# round 2 evidence: parity suite, smoke, the bench line (config 3 at full size), rocprofv3 kernel stats, PMC traffic passes
MODE=${1:-all}
mkdir -p gpurun_out
export LCB_WATCHDOG_S=600
R=$PWD
if [ "$MODE" = all ] || [ "$MODE" = tests ]; then
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -x 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
fi
if [ "$MODE" = all ] || [ "$MODE" = bench ]; then
LCB_VERBOSE=1 timeout 2400 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -12 gpurun_out/bench_n1.err; cat gpurun_out/bench_n1.json | cut -c1-1500
LCB_VERBOSE=1 timeout 900 python bench.py --workload ecoli10 --steps 3 --warmup 1 > gpurun_out/bench_n1_config2.json 2> gpurun_out/bench_n1_config2.err
tail -8 gpurun_out/bench_n1_config2.err; cat gpurun_out/bench_n1_config2.json | cut -c1-600
fi
if [ "$MODE" = all ] || [ "$MODE" = prof ]; then
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $R/gpurun_out/prof.log 2>&1
cat $R/gpurun_out/prof/*kernel_stats.csv | head -12
rm -f $R/gpurun_out/prof/*kernel_trace.csv
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os, json
tot = {}; launches = 0
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    if not files: print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][-44:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k] += 1
    with open("gpurun_out/%s_summary.txt" % d, "w") as f:
        for k, v in agg.items():
            line = k + "  dispatches=%d  " % calls[k] + "  ".join("%s=%.6g" % kv for kv in sorted(v.items()))
            print(line); f.write(line + "\n")
    ship = lambda k: "lcb_process_kernel" in k
    tot[name] = sum(v[name] for k, v in agg.items() if ship(k))
    launches = sum(c for k, c in calls.items() if ship(k))
    for fn in files: os.remove(fn)
    for fn in glob.glob("gpurun_out/%s/**/*kernel_trace.csv" % d, recursive=True): os.remove(fn)
if len(tot) == 2:
    b = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/gpu_r2_evidence.sh) over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline` (config 3 at full size), summed over the lcb_process_kernel instantiations; KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction, an upper bound for narrow gathers)",
               "fetch_kb_raw": tot["FETCH_SIZE"], "write_kb_raw": tot["WRITE_SIZE"], "launches": launches, "hbm_bytes_per_pass": b,
               "hbm_bytes_per_launch": b / max(1, launches)}, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
    print(open("gpurun_out/pmc_traffic.json").read())
PY
fi
