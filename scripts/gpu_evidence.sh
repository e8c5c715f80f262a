# GPU evidence for one round: usage  bash scripts/gpu_evidence.sh [light|full]
#   light: parity tests, smoke, bench line (N=1, with the reference as cpu_baseline), rocprofv3 kernel stats
#   full : PMC traffic passes first (so the bench line quotes this build's traffic), then everything in light,
#          the 62-strain workload and the 2-rank plumbing check on one GPU (gloo)
MODE=${1:-light}
mkdir -p gpurun_out
export LCB_WATCHDOG_S=180
R=$PWD
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10')"
if [ "$MODE" = full ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc3.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc4 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc4.log 2>&1
  cd $R
  python - <<'PY'
import csv, glob, collections, os, json
tot = {}; launches = 0
for d, name in (("pmc3", "FETCH_SIZE"), ("pmc4", "WRITE_SIZE")):
    files = glob.glob("gpurun_out/%s/*counter_collection.csv" % d)
    if not files: print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][-44:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k] += 1
    with open("gpurun_out/%s_summary.txt" % d, "w") as f:
        for k, v in agg.items():
            line = k + "  dispatches=%d  " % calls[k] + "  ".join("%s=%.6g" % kv for kv in sorted(v.items()))
            print(line); f.write(line + "\n")
    # the shipped instantiations only: <MODE, false, NW, false>; a fresh box also runs bench.py's one-time stats-mode pass (<MODE, true, ...>)
    ship = lambda k: "lcb_process_kernel" in k and ", true," not in k
    tot[name] = sum(v[name] for k, v in agg.items() if ship(k))
    launches = sum(c for k, c in calls.items() if ship(k))
    for fn in files: os.remove(fn)
    for fn in glob.glob("gpurun_out/%s/*kernel_trace.csv" % d): os.remove(fn)
if len(tot) == 2:
    b = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/gpu_evidence.sh full) over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline`, summed over the lcb_process_kernel instantiations; KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction, an upper bound for narrow gathers)",
               "fetch_kb_raw": tot["FETCH_SIZE"], "write_kb_raw": tot["WRITE_SIZE"], "launches": launches, "hbm_bytes_per_pass": b,
               "hbm_bytes_per_launch": b / max(1, launches)}, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
    print(open("gpurun_out/pmc_traffic.json").read())
PY
fi
timeout 600 python -m pytest tests -m gpu -q --timeout 150 -x 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -2 gpurun_out/bench_n1.err; cat gpurun_out/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cat $R/gpurun_out/prof/*kernel_stats.csv | head -8
rm -f $R/gpurun_out/prof/*kernel_trace.csv
cd $R
if [ "$MODE" = full ]; then
  timeout 600 python bench.py --workload ecoli62_small --steps 2 --warmup 0 > gpurun_out/bench_62small.json 2> gpurun_out/bench_62small.err; cat gpurun_out/bench_62small.json
  LCB_BENCH_BACKEND=gloo LCB_BENCH_SAME_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err
  cat gpurun_out/bench_2rank_gloo.json | cut -c1-400
fi
