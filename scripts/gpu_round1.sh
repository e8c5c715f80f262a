# Round-1 GPU evidence: parity tests, smoke, bench line (N=1), 2-rank plumbing check on one GPU (gloo), rocprofv3 kernel stats, PMC traffic.
mkdir -p gpurun_out
export LCB_WATCHDOG_S=180
R=$PWD
timeout 600 python -m pytest tests -m gpu -q --timeout 150 -x 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
tail -2 gpurun_out/bench_r1.err; cat gpurun_out/bench_r1.json
timeout 600 python bench.py --workload ecoli62_small --steps 2 --warmup 0 > gpurun_out/bench_62small.json 2> gpurun_out/bench_62small.err; cat gpurun_out/bench_62small.json
LCB_BENCH_BACKEND=gloo LCB_BENCH_SAME_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err
cat gpurun_out/bench_2rank_gloo.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o r1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_r1.log 2>&1
cat $R/gpurun_out/prof_r1/*kernel_stats.csv | head -6
rm -f $R/gpurun_out/prof_r1/*kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc4 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc4.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in ("pmc3","pmc4"):
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
    for fn in files: os.remove(fn)
    for fn in glob.glob("gpurun_out/%s/*kernel_trace.csv" % d): os.remove(fn)
PY
