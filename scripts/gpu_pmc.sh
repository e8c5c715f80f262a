# PMC passes for the process kernel (separate runs, counters only with --kernel-trace)
mkdir -p gpurun_out
export LCB_WATCHDOG_S=180
R=$PWD
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10')"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $R/gpurun_out/pmc1 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmc2 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc2.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc3.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc4 -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc4.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in ("pmc1","pmc2","pmc3","pmc4"):
    files = glob.glob("gpurun_out/%s/*counter_collection.csv" % d)
    if not files:
        print(d, "no counter file", os.listdir("gpurun_out/"+d) if os.path.isdir("gpurun_out/"+d) else "")
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][-40:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    with open("gpurun_out/%s_summary.txt" % d, "w") as f:
        for k, v in agg.items():
            line = k + "  " + "  ".join("%s=%.4g" % kv for kv in sorted(v.items()))
            print(line); f.write(line + "\n")
    for fn in files: os.remove(fn)
    for fn in glob.glob("gpurun_out/%s/*kernel_trace.csv" % d): os.remove(fn)
PY
