# Round 3, GPU call 4: is the process kernel instruction-fetch bound? (kernels are 85-135 KB of code; the instruction cache is shared by CUs)
mkdir -p gpurun_out/r3e4
R=$PWD; O=$R/gpurun_out/r3e4
export LCB_WATCHDOG_S=180
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10')"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | head -60 > $O/counters_list.txt
head -40 $O/counters_list.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/$tag -o p -- python $R/bench.py --workload ecoli10 --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/$tag.log 2>&1
  tail -2 $O/$tag.log | cut -c1-300
done
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in glob.glob("gpurun_out/r3e4/S*"):
    if not os.path.isdir(d): continue
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][-60:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    with open(d + "_summary.txt", "w") as f:
        for k, v in agg.items():
            line = k + "  " + "  ".join("%s=%.4g" % kv for kv in sorted(v.items()))
            print(line); f.write(line + "\n")
    for fn in glob.glob(d + "/**/*.csv", recursive=True): os.remove(fn)
PY
