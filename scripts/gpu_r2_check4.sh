mkdir -p gpurun_out
export LCB_WATCHDOG_S=900
R=$PWD
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -x 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee gpurun_out/pytest_gpu.log
( python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('primates8_scaled')" > gpurun_out/gen4.log 2>&1 ) &
for v in 0 1; do
  EXTRA=""; if [ $v = 0 ]; then EXTRA="--overlap"; fi
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli $EXTRA > gpurun_out/ab4_c3_$v.json 2> gpurun_out/ab4_c3_$v.err
  timeout 600 python bench.py --workload ecoli10 --steps 3 --warmup 1 --no-cpu-baseline --no-cli $EXTRA > gpurun_out/ab4_c2_$v.json 2> gpurun_out/ab4_c2_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab4_c3_$v.json")); e=json.load(open("gpurun_out/ab4_c2_$v.json"))
print("no_overlap=$v: c3 %.0f seeds/s ms %.1f kernel %.1f early %s host %s | c2 %.0f seeds/s ms %.1f kernel %.1f early %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["config"]["early_rounds"], {k: round(v) for k, v in d["config"]["host_ms_per_step"].items()}, e["value"], e["ms_per_step"], e["roofline"]["kernel_ms_per_step"], e["config"]["early_rounds"]))
PY
done
wait
D=/tmp/lcb_bench/primates8_scaled
( time LCB_VERBOSE=1 $R/sibeliaz_amd/bin/sibeliaz-lcb --graph $D/graph.bin $D/genomes.fa -k 25 -b 200 -m 50 -a 150 -t 32 -o $D/cli_out --noseq ) > gpurun_out/c45b_ours.log 2>&1
grep -E "lcb:|seeds per|overflows|real" gpurun_out/c45b_ours.log
md5sum $D/cli_out/blocks_coords.gff
