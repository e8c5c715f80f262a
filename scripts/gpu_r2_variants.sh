# compares library variants (LCB_LIB) on config 2 and config 3: value, kernel ms
mkdir -p gpurun_out
export LCB_WATCHDOG_S=300
R=$PWD
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10'); bench.ensure_workload('ecoli62')" > gpurun_out/gen.log 2>&1
for v in "" _nw8 _nw4; do
  export LCB_LIB=$R/sibeliaz_amd/libsibeliaz_amd$v.so
  timeout 600 python bench.py --workload ecoli10 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/var_c2$v.json 2> gpurun_out/var_c2$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/var_c2$v.json"))
print("c2 variant '$v': %.0f seeds/s  ms_per_step %.1f  kernel_ms %.1f launches %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"]))
PY
done
for v in "" _nw8; do
  export LCB_LIB=$R/sibeliaz_amd/libsibeliaz_amd$v.so
  timeout 900 python bench.py --workload ecoli62 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/var_c3$v.json 2> gpurun_out/var_c3$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/var_c3$v.json"))
print("c3 variant '$v': %.0f seeds/s  ms_per_step %.1f  kernel_ms %.1f launches %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"]))
PY
done
unset LCB_LIB
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c2s.tsv LCB_TRACE_SEEDS=1 timeout 600 python bench.py --workload ecoli10 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_c2s.json 2> gpurun_out/bench_c2s.err
python scripts/analyze_trace.py gpurun_out/trace_c2s.tsv 2>&1 | head -12
rm -f gpurun_out/trace_c2s.tsv
