mkdir -p gpurun_out
export LCB_WATCHDOG_S=600
R=$PWD
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/pytest_gpu.log
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli62'); bench.ensure_workload('ecoli10')" > gpurun_out/gen.log 2>&1
for v in "" _nwc2 _nwc4; do
  export LCB_LIB=$R/sibeliaz_amd/libsibeliaz_amd$v.so
  LCB_TRACE_LAUNCHES=$R/gpurun_out/ab3_trace$v.tsv timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/ab3_c3$v.json 2> gpurun_out/ab3_c3$v.err
  timeout 600 python bench.py --workload ecoli10 --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/ab3_c2$v.json 2> gpurun_out/ab3_c2$v.err
  python - <<PY
import json, collections
d=json.load(open("gpurun_out/ab3_c3$v.json")); e=json.load(open("gpurun_out/ab3_c2$v.json"))
print("variant '$v': c3 %.0f seeds/s ms %.1f kernel %.1f host %s | c2 %.0f seeds/s ms %.1f kernel %.1f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], {k: round(v) for k, v in d["config"]["host_ms_per_step"].items()}, e["value"], e["ms_per_step"], e["roofline"]["kernel_ms_per_step"]))
t=collections.Counter(); c=collections.Counter()
for ln in open("gpurun_out/ab3_trace$v.tsv"):
    f=ln.split("\t")
    if f[0].startswith("#"): continue
    t[f[3]]+=float(f[4]); c[f[3]]+=1
print("   c3 per pass:", {k:(c[k], round(t[k],1)) for k in t})
PY
done
