mkdir -p gpurun_out
R=$PWD
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10')" > gpurun_out/gen.log 2>&1
for v in "" _precow "" _precow; do
  export LCB_LIB=$R/sibeliaz_amd/libsibeliaz_amd$v.so
  SL=""; if [ "$v" = "_vc512" ]; then SL=1536; fi
  LCB_TRACE_LAUNCHES=$R/gpurun_out/ab_trace$v.tsv timeout 600 python bench.py --workload ecoli10 --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/ab_c2$v.json 2> gpurun_out/ab_c2$v.err
  python - <<PY
import json, collections
d=json.load(open("gpurun_out/ab_c2$v.json"))
print("c2 variant '$v': %.0f seeds/s  ms_per_step %.1f  kernel_ms %.1f launches %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"]))
t=collections.Counter(); c=collections.Counter()
for ln in open("gpurun_out/ab_trace$v.tsv"):
    f=ln.split("\t")
    if f[0].startswith("#"): continue
    t[f[3]]+=float(f[4]); c[f[3]]+=1
print("   per 4 passes:", {k:(c[k], round(t[k],1)) for k in t})
PY
done
