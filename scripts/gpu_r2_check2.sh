mkdir -p gpurun_out
export LCB_WATCHDOG_S=600
R=$PWD
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -x 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
for v in "" _precow; do
  export LCB_LIB=$R/sibeliaz_amd/libsibeliaz_amd$v.so
  LCB_TRACE_LAUNCHES=$R/gpurun_out/ab_trace$v.tsv timeout 600 python bench.py --workload ecoli10 --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/ab_c2$v.json 2> gpurun_out/ab_c2$v.err
  python - <<PY
import json, collections
d=json.load(open("gpurun_out/ab_c2$v.json"))
print("c2 variant '$v': %.0f seeds/s  ms_per_step %.1f  kernel_ms %.1f launches %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"]), d["config"]["untimed_s"])
t=collections.Counter(); c=collections.Counter()
for ln in open("gpurun_out/ab_trace$v.tsv"):
    f=ln.split("\t")
    if f[0].startswith("#"): continue
    t[f[3]]+=float(f[4]); c[f[3]]+=1
print("   per 4 passes:", {k:(c[k], round(t[k],1)) for k in t})
PY
done
unset LCB_LIB
LCB_VERBOSE=1 LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c3.tsv timeout 1500 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/chk_c3.json 2> gpurun_out/chk_c3.err
tail -7 gpurun_out/chk_c3.err
python - <<'PY'
import json, collections
d=json.load(open("gpurun_out/chk_c3.json"))
print("c3 %.0f seeds/s ms %.1f kernel %.1f launches %.0f frac %.5f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"], d["roofline"]["frac"]), d["config"]["host_ms_per_step"], d["config"]["untimed_s"])
t=collections.Counter(); c=collections.Counter()
for ln in open("gpurun_out/trace_c3.tsv"):
    f=ln.split("\t")
    if f[0].startswith("#"): continue
    n=int(f[1]); b="<=16" if n<=16 else "<=256" if n<=256 else "<=512" if n<=512 else "<=1280" if n<=1280 else "<=4096" if n<=4096 else ">4096"
    t[(f[3],b)]+=float(f[4]); c[(f[3],b)]+=1
for k in sorted(t): print("   ", k, c[k], round(t[k],1))
PY
md5sum /tmp/lcb_bench/ecoli62/gpu_out/blocks_coords.gff
