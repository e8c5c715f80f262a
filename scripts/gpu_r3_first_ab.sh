# Round 3, first GPU call: A/B of the candidates prepared at the end of round 2 (no GPU budget was left to measure them).
# Same box, config 3 at full size, one pass each (hints are per pass), plus the parity tests of each variant build.
#   base     the shipped build
#   bighot   -DLCB_BIG_HOT=256u : fields of the first 256 pool entries of the big variant in LDS (build: python sibeliaz_amd/build.py variant bighot -DLCB_BIG_HOT=256u)
#   jobs64 / jobs256 / jobs512   lcb_hooks.max_jobs (job launches in the wide variant; 0.8 % of the 1 280 jobs per stop are used)
#   nwbig16  -DLCB_NW_BIG=16 : 16 wavefronts in the big variant
#   pathsig  -DLCB_PATH_SIG=1 + lcb_hooks.relax_views = 1 : the kernels report the path's vertices, a predicted mark that did not come
#            true voids a job's result only if the job can have read it (model: jobs -30 %, critical path -11 %)
# and a per-seed section trace (LCB_TRACE_SEEDS=1: vote / push / score split of the slowest seeds) of the shipped build
mkdir -p gpurun_out
export LCB_WATCHDOG_S=300
python sibeliaz_amd/build.py variant bighot -DLCB_BIG_HOT=256u
python sibeliaz_amd/build.py variant nwbig16 -DLCB_NW_BIG=16
python sibeliaz_amd/build.py variant pathsig -DLCB_PATH_SIG=1
for v in base bighot nwbig16 pathsig jobs64 jobs256 jobs512; do
  LIB=""; EXTRA=""
  if [ $v = bighot ] || [ $v = nwbig16 ] || [ $v = pathsig ]; then LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_$v.so; fi
  if [ $v = pathsig ]; then EXTRA="--engine-opt relax_views=1"; fi
  if [ $v = jobs64 ]; then EXTRA="--engine-opt max_jobs=64"; fi
  if [ $v = jobs256 ]; then EXTRA="--engine-opt max_jobs=256"; fi
  if [ $v = jobs512 ]; then EXTRA="--engine-opt max_jobs=512"; fi
  if [ -n "$LIB" ]; then LCB_LIB=$LIB timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 100 -x -k "variant or overflow" 2>&1 | grep -E "passed|failed|rror" | tail -2; fi
  LCB_LIB=$LIB LCB_VERBOSE=1 LCB_TRACE_LAUNCHES=gpurun_out/r3ab_trace_$v.tsv timeout 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline $EXTRA > gpurun_out/r3ab_$v.json 2> gpurun_out/r3ab_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r3ab_$v.json"))
print("$v: %.0f seeds/s, %.1f ms, kernel %.1f ms, launches %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"].get("launches_per_step")))
PY
done
LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=gpurun_out/r3ab_seedtrace.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > gpurun_out/r3ab_seedtrace.json 2> gpurun_out/r3ab_seedtrace.err
python scripts/analyze_trace.py gpurun_out/r3ab_seedtrace.tsv | tee gpurun_out/r3ab_seedtrace_summary.txt
gzip -f gpurun_out/r3ab_seedtrace.tsv
