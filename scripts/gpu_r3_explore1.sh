# Round 3, GPU call 1: where a config-3 pass of the round-2 build goes (launch + per-seed section trace) and the A/B of the
# candidates prepared at the end of round 2 (variant libraries are prebuilt in the build container and travel with the snapshot).
mkdir -p gpurun_out/r3e1
O=gpurun_out/r3e1
export LCB_WATCHDOG_S=300
run() {  # name lib extra...
  local v=$1 lib=$2; shift 2
  LCB_LIB=$lib LCB_VERBOSE=1 LCB_TRACE_LAUNCHES=$O/trace_$v.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json"))
    c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel %.1f ms, launches %s, jobs %s used %s joblaunches %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"].get("launches_per_step"), c["jobs"], c["jobs_used"], c["job_launches"], c["host_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e)
PY
}
P=$PWD/sibeliaz_amd
run base ""
run bighot256 $P/libsibeliaz_amd_bighot.so
run bighot512 $P/libsibeliaz_amd_bighot512.so
run nwbig16 $P/libsibeliaz_amd_nwbig16.so
run pathsig $P/libsibeliaz_amd_pathsig.so --engine-opt relax_views=1
run jobs256 "" --engine-opt max_jobs=256
for v in bighot bighot512 pathsig; do
  LCB_LIB=$P/libsibeliaz_amd_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 100 -x -k "variant or overflow" 2>&1 | grep -E "passed|failed|rror" | tail -2
done
LCB_TRACE_SEEDS=1 run seedtrace ""
python scripts/analyze_trace.py $O/trace_seedtrace.tsv | tee $O/seedtrace_summary.txt
python scripts/analyze_trace.py $O/trace_base.tsv | tee $O/base_summary.txt
gzip -f $O/trace_seedtrace.tsv
