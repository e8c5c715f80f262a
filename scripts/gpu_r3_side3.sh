# Round 3, GPU call 7: branch-free vote loads; side lanes with frozen phases; wide variant with 8 or 16 wavefronts
mkdir -p gpurun_out/r3e7
O=gpurun_out/r3e7
export LCB_WATCHDOG_S=120
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() {
  local v=$1 lib=$2; shift 2
  LCB_LIB=$lib LCB_VERBOSE=1 LCB_TRACE_LAUNCHES=$O/trace_$v.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, launches %s, jobs %s used %s stops %s side %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"].get("launches_per_step"), c["jobs"], c["jobs_used"], c["job_launches"], c.get("side"), c["host_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-1500:])
PY
}
P=$PWD/sibeliaz_amd
run side ""
run sync "" --engine-opt sync_jobs=1
run side_nww8 $P/libsibeliaz_amd_nww8.so
run sync_nww8 $P/libsibeliaz_amd_nww8.so --engine-opt sync_jobs=1
run side_c2 "" --workload ecoli10
run side_c2_nww8 $P/libsibeliaz_amd_nww8.so --workload ecoli10
python scripts/analyze_trace.py $O/trace_sync.tsv | head -6
