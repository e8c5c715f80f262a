# Round 3, GPU call 11: round-2 vote walk restored (+ deferred path verification): parity suite, then A/B on one box
mkdir -p gpurun_out/r3e11
O=gpurun_out/r3e11
export LCB_WATCHDOG_S=120
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
run() {
  local v=$1 lib=$2; shift 2
  LCB_LIB=$lib timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, stops %s jobs %s side %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], c["job_launches"], c["jobs"], c.get("side")))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
P=$PWD/sibeliaz_amd
for w in ecoli62 mice16_test; do
run r2_$w $P/libsibeliaz_amd_r2.so --workload $w
run nodefer_sync_$w $P/libsibeliaz_amd_nodefer.so --workload $w --engine-opt sync_jobs=1
run defer_sync_$w "" --workload $w --engine-opt sync_jobs=1
run defer_side_$w "" --workload $w
done
run defer_sync_ecoli10 "" --workload ecoli10 --engine-opt sync_jobs=1
run defer_side_ecoli10 "" --workload ecoli10
