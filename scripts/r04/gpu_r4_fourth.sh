# Round 4, GPU call 4: commit kernel with lane-parallel scans of the seeds' states (the walk over them by one wavefront was what cost 30-40 ms
# per round) and batched range tests; search / classification of a push with a wave-uniform binary descent and unconditional reads.
mkdir -p gpurun_out/r4d
O=gpurun_out/r4d
git rev-parse HEAD > $O/head.txt 2>/dev/null
export LCB_WATCHDOG_S=300
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "resident or variant or footprints or screened or overflow or seeds or gpus" > $O/pytest_gpu_subset.log 2>&1; tail -2 $O/pytest_gpu_subset.log
V="base hostc:host_commit=1 base_again hostc_again:host_commit=1"
for w in ecoli62 primates8_test mice16_test ecoli10; do
  LCB_VERBOSE=1 timeout 900 python scripts/ab_engine.py --workload $w $V > $O/ab_$w.txt 2> $O/ab_$w.err; cat $O/ab_$w.txt; grep -E "lcb engine" $O/ab_$w.err | head -2
done
run() {
  local v=$1; shift
  timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline --engine-opt host_commit=1 "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel busy %.1f ms (sum %.1f, side %.1f), launches %s stops %s jobs %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["kernel_ms_sum_over_streams_per_step"], d["roofline"]["kernel_ms_on_side_lanes_per_step"], d["roofline"]["launches_per_step"], c["job_launches"], c["jobs"], c["host_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
for w in ecoli62 primates8_test mice16_test; do LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_prev.so run prev_$w --workload $w; run stock_$w --workload $w; done
LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_prev.so run prev_ecoli62_2 --workload ecoli62; run stock_ecoli62_2 --workload ecoli62
