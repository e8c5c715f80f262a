# Round 3, GPU call 12: 16 wavefronts in the big variant (same box A/B, side lanes on)
mkdir -p gpurun_out/r3e12
O=gpurun_out/r3e12
export LCB_WATCHDOG_S=120
run() {
  local v=$1 lib=$2; shift 2
  LCB_LIB=$lib timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, stops %s jobs %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], c["job_launches"], c["jobs"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
P=$PWD/sibeliaz_amd
LCB_LIB=$P/libsibeliaz_amd_nwbig16.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "variant or overflow or footprints or side_lanes" 2>&1 | grep -E "passed|failed" | tail -2
for w in ecoli62 primates8_test mice16_test; do
run base_$w "" --workload $w
run nwbig16_$w $P/libsibeliaz_amd_nwbig16.so --workload $w
done
