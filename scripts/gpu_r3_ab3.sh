# Round 3, GPU call 9: the round-2 library and the current one on the SAME box (config 3, config 2, configs 4/5 at test size)
mkdir -p gpurun_out/r3e9
O=gpurun_out/r3e9
export LCB_WATCHDOG_S=120
run() {
  local v=$1 lib=$2; shift 2
  LCB_LIB=$lib timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, launches %s stops %s jobs %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"].get("launches_per_step"), c["job_launches"], c["jobs"], c["host_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
P=$PWD/sibeliaz_amd
run r2_c3 $P/libsibeliaz_amd_r2.so
run cur_sync_c3 "" --engine-opt sync_jobs=1
run cur_side_c3 ""
run r2_c3_again $P/libsibeliaz_amd_r2.so
for w in primates8_test mice16_test ecoli10; do
run r2_$w $P/libsibeliaz_amd_r2.so --workload $w
run cur_sync_$w "" --engine-opt sync_jobs=1 --workload $w
run cur_side_$w "" --workload $w
done
