# Round 3, GPU call 8: which of the vote changes pay (synchronous engine, config 3 and 2)
mkdir -p gpurun_out/r3e8
O=gpurun_out/r3e8
export LCB_WATCHDOG_S=120
run() {
  local v=$1 lib=$2; shift 2
  LCB_LIB=$lib timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, launches %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"].get("launches_per_step"), c["host_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-1500:])
PY
}
P=$PWD/sibeliaz_amd
for w in ecoli62 ecoli10; do
run base_$w "" --engine-opt sync_jobs=1 --workload $w
run nodefer_$w $P/libsibeliaz_amd_nodefer.so --engine-opt sync_jobs=1 --workload $w
run noahead_$w $P/libsibeliaz_amd_noahead.so --engine-opt sync_jobs=1 --workload $w
run neither_$w $P/libsibeliaz_amd_neither.so --engine-opt sync_jobs=1 --workload $w
done
