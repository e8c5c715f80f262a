# Round 3, GPU call 10: bisecting the kernel regression against the round-2 library on one box (synchronous engine)
mkdir -p gpurun_out/r3e10
O=gpurun_out/r3e10
export LCB_WATCHDOG_S=120
run() {
  local v=$1 lib=$2; shift 2
  LCB_LIB=$lib timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline --engine-opt sync_jobs=1 "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
P=$PWD/sibeliaz_amd
for w in mice16_test ecoli62; do
run r2_$w $P/libsibeliaz_amd_r2.so --workload $w
run r2walk_$w $P/libsibeliaz_amd_r2walk.so --workload $w
run nodefer_$w $P/libsibeliaz_amd_nodefer.so --workload $w
run base_$w "" --workload $w
done
