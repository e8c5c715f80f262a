# last measurement of round 2: the ladder routes a wide path overflow by the pool size at the end of the seed; config 3, one pass (no hints)
mkdir -p gpurun_out
export LCB_WATCHDOG_S=300
LCB_VERBOSE=1 timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > gpurun_out/ladder2_c3.json 2> gpurun_out/ladder2_c3.err
grep -E "compact path set|overflows out of (compact|wide)|seeds per variant" gpurun_out/ladder2_c3.err | tail -4
python - <<PY
import json
d = json.load(open("gpurun_out/ladder2_c3.json"))
print("%.0f seeds/s, %.1f ms, kernel %.1f ms, launches %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"].get("launches_per_step")))
PY
