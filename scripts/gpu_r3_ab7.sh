# Round 3, GPU call 13: compact variant hands seeds with >= 24 voters to the wide variant (same box A/B)
mkdir -p gpurun_out/r3e13
O=gpurun_out/r3e13
export LCB_WATCHDOG_S=120
run() {
  local v=$1 lib=$2; shift 2
  LCB_LIB=$lib LCB_VERBOSE=1 LCB_TRACE_LAUNCHES=$O/trace_$v.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, launches %s stops %s jobs %s variants %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"], c["job_launches"], c["jobs"], c["seeds_per_kernel_variant"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
P=$PWD/sibeliaz_amd

for w in ecoli62; do
run noheavy_$w $P/libsibeliaz_amd_noheavy.so --workload $w
run heavy384_$w "" --workload $w
run heavy1k_$w $P/libsibeliaz_amd_heavy1k.so --workload $w
done
python scripts/analyze_trace.py $O/trace_heavy384_ecoli62.tsv | head -6
