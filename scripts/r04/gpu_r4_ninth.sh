# Round 4, GPU call 9 (last): the bench line of a k = 25 shape with the reordered reference protocol of the final bench.py, bounded.
mkdir -p gpurun_out/r4i
timeout 230 python bench.py --workload mice16_test --steps 1 --warmup 0 --cpu-baseline-budget 100 --no-cli > gpurun_out/r4i/bench_mice16_test.json 2> gpurun_out/r4i/bench_mice16_test.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4i/bench_mice16_test.json")); cb = d.get("cpu_baseline", {})
print(d["value"], d["ms_per_step"], cb.get("value"), cb.get("gff_md5_equal"), cb.get("sample", "")[:120], {k: (v.get("analyze_s_median"), v.get("timeout_s"), v.get("skipped")) for k, v in cb.get("legs", {}).items()}, d.get("cpu_baseline_error"))
PY
