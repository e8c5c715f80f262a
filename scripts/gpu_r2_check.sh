# quick check of a build: parity suite, then one pass of config 3 and config 2 (no CPU baseline, no CLI run)
mkdir -p gpurun_out
export LCB_WATCHDOG_S=600
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -x 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
LCB_VERBOSE=1 timeout 1500 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/chk_c3.json 2> gpurun_out/chk_c3.err
tail -8 gpurun_out/chk_c3.err
LCB_VERBOSE=1 timeout 600 python bench.py --workload ecoli10 --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/chk_c2.json 2> gpurun_out/chk_c2.err
tail -7 gpurun_out/chk_c2.err
python - <<'PY'
import json
for n in ("c3","c2"):
    d=json.load(open("gpurun_out/chk_%s.json"%n))
    print(n, "%.0f seeds/s ms %.1f kernel %.1f launches %.0f frac %.5f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"], d["roofline"]["frac"]), d["config"]["host_ms_per_step"], d["config"]["seeds_per_kernel_variant"])
PY
md5sum /tmp/lcb_bench/ecoli62/gpu_out/blocks_coords.gff
