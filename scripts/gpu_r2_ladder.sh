# after the final evidence run: the variant ladder routes by the pool size at the overflow; ladder tests + one bench of config 3
mkdir -p gpurun_out
export LCB_WATCHDOG_S=300
timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 100 -x -k "overflow or fallback or variant" 2>&1 | grep -E "passed|failed|error" | tail -3
LCB_VERBOSE=1 timeout 260 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-roofline > gpurun_out/ladder_c3.json 2> gpurun_out/ladder_c3.err
grep -E "overflows out of|seeds per variant" gpurun_out/ladder_c3.err | tail -5; cut -c1-400 gpurun_out/ladder_c3.json
