# GPU test tier: parity tests through the C ABI. Bounded: per-test timeout + global timeout.
mkdir -p gpurun_out
export LCB_WATCHDOG_S=${LCB_WATCHDOG_S:-60}
timeout ${GPU_TESTS_TIMEOUT:-600} python -m pytest tests -m gpu -q --timeout 150 -x "$@" 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
