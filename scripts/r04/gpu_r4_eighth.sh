# Round 4, GPU call 8 (the last minutes of the budget): the randomized footprint-completeness test added after the evidence run.
mkdir -p gpurun_out/r4h
export LCB_WATCHDOG_S=120
timeout 280 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 -x -k "random_inputs" > gpurun_out/r4h/pytest_random_fp.log 2>&1; tail -5 gpurun_out/r4h/pytest_random_fp.log
