# Round 4, GPU call 5: the commit kernel hands a phase over to the host where a footprint lies near a mark (no exact range tests on the
# device); the search / classification rewrite of call 4 reverted. Is the device-resident commit neutral now?
mkdir -p gpurun_out/r4e
O=gpurun_out/r4e
git rev-parse HEAD > $O/head.txt 2>/dev/null
export LCB_WATCHDOG_S=300
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "resident or screened or early" > $O/pytest_gpu_subset.log 2>&1; tail -2 $O/pytest_gpu_subset.log
V="base hostc:host_commit=1 base_again hostc_again:host_commit=1"
for w in ecoli62 ecoli10 primates8_test; do
  LCB_VERBOSE=1 timeout 900 python scripts/ab_engine.py --workload $w $V > $O/ab_$w.txt 2> $O/ab_$w.err; cat $O/ab_$w.txt; grep -E "lcb engine" $O/ab_$w.err | head -2
done
