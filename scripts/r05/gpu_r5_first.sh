# Round 5, GPU call 1: (1) the (segment, offset) positions on the MI355X - the SEG kernels on the goldens with flat indices beyond 2^32,
# the footprints of launches that keep every CU busy (tests/test_gpu_segments.py) - and a sample of the parity suite after the refactoring;
# (2) same-box A/B of the lazy round tails (lcb_hooks.lazy_span) on config 3 and on config 4's shape.
mkdir -p gpurun_out/r5a
R=$PWD; O=$R/gpurun_out/r5a
export LCB_WATCHDOG_S=300
cp $R/.evidence_head $O/head.txt 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_segments.py -m gpu -q --timeout 300 -x > $O/pytest_segments.log 2>&1; tail -5 $O/pytest_segments.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "per_seed_parity or each_kernel_variant or find_blocks_matches_reference or footprints_cover_every_read_on_gpu or device_resident" > $O/pytest_parity_sample.log 2>&1; tail -3 $O/pytest_parity_sample.log
timeout 600 python scripts/ab_engine.py --workload ecoli62 --passes 1 warm base lazyoff:lazy_span=-1 lazy16:lazy_span=16 lazy4:lazy_span=4 base_again lazyoff_again:lazy_span=-1 > $O/ab_ecoli62.txt 2>&1; cat $O/ab_ecoli62.txt | cut -c1-330
timeout 300 python scripts/ab_engine.py --workload primates8_test --passes 2 warm base lazyoff:lazy_span=-1 lazy16:lazy_span=16 > $O/ab_primates8_test.txt 2>&1; cat $O/ab_primates8_test.txt | cut -c1-330
timeout 300 python scripts/ab_engine.py --workload mice16_test --passes 2 warm base lazyoff:lazy_span=-1 lazy16:lazy_span=16 > $O/ab_mice16_test.txt 2>&1; cat $O/ab_mice16_test.txt | cut -c1-330
