# Round 5, GPU call 2: (1) the parity tests the first call's sample stopped in front of (device-resident commit, early critical launch, side lanes);
# (2) can the footprints-under-load test see the memory-model hazard commit 0d6481f closed? - the same test against a library built with
# -DLCB_TEST_PLAIN_FP_READ (plain loads / stores of the footprint slots in the HBM workspace, as before that commit);
# (3) same-box A/B of lcb_hooks.lazy_span after the per-plan memo of the view checks, with the engine's host-time split (LCB_VERBOSE),
# and the SEG kernel instantiations on config 3 (dev.seg_cap set: same input, segment-aware kernels) against the plain ones.
mkdir -p gpurun_out/r5b
R=$PWD; O=$R/gpurun_out/r5b
export LCB_WATCHDOG_S=300
cp $R/.evidence_head $O/head.txt 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "device_resident or early_critical or side_lanes or gpu_set or rccl" > $O/pytest_parity_engine.log 2>&1; tail -3 $O/pytest_parity_engine.log
LCB_LIB=$R/sibeliaz_amd/libsibeliaz_amd_plainfp.so timeout 600 python -m pytest tests/test_gpu_segments.py -m gpu -q --timeout 300 -k "under_load" > $O/pytest_under_load_plain_reads.log 2>&1; tail -4 $O/pytest_under_load_plain_reads.log
export LCB_VERBOSE=1
timeout 700 python scripts/ab_engine.py --workload ecoli62 --passes 1 warm base lazy4:lazy_span=4 lazyoff:lazy_span=-1 lazy16:lazy_span=16 base_again lazy4_again:lazy_span=4 hostc:host_commit=1 seg:dev.seg_cap=4000000000 > $O/ab_ecoli62.txt 2>&1; grep -E "seeds/s|lcb engine|DIFFER|rror" $O/ab_ecoli62.txt | cut -c1-420
for w in primates8_test mice16_test; do
timeout 300 python scripts/ab_engine.py --workload $w --passes 2 warm base lazy4:lazy_span=4 lazy16:lazy_span=16 lazy32:lazy_span=32 lazyoff:lazy_span=-1 seg:dev.seg_cap=4000000000 > $O/ab_$w.txt 2>&1; grep -E "seeds/s|DIFFER|rror" $O/ab_$w.txt | cut -c1-330
done
