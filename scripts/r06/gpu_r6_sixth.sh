# Round 6, GPU call 6: more seeds in flight where a pass is bound by the work of its round launches (call 5: at 1.24 Gbp the launches of > 4 096 seeds are
# 72 % work / 1 280 workgroups, 28 % tail). A compact variant with half the pools (128 instances / 512 vote slots: ~17 KB of LDS instead of 32 KB) fits 8
# workgroups per CU (register-bound) instead of 5: libsibeliaz_amd_s128.so with lcb_device_opts.compact_slots 1280 / 1792 / 2048 against the shipped library.
mkdir -p gpurun_out/r6f
R=$PWD; O=$R/gpurun_out/r6f
export LCB_WATCHDOG_S=300
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
S=$R/sibeliaz_amd/libsibeliaz_amd_s128.so
LCB_LIB=$S timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "per_seed or each_kernel or find_blocks_matches or overflow" > $O/pytest_s128.log 2>&1; grep -E "passed|failed|error" $O/pytest_s128.log | tail -3
for w in primates8_scaled mice16_scaled primates8_test mice16_test ecoli62; do
  p=1; [ $w = primates8_test -o $w = mice16_test ] && p=3
  timeout 900 python scripts/ab_engine.py --workload $w --passes $p base base6:dev.compact_slots=1536 > $O/ab_base_$w.txt 2>&1; grep -E "^base|seeds, loaded|DIFFER|rror" $O/ab_base_$w.txt | cut -c1-330
  LCB_LIB=$S timeout 900 python scripts/ab_engine.py --workload $w --passes $p s128x5:dev.compact_slots=1280 s128x7:dev.compact_slots=1792 s128x8:dev.compact_slots=2048 > $O/ab_s128_$w.txt 2>&1; grep -E "^s128|DIFFER|rror" $O/ab_s128_$w.txt | cut -c1-330
done
