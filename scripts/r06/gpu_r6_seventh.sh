# Round 6, GPU call 7: the compact variant's pools chosen by the input (lcb_device_opts.compact_pools 0: 128 instances / 512 vote slots and 8 workgroups per
# CU where a vertex has at most 20 occurrences on average, given up if > 3 % of the live seeds outgrow them; 1 = 256 / 1 024, 5 per CU as until now; 2 = small).
# Parity of both instantiations (goldens per seed, engine, full-size hashes), then same-box A/B auto / large / small on configs 2, 3 and the k = 25 shapes.
mkdir -p gpurun_out/r6g
R=$PWD; O=$R/gpurun_out/r6g
export LCB_WATCHDOG_S=300
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -x -k "compact_pools or sparse_rounds or per_seed or each_kernel or find_blocks_matches or fullsize or overflow or footprints_cover" > $O/pytest_pools.log 2>&1; grep -E "passed|failed|error" $O/pytest_pools.log | tail -3
for w in ecoli10 ecoli62 primates8_test mice16_test mice16_scaled primates8_scaled; do
  p=2; [ $w = primates8_scaled -o $w = mice16_scaled -o $w = ecoli62 ] && p=1
  LCB_VERBOSE=1 timeout 900 python scripts/ab_engine.py --workload $w --passes $p auto large:dev.compact_pools=1 small:dev.compact_pools=2 > $O/ab_$w.txt 2>&1; grep -E "^auto:|^large:|^small:|seeds, loaded|DIFFER|rror|compact pools" $O/ab_$w.txt | cut -c1-330
done
