# Round 6, GPU call 10: results whose instances span more junctions than the wide variant's LDS path set holds vertices are remembered like an overflow (the next
# computation of that seed begins in the compact variant, on a lane in the big one, instead of filling the wide variant's path set first). Same-box A/B against a
# build without the rule (libsibeliaz_amd_nohint.so = -DLCB_NO_LONG_HINT) on the k = 25 shapes (test size and Gbp scale) and config 3; engine parity.
mkdir -p gpurun_out/r6h
R=$PWD; O=$R/gpurun_out/r6h
export LCB_WATCHDOG_S=300
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -x -k "find_blocks or fullsize or side_lanes or lazy or early or compact_pools or overflow" > $O/pytest_engine.log 2>&1; grep -E "passed|failed|error" $O/pytest_engine.log | tail -3
N=$R/sibeliaz_amd/libsibeliaz_amd_nohint.so
for w in primates8_test mice16_test ecoli62 primates8_scaled mice16_scaled; do
  p=3; [ $w = primates8_scaled -o $w = mice16_scaled -o $w = ecoli62 ] && p=1
  LCB_VERBOSE=1 LCB_LIB=$N timeout 900 python scripts/ab_engine.py --workload $w --passes $p warm nohint nohint2 > $O/ab_nohint_$w.txt 2>&1; grep -E "^nohint|seeds, loaded|DIFFER|rror" $O/ab_nohint_$w.txt | cut -c1-330
  LCB_VERBOSE=1 timeout 900 python scripts/ab_engine.py --workload $w --passes $p warm hint hint2 > $O/ab_hint_$w.txt 2>&1; grep -E "^hint|DIFFER|rror|too long" $O/ab_hint_$w.txt | cut -c1-330
done
