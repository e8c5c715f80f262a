# Round 6, GPU call 4: sparse speculative launches + host-settled dead seeds (lcb_hooks.sparse_rounds, engine.cpp) on the MI355X:
# engine parity (goldens + the five full-size reference hashes, sparse rounds on by default), same-box A/B sparse on / off on config 3 and both
# k = 25 shapes, then the Gbp-scale k = 25 shapes (where the round launches of the compact variant are 2/3 of the pass) with both settings.
mkdir -p gpurun_out/r6d
R=$PWD; O=$R/gpurun_out/r6d
export LCB_WATCHDOG_S=300
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -x -k "find_blocks or fullsize or side_lanes or lazy or early or cli or gpus or smoke" > $O/pytest_engine.log 2>&1; grep -E "passed|failed|error" $O/pytest_engine.log | tail -3
for w in primates8_test mice16_test ecoli62; do
  p=3; [ $w = ecoli62 ] && p=2
  timeout 600 python scripts/ab_engine.py --workload $w --passes $p warm sparse nosparse:sparse_rounds=-1 > $O/ab_$w.txt 2>&1; grep -E "^sparse:|^nosparse:|DIFFER|rror" $O/ab_$w.txt | cut -c1-400
done
for w in primates8_scaled mice16_scaled; do
  timeout 900 python scripts/ab_engine.py --workload $w --passes 1 sparse nosparse:sparse_rounds=-1 > $O/scale_$w.txt 2>&1; grep -E "^sparse:|^nosparse:|seeds, loaded|DIFFER|rror" $O/scale_$w.txt | cut -c1-400
  c=config4_$w; [ $w = mice16_scaled ] && c=config5_$w
  python scripts/check_fullsize_scaled.py $c > $O/hash_$w.txt 2>&1; tail -2 $O/hash_$w.txt
done
