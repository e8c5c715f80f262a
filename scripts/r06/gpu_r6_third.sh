# Round 6, GPU call 3: which kernel variant walks the look-ahead vote which way (LCB_WALK_V2_MODES: the round-5 walk everywhere / the window-table
# walk in the compact variant only (default) / everywhere), push side as in round 5 + the footprint-slot segment check; same-box A/B against the
# round-5 library. Then where the time of a Gbp-scale k = 25 pass goes: primates8_scaled (1.24 Gbp) against primates8_test (186 Mbp), per launch.
mkdir -p gpurun_out/r6c
R=$PWD; O=$R/gpurun_out/r6c
export LCB_WATCHDOG_S=300
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "per_seed or each_kernel or find_blocks_matches or side_lanes or footprints_cover_every_read_on_gpu or event_counters" > $O/pytest_parity_subset.log 2>&1; grep -E "passed|failed|error" $O/pytest_parity_subset.log | tail -3
for w in ecoli62 primates8_test mice16_test; do
  p=3; [ $w = ecoli62 ] && p=2
  for v in r5 v1all default v2all; do
    lib=$R/sibeliaz_amd/libsibeliaz_amd_$v.so; [ $v = default ] && lib=$R/sibeliaz_amd/libsibeliaz_amd.so
    LCB_LIB=$lib timeout 300 python scripts/ab_engine.py --workload $w --passes $p warm $v > $O/ab_${v}_$w.txt 2>&1; grep -E "^$v:|DIFFER|rror" $O/ab_${v}_$w.txt | cut -c1-330
  done
done
for w in primates8_test primates8_scaled; do
  LCB_TRACE_LAUNCHES=$O/trace_$w.tsv timeout 900 python scripts/ab_engine.py --workload $w --passes 1 one > $O/scale_$w.txt 2>&1; grep -E "^one:|seeds, loaded|rror" $O/scale_$w.txt | cut -c1-400
  python scripts/analyze_trace.py $O/trace_$w.tsv > $O/trace_summary_$w.txt 2>&1; cat $O/trace_summary_$w.txt
  gzip -f $O/trace_$w.tsv
done
