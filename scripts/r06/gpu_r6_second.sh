# Round 6, GPU call 2: calibration of the dependent steps of one wavefront (scripts/r06/microbench.hip), then the push-side changes (home-key
# prefetch, wavefront-scope ordering of the path-set insert, `used` words of a predicted view in two round trips, next edge's dependent
# loads requested ahead) with the cheaper stage A of the vote: parity subset, same-box A/B against the round-5 library, push sections.
mkdir -p gpurun_out/r6b
R=$PWD; O=$R/gpurun_out/r6b
export LCB_WATCHDOG_S=300
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
timeout 120 scripts/r06/microbench 1 > $O/microbench_1.txt 2>&1; cat $O/microbench_1.txt
timeout 120 scripts/r06/microbench 1280 > $O/microbench_1280.txt 2>&1; cat $O/microbench_1280.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "per_seed or each_kernel or find_blocks_matches or side_lanes or footprints_cover_every_read_on_gpu or event_counters" > $O/pytest_parity_subset.log 2>&1; grep -E "passed|failed|error" $O/pytest_parity_subset.log | tail -3
R5=$R/sibeliaz_amd/libsibeliaz_amd_r5.so
NA=$R/sibeliaz_amd/libsibeliaz_amd_noahead.so
for w in ecoli62 primates8_test mice16_test; do
  p=3; [ $w = ecoli62 ] && p=2
  LCB_LIB=$R5 timeout 300 python scripts/ab_engine.py --workload $w --passes $p warm r5 > $O/ab_r5_$w.txt 2>&1; grep -E "^r5:|DIFFER|rror" $O/ab_r5_$w.txt | cut -c1-200
  timeout 300 python scripts/ab_engine.py --workload $w --passes $p warm new > $O/ab_new_$w.txt 2>&1; grep -E "^new:|DIFFER|rror" $O/ab_new_$w.txt | cut -c1-200
  LCB_LIB=$NA timeout 300 python scripts/ab_engine.py --workload $w --passes $p warm noahead > $O/ab_noahead_$w.txt 2>&1; grep -E "^noahead:|DIFFER|rror" $O/ab_noahead_$w.txt | cut -c1-200
done
prof() {
  local tag=$1; local kind=$2; shift 2
  LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace_$tag.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/prof_$tag.json 2> $O/prof_$tag.err
  python scripts/vote_sections.py $O/trace_$tag.tsv $tag $kind | tee $O/sections_$tag.txt
  rm -f $O/trace_$tag.tsv
}
prof vote_new_mice16 vote --workload mice16_test
LCB_LIB=$R/sibeliaz_amd/libsibeliaz_amd_profpush.so prof push_new_mice16 push --workload mice16_test
LCB_LIB=$R/sibeliaz_amd/libsibeliaz_amd_profpush.so prof push_new_ecoli62 push --workload ecoli62
