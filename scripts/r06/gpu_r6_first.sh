# Round 6, GPU call 1: the look-ahead vote with window table + batched requests (lcb_vote_walk, round 6) against the round-5 library on one box:
# parity (per-seed + engine goldens, full-size hashes), same-box A/B on config 3 and both k = 25 shapes, section timers of the votes.
mkdir -p gpurun_out/r6a
R=$PWD; O=$R/gpurun_out/r6a
export LCB_WATCHDOG_S=300
git -C $R rev-parse HEAD > $O/head.txt 2>/dev/null
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -x > $O/pytest_parity_fullsize.log 2>&1; grep -E "passed|failed|error" $O/pytest_parity_fullsize.log | tail -3
R5=$R/sibeliaz_amd/libsibeliaz_amd_r5.so
for w in ecoli62 primates8_test mice16_test; do
  p=3; [ $w = ecoli62 ] && p=2
  LCB_LIB=$R5 timeout 300 python scripts/ab_engine.py --workload $w --passes $p warm r5 > $O/ab_r5_$w.txt 2>&1; grep -E "^r5:|DIFFER|rror" $O/ab_r5_$w.txt | cut -c1-200
  timeout 300 python scripts/ab_engine.py --workload $w --passes $p warm new > $O/ab_new_$w.txt 2>&1; grep -E "^new:|DIFFER|rror" $O/ab_new_$w.txt | cut -c1-200
done
prof() {
  local tag=$1; shift
  LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace_$tag.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/prof_$tag.json 2> $O/prof_$tag.err
  python scripts/vote_sections.py $O/trace_$tag.tsv $tag | tee $O/sections_$tag.txt
  rm -f $O/trace_$tag.tsv
}
prof new_ecoli62 --workload ecoli62
prof new_mice16 --workload mice16_test
LCB_LIB=$R5 prof r5_mice16 --workload mice16_test
