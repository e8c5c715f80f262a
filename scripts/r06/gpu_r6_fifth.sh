# Round 6, GPU call 5: host-settled dead seeds alone (lcb_hooks.sparse_rounds 0, the new default) against neither (-1) and against sparse launches (1) on one box;
# then where the time of a Gbp-scale k = 25 pass goes, per seed: primates8_scaled with the instrumented kernels (LCB_TRACE_SEEDS=1: ticks, pushes, votes, pool
# of every seed above 20 us), launch by launch - is a round launch as long as its longest seed or as the work of its busiest workgroup?
mkdir -p gpurun_out/r6e
R=$PWD; O=$R/gpurun_out/r6e
export LCB_WATCHDOG_S=300
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "find_blocks or side_lanes or lazy or early" > $O/pytest_engine.log 2>&1; grep -E "passed|failed|error" $O/pytest_engine.log | tail -3
for w in primates8_test mice16_test ecoli62; do
  p=3; [ $w = ecoli62 ] && p=2
  timeout 600 python scripts/ab_engine.py --workload $w --passes $p warm screen neither:sparse_rounds=-1 screen2 neither2:sparse_rounds=-1 > $O/ab_$w.txt 2>&1; grep -E "^screen|^neither|DIFFER|rror" $O/ab_$w.txt | cut -c1-400
done
for w in primates8_scaled; do
  timeout 900 python scripts/ab_engine.py --workload $w --passes 1 screen neither:sparse_rounds=-1 > $O/scale_$w.txt 2>&1; grep -E "^screen|^neither|seeds, loaded|DIFFER|rror" $O/scale_$w.txt | cut -c1-400
  LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace_$w.tsv timeout 900 python scripts/ab_engine.py --workload $w --passes 1 traced > $O/traced_$w.txt 2>&1; grep -E "^traced|rror" $O/traced_$w.txt | cut -c1-400
  python scripts/analyze_trace.py $O/trace_$w.tsv > $O/trace_summary_$w.txt 2>&1; cat $O/trace_summary_$w.txt
  ls -la $O/trace_$w.tsv; rm -f $O/trace_$w.tsv     # (hundreds of MB: the summary is what travels back)
done
