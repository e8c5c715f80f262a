# Round 6, GPU call 14: the per-seed trace of call 5 again on the final build (small pools, 8 workgroups per CU, long results remembered): primates8_scaled with the
# instrumented kernels, launch by launch - work / workgroups against longest seed.
mkdir -p gpurun_out/r6k
R=$PWD; O=$R/gpurun_out/r6k
export LCB_WATCHDOG_S=600
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
w=primates8_scaled
LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace_$w.tsv timeout 900 python scripts/ab_engine.py --workload $w --passes 1 traced > $O/traced_$w.txt 2>&1; grep -E "^traced|rror" $O/traced_$w.txt | cut -c1-400
python scripts/analyze_trace.py $O/trace_$w.tsv > $O/trace_summary_$w.txt 2>&1; cat $O/trace_summary_$w.txt
rm -f $O/trace_$w.tsv
