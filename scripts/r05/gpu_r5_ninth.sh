# Round 5, GPU call 9 (the last minutes of the budget): the build with the big-job cap of a background batch as the default - engine-level parity
# tests on the goldens, the bench line of config 3 (no reference legs: those are in the evidence run), and the rocprofv3 kernel table of that command.
mkdir -p gpurun_out/r5i
R=$PWD; O=$R/gpurun_out/r5i
export LCB_WATCHDOG_S=300
cp $R/.evidence_head $O/head.txt 2>/dev/null
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 120 -x -k "find_blocks_matches_reference or side_lanes or early_critical" > $O/pytest_engine_sample.log 2>&1; tail -2 $O/pytest_engine_sample.log
LCB_VERBOSE=1 timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > $O/bench_n1_final.json 2> $O/bench_n1_final.err; grep "lcb engine" $O/bench_n1_final.err | tail -1 | cut -c1-300; cut -c1-260 $O/bench_n1_final.json
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_final.csv \; ; head -5 $O/rocprofv3_kernel_stats_final.csv | cut -c1-60,225-330
find $O/prof -name "*kernel_trace.csv" -delete
