# round 2 iteration: parity suite, then config 2 / config 3 passes with launch traces
mkdir -p gpurun_out
export LCB_WATCHDOG_S=300
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c2.tsv timeout 600 python bench.py --workload ecoli10 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
tail -2 gpurun_out/bench_c2.err; cat gpurun_out/bench_c2.json | cut -c1-400
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c2s.tsv LCB_TRACE_SEEDS=1 timeout 600 python bench.py --workload ecoli10 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_c2s.json 2> gpurun_out/bench_c2s.err
python scripts/analyze_trace.py gpurun_out/trace_c2s.tsv > gpurun_out/trace_c2s_summary.txt 2>&1; cat gpurun_out/trace_c2s_summary.txt
if [ "$1" != "c2only" ]; then
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli62')" > gpurun_out/gen_c3.log 2>&1
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c3.tsv timeout 1500 python bench.py --workload ecoli62 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
tail -3 gpurun_out/bench_c3.err; cat gpurun_out/bench_c3.json | cut -c1-300
md5sum /tmp/lcb_bench/ecoli62/gpu_out/blocks_coords.gff | tee gpurun_out/c3_md5.txt
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c3s.tsv LCB_TRACE_SEEDS=1 timeout 1500 python bench.py --workload ecoli62 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_c3s.json 2> gpurun_out/bench_c3s.err
python scripts/analyze_trace.py gpurun_out/trace_c3s.tsv > gpurun_out/trace_c3s_summary.txt 2>&1
grep -v "^#seed" gpurun_out/trace_c3s.tsv > gpurun_out/trace_c3s_launches.tsv
rm -f gpurun_out/trace_c3s.tsv
cat gpurun_out/trace_c3s_summary.txt
fi
gzip -f gpurun_out/trace_c2s.tsv
