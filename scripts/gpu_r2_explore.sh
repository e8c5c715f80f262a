# round 2, first GPU visit: state of the round-1 build on config 2 (traced) and on config 3 at full size
mkdir -p gpurun_out
export LCB_WATCHDOG_S=600
R=$PWD
timeout 400 python -m pytest tests -m gpu -q --timeout 150 -x 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
nproc > gpurun_out/host.txt; lscpu | head -20 >> gpurun_out/host.txt; free -g >> gpurun_out/host.txt
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c2.tsv timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
cat gpurun_out/bench_c2.json | cut -c1-600
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c2s.tsv LCB_TRACE_SEEDS=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_c2s.json 2> gpurun_out/bench_c2s.err
( time python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli62')" ) > gpurun_out/gen_c3.log 2>&1
tail -4 gpurun_out/gen_c3.log
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c3.tsv timeout 1500 python bench.py --workload ecoli62 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
tail -3 gpurun_out/bench_c3.err; cat gpurun_out/bench_c3.json | cut -c1-900
D=/tmp/lcb_bench/ecoli62
md5sum $D/gpu_out/blocks_coords.gff | tee gpurun_out/c3_md5.txt
wc -l $D/gpu_out/blocks_coords.gff >> gpurun_out/c3_md5.txt
LCB_TRACE_LAUNCHES=$R/gpurun_out/trace_c3s.tsv LCB_TRACE_SEEDS=1 timeout 1500 python bench.py --workload ecoli62 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_c3s.json 2> gpurun_out/bench_c3s.err
python scripts/analyze_trace.py gpurun_out/trace_c2s.tsv > gpurun_out/trace_c2s_summary.txt 2>&1
python scripts/analyze_trace.py gpurun_out/trace_c3s.tsv > gpurun_out/trace_c3s_summary.txt 2>&1
python scripts/analyze_trace.py gpurun_out/trace_c2.tsv > gpurun_out/trace_c2_summary.txt 2>&1
python scripts/analyze_trace.py gpurun_out/trace_c3.tsv > gpurun_out/trace_c3_summary.txt 2>&1
grep -v "^#seed" gpurun_out/trace_c3s.tsv > gpurun_out/trace_c3s_launches.tsv
grep "^#seed" gpurun_out/trace_c3s.tsv | awk -F'\t' '{split($7,a,"="); if (a[2]+0 > 100000) print}' | head -20000 > gpurun_out/trace_c3s_slowseeds.tsv
rm -f gpurun_out/trace_c3s.tsv
gzip -f gpurun_out/trace_c2s.tsv
cat gpurun_out/trace_c3s_summary.txt
ls -la gpurun_out | tail -12
