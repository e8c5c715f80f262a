# Measurements on the MI355X: per-launch trace, bench line, rocprofv3 kernel trace (CSV).
mkdir -p gpurun_out
export LCB_WATCHDOG_S=120
R=$PWD
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10')"
D=/tmp/lcb_bench/ecoli10
( time LCB_TRACE_LAUNCHES=$R/gpurun_out/launches.tsv LCB_VERBOSE=1 timeout 600 sibeliaz_amd/bin/sibeliaz-lcb --graph $D/graph.bin $D/genomes.fa -k 15 -b 200 -m 50 -a 150 -t 32 -o $D/cli_out --noseq ) > gpurun_out/cli_e10.log 2>&1
tail -9 gpurun_out/cli_e10.log
md5sum $D/cli_out/blocks_coords.gff | tee -a gpurun_out/cli_e10.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_first.json 2> gpurun_out/bench_first.err
tail -3 gpurun_out/bench_first.err; cat gpurun_out/bench_first.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o r1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_r1.log 2>&1
ls -R $R/gpurun_out/prof_r1 | head; cat $R/gpurun_out/prof_r1/*kernel_stats.csv | head -8
rm -f $R/gpurun_out/prof_r1/*kernel_trace.csv
