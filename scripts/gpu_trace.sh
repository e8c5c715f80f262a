mkdir -p gpurun_out
export LCB_WATCHDOG_S=120
R=$PWD
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10')"
D=/tmp/lcb_bench/ecoli10
( time LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$R/gpurun_out/launches2.tsv LCB_VERBOSE=1 timeout 600 sibeliaz_amd/bin/sibeliaz-lcb --graph $D/graph.bin $D/genomes.fa -k 15 -b 200 -m 50 -a 150 -t 32 -o $D/cli_out --noseq ) > gpurun_out/cli_e10b.log 2>&1
tail -9 gpurun_out/cli_e10b.log
gzip -f gpurun_out/launches2.tsv
