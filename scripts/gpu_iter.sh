# one optimisation iteration on the GPU: parity tests, then a traced CLI run on the bench workload
mkdir -p gpurun_out
export LCB_WATCHDOG_S=40
R=$PWD
timeout 400 python -m pytest tests -m gpu -q --timeout 100 -x 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10')"
D=/tmp/lcb_bench/ecoli10
( time LCB_VERBOSE=1 timeout 300 sibeliaz_amd/bin/sibeliaz-lcb --graph $D/graph.bin $D/genomes.fa -k 15 -b 200 -m 50 -a 150 -t 32 -o $D/cli_out --noseq ) > gpurun_out/cli_iter.log 2>&1
tail -8 gpurun_out/cli_iter.log; md5sum $D/cli_out/blocks_coords.gff
