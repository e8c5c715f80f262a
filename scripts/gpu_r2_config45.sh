# configs 4 / 5 dry run at a scaled size: generation, our CLI (whole process), the reference at -t 32, md5 comparison
W=${1:-primates8_scaled}
mkdir -p gpurun_out
export LCB_WATCHDOG_S=900
R=$PWD
( time python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
w=bench.ensure_workload('$W'); print(w['fasta'], os.path.getsize(w['fasta']), os.path.getsize(w['graph']))" ) > gpurun_out/c45_$W.gen.log 2>&1
tail -5 gpurun_out/c45_$W.gen.log
D=/tmp/lcb_bench/$W
( time LCB_VERBOSE=1 $R/sibeliaz_amd/bin/sibeliaz-lcb --graph $D/graph.bin $D/genomes.fa -k 25 -b 200 -m 50 -a 150 -t 32 -o $D/cli_out --noseq ) > gpurun_out/c45_$W.ours.log 2>&1
tail -14 gpurun_out/c45_$W.ours.log
( time $R/oracle/_ref/sibeliaz-lcb-ref --graph $D/graph.bin $D/genomes.fa -k 25 -b 200 -m 50 -a 150 -t 32 -o $D/ref_out --noseq ) > gpurun_out/c45_$W.ref.log 2>&1
tail -6 gpurun_out/c45_$W.ref.log
md5sum $D/cli_out/blocks_coords.gff $D/ref_out/blocks_coords.gff | tee gpurun_out/c45_$W.md5.txt
wc -l $D/cli_out/blocks_coords.gff >> gpurun_out/c45_$W.md5.txt
