# engine-knob sweep on the config-2 workload (CLI phase-loop time; md5 must stay the same)
mkdir -p gpurun_out
export LCB_WATCHDOG_S=60
python -c "
import sys, os
sys.path.insert(0, os.getcwd())
import bench
bench.ensure_workload('ecoli10')"
D=/tmp/lcb_bench/ecoli10
run() {
  r=$( ( env "$@" LCB_VERBOSE=1 timeout 120 sibeliaz_amd/bin/sibeliaz-lcb --graph $D/graph.bin $D/genomes.fa -k 15 -b 200 -m 50 -a 150 -t 32 -o $D/cli_out --noseq ) 2>&1 | grep "^lcb:" | sed 's/.*conflicts=[0-9]* //' )
  echo "$* | $r | $(md5sum < $D/cli_out/blocks_coords.gff | cut -c1-8)"
}
run LCB_ROUND_PHASES=64
run LCB_VIEWS=256
run LCB_EAGER_PHASES=256 LCB_ROUND_PHASES=256
run LCB_EAGER_PHASES=256 LCB_ROUND_PHASES=256 LCB_VIEWS=256
run LCB_EAGER_PHASES=1024 LCB_ROUND_PHASES=1024 LCB_VIEWS=256
run LCB_EAGER_PHASES=1024 LCB_ROUND_PHASES=1024 LCB_VIEWS=256 LCB_ROUND_FIXED=1
