# second full-size parity check: the config-2 genomes with k=25, b=400, m=100 (reference timed in the same run, md5 compared)
mkdir -p gpurun_out
export LCB_WATCHDOG_S=120
timeout 800 python bench.py --workload ecoli10_k25 --steps 2 --warmup 0 > gpurun_out/bench_k25.json 2> gpurun_out/bench_k25.err
tail -3 gpurun_out/bench_k25.err; cat gpurun_out/bench_k25.json
