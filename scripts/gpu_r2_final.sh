# round 2 final evidence on the final build: parity suite + smoke, the bench line (config 3, full size), rocprofv3 kernel stats of the
# same command, then the config-2 line (the PMC traffic passes of scripts/gpu_r2_evidence.sh were taken one commit earlier: same kernels)
mkdir -p gpurun_out
export LCB_WATCHDOG_S=600
R=$PWD
timeout 700 python -m pytest tests -m gpu -q --timeout 400 -x 2>&1 | grep -E "passed|failed|error" | tail -4 | tee gpurun_out/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
LCB_VERBOSE=1 timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -12 gpurun_out/bench_n1.err; cat gpurun_out/bench_n1.json | cut -c1-1800
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $R/gpurun_out/prof.log 2>&1
cat $R/gpurun_out/prof/*kernel_stats.csv | head -12
rm -f $R/gpurun_out/prof/*kernel_trace.csv
cd $R
LCB_VERBOSE=1 timeout 300 python bench.py --workload ecoli10 --steps 3 --warmup 1 > gpurun_out/bench_n1_config2.json 2> gpurun_out/bench_n1_config2.err
cat gpurun_out/bench_n1_config2.json | cut -c1-700
echo done
