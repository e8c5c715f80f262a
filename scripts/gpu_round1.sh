# Round-1 GPU evidence: parity tests, smoke, bench line (N=1), 2-rank plumbing check on one GPU (gloo), rocprofv3 kernel trace.
mkdir -p gpurun_out
export LCB_WATCHDOG_S=180
R=$PWD
timeout 600 python -m pytest tests -m gpu -q --timeout 150 -x 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
tail -2 gpurun_out/bench_r1.err; cat gpurun_out/bench_r1.json
LCB_BENCH_BACKEND=gloo LCB_BENCH_SAME_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err
tail -3 gpurun_out/bench_2rank_gloo.err; cat gpurun_out/bench_2rank_gloo.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o r1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_r1.log 2>&1
cat $R/gpurun_out/prof_r1/*kernel_stats.csv | head -8
rm -f $R/gpurun_out/prof_r1/*kernel_trace.csv
