export LCB_WATCHDOG_S=120
timeout 800 python bench.py --workload ecoli62_small --steps 1 --warmup 0 > gpurun_out/bench_62small.json 2> gpurun_out/bench_62small.err
tail -3 gpurun_out/bench_62small.err; python -c "
import json; d=json.load(open('gpurun_out/bench_62small.json')); print({k:d[k] for k in ('value','ms_per_step')}, d['config']['seeds'], d['config']['blocks_found'], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('analyze_s'), d.get('cpu_baseline',{}).get('gff_md5_equal'))"
