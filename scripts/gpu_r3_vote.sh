# Round 3, GPU call 2: the deferred path-membership vote (lcb_vote_pass) - parity suite, configs 4/5 at test size, config 3 pass with seed trace
mkdir -p gpurun_out/r3e2
O=gpurun_out/r3e2
export LCB_WATCHDOG_S=300
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 500 -x -s -k "config4 or config5" 2>&1 | grep -E "seeds,|passed|failed|rror" | tail -6
LCB_VERBOSE=1 LCB_TRACE_SEEDS=1 LCB_TRACE_LAUNCHES=$O/trace_vote.tsv timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline > $O/vote.json 2> $O/vote.err
python - <<PY
import json
d = json.load(open("$O/vote.json")); c = d["config"]
print("vote: %.0f seeds/s, %.1f ms, kernel %.1f ms, launches %s, jobs %s used %s joblaunches %s host %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"].get("launches_per_step"), c["jobs"], c["jobs_used"], c["job_launches"], c["host_ms_per_step"]))
PY
python scripts/analyze_trace.py $O/trace_vote.tsv | tee $O/vote_summary.txt
gzip -f $O/trace_vote.tsv
