# Round 4, GPU call 7: latency cuts on the seed's wavefront, each behind a -D for this A/B (variant libraries built beforehand with
# `python sibeliaz_amd/build.py variant <name> <-D...>`):
#   stock    = HEAD: one-load instance fields (voter draw, extension), hoisted field reads of a push (HBM pools), late path-set insert,
#              kernel-argument tuples split into registers of their own
#   ref      = all four off (the code of GPU call 6)
#   nofields / nohoist / nolate / nosplit = one of them off
#   stage    = stock + LCB_WALK_STAGE=1 (chunks 1-2 of a voter's window requested straight into LDS: wide, big, huge)
#   stage2   = stock + LCB_WALK_STAGE=2 (also chunk 1 in the compact variant, in LDS that is idle during a vote)
mkdir -p gpurun_out/r4g
O=gpurun_out/r4g
git rev-parse HEAD > $O/head.txt 2>/dev/null
export LCB_WATCHDOG_S=300
K="variant or footprints or overflow or per_seed or event or find_blocks_matches or resident"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "$K" > $O/pytest_stock.log 2>&1; tail -2 $O/pytest_stock.log
LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_stage2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "$K" > $O/pytest_stage2.log 2>&1; tail -2 $O/pytest_stage2.log
run() {
  local v=$1; shift
  timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline "$@" > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$v.json")); c = d["config"]
    print("$v: %.0f seeds/s, %.1f ms, kernel busy %.1f ms (sum %.1f, side %.1f), launches %s stops %s jobs %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["kernel_ms_sum_over_streams_per_step"], d["roofline"]["kernel_ms_on_side_lanes_per_step"], d["roofline"]["launches_per_step"], c["job_launches"], c["jobs"]))
except Exception as e:
    print("$v: FAILED", e); print(open("$O/$v.err").read()[-800:])
PY
}
for w in ecoli62 primates8_test; do
  run stock_$w --workload $w
  for v in ref stage stage2 nolate nohoist nofields nosplit; do LCB_LIB=$PWD/sibeliaz_amd/libsibeliaz_amd_$v.so run ${v}_$w --workload $w; done
  run stock2_$w --workload $w
done
