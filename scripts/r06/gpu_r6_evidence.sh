# Round 6 evidence on ONE box for the final build (commit in evidence_head.txt, hash of the kernel / engine sources in kernel_source_hash.txt):
#   tests  the whole `pytest -m gpu` suite + smoke
#   prof   rocprofv3 --kernel-trace --stats of the bench command (config 3, one pass), the two PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) and one
#          SQ pass (wave cycles, busy cycles, VALU / SALU / LDS instructions, issue stalls) - per kernel instantiation, into profiles/r06/pmc_traffic.json
#   bench  the bench line of config 3 with the bounded reference legs and the LIVE k = 25 secondary lines; config 3 with a = 868
#   big    configs 4 / 5 at 1.2 / 1.0 Gbp and at 4.2 / 4.1 Gbp against the reference's GFF hashes (tests/golden/fullsize_scaled.json), engine breakdown of each
# Outputs under gpurun_out/r6ev (copied into profiles/r06 afterwards). Every step has its own time limit.
MODE=${1:-all}
mkdir -p gpurun_out/r6ev
R=$PWD; O=$R/gpurun_out/r6ev
export LCB_WATCHDOG_S=600
cp $R/.evidence_head $O/evidence_head.txt 2>/dev/null; cat $O/evidence_head.txt
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
has() { case ",$MODE," in *,all,*|*,$1,*) return 0;; esac; return 1; }
if has tests; then
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|skipped" $O/pytest_gpu.log | tail -3
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
fi
if has prof; then
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-roofline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- $B > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats.csv \; ; head -8 $O/rocprofv3_kernel_stats.csv | cut -c1-200
find $O/prof -name "*kernel_trace.csv" -delete
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $B > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o p -- $B > $O/pmc_write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o p -- $B > $O/pmc_sq.log 2>&1
cd $R
python scripts/r06/pmc_summary.py $O
fi
if has bench; then
LCB_VERBOSE=1 timeout 1500 python bench.py --steps 3 --warmup 1 --cpu-baseline-budget 330 > $O/bench_n1.json 2> $O/bench_n1.err
grep "lcb engine" $O/bench_n1.err | tail -2 | cut -c1-400; cut -c1-1500 $O/bench_n1.json
LCB_VERBOSE=1 timeout 400 python bench.py --workload ecoli62_a868 --steps 2 --warmup 1 --no-roofline --no-secondary > $O/bench_n1_ecoli62_a868.json 2> $O/bench_n1_ecoli62_a868.err; cut -c1-300 $O/bench_n1_ecoli62_a868.json
fi
big() {
  LCB_VERBOSE=1 LCB_TRACE_LAUNCHES=$O/trace_$1.tsv timeout $2 python scripts/check_fullsize_scaled.py $1 > $O/fullsize_$1.log 2>&1; grep -E "equal|FAILED|phase loop|lcb engine" $O/fullsize_$1.log | cut -c1-500
  python scripts/analyze_trace.py $O/trace_$1.tsv > $O/trace_summary_$1.txt 2>&1; head -6 $O/trace_summary_$1.txt; rm -f $O/trace_$1.tsv
}
if has scaled; then big config4_primates8_scaled 900; big config5_mice16_scaled 900; fi
if has big4; then big config4_primates8_4g_scaled 2400; fi
if has big5; then big config5_mice16_4g_scaled 2400; fi
