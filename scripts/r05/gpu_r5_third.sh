# Round 5, GPU call 3: same-box A/B of the engine's speculation knobs on top of the lazy round tails (job cap, planning horizon, number of
# side lanes), config 3 and config 4's shape; the default build is `base`.
mkdir -p gpurun_out/r5c
R=$PWD; O=$R/gpurun_out/r5c
export LCB_WATCHDOG_S=300 LCB_VERBOSE=1
cp $R/.evidence_head $O/head.txt 2>/dev/null
timeout 700 python scripts/ab_engine.py --workload ecoli62 --passes 1 warm base jobs512:max_jobs=512 jobs2560:max_jobs=2560 eager32:eager_phases=32 lanes8:dev.side_lanes=8 base_again > $O/ab_ecoli62.txt 2>&1; grep -E "seeds/s|DIFFER|rror" $O/ab_ecoli62.txt | cut -c1-330
timeout 300 python scripts/ab_engine.py --workload primates8_test --passes 2 warm base jobs512:max_jobs=512 jobs2560:max_jobs=2560 eager32:eager_phases=32 lanes8:dev.side_lanes=8 > $O/ab_primates8_test.txt 2>&1; grep -E "seeds/s|DIFFER|rror" $O/ab_primates8_test.txt | cut -c1-330
