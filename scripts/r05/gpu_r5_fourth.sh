# Round 5, GPU call 4: the adaptive job cap (default build: `base`) against the fixed caps it moves between, on config 3 and both k = 25 shapes.
mkdir -p gpurun_out/r5d
R=$PWD; O=$R/gpurun_out/r5d
export LCB_WATCHDOG_S=300 LCB_VERBOSE=1
cp $R/.evidence_head $O/head.txt 2>/dev/null
timeout 700 python scripts/ab_engine.py --workload ecoli62 --passes 1 warm base jobs1280:max_jobs=1280 jobs512:max_jobs=512 base_again jobs1280_again:max_jobs=1280 > $O/ab_ecoli62.txt 2>&1; grep -E "seeds/s|DIFFER|rror" $O/ab_ecoli62.txt | cut -c1-330
for w in primates8_test mice16_test; do
timeout 300 python scripts/ab_engine.py --workload $w --passes 2 warm base jobs1280:max_jobs=1280 jobs512:max_jobs=512 > $O/ab_$w.txt 2>&1; grep -E "seeds/s|DIFFER|rror" $O/ab_$w.txt | cut -c1-330
done
