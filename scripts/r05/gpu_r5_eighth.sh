# Round 5, GPU call 8: a cap on the big-variant jobs of one background batch (lcb_device_opts.side_big_cap) - does it give config 3 what a job
# cap of 512 gave it (fewer heavy void jobs in the way of the commit's own launches) without costing the k = 25 shapes, which have few such jobs?
mkdir -p gpurun_out/r5h
R=$PWD; O=$R/gpurun_out/r5h
export LCB_WATCHDOG_S=300
cp $R/.evidence_head $O/head.txt 2>/dev/null
timeout 420 python scripts/ab_engine.py --workload ecoli62 --passes 1 warm base bigcap96:dev.side_big_cap=96 bigcap32:dev.side_big_cap=32 base_again bigcap96_again:dev.side_big_cap=96 > $O/ab_ecoli62.txt 2>&1; grep -E "seeds/s|DIFFER|rror" $O/ab_ecoli62.txt | cut -c1-330
timeout 120 python scripts/ab_engine.py --workload primates8_test --passes 2 warm base bigcap96:dev.side_big_cap=96 bigcap32:dev.side_big_cap=32 > $O/ab_primates8_test.txt 2>&1; grep -E "seeds/s|DIFFER|rror" $O/ab_primates8_test.txt | cut -c1-330
