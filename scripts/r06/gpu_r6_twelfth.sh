# Round 6, GPU call 12: larger rounds where the round launches are work-bound. At 4.19 Gbp a pass is 2 368 compact launches (median 37 ms, 90 % 290 ms) and every launch drains to
# its last block-sized seed before the commit goes on: rounds of up to 512 / 1 024 phases (lcb_hooks.round_phases; lcb_device_opts.batch raised with it so that a round stays ONE launch)
# against the default 256 on the Gbp-scale k = 25 shapes, the test shapes and config 3 (round 4 measured 1 024 phases on config 3: +20 %).
mkdir -p gpurun_out/r6i
R=$PWD; O=$R/gpurun_out/r6i
export LCB_WATCHDOG_S=600
python -c "import bench; print(bench.source_hash())" > $O/kernel_source_hash.txt; cat $O/kernel_source_hash.txt
for w in primates8_scaled mice16_scaled primates8_test ecoli62; do
  p=1; [ $w = primates8_test ] && p=2
  timeout 1200 python scripts/ab_engine.py --workload $w --passes $p base r512:round_phases=512,dev.batch=131072 r1024:round_phases=1024,dev.batch=262144 base2 > $O/ab_$w.txt 2>&1; grep -E "^base|^r512|^r1024|seeds, loaded|DIFFER|rror" $O/ab_$w.txt | cut -c1-330
done
