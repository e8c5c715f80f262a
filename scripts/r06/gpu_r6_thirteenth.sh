# Round 6, GPU call 13: the compact variants without their vote helper (build -DLCB_NW_COMPACT=1 = libsibeliaz_amd_nw1.so: one wavefront per workgroup) where a pass is bound by the
# work of its round launches - at 8 workgroups per CU the helpers are half of the resident wavefronts; 8 / 9 (LDS-bound) / 12 workgroups per CU against the shipped library.
mkdir -p gpurun_out/r6j
R=$PWD; O=$R/gpurun_out/r6j
export LCB_WATCHDOG_S=600
N=$R/sibeliaz_amd/libsibeliaz_amd_nw1.so
LCB_LIB=$N timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "per_seed or compact_pools or find_blocks_matches" > $O/pytest_nw1.log 2>&1; grep -E "passed|failed|error" $O/pytest_nw1.log | tail -2
for w in primates8_scaled mice16_scaled primates8_test; do
  p=1; [ $w = primates8_test ] && p=2
  timeout 900 python scripts/ab_engine.py --workload $w --passes $p base base2 > $O/ab_base_$w.txt 2>&1; grep -E "^base|seeds, loaded|DIFFER|rror" $O/ab_base_$w.txt | cut -c1-330
  LCB_LIB=$N timeout 900 python scripts/ab_engine.py --workload $w --passes $p nw1x8 nw1x9:dev.compact_slots=2304 nw1x12:dev.compact_slots=3072 > $O/ab_nw1_$w.txt 2>&1; grep -E "^nw1|DIFFER|rror" $O/ab_nw1_$w.txt | cut -c1-330
done
