# Round 4, GPU call 6: compute units reserved for the synchronous launches (a CU mask on the side lanes' streams instead of a low priority)
mkdir -p gpurun_out/r4f
O=gpurun_out/r4f
git rev-parse HEAD > $O/head.txt 2>/dev/null
export LCB_WATCHDOG_S=300
V="base cus224:dev.side_cus=224 cus192:dev.side_cus=192 cus128:dev.side_cus=128 base_again"
for w in ecoli62 primates8_test mice16_test; do
  timeout 900 python scripts/ab_engine.py --workload $w $V > $O/ab_$w.txt 2> $O/ab_$w.err; cat $O/ab_$w.txt
done
