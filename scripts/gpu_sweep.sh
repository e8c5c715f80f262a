export LCB_WATCHDOG_S=60
timeout 300 python -m pytest tests -m gpu -q --timeout 100 -x 2>&1 | tail -2
for wl in ecoli10 ecoli62_small; do
    r=$(timeout 200 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f ms/step  kernel %.0f ms  launches %d' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['launches_per_step']))")
    echo "nw16+thr4 $wl :: $r"
done
