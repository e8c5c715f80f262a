# full GPU suite + bench on a library variant next to the shipped one
mkdir -p gpurun_out
export LCB_WATCHDOG_S=20
V=$PWD/sibeliaz_amd/variants/fr0_v0.so
LCB_LIB=$V timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 100 -x -k "not cli" 2>&1 | tail -4
for i in 1 2; do
LCB_LIB=$V python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fr0', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fr1', d['value'], d['ms_per_step'])"
done
