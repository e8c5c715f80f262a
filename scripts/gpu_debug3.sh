export LCB_WATCHDOG_S=12
LCB_SLOTS=1 LCB_MEDIUM_SLOTS=1 LCB_BIG_SLOTS=1 LCB_LIB=$PWD/sibeliaz_amd/variants/plain_nw1.so timeout 60 python - <<'PY' 2>&1 | grep -v "^  File\|Extension modules" | tail -12
import sys, os, gzip, time
sys.path.insert(0, os.getcwd())
import sibeliaz_amd
d = "tests/golden/collinear6"
fa, gr = "/tmp/c6.fa", "/tmp/c6.bin"
for s, t in (("genomes.fa.gz", fa), ("graph.bin.gz", gr)):
    open(t, "wb").write(gzip.open(os.path.join(d, s)).read())
st = sibeliaz_amd.JunctionStorage(gr, [fa], 15, 4, 150)
dev = sibeliaz_amd.Device(st, sibeliaz_amd.Params.make(15, 200, 50), 0)
seeds = st.seeds(4)
t = time.time()
try:
    off, inst, sc, _ = dev.process_seeds(seeds[:400])
    print("completed")
except Exception as e:
    print("EXC", e)
PY
