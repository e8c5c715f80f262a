# which of {blocking sync, flight recorder} matters? 4 combos, 25 s cap each
for combo in "0 0" "0 15" "1 0" "1 15"; do
  set -- $combo
  echo "=== LCB_DEBUG=$1 LCB_WATCHDOG_S=$2"
  LCB_DEBUG=$1 LCB_WATCHDOG_S=$2 timeout 25 python - <<'PY'
import sys, os, gzip, time
sys.path.insert(0, os.getcwd())
import sibeliaz_amd
d = "tests/golden/inv_k25"
fa, gr = "/tmp/c.fa", "/tmp/c.bin"
for s, t in (("genomes.fa.gz", fa), ("graph.bin.gz", gr)):
    open(t, "wb").write(gzip.open(os.path.join(d, s)).read())
st = sibeliaz_amd.JunctionStorage(gr, [fa], 25, 4, 150)
dev = sibeliaz_amd.Device(st, sibeliaz_amd.Params.make(25, 200, 200), 0)
seeds = st.seeds(4)
for n in (1, 64, 1791):
    t = time.time(); off, inst, sc, _ = dev.process_seeds(seeds[:n]); print(n, "seeds ok in %.3fs" % (time.time() - t), dev.kernel_time(), flush=True)
PY
  echo "rc=$?"
done
