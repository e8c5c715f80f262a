"""Stepwise GPU bring-up: each step runs in its own subprocess with a short timeout and prints timings."""
import gzip, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STEP = r'''
import sys, time, os, gzip
sys.path.insert(0, %(root)r)
import numpy as np
import sibeliaz_amd
from tests.oracle_binding import Oracle
t0 = time.time()
case = %(case)r; n = %(n)d; k, b, m, a = %(k)d, %(b)d, %(m)d, %(a)d
d = os.path.join(%(root)r, "tests", "golden", case)
fa, gr = "/tmp/dbg_%%s.fa" %% case, "/tmp/dbg_%%s.bin" %% case
for s, t in (("genomes.fa.gz", fa), ("graph.bin.gz", gr)):
    open(t, "wb").write(gzip.open(os.path.join(d, s)).read())
st = sibeliaz_amd.JunctionStorage(gr, [fa], k, 4, a)
p = sibeliaz_amd.Params.make(k, b=b, m=m)
dev = sibeliaz_amd.Device(st, p, 0)
print("setup %%.2fs" %% (time.time() - t0), flush=True)
seeds = st.seeds(4)
sub = seeds[%(start)d:%(start)d + n]
t1 = time.time()
off, inst, score, _ = dev.process_seeds(sub)
t2 = time.time()
ms, launches = dev.kernel_time()
print("process %%d seeds: wall %%.3fs kernel %%.3f ms launches %%d insts %%d" %% (len(sub), t2 - t1, ms, launches, len(inst)), flush=True)
orc = Oracle(gr, [fa], k, a)
bad = 0
for i in range(len(sub)):
    ref, rs = orc.process_seed(k, b, m, int(sub["vid"][i]), int(sub["ch"][i]))
    got = [(int(x["chr"]), int(x["front_idx"]), int(x["back_idx"]), int(x["positive"]) != 0) for x in inst[int(off[i]):int(off[i+1])]]
    if got != ref or int(score[i]) != rs:
        bad += 1
        if bad < 3: print("DIFF seed", i, got[:4], int(score[i]), "|", ref[:4], rs, flush=True)
print("mismatches", bad, flush=True)
'''

def run(case, n, start, k, b, m, a, timeout):
    code = STEP % dict(root=ROOT, case=case, n=n, start=start, k=k, b=b, m=m, a=a)
    t = time.time()
    try:
        env = dict(os.environ, LCB_DEBUG="1", LCB_WATCHDOG_S=os.environ.get("LCB_WATCHDOG_S", "15"))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env)
        print("[%s n=%d start=%d] rc=%d %.1fs\n%s%s" % (case, n, start, r.returncode, time.time() - t, r.stdout, r.stderr[-1500:]), flush=True)
        return r.returncode == 0 and "mismatches 0" in r.stdout
    except subprocess.TimeoutExpired as e:
        print("[%s n=%d start=%d] TIMEOUT after %ds\n%s" % (case, n, start, timeout, (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else e.stdout), flush=True)
        return False

if __name__ == "__main__":
    steps = [("collinear6", 64, 0, 15, 200, 50, 150, 60), ("collinear6", 512, 0, 15, 200, 50, 150, 60), ("collinear6", 3373, 0, 15, 200, 50, 150, 90)]
    for s in steps:
        if not run(*s):
            break
