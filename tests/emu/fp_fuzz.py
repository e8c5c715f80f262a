"""TEST-ONLY: footprint-completeness campaign. For random inputs (the generator of fuzz.py) every seed's footprint is checked with
EMU_FP_CHECK: all unused positions outside it are set to used and the oracle must still reproduce the kernel's result.

    python tests/emu/fp_fuzz.py <emu_check binary> <first case> <number of cases>
"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz
exe = sys.argv[1]; first = int(sys.argv[2]); n = int(sys.argv[3])
work = "/tmp/ana/fpfuzz_%d" % first; os.makedirs(work, exist_ok=True)
bad = 0
for i in range(first, first + n):
    synth, (k, b, m, a), runs, _ = fuzz.case_params(i)
    fa = work + "/g.fa"; gr = work + "/g.bin"
    try:
        subprocess.check_call([fuzz.BIN + "/lcb-synth"] + synth + ["-o", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([fuzz.BIN + "/lcb-mkgraph", "-k", str(k), "-o", gr, fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception as e:
        print("case", i, "generation failed"); continue
    for mode in ("seeds-init", "seeds-final"):
        try:
            r = subprocess.run([exe, gr, fa, str(k), str(b), str(m), str(a), mode], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, EMU_FP_CHECK="1", EMU_NOSTATS="1", EMU_LIMIT="1500"))
        except subprocess.TimeoutExpired:
            print("case", i, mode, "timeout"); continue
        fails = [l for l in r.stderr.splitlines() if "FAIL" in l or "MISMATCH" in l]
        if r.returncode != 0 or fails:
            bad += 1
            print("case", i, mode, "k b m a", k, b, m, a, "rc", r.returncode, fails[:2], flush=True)
print("done", first, n, "bad", bad)
