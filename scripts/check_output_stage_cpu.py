"""TEST-SIDE check (uses the oracle): the product output stage (lcb_generate_output) on the oracle's pre-trim blocks of a full-size bench workload,
blocks_coords.gff against the reference hash in tests/golden/fullsize.json. CPU only.   python scripts/check_output_stage_cpu.py ecoli10 config2_ecoli10_a150"""
import sys, os, json, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, sibeliaz_amd
from tests.oracle_binding import Oracle
name, key = sys.argv[1], sys.argv[2]
w = bench.ensure_workload(name)
t = time.time()
orc = Oracle(w["graph"], [w["fasta"]], w["k"], w["a"])
ob, st = orc.find_blocks(w["k"], w["b"], w["m"])
print("oracle find_blocks %.1f s, %d instances, %d blocks" % (time.time() - t, len(ob), st["blocks_found"]), flush=True)
blocks = np.zeros(len(ob), dtype=sibeliaz_amd.BLOCK_DTYPE)
for f in ("id", "chr", "start", "end"): blocks[f] = ob[f]
s = sibeliaz_amd.JunctionStorage(w["graph"], [w["fasta"]], w["k"], threads=8, abundance=w["a"])
finder = sibeliaz_amd.BlocksFinder(s, w["k"]); finder.params = sibeliaz_amd.Params.make(w["k"], w["b"], w["m"])
t = time.time()
nt, cov = finder.GenerateOutput("/tmp/fs_out_" + name, blocks=blocks, blocks_found=st["blocks_found"])
dt = time.time() - t
h = hashlib.sha256(open("/tmp/fs_out_%s/blocks_coords.gff" % name, "rb").read()).hexdigest()
ref = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize.json")))[key]
print(name, "GenerateOutput %.2f s" % dt, nt, cov, h, "EQUAL" if h == ref["gff_sha256"] else "DIFFERENT from " + ref["gff_sha256"])
