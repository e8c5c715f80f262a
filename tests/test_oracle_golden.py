"""Pins the CPU oracle (oracle/lcb_oracle.c) against outputs of the REAL reference, committed under
tests/golden/ by tests/golden/make_golden.py: blocks_coords.gff, the sorted seed list, the pre-trim
block instances and the per-seed Process() results (final and initial `used` state)."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def test_oracle_matches_reference_goldens(built, case, tmp_path):
    out, dump = str(tmp_path / "out"), str(tmp_path / "dump")
    subprocess.check_call([os.path.join(ROOT, "oracle", "lcb_oracle"), "--graph", case.graph, case.fasta, "-k", str(case.k),
                           "-b", str(case.b), "-m", str(case.m), "-a", str(case.a), "-o", out, "--dump", dump],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert open(os.path.join(out, "blocks_coords.gff")).read() == case.golden("ref.gff")
    assert open(os.path.join(dump, "pretrim.tsv")).read() == case.golden("pretrim.tsv")
    assert open(os.path.join(dump, "summary.txt")).read() == case.golden("summary.txt")
    assert open(os.path.join(dump, "seeds_final.tsv")).read() == case.golden("seeds_final.tsv")
    for fn in ("bundles.tsv", "seeds_init.tsv"):
        assert sha(os.path.join(dump, fn)) == case.meta["sha256"][fn], fn
