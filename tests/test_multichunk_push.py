"""Regression for a push whose vertex has more than 64 occurrences (found by tests/emu/fuzz.py): an occurrence handled by the second
64-lane chunk must not extend an instance that the first chunk just created at the pushed vertex
(`inst->Back().GetVertexId() != vertex`, path.h:541 / :472). Expected rows come from the reference itself (ref_dump)."""
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "multichunk_push")
BIN = os.path.join(ROOT, "sibeliaz_amd", "bin")


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


@pytest.fixture(scope="module")
def mc(built, tmp_path_factory):
    meta = json.load(open(os.path.join(GOLD, "golden.json")))
    d = str(tmp_path_factory.mktemp("multichunk"))
    fa, gr = os.path.join(d, "g.fa"), os.path.join(d, "g.bin")
    subprocess.check_call([os.path.join(BIN, "lcb-synth"), "-o", fa] + meta["synth"].split(), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(BIN, "lcb-mkgraph"), "-k", str(meta["k"]), "-o", gr, fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert _sha(fa) == meta["sha256"]["genomes.fa"] and _sha(gr) == meta["sha256"]["graph.bin"], "the generators no longer reproduce the fixture"
    rows = {}
    for ln in open(os.path.join(GOLD, "seeds_init.ref.tsv")).read().splitlines():
        f = ln.split("\t")
        inst = []
        for x in f[3:]:
            s, c, fr, bk = x.split(",")
            inst.append((int(c), int(fr), int(bk), s == "+"))
        assert len(inst) == int(f[2])
        rows[int(f[0])] = (int(f[1]), inst)
    return dict(meta, fasta=fa, graph=gr, rows=rows, dir=d)


def test_oracle_matches_reference_rows(mc):
    from tests.oracle_binding import Oracle
    orc = Oracle(mc["graph"], [mc["fasta"]], mc["k"], mc["a"])
    seeds = orc.seeds()
    for idx, (score, inst) in mc["rows"].items():
        got, got_score = orc.process_seed(mc["k"], mc["b"], mc["m"], int(seeds[idx][0]), int(seeds[idx][1]))
        assert (got_score, got) == (score, inst), "seed %d" % idx


@pytest.mark.parametrize("idx,env", [(1310, {}), (1420, {"EMU_NOSTATS": "1"})])
def test_kernel_logic_under_emulator(mc, idx, env):
    emu = os.path.join(ROOT, "tests", "emu", "build", "emu_check")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    r = subprocess.run([emu, mc["graph"], mc["fasta"], str(mc["k"]), str(mc["b"]), str(mc["m"]), str(mc["a"]), "seeds-init", os.path.join(mc["dir"], "emu")],
                       capture_output=True, text=True, env=dict(os.environ, EMU_ONLY=str(idx), EMU_THREADS="1", **env))
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.gpu
def test_gpu_matches_reference_rows(mc):
    import sibeliaz_amd
    st = sibeliaz_amd.JunctionStorage(mc["graph"], [mc["fasta"]], mc["k"], threads=4, abundance=mc["a"])
    dev = sibeliaz_amd.Device(st, sibeliaz_amd.Params.make(mc["k"], b=mc["b"], m=mc["m"]), 0)
    seeds = st.seeds(4)
    idxs = sorted(mc["rows"])
    off, inst, score, _ = dev.process_seeds(seeds[idxs])
    for k, idx in enumerate(idxs):
        got = [(int(a["chr"]), int(a["front_idx"]), int(a["back_idx"]), int(a["positive"]) != 0) for a in inst[int(off[k]):int(off[k + 1])]]
        assert (int(score[k]), got) == mc["rows"][idx], "seed %d" % idx
