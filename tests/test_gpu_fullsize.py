"""GPU parity at the BASELINE.json configurations' stated size (run with -m gpu on an MI355X): config 2 (10 strains, 45 Mbp)
and config 3 (62 strains, 281 Mbp; a = 150 and a = 2*62*7 = 868, reference README.md:161-175), and the shapes of configs 4 / 5
(k = 25, 8 strains x 24 chromosomes at 1 % divergence and 16 x 20 at 0.5 %, repeat families of 100 copies under a = 150: long
paths over few genomes, ten commit conflicts per block) at a size the suite can afford (186 / 217 Mbp). The inputs are regenerated on
the box with the deterministic tools (lcb-synth seed 1001 / 1002 + lcb-mkgraph) and checked by hash; the expected
blocks_coords.gff hashes were produced by the UNMODIFIED reference in the build container (tests/golden/make_fullsize.py ->
tests/golden/fullsize.json). The product path is the C ABI: lcb_graph_load -> lcb_enumerate_seeds -> lcb_find_blocks_ex ->
lcb_generate_output."""
import hashlib
import json
import os
import sys

import pytest

import sibeliaz_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

FULL = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))


def _sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def check_case(name, g, out_dir, device_opts=None):
    """One full-size case through the C ABI against the reference's hashes (also used by scripts/check_fullsize_scaled.py for the
    Gbp-scale cases of tests/golden/fullsize_scaled.json, which the default suite cannot afford: 2.5 minutes of generation each)."""
    import bench
    w = bench.ensure_workload(g["workload"])
    assert _sha256(w["fasta"]) == g["fasta_sha256"], "lcb-synth did not reproduce the genomes the reference was run on"
    assert _sha256(w["graph"]) == g["graph_sha256"], "lcb-mkgraph did not reproduce the junction file the reference was run on"
    threads = min(32, os.cpu_count() or 1)
    st = sibeliaz_amd.JunctionStorage(w["graph"], [w["fasta"]], g["k"], threads=threads, abundance=g["a"])
    p = sibeliaz_amd.Params.make(g["k"], b=g["b"], m=g["m"])
    dev = sibeliaz_amd.Device(st, p, 0, **(device_opts or {}))
    finder = sibeliaz_amd.BlocksFinder(st, g["k"])
    finder.FindBlocks(g["m"], g["b"], device=dev, threads=threads)
    out = os.path.join(out_dir, "out")
    n_trimmed, cov = finder.GenerateOutput(out)
    assert n_trimmed == g["blocks_found"] and "%.2f" % cov == g["coverage"]
    gff = os.path.join(out, "blocks_coords.gff")
    assert sum(1 for _ in open(gff)) == g["gff_lines"]
    assert _sha256(gff) == g["gff_sha256"], "blocks_coords.gff differs from the reference's"
    print("%s: %d seeds, %.1f s phase loop, %d launches, modes %s" % (name, finder.stats["seeds"], finder.stats["wall_ms"] / 1e3,
                                                                        finder.stats["launches"], dev.mode_seeds()))
    dev.close()
    st.close()


@pytest.mark.parametrize("name", sorted(FULL))
def test_fullsize_gff_equals_reference(built, name, tmp_path):
    check_case(name, FULL[name], str(tmp_path))
