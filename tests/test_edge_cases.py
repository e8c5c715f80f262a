"""Edge cases (empty seed list, identical genomes, tiny records, several FASTA files, IUPAC / lower case, everything filtered by
the abundance threshold, small b / large m, k = 25 with repeats).
CPU: the C oracle must agree with the real reference binary (where oracle/_ref exists). GPU: the product CLI must agree with it."""
import os
import re

import pytest

from tests import edge_inputs as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _summary(stdout):
    return re.findall(r"^(Blocks found: .*|Coverage: .*)$", stdout, re.M)


@pytest.mark.parametrize("name", E.NAMES)
def test_oracle_agrees_with_reference_on_edge_cases(built, name, tmp_path):
    ref = os.path.join(ROOT, "oracle", "_ref", "sibeliaz-lcb-ref")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref is only built where /root/reference exists")
    case = E.build(name, str(tmp_path))
    a = E.run_cli(ref, case, str(tmp_path / "ref"), ["--noseq"])
    b = E.run_cli(os.path.join(ROOT, "oracle", "lcb_oracle"), case, str(tmp_path / "orc"))
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert open(str(tmp_path / "ref" / "blocks_coords.gff")).read() == open(str(tmp_path / "orc" / "blocks_coords.gff")).read()
    assert _summary(a.stdout) == _summary(b.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("name", E.NAMES)
def test_cli_matches_reference_on_edge_cases(built, name, tmp_path):
    case = E.build(name, str(tmp_path))
    exe = os.path.join(ROOT, "sibeliaz_amd", "bin", "sibeliaz-lcb")
    a = E.run_cli(E.reference_exe(), case, str(tmp_path / "ref"), ["--noseq"])
    b = E.run_cli(exe, case, str(tmp_path / "gpu"), ["--chunks", "3"])
    assert a.returncode == 0, a.stderr
    assert b.returncode == 0, b.stderr
    assert open(str(tmp_path / "gpu" / "blocks_coords.gff")).read() == open(str(tmp_path / "ref" / "blocks_coords.gff")).read()
    assert _summary(a.stdout) == _summary(b.stdout)
    if os.path.basename(E.reference_exe()) == "sibeliaz-lcb-ref":          # block sequences for the aligner (blocksfinder.h:533-582)
        c = E.run_cli(E.reference_exe(), case, str(tmp_path / "refseq"), ["--chunks", "3"])
        assert c.returncode == 0
        for i in range(3):
            assert open(str(tmp_path / "gpu" / ("%d.tmp" % i))).read() == open(str(tmp_path / "refseq" / ("%d.tmp" % i))).read()
