"""A bounded, fixed-seed slice of the randomized emulator campaign (tests/emu/fuzz.py) inside the CPU suite: random genome
sets and parameters, the device code + the round engine under the wavefront emulator against the oracle — single-wave and
multi-wavefront kernel variants, random engine knobs. The open-ended campaign stays a manual tool; this pins a handful of
cases (about a minute) so that a kernel or engine change cannot land without passing them."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lcb_fuzz", os.path.join(ROOT, "tests", "emu", "fuzz.py"))
fuzz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fuzz)


@pytest.fixture(scope="module")
def emu_built(built):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])


@pytest.mark.parametrize("case_no", [72, 73, 77, 81, 83, 86, 101, 108, 128, 138])
def test_fixed_fuzz_cases(emu_built, case_no, tmp_path):
    desc, res, synth = fuzz.run_case(case_no, str(tmp_path), small=True, timeout=120)
    bad = [(m, e, t) for (m, e, ok, t) in res if not ok]
    assert not bad, "%s synth=%s: %s" % (desc, " ".join(synth), bad)


@pytest.mark.parametrize("case_no,mode,limit", [(2003, "seeds-init", 450), (2007, "seeds-init", 450), (2009, "seeds-init", 220), (2009, "seeds-final", 60)])
def test_footprints_cover_every_read(emu_built, case_no, mode, limit, tmp_path):
    """Rule (2) of the engine rests on the footprint covering every position the computation read as 0. Property test: with EVERY
    unused position outside a seed's footprint set to used, Process() must still give the seed's result (EMU_FP_CHECK). The
    round-2 kernel failed it on these inputs: the backward extension re-used the footprint slots of instances the replay had
    dropped, so the creating read of the new instance went unrecorded (fixed: lcb_fp_slot)."""
    synth, (k, b, m, a), _, _ = fuzz.case_params(case_no)
    fa, gr = str(tmp_path / "g.fa"), str(tmp_path / "g.bin")
    subprocess.check_call([fuzz.BIN + "/lcb-synth", "-o", fa] + synth, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([fuzz.BIN + "/lcb-mkgraph", "-k", str(k), "-o", gr, fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([fuzz.EMU, gr, fa, str(k), str(b), str(m), str(a), mode], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, EMU_FP_CHECK="1", EMU_NOSTATS="1", EMU_LIMIT=str(limit)))      # (the seeds that failed lie below the limits)
    assert r.returncode == 0 and "FAIL" not in r.stderr and "MISMATCH" not in r.stderr, r.stderr[-1500:]
