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
