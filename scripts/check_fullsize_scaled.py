#!/usr/bin/env python3
"""The Gbp-scale parity cases (configs 4 / 5 of BASELINE.json as 8 x 24 chromosomes = 1.24 Gbp and 16 x 20 = 1.0 Gbp, k = 25): the GFF of
the MI355X run against the hashes the UNMODIFIED reference produced in the build container (tests/golden/fullsize_scaled.json, written by
tests/golden/make_fullsize.py). Not part of `pytest -m gpu`: generating each input takes 2.5 minutes on the GPU box.

    python scripts/check_fullsize_scaled.py [case ...]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_fullsize import check_case  # noqa: E402

cases = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_scaled.json")))
bad = 0
for name in sorted(cases):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    t = time.time()
    try:
        with tempfile.TemporaryDirectory() as tmp:
            check_case(name, cases[name], tmp)
        print("%s: blocks_coords.gff equal to the reference's (%d blocks, %d lines); %.0f s with generation" % (name, cases[name]["blocks_found"], cases[name]["gff_lines"], time.time() - t), flush=True)
    except AssertionError as e:
        bad += 1
        print("%s: FAILED %s" % (name, e), flush=True)
sys.exit(1 if bad else 0)
