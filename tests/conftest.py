"""Shared fixtures. GPU tests are marked @pytest.mark.gpu (run with -m gpu on an MI355X box)."""
import gzip
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

CASES = sorted(d for d in os.listdir(GOLDEN) if os.path.isfile(os.path.join(GOLDEN, d, "genomes.fa.gz")))   # full cases (inputs + all goldens)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Build everything once per session (tools, HIP library, CLI, oracle)."""
    subprocess.check_call([sys.executable, os.path.join(ROOT, "sibeliaz_amd", "build.py"), "all"])
    return True


class Case:
    def __init__(self, name, tmp):
        self.name = name
        self.dir = os.path.join(GOLDEN, name)
        with open(os.path.join(self.dir, "golden.json")) as f:
            self.meta = json.load(f)
        self.k, self.b, self.m, self.a = (self.meta[x] for x in "kbma")
        self.fasta = os.path.join(tmp, name + ".fa")
        self.graph = os.path.join(tmp, name + ".bin")
        for src, dst in (("genomes.fa.gz", self.fasta), ("graph.bin.gz", self.graph)):
            if not os.path.exists(dst):
                with gzip.open(os.path.join(self.dir, src), "rb") as f, open(dst, "wb") as g:
                    g.write(f.read())

    def golden(self, fn, mode="r"):
        with open(os.path.join(self.dir, fn), mode) as f:
            return f.read()


@pytest.fixture(scope="session")
def case_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("cases"))


@pytest.fixture(params=CASES)
def case(request, case_dir):
    return Case(request.param, case_dir)
