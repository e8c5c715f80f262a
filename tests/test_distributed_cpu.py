"""world_size-2 test of the multi-rank path on CPU (gloo): the native round engine with seeds dealt across ranks, results
all-gathered through torch.distributed, identical commit on every rank. A TEST stand-in built on the oracle plays the
per-rank device; the sharding / gather / commit code is exactly what runs with RCCL on GPUs."""
import os
import re
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch.distributed as dist
import sibeliaz_amd
from sibeliaz_amd import parallel
from conftest import Case
from test_host_cpu import OracleProcessor
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
case = Case(%(name)r, %(tmp)r)
st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, 2, case.a)
hooks, keep = parallel.make_hooks(rank, world, processor=OracleProcessor(case, st), round_phases=%(rounds)d)
finder = sibeliaz_amd.BlocksFinder(st, case.k)
blocks = finder.FindBlocks(case.m, case.b, hooks=hooks, threads=2)
got = "".join("%%d\t%%d\t%%d\t%%d\n" %% (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
assert got == case.golden("pretrim.tsv"), "rank %%d: blocks differ from the reference" %% rank
assert finder.stats["exchanges"] > 0
sys.stdout.write("rank %%d ok %%d %%d\n" %% (rank, finder.stats["rounds"], finder.stats["exchanges"]))   # one write: the ranks share the pipe
sys.stdout.flush()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("name,rounds", [("inv_k25", 4), ("twogenomes", 64)])
def test_two_rank_gloo_matches_reference(built, tmp_path, name, rounds):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Case
    Case(name, str(tmp_path))            # unpack the fixture once, before the ranks start
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, name=name, tmp=str(tmp_path), rounds=rounds))
    for attempt in range(2):             # one retry: the rendezvous port can be taken between probing and binding
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
               str(_free_port()), str(script)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if r.returncode == 0 or "rank" in r.stdout and "differ" in (r.stdout + r.stderr):
            break
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert len(re.findall(r"rank \d ok ", r.stdout)) == 2, r.stdout[-500:]


@pytest.mark.parametrize("name,ranks,env", [("inv_k25", 2, {}), ("nruns_abund", 3, {"EMU_VIEWS": "3"}), ("tandem4", 4, {"EMU_ROUNDS": "7"}),
                                            # the multi-rank engine is the single-rank engine: asynchronous job batches dealt to the ranks' side lanes, results
                                            # published through collective exchanges (batches computed at once / late / visible late / refused by a lane)
                                            ("nruns_abund", 2, {"EMU_SIDE_LANES": "2", "LCB_LAZY_SPAN": "8"}), ("tandem4", 3, {"EMU_SIDE_LANES": "2", "EMU_SIDE_LATE": "1", "EMU_SIDE_DELAY": "2", "EMU_ROUNDS": "8"}),
                                            ("nruns_abund", 4, {"EMU_SIDE_LANES": "1", "EMU_SIDE_CAP": "5", "LCB_MAX_JOBS": "16"}),
                                            ("inv_k25", 2, {"EMU_SIDE_LANES": "3", "EMU_SIDE_DELAY": "1000", "EMU_ROUNDS": "64"}),
                                            # eight ranks (results do not depend on 1 / 2 / 4 / 8 ranks), with and without background batches
                                            ("inv_k25", 8, {}), ("nruns_abund", 8, {"EMU_SIDE_LANES": "2", "EMU_SIDE_DELAY": "1", "LCB_LAZY_SPAN": "8"}),
                                            # sparse speculative launches / no host-settled seeds (lcb_hooks.sparse_rounds 1 / -1): every rank settles the same seeds
                                            ("nruns_abund", 3, {"EMU_SIDE_LANES": "2", "LCB_LAZY_SPAN": "8", "LCB_SPARSE_ROUNDS": "1"}), ("tandem4", 2, {"EMU_SIDE_LANES": "2", "EMU_ROUNDS": "8", "LCB_SPARSE_ROUNDS": "-1"}),
                                            # ... and with positions as (segment, offset) pairs (the SEG kernels; test_host_cpu.py)
                                            ("tandem4", 4, {"EMU_SEG_CAP": "3000", "EMU_SEG_GAP": "99991", "EMU_SIDE_LANES": "2"})])
def test_multi_rank_engine_with_real_footprints(built, tmp_path, name, ranks, env):
    """The multi-rank round engine with the wave emulator as each rank's device: every launch (rounds AND job launches against
    predicted `used` views) is dealt to the ranks, per-seed results and REAL footprints cross pack / all-gather / unpack, every
    rank commits identically and ends with the reference's block list (emu_check find-ranks). With side lanes a stop's speculative
    jobs run in the background of every rank and their results are published through exchangeSide."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Case
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "all"])
    c = Case(name, str(tmp_path))
    r = subprocess.run([os.path.join(ROOT, "tests", "emu", "build", "emu_check"), c.graph, c.fasta, str(c.k), str(c.b), str(c.m), str(c.a), "find-ranks",
                        str(tmp_path / "emu")], capture_output=True, text=True,
                       env=dict(os.environ, EMU_NOSTATS="1", EMU_THREADS="2", EMU_RANKS=str(ranks), **dict({"LCB_LAZY_SPAN": "0"}, **env)))     # (lazy round tails: the cases that name them)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = re.findall(r"find-ranks rank \d+/%d: .* exchanges (\d+) .*diffs 0" % ranks, r.stderr)
    assert len(lines) == ranks and all(int(x) > 0 for x in lines), r.stderr[-1000:]
