"""CPU tests (no GPU): the C-ABI library loads and exports everything include/lcb.h declares, the host code (SoA loader,
seed enumeration, ordered commit, round engine, output) matches the oracle / the reference goldens, and the kernel LOGIC
passes under the wavefront emulator. No compute call reaches the HIP kernels here."""
import ctypes as C
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import sibeliaz_amd
from sibeliaz_amd import parallel
from tests.oracle_binding import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    lib = sibeliaz_amd.load_library()
    header = open(os.path.join(ROOT, "include", "lcb.h")).read()
    declared = set(re.findall(r"\b(lcb_[a-z0-9_]+)\s*\(", header)) - {"lcb_reprocess_fn", "lcb_allgather_cb", "lcb_process_cb", "lcb_mark_cb", "lcb_reset_cb"}
    assert declared, "no declarations parsed"
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(sibeliaz_amd.api.EXPORTS) <= declared


def test_ctypes_structs_have_the_layout_of_the_header(built, tmp_path):
    """The binding's ctypes mirrors of the C-ABI structs (sibeliaz_amd/api.py) against include/lcb.h compiled by gcc: sizes of all of them,
    offsets of the last fields of the ones that keep growing."""
    import ctypes as C
    from sibeliaz_amd import api
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lcb.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(lcb_stats), sizeof(lcb_hooks), '
                   'sizeof(lcb_device_opts), sizeof(lcb_seed), sizeof(lcb_block), sizeof(lcb_instance), offsetof(lcb_stats, host_dead), offsetof(lcb_hooks, sparse_rounds), '
                   'offsetof(lcb_device_opts, seg_gap)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    want = [C.sizeof(api.Stats), C.sizeof(api.Hooks), C.sizeof(api.DeviceOpts), sibeliaz_amd.SEED_DTYPE.itemsize, sibeliaz_amd.BLOCK_DTYPE.itemsize,
            sibeliaz_amd.INSTANCE_DTYPE.itemsize, api.Stats.host_dead.offset, api.Hooks.sparse_rounds.offset, api.DeviceOpts.seg_gap.offset]
    assert got == want
    header = open(os.path.join(ROOT, "include", "lcb.h")).read()
    assert int(re.search(r"#define LCB_ABI_VERSION (\d+)", header).group(1)) == api.ABI_VERSION == sibeliaz_amd.load_library().lcb_abi_version()


def test_device_fails_loudly_without_gpu(built, case):
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, 2, case.a)
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(sibeliaz_amd.LcbError, match="no CPU fallback"):
        sibeliaz_amd.Device(st, sibeliaz_amd.Params.make(case.k, case.b, case.m))


def test_tables_and_seeds_match_oracle(built, case):
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, 4, case.a)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    seeds = st.seeds(4)
    ref = orc.seeds()
    assert len(seeds) == len(ref)
    got = [(int(s["vid"]), int(s["ch"]), int(s["count"]), int(s["rank"]), int(s["resolve_pos"]), int(s["resolve_chr"])) for s in seeds]
    assert got == ref
    text = "".join("%d\t%d\t%d\t%d\t%d\t%d\n" % t for t in got)
    assert hashlib.sha256(text.encode()).hexdigest() == case.meta["sha256"]["bundles.tsv"]     # the REAL reference's sorted bundle_


def test_loader_errors(built, case, tmp_path):
    with pytest.raises(sibeliaz_amd.LcbError, match="Can't read the input file"):
        sibeliaz_amd.JunctionStorage(str(tmp_path / "missing.bin"), [case.fasta], case.k, 1, case.a)
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n")
    with pytest.raises(sibeliaz_amd.LcbError, match="should start with a '>'"):
        sibeliaz_amd.JunctionStorage(case.graph, [str(bad)], case.k, 1, case.a)
    with pytest.raises(sibeliaz_amd.LcbError, match="must be odd"):
        sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], 16, 1, case.a)


def test_output_stage_matches_reference(built, case, tmp_path):
    """GenerateOutput + GFF writer on the reference's own pre-trim block instances -> the reference's GFF, byte for byte."""
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, 2, case.a)
    rows = [tuple(int(x) for x in ln.split("\t")) for ln in case.golden("pretrim.tsv").splitlines()]
    blocks = np.array(rows, dtype=sibeliaz_amd.BLOCK_DTYPE) if rows else np.zeros(0, dtype=sibeliaz_amd.BLOCK_DTYPE)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    finder.params = sibeliaz_amd.Params.make(case.k, case.b, case.m)
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    finder.GenerateOutput(str(tmp_path / "o"), blocks=blocks, blocks_found=int(summary["blocksFound"]))
    assert open(str(tmp_path / "o" / "blocks_coords.gff")).read() == case.golden("ref.gff")


def _random_instances(rnd, lens, n_blocks, style):
    """Pre-trim block instances that overlap each other the ways the trimming has to tell apart: nested, touching, sharing an end,
    reaching position 0 or the chromosome's end, single-instance blocks on top of held ranges."""
    rows = []
    for bid in range(1, n_blocks + 1):
        copies = rnd.choice([1, 2, 2, 3, 5]) if style != "singles" else rnd.choice([1, 1, 1, 2])
        for _ in range(copies):
            c = rnd.randrange(len(lens))
            n = lens[c]
            if style == "dense":                 # few anchor points: equal starts / ends, touching ranges
                grid = max(1, n // 12)
                a = rnd.randrange(0, 12) * grid
                b = min(n, a + rnd.choice([1, 2, 3]) * grid + rnd.choice([0, 0, 1, -1]))
            else:
                a = rnd.choice([0, rnd.randrange(n), rnd.randrange(n)])
                b = rnd.choice([n, min(n, a + rnd.randrange(1, max(2, n // 3)))])
            if b <= a:
                b = min(n, a + 1)
            if b > a:
                rows.append((bid if rnd.random() < 0.5 else -bid, c, a, b))
    rnd.shuffle(rows)
    return rows


@pytest.mark.parametrize("style", ["loose", "dense", "singles"])
def test_output_trimming_on_random_overlaps(built, case, tmp_path, style):
    """The overlap trimming (held runs per chromosome, output.cpp) against the oracle's flag-per-base restatement of
    blocksfinder.h:605-656 on random instance lists that overlap far more than real ones: same GFF, bytes and order, same number of
    blocks and coverage, for several minimal block sizes."""
    import random
    import zlib
    from tests.oracle_binding import ORC_BLOCK, lib
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, 2, case.a)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    lens = [len("".join(rec.split("\n")[1:])) for rec in open(case.fasta).read().split(">")[1:]]
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    L = lib()
    rnd = random.Random(zlib.crc32((case.name + style).encode()))
    for trial in range(40):
        n_blocks = rnd.choice([1, 3, 10, 60, 300])
        rows = _random_instances(rnd, lens, n_blocks, style)
        min_block = rnd.choice([1, 1, 7, 50, max(lens) // 20 + 1])
        blocks = np.array(rows, dtype=sibeliaz_amd.BLOCK_DTYPE)
        finder.params = sibeliaz_amd.Params.make(case.k, case.b, min_block)
        out = str(tmp_path / ("p%d" % trial))
        nt, cov = finder.GenerateOutput(out, blocks=blocks, blocks_found=n_blocks)
        ob = np.zeros(len(rows), dtype=ORC_BLOCK)
        for f in ("id", "chr", "start", "end"):
            ob[f] = blocks[f]
        oout = str(tmp_path / ("o%d" % trial))
        ocov, err = C.c_double(), C.create_string_buffer(512)
        ont = L.orc_generate_output(orc.h, min_block, ob.ctypes.data, len(ob), n_blocks, oout.encode(), C.byref(ocov), err, 512)
        assert ont >= 0, err.value
        assert open(out + "/blocks_coords.gff").read() == open(oout + "/blocks_coords.gff").read(), "trial %d (%s, min block %d, %d instances)" % (trial, style, min_block, len(rows))
        assert (nt, cov) == (ont, ocov.value)


class OracleProcessor:
    """TEST stand-in for the device: per-seed results from the CPU oracle (never used by the product)."""

    def __init__(self, case, storage):
        self.case = case
        self.orc = Oracle(case.graph, [case.fasta], case.k, case.a)
        self.chr_start = storage.chr_start()
        self.stride = self.orc.L.orc_used_stride()

    def process(self, seeds):
        off = np.zeros(len(seeds) + 1, dtype="<u8")
        rows = []
        for i in range(len(seeds)):
            r, _ = self.orc.process_seed(self.case.k, self.case.b, self.case.m, int(seeds["vid"][i]), int(seeds["ch"][i]))
            rows += [(c, f, b, 1 if p else 0) for (c, f, b, p) in r]
            off[i + 1] = len(rows)
        return off, np.array(rows, dtype=sibeliaz_amd.INSTANCE_DTYPE) if rows else np.zeros(0, dtype=sibeliaz_amd.INSTANCE_DTYPE)

    def mark(self, ranges):
        for lo, hi in np.asarray(ranges, dtype=np.uint64).reshape(-1, 2):
            c = int(np.searchsorted(self.chr_start, lo, side="right") - 1)
            n = self.orc.L.orc_chr_n_pos(self.orc.h, c)
            raw = (C.c_uint8 * (n * self.stride)).from_address(self.orc.L.orc_chr_used(self.orc.h, c))
            for q in range(int(lo), int(hi)):
                raw[(q - int(self.chr_start[c])) * self.stride] = 1

    def reset(self):
        self.orc.reset_used()


@pytest.mark.parametrize("round_phases", [1, 5])
def test_round_engine_with_callback_processor(built, case, round_phases):
    """engine.cpp (rounds, invalidation, batched conflicts, ordered commit) driven through lcb_find_blocks_ex with a callback
    engine: must reproduce the reference's pre-trim block list exactly."""
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, 2, case.a)
    hooks, keep = parallel.make_hooks(processor=OracleProcessor(case, st), round_phases=round_phases)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, hooks=hooks, threads=2)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    assert finder.stats["failures"] == int(summary["failure"]) and finder.stats["blocks_found"] == int(summary["blocksFound"])


def test_cli_usage_errors(built, case, tmp_path):
    exe = os.path.join(ROOT, "sibeliaz_amd", "bin", "sibeliaz-lcb")
    r = subprocess.run([exe, case.fasta, "-k", "15"], capture_output=True, text=True)
    assert r.returncode == 1 and "graph" in r.stderr
    r = subprocess.run([exe, "--graph", case.graph, case.fasta, "-k", "16"], capture_output=True, text=True)
    assert r.returncode == 1 and "odd" in r.stderr
    r = subprocess.run([exe, "--graph", str(tmp_path / "nope.bin"), case.fasta, "-k", "15"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("error: ") and r.stdout.startswith("Loading the graph...")


EMU = os.path.join(ROOT, "tests", "emu", "build", "emu_check")
EMU_SHARE = os.path.join(ROOT, "tests", "emu", "build", "emu_check_share8")   # tiny LCB_VOTE_SHARE_MIN: all-waves reduce/clear path


@pytest.fixture(scope="session")
def emu_built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "all"])
    return EMU


@pytest.mark.parametrize("name,mode,env", [("inv_k25", "seeds-final", {}), ("inv_k25", "find", {}), ("twogenomes", "seeds-final", {"EMU_NW": "4"}),
                                            ("nruns_abund", "find", {"EMU_ROUNDS": "64"}),
                                            ("nruns_abund", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "LCB_LAZY_SPAN": "0"}),     # the shipped kernels, no lazy round tails
                                            # the shipped (non-stats) instantiation: checkpointed replay instead of a replay from Init
                                            ("inv_k25", "seeds-init", {"EMU_NOSTATS": "1"}), ("twogenomes", "medium", {"EMU_NOSTATS": "1", "EMU_LIMIT": "1500"}),
                                            # every kernel variant: wide (LDS path set), big (index in LDS, fields in the workspace), huge (all in the workspace)
                                            ("twogenomes", "medium", {"EMU_LIMIT": "1500"}), ("twogenomes", "big", {"EMU_LIMIT": "1500"}), ("inv_k25", "big", {"EMU_NOSTATS": "1", "EMU_LIMIT": "600"}),
                                            ("twogenomes", "huge", {"EMU_LIMIT": "1500"}), ("inv_k25", "huge", {"EMU_NOSTATS": "1", "EMU_LIMIT": "600"}),
                                            # helper wavefronts: the heaviest seeds with 16 / 8 / 4 wavefronts per workgroup, both vote protocols
                                            ("inv_k25", "medium", {"EMU_NW": "16", "EMU_NOSTATS": "1", "EMU_LIMIT": "200"}),
                                            ("inv_k25", "medium", {"EMU_NW": "16", "EMU_LIMIT": "120", "EMU_SHARE": "1"}),
                                            ("inv_k25", "big", {"EMU_NW": "8", "EMU_NOSTATS": "1", "EMU_LIMIT": "200", "EMU_SHARE": "1"}),
                                            ("inv_k25", "big", {"EMU_NW": "16", "EMU_NOSTATS": "1", "EMU_LIMIT": "200"}),      # the shipped big variant: 16 wavefronts
                                            # other schedules of the emulated wavefronts (the default runs wave 0 as far as it gets first: it then draws every
                                            # voter ticket itself): taking turns at every wave-level operation, and the last wavefront first
                                            ("inv_k25", "medium", {"EMU_NW": "16", "EMU_NOSTATS": "1", "EMU_LIMIT": "200", "EMU_SCHED": "rr"}),
                                            ("inv_k25", "medium", {"EMU_NW": "16", "EMU_LIMIT": "120", "EMU_SCHED": "rev"}),
                                            ("inv_k25", "big", {"EMU_NW": "16", "EMU_NOSTATS": "1", "EMU_LIMIT": "200", "EMU_SCHED": "rr", "EMU_FP_CHECK": "1"}),
                                            ("twogenomes", "medium", {"EMU_NW": "8", "EMU_LIMIT": "300", "EMU_SCHED": "rr"}),
                                            ("inv_k25", "huge", {"EMU_NW": "4", "EMU_LIMIT": "200"}),
                                            ("inv_k25", "seeds-init", {"EMU_NW": "2", "EMU_NOSTATS": "1", "EMU_LIMIT": "300"}),     # the shipped compact variant: 2 wavefronts
                                            ("collinear6", "seeds-init", {"EMU_NW": "4", "EMU_LIMIT": "100", "EMU_SHARE": "1"}),
                                            # predicted `used` views spanning many copy-on-write pages (the EMU_SHARE build has 128-position pages)
                                            ("inv_k25", "find", {"EMU_NOSTATS": "1", "EMU_SHARE": "1", "EMU_ROUNDS": "64"}),
                                            ("nruns_abund", "find", {"EMU_NOSTATS": "1", "EMU_SHARE": "1", "EMU_ROUNDS": "64"}),
                                            # footprint completeness (EMU_FP_CHECK): every unused position outside a seed's footprint set to used -> same result
                                            ("nruns_abund", "seeds-init", {"EMU_NOSTATS": "1", "EMU_LIMIT": "700", "EMU_FP_CHECK": "1"}),
                                            ("inv_k25", "medium", {"EMU_NW": "16", "EMU_NOSTATS": "1", "EMU_LIMIT": "150", "EMU_FP_CHECK": "1"}),
                                            ("inv_k25", "big", {"EMU_NW": "8", "EMU_NOSTATS": "1", "EMU_LIMIT": "150", "EMU_FP_CHECK": "1"}),
                                            # asynchronous job batches (side lanes): computed at once / when first asked for (the two extremes of what a batch
                                            # that reads the live state while it is being marked can see), results visible late, one lane, tiny job cap
                                            ("nruns_abund", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "EMU_SIDE_LANES": "2", "EMU_SIDE_DELAY": "1", "LCB_LAZY_SPAN": "8"}),
                                            ("tandem4", "find", {"EMU_ROUNDS": "8", "EMU_NOSTATS": "1", "EMU_SIDE_LANES": "1", "EMU_SIDE_LATE": "1", "LCB_MAX_JOBS": "16", "EMU_SIDE_DELAY": "2"}),
                                            ("twogenomes", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "EMU_SIDE_LANES": "3", "EMU_SIDE_DELAY": "1000"}),
                                            ("inv_k25", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "EMU_SHARE": "1", "EMU_SIDE_LANES": "2", "EMU_SIDE_LATE": "1"}),
                                            # early critical launch (always with side lanes): the stop's own jobs are begun before the dry run that plans the rest;
                                            # with a lane for the rest, with batches the lanes refuse (the rest then runs synchronously behind the early jobs), and
                                            # with a processor that refuses the early launch (EMU_NO_EARLY: the stop's own jobs run after the dry run)
                                            ("tandem4", "find", {"EMU_NOSTATS": "1", "EMU_EXPECT_EARLY": "1", "EMU_ROUNDS": "8", "EMU_SIDE_LANES": "2", "LCB_LAZY_SPAN": "3"}),
                                            ("smallb", "find", {"EMU_NOSTATS": "1", "EMU_EXPECT_EARLY": "1", "EMU_ROUNDS": "8", "EMU_SIDE_LANES": "2", "LCB_MAX_JOBS": "4", "LCB_LAZY_SPAN": "8"}),
                                            ("nruns_abund", "find", {"EMU_NOSTATS": "1", "EMU_EXPECT_EARLY": "1", "EMU_ROUNDS": "64", "EMU_SIDE_LANES": "2", "EMU_SIDE_CAP": "20", "EMU_SIDE_LATE": "1"}),
                                            ("nruns_abund", "find", {"EMU_NOSTATS": "1", "EMU_NO_EARLY": "1", "EMU_ROUNDS": "64", "EMU_SIDE_LANES": "2"}),
                                            # round 6: results the host settles itself (dead seeds; default) / also sparse speculative launches (lcb_hooks.sparse_rounds = 1:
                                            # only the first phase of every cluster of seeds is launched, rounds of up to 1 024 phases) / neither
                                            ("nruns_abund", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "EMU_SIDE_LANES": "2", "LCB_LAZY_SPAN": "8", "LCB_SPARSE_ROUNDS": "1"}),
                                            ("tandem4", "find", {"EMU_ROUNDS": "8", "EMU_NOSTATS": "1", "EMU_SIDE_LANES": "2", "EMU_SIDE_LATE": "1", "LCB_LAZY_SPAN": "2", "LCB_SPARSE_ROUNDS": "1", "LCB_CLUSTER_GAP": "50"}),
                                            ("inv_k25", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "EMU_SIDE_LANES": "1", "EMU_SIDE_DELAY": "2", "LCB_LAZY_SPAN": "8", "LCB_SPARSE_ROUNDS": "1", "LCB_CLUSTER_GAP": "1000000"}),
                                            ("nruns_abund", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "EMU_SIDE_LANES": "2", "LCB_LAZY_SPAN": "8", "LCB_SPARSE_ROUNDS": "-1"}),
                                            ("twogenomes", "find", {"EMU_ROUNDS": "8", "EMU_NOSTATS": "1", "LCB_SPARSE_ROUNDS": "-1"}),
                                            # the compact variant with the small pools (128 instances / 512 vote slots; lcb_device_opts.compact_pools = 2): per-seed results
                                            # with the footprint check, the overflow into the wide variant (tandem4's pools outgrow 128), the whole engine, with segments
                                            ("inv_k25", "seeds-init", {"EMU_COMPACT_SMALL": "1", "EMU_NW": "2", "EMU_NOSTATS": "1", "EMU_LIMIT": "600", "EMU_FP_CHECK": "1"}),
                                            ("tandem4", "seeds-final", {"EMU_COMPACT_SMALL": "1", "EMU_LIMIT": "500"}),
                                            ("tandem4", "seeds-init", {"EMU_COMPACT_SMALL": "1", "EMU_NW": "2", "EMU_NOSTATS": "1", "EMU_LIMIT": "500", "EMU_FP_CHECK": "1"}),   # (two wavefronts; the seeds that outgrow the pools re-run in the wide variant)
                                            ("nruns_abund", "find", {"EMU_COMPACT_SMALL": "1", "EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "EMU_NW": "2", "EMU_SIDE_LANES": "2", "LCB_LAZY_SPAN": "8"}),
                                            ("twogenomes", "find", {"EMU_COMPACT_SMALL": "1", "EMU_ROUNDS": "8"}),
                                            ("inv_k25", "seeds-init", {"EMU_COMPACT_SMALL": "1", "EMU_SEG_CAP": "2000", "EMU_SEG_GAP": "4300000000", "EMU_SEG_MAX": "2", "EMU_NOSTATS": "1", "EMU_LIMIT": "300", "EMU_FP_CHECK": "1"}),
                                            # predictive engine: no F prediction / tiny job cap / view starvation, fixed whole-input round
                                            ("nruns_abund", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "LCB_PREDICT_F": "1", "LCB_MAX_JOBS": "8"}),
                                            ("nruns_abund", "find", {"EMU_ROUNDS": "64", "EMU_NOSTATS": "1", "EMU_VIEWS": "3", "LCB_ROUND_FIXED": "1", "LCB_MAX_JOBS": "64"}),
                                            # Positions as (segment, 32-bit offset) pairs - the SEG kernels that inputs of 2^32 junction occurrences and more run.
                                            # EMU_SEG_CAP cuts a golden into segments of a few chromosomes (one chromosome each at the small values), EMU_SEG_GAP
                                            # puts unused table space between them: with more than 2^32 the flat indices no longer fit 32 bits, i.e. every 64-bit
                                            # address computation of the device code runs here (the gapped tables are lazily zeroed allocations)
                                            ("collinear6", "seeds-init", {"EMU_SEG_CAP": "3000", "EMU_LIMIT": "500"}),
                                            ("twogenomes", "seeds-init", {"EMU_SEG_CAP": "4000", "EMU_SEG_GAP": "4300000000", "EMU_LIMIT": "400", "EMU_FP_CHECK": "1"}),
                                            ("twogenomes", "seeds-final", {"EMU_SEG_CAP": "4000", "EMU_SEG_GAP": "4300000000", "EMU_NOSTATS": "1", "EMU_NW": "2"}),
                                            ("inv_k25", "medium", {"EMU_SEG_CAP": "2000", "EMU_SEG_GAP": "777", "EMU_NW": "16", "EMU_NOSTATS": "1", "EMU_LIMIT": "200", "EMU_FP_CHECK": "1"}),
                                            ("inv_k25", "big", {"EMU_SEG_CAP": "2000", "EMU_SEG_GAP": "100000", "EMU_NW": "16", "EMU_NOSTATS": "1", "EMU_LIMIT": "200", "EMU_SCHED": "rr", "EMU_FP_CHECK": "1"}),
                                            ("twogenomes", "huge", {"EMU_SEG_CAP": "1000", "EMU_LIMIT": "800"}),
                                            # footprint slots shared across segments (the EMU_SHARE build has a sixteenth of the slots): an instance of the backward
                                            # extension that would have to share a slot of ANOTHER segment ends the seed with an overflow status and the next variant
                                            # re-runs it (the shared hull kept the old segment and did not cover the new instance's reads)
                                            ("twogenomes", "seeds-init", {"EMU_SEG_CAP": "1000", "EMU_SHARE": "1", "EMU_LIMIT": "900", "EMU_FP_CHECK": "1"}),
                                            ("tandem4", "seeds-init", {"EMU_SEG_CAP": "1500", "EMU_SHARE": "1", "EMU_LIMIT": "700", "EMU_FP_CHECK": "1", "EMU_NOSTATS": "1"}),
                                            ("nruns_abund", "find", {"EMU_SEG_CAP": "1500", "EMU_SEG_GAP": "1000", "EMU_ROUNDS": "64"}),
                                            ("twogenomes", "find", {"EMU_SEG_CAP": "2000", "EMU_SEG_GAP": "70000", "EMU_NOSTATS": "1", "EMU_SIDE_LANES": "2", "EMU_NW": "2"}),
                                            ("tandem4", "find", {"EMU_SEG_CAP": "3000", "EMU_SEG_GAP": "123457", "EMU_NOSTATS": "1", "EMU_SHARE": "1", "EMU_ROUNDS": "8", "EMU_SIDE_LANES": "1", "EMU_SIDE_LATE": "1", "LCB_LAZY_SPAN": "8"})])
def test_kernel_logic_under_wave_emulator(built, emu_built, case_dir, name, mode, env, tmp_path):
    """The unmodified device code of lcb_kernel.h on the CPU wavefront emulator (tests/emu) vs the oracle: per-seed results,
    event counters and the whole round engine. Logic only — the GPU tests are the parity tests proper."""
    from tests.conftest import Case
    c = Case(name, case_dir)
    exe = EMU_SHARE if env.get("EMU_SHARE") else emu_built
    env = dict(env)
    # (lazy round tails multiply the emulated work of a golden, whose rounds are a handful of phases: the cases that are about something
    # else run without them, the ones that name LCB_LAZY_SPAN are about them - 8 is the product's default)
    if mode == "find": env.setdefault("LCB_LAZY_SPAN", "0")
    r = subprocess.run([exe, c.graph, c.fasta, str(c.k), str(c.b), str(c.m), str(c.a), mode, str(tmp_path / "emu")], capture_output=True, text=True,
                       env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-2000:]
    if mode == "find":
        assert open(str(tmp_path / "emu" / "blocks_coords.gff")).read() == c.golden("ref.gff")


@pytest.mark.parametrize("name", ["nruns_abund", "inv_k25", "collinear6"])
def test_engine_model_reproduces_the_oracle(built, case_dir, name):
    """tests/emu/engine_model: the product's round engine driven by the CPU oracle (with the oracle's own footprints) instead of
    the device - the tool that prices engine policies at sizes the emulator cannot reach. It must reproduce the oracle's
    FindBlocks exactly."""
    from tests.conftest import Case
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "build/engine_model"])
    c = Case(name, case_dir)
    r = subprocess.run([os.path.join(ROOT, "tests", "emu", "build", "engine_model"), c.graph, c.fasta, str(c.k), str(c.b), str(c.m), str(c.a)],
                       capture_output=True, text=True, env=dict(os.environ, MODEL_THREADS="2"))
    assert r.returncode == 0 and "FindBlocks: equal" in r.stderr, r.stderr[-1500:]


@pytest.mark.parametrize("env", [{}, {"LCB_MAX_JOBS": "64", "LCB_EAGER_PHASES": "2"}, {"LCB_PREDICT_F": "2"},
                                 {"MODEL_SIDE_LANES": "2"}])      # asynchronous job batches on a virtual clock + the early critical launch
def test_engine_model_at_scale(built, tmp_path_factory, env):
    """The round engine over 176 000 seeds (config 2 at a tenth of the segments: the emulator cannot reach this size) with the
    oracle as the processor: rounds that grow to 256 phases, hundreds of job launches against predicted views, all with the
    oracle's footprints - the blocks must be the oracle's FindBlocks for every knob set."""
    import bench
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "build/engine_model"])
    work = os.environ.get("LCB_TEST_WORKLOADS") or str(tmp_path_factory.getbasetemp().parent / "lcb_model_workloads")
    old = os.environ.get("LCB_BENCH_DIR")
    os.environ["LCB_BENCH_DIR"] = work
    try:
        w = bench.ensure_workload("ecoli10_small")
    finally:
        if old is None:
            os.environ.pop("LCB_BENCH_DIR", None)
        else:
            os.environ["LCB_BENCH_DIR"] = old
    r = subprocess.run([os.path.join(ROOT, "tests", "emu", "build", "engine_model"), w["graph"], w["fasta"], str(w["k"]), str(w["b"]), str(w["m"]), str(w["a"])],
                       capture_output=True, text=True, env=dict(os.environ, MODEL_THREADS="4", **env))
    assert r.returncode == 0 and "FindBlocks: equal" in r.stderr, r.stderr[-1500:]




def test_segment_plan_maps_positions(built, tmp_path):
    """csrc/lcb_segments.h: the plan that cuts the flat position space into segments of whole chromosomes and translates between the host's dense
    positions and the device's (segment, offset) layout - capacities that force a segment per chromosome, chromosomes larger than the capacity,
    gaps between the segments (toDev / rangeToDev / toHost round trips at every chromosome boundary), and the limits it reports."""
    src = tmp_path / "segplan.cpp"
    src.write_text(r'''
#include <cstdio>
#include <string>
#include "lcb_segments.h"
void lcb_set_error(const std::string&) {}
static lcb_graph make(const std::vector<uint64_t>& len) {
    lcb_graph g; g.chrStart.assign(1, 0);
    for (uint64_t l : len) { g.chrStart.push_back(g.chrStart.back() + l); g.chrName.push_back("c"); }
    g.posId.assign((size_t)g.chrStart.back(), 1);
    return g;
}
int main() {
    int bad = 0;
    const std::vector<uint64_t> len = {100, 1, 250, 40, 40, 999, 3};
    const lcb_graph g = make(len);
    for (uint64_t cap : {0ull, 1ull, 41ull, 100ull, 290ull, 5000ull}) for (uint64_t gap : {0ull, 64ull, 70001ull, (1ull << 32) + 12345}) {
        const LcbSegPlan p = lcb_plan_segments(g, cap, gap);
        if (p.segStart.front() != 0 || p.segStart.back() != g.nPos()) bad++;
        if (cap == 0 && p.nSeg() != 1) bad++;
        if (cap == 1 && p.nSeg() != len.size()) bad++;                 // a chromosome longer than the capacity gets a segment of its own
        if (p.devPositions != g.nPos() + (uint64_t)(p.nSeg() - 1) * gap) bad++;
        for (uint32_t sg = 0; sg < p.nSeg(); sg++) {                   // a segment of several chromosomes holds at most `cap` positions
            size_t nChr = 0;
            for (size_t c = 0; c < len.size(); c++) nChr += (p.chrWord[c] >> LCB_SEG_SHIFT) == sg;
            if (nChr == 0 || (nChr > 1 && cap && p.segStart[sg + 1] - p.segStart[sg] > cap)) bad++;
        }
        for (size_t c = 0; c < len.size(); c++) {
            const uint32_t s = p.chrWord[c] >> LCB_SEG_SHIFT;
            if ((p.chrWord[c] & LCB_CHR_MASK) != c || s >= p.nSeg()) { bad++; continue; }
            if (p.segStart[s] + p.chrLo[c] != g.chrStart[c] || p.segStart[s] + p.chrHi[c] != g.chrStart[c + 1]) bad++;      // g bounds of the chromosome in its segment
            if (p.chrDev[c] != p.segDev[s] + p.chrLo[c] || p.segDev[s] != p.segStart[s] + (uint64_t)s * gap) bad++;
            for (uint64_t f : {g.chrStart[c], g.chrStart[c] + len[c] / 2, g.chrStart[c + 1] - 1}) {
                if (p.segOfHost(f) != s) bad++;
                if (p.toDev(f) != p.segDev[s] + (f - p.segStart[s]) || p.toHost(p.toDev(f)) != f) bad++;
            }
            uint64_t lo, hi;
            p.rangeToDev(g.chrStart[c], g.chrStart[c + 1], lo, hi);      // a whole chromosome: its end may be the next segment's start
            if (lo != p.chrDev[c] || hi - lo != len[c]) bad++;
        }
    }
    // limits
    try { lcb_plan_segments(make(std::vector<uint64_t>(40, 10)), 1, 0); bad++; } catch (LcbError&) {}      // 40 segments
    printf("%d\n", bad);
    return bad ? 1 : 0;
}
''')
    exe = tmp_path / "segplan"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "sibeliaz_amd", "csrc"), str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)], text=True).strip() == "0"
