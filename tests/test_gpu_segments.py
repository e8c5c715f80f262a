"""GPU tests (-m gpu) of the (segment, 32-bit offset) positions: the SEG instantiations of the kernels - what an input of 2^32 junction
occurrences or more runs (the reference bounds a CHROMOSOME by 2^32, junctionstorage.h:120-151, README.md:25-26, not the input) - on
the goldens, through the test hooks of lcb_device_opts: `seg_cap` cuts a small input into several segments, `seg_gap` puts unused
table space between the segments of the device tables, so that with a gap of more than 2^32 positions the flat indices of a golden
no longer fit 32 bits and every 64-bit address computation of the device code runs (tables, `used` bitmap, copy-on-write view pages,
footprints, marks). Results never depend on the hooks: same oracle, same reference goldens.

Plus the memory-model side of the footprints (round-4 review): footprints and results of launches that keep every CU busy with heavy
seeds in the wide / big / huge variants - where helper wavefronts update the footprint slots in the HBM workspace while wave 0 works."""
import os

import numpy as np
import pytest

import sibeliaz_amd
from tests.oracle_binding import Oracle
from tests.test_gpu_parity import _compare_all, _setup

pytestmark = pytest.mark.gpu

BIG_GAP = (1 << 32) + 12345          # flat indices of the second segment start beyond 2^32


def _two_segments(st):
    """A segment capacity that cuts the case into two (occasionally three) segments."""
    return max(64, (st.n_positions() * 2) // 3)


def _seg_setup(case, flavour, **opts):
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, threads=4, abundance=case.a)
    p = sibeliaz_amd.Params.make(case.k, b=case.b, m=case.m)
    if flavour == "many":            # a segment per chromosome or two, packed
        o = dict(seg_cap=max(64, st.n_positions() // 5))
    elif flavour == "gap":           # a few segments with a small odd gap (segment bases that are not multiples of 32)
        o = dict(seg_cap=max(64, st.n_positions() // 3), seg_gap=70001)
    else:                            # "wide": two segments, the second one beyond 2^32
        o = dict(seg_cap=_two_segments(st), seg_gap=BIG_GAP)
    o.update(opts)
    return st, p, sibeliaz_amd.Device(st, p, 0, **o)


@pytest.mark.parametrize("flavour", ["many", "gap", "wide"])
def test_segmented_per_seed_parity(built, case, flavour):
    """ProcessVertex::Process of every seed vs the oracle, unused and final `used` state (set_used lays the bitmap out segment by segment)."""
    st, p, dev = _seg_setup(case, flavour)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    seeds = st.seeds(4)
    _compare_all(case, st, dev, orc, seeds, "segments (%s), unused state" % flavour)
    orc.find_blocks(case.k, case.b, case.m)
    dev.set_used(orc.used_bitmap(st.chr_start()))
    _compare_all(case, st, dev, orc, seeds, "segments (%s), final state" % flavour)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_segmented_each_kernel_variant(built, case, mode):
    """Each of the four kernel variants alone with flat indices beyond 2^32, screened launches included."""
    st, p, dev = _seg_setup(case, "wide", start_mode=mode, screen_min=1 if mode == 1 else 0)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    seeds = st.seeds(4)
    seeds = seeds[:700] if mode != 1 else seeds
    _compare_all(case, st, dev, orc, seeds, "segments, variant %d" % mode)
    counts = dev.mode_seeds()
    assert counts[mode - 1] >= len(seeds) and all(c == 0 for i, c in enumerate(counts) if i < mode - 1), counts


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_segmented_footprints_cover_every_read(built, case, mode):
    """Footprints come back as flat 64-bit positions of the HOST tables whatever the device's layout: with every unused position outside
    a seed's footprint set to used the seed must give the same result."""
    st, p, dev = _seg_setup(case, "wide", start_mode=mode)
    seeds = st.seeds(4)[:1000]
    off, inst, fp_off, fp = dev.process_seeds_fp(seeds)
    n_pos = st.n_positions()
    assert fp.dtype == np.dtype("<u8") and (len(fp) == 0 or int(fp[:, 1].max()) < n_pos)
    words = (n_pos + 31) // 32 + 1
    picked = [i for i in range(len(seeds)) if off[i + 1] - off[i] > 1][:: max(1, (len(seeds) // 30))][:30]
    assert picked, "no seed of the case yields a block"
    for i in picked:
        bits = np.ones(words * 32, dtype=bool)
        for lo, hi in fp[int(fp_off[i]):int(fp_off[i + 1])]:
            bits[int(lo):int(hi) + 1] = False
        dev.set_used(np.packbits(bits, bitorder="little").view("<u4"))
        off2, inst2, _, _ = dev.process_seeds_fp(seeds[i:i + 1])
        assert inst2.tobytes() == inst[int(off[i]):int(off[i + 1])].tobytes(), "seed %d: a read outside its footprint changed the result (variant %d)" % (i, mode)


@pytest.mark.parametrize("flavour,knobs", [("many", {}), ("gap", dict(round_phases=3, round_fixed=1)), ("wide", {}), ("wide", dict(lazy_span=-1, sync_jobs=1)),
                                           ("wide", dict(round_phases=1, round_fixed=1, max_jobs=8))])
def test_segmented_find_blocks_matches_reference(built, case, tmp_path, flavour, knobs):
    """Whole FindBlocks - speculative rounds with lazy tails, predicted views (copy-on-write pages of a bitmap with a hole of 2^32 bits),
    side lanes - and GenerateOutput against the REAL reference's goldens."""
    st, p, dev = _seg_setup(case, flavour)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4, **knobs)
    text = "".join("%d\t%d\t%d\t%d\n" % (int(b["id"]), int(b["chr"]), int(b["start"]), int(b["end"])) for b in blocks)
    assert text == case.golden("pretrim.tsv")
    out = str(tmp_path / "out")
    finder.GenerateOutput(out)
    assert open(os.path.join(out, "blocks_coords.gff")).read() == case.golden("ref.gff")


def test_segmented_k25_shape_matches_reference_hash(built, tmp_path):
    """Config 4's shape at test size (8 x 24 chromosomes, k = 25, 186 Mbp) cut into segments of ~40 chromosomes with flat indices beyond
    2^32: blocks_coords.gff against the hash of the unmodified reference's (tests/golden/fullsize.json)."""
    import json

    from tests.test_gpu_fullsize import check_case
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullsize.json")))
    check_case("config4_primates8_test", cases["config4_primates8_test"], str(tmp_path), device_opts=dict(seg_cap=8_000_000, seg_gap=BIG_GAP))


# ---- footprints and results of launches that keep the whole GPU busy with heavy seeds --------------------------------------------------
@pytest.fixture(scope="module")
def heavy():
    """62 strains, 8 Mbp (config 3 with 1/40 of the segments): the head of its seed order is paths of thousands of vertices over dozens of
    voters - the seeds the wide / big variants exist for."""
    import bench
    w = bench.ensure_workload("ecoli62_tiny")
    st = sibeliaz_amd.JunctionStorage(w["graph"], [w["fasta"]], w["k"], threads=8, abundance=w["a"])
    return w, st


@pytest.mark.parametrize("mode,n_seeds", [(2, 1536), (3, 1024), (4, 256)])
def test_footprints_and_results_under_load(built, heavy, mode, n_seeds):
    """Every CU busy with a heavy seed of ONE launch (wide: 16 wavefronts per seed, big: instance fields and footprint slots in the HBM
    workspace, updated by all wavefronts of the workgroup with atomics while wave 0 works; huge: everything there). (1) every result of
    the launch equals the oracle's; (2) the footprints THIS launch reported - not those of a quiet one-seed relaunch - cover what was
    read: with every other unused position set to used, each checked seed reproduces its result. A footprint read through a stale cache
    line (the hazard commit 0d6481f closed) would show here as a hull that is too small."""
    w, st = heavy
    p = sibeliaz_amd.Params.make(w["k"], b=w["b"], m=w["m"])
    dev = sibeliaz_amd.Device(st, p, 0, start_mode=mode)
    seeds = st.seeds(8)[:n_seeds]
    off, inst, fp_off, fp = dev.process_seeds_fp(seeds)
    counts = dev.mode_seeds()
    assert counts[mode - 1] >= len(seeds), counts
    orc = Oracle(w["graph"], [w["fasta"]], w["k"], w["a"])
    bad = 0
    for i in range(len(seeds)):
        ref, _ = orc.process_seed(w["k"], w["b"], w["m"], int(seeds["vid"][i]), int(seeds["ch"][i]))
        got = [(int(a["chr"]), int(a["front_idx"]), int(a["back_idx"]), int(a["positive"]) != 0) for a in inst[int(off[i]):int(off[i + 1])]]
        bad += got != ref
    assert bad == 0, "%d of %d seeds of a full launch differ from the oracle (variant %d)" % (bad, len(seeds), mode)
    n_pos = st.n_positions()
    words = (n_pos + 31) // 32 + 1
    with_block = [i for i in range(len(seeds)) if off[i + 1] - off[i] > 1]
    picked = with_block[:: max(1, len(with_block) // 48)][:48]
    assert len(picked) >= 8, "the head of the seed order yields too few blocks"
    for i in picked:
        bits = np.ones(words * 32, dtype=bool)
        for lo, hi in fp[int(fp_off[i]):int(fp_off[i + 1])]:
            bits[int(lo):int(hi) + 1] = False
        dev.set_used(np.packbits(bits, bitorder="little").view("<u4"))
        off2, inst2, _, _ = dev.process_seeds_fp(seeds[i:i + 1])
        assert inst2.tobytes() == inst[int(off[i]):int(off[i + 1])].tobytes(), "seed %d: the footprint reported by the busy launch does not cover what the seed read (variant %d)" % (i, mode)
