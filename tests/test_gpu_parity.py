"""GPU parity tests (run with -m gpu on an MI355X): the HIP hot path, called through the C ABI, against the CPU oracle
on the same inputs and against the goldens produced by the real reference. Integer/index work: bit-exact."""
import os
import subprocess

import numpy as np
import pytest

import sibeliaz_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests.oracle_binding import Oracle, OrcCounters

pytestmark = pytest.mark.gpu


def _setup(case, **device_opts):
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, threads=4, abundance=case.a)
    p = sibeliaz_amd.Params.make(case.k, b=case.b, m=case.m)
    dev = sibeliaz_amd.Device(st, p, 0, **device_opts)
    return st, p, dev


def _compare_all(case, st, dev, orc, seeds, what):
    off, inst, score, _ = dev.process_seeds(seeds)
    bad = 0
    for i in range(len(seeds)):
        ref, ref_score = orc.process_seed(case.k, case.b, case.m, int(seeds["vid"][i]), int(seeds["ch"][i]))
        got = inst[int(off[i]):int(off[i + 1])]
        tup = [(int(a["chr"]), int(a["front_idx"]), int(a["back_idx"]), int(a["positive"]) != 0) for a in got]
        if tup != ref or int(score[i]) != ref_score:
            bad += 1
            if bad <= 3:
                print("%s: seed %d (vid %d) differs:\n  gpu    %s %d\n  oracle %s %d" % (what, i, seeds["vid"][i], tup[:6], score[i], ref[:6], ref_score))
    assert bad == 0, "%d of %d seeds differ from the oracle (%s)" % (bad, len(seeds), what)


def test_seeds_match_golden(built, case):
    """The whole sorted bundle list (blocksfinder.h:461-517) against the sha256 of the REAL reference's bundle_."""
    import hashlib
    st, p, dev = _setup(case)
    seeds = st.seeds(4)
    text = "".join("%d\t%d\t%d\t%d\t%d\t%d\n" % (int(s["vid"]), int(s["ch"]), int(s["count"]), int(s["rank"]), int(s["resolve_pos"]), int(s["resolve_chr"])) for s in seeds)
    assert hashlib.sha256(text.encode()).hexdigest() == case.meta["sha256"]["bundles.tsv"]


def test_per_seed_parity_unused_state(built, case):
    """ProcessVertex::Process on the GPU vs the oracle, every seed, all-unused table (every seed does real work)."""
    st, p, dev = _setup(case)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    _compare_all(case, st, dev, orc, st.seeds(4), "unused state")


def test_per_seed_parity_final_state(built, case):
    """Same against the final `used` state (exercises IsUsed, Finish*, tryUsed and the Compatible bitmap test)."""
    st, p, dev = _setup(case)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    orc.find_blocks(case.k, case.b, case.m)
    dev.set_used(orc.used_bitmap(st.chr_start()))
    _compare_all(case, st, dev, orc, st.seeds(4), "final state")


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_each_kernel_variant_parity(built, case, mode):
    """Every seed through ONE kernel variant (1 compact: one wavefront per seed; 2 wide: 16 wavefronts share the votes;
    3 big: index and vote table in LDS, instance fields in HBM; 4 huge: all per-path state in the global-memory workspace),
    checked by the per-variant seed counters."""
    st, p, dev = _setup(case, start_mode=mode)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    seeds = st.seeds(4)
    seeds = seeds[:900] if mode != 1 else seeds
    _compare_all(case, st, dev, orc, seeds, "variant %d" % mode)
    counts = dev.mode_seeds()
    assert counts[mode - 1] >= len(seeds) and all(c == 0 for i, c in enumerate(counts) if i < mode - 1), counts


def test_overflow_chain_reaches_big_mode(built, case):
    """Tiny path sets in the compact AND the wide slots: a seed that pushes more than a few vertices overflows the compact
    variant's path set, skips the wide one (whose LDS path set is the smaller of the two) and ends in the big variant; results
    are the oracle's."""
    st, p, dev = _setup(case, path_cap=16, path_cap_max=16, wide_path_cap=16, start_mode=1)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    seeds = st.seeds(4)[:600]
    _compare_all(case, st, dev, orc, seeds, "overflow chain")
    counts = dev.mode_seeds()
    assert counts[0] == len(seeds) and counts[2] > 0, counts


def test_compact_path_set_grows_on_demand(built, case):
    """The compact variant's HBM path set starts small (so that the sets of all slots stay cache-resident) and is enlarged x4
    when a path of few instances overflows it: with a 16-vertex start every longer path is re-run in the compact variant after
    a growth, nothing reaches the big variant, results are the oracle's."""
    st, p, dev = _setup(case, path_cap=16, start_mode=1)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    seeds = st.seeds(4)[:600]
    _compare_all(case, st, dev, orc, seeds, "growing compact path set")
    counts = dev.mode_seeds()
    assert counts[0] > len(seeds) and counts[2] == 0 and counts[3] == 0, counts


def test_result_arena_grows_when_a_launch_fills_it(built, case):
    """A launch whose results do not fit the pinned result arena re-runs the unlucky seeds after the arena was enlarged (x4 at
    the first overflow, so that a workload pays for it once or twice, not in every round): a 64-instance start still gives the
    oracle's results for every seed."""
    st, p, dev = _setup(case, arena=64)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    _compare_all(case, st, dev, orc, st.seeds(4), "tiny result arena")


def test_long_paths_fall_back_from_wide_to_compact(built, case):
    """The wide variant keeps its path set in LDS (4096 vertices), the compact one in HBM: a path that overflows the wide
    variant's set is re-run in the compact variant, not in the slow big one."""
    st, p, dev = _setup(case, wide_path_cap=16, start_mode=2)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    seeds = st.seeds(4)[:600]
    _compare_all(case, st, dev, orc, seeds, "wide -> compact")
    counts = dev.mode_seeds()
    assert counts[1] == len(seeds) and counts[0] > 0 and counts[2] == 0, counts


def test_screened_launch_parity(built, case):
    """A launch that goes through the screening kernel first (dead seeds finalised there, live ones queued) gives the same
    per-seed results as unscreened launches, in the final `used` state where most seeds are dead."""
    st, p, dev = _setup(case, screen_min=1, start_mode=1)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    orc.find_blocks(case.k, case.b, case.m)
    dev.set_used(orc.used_bitmap(st.chr_start()))
    _compare_all(case, st, dev, orc, st.seeds(4), "screened, final state")
    orc2 = Oracle(case.graph, [case.fasta], case.k, case.a)
    dev.reset_used()
    _compare_all(case, st, dev, orc2, st.seeds(4), "screened, unused state")


def test_event_counters_match_oracle(built, case):
    st, p, dev = _setup(case)
    dev.set_stats_mode(True)
    seeds = st.seeds(4)
    _, _, _, ctr = dev.process_seeds(seeds, counters=True)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    oc = OrcCounters()
    for i in range(len(seeds)):
        orc.process_seed(case.k, case.b, case.m, int(seeds["vid"][i]), int(seeds["ch"][i]), counters=oc)
    assert ctr == oc.as_dict()


def test_batch_split_invariance(built, case):
    """A seed's result is a pure function of (tables, used, seed): any batching gives identical results."""
    st, p, dev = _setup(case)
    seeds = st.seeds(4)[:700]
    off, inst, score, _ = dev.process_seeds(seeds)
    pieces = [dev.process_seeds(seeds[a:a + 97]) for a in range(0, len(seeds), 97)]
    inst2 = np.concatenate([x[1] for x in pieces])
    score2 = np.concatenate([x[2] for x in pieces])
    assert inst.tobytes() == inst2.tobytes() and score.tobytes() == score2.tobytes()


def test_find_blocks_matches_reference(built, case, tmp_path):
    """Whole FindBlocks (phase loop + ordered commit with GPU re-processing) and GenerateOutput vs the REAL reference."""
    st, p, dev = _setup(case)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    assert finder.stats["blocks_found"] == int(summary["blocksFound"])
    assert finder.stats["failures"] == int(summary["failure"])
    finder.GenerateOutput(str(tmp_path / "out"))
    assert open(str(tmp_path / "out" / "blocks_coords.gff")).read() == case.golden("ref.gff")
    # idempotence: a second run on the same device gives the same blocks
    blocks2 = finder.FindBlocks(case.m, case.b, device=dev, threads=4)
    assert blocks.tobytes() == blocks2.tobytes()


@pytest.mark.parametrize("pools", [1, 2])
def test_compact_pools_parity(built, case, pools):
    """The compact variant with either of its pool sizes (lcb_device_opts.compact_pools: 1 = 256 instances / 1 024 vote slots, 2 = 128 / 512 with 8
    workgroups per CU; 0 lets the input choose): every seed started there in both `used` states (the seeds that outgrow the pools go up the ladder),
    then the whole FindBlocks + GFF against the reference."""
    st, p, dev = _setup(case, compact_pools=pools, start_mode=1)
    orc = Oracle(case.graph, [case.fasta], case.k, case.a)
    seeds = st.seeds(4)
    _compare_all(case, st, dev, orc, seeds, "compact pools %d, unused state" % pools)
    assert dev.mode_seeds()[0] >= len(seeds)
    orc.find_blocks(case.k, case.b, case.m)
    dev.set_used(orc.used_bitmap(st.chr_start()))
    _compare_all(case, st, dev, orc, seeds, "compact pools %d, final state" % pools)
    dev.close()
    st, p, dev = _setup(case, compact_pools=pools, wide_threshold=1)       # (every launch of two seeds or more begins in the compact variant)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")


@pytest.mark.parametrize("sparse", [-1, 0, 1])
def test_host_settled_seeds_and_sparse_rounds_on_gpu(built, case, sparse):
    """lcb_hooks.sparse_rounds: -1 = every result from the device, 0 (default) = the host settles the seeds without an unused occurrence, 1 = also sparse
    speculative launches (only the first phase of every cluster of seeds). Blocks, failure_ and blocksFound_ are the reference's in all three."""
    st, p, dev = _setup(case)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4, sparse_rounds=sparse)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    assert finder.stats["failures"] == int(summary["failure"]) and finder.stats["blocks_found"] == int(summary["blocksFound"])
    assert sparse >= 0 or finder.stats["host_dead"] == 0


@pytest.mark.parametrize("fixed,phases", [(1, 1), (1, 7), (0, 64)])
def test_round_engine_variants_on_gpu(built, case, fixed, phases):
    """Round size must not change the result: one phase per launch (the reference's schedule), a fixed speculative round of
    7 phases, and the adaptive default all give the reference's block list."""
    st, p, dev = _setup(case)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4, round_fixed=fixed, round_phases=phases)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    assert finder.stats["failures"] == int(summary["failure"])


@pytest.mark.parametrize("knobs", [{"max_views": -1}, {"max_views": 2, "round_fixed": 1, "round_phases": 64}, {"predict_f": 1},
                                   {"predict_f": 2, "max_jobs": 8}, {"eager_phases": -1}, {"max_jobs": 100000}])
def test_predictive_engine_knobs_on_gpu(built, case, knobs):
    """Predictions only cost launches, never correctness: without predicted views, with too few of them, with other F
    predictions, a tiny or a huge job cap or no look-ahead the block list is the reference's."""
    st, p, dev = _setup(case)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4, **knobs)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    assert finder.stats["failures"] == int(summary["failure"]) and finder.stats["blocks_found"] == int(summary["blocksFound"])


def test_cli_drop_in(built, case, tmp_path):
    """The sibeliaz-lcb executable with the wrapper's argv (sibeliaz:146) writes the reference's blocks_coords.gff."""
    out = str(tmp_path / "cli")
    r = subprocess.run([os.path.join(ROOT, "sibeliaz_amd", "bin", "sibeliaz-lcb"), "--graph", case.graph, case.fasta, "-k", str(case.k), "-b", str(case.b),
                        "-o", out, "-m", str(case.m), "-t", "4", "--abundance", str(case.a), "--chunks", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("Loading the graph...\nAnalyzing the graph...\n[")
    assert "]\nGenerating the output...\nBlocks found: " in r.stdout
    assert open(os.path.join(out, "blocks_coords.gff")).read() == case.golden("ref.gff")
    chunks = b"".join(open(os.path.join(out, "%d.tmp" % i), "rb").read() for i in range(4))
    if "chunks4" in case.meta["sha256"]:
        import hashlib
        assert hashlib.sha256(chunks).hexdigest() == case.meta["sha256"]["chunks4"]


def test_rccl_exchange_path_single_rank(built, case):
    """The native multi-rank path on the one GPU of the test box: an RCCL communicator of world 1 (ncclCommInitRank), every
    launch packed, all-gathered with ncclAllGather through device staging buffers and unpacked (exchange_always), identical
    commit. Same blocks as the reference. (More ranks need more GPUs: RCCL refuses two ranks on one device; the multi-rank
    logic itself runs with 2-4 ranks under the emulator and over gloo in the CPU suite.)"""
    st, p, dev = _setup(case)
    comm = sibeliaz_amd.Comm(dev, sibeliaz_amd.Comm.unique_id(), 0, 1)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4, comm=comm, exchange_always=1)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    assert finder.stats["exchanges"] > 0
    comm.close()


def test_find_blocks_gpus_one_process(built, case):
    """lcb_find_blocks_gpus (what sibeliaz-lcb does with LCB_GPUS=N): devices, ncclCommInitAll and one host thread per GPU
    inside one process — with the single GPU of the test box, forced through the exchange path."""
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, threads=4, abundance=case.a)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocksGpus(case.m, case.b, [0], threads=4, exchange_always=1)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    assert finder.stats["exchanges"] > 0


@pytest.mark.parametrize("lazy_span", [-1, 0])
def test_screened_compact_rounds(built, case, lazy_span):
    """wide_threshold=1 / screen_min=1 make even the small rounds of the golden cases take the screened compact path (the screening
    kernel finalises the seeds whose Path::Init finds nothing), with and without lazy round tails: the reference's blocks."""
    st, p, dev = _setup(case, wide_threshold=1, screen_min=1)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4, lazy_span=lazy_span)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    assert finder.stats["failures"] == int(summary["failure"]) and finder.stats["blocks_found"] == int(summary["blocksFound"])
    assert (finder.stats["lazy_seeds"] > 0) == (lazy_span == 0 and finder.stats["rounds"] > 0 and len(st.seeds(4)) > 256)


@pytest.mark.parametrize("knobs,dev_opts", [({"sync_jobs": 1}, {}), ({}, {"side_lanes": 1}), ({"max_jobs": 6}, {"side_lanes": 2}),
                                             ({"round_fixed": 1, "round_phases": 64, "max_views": 3}, {}), ({"predict_f": 1}, {"side_lanes": 3, "wide_slots": 4, "big_slots": 2})])
def test_side_lanes_on_gpu(built, case, knobs, dev_opts):
    """Asynchronous job batches: a stop of the ordered commit waits only for the results it cannot go on without, the rest of its
    plan runs on a side lane (own streams, buffers, slots, views) while the commit goes on and reads/marks the live bitmap. Same
    blocks as the reference with the lanes off (sync_jobs), with one lane (batches give way to each other), with a tiny job cap,
    with view starvation in one whole-input round, and with few workgroups per lane."""
    st, p, dev = _setup(case, **dev_opts)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4, **knobs)
    got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
    assert got == case.golden("pretrim.tsv")
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    assert finder.stats["failures"] == int(summary["failure"]) and finder.stats["blocks_found"] == int(summary["blocksFound"])
    if knobs.get("sync_jobs"):
        assert finder.stats["side_batches"] == 0
    elif finder.stats["recompute_launches"] > 2:
        assert finder.stats["side_batches"] > 0 and finder.stats["side_jobs"] >= finder.stats["side_taken"]
    # a second pass on the same device (lanes drained and reused) gives the same blocks
    blocks2 = finder.FindBlocks(case.m, case.b, device=dev, threads=4, **knobs)
    assert blocks.tobytes() == blocks2.tobytes()


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_footprints_cover_every_read_on_gpu(built, case, mode):
    """The property behind the engine's exactness rule, on the device, for every kernel variant: a seed's footprint covers every
    position whose `used` bit its computation read as 0 - so with EVERY other unused position set to used the seed must give the
    same result (and with the footprint itself unused nothing it read as 0 changed). Checked for the seeds that produce a block."""
    st, p, dev = _setup(case, start_mode=mode)
    seeds = st.seeds(4)[:1200]
    off, inst, fp_off, fp = dev.process_seeds_fp(seeds)
    n_pos = st.n_positions()
    words = (n_pos + 31) // 32 + 1
    picked = [i for i in range(len(seeds)) if off[i + 1] - off[i] > 1][:: max(1, (len(seeds) // 40))][:40]
    assert picked, "no seed of the case yields a block"
    for i in picked:
        bits = np.ones(words * 32, dtype=bool)
        for lo, hi in fp[int(fp_off[i]):int(fp_off[i + 1])]:
            bits[int(lo):int(hi) + 1] = False
        dev.set_used(np.packbits(bits, bitorder="little").view("<u4"))
        off2, inst2, _, _ = dev.process_seeds_fp(seeds[i:i + 1])
        assert inst2.tobytes() == inst[int(off[i]):int(off[i + 1])].tobytes(), "seed %d: a read outside its footprint changed the result (variant %d)" % (i, mode)
    dev.reset_used()


@pytest.mark.parametrize("synth_seed", [7101, 7102, 7103, 7104, 7105, 7106, 7107, 7108])
def test_footprints_cover_every_read_on_random_inputs(built, tmp_path, synth_seed):
    """The same property on RANDOM inputs (the bug class it guards - a read outside the footprint - showed on 6 of 11 random inputs under
    the emulator in round 2 and on none of the goldens), for all four kernel variants, in three `used` states - all unused,
    random runs of used positions (any bitmap is a legal state for the kernels), the final state of a whole FindBlocks - and for up to
    120 seeds that yield a block per state (the heavy head of the seed order and a sample of the rest)."""
    import bench
    fa, gr = str(tmp_path / "g.fa"), str(tmp_path / "g.bin")
    subprocess.check_call([os.path.join(bench.BIN, "lcb-synth"), "-o", fa] + ("--strains 7 --segments 50 --keep 0.8 --swap 0.08 --invert 0.15 --sub 0.03 --indel 0.004 "
                          "--filler-frac 0.3 --filler-min 100 --filler-max 1500 --repeat-families 4 --repeat-copies 6 --repeat-len 500 --seg-min 300 --seg-max 4000 --seed %d" % synth_seed).split())
    subprocess.check_call([os.path.join(bench.BIN, "lcb-mkgraph"), "-k", "15", "-o", gr, fa], stderr=subprocess.DEVNULL)
    k, b, m, a = 15, 200, 50, 150
    st = sibeliaz_amd.JunctionStorage(gr, [fa], k, threads=4, abundance=a)
    p = sibeliaz_amd.Params.make(k, b=b, m=m)
    orc = Oracle(gr, [fa], k, a)
    orc.find_blocks(k, b, m)
    final = np.array(orc.used_bitmap(st.chr_start()), dtype="<u4")
    n_pos = st.n_positions()
    words = (n_pos + 31) // 32 + 1
    rng = np.random.default_rng(synth_seed)
    rnd = np.zeros(words * 32, dtype=bool)
    for _ in range(max(1, n_pos // 2000)):
        q = int(rng.integers(0, n_pos)); rnd[q:q + int(rng.integers(20, 300))] = True
    rnd[n_pos:] = False
    states = {"unused": np.zeros(words * 32, dtype=bool), "random": rnd, "final": np.unpackbits(np.resize(final, words).view(np.uint8), bitorder="little").astype(bool)}
    checked = 0
    for mode in (1, 2, 3, 4):
        dev = sibeliaz_amd.Device(st, p, 0, start_mode=mode)
        seeds = st.seeds(4)
        for name, base in states.items():
            dev.set_used(np.packbits(base, bitorder="little").view("<u4")[:words])
            off, inst, fp_off, fp = dev.process_seeds_fp(seeds)
            good = [i for i in range(len(seeds)) if off[i + 1] - off[i] > 1]
            picked = sorted(set(good[:50] + good[50:: max(1, len(good) // 70)][:70]))       # the heavy head of the seed order and a sample of the rest
            if mode >= 3: picked = picked[:40]                                                # (big, huge: one seed per CU or fewer, every launch is slow)
            for i in picked:
                bits = np.ones(words * 32, dtype=bool)
                for lo, hi in fp[int(fp_off[i]):int(fp_off[i + 1])]:
                    bits[int(lo):int(hi) + 1] = base[int(lo):int(hi) + 1]            # inside the footprint: the state the seed saw
                dev.set_used(np.packbits(bits, bitorder="little").view("<u4")[:words])
                off2, inst2, _, _ = dev.process_seeds_fp(seeds[i:i + 1])
                assert inst2.tobytes() == inst[int(off[i]):int(off[i + 1])].tobytes(), "input %d, %s state, seed %d: a read outside its footprint changed the result (variant %d)" % (synth_seed, name, i, mode)
                checked += 1
        dev.close()
    assert checked >= 20, "too few block-producing seeds (%d) - the generator parameters no longer fit" % checked


def test_persistent_gpu_set(built, case):
    """lcb_gpus_create / lcb_gpus_find_blocks / lcb_gpus_destroy: devices, tables and the RCCL communicator live across passes (the handle
    bench.py --gpus N and sibeliaz-lcb LCB_GPUS=N use) - with the one GPU of the test box, forced through the exchange path; two passes."""
    st = sibeliaz_amd.JunctionStorage(case.graph, [case.fasta], case.k, threads=4, abundance=case.a)
    p = sibeliaz_amd.Params.make(case.k, b=case.b, m=case.m)
    gpus = sibeliaz_amd.GpuSet(st, p, [0], always_comm=True)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    for _ in range(2):
        blocks = finder.FindBlocksOnSet(gpus, threads=4, exchange_always=1)
        got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
        assert got == case.golden("pretrim.tsv")
        assert finder.stats["exchanges"] > 0
    gpus.close()


@pytest.mark.parametrize("knobs,dev_opts", [({}, {}), ({"max_jobs": 6}, {"side_lanes": 1}), ({"round_fixed": 1, "round_phases": 1}, {}), ({}, {"start_mode": 3}),
                                            ({"round_fixed": 1, "round_phases": 64}, {"wide_slots": 8, "big_slots": 8})])
def test_early_critical_launch_on_gpu(built, case, knobs, dev_opts):
    """With side lanes the results a stop cannot go on without (the re-processing of the stopping seed, the missing phase-start results of
    the phase about to start) are launched before the dry run that plans the rest of the stop's jobs (an asynchronous first launch in
    the variant a synchronous call would start with, lcb_device_process_begin) and collected after the side batch has been started.
    Same blocks and conflict count as the reference - with the default options, one lane and a tiny job cap, one-phase rounds, every
    seed forced into the big variant, and lanes so small that batches are refused (the rest of a plan then runs synchronously)."""
    st, p, dev = _setup(case, **dev_opts)
    finder = sibeliaz_amd.BlocksFinder(st, case.k)
    summary = dict(ln.split("\t") for ln in case.golden("summary.txt").splitlines())
    for _ in range(2):
        blocks = finder.FindBlocks(case.m, case.b, device=dev, threads=4, **knobs)
        got = "".join("%d\t%d\t%d\t%d\n" % (b["id"], b["chr"], b["start"], b["end"]) for b in blocks)
        assert got == case.golden("pretrim.tsv")
        assert finder.stats["failures"] == int(summary["failure"]) and finder.stats["blocks_found"] == int(summary["blocksFound"])
        if finder.stats["recompute_launches"] > 0 and not dev_opts.get("start_mode"):
            assert finder.stats["early_critical"] > 0
