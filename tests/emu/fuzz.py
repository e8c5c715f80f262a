"""TEST-ONLY randomized parity campaign: random small genome sets and parameters, the unmodified device
code + the product's round engine under the wavefront emulator (emu_check) against the CPU oracle.

    python tests/emu/fuzz.py <seconds> [first case number]        # log: $LCB_FUZZ_DIR/fuzz.log (default /tmp/lcb_fuzz)

Each case runs `find` with the shipped (non-stats) kernel instantiation and random engine knobs (round size, number of predicted
views, F prediction, job cap), `seeds-init` / `seeds-final` with the footprint-completeness check (EMU_FP_CHECK: every unused position
outside a seed's footprint is set to used and the oracle must still reproduce the result), `seeds-init` with event counters, and one
multi-wavefront variant (wide or big mode
with helper wavefronts) on the heaviest seeds, and `find` once more with the engine's asynchronous features (side lanes with random delays,
late batches and refused batches, the early critical launch, lazy round tails of several spans), and two runs with positions as (segment, offset)
pairs (small segments, gaps between them, flat indices beyond 2^32). Failing cases keep their inputs. tests/test_fuzz_emu.py runs a fixed handful of
cases inside the CPU suite; the open-ended campaign is this script.
"""
import os, random, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = ROOT + "/sibeliaz_amd/bin"
EMU = os.environ.get("LCB_FUZZ_EMU") or ROOT + "/tests/emu/build/emu_check"      # (another build of the harness)


def case_params(i, small=False):
    """Deterministic in i: generator arguments, (k, b, m, a) and the emulator runs of case i."""
    rnd = random.Random(i)
    strains = rnd.choice([2, 3, 5, 8] if small else [2, 3, 5, 8, 12, 20])
    segs = rnd.choice([2, 4] if small else [2, 4, 8, 12])
    k = rnd.choice([11, 15, 21, 25])
    b = rnd.choice([40, 100, 200, 400])
    m = rnd.choice([30, 50, 100, 250])
    a = rnd.choice([4, 8, 150])
    synth = ["--strains", str(strains), "--segments", str(segs), "--seg-min", str(rnd.choice([300, 1000, 3000])), "--seg-max", str(rnd.choice([3000, 6000])),
             "--keep", str(rnd.choice([0.7, 0.9, 1.0])), "--swap", str(rnd.choice([0, 0.1, 0.3])), "--invert", str(rnd.choice([0, 0.1, 0.3])),
             "--sub", str(rnd.choice([0.005, 0.02, 0.05])), "--indel", str(rnd.choice([0, 0.002, 0.01])), "--filler-frac", str(rnd.choice([0, 0.2])),
             "--repeat-families", str(rnd.choice([0, 1, 3])), "--repeat-copies", str(rnd.choice([2, 5, 10])), "--repeat-len", str(rnd.choice([200, 800])),
             "--tandem", str(rnd.choice([0, 0.2])), "--nrun", str(rnd.choice([0, 0.1])), "--chromosomes", str(rnd.choice([1, 1, 3])), "--seed", str(1000 + i)]
    runs = [("find", {"EMU_NOSTATS": "1", "EMU_ROUNDS": rnd.choice(["1", "7", "256"]), "EMU_VIEWS": rnd.choice(["0", "3", "64"]),
                      "LCB_PREDICT_F": rnd.choice(["1", "2", "3"]), "EMU_CONCURRENCY": rnd.choice(["4", "64", "16384"])}),
            ("seeds-init", {"EMU_NOSTATS": "1", "EMU_FP_CHECK": "1", "EMU_LIMIT": "1500"}), ("seeds-final", {"EMU_NOSTATS": "1", "EMU_FP_CHECK": "1", "EMU_LIMIT": "1500"}),
            ("seeds-init", {}),
            rnd.choice([("medium", {"EMU_NW": "16", "EMU_NOSTATS": "1", "EMU_LIMIT": "150"}), ("big", {"EMU_NW": "8", "EMU_NOSTATS": "1", "EMU_LIMIT": "150"}),
                        ("medium", {"EMU_NW": "8", "EMU_LIMIT": "150"}), ("big", {"EMU_NW": "4", "EMU_LIMIT": "150"})])]
    # the engine's asynchronous job batches (eager / late stand-ins, delayed visibility, refused batches) with the early critical launch
    # (or a processor that refuses it), and lazy round tails of several spans (drawn after everything else: earlier cases keep their runs)
    env3 = {"EMU_NOSTATS": "1", "EMU_ROUNDS": rnd.choice(["1", "7", "256"]), "EMU_SIDE_LANES": rnd.choice(["1", "2", "4"]),
            "EMU_SIDE_DELAY": rnd.choice(["0", "1", "3", "1000"]), "EMU_CONCURRENCY": rnd.choice(["4", "64", "16384"])}
    if rnd.random() < 0.5: env3["EMU_SIDE_LATE"] = "1"
    if rnd.random() >= 0.6: env3["EMU_NO_EARLY"] = "1"
    if rnd.random() < 0.3: env3["EMU_SIDE_CAP"] = rnd.choice(["8", "50"])
    if rnd.random() >= 0.4: env3["LCB_LAZY_SPAN"] = rnd.choice(["0", "2", "16", "64"])      # (0: off; default 8)
    runs.append(("find", env3))
    # positions as (segment, offset) pairs - the SEG kernel instantiations: segments of a few hundred to a few thousand positions, with and
    # without unused table space between them (every eighth case: more than 2^32 positions of it, i.e. flat indices beyond 32 bits), per-seed
    # results with the footprint check and the whole engine (drawn last: earlier runs of a case stay what they were)
    seg = {"EMU_SEG_CAP": rnd.choice(["300", "1000", "4000"])}
    gap = rnd.choice(["0", "0", "77", "65536", "100003", "100003", "4300000000", "4300000000"])
    if gap != "0": seg["EMU_SEG_GAP"] = gap
    big = gap == "4300000000"
    # big: two or three segments at most (the gapped tables are address space, the bitmap is real memory); otherwise at most about 16 -
    # the device tables take 32, and many strains x chromosomes with a small cap would ask for more (case 7007: the plan's loud error)
    seg["EMU_SEG_MAX"] = "2" if big else "16"
    runs.append((rnd.choice(["seeds-init", "seeds-final"]), dict(seg, EMU_NOSTATS="1", EMU_FP_CHECK="1", EMU_LIMIT="600", EMU_NW=rnd.choice(["1", "2"]))))
    if not big: runs.append(("find", dict(seg, EMU_NOSTATS="1", EMU_ROUNDS=rnd.choice(["1", "7", "256"]), EMU_SIDE_LANES=rnd.choice(["1", "3"]), EMU_SIDE_DELAY=rnd.choice(["0", "2"]))))
    # round 6 (drawn last of all): the engine runs with sparse speculative launches / without host-settled seeds in half of the cases (default: host-settled seeds only)
    sr = rnd.choice(["1", "1", "-1", None, None, None])
    if sr is not None:
        for mode, env in runs:
            if mode == "find": env["LCB_SPARSE_ROUNDS"] = sr
    # ... and a third of the cases run the compact variant with its small pools (128 instances / 512 vote slots; lcb_device_opts.compact_pools): every run of the case
    if rnd.random() < 1.0 / 3:
        for mode, env in runs: env["EMU_COMPACT_SMALL"] = "1"
    return synth, (k, b, m, a), runs, (strains, segs)


def run_case(i, work, small=False, timeout=900):
    """-> (description, [(mode, env, ok (True / False / None = timeout), stderr tail)])"""
    synth, (k, b, m, a), runs, (strains, segs) = case_params(i, small)
    d = os.path.join(work, "c%d" % i)
    os.makedirs(d, exist_ok=True)
    fa, gr = d + "/g.fa", d + "/g.bin"
    subprocess.check_call([BIN + "/lcb-synth", "-o", fa] + synth, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([BIN + "/lcb-mkgraph", "-k", str(k), "-o", gr, fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res = []
    for mode, env in runs:
        try:
            r = subprocess.run([EMU, gr, fa, str(k), str(b), str(m), str(a), mode, d + "/out"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=timeout)
            ok = r.returncode == 0
            tail = "" if ok else r.stderr[-400:].replace("\n", " | ")
        except subprocess.TimeoutExpired:
            ok = None; tail = "timeout"
        res.append((mode, env, ok, tail))
    desc = "case %d strains=%d segs=%d k=%d b=%d m=%d a=%d" % (i, strains, segs, k, b, m, a)
    if all(x[2] for x in res):
        subprocess.call(["rm", "-rf", d])
    return desc, res, synth


def main():
    work = os.environ.get("LCB_FUZZ_DIR", "/tmp/lcb_fuzz")
    os.makedirs(work, exist_ok=True)
    t_end = time.time() + float(sys.argv[1])
    i = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    log = open(os.path.join(work, "fuzz.log"), "a")
    nfail = 0
    while time.time() < t_end:
        try:
            desc, res, synth = run_case(i, work)
        except Exception as e:
            log.write("case %d: generation failed %s\n" % (i, e)); log.flush(); i += 1; continue
        bad = [x for x in res if x[2] is False]
        nfail += len(bad)
        log.write("%s : %s\n" % (desc, " ".join("%s=%s" % (x[0], "ok" if x[2] else ("TIMEOUT" if x[2] is None else "FAIL")) for x in res)))
        for x in bad:
            log.write("   FAIL %s %s synth=%s :: %s\n" % (x[0], x[1], " ".join(synth), x[3]))
        log.flush()
        i += 1
    log.write("done: next case %d, failures %d\n" % (i, nfail)); log.flush()


if __name__ == "__main__":
    main()
