#!/usr/bin/env python3
"""bench.py — seed vertices/sec through the MI355X block finder (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ecoli62|ecoli10|...]

A "step" is one full pass of the hot path over the workload's sorted seed list: BlocksFinder::FindBlocks' phase loop
(every seed through the per-seed HIP kernels, ordered commit, GPU re-processing of conflicts) with the junction tables
already resident in HBM. value = seeds / step time (whole job over all GPUs).

Workload at N = 1: the largest single-GPU configuration of BASELINE.json, configs[2] "62 E. coli strains, k=15", restated
as the synthetic pangenome of SURVEY.md §8d at its stated size (62 strains, 281 Mbp, lcb-synth seed 1002 + lcb-mkgraph,
untimed; there are no genomes and no network here). `--workload ecoli10` is configs[1] (10 strains, 45 Mbp).

The JSON line also carries
  roofline      the process kernels (lcb_process_kernel, all variants): ALGORITHMIC bytes per launch (SURVEY.md §8d formula
                over the reference-semantics event counters, counted by the kernels themselves in one untimed stats-mode pass of
                the engine) / average launch duration, measured in this run with HIP events on the kernels' stream; against the
                8 TB/s HBM peak, with this GPU's measured STREAM-triad rate beside it. `traffic` (PMC) cannot be measured inside
                this run and is null; `traffic_profiled` quotes the per-launch figure of the committed rocprofv3 --pmc
                passes of this same command (profiles/r04/pmc_traffic.json) for the default workload.
  cpu_baseline  the UNMODIFIED reference sibeliaz-lcb (oracle/_ref, built from /root/reference in the build container) timed
                on this box's host cores: once at -t 32 (the cap of the reference's wrapper script, sibeliaz:139) on the WHOLE
                workload `value` is measured on, its blocks_coords.gff compared (md5) with the timed run's; and on bounded samples
                of the same workload (same generator and parameters, 1/10 of the ancestor's segments; 1/40 for -t 1), median of
                3, banner to banner, at -t 1, -t 64 and -t <all hardware threads>. `--sample-cpu-baseline` skips the whole-workload
                run (the round-2 protocol).
  wall_clock    metric part 2: the whole sibeliaz-lcb process (load, seeds, upload, phase loop, GFF) on the workload.
N > 1: either under torch.distributed.run (one rank per GPU: torch only ships the RCCL unique id and takes the max of the step
times) or called directly (`python bench.py --gpus N`): one process, a persistent lcb_gpus set - one host thread per GPU inside the
library, ncclCommInitAll. Either way the all-gathers of the engine are ncclAllGather calls inside the C++ library (csrc/comm.hip).
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "sibeliaz_amd", "bin")

_COMMON = ("--keep 0.8 --swap 0.05 --invert 0.10 --sub 0.02 --indel 0.002 --filler-frac 0.25 --filler-min 200 --filler-max 3000 "
           "--repeat-families 3 --repeat-copies 10 --repeat-len 800 --seg-min 500 --seg-max 8000 ")


def _wl(strains, segments, seed, desc, k=15, b=200, m=50, a=150):
    return dict(synth="--strains %d --segments %d %s--seed %d" % (strains, segments, _COMMON, seed), k=k, b=b, m=m, a=a, desc=desc)


WORKLOADS = {
    # SURVEY.md §8d config 3 at its stated size: 62 strains x ~4.5 Mbp = 281 Mbp, seed 1002
    "ecoli62": _wl(62, 1200, 1002, "62 synthetic E. coli-like strains (281 Mbp), k=15, b=200, m=50, a=150 [BASELINE configs[2]; SURVEY.md §8d config 3, lcb-synth seed 1002]"),
    # the same genomes and graph with the abundance threshold the reference's README derives for them: a = 2 * 62 * 7 (README.md:161-175; SURVEY.md section 8d "report both")
    "ecoli62_a868": dict(_wl(62, 1200, 1002, "62 synthetic E. coli-like strains (281 Mbp), k=15, b=200, m=50, a=868 = 2 * N * D [SURVEY.md §8d config 3, second abundance]", a=868), files="ecoli62"),
    # bounded samples of it for the CPU baseline: same generator and parameters, 1/10 and 1/40 of the ancestor's segments
    "ecoli62_small": _wl(62, 120, 1002, "62 synthetic strains (32 Mbp: config 3 with 1/10 of the segments), k=15, b=200, m=50, a=150"),
    "ecoli62_tiny": _wl(62, 30, 1002, "62 synthetic strains (8 Mbp: config 3 with 1/40 of the segments), k=15, b=200, m=50, a=150"),
    # SURVEY.md §8d config 2
    "ecoli10": _wl(10, 1200, 1001, "10 synthetic E. coli-like strains (45 Mbp), k=15, b=200, m=50, a=150 [BASELINE configs[1]; SURVEY.md §8d config 2, lcb-synth seed 1001]"),
    "ecoli10_small": _wl(10, 120, 1001, "10 synthetic strains (4.5 Mbp: config 2 with 1/10 of the segments), k=15, b=200, m=50, a=150"),
    "ecoli10_tiny": _wl(10, 30, 1001, "10 synthetic strains (1.1 Mbp), k=15, b=200, m=50, a=150"),
    # SURVEY.md §8d configs 4 / 5 (k=25, many chromosomes, repeat families that exercise the abundance filter), scaled to what one
    # box generates in about a minute: 8 x 24 chromosomes x ~6.5 Mbp = 1.24 Gbp and 16 x 20 x ~3.2 Mbp = 1.0 Gbp
    "primates8_scaled": dict(synth="--strains 8 --chromosomes 24 --segments 1200 --seg-min 5000 --seg-max 200000 --keep 0.9 --swap 0.05 --invert 0.05 --sub 0.01 "
                                   "--indel 0.001 --filler-frac 0.25 --filler-min 200 --filler-max 3000 --repeat-families 20 --repeat-copies 100 --repeat-len 800 --seed 1003",
                             k=25, b=200, m=50, a=150, desc="8 synthetic strains x 24 chromosomes (1.24 Gbp), k=25, b=200, m=50, a=150 [SURVEY.md §8d config 4, scaled]"),
    "mice16_scaled": dict(synth="--strains 16 --chromosomes 20 --segments 600 --seg-min 5000 --seg-max 200000 --keep 0.9 --swap 0.05 --invert 0.05 --sub 0.005 "
                                "--indel 0.0005 --filler-frac 0.25 --filler-min 200 --filler-max 3000 --repeat-families 20 --repeat-copies 100 --repeat-len 800 --seed 1004",
                          k=25, b=200, m=50, a=150, desc="16 synthetic strains x 20 chromosomes (1.0 Gbp), k=25, b=200, m=50, a=150 [SURVEY.md §8d config 5, scaled]"),
    # config 4's shape at the largest size the build container can run the unmodified reference on (its hash: tests/golden/fullsize_scaled.json):
    # 8 x 24 chromosomes x ~21.5 Mbp = 4.1 Gbp (5 600 segments)
    "primates8_4g": dict(synth="--strains 8 --chromosomes 24 --segments 5600 --seg-min 5000 --seg-max 200000 --keep 0.9 --swap 0.05 --invert 0.05 --sub 0.01 "
                               "--indel 0.001 --filler-frac 0.25 --filler-min 200 --filler-max 3000 --repeat-families 20 --repeat-copies 100 --repeat-len 800 --seed 1003",
                         k=25, b=200, m=50, a=150, desc="8 synthetic strains x 24 chromosomes (4.1 Gbp), k=25, b=200, m=50, a=150 [SURVEY.md §8d config 4, scaled]"),
    # config 5's shape at the same scale: 16 x 20 chromosomes x ~13 Mbp = 4.1 Gbp (2 720 segments); reference hash in tests/golden/fullsize_scaled.json
    "mice16_4g": dict(synth="--strains 16 --chromosomes 20 --segments 2720 --seg-min 5000 --seg-max 200000 --keep 0.9 --swap 0.05 --invert 0.05 --sub 0.005 "
                            "--indel 0.0005 --filler-frac 0.25 --filler-min 200 --filler-max 3000 --repeat-families 20 --repeat-copies 100 --repeat-len 800 --seed 1004",
                      k=25, b=200, m=50, a=150, desc="16 synthetic strains x 20 chromosomes (4.1 Gbp), k=25, b=200, m=50, a=150 [SURVEY.md §8d config 5, scaled]"),
    # the same two shapes at the size of a parity test (tests/test_gpu_fullsize.py; the reference needs ~20 s for each on 8 cores):
    # 8 x 24 chromosomes = 186 Mbp at 1 % divergence, 16 x 20 chromosomes = 217 Mbp at 0.5 %
    "primates8_test": dict(synth="--strains 8 --chromosomes 24 --segments 240 --seg-min 5000 --seg-max 200000 --keep 0.9 --swap 0.05 --invert 0.05 --sub 0.01 "
                                 "--indel 0.001 --filler-frac 0.25 --filler-min 200 --filler-max 3000 --repeat-families 20 --repeat-copies 100 --repeat-len 800 --seed 1003",
                           k=25, b=200, m=50, a=150, desc="8 synthetic strains x 24 chromosomes (186 Mbp), k=25, b=200, m=50, a=150 [SURVEY.md §8d config 4 at test size]"),
    "mice16_test": dict(synth="--strains 16 --chromosomes 20 --segments 120 --seg-min 5000 --seg-max 200000 --keep 0.9 --swap 0.05 --invert 0.05 --sub 0.005 "
                              "--indel 0.0005 --filler-frac 0.25 --filler-min 200 --filler-max 3000 --repeat-families 20 --repeat-copies 100 --repeat-len 800 --seed 1004",
                        k=25, b=200, m=50, a=150, desc="16 synthetic strains x 20 chromosomes (217 Mbp), k=25, b=200, m=50, a=150 [SURVEY.md §8d config 5 at test size]"),
    # bounded samples of the two k = 25 shapes for the -t 1 leg of the CPU baseline (a quarter of the segments)
    "primates8_tiny": dict(synth="--strains 8 --chromosomes 24 --segments 60 --seg-min 5000 --seg-max 200000 --keep 0.9 --swap 0.05 --invert 0.05 --sub 0.01 "
                                 "--indel 0.001 --filler-frac 0.25 --filler-min 200 --filler-max 3000 --repeat-families 20 --repeat-copies 100 --repeat-len 800 --seed 1003",
                           k=25, b=200, m=50, a=150, desc="8 synthetic strains x 24 chromosomes (config 4 shape with 1/4 of the test size's segments), k=25, b=200, m=50, a=150"),
    "mice16_tiny": dict(synth="--strains 16 --chromosomes 20 --segments 30 --seg-min 5000 --seg-max 200000 --keep 0.9 --swap 0.05 --invert 0.05 --sub 0.005 "
                              "--indel 0.0005 --filler-frac 0.25 --filler-min 200 --filler-max 3000 --repeat-families 20 --repeat-copies 100 --repeat-len 800 --seed 1004",
                        k=25, b=200, m=50, a=150, desc="16 synthetic strains x 20 chromosomes (config 5 shape with 1/4 of the test size's segments), k=25, b=200, m=50, a=150"),
    # same genomes as config 2, other parameters (k=25, b=400, m=100)
    "ecoli10_k25": _wl(10, 1200, 1001, "10 synthetic E. coli-like strains (45 Mbp), k=25, b=400, m=100, a=150", k=25, b=400, m=100),
}
SAMPLES = {"ecoli62": ("ecoli62_small", "ecoli62_tiny"), "ecoli10": ("ecoli10", "ecoli10_small"), "ecoli10_k25": ("ecoli10_k25", "ecoli10_small"),
           # the k = 25 shapes (configs 4 / 5 at test size): the reference needs seconds for them, so the "sample" of the -t 32 / -t 64 / all-threads legs
           # is the workload itself
           "primates8_test": ("primates8_test", "primates8_tiny"), "mice16_test": ("mice16_test", "mice16_tiny")}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def ensure_workload(name):
    w = WORKLOADS[name]
    d = os.path.join(os.environ.get("LCB_BENCH_DIR", "/tmp/lcb_bench"), w.get("files", name))       # (workloads that differ in parameters only share their files)
    os.makedirs(d, exist_ok=True)
    fa, gr = os.path.join(d, "genomes.fa"), os.path.join(d, "graph.bin")
    import fcntl
    with open(os.path.join(d, ".lock"), "w") as lock:         # (several processes may ask for the same workload: pytest-xdist workers, ranks)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not (os.path.exists(fa) and os.path.exists(gr) and os.path.exists(gr + ".ok")):
            t = time.time()
            subprocess.check_call([os.path.join(BIN, "lcb-synth"), "-o", fa] + w["synth"].split())
            subprocess.check_call([os.path.join(BIN, "lcb-mkgraph"), "-k", str(w["k"]), "-o", gr, fa], stderr=subprocess.DEVNULL)
            open(gr + ".ok", "w").write("ok")
            log("bench: generated workload %s in %.1fs" % (name, time.time() - t))
    return dict(w, name=name, fasta=fa, graph=gr, dir=d)


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def run_reference(w, threads, tag, limit_s=None):
    """One run of the unmodified reference: (analyze seconds banner to banner, whole-process seconds, gff path); None if it is not
    there or failed; the string "timeout" if it ran longer than limit_s (it is killed: every leg of the protocol is bounded)."""
    import threading
    ref = os.path.join(ROOT, "oracle", "_ref", "sibeliaz-lcb-ref")
    if not os.path.exists(ref):
        return None
    out = os.path.join(w["dir"], "ref_out_%s" % tag)
    cmd = [ref, "--graph", w["graph"], w["fasta"], "-k", str(w["k"]), "-b", str(w["b"]), "-m", str(w["m"]), "-a", str(w["a"]), "-t", str(threads),
           "-o", out, "--noseq"]
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=0)
    killed = []
    timer = None
    if limit_s:
        timer = threading.Timer(limit_s, lambda: (killed.append(1), p.kill()))
        timer.daemon = True
        timer.start()
    marks, buf = {}, ""
    while True:
        ch = p.stdout.read(1)
        if not ch:
            break
        buf += ch
        if ch == "\n":
            for key in ("Analyzing the graph...", "Generating the output..."):
                if key in buf and key not in marks:
                    marks[key] = time.time()
            buf = ""
    p.wait()
    if timer:
        timer.cancel()
    wall = time.time() - t0
    if killed:
        return "timeout"
    if p.returncode != 0 or len(marks) < 2:
        return None
    return marks["Generating the output..."] - marks["Analyzing the graph..."], wall, os.path.join(out, "blocks_coords.gff")


def n_seeds_of(w, threads):
    import sibeliaz_amd
    st = sibeliaz_amd.JunctionStorage(w["graph"], [w["fasta"]], w["k"], threads=threads, abundance=w["a"])
    n = len(st.seeds(threads))
    st.close()
    return n


def our_gff(w, threads, dev_ordinal=0):
    import sibeliaz_amd
    st = sibeliaz_amd.JunctionStorage(w["graph"], [w["fasta"]], w["k"], threads=threads, abundance=w["a"])
    p = sibeliaz_amd.Params.make(w["k"], b=w["b"], m=w["m"])
    dev = sibeliaz_amd.Device(st, p, dev_ordinal)
    f = sibeliaz_amd.BlocksFinder(st, w["k"])
    f.FindBlocks(w["m"], w["b"], device=dev, threads=threads)
    out = os.path.join(w["dir"], "gpu_out")
    f.GenerateOutput(out)
    dev.close()
    st.close()
    return os.path.join(out, "blocks_coords.gff")


def cpu_baseline_quick(workload, threads, our_gff_path, limit_s=120.0):
    """The bounded leg of a `secondary` entry: ONE run of the unmodified reference at -t 32 on the WHOLE workload (the k = 25 test shapes take it
    ~10 s), its GFF compared with the timed run's."""
    host = os.cpu_count() or 1
    t32 = min(32, host)
    whole = ensure_workload(workload)
    s_whole = n_seeds_of(whole, threads)
    r = run_reference(whole, t32, "t%d_whole" % t32, limit_s)
    if r is None or r == "timeout":
        return None
    return {"value": s_whole / r[0], "unit": "seeds/s", "cores": t32, "kind": "reference", "gff_md5_equal": md5(r[2]) == md5(our_gff_path),
            "sample": "the WHOLE benchmarked workload, %s: unmodified reference sibeliaz-lcb (g++ -O3 -DNDEBUG -fopenmp) at -t %d, 1 run, 'Analyzing' to 'Generating' banner "
                      "%.2f s (includes its serial seed enumeration), whole process %.2f s; host has %d hardware threads" % (whole["desc"], t32, r[0], r[1], host)}


def secondary_live(names, threads, no_cpu=False):
    """The other shapes of BASELINE.json, timed by THIS run: `bench.py --workload <name> --steps 3 --warmup 1 --secondary-leg` as a child process each
    (its own device and tables; the parent's are closed by then), its line condensed into one entry."""
    out = []
    for name in names:
        t = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", "3", "--warmup", "1", "--no-cli", "--secondary-leg",
                                "--threads", str(threads)] + (["--no-cpu-baseline"] if no_cpu else []), capture_output=True, text=True, timeout=420)
            line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:       # a secondary entry must never cost the line itself
            out.append({"workload": name, "measured_in_this_run": True, "error": repr(e)})
            continue
        cb = line.get("cpu_baseline")
        out.append({"what": WORKLOADS[name]["desc"], "workload": name, "measured_in_this_run": True, "value": line["value"], "unit": line["unit"],
                    "ms_per_step": line["ms_per_step"], "steps": line["steps"], "warmup": line["warmup"], "seeds": line["config"]["seeds"],
                    "blocks_found": line["config"]["blocks_found"],
                    "roofline": {k: line["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms_per_step", "algorithmic_bytes_per_step",
                                                                  "launches_per_step", "per_kernel", "frac_of_measured_peak")},
                    "cpu_baseline": cb, "speedup_over_cpu_baseline": line["value"] / cb["value"] if cb else None,
                    "sources": source_hash(), "wall_s_of_this_entry": time.time() - t})
    return out


def cpu_baseline(workload, threads, full, our_gff_path, budget_s=600.0):
    """SURVEY.md §8d protocol, every leg bounded: the reference on bounded samples at -t 1 / -t 32 / -t 64 / -t <all hardware threads>
    (median of 3, analyze time banner to banner; its GFF on the -t 32 sample must equal ours) and ONE run at -t 32 on the WHOLE
    workload `value` is measured on - right after the -t 32 sample, with what is left of budget_s as its limit; if it finishes, its GFF
    must equal the timed run's and it is the reported figure, otherwise the -t 32 sample is. The other thread counts get what is left."""
    host = os.cpu_count() or 1
    t_start = time.time()
    small_name, tiny_name = SAMPLES[workload]
    whole, small, tiny = ensure_workload(workload), ensure_workload(small_name), ensure_workload(tiny_name)
    s_whole, s_small, s_tiny = n_seeds_of(whole, threads), n_seeds_of(small, threads), n_seeds_of(tiny, threads)
    per_t = {}

    def leg(t, w, s, reps, tag, limit_s):
        runs = []
        for rep in range(reps):
            limit = max(10.0, min(limit_s, budget_s - (time.time() - t_start)))     # no leg outlives the budget of the whole protocol
            r = run_reference(w, t, tag, limit)
            if r == "timeout":
                per_t[tag] = {"threads": t, "timeout_s": limit, "sample": w["desc"], "sample_seeds": s}
                return "timeout"
            if r is None:
                return None
            runs.append(r)
        an = statistics.median(x[0] for x in runs)
        per_t[tag] = {"threads": t, "seeds_per_s": s / an, "analyze_s_median": an, "wall_s_median": statistics.median(x[1] for x in runs), "runs": len(runs),
                      "sample": w["desc"], "sample_seeds": s}
        return runs[-1][2]

    t32 = min(32, host)
    if leg(1, tiny, s_tiny, 3, "t1_sample", 120) is None:
        return None
    gff_small = leg(t32, small, s_small, 3, "t%d_sample" % t32, 120)
    if gff_small is None or gff_small == "timeout":
        return None
    same_small = md5(gff_small) == md5(our_gff(small, threads))
    # the figure that matters next: -t 32 on the WHOLE workload (before the other thread counts, which get what is left of the budget)
    same_full, whole_tag = None, "t%d_whole" % t32
    if full and small_name == workload:
        per_t[whole_tag] = dict(per_t["t%d_sample" % t32])            # the "sample" is the workload itself (the k = 25 shapes)
        same_full = md5(gff_small) == md5(our_gff_path)
    elif full:
        left = budget_s - (time.time() - t_start) - 45.0               # (45 s kept for the legs below)
        if left > 60:
            gff_ref = leg(t32, whole, s_whole, 1, whole_tag, left)
            if gff_ref is None:
                return None
            if gff_ref != "timeout":
                same_full = md5(gff_ref) == md5(our_gff_path)
    for t in sorted({min(64, host), host} - {t32}):
        if budget_s - (time.time() - t_start) < 15.0:
            per_t["t%d_sample" % t] = {"threads": t, "skipped": "the budget of the protocol was spent"}
            continue
        if leg(t, small, s_small, 1 if t == host and host > 64 else 3, "t%d_sample" % t, 90) is None:
            return None
    on_whole = same_full is not None
    top = per_t[whole_tag] if on_whole else per_t["t%d_sample" % t32]
    return {"value": top["seeds_per_s"], "unit": "seeds/s", "cores": t32, "kind": "reference",
            "sample": "%s%s: unmodified reference sibeliaz-lcb (g++ -O3 -DNDEBUG -fopenmp) at -t %d, %d run(s), 'Analyzing' to 'Generating' banner %.2f s "
                      "(includes its serial seed enumeration); host has %d hardware threads%s" % (
                          "the WHOLE benchmarked workload, " if on_whole else "", top["sample"], t32, top["runs"], top["analyze_s_median"], host,
                          "" if on_whole or not full else "; the run on the whole workload did not finish within its limit (legs['%s'])" % whole_tag),
            "gff_md5_equal": same_full if on_whole else bool(same_small), "gff_md5_equal_on_sample": bool(same_small),
            "stated_comparison": "-t %d: the reference's own wrapper never asks for more (sibeliaz:139), and on this host -t 64 is SLOWER than -t %d (legs['t64_sample']; "
                                 "north_star's 64-thread figure is that leg)" % (t32, t32),
            "legs": per_t,
            "note": "legs named *_sample run on the bounded samples named in them (1/10 of the segments; 1/40 for -t 1); the reference's own wrapper caps -t at 32 "
                    "(sibeliaz:139); every leg has a time limit so that the default run stays within minutes"}


def source_hash():
    """sha256 (16 hex digits) over the sources of the device code and the engine: the event counts were counted by a build of these."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sibeliaz_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".h", ".hip", ".cpp")):
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(c):
    """bytes = 9 N_walk + 15 N_occ + 16 N_compat_call + 1 N_compat_step + 13 N_inst_out."""
    return 9 * c["n_walk"] + 15 * c["n_occ"] + 16 * c["n_compat_call"] + c["n_compat_step"] + 13 * c["n_inst_out"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ecoli62")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sample-cpu-baseline", action="store_true", help="time the reference on the bounded samples only (not on the whole workload)")
    ap.add_argument("--cpu-baseline-budget", type=float, default=360.0, help="seconds all legs of the reference protocol may take together (every leg is cut off at what is left)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the stats-mode counting pass (exploration runs)")
    ap.add_argument("--recount", action="store_true", help="repeat the stats-mode counting pass even if the workload's event counts are known")
    ap.add_argument("--verify-counts", action="store_true", help="count the events again (a 3-minute stats-mode pass) when the committed counts were counted by a build of other sources; "
                    "they are a property of input and parameters, so the default run uses them as they are")
    ap.add_argument("--write-counts", action="store_true", help="store the counts of this run's counting pass in bench_event_counts.json (with the hash of the sources)")
    ap.add_argument("--no-cli", action="store_true", help="skip the whole-process wall-clock run of sibeliaz-lcb")
    ap.add_argument("--device-opt", action="append", default=[], metavar="FIELD=VALUE", help="A/B: a field of lcb_device_opts (e.g. path_cap=8192)")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="FIELD=VALUE", help="A/B: an engine field of lcb_hooks (e.g. max_jobs=256)")
    ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--no-secondary", action="store_true", help="skip the live lines of the k = 25 shapes (configs 4 / 5 at test size) that a default run appends")
    ap.add_argument("--secondary-leg", action="store_true", help="(internal) this process times one `secondary` entry: one bounded reference leg, no entries of its own")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    in_process = world == 1 and args.gpus > 1       # called directly with --gpus N: one process, a persistent lcb_gpus set
    if world != args.gpus and not in_process and not (world == 1 and args.gpus == 1):
        log("bench: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        sys.exit(2)

    import sibeliaz_amd

    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if rank == 0:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "sibeliaz_amd", "build.py"), "tools", "lib", "cli"], stdout=subprocess.DEVNULL)
        ensure_workload(args.workload)
    if world > 1:
        dist.barrier()
    w = ensure_workload(args.workload)

    t = time.time()
    storage = sibeliaz_amd.JunctionStorage(w["graph"], [w["fasta"]], w["k"], threads=args.threads, abundance=w["a"])
    t_load = time.time() - t
    t = time.time()
    seeds = storage.seeds(args.threads)
    t_seeds = time.time() - t
    params = sibeliaz_amd.Params.make(w["k"], b=w["b"], m=w["m"])
    t = time.time()
    dev_opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.device_opt}
    engine_opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.engine_opt}
    gpus = None
    if in_process:
        gpus = sibeliaz_amd.GpuSet(storage, params, list(range(args.gpus)), **dev_opts)      # tables resident in the HBM of every GPU, RCCL initialised
        dev = None
    else:
        dev = sibeliaz_amd.Device(storage, params, local_rank, **dev_opts)          # tables now resident in HBM
    t_upload = time.time() - t
    S = len(seeds)
    log("bench[%d]: P=%d V=%d S=%d load %.2fs seeds %.2fs upload %.2fs" % (rank, storage.n_positions(), storage.GetVerticesNumber(), S, t_load, t_seeds, t_upload))

    comm = None
    if world > 1:
        # the only use of torch in the multi-rank path: ship rank 0's RCCL unique id; the engine's all-gathers are native
        box = [sibeliaz_amd.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = sibeliaz_amd.Comm(dev, box[0], rank, world)

    finder = sibeliaz_amd.BlocksFinder(storage, w["k"])

    def step():
        if gpus is not None:
            finder.FindBlocksOnSet(gpus, seeds=seeds, **engine_opts)
        else:
            finder.FindBlocks(w["m"], w["b"], device=dev, seeds=seeds, comm=comm, **engine_opts)
        return finder.blocks, dict(finder.stats)

    def sync():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if dev is not None:
        dev.kernel_time()
    mode_t0 = dev.mode_time() if dev is not None else None
    sync()
    t0 = time.time()
    last = None
    kernel_ms, kernel_busy_ms, kernel_side_ms, launches, process_ms, plan_ms = 0.0, 0.0, 0.0, 0, 0.0, 0.0
    for _ in range(args.steps):
        last = step()
        kernel_ms += last[1]["kernel_ms"]    # the native engine reports (and resets) this rank's hipEvent totals per call
        kernel_busy_ms += last[1].get("kernel_busy_ms", 0.0) or last[1]["kernel_ms"]   # union of the kernels' intervals over all streams
        kernel_side_ms += last[1].get("kernel_side_ms", 0.0)
        launches += last[1]["launches"]
        process_ms += last[1]["process_ms"]
        plan_ms += last[1]["plan_ms"]
    sync()
    elapsed = time.time() - t0
    mode_t1 = dev.mode_time() if dev is not None else None
    if world > 1:
        tmax = torch.tensor([elapsed, kernel_ms, float(launches), kernel_busy_ms], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms, launches, kernel_busy_ms = float(tmax[0].item()), float(tmax[1].item()), int(tmax[2].item()), float(tmax[3].item())

    if rank == 0:
        blocks, st = last
        ms_per_step = 1000.0 * elapsed / args.steps
        value = S / (elapsed / args.steps)
        # parity artefact of the timed run: GFF through the product's GenerateOutput
        out_dir = os.path.join(w["dir"], "gpu_out")
        finder.params = params
        finder.GenerateOutput(out_dir, blocks=blocks, blocks_found=int(st["blocks_found"]))
        gff = os.path.join(out_dir, "blocks_coords.gff")
        # algorithmic bytes: one untimed pass of the engine in stats mode; the engine sums the event counters of exactly the
        # Process() calls the reference makes (phase-start result of every seed + re-processing of every conflict)
        ctr_file = os.path.join(w["dir"], "counters.json")
        ctr = None
        n_gpus = args.gpus if in_process else world
        if gpus is not None and not args.no_roofline:
            # the counting pass and the triad run on one device: a plain one beside the set (the counts do not depend on the GPU count)
            dev = sibeliaz_amd.Device(storage, params, 0, **dev_opts)
        if not args.no_roofline:
            # the counts are a property of (input, parameters): the ones of the named workloads are kept in bench_event_counts.json
            # (counted by `bench.py --recount` on the MI355X) so that a default run need not repeat the 3-minute counting pass
            known = json.load(open(os.path.join(ROOT, "bench_event_counts.json"))).get(args.workload) if os.path.exists(os.path.join(ROOT, "bench_event_counts.json")) else None
            counts_src = None
            # (--verify-counts counts them again when other sources than the ones of this build counted them)
            if known and not args.recount and known["lcb_synth"] == w["synth"] and known["seeds"] == S and (known.get("source_hash") == source_hash() or not args.verify_counts):
                ctr = known["event_counts"]
                # the counts are a property of (input, parameters), pinned to the CPU oracle at full size; whether the sources of this build are the
                # ones that last re-counted them on the device is reported, not required
                counts_src = {"file": "bench_event_counts.json", "counted_by": known.get("source", ""), "source_hash_of_that_build": known.get("source_hash"),
                              "source_hash_of_this_build": source_hash(), "same_sources": known.get("source_hash") == source_hash()}
            elif os.path.exists(ctr_file) and not args.recount:
                ctr = json.load(open(ctr_file))
            else:
                t = time.time()
                dev.set_stats_mode(True)
                finder.FindBlocks(w["m"], w["b"], device=dev, seeds=seeds, count_events=1)
                dev.set_stats_mode(False)
                ctr = {k: int(finder.stats["ev_" + k]) for k in sibeliaz_amd.api.COUNTER_NAMES}
                json.dump(ctr, open(ctr_file, "w"))
                if args.write_counts:
                    allc = json.load(open(os.path.join(ROOT, "bench_event_counts.json"))) if os.path.exists(os.path.join(ROOT, "bench_event_counts.json")) else {}
                    allc[args.workload] = {"event_counts": ctr, "lcb_synth": w["synth"], "seeds": S, "workload": w["desc"], "source_hash": source_hash(),
                                           "source": "stats-mode pass of the engine on the MI355X (bench.py --recount --write-counts); reference-semantics counts are a property of input + parameters"}
                    json.dump(allc, open(os.path.join(ROOT, "bench_event_counts.json"), "w"), indent=1, sort_keys=True)
                dev.kernel_time()
                log("bench: stats-mode pass %.1fs: %s" % (time.time() - t, ctr))
        abytes = algorithmic_bytes(ctr) if ctr else 0
        kernel_s_per_step = kernel_busy_ms / 1000.0 / args.steps     # GPU time = the union of the kernels' intervals (never more than the step)
        lps = launches / float(args.steps)
        achieved = abytes / kernel_s_per_step / 1e9 if kernel_s_per_step > 0 else 0.0
        triad = dev.hbm_triad() if dev is not None else 0.0
        # HBM bytes per launch from the PMC counters: collected in separate rocprofv3 --pmc passes of this same command
        # (scripts/gpu_r2_evidence.sh), committed with the profile summaries; only for the workload they were taken on
        traffic, traffic_src, pmc = None, None, None
        pmc_file = next((f for f in (os.path.join(ROOT, "profiles", r, "pmc_traffic.json") for r in ("r06", "r05")) if os.path.exists(f)), None)
        if args.workload == "ecoli62" and n_gpus == 1 and pmc_file:
            pmc = json.load(open(pmc_file))
            traffic = pmc["hbm_bytes_per_launch"]
            traffic_src = {"file": os.path.relpath(pmc_file, ROOT), "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs) of `bench.py --steps 1 --warmup 0` on this "
                           "workload, FETCH_SIZE doubled (gfx950 correction of MI355X_MICROARCH.md), per process-kernel launch",
                           "source_hash_of_that_build": pmc.get("kernel_source_hash"), "source_hash_of_this_build": source_hash(),
                           "same_sources": pmc.get("kernel_source_hash") == source_hash()}
        # per kernel variant: launches and hipEvent-timed kernel time of the timed region (all streams); HBM bytes from the PMC passes where they are of
        # this build's sources (the reference-semantics event counts are per seed, not per variant: the algorithmic figure stays a whole-pass one)
        per_kernel = None
        if mode_t0 is not None and mode_t1 is not None:
            per_kernel = {}
            pk_bytes = (pmc or {}).get("per_kernel_hbm_bytes_per_step") if pmc and pmc.get("kernel_source_hash") == source_hash() else None
            for i, nm in enumerate(("compact", "wide", "big", "huge")):
                ms_k = (mode_t1[0][i] - mode_t0[0][i]) / args.steps
                n_k = (mode_t1[1][i] - mode_t0[1][i]) / float(args.steps)
                if n_k == 0:
                    continue
                b_k = pk_bytes.get(nm) if pk_bytes else None
                per_kernel["lcb_process_kernel<%s>" % nm] = {
                    "calls_per_step": n_k, "ms_per_step": ms_k, "avg_launch_ms": ms_k / n_k, "share_of_kernel_ms": ms_k / max(1e-9, kernel_ms / args.steps),
                    "hbm_bytes_per_step": b_k, "achieved_gb_s": (b_k / (ms_k / 1000.0) / 1e9) if b_k and ms_k > 0 else None,
                    "frac": (b_k / (ms_k / 1000.0) / 1e9 / HBM_PEAK_GBS) if b_k and ms_k > 0 else None}
        line = {
            "metric": "seed vertices/sec through BlocksFinder", "value": value, "unit": "seeds/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": w["desc"], "lcb_synth": w["synth"], "seeds": S, "junction_occurrences": storage.n_positions(),
                       "vertices": storage.GetVerticesNumber(), "phase_size": 256,
                       "parallelism": "one seed per workgroup: compact variant (2 wavefronts; 5 workgroups per CU, or 8 with the small pools an input of few occurrences per vertex gets) for launches of many seeds, wide variant (16 wavefronts "
                                      "sharing the votes) for launches of few, big variant for seeds with thousands of instances; speculative rounds of up to 256 phases with lazy tails and dry-run job launches against predicted used views, exact footprint "
                                      "validation, ordered commit on the host; %d GPU(s)%s" % (n_gpus, ", every launch dealt to the ranks, ncclAllGather of results" if n_gpus > 1 else ", a stop's speculative jobs on side lanes"),
                       "blocks_found": int(st["blocks_found"]), "commit_conflicts": int(st["failures"]), "rounds": int(st["rounds"]),
                       "job_launches": int(st["recompute_launches"]), "jobs": int(st["recomputed_seeds"]), "jobs_used": int(st["jobs_used"]),
                       "views_built": int(st["views_built"]), "over_predicted": int(st["over_predicted"]),
                       "conflict_launches": int(st["conflict_launches"]), "exchanges": int(st["exchanges"]),
                       "side": {k: int(st["side_" + k]) for k in ("batches", "jobs", "taken", "void", "failed")},
                       "early_critical_launches": int(st.get("early_critical", 0)),
                       "lazy_seeds": int(st.get("lazy_seeds", 0)),
                       "host_settled_seeds": int(st.get("host_dead", 0)), "collectives": int(st.get("collectives", 0)),
                       "seeds_per_kernel_variant": dict(zip(("compact", "wide", "big", "huge"), dev.mode_seeds())) if gpus is None else None,
                       "host_ms_per_step": {"in_processor_incl_kernels": process_ms / args.steps, "dry_runs": plan_ms / args.steps,
                                            "commit_validation_other": ms_per_step - (process_ms + plan_ms) / args.steps},
                       "untimed_s": {"load_graph": t_load, "enumerate_seeds": t_seeds, "create_device_upload_tables": t_upload}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "lcb_process_kernel (all variants)", "launches_per_step": lps,
                         "algorithmic_bytes_per_launch": abytes / max(1.0, lps), "avg_launch_ms": kernel_ms / max(1, launches),
                         "algorithmic_bytes_per_step": abytes, "kernel_ms_per_step": kernel_busy_ms / args.steps, "bytes_per_seed": abytes / S,
                         "kernel_ms_sum_over_streams_per_step": kernel_ms / args.steps, "kernel_ms_on_side_lanes_per_step": kernel_side_ms / args.steps,
                         "event_counts_source": counts_src if not args.no_roofline else None,
                         "per_kernel": per_kernel,
                         "peak_measured_stream_triad": triad, "frac_of_measured_peak": achieved / triad if triad > 0 else None,
                         "event_counts": ctr,
                         "note": "latency-bound integer walk: a launch is as long as its longest seed; the event counts of the named workloads are the CPU oracle's at full size "
                                 "(bench_event_counts.json; a stats-mode pass of the device reproduces them, n_compat_call / n_compat_step as upper bounds within 0.02%: "
                                 "speculative results walk older bitmaps). achieved = algorithmic bytes / kernel_ms_per_step, and kernel_ms_per_step is the UNION of the "
                                 "hipEvent-timed intervals of every process-kernel launch on every stream (the time the GPU was busy with them: the side-lane kernels run "
                                 "beside the synchronous ones); kernel_ms_sum_over_streams_per_step is their plain sum - what a rocprofv3 --stats table of the same command sums to"},
        }
        if dev is not None:
            dev.close()
        if gpus is not None:
            gpus.close()
        if n_gpus == 1 and not args.no_cli:
            # metric part 2: wall-clock of the whole drop-in process to blocks_coords.gff
            cli_out = os.path.join(w["dir"], "cli_out")
            t = time.time()
            r = subprocess.run([os.path.join(BIN, "sibeliaz-lcb"), "--graph", w["graph"], w["fasta"], "-k", str(w["k"]), "-b", str(w["b"]), "-m", str(w["m"]),
                                "-a", str(w["a"]), "-t", str(args.threads), "-o", cli_out, "--noseq"], capture_output=True, text=True)
            line["wall_clock"] = {"sibeliaz_lcb_process_s": time.time() - t, "rc": r.returncode, "threads": args.threads,
                                  "gff_md5_equal_to_timed_run": r.returncode == 0 and md5(os.path.join(cli_out, "blocks_coords.gff")) == md5(gff)}
        if n_gpus == 1 and not args.no_cpu_baseline and args.workload in SAMPLES:
            try:
                cb = cpu_baseline_quick(args.workload, args.threads, gff) if args.secondary_leg else cpu_baseline(args.workload, args.threads, not args.sample_cpu_baseline, gff, args.cpu_baseline_budget)
                if cb:
                    line["cpu_baseline"] = cb
            except Exception as e:       # the reference's legs must never cost the line itself
                line["cpu_baseline_error"] = repr(e)
        # the other shapes of BASELINE.json on the scoreboard. The k = 25 shapes (configs 4 / 5 at test size) are timed LIVE by this run, each with its own
        # roofline and a whole-workload reference leg ("measured_in_this_run": true). Entries of the round's evidence run (other workloads: a = 868, the
        # Gbp-scale shapes) are attached from profiles/ only if they were measured on a build of THESE sources, and say "measured_in_this_run": false.
        if args.workload == "ecoli62" and n_gpus == 1 and not args.no_secondary and not args.secondary_leg:
            sec = secondary_live(("primates8_test", "mice16_test"), args.threads, args.no_cpu_baseline)
            sec_file = os.path.join(ROOT, "profiles", "r06", "secondary.json")
            if os.path.exists(sec_file):
                for e in json.load(open(sec_file)):
                    if e.get("sources") == source_hash() and e.get("workload") not in ("primates8_test", "mice16_test"):
                        sec.append(dict(e, measured_in_this_run=False))
            line["secondary"] = sec
        print(json.dumps(line), flush=True)
    if world > 1:
        comm.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
