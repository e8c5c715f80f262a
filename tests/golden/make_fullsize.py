#!/usr/bin/env python3
"""Generates tests/golden/fullsize.json: hashes of what the UNMODIFIED reference (oracle/_ref/sibeliaz-lcb-ref, built from
/root/reference by oracle/Makefile) writes for the BASELINE.json configurations at their stated size. Run in the build
container only (the reference does not exist on the GPU box); the GPU tests regenerate the inputs with the same
deterministic tools (lcb-synth, lcb-mkgraph) and compare hashes.

    python tests/golden/make_fullsize.py [workdir [case ...]]   # default /tmp/lcb_fullsize; ~25 min of CPU on 8 cores; named cases only:
                                                                # the entries of the others are kept as they are

Only hashes, line counts and the banner figures are committed — no reference source, no genome data.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload definitions are shared with bench.py)

CASES = [
    # name, bench workload (genomes + graph), abundance
    ("config2_ecoli10_a150", "ecoli10", 150),
    ("config3_ecoli62_a150", "ecoli62", 150),
    ("config3_ecoli62_a868", "ecoli62", 868),       # a = 2 * N * D = 2 * 62 * 7 (reference README.md:161-175)
    # configs 4 / 5 (k = 25, many chromosomes, repeat families filtered by a = 150) at the size of a parity test
    ("config4_primates8_test", "primates8_test", 150),
    ("config5_mice16_test", "mice16_test", 150),
    # ... and at the Gbp scale one box generates in about a minute (8 x 24 chromosomes = 1.24 Gbp, 16 x 20 = 1.0 Gbp)
    ("config4_primates8_scaled", "primates8_scaled", 150),
    ("config5_mice16_scaled", "mice16_scaled", 150),
    # ... and config 4's shape at 4.1 Gbp (P = 0.8 G occurrences: ~30 GB in the reference, most of the build container's memory)
    ("config4_primates8_4g_scaled", "primates8_4g", 150),
    # ... and config 5's shape at 4.1 Gbp (round 6)
    ("config5_mice16_4g_scaled", "mice16_4g", 150),
]


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/lcb_fullsize"
    os.environ["LCB_BENCH_DIR"] = work
    ref = os.path.join(ROOT, "oracle", "_ref", "sibeliaz-lcb-ref")
    only = set(sys.argv[2:])
    target = os.path.join(ROOT, "tests", "golden", "fullsize.json")
    target_scaled = os.path.join(ROOT, "tests", "golden", "fullsize_scaled.json")     # the Gbp-scale cases: checked by scripts/check_fullsize_scaled.py, not by the default suite
    out = json.load(open(target)) if only and os.path.exists(target) else {}
    if only and os.path.exists(target_scaled):
        out.update(json.load(open(target_scaled)))
    for name, wl, a in CASES:
        if only and name not in only:
            continue
        w = bench.ensure_workload(wl)
        od = os.path.join(w["dir"], "ref_a%d" % a)
        gff = os.path.join(od, "blocks_coords.gff")
        log = os.path.join(od, "stdout.txt")
        if not (os.path.exists(gff) and os.path.exists(log)):
            os.makedirs(od, exist_ok=True)
            t = time.time()
            r = subprocess.run([ref, "--graph", w["graph"], w["fasta"], "-k", str(w["k"]), "-b", str(w["b"]), "-m", str(w["m"]), "-a", str(a),
                                "-t", str(os.cpu_count() or 1), "-o", od, "--noseq"], capture_output=True, text=True, check=True)
            open(log, "w").write(r.stdout)
            print("%s: reference ran %.0f s" % (name, time.time() - t), flush=True)
        banner = open(log).read()
        out[name] = {
            "workload": wl, "synth": w["synth"], "k": w["k"], "b": w["b"], "m": w["m"], "a": a,
            "fasta_sha256": sha256(w["fasta"]), "graph_sha256": sha256(w["graph"]),
            "gff_sha256": sha256(gff), "gff_lines": sum(1 for _ in open(gff)),
            "blocks_found": int(re.search(r"Blocks found: (\d+)", banner).group(1)),
            "coverage": re.search(r"Coverage: ([0-9.]+)", banner).group(1),
        }
        print(name, out[name]["gff_sha256"], out[name]["blocks_found"], flush=True)
    for path, keep in ((target, lambda k: not k.endswith("_scaled")), (target_scaled, lambda k: k.endswith("_scaled"))):
        part = {k: v for k, v in out.items() if keep(k)}
        if part:
            with open(path, "w") as f:
                json.dump(part, f, indent=1, sort_keys=True)
                f.write("\n")


if __name__ == "__main__":
    main()
