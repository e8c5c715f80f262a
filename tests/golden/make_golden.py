#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/<case>/ from the REAL reference.

Runs only in the build container (needs /root/reference, through oracle/_ref built by
`make -C oracle ref`). For every case it
  1. generates synthetic strains (sibeliaz_amd/bin/lcb-synth) and the TwoPaCo-format junction
     file (sibeliaz_amd/bin/lcb-mkgraph) — `twopaco` and the example genomes are absent from the
     reference checkout;
  2. runs the unmodified reference binary (oracle/_ref/sibeliaz-lcb-ref) at -t 1 and -t 4 and
     checks both give the same blocks_coords.gff (NEWS.md:46);
  3. runs oracle/_ref/ref_dump for the intermediate goldens (sorted seed list, pre-trim block
     instances, per-seed Process() results against the final and the initial `used` state).
What is committed is DATA only: inputs (gzip), expected outputs, and sha256 of the big dumps plus
a line sample for debugging. No reference source is copied.
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BIN = os.path.join(ROOT, "sibeliaz_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")

CASES = {
    # name: (synth args, k, b, m, a)
    "tandem4": ("--strains 4 --segments 8 --seg-min 1000 --seg-max 5000 --seed 7 --tandem 0.3", 15, 200, 50, 150),
    "twogenomes": ("--strains 2 --chromosomes 4 --segments 24 --seg-min 800 --seg-max 3000 --keep 0.9 --sub 0.03 "
                   "--seed 11", 15, 200, 50, 150),
    "nruns_abund": ("--strains 5 --segments 10 --seg-min 600 --seg-max 2500 --repeat-families 3 --repeat-copies 9 "
                    "--repeat-len 300 --nrun 0.5 --sub 0.01 --seed 23", 13, 100, 30, 8),
    "inv_k25": ("--strains 3 --segments 10 --seg-min 1500 --seg-max 5000 --invert 0.4 --swap 0.2 --sub 0.01 "
                "--seed 31", 25, 200, 200, 150),
    "collinear6": ("--strains 6 --segments 4 --seg-min 3000 --seg-max 6000 --keep 1.0 --swap 0 --invert 0 "
                   "--sub 0.015 --indel 0.001 --filler-frac 0 --repeat-families 1 --repeat-copies 3 --seed 41",
                   15, 200, 50, 150),
    "smallb": ("--strains 4 --chromosomes 2 --segments 10 --seg-min 500 --seg-max 2500 --sub 0.04 --indel 0.004 "
               "--tandem 0.2 --seed 53", 11, 40, 60, 150),
}


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        g.write(f.read())


def sample(path, every):
    with open(path) as f:
        lines = f.readlines()
    keep = lines[:50] + lines[50::every]
    return "".join(keep)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "sibeliaz_amd", "build.py"), "tools"])
    only = sys.argv[1:]
    for name, (synth, k, b, m, a) in CASES.items():
        if only and name not in only:
            continue
        out = os.path.join(HERE, name)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        with tempfile.TemporaryDirectory() as tmp:
            fa, gr = os.path.join(tmp, "genomes.fa"), os.path.join(tmp, "graph.bin")
            subprocess.check_call([os.path.join(BIN, "lcb-synth"), "-o", fa] + synth.split())
            subprocess.check_call([os.path.join(BIN, "lcb-mkgraph"), "-k", str(k), "-o", gr, fa])
            gffs = []
            for t in (1, 4):
                od = os.path.join(tmp, "out%d" % t)
                subprocess.check_call([os.path.join(REF, "sibeliaz-lcb-ref"), "--graph", gr, fa, "-k", str(k), "-b", str(b),
                                       "-m", str(m), "-a", str(a), "-t", str(t), "-o", od, "--noseq"],
                                      stdout=subprocess.DEVNULL)
                gffs.append(open(os.path.join(od, "blocks_coords.gff"), "rb").read())
            assert gffs[0] == gffs[1], "reference GFF depends on -t for " + name
            dump = os.path.join(tmp, "dump")
            subprocess.check_call([os.path.join(REF, "ref_dump"), gr, str(k), str(b), str(m), str(a), dump, fa])
            gz(fa, os.path.join(out, "genomes.fa.gz"))
            gz(gr, os.path.join(out, "graph.bin.gz"))
            with open(os.path.join(out, "ref.gff"), "wb") as f:
                f.write(gffs[0])
            for fn in ("pretrim.tsv", "summary.txt", "seeds_final.tsv"):
                shutil.copy(os.path.join(dump, fn), os.path.join(out, fn))
            meta = {"k": k, "b": b, "m": m, "a": a, "synth": synth, "sha256": {}}
            for fn in ("bundles.tsv", "seeds_init.tsv", "seeds_final.tsv", "pretrim.tsv"):
                meta["sha256"][fn] = sha(os.path.join(dump, fn))
            meta["sha256"]["ref.gff"] = hashlib.sha256(gffs[0]).hexdigest()
            meta["sha256"]["genomes.fa"] = sha(fa)
            meta["sha256"]["graph.bin"] = sha(gr)
            with open(os.path.join(out, "bundles.sample.tsv"), "w") as f:
                f.write(sample(os.path.join(dump, "bundles.tsv"), 97))
            with open(os.path.join(out, "seeds_init.sample.tsv"), "w") as f:
                f.write(sample(os.path.join(dump, "seeds_init.tsv"), 97))
            with open(os.path.join(out, "golden.json"), "w") as f:
                json.dump(meta, f, indent=1, sort_keys=True)
            print(name, open(os.path.join(dump, "summary.txt")).read().replace("\n", " "))


if __name__ == "__main__":
    main()
