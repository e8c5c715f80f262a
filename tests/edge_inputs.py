"""Small, deterministic edge-case inputs shared by the CPU and GPU edge tests (data, generated at test time)."""
import os
import random
import zlib
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "sibeliaz_amd", "bin")


def _seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def _mut(rng, s, rate):
    return "".join((rng.choice("ACGT") if rng.random() < rate else c) for c in s)


def _wrap(s, w=70):
    return "\n".join(s[i:i + w] for i in range(0, len(s), w))


def build(name, tmp):
    """Returns dict(fasta=[...], graph=..., k, b, m, a) for edge case `name`."""
    rng = random.Random(zlib.crc32(name.encode()))
    k, b, m, a = 15, 200, 50, 150
    files = {}
    if name == "single_strain":                      # no vertex occurs twice -> no seeds at all
        files["g.fa"] = ">only\n" + _wrap(_seq(rng, 6000)) + "\n"
    elif name == "identical_pair":                   # everything collinear
        s = _seq(rng, 5000)
        files["g.fa"] = ">a\n" + _wrap(s) + "\n>b\n" + _wrap(s) + "\n"
    elif name == "tiny_records":                     # records barely longer than k, next to normal ones
        s = _seq(rng, 3000)
        files["g.fa"] = ">a\n" + _wrap(s) + "\n>t1\n" + s[100:120] + "\n>b\n" + _wrap(_mut(rng, s, 0.02)) + "\n>t2\n" + s[500:516] + "\n"
    elif name == "multi_file":                       # the same records spread over several FASTA files (sibeliaz.cpp:112-120)
        s = _seq(rng, 4000)
        files["g1.fa"] = ">a desc text\n" + _wrap(s) + "\n"
        files["g2.fa"] = ">b\tmore\n" + _wrap(_mut(rng, s, 0.03), 61) + "\n>c\n" + _wrap(_mut(rng, s[::-1].translate(str.maketrans("ACGT", "TGCA")), 0.03)) + "\n"
    elif name == "iupac_lowercase":                  # lower case + IUPAC codes + blank lines (streamfastaparser.cpp:60-92)
        s = _seq(rng, 4000)
        t = list(_mut(rng, s, 0.02))
        for i in range(200, 4000, 517):
            t[i] = rng.choice("NRYKMSW")
        files["g.fa"] = ">a\n" + _wrap(s.lower()) + "\n\n>b\n" + _wrap("".join(t)) + "\n  \n>c\n" + _wrap(_mut(rng, s, 0.05)) + "\n"
    elif name == "all_filtered":                     # abundance 2: every shared junction is dropped
        s = _seq(rng, 3000)
        files["g.fa"] = ">a\n" + _wrap(s) + "\n>b\n" + _wrap(_mut(rng, s, 0.02)) + "\n>c\n" + _wrap(_mut(rng, s, 0.02)) + "\n"
        a = 2
    elif name == "small_b_large_m":
        s = _seq(rng, 6000)
        files["g.fa"] = "".join(">s%d\n%s\n" % (i, _wrap(_mut(rng, s, 0.04))) for i in range(5))
        b, m = 12, 300
    elif name == "k25_repeats":
        r = _seq(rng, 400)
        s = _seq(rng, 1500) + r + _seq(rng, 1200) + r + _seq(rng, 900) + r
        files["g.fa"] = "".join(">s%d\n%s\n" % (i, _wrap(_mut(rng, s, 0.01))) for i in range(3))
        k, m = 25, 100
    else:
        raise KeyError(name)
    paths = []
    for fn, txt in files.items():
        p = os.path.join(tmp, name + "_" + fn)
        with open(p, "w") as f:
            f.write(txt)
        paths.append(p)
    graph = os.path.join(tmp, name + ".bin")
    subprocess.check_call([os.path.join(BIN, "lcb-mkgraph"), "-k", str(k), "-o", graph] + paths, stderr=subprocess.DEVNULL)
    return dict(fasta=paths, graph=graph, k=k, b=b, m=m, a=a)


NAMES = ["single_strain", "identical_pair", "tiny_records", "multi_file", "iupac_lowercase", "all_filtered", "small_b_large_m", "k25_repeats"]


def run_cli(exe, case, out, extra=()):
    cmd = [exe, "--graph", case["graph"]] + case["fasta"] + ["-k", str(case["k"]), "-b", str(case["b"]), "-m", str(case["m"]), "-a", str(case["a"]),
                                                               "-t", "2", "-o", out] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True)


def reference_exe():
    ref = os.path.join(ROOT, "oracle", "_ref", "sibeliaz-lcb-ref")
    return ref if os.path.exists(ref) else os.path.join(ROOT, "oracle", "lcb_oracle")
