"""lcb-mkgraph (the build-owned stand-in for the absent `twopaco`) against an independent brute-force
junction finder: the dict-of-canonical-k-mers definition of SURVEY.md Appendix A, written here in plain Python and sharing
no code with csrc/tools/mkgraph.cpp. The file format is the one the reference reads (common/junctionapi.h:80-98):
packed little-endian {uint32 pos, int64 id}, separator record {0xFFFFFFFF, INT64_MAX} after every sequence."""
import gzip
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def read_fasta(path):
    recs = []
    for line in open(path):
        if line.startswith(">"):
            recs.append([line[1:].split()[0], []])
        elif recs:
            recs[-1][1].append("".join(line.split()).upper())
    return [(n, "".join(s)) for n, s in recs]


def brute_force_junctions(recs, k):
    """-> per sequence, the list of (pos, signed id)."""
    succ, pred, forced = {}, {}, set()

    def windows(seq):
        for p in range(len(seq) - k + 1):
            w = seq[p:p + k]
            if all(c in COMP for c in w):
                yield p, w

    def canon(w):
        rc = "".join(COMP[c] for c in reversed(w))
        return (w, True) if w < rc else (rc, False)

    for _, seq in recs:
        for p, w in windows(seq):
            c, fwd = canon(w)
            nx = seq[p + k] if p + k < len(seq) and seq[p + k] in COMP else None
            pv = seq[p - 1] if p > 0 and seq[p - 1] in COMP else None
            if nx is None or pv is None:                      # sequence end or a neighbour that is not ACGT
                forced.add(c)
            if fwd:
                if nx: succ.setdefault(c, set()).add(nx)
                if pv: pred.setdefault(c, set()).add(pv)
            else:                                             # read on the other strand: roles swap, characters complement
                if pv: succ.setdefault(c, set()).add(COMP[pv])
                if nx: pred.setdefault(c, set()).add(COMP[nx])
    ids, out = {}, []
    for _, seq in recs:
        rows = []
        for p, w in windows(seq):
            c, fwd = canon(w)
            if c in forced or len(succ.get(c, ())) >= 2 or len(pred.get(c, ())) >= 2:
                i = ids.setdefault(c, len(ids) + 1)           # ids in order of first appearance among junction occurrences
                rows.append((p, i if fwd else -i))
        out.append(rows)
    return out


def read_junction_file(path):
    data = open(path, "rb").read()
    assert len(data) % 12 == 0
    seqs, cur = [], []
    for o in range(0, len(data), 12):
        pos, vid = struct.unpack_from("<Iq", data, o)
        if pos == 0xFFFFFFFF or vid == 0x7FFFFFFFFFFFFFFF:
            seqs.append(cur)
            cur = []
        else:
            cur.append((pos, vid))
    assert not cur, "file does not end with a separator"
    return seqs


@pytest.mark.parametrize("k", [None, 11])
def test_mkgraph_matches_brute_force(built, case, k, tmp_path):
    k = k or case.k
    out = str(tmp_path / "g.bin")
    subprocess.check_call([os.path.join(ROOT, "sibeliaz_amd", "bin", "lcb-mkgraph"), "-k", str(k), "-o", out, case.fasta], stderr=subprocess.DEVNULL)
    got = read_junction_file(out)
    want = brute_force_junctions(read_fasta(case.fasta), k)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w
    if k == case.k:       # the committed golden graph is what this tool wrote when the reference produced the goldens
        with gzip.open(os.path.join(case.dir, "graph.bin.gz"), "rb") as f:
            assert f.read() == open(out, "rb").read()


def test_mkgraph_low_redundancy_many_threads(built, tmp_path):
    """A random (low-redundancy) sequence fills the first k-mer table of the parallel build: every thread has to notice,
    stop inserting and start over with a larger table instead of probing a full table forever; same file for any thread count."""
    import hashlib
    import random

    rnd = random.Random(5)
    fa = tmp_path / "rand.fa"
    fa.write_text(">r\n" + "".join(rnd.choice("ACGT") for _ in range(3_000_000)) + "\n")
    digests = set()
    for threads in (1, 4):
        out = str(tmp_path / ("g%d.bin" % threads))
        subprocess.run([os.path.join(ROOT, "sibeliaz_amd", "bin", "lcb-mkgraph"), "-k", "25", "-o", out, str(fa)], check=True, timeout=120,
                       stderr=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
        digests.add(hashlib.md5(open(out, "rb").read()).hexdigest())
    assert len(digests) == 1
