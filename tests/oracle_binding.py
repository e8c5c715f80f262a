"""ctypes binding of oracle/liblcb_oracle.so — TEST INFRASTRUCTURE ONLY (the checker, never the product)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_INST = np.dtype([("positive", "<i4"), ("chr", "<u4"), ("front_idx", "<u4"), ("back_idx", "<u4")])
ORC_BLOCK = np.dtype([("id", "<i4"), ("_pad", "<i4"), ("chr", "<u8"), ("start", "<u8"), ("end", "<u8")])


class OrcParams(C.Structure):
    _fields_ = [("k", C.c_int64), ("min_block", C.c_int64), ("max_branch", C.c_int64), ("max_flank", C.c_int64), ("looking_depth", C.c_int64)]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_walk", "n_occ", "n_compat_call", "n_compat_step", "n_inst_out", "n_vote", "n_push", "n_process")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OrcStats(C.Structure):
    _fields_ = [("blocks_found", C.c_int64), ("failures", C.c_int64), ("seeds", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "liblcb_oracle.so"))
        vp, i64 = C.c_void_p, C.c_int64
        L.orc_load.restype = vp
        L.orc_load.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, i64, i64, C.c_char_p, C.c_size_t]
        L.orc_free.argtypes = [vp]
        for f in ("orc_n_chr", "orc_n_vertices", "orc_build_bundles"):
            getattr(L, f).restype = i64
            getattr(L, f).argtypes = [vp]
        for f in ("orc_chr_len", "orc_chr_n_pos"):
            getattr(L, f).restype = i64
            getattr(L, f).argtypes = [vp, i64]
        L.orc_chr_used.restype = vp
        L.orc_chr_used.argtypes = [vp, i64]
        L.orc_used_stride.restype = C.c_size_t
        L.orc_reset_used.argtypes = [vp]
        L.orc_get_bundle.argtypes = [vp, i64, C.POINTER(i64), C.POINTER(C.c_int32)] + [C.POINTER(C.c_uint64)] * 4
        L.orc_process_seed.restype = i64
        L.orc_process_seed.argtypes = [vp, C.POINTER(OrcParams), i64, C.c_int32, vp, i64, C.POINTER(i64), C.POINTER(OrcCounters)]
        L.orc_find_blocks.restype = i64
        L.orc_find_blocks.argtypes = [vp, C.POINTER(OrcParams), C.POINTER(vp), C.POINTER(OrcStats), C.POINTER(OrcCounters)]
        L.orc_free_blocks.argtypes = [vp]
        L.orc_generate_output.restype = i64
        L.orc_generate_output.argtypes = [vp, i64, vp, i64, i64, C.c_char_p, C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


class Oracle:
    def __init__(self, graph, fasta, k, abundance):
        self.L = lib()
        arr = (C.c_char_p * len(fasta))(*[f.encode() for f in fasta])
        err = C.create_string_buffer(512)
        self.h = self.L.orc_load(graph.encode(), arr, len(fasta), k, abundance, err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self._buf = np.zeros(1 << 16, dtype=ORC_INST)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_free(self.h)
            self.h = None

    @staticmethod
    def params(k, b, m):
        return OrcParams(k, m, b, b, 8)

    def seeds(self):
        n = self.L.orc_build_bundles(self.h)
        out = []
        vid, ch = C.c_int64(), C.c_int32()
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        for i in range(n):
            self.L.orc_get_bundle(self.h, i, C.byref(vid), C.byref(ch), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
            out.append((vid.value, ch.value, a.value, b.value, c.value, d.value))
        return out

    def process_seed(self, k, b, m, vid, ch, counters=None):
        p = self.params(k, b, m)
        score = C.c_int64()
        n = self.L.orc_process_seed(self.h, C.byref(p), vid, ch, self._buf.ctypes.data, len(self._buf), C.byref(score),
                                    C.byref(counters) if counters is not None else None)
        r = self._buf[:n]
        return [(int(x["chr"]), int(x["front_idx"]), int(x["back_idx"]), int(x["positive"]) != 0) for x in r], score.value

    def find_blocks(self, k, b, m, counters=None):
        p = self.params(k, b, m)
        out, st = C.c_void_p(), OrcStats()
        n = self.L.orc_find_blocks(self.h, C.byref(p), C.byref(out), C.byref(st), C.byref(counters) if counters is not None else None)
        buf = (C.c_char * (n * ORC_BLOCK.itemsize)).from_address(out.value) if n else b""
        blocks = np.frombuffer(buf, dtype=ORC_BLOCK, count=n).copy() if n else np.zeros(0, dtype=ORC_BLOCK)
        self.L.orc_free_blocks(out)
        return blocks, {"blocks_found": st.blocks_found, "failures": st.failures, "seeds": st.seeds}

    def used_bitmap(self, chr_start):
        """Flat `used` bitmap (uint32 words over g) of the oracle's current state."""
        n_pos = int(chr_start[-1])
        bits = np.zeros(n_pos + 64, dtype=np.uint8)
        stride = self.L.orc_used_stride()
        for c in range(self.L.orc_n_chr(self.h)):
            n = self.L.orc_chr_n_pos(self.h, c)
            if n == 0:
                continue
            raw = (C.c_uint8 * (n * stride)).from_address(self.L.orc_chr_used(self.h, c))
            bits[int(chr_start[c]):int(chr_start[c]) + n] = np.frombuffer(raw, dtype=np.uint8)[::stride][:n]
        words = np.packbits(bits[: (n_pos // 32 + 2) * 32].reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype("<u4").ravel()
        return words

    def reset_used(self):
        self.L.orc_reset_used(self.h)
