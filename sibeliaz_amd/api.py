"""ctypes binding of include/lcb.h. Fails loudly if the library is missing: there is no Python or CPU fallback."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    return os.environ.get("LCB_LIB") or os.path.join(PKG, "libsibeliaz_amd.so")


class LcbError(RuntimeError):
    pass


SEED_DTYPE = np.dtype([("vid", "<i4"), ("ch", "<i4"), ("count", "<u8"), ("rank", "<u8"), ("resolve_pos", "<u8"), ("resolve_chr", "<u8")])
INSTANCE_DTYPE = np.dtype([("chr", "<u4"), ("front_idx", "<u4"), ("back_idx", "<u4"), ("positive", "<u4")])
BLOCK_DTYPE = np.dtype([("id", "<i4"), ("chr", "<u4"), ("start", "<u8"), ("end", "<u8")])
COUNTER_NAMES = ("n_walk", "n_occ", "n_compat_call", "n_compat_step", "n_inst_out", "n_vote", "n_push", "n_process")


class Params(C.Structure):
    _fields_ = [("k", C.c_int32), ("min_block", C.c_int32), ("max_branch", C.c_int32), ("max_flank", C.c_int32),
                ("looking_depth", C.c_int32), ("phase_size", C.c_int32)]

    @classmethod
    def make(cls, k, b=200, m=50):
        """sibeliaz.cpp:133-138: maxFlankingSize = maxBranchSize = b, lookingDepth = 8; blocksfinder.h:519: phase 256."""
        return cls(k, m, b, b, 8, 256)


class Stats(C.Structure):
    _fields_ = [("seeds", C.c_int64), ("blocks_found", C.c_int64), ("failures", C.c_int64), ("launches", C.c_int64),
                ("big_retries", C.c_int64), ("kernel_ms", C.c_double), ("wall_ms", C.c_double), ("rounds", C.c_int64),
                ("recompute_launches", C.c_int64), ("recomputed_seeds", C.c_int64), ("conflict_launches", C.c_int64),
                ("conflict_seeds", C.c_int64), ("exchanges", C.c_int64), ("jobs_used", C.c_int64), ("views_built", C.c_int64),
                ("over_predicted", C.c_int64), ("process_ms", C.c_double), ("plan_ms", C.c_double)] + [("ev_" + n, C.c_uint64) for n in COUNTER_NAMES] + [
                    (n, C.c_int64) for n in ("side_batches", "side_jobs", "side_taken", "side_void", "side_failed", "early_critical")] + [
                    ("kernel_busy_ms", C.c_double), ("kernel_side_ms", C.c_double), ("lazy_seeds", C.c_int64), ("collectives", C.c_int64), ("host_dead", C.c_int64)]


ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
PROCESS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64)
MARK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
RESET_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)


ABI_VERSION = 6        # LCB_ABI_VERSION of include/lcb.h: layout of Stats, Hooks, DeviceOpts (Hooks / DeviceOpts carry it in their first field)


class Hooks(C.Structure):
    _fields_ = [("abi", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("allgather", ALLGATHER_CB), ("allgather_user", C.c_void_p),
                ("process", PROCESS_CB), ("mark", MARK_CB), ("reset", RESET_CB), ("engine_user", C.c_void_p),
                ("round_phases", C.c_int32), ("progress", C.c_int32),
                # engine tuning (0 = default; results never depend on it)
                ("round_fixed", C.c_int32), ("eager_phases", C.c_int32), ("max_views", C.c_int32), ("max_jobs", C.c_int32),
                ("predict_f", C.c_int32), ("exchange_always", C.c_int32), ("count_events", C.c_int32), ("sync_jobs", C.c_int32), ("lazy_span", C.c_int32), ("sparse_rounds", C.c_int32)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.abi = ABI_VERSION


ENGINE_KNOBS = ("round_phases", "round_fixed", "eager_phases", "max_views", "max_jobs", "predict_f", "exchange_always", "count_events", "sync_jobs", "lazy_span", "sparse_rounds")


class DeviceOpts(C.Structure):
    """lcb_device_opts: tuning knobs of a device, 0 = default."""
    _fields_ = [(n, C.c_uint32) for n in ("abi", "compact_slots", "wide_slots", "big_slots", "huge_slots", "path_cap", "wide_path_cap", "max_views", "batch",
                                          "wide_threshold", "start_mode", "screen_min", "path_cap_max", "arena", "side_lanes")] + [
                    # test hooks of the (segment, offset) positions: small segments on small inputs, flat indices beyond 2^32 (lcb.h)
                    ("seg_cap", C.c_uint64), ("side_big_cap", C.c_uint32), ("compact_pools", C.c_uint32), ("seg_gap", C.c_uint64)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.abi = ABI_VERSION


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


REPROCESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64))

_lib = None

EXPORTS = [
    "lcb_last_error", "lcb_version", "lcb_abi_version", "lcb_graph_load", "lcb_graph_free", "lcb_graph_n_chr", "lcb_graph_n_pos", "lcb_graph_n_vertices",
    "lcb_graph_chr_len", "lcb_graph_chr_n_pos", "lcb_graph_chr_name", "lcb_graph_chr_start", "lcb_graph_pos_id", "lcb_graph_pos_pos",
    "lcb_enumerate_seeds", "lcb_free", "lcb_device_create", "lcb_device_create_ex", "lcb_device_mode_seeds", "lcb_device_mode_time", "lcb_device_destroy", "lcb_device_reset_used", "lcb_device_mark_used",
    "lcb_device_set_used", "lcb_device_set_stats_mode", "lcb_device_hbm_triad", "lcb_process_seeds", "lcb_process_seeds_fp", "lcb_device_kernel_time", "lcb_committer_create",
    "lcb_committer_free", "lcb_committer_commit_phase", "lcb_committer_take_marks", "lcb_committer_n_blocks", "lcb_committer_blocks",
    "lcb_committer_blocks_found", "lcb_committer_failures", "lcb_committer_used_words", "lcb_find_blocks", "lcb_find_blocks_ex",
    "lcb_generate_output", "lcb_comm_unique_id", "lcb_comm_create", "lcb_comm_destroy", "lcb_find_blocks_comm", "lcb_find_blocks_gpus", "lcb_gpus_create", "lcb_gpus_find_blocks", "lcb_gpus_destroy",
]


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise LcbError("%s is missing: build it with `python sibeliaz_amd/build.py lib` (no CPU fallback exists)" % path)
    L = C.CDLL(path)
    vp, i64, u64p = C.c_void_p, C.c_int64, C.POINTER(C.c_uint64)
    L.lcb_last_error.restype = C.c_char_p
    L.lcb_version.restype = C.c_char_p
    if not hasattr(L, "lcb_abi_version") or L.lcb_abi_version() != ABI_VERSION:
        raise LcbError("%s has another struct layout version than this binding (LCB_ABI_VERSION %d): rebuild it" % (path, ABI_VERSION))
    L.lcb_graph_load.restype = vp
    L.lcb_graph_load.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int]
    L.lcb_graph_free.argtypes = [vp]
    for f in ("lcb_graph_n_chr", "lcb_graph_n_pos", "lcb_graph_n_vertices"):
        getattr(L, f).restype = i64
        getattr(L, f).argtypes = [vp]
    for f in ("lcb_graph_chr_len", "lcb_graph_chr_n_pos"):
        getattr(L, f).restype = i64
        getattr(L, f).argtypes = [vp, i64]
    L.lcb_graph_chr_name.restype = C.c_char_p
    L.lcb_graph_chr_name.argtypes = [vp, i64]
    L.lcb_graph_chr_start.restype = vp
    L.lcb_graph_chr_start.argtypes = [vp]
    L.lcb_graph_pos_id.restype = vp
    L.lcb_graph_pos_id.argtypes = [vp]
    L.lcb_graph_pos_pos.restype = vp
    L.lcb_graph_pos_pos.argtypes = [vp]
    L.lcb_enumerate_seeds.restype = i64
    L.lcb_enumerate_seeds.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.lcb_free.argtypes = [vp]
    L.lcb_device_create.restype = vp
    L.lcb_device_create.argtypes = [vp, C.POINTER(Params), C.c_int]
    L.lcb_device_create_ex.restype = vp
    L.lcb_device_create_ex.argtypes = [vp, C.POINTER(Params), C.c_int, C.POINTER(DeviceOpts)]
    L.lcb_device_mode_seeds.argtypes = [vp, C.POINTER(i64)]
    if hasattr(L, "lcb_device_mode_time"):      # (an A/B library of an earlier round, LCB_LIB, lacks it)
        L.lcb_device_mode_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64)]
    L.lcb_device_destroy.argtypes = [vp]
    L.lcb_device_reset_used.argtypes = [vp]
    L.lcb_device_mark_used.argtypes = [vp, vp, i64]
    L.lcb_device_set_used.argtypes = [vp, vp, i64]
    L.lcb_device_set_stats_mode.argtypes = [vp, C.c_int]
    L.lcb_process_seeds.argtypes = [vp, vp, i64, vp, vp, C.c_uint64, vp, C.POINTER(Counters)]
    if hasattr(L, "lcb_gpus_create"):
        L.lcb_gpus_create.restype = vp
        L.lcb_gpus_create.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
        L.lcb_gpus_find_blocks.argtypes = [vp, vp, i64, vp, vp, vp, vp]
        L.lcb_gpus_destroy.argtypes = [vp]
    if hasattr(L, "lcb_process_seeds_fp"):      # (absent from older experiment libraries named by LCB_LIB)
        L.lcb_process_seeds_fp.argtypes = [vp, vp, i64, vp, vp, C.c_uint64, vp, vp, C.c_uint64]
    L.lcb_device_hbm_triad.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.lcb_device_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64)]
    L.lcb_committer_create.restype = vp
    L.lcb_committer_create.argtypes = [vp, C.POINTER(Params)]
    L.lcb_committer_free.argtypes = [vp]
    L.lcb_committer_commit_phase.argtypes = [vp, vp, i64, vp, vp, REPROCESS_FN, vp]
    L.lcb_committer_take_marks.restype = i64
    L.lcb_committer_take_marks.argtypes = [vp, vp, i64]
    L.lcb_committer_n_blocks.restype = i64
    L.lcb_committer_n_blocks.argtypes = [vp]
    L.lcb_committer_blocks.restype = vp
    L.lcb_committer_blocks.argtypes = [vp]
    L.lcb_committer_blocks_found.restype = i64
    L.lcb_committer_blocks_found.argtypes = [vp]
    L.lcb_committer_failures.restype = i64
    L.lcb_committer_failures.argtypes = [vp]
    L.lcb_committer_used_words.restype = vp
    L.lcb_committer_used_words.argtypes = [vp, C.POINTER(i64)]
    L.lcb_find_blocks.argtypes = [vp, vp, C.POINTER(Params), vp, i64, C.c_int, C.POINTER(vp), C.POINTER(i64), C.POINTER(Stats)]
    L.lcb_find_blocks_ex.argtypes = [vp, vp, C.POINTER(Params), vp, i64, C.POINTER(Hooks), C.POINTER(vp), C.POINTER(i64), C.POINTER(Stats)]
    L.lcb_comm_unique_id.argtypes = [vp]
    L.lcb_comm_create.restype = vp
    L.lcb_comm_create.argtypes = [vp, vp, C.c_int, C.c_int]
    L.lcb_comm_destroy.argtypes = [vp]
    L.lcb_find_blocks_comm.argtypes = [vp, vp, vp, C.POINTER(Params), vp, i64, C.POINTER(Hooks), C.POINTER(vp), C.POINTER(i64), C.POINTER(Stats)]
    L.lcb_find_blocks_gpus.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.POINTER(Params), C.POINTER(DeviceOpts), vp, i64, C.POINTER(Hooks),
                                       C.POINTER(vp), C.POINTER(i64), C.POINTER(Stats)]
    L.lcb_generate_output.argtypes = [vp, i64, vp, i64, i64, C.c_char_p, C.c_int, i64, C.POINTER(i64), C.POINTER(C.c_double)]
    _lib = L
    return L


def _err(L):
    return LcbError(L.lcb_last_error().decode("utf-8", "replace"))


def _np_from(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class JunctionStorage:
    """Sibelia::JunctionStorage (junctionstorage.h:653): junction file + FASTA -> SoA tables."""

    def __init__(self, graph_file, fasta_files, k, threads=1, abundance=150):
        self.L = load_library()
        if isinstance(fasta_files, str):
            fasta_files = [fasta_files]
        arr = (C.c_char_p * len(fasta_files))(*[f.encode() for f in fasta_files])
        self.h = self.L.lcb_graph_load(graph_file.encode(), arr, len(fasta_files), k, abundance, threads)
        if not self.h:
            raise _err(self.L)
        self.k = k

    def close(self):
        if getattr(self, "h", None):
            self.L.lcb_graph_free(self.h)
            self.h = None

    __del__ = close

    def GetChrNumber(self):
        return self.L.lcb_graph_n_chr(self.h)

    def GetVerticesNumber(self):
        return self.L.lcb_graph_n_vertices(self.h)

    def n_positions(self):
        return self.L.lcb_graph_n_pos(self.h)

    def GetChrDescription(self, c):
        return self.L.lcb_graph_chr_name(self.h, c).decode()

    def chr_len(self, c):
        return self.L.lcb_graph_chr_len(self.h, c)

    def chr_start(self):
        return _np_from(self.L.lcb_graph_chr_start(self.h), self.GetChrNumber() + 1, np.dtype("<u8"))

    def pos_id(self):
        return _np_from(self.L.lcb_graph_pos_id(self.h), self.n_positions(), np.dtype("<i4"))

    def pos_pos(self):
        return _np_from(self.L.lcb_graph_pos_pos(self.h), self.n_positions(), np.dtype("<u4"))

    def seeds(self, threads=1):
        """Bundle enumeration + sort (blocksfinder.h:461-503,517) -> structured array in processing order."""
        out = C.c_void_p()
        n = self.L.lcb_enumerate_seeds(self.h, threads, C.byref(out))
        if n < 0:
            raise _err(self.L)
        arr = _np_from(out.value, n, SEED_DTYPE)
        self.L.lcb_free(out)
        return arr


class Device:
    """One MI355X holding the tables and the `used` bitmap in HBM."""

    def __init__(self, storage, params, ordinal=0, **opts):
        """opts: fields of lcb_device_opts (compact_slots, wide_slots, big_slots, path_cap, wide_path_cap, max_views, batch,
        wide_threshold, start_mode, screen_min)."""
        self.L = load_library()
        self.storage = storage
        self.params = params
        o = DeviceOpts()
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError("unknown device option %r" % k)
            setattr(o, k, int(v))
        self.h = self.L.lcb_device_create_ex(storage.h, C.byref(params), ordinal, C.byref(o))
        if not self.h:
            raise _err(self.L)

    def mode_seeds(self):
        """Seeds handed to the (compact, wide, big, huge) kernel variants since creation."""
        out = (C.c_int64 * 4)()
        if self.L.lcb_device_mode_seeds(self.h, out):
            raise _err(self.L)
        return tuple(int(x) for x in out)

    def mode_time(self):
        """(ms, launches) of the (compact, wide, big, huge) kernel variants since creation: hipEvent-timed, all streams."""
        ms, n = (C.c_double * 4)(), (C.c_int64 * 4)()
        if self.L.lcb_device_mode_time(self.h, ms, n):
            raise _err(self.L)
        return tuple(float(x) for x in ms), tuple(int(x) for x in n)

    def close(self):
        if getattr(self, "h", None):
            self.L.lcb_device_destroy(self.h)
            self.h = None

    __del__ = close

    def reset_used(self):
        if self.L.lcb_device_reset_used(self.h):
            raise _err(self.L)

    def mark_used(self, ranges):
        r = np.ascontiguousarray(ranges, dtype="<u8").reshape(-1, 2)
        if len(r) and self.L.lcb_device_mark_used(self.h, r.ctypes.data, len(r)):
            raise _err(self.L)

    def set_used(self, words):
        w = np.ascontiguousarray(words, dtype="<u4")
        if self.L.lcb_device_set_used(self.h, w.ctypes.data, len(w)):
            raise _err(self.L)

    def set_stats_mode(self, on):
        self.L.lcb_device_set_stats_mode(self.h, 1 if on else 0)

    def process_seeds(self, seeds, counters=False):
        """ProcessVertex::Process (blocksfinder.h:228-310) for a batch -> (offsets[n+1], instances, best_score[n], counters)."""
        s = np.ascontiguousarray(seeds, dtype=SEED_DTYPE)
        n = len(s)
        offsets = np.zeros(n + 1, dtype="<u8")
        score = np.zeros(n, dtype="<i8")
        ctr = Counters()
        cap = max(1024, 64 * n)
        while True:
            inst = np.zeros(cap, dtype=INSTANCE_DTYPE)
            rc = self.L.lcb_process_seeds(self.h, s.ctypes.data, n, offsets.ctypes.data, inst.ctypes.data, cap, score.ctypes.data, C.byref(ctr))
            if rc == 0:
                break
            if int(offsets[n]) > cap:
                cap = int(offsets[n])
                ctr = Counters()
                continue
            raise _err(self.L)
        return offsets, inst[: int(offsets[n])], score, (ctr.as_dict() if counters else None)

    def process_seeds_fp(self, seeds):
        """process_seeds plus every seed's footprint -> (offsets[n+1], instances, fp_offsets[n+1], fp[m, 2] of flat positions (lo, hi))."""
        s = np.ascontiguousarray(seeds, dtype=SEED_DTYPE)
        n = len(s)
        offsets, fp_off = np.zeros(n + 1, dtype="<u8"), np.zeros(n + 1, dtype="<u8")
        cap, fcap = max(1024, 64 * n), max(4096, 256 * n)
        while True:
            inst = np.zeros(cap, dtype=INSTANCE_DTYPE)
            fp = np.zeros((fcap, 2), dtype="<u8")
            rc = self.L.lcb_process_seeds_fp(self.h, s.ctypes.data, n, offsets.ctypes.data, inst.ctypes.data, cap, fp_off.ctypes.data, fp.ctypes.data, fcap)
            if rc == 0:
                break
            if int(offsets[n]) > cap or int(fp_off[n]) > fcap:
                cap, fcap = max(cap, int(offsets[n])), max(fcap, int(fp_off[n]))
                continue
            raise _err(self.L)
        return offsets, inst[: int(offsets[n])], fp_off, fp[: int(fp_off[n])]

    def hbm_triad(self, nbytes=1 << 30, reps=5):
        """Measured HBM rate (STREAM triad, GB/s)."""
        v = C.c_double()
        if self.L.lcb_device_hbm_triad(self.h, nbytes, reps, C.byref(v)):
            raise _err(self.L)
        return v.value

    def kernel_time(self):
        ms, n = C.c_double(), C.c_int64()
        self.L.lcb_device_kernel_time(self.h, C.byref(ms), C.byref(n))
        return ms.value, n.value


class Comm:
    """RCCL communicator of one rank (lcb_comm): the round engine's all-gather runs natively over it (ncclAllGather)."""

    @staticmethod
    def unique_id():
        L = load_library()
        buf = (C.c_ubyte * 128)()
        if L.lcb_comm_unique_id(buf):
            raise _err(L)
        return bytes(buf)

    def __init__(self, device, unique_id, rank, world):
        self.L = load_library()
        self.rank, self.world = rank, world
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self.h = self.L.lcb_comm_create(device.h, buf, rank, world)
        if not self.h:
            raise _err(self.L)

    def close(self):
        if getattr(self, "h", None):
            self.L.lcb_comm_destroy(self.h)
            self.h = None

    __del__ = close


class Committer:
    """Ordered commit of per-seed results (blocksfinder.h:312-332,372-427). Host-only."""

    def __init__(self, storage, params):
        self.L = load_library()
        self.storage = storage
        self.h = self.L.lcb_committer_create(storage.h, C.byref(params))
        if not self.h:
            raise _err(self.L)

    def close(self):
        if getattr(self, "h", None):
            self.L.lcb_committer_free(self.h)
            self.h = None

    __del__ = close

    def commit_phase(self, seeds, offsets, inst, reprocess):
        """reprocess(seed_record) -> structured INSTANCE array computed against the live state."""
        s = np.ascontiguousarray(seeds, dtype=SEED_DTYPE)
        off = np.ascontiguousarray(offsets, dtype="<u8")
        ins = np.ascontiguousarray(inst, dtype=INSTANCE_DTYPE)
        err = []

        def cb(_user, seed_ptr, out_ptr, cap, n_out):
            try:
                seed = _np_from(seed_ptr, 1, SEED_DTYPE)
                res = np.ascontiguousarray(reprocess(seed), dtype=INSTANCE_DTYPE)
                n_out[0] = len(res)
                if len(res) <= cap and len(res):
                    C.memmove(out_ptr, res.ctypes.data, res.nbytes)
                return 0
            except Exception as e:  # noqa: BLE001 - reported through the C return code
                err.append(e)
                return -1

        fn = REPROCESS_FN(cb)
        rc = self.L.lcb_committer_commit_phase(self.h, s.ctypes.data, len(s), off.ctypes.data, ins.ctypes.data if len(ins) else None, fn, None)
        if err:
            raise err[0]
        if rc:
            raise _err(self.L)

    def take_marks(self):
        out = []
        buf = np.zeros((4096, 2), dtype="<u8")
        while True:
            n = self.L.lcb_committer_take_marks(self.h, buf.ctypes.data, len(buf))
            out.append(buf[:n].copy())
            if n < len(buf):
                break
        return np.concatenate(out) if out else np.zeros((0, 2), dtype="<u8")

    def blocks(self):
        return _np_from(self.L.lcb_committer_blocks(self.h), self.L.lcb_committer_n_blocks(self.h), BLOCK_DTYPE)

    def blocks_found(self):
        return self.L.lcb_committer_blocks_found(self.h)

    def failures(self):
        return self.L.lcb_committer_failures(self.h)

    def used_words(self):
        n = C.c_int64()
        p = self.L.lcb_committer_used_words(self.h, C.byref(n))
        return _np_from(p, n.value, np.dtype("<u4"))


class GpuSet:
    """lcb_gpus: several GPUs of this node driven from one process - devices created, tables uploaded and RCCL initialised once."""

    def __init__(self, storage, params, ordinals, always_comm=False, **device_opts):
        self.L = load_library()
        self.params = params
        o = DeviceOpts()
        for k, v in device_opts.items():
            setattr(o, k, int(v))
        ords = (C.c_int * len(ordinals))(*ordinals)
        self.h = self.L.lcb_gpus_create(storage.h, ords, len(ordinals), C.byref(params), C.byref(o), 1 if always_comm else 0)
        if not self.h:
            raise _err(self.L)

    def close(self):
        if getattr(self, "h", None):
            self.L.lcb_gpus_destroy(self.h)
            self.h = None

    __del__ = close


class BlocksFinder:
    """Sibelia::BlocksFinder (blocksfinder.h:178): FindBlocks on one GPU, then GenerateOutput."""

    def __init__(self, storage, k):
        self.L = load_library()
        self.storage = storage
        self.k = k
        self.blocks = None
        self.stats = None
        self.params = None

    def FindBlocksOnSet(self, gpus, seeds=None, threads=1, **engine):
        """One pass on a persistent GpuSet (tables resident, RCCL initialised): what bench.py --gpus N times."""
        self.params = gpus.params
        hooks = Hooks()
        hooks.world = 1
        for k, v in engine.items():
            if k not in ENGINE_KNOBS:
                raise TypeError("unknown engine knob %r" % k)
            setattr(hooks, k, int(v))
        s = self.storage.seeds(threads) if seeds is None else np.ascontiguousarray(seeds, dtype=SEED_DTYPE)
        out, n, st = C.c_void_p(), C.c_int64(), Stats()
        if self.L.lcb_gpus_find_blocks(gpus.h, s.ctypes.data, len(s), C.byref(hooks), C.byref(out), C.byref(n), C.byref(st)):
            raise _err(self.L)
        self.blocks = _np_from(out.value, n.value, BLOCK_DTYPE)
        self.L.lcb_free(out)
        self.stats = {f: getattr(st, f) for f, _ in Stats._fields_}
        return self.blocks

    def FindBlocksGpus(self, minBlockSize, maxBranchSize, ordinals, seeds=None, threads=1, device_opts=None, **engine):
        """FindBlocks on several GPUs from this process (one host thread per GPU, RCCL all-gather between them)."""
        p = Params(self.k, minBlockSize, maxBranchSize, maxBranchSize, 8, 256)
        self.params = p
        hooks = Hooks()
        hooks.world = 1
        for k, v in engine.items():
            if k not in ENGINE_KNOBS:
                raise TypeError("unknown engine knob %r" % k)
            setattr(hooks, k, int(v))
        o = DeviceOpts()
        for k, v in (device_opts or {}).items():
            setattr(o, k, int(v))
        s = self.storage.seeds(threads) if seeds is None else np.ascontiguousarray(seeds, dtype=SEED_DTYPE)
        ords = (C.c_int * len(ordinals))(*ordinals)
        out, n, st = C.c_void_p(), C.c_int64(), Stats()
        if self.L.lcb_find_blocks_gpus(self.storage.h, ords, len(ordinals), C.byref(p), C.byref(o), s.ctypes.data, len(s), C.byref(hooks),
                                       C.byref(out), C.byref(n), C.byref(st)):
            raise _err(self.L)
        self.blocks = _np_from(out.value, n.value, BLOCK_DTYPE)
        self.L.lcb_free(out)
        self.stats = {f: getattr(st, f) for f, _ in Stats._fields_}
        return self.blocks

    def FindBlocks(self, minBlockSize, maxBranchSize, maxFlankingSize=None, lookingDepth=8, sampleSize=0, threads=1, device=None,
                   seeds=None, hooks=None, comm=None, **engine):
        """hooks: an api.Hooks (multi-rank all-gather and/or callback engine); device may be None only with callback hooks.
        engine: tuning knobs of the round engine (ENGINE_KNOBS), e.g. round_phases=7, round_fixed=1, max_views=-1."""
        if engine:
            if hooks is None:
                hooks = Hooks()
                hooks.world = 1
            for k, v in engine.items():
                if k not in ENGINE_KNOBS:
                    raise TypeError("unknown engine knob %r" % k)
                setattr(hooks, k, int(v))
        p = Params(self.k, minBlockSize, maxBranchSize, maxBranchSize if maxFlankingSize is None else maxFlankingSize, lookingDepth, 256)
        self.params = p
        callback_engine = hooks is not None and bool(hooks.process)
        own = device is None and not callback_engine
        dev = Device(self.storage, p) if own else device
        try:
            s = self.storage.seeds(threads) if seeds is None else np.ascontiguousarray(seeds, dtype=SEED_DTYPE)
            out, n, st = C.c_void_p(), C.c_int64(), Stats()
            if comm is not None:     # native multi-rank path: RCCL all-gather inside the C++ engine
                rc = self.L.lcb_find_blocks_comm(self.storage.h, dev.h, comm.h, C.byref(p), s.ctypes.data, len(s),
                                                 C.byref(hooks) if hooks is not None else None, C.byref(out), C.byref(n), C.byref(st))
            else:
                rc = self.L.lcb_find_blocks_ex(self.storage.h, dev.h if dev is not None else None, C.byref(p), s.ctypes.data, len(s),
                                               C.byref(hooks) if hooks is not None else None, C.byref(out), C.byref(n), C.byref(st))
            if rc:
                raise _err(self.L)
            self.blocks = _np_from(out.value, n.value, BLOCK_DTYPE)
            self.L.lcb_free(out)
            self.stats = {f: getattr(st, f) for f, _ in Stats._fields_}
        finally:
            if own:
                dev.close()
        return self.blocks

    def GenerateOutput(self, outDir, genSeq=False, chunks=0, blocks=None, blocks_found=None):
        b = np.ascontiguousarray(self.blocks if blocks is None else blocks, dtype=BLOCK_DTYPE)
        found = int(np.abs(b["id"]).max()) if blocks_found is None and len(b) else (blocks_found or 0)
        if blocks_found is None and self.stats is not None and blocks is None:
            found = self.stats["blocks_found"]
        nt, cov = C.c_int64(), C.c_double()
        rc = self.L.lcb_generate_output(self.storage.h, self.params.min_block if self.params else 0, b.ctypes.data if len(b) else None, len(b),
                                        found, outDir.encode(), 1 if genSeq else 0, chunks, C.byref(nt), C.byref(cov))
        if rc:
            raise _err(self.L)
        return nt.value, cov.value
