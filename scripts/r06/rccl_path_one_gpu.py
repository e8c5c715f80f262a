#!/usr/bin/env python3
"""The multi-rank path on ONE GPU (an RCCL communicator of world 1, lcb_hooks.exchange_always: every dealt launch and every background result goes through
pack / ncclAllGather via the device staging buffers / unpack) against the plain single-rank path, same device, same seeds: what the exchange path costs per pass
and how many exchanges / all-gathers a pass makes (round 6: the stop's own jobs are computed by every rank itself, small exchanges are one collective).

    python scripts/r06/rccl_path_one_gpu.py ecoli62 primates8_test"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench            # noqa: E402
import sibeliaz_amd     # noqa: E402

for name in sys.argv[1:]:
    w = bench.ensure_workload(name)
    storage = sibeliaz_amd.JunctionStorage(w["graph"], [w["fasta"]], w["k"], threads=32, abundance=w["a"])
    seeds = storage.seeds(32)
    params = sibeliaz_amd.Params.make(w["k"], b=w["b"], m=w["m"])
    dev = sibeliaz_amd.Device(storage, params, 0)
    comm = sibeliaz_amd.Comm(dev, sibeliaz_amd.Comm.unique_id(), 0, 1)
    finder = sibeliaz_amd.BlocksFinder(storage, w["k"])
    first = None
    for tag, kw in (("warm-up", {}), ("single rank", {}), ("RCCL path, world 1", dict(comm=comm, exchange_always=1)), ("single rank", {}), ("RCCL path, world 1", dict(comm=comm, exchange_always=1))):
        t = time.time()
        blocks = finder.FindBlocks(w["m"], w["b"], device=dev, seeds=seeds, threads=32, **kw)
        dt = time.time() - t
        h = hashlib.md5(blocks.tobytes()).hexdigest()
        first = first or h
        st = finder.stats
        print("%s: %-20s %8.1f ms per pass | launches %d, stops %d, exchanges %d, all-gathers %d, early launches %d%s" % (
            name, tag, 1000 * dt, st["launches"], st["recompute_launches"], st["exchanges"], st.get("collectives", 0), st.get("early_critical", 0), "" if h == first else "  BLOCKS DIFFER"), flush=True)
    comm.close()
    dev.close()
