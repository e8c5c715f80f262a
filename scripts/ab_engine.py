#!/usr/bin/env python3
"""Same-box A/B of engine / device knobs on one workload: the graph is loaded and the seeds are enumerated ONCE, every variant runs
`--passes` passes of BlocksFinder::FindBlocks on the GPU and must produce the blocks of the first variant (the knobs never change
results). Variants that differ in device options get a device of their own.

    python scripts/ab_engine.py --workload ecoli62 base lazyoff:lazy_span=-1 sync:sync_jobs=1 jobs512:max_jobs=512 lanes2:dev.side_lanes=2

A variant is `name[:knob=value[,knob=value...]]`; knobs prefixed with `dev.` are fields of lcb_device_opts, the others of lcb_hooks.
One line per variant: seeds/s, ms per pass (best of the passes), kernel time, launches, stops, jobs, side-lane, early-launch and lazy-tail
counters, host time split."""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_variant(v):
    name, _, rest = v.partition(":")
    dev, eng = {}, {}
    for kv in filter(None, rest.split(",")):
        k, _, val = kv.partition("=")
        (dev if k.startswith("dev.") else eng)[k[4:] if k.startswith("dev.") else k] = int(val)
    return name, dev, eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ecoli62")
    ap.add_argument("--passes", type=int, default=1)
    ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    import bench
    import sibeliaz_amd
    w = bench.ensure_workload(args.workload)
    t = time.time()
    storage = sibeliaz_amd.JunctionStorage(w["graph"], [w["fasta"]], w["k"], threads=args.threads, abundance=w["a"])
    seeds = storage.seeds(args.threads)
    params = sibeliaz_amd.Params.make(w["k"], b=w["b"], m=w["m"])
    print("%s: %d seeds, loaded in %.1f s" % (args.workload, len(seeds), time.time() - t), flush=True)
    finder = sibeliaz_amd.BlocksFinder(storage, w["k"])
    devices = {}
    first = None
    for v in args.variants:
        name, dopt, eopt = parse_variant(v)
        key = tuple(sorted(dopt.items()))
        if key not in devices:
            devices[key] = sibeliaz_amd.Device(storage, params, 0, **dopt)
        dev = devices[key]
        best = None
        for _ in range(args.passes):
            dev.kernel_time()
            t = time.time()
            blocks = finder.FindBlocks(w["m"], w["b"], device=dev, seeds=seeds, threads=args.threads, **eopt)
            dt = time.time() - t
            st = dict(finder.stats)
            if best is None or dt < best[0]:
                best = (dt, st)
            h = hashlib.md5(blocks.tobytes()).hexdigest()
            if first is None:
                first = h
            if h != first:
                print("%s: BLOCKS DIFFER from the first variant" % name, flush=True)
        dt, st = best
        print("%s: %.0f seeds/s, %.1f ms, kernel(sum) %.1f ms, launches %d, stops %d, jobs %d (used %d) | side batches %d jobs %d taken %d void %d failed %d | early %d | "
              "lazy seeds %d host-settled %d | host ms: processor %.0f dry runs %.0f other %.0f" % (
                  name, len(seeds) / dt, 1000 * dt, st["kernel_ms"], st["launches"], st["recompute_launches"], st["recomputed_seeds"], st["jobs_used"],
                  st["side_batches"], st["side_jobs"], st["side_taken"], st["side_void"], st["side_failed"], st.get("early_critical", 0),
                  st.get("lazy_seeds", 0), st.get("host_dead", 0), st["process_ms"], st["plan_ms"], 1000 * dt - st["process_ms"] - st["plan_ms"]), flush=True)
    for d in devices.values():
        d.close()


if __name__ == "__main__":
    main()
