#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of scripts/r06/gpu_r6_evidence.sh (FETCH_SIZE, WRITE_SIZE, SQ_*: separate runs of one bench pass of config 3) into
per-kernel-instantiation summaries and profiles/r06/pmc_traffic.json (what bench.py quotes as roofline.traffic and roofline.per_kernel[...].hbm_bytes_per_step).
FETCH_SIZE is doubled (gfx950 correction of MI355X_MICROARCH.md: the counter tallies 128-B requests at 64 B; an upper bound for narrow gathers); KB units."""
import collections
import csv
import glob
import json
import os
import re
import sys

O = sys.argv[1]
VARIANT = {"0": "compact", "4": "compact", "1": "wide", "2": "big", "3": "huge"}


def read(d):
    files = glob.glob("%s/%s/**/*counter_collection.csv" % (O, d), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for fn in files:
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"].split("(")[0][-60:]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
    return agg, {k: max(v.values()) for k, v in calls.items()}, files


tot, per, launches = {}, collections.defaultdict(dict), 0
for d, names in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_sq", None)):
    agg, calls, files = read(d)
    if not files:
        print(d, "no counter file")
        continue
    with open("%s/%s_summary.txt" % (O, d), "w") as f:
        for k, v in sorted(agg.items()):
            line = k + "  dispatches=%d  " % calls[k] + "  ".join("%s=%.6g" % kv for kv in sorted(v.items()))
            print(line)
            f.write(line + "\n")
    for k, v in agg.items():
        m = re.search(r"lcb_process_kernel<(\d)", k)
        if not m:
            continue
        var = VARIANT[m.group(1)]
        for n, x in v.items():
            per[var][n] = per[var].get(n, 0.0) + x
        if d == "pmc_fetch":
            per[var]["dispatches"] = per[var].get("dispatches", 0) + calls[k]
    if names:
        tot[names[0]] = sum(v[names[0]] for k, v in agg.items() if "lcb_process_kernel" in k)
        if d == "pmc_fetch":
            launches = sum(c for k, c in calls.items() if "lcb_process_kernel" in k)
    for fn in files:
        os.remove(fn)
if len(tot) == 2:
    b = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
    pk = {var: (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 for var, v in per.items()}
    sq = {var: {n: x for n, x in v.items() if n.startswith("SQ_")} for var, v in per.items()}
    for var, v in sq.items():
        if v.get("SQ_WAVE_CYCLES"):
            insts = v.get("SQ_INSTS_VALU", 0) + v.get("SQ_INSTS_SALU", 0) + v.get("SQ_INSTS_LDS", 0)
            v["derived"] = {"insts_valu_salu_lds_per_wave_quad_cycle": insts / v["SQ_WAVE_CYCLES"],
                            "active_share_of_wave_cycles": v.get("SQ_ACTIVE_INST_ANY", 0) / v["SQ_WAVE_CYCLES"],
                            "issue_stall_share_of_wave_cycles": v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"],
                            "parked_share_of_wave_cycles": v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"]}
    js = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes, scripts/r06/gpu_r6_evidence.sh) over `python bench.py --steps 1 --warmup 0 "
                    "--no-cpu-baseline --no-cli --no-roofline --no-secondary` (config 3 at full size), summed over the lcb_process_kernel instantiations; KB units; FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 correction, an upper bound for narrow gathers); SQ counters in quad-cycles, summed over all wavefronts of the dispatches",
          "kernel_source_hash": open("%s/kernel_source_hash.txt" % O).read().strip(),
          "commit": open("%s/evidence_head.txt" % O).read().strip() if os.path.exists("%s/evidence_head.txt" % O) else None,
          "fetch_kb_raw": tot["FETCH_SIZE"], "write_kb_raw": tot["WRITE_SIZE"], "launches": launches, "hbm_bytes_per_pass": b, "hbm_bytes_per_launch": b / max(1, launches),
          "per_kernel_hbm_bytes_per_step": pk, "per_kernel_write_bytes_per_step": {var: v.get("WRITE_SIZE", 0.0) * 1024.0 for var, v in per.items()}, "per_kernel_sq": sq, "per_kernel_dispatches": {var: v.get("dispatches", 0) for var, v in per.items()}}
    json.dump(js, open("%s/pmc_traffic.json" % O, "w"), indent=1)
    print(json.dumps(js)[:1500])
    os.makedirs("profiles/r06", exist_ok=True)
    json.dump(js, open("profiles/r06/pmc_traffic.json", "w"), indent=1)   # the bench line of the same run quotes it
