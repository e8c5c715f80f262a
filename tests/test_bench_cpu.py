"""bench.py's own plumbing on the CPU (no GPU): the JSON line is assembled from the engine's statistics, the reference's legs are
time-bounded, and a failing leg never costs the line. The device is replaced by a stand-in that reports zeros - what is checked is
the harness, not a measurement (the measurement is the driver's run on the MI355X)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _FakeDevice:
    def __init__(self, storage, params, ordinal, **opts):
        self.opts = opts

    def kernel_time(self):
        return 0.0, 0

    def set_stats_mode(self, on):
        pass

    def hbm_triad(self, nbytes=0, reps=0):
        return 4700.0

    def mode_seeds(self):
        return [0, 0, 0, 0]

    def mode_time(self):
        self.calls = getattr(self, "calls", 0) + 1
        return (4.0 * self.calls, 6.0 * self.calls, 0.0, 0.0), (2 * self.calls, self.calls, 0, 0)

    def close(self):
        pass


def _fake_find_blocks(self, m, b, device=None, seeds=None, comm=None, threads=1, **engine):
    import sibeliaz_amd
    from sibeliaz_amd import api
    self.blocks = np.zeros(0, dtype=sibeliaz_amd.BLOCK_DTYPE)
    self.stats = {f: 0 for f, _ in api.Stats._fields_}
    self.stats.update(kernel_ms=12.5, launches=3, process_ms=20.0, plan_ms=1.0, seeds=len(seeds) if seeds is not None else 0)
    return self.blocks


@pytest.fixture()
def fake_gpu(built, monkeypatch, tmp_path):
    import sibeliaz_amd
    import bench
    monkeypatch.setenv("LCB_BENCH_DIR", str(tmp_path / "wl"))
    monkeypatch.setattr(sibeliaz_amd, "Device", _FakeDevice)
    monkeypatch.setattr(sibeliaz_amd.BlocksFinder, "FindBlocks", _fake_find_blocks)
    return bench


def _run_main(bench, monkeypatch, capsys, argv):
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1, out
    return json.loads(out[0])


def test_bench_line_fields(fake_gpu, monkeypatch, capsys):
    """One JSON line with the contract's fields, the roofline object from the committed-format event counts, the side-lane statistics."""
    bench = fake_gpu
    line = _run_main(bench, monkeypatch, capsys, ["--workload", "ecoli10_tiny", "--steps", "2", "--warmup", "1", "--no-cli", "--no-cpu-baseline", "--no-roofline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["unit"] == "seeds/s" and line["config"]["workload"].startswith("10 synthetic strains")
    assert set(line["config"]["side"]) == {"batches", "jobs", "taken", "void", "failed"}
    assert line["roofline"]["kernel_ms_per_step"] == pytest.approx(12.5) and line["roofline"]["launches_per_step"] == 3
    # per kernel variant: the difference of the device's cumulative (ms, launches) over the timed region, per step; variants that did not run are left out
    pk = line["roofline"]["per_kernel"]
    assert set(pk) == {"lcb_process_kernel<compact>", "lcb_process_kernel<wide>"}
    assert pk["lcb_process_kernel<compact>"]["calls_per_step"] == 1 and pk["lcb_process_kernel<compact>"]["ms_per_step"] == pytest.approx(2.0)
    assert pk["lcb_process_kernel<wide>"]["avg_launch_ms"] == pytest.approx(6.0) and pk["lcb_process_kernel<wide>"]["hbm_bytes_per_step"] is None
    assert "secondary" not in line                      # (only the headline workload carries the other shapes)
    # with the roofline: the counting pass (a workload without committed counts), the triad, the traffic note
    line = _run_main(bench, monkeypatch, capsys, ["--workload", "ecoli10_tiny", "--steps", "1", "--warmup", "0", "--no-cli", "--no-cpu-baseline"])
    assert set(line["roofline"]["event_counts"]) == set(__import__("sibeliaz_amd").api.COUNTER_NAMES) and line["roofline"]["peak_measured_stream_triad"] == 4700.0
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["peak"] == 8000.0 and "frac" in line["roofline"]


def test_reference_legs_are_bounded_and_never_cost_the_line(fake_gpu, monkeypatch, capsys):
    bench = fake_gpu
    w = bench.ensure_workload("ecoli10_tiny")
    # a run that exceeds its limit is killed and reported as such
    assert bench.run_reference(w, 1, "limit", 0.2) == "timeout"
    r = bench.run_reference(w, 2, "ok", 120)
    assert isinstance(r, tuple) and os.path.exists(r[2])
    # the whole protocol on a tiny workload (the GPU leg of the md5 comparison stands in with the reference's own output)
    monkeypatch.setitem(bench.SAMPLES, "ecoli10_tiny", ("ecoli10_tiny", "ecoli10_tiny"))
    monkeypatch.setattr(bench, "our_gff", lambda wl, threads, dev_ordinal=0: r[2])
    cb = bench.cpu_baseline("ecoli10_tiny", 2, True, r[2], budget_s=300.0)
    assert cb["kind"] == "reference" and cb["gff_md5_equal"] is True and cb["value"] > 0 and any(k.endswith("_whole") for k in cb["legs"])
    # a leg that raises does not cost the line
    monkeypatch.setattr(bench, "cpu_baseline", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("boom")))
    line = _run_main(bench, monkeypatch, capsys, ["--workload", "ecoli10_tiny", "--steps", "1", "--warmup", "0", "--no-cli", "--no-roofline"])
    assert "cpu_baseline" not in line and "boom" in line["cpu_baseline_error"]


def test_committed_event_counts_are_the_oracles_at_full_size(built, tmp_path_factory):
    """bench_event_counts.json is the numerator of the roofline (9 N_walk + 15 N_occ + 16 N_compat_call + N_compat_step +
    13 N_inst_out). Its counts are the CPU oracle's from runs at full size; config 2 (1.9 M seeds, half a minute of oracle time) is repeated
    here - counts, conflict count and the hash of the oracle's GFF (= the reference's, tests/golden/fullsize.json). The config-3 line of
    the file was made the same way (27 minutes of one thread: profiles/r03/README.md)."""
    import hashlib
    import subprocess
    import bench
    work = os.environ.get("LCB_TEST_WORKLOADS") or str(tmp_path_factory.getbasetemp().parent / "lcb_model_workloads")
    old = os.environ.get("LCB_BENCH_DIR")
    os.environ["LCB_BENCH_DIR"] = work
    try:
        w = bench.ensure_workload("ecoli10")
    finally:
        if old is None:
            os.environ.pop("LCB_BENCH_DIR", None)
        else:
            os.environ["LCB_BENCH_DIR"] = old
    out = str(tmp_path_factory.mktemp("orc10"))
    r = subprocess.run([os.path.join(ROOT, "oracle", "lcb_oracle"), "--graph", w["graph"], w["fasta"], "-k", str(w["k"]), "-b", str(w["b"]), "-m", str(w["m"]),
                        "-a", str(w["a"]), "-o", out, "--noseq"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    line = [ln for ln in r.stderr.splitlines() if ln.startswith("oracle:")][-1]
    kv = dict(x.split("=") for x in line.split()[1:])
    known = json.load(open(os.path.join(ROOT, "bench_event_counts.json")))["ecoli10"]
    assert known["lcb_synth"] == w["synth"] and known["seeds"] == int(kv["seeds"])
    for name, key in (("n_walk", "walk"), ("n_occ", "occ"), ("n_compat_call", "compat_call"), ("n_compat_step", "compat_step"), ("n_inst_out", "inst_out"),
                      ("n_vote", "vote"), ("n_push", "push"), ("n_process", "process")):
        assert known["event_counts"][name] == int(kv[key]), name
    assert known["failures"] == int(kv["failures"])
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["config2_ecoli10_a150"]
    assert hashlib.sha256(open(os.path.join(out, "blocks_coords.gff"), "rb").read()).hexdigest() == full["gff_sha256"]


def test_ab_driver_runs_every_variant_once_loaded(fake_gpu, monkeypatch, capsys):
    """scripts/ab_engine.py (same-box A/B of engine / device knobs with one load of the workload): variants are parsed into engine and
    device options, a device is shared by the variants with equal device options, one summary line per variant."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ab_engine", os.path.join(ROOT, "scripts", "ab_engine.py"))
    ab = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ab)
    assert ab.parse_variant("x:lazy_span=-1,dev.side_lanes=2,max_jobs=64") == ("x", {"side_lanes": 2}, {"lazy_span": -1, "max_jobs": 64})
    assert ab.parse_variant("base") == ("base", {}, {})
    monkeypatch.setattr(sys, "argv", ["ab_engine.py", "--workload", "ecoli10_tiny", "--threads", "2", "base", "nolazy:lazy_span=-1", "lanes2:dev.side_lanes=2"])
    ab.main()
    out = capsys.readouterr().out.strip().splitlines()
    assert out[0].startswith("ecoli10_tiny:") and [ln.split(":")[0] for ln in out[1:]] == ["base", "nolazy", "lanes2"]
    assert "DIFFER" not in "".join(out)


def test_secondary_entries_are_timed_by_child_runs_and_condensed(fake_gpu, monkeypatch):
    """The k = 25 lines a default run appends: one child `bench.py --secondary-leg` per shape, its line condensed; a child that fails costs an entry, not the line."""
    bench = fake_gpu
    child = {"value": 1.0e6, "unit": "seeds/s", "ms_per_step": 3500.0, "steps": 3, "warmup": 1, "config": {"seeds": 3500465, "blocks_found": 700},
             "roofline": {"bound": "hbm", "achieved": 9.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0011, "kernel_ms_per_step": 3300.0, "algorithmic_bytes_per_step": 3.2e10,
                          "launches_per_step": 1000.0, "per_kernel": {}, "frac_of_measured_peak": 0.002},
             "cpu_baseline": {"value": 4.0e5, "unit": "seeds/s", "cores": 32, "kind": "reference", "gff_md5_equal": True, "sample": "whole"}}
    seen = []

    class R:
        def __init__(self, out):
            self.stdout = out

    def fake_run(cmd, **kw):
        seen.append(cmd)
        if "mice16_test" in cmd:
            return R("no line here\n")
        return R("noise\n" + json.dumps(child) + "\n")
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    sec = bench.secondary_live(("primates8_test", "mice16_test"), 4)
    assert all("--secondary-leg" in c and "--no-cli" in c for c in seen) and len(seen) == 2
    assert sec[0]["measured_in_this_run"] is True and sec[0]["speedup_over_cpu_baseline"] == pytest.approx(2.5) and sec[0]["roofline"]["frac"] == 0.0011
    assert sec[0]["sources"] == bench.source_hash() and sec[0]["cpu_baseline"]["gff_md5_equal"] is True
    assert sec[1]["workload"] == "mice16_test" and "error" in sec[1]


def test_pmc_summary_condenses_the_counter_passes_per_kernel_variant(tmp_path):
    """scripts/r06/pmc_summary.py: the FETCH / WRITE / SQ passes of the evidence script become per-variant HBM bytes (FETCH doubled + WRITE, KB units) and SQ shares in
    profiles/r06/pmc_traffic.json - what bench.py quotes as roofline.traffic and roofline.per_kernel[...].hbm_bytes_per_step. Both compact instantiations count as `compact`."""
    import csv
    out = tmp_path / "ev"
    kernels = [("void lcb_process_kernel<0, false, 2, false, false>(LcbTables, int)", 100.0), ("void lcb_process_kernel<4, false, 2, false, false>(LcbTables, int)", 50.0),
               ("void lcb_process_kernel<1, false, 16, false, false>(LcbTables, int)", 10.0), ("lcb_screen_kernel(LcbTables)", 5.0)]
    for d, names in (("pmc_fetch", {"FETCH_SIZE": 1.0}), ("pmc_write", {"WRITE_SIZE": 0.5}),
                     ("pmc_sq", {"SQ_WAVE_CYCLES": 10.0, "SQ_INSTS_VALU": 2.0, "SQ_INSTS_SALU": 1.5, "SQ_INSTS_LDS": 0.5, "SQ_ACTIVE_INST_ANY": 4.0, "SQ_WAIT_INST_ANY": 1.0, "SQ_WAIT_ANY": 5.0})):
        os.makedirs(out / d / "x")
        with open(out / d / "x" / "p_counter_collection.csv", "w") as f:
            w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writeheader()
            for k, v in kernels:
                for n, scale in names.items():
                    w.writerow({"Kernel_Name": k, "Counter_Name": n, "Counter_Value": v * scale})
    (out / "kernel_source_hash.txt").write_text("abc\n")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "r06", "pmc_summary.py"), str(out)], cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    js = json.load(open(out / "pmc_traffic.json"))
    assert js["kernel_source_hash"] == "abc" and js["launches"] == 3
    assert js["per_kernel_hbm_bytes_per_step"] == {"compact": (2 * 150.0 + 75.0) * 1024, "wide": (2 * 10.0 + 5.0) * 1024}
    assert js["hbm_bytes_per_pass"] == pytest.approx((2 * 160.0 + 80.0) * 1024) and js["per_kernel_dispatches"] == {"compact": 2, "wide": 1}
    assert js["per_kernel_sq"]["compact"]["derived"]["active_share_of_wave_cycles"] == pytest.approx(0.4)
    assert json.load(open(tmp_path / "profiles" / "r06" / "pmc_traffic.json")) == js
