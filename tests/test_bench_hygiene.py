"""bench.py's bookkeeping (CPU): the HBM-traffic figure of the JSON line is only printed while the committed PMC pass
still describes the kernel the library runs, the kernel names bench.py expects are the ones nrv_api.hip launches, and
the BLAS-grade CPU baseline (oracle/torch_cpu.py) evaluates the same graph as the fp64 oracle."""
import json
import os
import re

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pmc(tmp_path, **over):
    rec = {"T": 13, "batch": 4096, "kernel_name": "void nrv::" + bench.KERNEL_SIGNATURE["f16x2"] + "(nrv::LstmH2Args)",
           "source_sha256_16": bench.kernel_source_sha("f16x2"), "commit": "abc1234", "hbm_bytes_per_launch": 2.8e8,
           "source": "test"}
    rec.update(over)
    (tmp_path / "r99_pmc_lstm3.json").write_text(json.dumps({"f16x2": rec}))
    return str(tmp_path)


def test_traffic_is_printed_only_for_the_kernel_it_was_measured_on(tmp_path):
    v, src, _ = bench.load_traffic(13, 4096, "f16x2", _pmc(tmp_path))
    assert v == 2.8e8 and "r99_pmc_lstm3.json @ abc1234" in src
    v, src, _ = bench.load_traffic(13, 4096, "f16x2", _pmc(tmp_path, source_sha256_16="0" * 16))
    assert v is None and "stale" in src                                   # the kernel's source changed since
    v, src, _ = bench.load_traffic(13, 4096, "f16x2", _pmc(tmp_path, kernel_name="void nrv::lstm_h2o_kernel<32, 16, 128>"))
    assert v is None and "another kernel" in src
    rec = json.loads((tmp_path / "r99_pmc_lstm3.json").read_text())["f16x2"]
    del rec["kernel_name"]
    (tmp_path / "r99_pmc_lstm3.json").write_text(json.dumps({"f16x2": rec}))
    v, src, _ = bench.load_traffic(13, 4096, "f16x2", str(tmp_path))
    assert v is None and "not trusted" in src                             # a file of rounds 1-3: no kernel name
    assert bench.load_traffic(11, 4096, "f16x2", str(tmp_path))[0] is None   # another shape
    # the step-wide figure and the MFMA utilisation travel in the same record
    _, _, rec = bench.load_traffic(13, 4096, "f16x2", _pmc(tmp_path, mfma_busy_frac=0.67, traffic_step=7.6e8,
                                                           step_source_sha256_16=bench.step_source_sha()))
    assert rec["mfma_busy_frac"] == 0.67 and rec["step_source_sha256_16"] == bench.step_source_sha()
    # whatever is committed under profiles/ either matches the current kernel or is refused with a reason
    v, src, _ = bench.load_traffic(13, 4096, "f16x2")
    assert (v is None) == (not src.startswith("profiles/r")) or "@" not in src


def test_helper_child_answers_json_commands_and_never_imports_torch():
    """bench.py's helper (a child of rank 0 started before the first HIP call): one JSON answer per JSON command; the
    host-capacity leg runs INSIDE it, so it must stay free of torch / OpenMP pools."""
    h = bench.Helper()
    try:
        assert "unknown command" in h.ask({"cmd": "nope"}, timeout=60)["error"]
        r = h.ask({"cmd": "hostcap", "cores": 2, "min_s": 0.05}, timeout=300)
        assert "error" not in r, r
        for k in ("workers_1", "workers_2", "cli_1gpu", "cli_8gpu_workers"):
            assert isinstance(r[k], float) and r[k] > 0, (k, r)
        assert r["cli_8gpu_workers_detail"]["worker_processes"] == 8 and r["host_stage"].startswith("native")
    finally:
        h.close()
    assert h.p.wait(20) == 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def helper_main():"):src.index("def cli_e2e(")]
    assert "torch" not in body


def test_kernel_signatures_are_what_the_library_launches():
    api = open(os.path.join(ROOT, "nanoreviser_amd", "csrc", "nrv_api.hip")).read()
    # 192 -> 128 layer, f16x2: launch_lstm_h2w<KQ0, KQ1, H> -> lstm_h2w_kernel<KQ0, KQ1, H, ACT, NRV_L3_WS_NBG>
    m = re.search(r"launch_lstm_h2w<32, 16, 128>\(h, 2,", api)
    assert m, "the 192->128 launch of the f16x2 mode moved: update bench.KERNEL_SIGNATURE"
    nbg = re.search(r"#define NRV_L3_WS_NBG (\d+)", api).group(1)
    assert bench.KERNEL_SIGNATURE["f16x2"] == f"lstm_h2w_kernel<32, 16, 128, 0, {nbg}>"
    for f in sum(bench.KERNEL_SOURCES.values(), []):
        assert os.path.exists(os.path.join(ROOT, "nanoreviser_amd", "csrc", f))


def test_blas_grade_cpu_baseline_is_the_same_graph(species_models):
    from oracle import nrv_oracle as O, torch_cpu as TC
    for sp in ("ecoli", "human"):
        m1, m2 = species_models[sp]
        sig, rd = O.synth_windows(24, 11)
        q1, q2, b1, b2 = O.predict_pair(m1.tensors, m2.tensors, sig, rd, np.float64)
        p1, p2, a1, a2 = TC.predict_pair(TC.TorchCpuModel(m1.tensors), TC.TorchCpuModel(m2.tensors), sig, rd)
        assert np.abs(p1 - q1).max() < 1e-4 and np.abs(p2 - q2).max() < 1e-4
        assert np.array_equal(a1, b1) and np.array_equal(a2, b2)
    m1, m2 = species_models["ecoli"]
    sig, rd = O.synth_windows(64, 11)
    r = bench.blas_grade_baseline(m1, m2, 11, sig, rd, 2, 0.5)
    assert r.get("value", 0) > 0 and r["dtype"] == "f32" and r["cores"] == 2, r


def test_roofline_block_carries_mfma_utilisation_and_hbm_rate(tmp_path, monkeypatch):
    """north_star: "rocprof HBM GB/s and MFMA utilisation reported against gfx950 peak" - the JSON line takes them from the
    committed PMC record of the kernel it times, and prints null (with the reason) when that record is stale."""
    import argparse
    m = {"prof": {"cnn": (0.0, 0), "lstm1": (0.0, 0), "lstm2": (0.0, 0), "lstm3 192->128": (1.45, 10), "lstm4": (0.0, 0), "head": (0.0, 0)},
         "kernel_us": {"lstm3 192->128": 150.0}, "prof_mode": 3, "bracket_overhead_us": 5.0, "ms_per_step": 0.32}
    args = argparse.Namespace()
    rec = {"T": 13, "batch": 4096, "kernel_name": "void nrv::" + bench.KERNEL_SIGNATURE["f16x2"] + "(nrv::LstmH2Args)",
           "source_sha256_16": bench.kernel_source_sha("f16x2"), "commit": "abc", "hbm_bytes_per_launch": 2.88e8, "source": "t",
           "mfma_busy_frac": 0.68, "traffic_step": 7.6e8, "traffic_step_by_kernel": {"lstm3": 2.88e8},
           "step_source_sha256_16": bench.step_source_sha()}
    good = (rec["hbm_bytes_per_launch"], "x", rec)
    monkeypatch.setattr(bench, "load_traffic", lambda T, B, prec, d=None: good)
    r = bench.roofline_blocks(args, 13, 4096, "f16x2", m)["roofline"]
    assert abs(r["avg_launch_us"] - 140.0) < 1e-6                              # 145 us bracketed - 5 us of empty bracket
    assert abs(r["hbm_gbps"] - 2.88e8 / 140e-6 / 1e9) < 1e-6 and abs(r["hbm_frac_of_peak"] - r["hbm_gbps"] / 8000) < 1e-9
    assert r["mfma_busy_frac"] == 0.68 and r["traffic_step"] == 7.6e8 and "12" in r["traffic_step_note"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["frac"] < 1
    stale = dict(rec, step_source_sha256_16="0" * 16)
    monkeypatch.setattr(bench, "load_traffic", lambda T, B, prec, d=None: (rec["hbm_bytes_per_launch"], "x", stale))
    r = bench.roofline_blocks(args, 13, 4096, "f16x2", m)["roofline"]
    assert r["traffic_step"] is None and "another version" in r["traffic_step_note"] and r["hbm_gbps"] is not None
    monkeypatch.setattr(bench, "load_traffic", lambda T, B, prec, d=None: (None, "no committed PMC pass", {}))
    r = bench.roofline_blocks(args, 13, 4096, "f16x2", m)["roofline"]
    assert r["traffic"] is None and r["hbm_gbps"] is None and r["mfma_busy_frac"] is None and r["traffic_step"] is None


def test_scratch_base_prefers_a_tmpfs_and_can_be_overridden(monkeypatch, tmp_path):
    monkeypatch.setenv("NRV_BENCH_SCRATCH", str(tmp_path))
    assert bench.scratch_base() == (str(tmp_path), "NRV_BENCH_SCRATCH")
    monkeypatch.delenv("NRV_BENCH_SCRATCH")
    base, what = bench.scratch_base()
    assert (base == "/dev/shm" and "tmpfs" in what) or (base is None and "default temporary directory" in what)


def test_round6_line_helpers():
    """r06 additions of the bench line that need no GPU: the committed rocprof trace's duration of the dominant kernel (what
    `roofline.frac_rocprof` is computed from) and the device identity that decides whether N ranks ran on N devices."""
    for prec in ("f16x2", "f32"):
        us, src = bench.rocprof_duration_us(prec)
        assert us is not None and 50 < us < 2000 and src.endswith(f"_kernel_stats_timed_region_{prec}.csv"), (prec, us, src)
        fl = bench.flop_lstm3_launch(13, 4096, executed=True)
        frac = fl / (us * 1e-6) / 1e12 / bench.mode_peak(prec)[0]
        assert 0.3 < frac < 1.0, (prec, frac)
    us, why = bench.rocprof_duration_us("f16x2", profiles_dir=os.path.join(ROOT, "tests"))     # no trace there: says so
    assert us is None and "no committed kernel trace" in why
    a = {"pci": "0000:72:00", "uuid": "u1", "device": 0}
    b = {"pci": "0000:73:00", "uuid": "u2", "device": 1}
    assert bench.device_key(a) != bench.device_key(b) and bench.device_key(a) == bench.device_key(dict(a, device=5))
    # a runtime that reports no identity at all must never make N devices look like one
    z0, z1 = {"pci": "0000:00:00", "uuid": "", "device": 0}, {"pci": "0000:00:00", "uuid": "", "device": 1}
    assert bench.device_key(z0) != bench.device_key(z1)
