"""bench.py's host-side logic, exercised as CODE on CPU (no GPU, no committed result files): the argument contract
the driver relies on, the self-launch command for --gpus N, the W-trace field against the oracle's definition, the
roofline arithmetic of the JSON line."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_match_the_driver_contract(bench, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0          # no flags: N=1, finishes within minutes
    assert a.batch == 8 and a.resolution == 512 and a.diffusion_steps == 1000 and a.latent == 32   # BASELINE configs[2] shard
    assert a.workload == "real"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (4, 20, 5)


def test_self_launch_builds_a_one_rank_per_gpu_command(bench, monkeypatch):
    """`python bench.py --gpus N` without a launcher must re-execute itself under torch.distributed.run on 127.0.0.1."""
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(2)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_trace_field_is_the_oracles_thin_shell(bench):
    from oracle.gridfiller import analytic_field
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(20000, 3, generator=g) * 2 - 1
    np.testing.assert_allclose(bench.analytic_field_gpu(pts).numpy(), analytic_field(pts).numpy(), rtol=0, atol=1e-7)


def test_roofline_constants(bench):
    # 138 323 585 fp32 parameters, 5 308 416 FLOP per decoder query, HBM 8 TB/s, dense fp16 peak 2.5 PF
    assert bench.UNET_WEIGHT_BYTES == 138_323_585 * 4 - 0 or bench.UNET_WEIGHT_BYTES == 553_294_340
    assert bench.FWD_FLOP == 2 * (63 * 512 + 10 * 512 * 512 + 512)
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.F16_MFMA_PEAK_TF == 2500.0 and bench.FP32_MFMA_PEAK_TF == 157.3
    # per-evaluation roof at B=8, L=32 in f16x2 mode: the weight stream (69 us) dominates 3 x 16.5 GFLOP / 2.5 PF (20 us)
    roof = max(bench.UNET_WEIGHT_BYTES / (bench.HBM_PEAK_GBS * 1e9), 8 * bench.UNET_FLOP_PER_SAMPLE[32] / (bench.F16_MFMA_PEAK_TF / 3 * 1e12))
    assert roof == pytest.approx(69.16e-6, rel=1e-3)


def test_cpu_model_is_reported(bench):
    m = bench.cpu_model()
    assert isinstance(m, str) and len(m) > 0


def test_committed_traffic_is_what_the_tool_derives(tmp_path, bench):
    """roofline.traffic comes from the newest profiles/rNN_pmc_traffic.json; that file must be exactly what
    tools/pmc_to_traffic.py derives from the committed PMC summary (KiB units, FETCH_SIZE doubled on gfx950), not a hand-edited
    number — and bench.py quotes it only while the kernel sources it names are unchanged."""
    import glob
    import hashlib
    import json
    import subprocess
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
    src = newest.replace("_pmc_traffic.json", "_pmc_summary.json")
    out = tmp_path / "traffic.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_to_traffic.py"), os.path.relpath(src, ROOT), str(out)],
                          stdout=subprocess.DEVNULL, cwd=ROOT)
    derived = json.load(open(out))
    on_disk = json.load(open(newest))
    for k, v in derived.items():
        if isinstance(v, float):
            assert on_disk[k] == pytest.approx(v, rel=1e-12), k
    summary = json.load(open(src))
    dec = [v for k, v in summary["fetch"].items() if "decoder_fwd8_kernel" in k][0]
    assert on_disk["decoder_fwd_fetch_bytes_per_launch"] == pytest.approx(dec["FETCH_SIZE"] * 1024 * 2 / dec["dispatches"])
    prof, note = bench.committed_traffic()
    named = on_disk.get("source_sha256")
    fresh = bool(named) and all(hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest() == d for rel, d in named.items())
    if fresh:
        assert prof == on_disk and "unchanged" in note
    else:
        assert prof is None and ("stale" in note or "does not name" in note)


def test_rank_cpu_placement(bench):
    """every rank of an N-rank run keeps an equal contiguous slice of the host's CPUs for its loop threads, meshing
    threads and torch's pool (os.sched_setaffinity in setup_dist); one rank, or a host with fewer than two CPUs per rank, is
    left alone."""
    cpus = list(range(64))
    slices = [bench.rank_cpu_slice(cpus, 8, r) for r in range(8)]
    assert all(len(s) == 8 for s in slices) and sorted(c for s in slices for c in s) == cpus        # disjoint, complete
    assert slices[3] == list(range(24, 32))
    assert bench.rank_cpu_slice(cpus, 1, 0) is None and bench.rank_cpu_slice(list(range(8)), 8, 0) is None
    odd = [0, 1, 2, 3, 8, 9, 10, 11, 12, 13]                                                         # a restricted affinity mask
    assert bench.rank_cpu_slice(odd, 2, 1) == [9, 10, 11, 12, 13]
    assert bench._cpu_ranges(odd) == "0-3,8-13" and bench._cpu_ranges([5]) == "5"
    assert bench.MAX_CLOCK_GHZ == 2.4


def test_latency_form_roofline_and_build_flags(bench):
    """the driver line carries the latency form's fraction of its HBM roof (strict_c3.roofline), the fabric-side
    over-fetch of the wide loops (roofline_loop.traffic_over_algorithmic) and the library's compile-time configuration
    (config.build_flags).  The arithmetic, on round 5's measured 1.36 s per 1000-step request at 8 latents:"""
    r = bench.latency_form_roofline(1.36, 1000, 8, 32, "f16x2")
    assert r["bound"] == "hbm" and r["roof_us_hbm"] == pytest.approx(69.16, rel=1e-3) and r["us_per_evaluation"] == pytest.approx(1360.0)
    assert r["frac"] == pytest.approx(69.16 / 1360.0, rel=1e-3) and r["achieved"] == pytest.approx(0.553294340 / 1.36e-3, rel=1e-6)       # ~407 GB/s of 8 000
    cfg = bench.build_config()
    assert "unsafe_variants=0" in cfg.split() and "C2_GNW=1" in cfg.split() and any(kv.startswith("C2_GNPAD=") for kv in cfg.split())
    src = open(os.path.join(ROOT, "bench.py")).read()
    for field in ('"roofline": latency_form_roofline(', '"traffic_over_algorithmic"', '"build_flags": build_config()'):
        assert field in src, field


def test_committed_driver_line_carries_the_round_6_fields():
    """the newest committed line of the driver's command (profiles/rNN_bench_driver_cmd.json, round 6 on)"""
    import glob
    import json
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_driver_cmd.json")))[-1]
    if int(os.path.basename(newest)[1:3]) < 6:
        pytest.skip("no round-6 driver line committed yet")
    d = json.load(open(newest))
    assert 0 < d["strict_c3"]["roofline"]["frac"] < 1 and d["strict_c3"]["roofline"]["bound"] == "hbm"
    assert "traffic_over_algorithmic" in d["roofline_loop"] and "unsafe_variants=0" in d["config"]["build_flags"]
    # round 6, second half: the strict pass runs the headline's own steps (same shapes) and says where a request's time goes
    s = d["strict_c3"]
    assert s["steps_sampled"] == list(range(d["steps"])) and s["requests"] == d["steps"]
    assert abs(s["decoder_fwd_queries_per_shape"] / s["headline_decoder_fwd_queries_per_shape"] - 1.0) < 0.02
    parts = s["reverse_loop_s_per_request"] + s["decoder_fwd_s_per_request"] + s["decoder_fwd_bwd_s_per_request"] + s["other_s_per_request"]
    assert parts == pytest.approx(s["latency_s_per_request"], rel=1e-6) and 0 <= s["other_s_per_request"] < 0.1 * s["latency_s_per_request"]
    assert d["roofline"]["traffic"] is not None          # the committed PMC profile names the kernel sources this line was measured on


def test_committed_gate_output_is_of_the_committed_kernel_sources():
    """tools/gate.sh prints the sha256 of every kernel source it ran on; the newest committed gate output (profiles/rNN_gate_final.txt)
    must name the sources as they are in the tree — a kernel edit after the last gate shows up here, not at the judge's."""
    import glob
    import hashlib
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gate_final.txt")))[-1]
    text = open(newest).read()
    assert "pytest rc=0" in text and " failed" not in text.split("== determinism_check")[0].split("== pytest")[-1]
    seen = 0
    for line in text.splitlines():
        parts = line.split()
        if len(parts) == 2 and parts[1].startswith("surfd_amd/csrc/") and len(parts[0]) == 16:
            path = os.path.join(ROOT, parts[1])
            assert os.path.exists(path), parts[1]
            assert hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] == parts[0], f"{parts[1]} changed after the committed gate ran"
            seen += 1
    assert seen >= 10
    for n in ("lib=default B=8 wide=0: 1 distinct", "lib=default B=80 wide=80: 1 distinct", "lib=default B=160 wide=160: 1 distinct"):
        assert n in text, n
