"""bench.py run as a program on the GPU box: the default line's contract, and the N > 1 code path (launcher, per-rank
shape indices, barriers, per-rank gather, latents all_gather) with two ranks SHARING the one GPU over the gloo
development backend — RCCL itself needs one device per rank, which only the driver's multi-GPU node has."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "0", "--diffusion-steps", "20", "--resolution", "128", "--no-trace", "--no-e2", "--no-cpu-baseline"]


def _line(out: str) -> dict:
    rows = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(rows) == 1, out[-2000:]
    return json.loads(rows[0])


def test_single_gpu_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _line(out.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "shapes/s" and d["value"] > 0 and d["higher_is_better"] and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(8 * 2 / (d["ms_per_step"] * 2 / 1e3), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] < 1 and "traffic_from" in r
    assert d["config"]["schedule"] == "phased" and d["config"]["fp16_range_saturations"] == 0
    assert d["time_share"]["loops_frac"] > 0 and d["config"]["decoder_fwd_queries_per_shape"] >= 32 ** 3


def test_two_ranks_over_gloo_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SURFD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _line(out.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and "not_a_scaling_measurement" in d
    assert d["value"] == pytest.approx(2 * 8 * 2 / (d["ms_per_step"] * 2 / 1e3), rel=1e-6)       # whole-job shapes / max-over-ranks time
    pr = d["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and all(p["ms_per_step"] > 0 and p["decoder_fwd_queries"] > 0 for p in pr)
    assert pr[0]["decoder_fwd_queries"] != pr[1]["decoder_fwd_queries"]                         # the ranks sampled different shapes
    assert max(p["ms_per_step"] for p in pr) <= d["ms_per_step"] * 1.01


@pytest.mark.parametrize("mode", ["shape-parallel", "grid-shard"])
def test_eight_ranks_over_gloo_on_one_gpu(mode):
    """The command line the driver uses for its 8-GPU scaling run, exercised on the one GPU this box has: 8 ranks over the gloo
    development backend sharing the device (64^3, 20 DDIM steps).  What can fail for a reason knowable on one GPU fails here:
    launcher, rank placement, rendezvous, per-rank shape indices, barriers, the gathers of the line, and in grid-shard mode
    the native exchange protocol with 8 ranks (tiles r, r + 8, ..., per-level capacities, overflow counter)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SURFD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # 2 shapes per step and rank: eight processes take turns on one device, and the grid-shard mode meets on the host after every
    # level of every shape (gloo) — with 8 shapes per step that was 5 minutes of context switching inside the full suite
    small = ["--steps", "2", "--warmup", "0", "--diffusion-steps", "20", "--resolution", "64", "--batch", "2", "--no-trace", "--no-e2", "--no-cpu-baseline", "--mode", mode]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8"] + small
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _line(out.stdout)
    assert d["n_gpus"] == 8 and d["value"] > 0
    pr = d["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8)) and all(p["ms_per_step"] > 0 for p in pr)
    if mode == "shape-parallel":
        assert d["scaling"] == "weak" and d["rccl_ranks"] == 8
        assert d["value"] == pytest.approx(8 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3), rel=1e-6)
        firsts = [p["first_shape_index"] for p in pr]
        assert firsts == [r * 2 for r in range(8)]                      # step 0: rank r samples the global shapes 2 r, 2 r + 1
        assert all(p["decoder_fwd_queries"] >= 2 * 2 * 32 ** 3 for p in pr)
        assert all(p["startup_s"] > 0 for p in pr)
        ncpu = len(os.sched_getaffinity(0))
        if ncpu >= 16:                                                   # two or more CPUs per rank: every rank is pinned to its own slice
            cp = [p["cpus"] for p in pr]
            assert all(c for c in cp) and len(set(cp)) == 8
        assert "rank 7: ready after" in out.stderr
    else:
        assert d["scaling"] == "strong" and d["config"]["mode"] == "grid-shard"
        assert d["config"]["exchange_buffers_cut_in_timed_region"] == 0 and d["config"]["exchange_bytes_per_shape"] > 0
        assert len(d["config"]["exchange_capacity_points"]["per_level"]) == 2      # 64^3: the 32^3 lattice and one refinement, each with its own capacity
