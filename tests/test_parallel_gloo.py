"""world_size-2 (gloo, CPU) test of the multi-rank sampling path: shape-parallel sharding with
per-global-index noise and the final latent all_gather must reproduce the single-process result:
bit for bit with an elementwise (batch-size independent) denoiser, and to fp32 rounding with the
CPU oracle UNet on a reduced configuration (torch's CPU convolutions round differently for
different batch sizes).  The N>1 logic under test is host-side and identical for the HIP model."""
import os
import socket
import types

import torch
from conftest import spawn_bounded
import torch.distributed as dist
import torch.multiprocessing as mp

from surfd_amd.spec import UNetConfig

CFG = UNetConfig(channel_mult=(1,), num_res_blocks=1, attention_resolutions=(1,))
ARGS = types.SimpleNamespace(noise_schedule="cosine", sigma_small=True, clip_value=1.0)
TOTAL, L = 3, 32


class OracleModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        from surfd_amd import synth
        self.sd = synth.synth_unet_state_dict(CFG)
        self.anchor = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x, t, **kw):
        from oracle import unet as ounet
        return ounet.unet_forward(self.sd, x, t)


class ToyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.anchor = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x, t, **kw):
        return torch.tanh(x * 0.7 + 0.3) * (1.0 + t.view(-1, 1, 1).float() / 1000.0)


def _run(total, toy=False):
    from surfd_amd.diffusion import create_gaussian_diffusion
    from surfd_amd.parallel import sample_sharded
    if toy:
        diff = create_gaussian_diffusion(ARGS, "ddim50")
        return sample_sharded(diff, ToyModel(), total, L, sampler="ddpm", seed=7)
    diff = create_gaussian_diffusion(ARGS, "ddim4")
    return sample_sharded(diff, OracleModel(), total, L, sampler="ddim", seed=7)


def _grid_run(N=64):
    """Grid-shard mode on the CPU: the oracle grid filler driven by a ShardedField over the analytic field
    (gradients through torch autograd on each rank's slice)."""
    from oracle import decoder as odec
    from oracle import gridfiller as ogrid
    from surfd_amd.parallel import ShardedField
    field = ShardedField(ogrid.analytic_field, grad_func=lambda p: odec.sample_grads(ogrid.analytic_field, p, 2 ** 16))
    calls = []

    def counting(p):
        calls.append(int(p.shape[0]))
        return field(p)
    # the oracle's fill_grid calls sample_grads(udf_func, ...) itself: give it a callable whose autograd
    # path also goes through the sharded evaluation
    udf, grads, stats = ogrid.fill_grid(counting, N, 2 ** 30, with_grads=False)
    gi = (udf < ogrid.gradient_threshold(N)).nonzero()
    ax = ogrid.axis_coords(N)
    pts = torch.stack([ax[gi[:, 0]], ax[gi[:, 1]], ax[gi[:, 2]]], 1)
    g = field.grads(pts)
    return udf, g, stats


def _tile_exchange(rank, world, n=1000, cap=2048, tile=64):
    """The exchange protocol of the NATIVE grid-shard path (surfd_grid_shard_level_eval / _pack / _level_commit behind
    GridFiller.fill_grid_sharded) on the host: rank r evaluates the 64-point tiles r, r + world, ... of the level's list, compacts
    them into its segment of cap / world points (tile t of the list = tile t // world of rank t % world's segment), the segments
    are all-gathered rank-major (parallel.gather_segments), and every rank reads point e of the level at
    (t % world) * seg + (t // world) * 64 + e % 64.  No zero-fill, no sum: what arrives is the level, bit for bit."""
    from oracle import gridfiller as ogrid
    from surfd_amd.parallel import gather_segments
    assert cap % (tile * world) == 0
    seg = cap // world
    pts = torch.rand(n, 3, generator=torch.Generator().manual_seed(5)) * 2 - 1       # the same "level" on every rank
    own = torch.full((seg,), float("nan"))                                           # entries behind the list are never read
    for t in range(rank, -(-n // tile), world):
        lo, hi = t * tile, min((t + 1) * tile, n)
        own[(t // world) * tile:(t // world) * tile + hi - lo] = ogrid.analytic_field(pts[lo:hi])
    gathered = torch.empty(cap)
    gather_segments(gathered, own)
    e = torch.arange(n)
    t = e // tile
    level = gathered[(t % world) * seg + (t // world) * tile + e % tile]
    return level.clone(), ogrid.analytic_field(pts)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lat, (first, count) = _run(TOTAL)
        toy, _ = _run(TOTAL, toy=True)
        gu, gg, gstats = _grid_run()
        out[rank] = (lat.clone(), first, count, toy.clone(), gu.clone(), gg.clone(), gstats["fwd_per_level"], _tile_exchange(rank, world))
    finally:
        dist.destroy_process_group()


def test_sharded_equals_single_process():
    torch.set_num_threads(4)
    single, _ = _run(TOTAL)
    single_toy, _ = _run(TOTAL, toy=True)
    gu1, gg1, gstats1 = _grid_run()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    spawn_bounded(_worker, (2, port, out), 2)
    assert (out[0][1], out[0][2]) == (0, 2) and (out[1][1], out[1][2]) == (2, 1)
    for r in (0, 1):
        assert out[r][0].shape == (TOTAL, 1, L)
        assert torch.equal(out[r][3], single_toy), "sharded sampling differs from the single-process result"
        torch.testing.assert_close(out[r][0], single, rtol=1e-5, atol=1e-5)
        # grid-shard mode: every rank ends with the single-process grid, bit for bit
        assert torch.equal(out[r][4], gu1) and out[r][6] == gstats1["fwd_per_level"]
        assert torch.equal(out[r][5], gg1)
        # native grid-shard exchange: interleaved tiles -> compact segments -> all-gather == the whole level
        vals, want = out[r][7]
        assert torch.equal(vals, want)
