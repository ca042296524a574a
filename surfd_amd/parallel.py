"""Multi-GPU sampling: shapes are independent (no cross-sample op anywhere in the reverse loop or
the decoder — SURVEY.md §8e), so every rank owns a contiguous block of global shape indices for
BOTH the reverse loop and the grids, with replicated weights and no collective on the data path.
Noise (and conditioning) is seeded per *global* shape index, which makes a shape's result
independent of the world size.  The only collective is an optional all_gather of the final
latents ([B,1,L] floats — a few KB) over RCCL/xGMI (backend "nccl") or gloo (CPU tests).

The reference has no multi-GPU sampling path (utils/dist_util.py:18-41 is a stub).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """(first, count) of the contiguous block owned by `rank`; the remainder goes to the first ranks."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(world, rank, local_rank) from the torchrun environment; initialises the process group
    when world > 1 (backend default: nccl == RCCL when a GPU is present, else gloo)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            dist.init_process_group(backend=backend)
    return world, rank, local


def gather_latents(local: torch.Tensor, counts: List[int]) -> torch.Tensor:
    """all_gather of per-rank latent blocks with (possibly) different lengths -> [sum(counts), ...]
    in global shape order.  Single-process: identity."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    assert len(counts) == world
    m = max(counts)
    # gloo moves host memory: device tensors take a round trip through the host there (CPU tests; a bench run whose ranks
    # share one GPU); RCCL gathers device memory over xGMI
    via_host = local.is_cuda and dist.get_backend() == "gloo"
    work = local.cpu() if via_host else local
    pad = torch.zeros((m,) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
    pad[: work.shape[0]] = work
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    res = torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)
    return res.to(local.device) if via_host else res


def gather_segments(out: torch.Tensor, own: torch.Tensor) -> None:
    """Grid-shard mode's exchange: every rank contributes its compact segment `own` [seg] and receives all of
    them, rank-major, in `out` [world * seg] — ncclAllGather over xGMI with backend "nccl" (all_gather_into_tensor: one
    collective, no zero-filled padding, (world - 1) / world x the buffer received per rank, i.e. half of what summing
    point-indexed buffers with a ring all-reduce moved); gloo moves host memory, device tensors take a round trip through the host
    there (CPU tests; ranks sharing one GPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out.copy_(own)
        return
    world = dist.get_world_size()
    assert out.numel() == world * own.numel(), (out.numel(), world, own.numel())
    if own.is_cuda and dist.get_backend() == "gloo":
        h = own.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h)
        out.copy_(torch.cat(parts))
    elif dist.get_backend() == "gloo":
        parts = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(parts, own)
        out.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(out, own)


def sample_sharded(diffusion, model, total: int, latent_len: int, sampler: str = "ddpm", seed: int = 1234,
                   model_kwargs_fn=None, gather: bool = True, **loop_kwargs):
    """Runs the reverse loop for this rank's block of `total` shapes and (optionally) gathers all
    latents.  ``model_kwargs_fn(first, count)`` builds the per-block conditioning (default: {'y': {}})."""
    from . import synth
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    first, count = shard_range(total, world, rank)
    device = next(model.parameters()).device if hasattr(model, "parameters") else torch.device("cpu")
    T = diffusion.num_timesteps
    noise = synth.synth_noise_batch(T, first, count, latent_len, seed).to(device) if count else None
    lat = torch.zeros(0, 1, latent_len, device=device)
    if count:
        kw = model_kwargs_fn(first, count) if model_kwargs_fn else {"y": {}}
        fn = diffusion.p_sample_loop if sampler == "ddpm" else diffusion.ddim_sample_loop
        lat = fn(model, (count, 1, latent_len), clip_denoised=False, model_kwargs=kw, noise_stream=noise, device=device,
                 **loop_kwargs)
    if gather and world > 1:
        counts = [shard_range(total, world, r)[1] for r in range(world)]
        return gather_latents(lat, counts), (first, count)
    return lat, (first, count)


class ShardedField:
    """Grid-shard mode (one shape evaluated by several ranks — SURVEY.md §8e, the north-star's "shard the
    per-sample grid evaluation"): wraps a per-rank ``udf_func`` so that every call splits its query points
    evenly by index range over the ranks, evaluates the local slice and all_gathers the values
    (ncclAllGather over xGMI with backend "nccl"; <= 4 B x n per level).  Every rank then holds the full
    value vector, derives the same refine mask and issues the same next-level queries, so any grid filler
    that takes a ``udf_func`` (``surfd_amd.meshudf.GridFiller`` through its callback path, or the CPU
    oracle in the tests) produces the single-process grid on every rank.  ``grads`` does the same for
    ``sample_grads`` (12 B x n)."""

    def __init__(self, udf_func, grad_func=None):
        self.udf_func, self.grad_func = udf_func, grad_func

    @staticmethod
    def _world():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
        return 1, 0

    def _scatter_gather(self, fn, pts: torch.Tensor, width: int) -> torch.Tensor:
        import torch.distributed as dist
        world, rank = self._world()
        n = pts.shape[0]
        if world == 1 or n == 0:
            return fn(pts)
        first, count = shard_range(n, world, rank)
        local = fn(pts[first:first + count]) if count else pts.new_zeros((0,) + ((width,) if width else ()))
        counts = [shard_range(n, world, r)[1] for r in range(world)]
        return gather_latents(local.contiguous(), counts)

    def __call__(self, pts: torch.Tensor) -> torch.Tensor:
        return self._scatter_gather(self.udf_func, pts, 0)

    def grads(self, pts: torch.Tensor, max_batch: int = 2 ** 16) -> torch.Tensor:
        """-normalize(grad udf) for all points, evaluated in slices (autograd or the native sweep)."""
        from .meshudf import sample_grads
        fn = self.grad_func or (lambda p: sample_grads(self.udf_func, p, max_batch))
        return self._scatter_gather(fn, pts, 3)


class BatchPipeline:
    """Software pipeline over independent batches on ONE GPU.  One reverse loop is a chain of ~100 000 dependent,
    latency-bound launches that keeps a fraction of the chip busy, so `loop_chains` loops (of different batches)
    run concurrently, each on its own HIP stream and its own execution context (`MDM.replica()`: shared weights,
    private workspace / captured graph), while the grids of finished batches (matrix-pipe bound) are evaluated on
    another stream.  The decoder kernels are persistent (one workgroup per CU), so the chip is split simply by
    their grid size (`CbnDecoder.set_grid_blocks`): while loops of a later round are in flight they take
    `decoder_blocks` CUs, for the last round of batches all of them.  Results are identical to running the batches one after the other
    (tests/test_gpu_unet.py).

        pipe = BatchPipeline(decoder, sample_fn, fill_fn, decoder_blocks=160, loop_chains=2)
        pipe.run(n_batches)

    sample_fn(s, chain) -> latents   enqueues batch s's reverse loop on the current stream with context `chain`
    fill_fn(s, latents) -> None      enqueues batch s's grid evaluation on the current stream
    """

    def __init__(self, decoder, sample_fn, fill_fn, decoder_blocks: int = 128, loop_chains: int = 1):
        self.decoder, self.sample_fn, self.fill_fn = decoder, sample_fn, fill_fn
        self.decoder_blocks = int(decoder_blocks)
        self.loop_chains = max(1, int(loop_chains))
        # chain 0 is created at high stream priority.  Measured: without any high-priority chain the pipeline loses 15 %
        # (3.94 against 4.63 shapes/s: the loops then yield to the decoder's launches); all chains high = chain 0 high.
        # It does not make the first batch of the first round finish early (profiles/r02_pipeline_timeline.txt).
        self.loop_streams = [torch.cuda.Stream(priority=-1 if q == 0 else 0) for q in range(self.loop_chains)]
        self.fill_stream = torch.cuda.Stream()
        self.record_timeline = False        # True: run() leaves per-batch (loop done, grids start, grids done) times in .timeline
        self.timeline = []

    def run(self, n_batches: int) -> None:
        """One host thread per loop chain (a loop call blocks its caller while the device queue is full — the 1000
        graph replays of one loop do not fit it), the calling thread enqueues the grids in batch order."""
        import queue
        import threading
        Q = self.loop_chains
        cur = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        for st in self.loop_streams:
            st.wait_stream(cur)
        self.fill_stream.wait_stream(cur)
        ready = [queue.Queue() for _ in range(Q)]
        errors = []
        marks = []
        t_begin = None
        if self.record_timeline:
            t_begin = torch.cuda.Event(enable_timing=True)
            t_begin.record(cur)

        def chain_worker(q):
            try:
                torch.cuda.set_device(dev)
                for s in range(q, n_batches, Q):
                    with torch.cuda.stream(self.loop_streams[q]):
                        x = self.sample_fn(s, q)
                        x.record_stream(self.fill_stream)
                        ev = torch.cuda.Event(enable_timing=self.record_timeline)
                        ev.record(self.loop_streams[q])
                    ready[q].put((s, x, ev))
            except BaseException as e:          # surfaced on the calling thread
                errors.append(e)
                ready[q].put(None)

        workers = [threading.Thread(target=chain_worker, args=(q,), daemon=True) for q in range(min(Q, n_batches))]
        for w in workers:
            w.start()
        try:
            for f in range(n_batches):
                item = ready[f % Q].get()
                if item is None:
                    raise errors[0]
                s, x, ev = item
                assert s == f
                with torch.cuda.stream(self.fill_stream):
                    self.fill_stream.wait_event(ev)
                    # a loop of a LATER round is running next to these grids (the chains of one round finish together,
                    # so the grids of the last round have the chip to themselves)
                    overlapped = f < Q * ((n_batches - 1) // Q)
                    self.decoder.set_grid_blocks(self.decoder_blocks if overlapped else 0)
                    if self.record_timeline:
                        e0 = torch.cuda.Event(enable_timing=True); e0.record(self.fill_stream)
                    self.fill_fn(f, x)
                    if self.record_timeline:
                        e1 = torch.cuda.Event(enable_timing=True); e1.record(self.fill_stream)
                        marks.append((f, ev, e0, e1))
        finally:
            for w in workers:
                w.join()
            self.decoder.set_grid_blocks(0)
            for st in self.loop_streams:
                cur.wait_stream(st)
            cur.wait_stream(self.fill_stream)
        if errors:
            raise errors[0]
        if self.record_timeline:
            torch.cuda.synchronize()
            self.timeline = [{"batch": f, "loop_done_ms": t_begin.elapsed_time(ev), "grids_start_ms": t_begin.elapsed_time(e0),
                              "grids_done_ms": t_begin.elapsed_time(e1)} for f, ev, e0, e1 in marks]


class PhasedPipeline:
    """Schedule over independent batches on ONE GPU in ROUNDS: the reverse loops of a round of batches as `chains` wide loops
    at once (each over several batches' latents: the denoiser's weights are streamed once per evaluation for all of them;
    conv kernel in its wide form, `MDM.set_wide`), then the grids of the round.

    Time-sliced (overlap_blocks = 0, the default): a round's loops have the whole chip, then its grids have it — the
    decoder kernels (all registers of a CU) cannot co-reside with anything.
    Overlapped (overlap_blocks = D > 0, needs `decoder`): the loops of round r + 1 run NEXT TO the grids of round r, which
    then use D persistent decoder workgroups (the last round's grids all CUs).  Under the decoder kernel the chip is
    power-bound (profiles/r03_decoder_variants.md: 256 CUs sustain 1.70 GHz, 192 CUs 6 % less throughput, not 25 %), so
    the CUs lent to the latency-bound loops cost little; only the first round's loops run alone (`first_round_batches`).

    A shape's result does not depend on the round, the loop or the schedule it rode in (tests/test_gpu_unet.py).

        pipe = PhasedPipeline(loop_fn, fill_fn, chains=2, max_loop_batches=8)
        pipe.run(n_batches)

    loop_fn(first_batch, n_batches, chain) -> latents of batches [first_batch, first_batch + n_batches), batch-major;
                                              enqueues ONE reverse loop on the current stream with execution context `chain`
    fill_fn(batch, latents_of_batch) -> None  enqueues that batch's grid evaluation on the current stream
    loops_fn(parts, streams) -> [latents]     (optional, round 5) enqueues ALL loops of a round from the calling thread:
                                              parts = [(chain, first_batch, n_batches), ...], loop i on streams[i]
                                              (SpacedDiffusion.fused_loops_interleaved: one host thread hands graph replays to the
                                              loops in turn — a fixed submission order).  Without it: one host thread per loop,
                                              each inside one loop_fn call, racing for the driver's queue.
    """

    def __init__(self, loop_fn, fill_fn, chains: int = 2, max_loop_batches: int = 8, overlap_blocks: int = 0, decoder=None,
                 first_round_batches: int = 0, loops_fn=None):
        self.loop_fn, self.fill_fn, self.loops_fn = loop_fn, fill_fn, loops_fn
        self.chains = max(1, int(chains))
        self.max_loop_batches = max(1, int(max_loop_batches))
        self.overlap_blocks = int(overlap_blocks)
        self.decoder = decoder
        if self.overlap_blocks and decoder is None:
            raise ValueError("PhasedPipeline(overlap_blocks=...) needs the decoder whose launches it sizes")
        self.first_round_batches = int(first_round_batches)
        self.loop_streams = None                 # created on first run (plan() needs no device)
        self.record_timeline = False
        self.timeline = []

    def plan(self, n_batches: int):
        """[(first_batch, [(chain, first, count), ...]), ...]: rounds of at most chains * max_loop_batches batches — an
        optional smaller first round (overlapped schedule: its loops run alone), the rest spread evenly over the remaining
        rounds — and a round's batches spread evenly over the chains."""
        per_round = self.chains * self.max_loop_batches
        sizes = []
        rest = n_batches
        if self.first_round_batches and n_batches > self.first_round_batches:
            sizes.append(min(self.first_round_batches, per_round))
            rest -= sizes[0]
        n_rounds = max(1, -(-rest // per_round))
        sizes += [shard_range(rest, n_rounds, r)[1] for r in range(n_rounds)]
        rounds, first = [], 0
        for count in sizes:
            q = min(self.chains, count)
            parts = []
            for c in range(q):
                f, n = shard_range(count, q, c)
                if n:
                    parts.append((c, first + f, n))
            rounds.append((first, parts))
            first += count
        return rounds

    def run(self, n_batches: int) -> None:
        import threading
        if n_batches <= 0:
            return
        if self.loop_streams is None:
            self.loop_streams = [torch.cuda.Stream() for _ in range(self.chains)]
        cur = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        overlap = self.overlap_blocks > 0
        t_begin = None
        if self.record_timeline:
            self.timeline = []
            t_begin = torch.cuda.Event(enable_timing=True)
            t_begin.record(cur)
        marks = []
        for st in self.loop_streams:
            st.wait_stream(cur)                             # whatever the caller enqueued before (noise, weights)

        def start_round(parts):
            """one host thread per loop (a loop call blocks its caller while the device queue is full)"""
            results, errors = {}, []

            def worker(c, f, n):
                try:
                    torch.cuda.set_device(dev)
                    st = self.loop_streams[c]
                    with torch.cuda.stream(st):
                        if not overlap:
                            st.wait_stream(cur)             # time-sliced: the previous round's grids are done before these loops start
                        x = self.loop_fn(f, n, c)
                        x.record_stream(cur)
                        ev = torch.cuda.Event(enable_timing=self.record_timeline)
                        ev.record(st)
                    results[c] = (f, n, x, ev)
                except BaseException as e:                  # surfaced on the calling thread
                    errors.append(e)

            def round_worker():
                try:
                    torch.cuda.set_device(dev)
                    sts = [self.loop_streams[c] for c, _, _ in parts]
                    if not overlap:
                        for st in sts:
                            st.wait_stream(cur)             # time-sliced: the previous round's grids are done before these loops start
                    xs = self.loops_fn(parts, sts)
                    for (c, f, n), st, x in zip(parts, sts, xs):
                        x.record_stream(cur)
                        ev = torch.cuda.Event(enable_timing=self.record_timeline)
                        ev.record(st)
                        results[c] = (f, n, x, ev)
                except BaseException as e:
                    errors.append(e)

            if self.loops_fn is not None:
                threads = [threading.Thread(target=round_worker, daemon=True)]
            else:
                threads = [threading.Thread(target=worker, args=p, daemon=True) for p in parts]
            for t in threads:
                t.start()
            return threads, results, errors

        plan = self.plan(n_batches)
        pending = start_round(plan[0][1])
        try:
            for r, (first, parts) in enumerate(plan):
                threads, results, errors = pending
                for t in threads:
                    t.join()
                pending = None                              # joined; `finally` must only see rounds that are still running
                if errors:
                    errors[0].other_worker_errors = errors[1:]        # every failed loop worker of the round stays reachable
                    raise errors[0]
                nxt = None
                if overlap and r + 1 < len(plan):
                    nxt = pending = start_round(plan[r + 1][1])   # the next round's loops go next to this round's grids
                for c, _, _ in parts:
                    cur.wait_event(results[c][3])
                if overlap:
                    self.decoder.set_grid_blocks(self.overlap_blocks if nxt is not None else 0)
                for c, _, _ in parts:
                    f, n, x, ev = results[c]
                    per = x.shape[0] // n
                    for b in range(n):
                        self.fill_fn(f + b, x[b * per:(b + 1) * per])
                if self.record_timeline:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(cur)
                    marks.append((first, [results[c][3] for c, _, _ in parts], e1, sum(n for _, _, n in parts)))
                if not overlap and r + 1 < len(plan):
                    nxt = pending = start_round(plan[r + 1][1])   # (their streams wait for this round's grids)
                pending = nxt
        finally:
            if pending is not None:
                for t in pending[0]:
                    t.join()
            if overlap:
                self.decoder.set_grid_blocks(0)
            for st in self.loop_streams:
                cur.wait_stream(st)
        if self.record_timeline:
            torch.cuda.synchronize()
            self.timeline = [{"first_batch": f, "batches": nb, "loops_done_ms": max(t_begin.elapsed_time(e) for e in evs),
                              "grids_done_ms": t_begin.elapsed_time(e1)} for f, evs, e1, nb in marks]
