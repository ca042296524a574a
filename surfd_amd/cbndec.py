"""Drop-ins for the reference's field modules, executing on libsurfd_hip.so.

  CoordsEncoder   <- AutoEncoder/models/coordsenc.py:7-51   (same ctor, .out_dim, .encode)
  CbnDecoder      <- AutoEncoder/models/cbndec.py:106-134    (same ctor, state_dict keys, forward)
  make_udf_func   <- the closure every sample/generate_*.py builds (generate_uncond.py:96-101)

``CbnDecoder`` is an ``nn.Module`` whose parameters/buffers carry exactly the reference's
state_dict keys, so ``decoder.load_state_dict(ckpt["decoder"], strict=True)`` and
``.cuda().eval()`` work unchanged.  ``forward`` never computes in torch: it hands device
pointers to the fused HIP kernel.  The reference's closure
``decoder(coords_encoder.encode(c.unsqueeze(0)), lat)`` stays valid *and* fused because
``encode`` returns a lazy ``EncodedCoords`` that remembers the raw coordinates; autograd
through it (``sample_grads``' ``.backward()``) is served by the in-kernel reverse sweep.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import torch
from torch import Tensor, nn

from . import _native as N
from .spec import DecoderConfig, decoder_param_spec

UDF_MAX_DIST = 0.1


class EncodedCoords:
    """Lazy result of ``CoordsEncoder.encode``: keeps the raw xyz so the decoder can fuse the
    positional encoding; materialises the [..., 63] tensor only if someone really asks."""

    def __init__(self, raw: Tensor, enc: "CoordsEncoder"):
        self.raw, self._enc, self._t = raw, enc, None

    @property
    def shape(self):
        return tuple(self.raw.shape[:-1]) + (self._enc.out_dim,)

    def materialize(self) -> Tensor:
        if self._t is None:
            self._t = self._enc.encode_dense(self.raw)
        return self._t

    def __getattr__(self, name):           # anything tensor-like falls through to the dense tensor
        return getattr(self.materialize(), name)


class CoordsEncoder:
    def __init__(self, input_dims: int = 3, include_input: bool = True, max_freq_log2: int = 9,
                 num_freqs: int = 10, log_sampling: bool = True, periodic_fns=(torch.sin, torch.cos)) -> None:
        self.input_dims, self.include_input = input_dims, include_input
        self.max_freq_log2, self.num_freqs, self.log_sampling = max_freq_log2, num_freqs, log_sampling
        self.periodic_fns = periodic_fns
        self.out_dim = (input_dims if include_input else 0) + input_dims * num_freqs * len(periodic_fns)
        self._fusable = (input_dims == 3 and include_input and max_freq_log2 == 9 and num_freqs == 10
                         and log_sampling and tuple(periodic_fns) == (torch.sin, torch.cos))

    def encode_dense(self, inputs: Tensor) -> Tensor:
        """The explicit [..., out_dim] tensor (plain torch; only used when a caller needs the
        encoding itself — the decoder path never does)."""
        if self.log_sampling:
            freqs = 2.0 ** torch.linspace(0.0, self.max_freq_log2, steps=self.num_freqs)
        else:
            freqs = torch.linspace(2.0 ** 0.0, 2.0 ** self.max_freq_log2, steps=self.num_freqs)
        parts = [inputs] if self.include_input else []
        for f in freqs:
            for fn in self.periodic_fns:
                parts.append(fn(inputs * f))
        return torch.cat(parts, -1)

    def encode(self, inputs: Tensor):
        if self._fusable and inputs.is_cuda:
            return EncodedCoords(inputs, self)
        return self.encode_dense(inputs)


def _register(module: nn.Module, key: str, tensor: Tensor, is_param: bool) -> None:
    """Creates the nested module path of `key` and registers the leaf under it."""
    parts = key.split(".")
    m = module
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    if is_param:
        m.register_parameter(parts[-1], nn.Parameter(tensor))
    else:
        m.register_buffer(parts[-1], tensor)


class _DecoderFn(torch.autograd.Function):
    """logits = decoder(xyz) with d logits / d xyz from the kernel's reverse sweep."""

    @staticmethod
    def forward(ctx, pts: Tensor, dec: "CbnDecoder", sample: int):
        ctx.dec, ctx.sample = dec, sample
        ctx.save_for_backward(pts)
        return dec._logits_xyz(pts.detach(), sample)

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        (pts,) = ctx.saved_tensors
        dl = ctx.dec._dlogit_xyz(pts.detach(), ctx.sample)
        return grad_out.reshape(-1, 1) * dl, None, None


class CbnDecoder(nn.Module):
    def __init__(self, input_dim: int, latent_dim: int, hidden_dim: int, num_hidden_layers: int,
                 out_dim: int = 1, refine: bool = False) -> None:
        super().__init__()
        if out_dim != 1:
            raise NotImplementedError("libsurfd_hip implements the out_dim=1 UDF decoder")
        self.cfg = DecoderConfig(input_dim, latent_dim, hidden_dim, num_hidden_layers, out_dim)
        g = torch.Generator().manual_seed(0)
        for key, shape in decoder_param_spec(self.cfg):
            leaf = key.rsplit(".", 1)[-1]
            if leaf == "num_batches_tracked":
                _register(self, key, torch.zeros((), dtype=torch.long), False)
            elif leaf == "running_mean":
                _register(self, key, torch.zeros(shape), False)
            elif leaf == "running_var":
                _register(self, key, torch.ones(shape), False)
            else:
                # same defaults as the reference: gamma/beta convs and fc_1 start at zero
                # (cbndec.py:62-66,97); the rest a fan-in uniform init
                if ".conv_gamma.bias" in key:
                    t = torch.ones(shape)
                elif ".conv_gamma." in key or ".conv_beta." in key or ".fc_1.weight" in key or (refine and "fc_out.weight" in key):
                    t = torch.zeros(shape)
                else:
                    fan_in = shape[1] if len(shape) > 1 else hidden_dim
                    bound = 1.0 / fan_in ** 0.5
                    t = (torch.rand(shape, generator=g) * 2 - 1) * bound
                _register(self, key, t, True)
        self._handle = None
        self._bound_key = None          # (device, versions) the native copy was made from
        self._latents_key = None
        self._key_slots = None

    # ---- native handle management -------------------------------------------------------------
    def _state_key(self):
        # (storage address, version) of every tensor of the state_dict.  The SLOTS — (owning dict, leaf name) in state_dict
        # order — are collected once (the module tree never changes; rebuilding a 101-entry state_dict on every udf call is
        # measurable in callback mode); the tensors are looked up in them on every call, so any rebinding is seen whether or
        # not it goes through a hook: `.to()` / `.cuda()`, `load_state_dict(assign=True)` on this module or a parent,
        # `decoder.blocks.0.fc_0.weight = nn.Parameter(...)`.
        if self._key_slots is None:
            slots = []
            for mod_name, mod in self.named_modules():
                for name in mod._parameters:
                    slots.append((mod._parameters, name))
                for name in mod._buffers:
                    if name not in mod._non_persistent_buffers_set:
                        slots.append((mod._buffers, name))
            self._key_slots = slots
        return tuple((t.data_ptr(), t._version) for d, n in self._key_slots for t in (d[n],) if t is not None)

    def _native(self):
        first = next(self.parameters())
        if not first.is_cuda:
            raise RuntimeError("CbnDecoder runs only on the GPU through libsurfd_hip.so (no CPU fallback); call .cuda() first")
        L = N.lib()
        key = self._state_key()
        if self._handle is None:
            h = C.c_void_p()
            c = self.cfg
            N.check(L.surfd_decoder_create(c.input_dim, c.latent_dim, c.hidden_dim, c.num_hidden_layers, C.byref(h)))
            self._handle = h
        if self._bound_key != key:
            st = N.stream()
            for k, v in self.state_dict(keep_vars=True).items():
                if k.endswith("num_batches_tracked"):
                    N.check(L.surfd_decoder_set_param(self._handle, k.encode(), None, N.shape_arr(()), 0, st))
                    continue
                t = v.detach()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.float().contiguous()
                N.check(L.surfd_decoder_set_param(self._handle, k.encode(), N.ptr(t), N.shape_arr(tuple(t.shape)), t.dim(), st))
            N.check(L.surfd_decoder_finalize(self._handle, st))
            torch.cuda.current_stream().synchronize()      # temporaries made above may be freed now
            self._bound_key = key
            self._latents_key = None
        return L, self._handle

    def set_precision(self, mode: str) -> None:
        """Arithmetic of the forward and the gradient kernel: 'f16x2' (default: split-fp16 on the fp16 matrix pipe,
        fp32-class accuracy, see csrc/decoder.hip) or 'fp32' (exact fp32 MFMA, bitwise independent of the tiling)."""
        L, h = self._native()
        N.check(L.surfd_decoder_set_precision(h, {"fp32": 0, "f16x2": 1}[mode]))

    def saturation_count(self, reset: bool = True) -> int:
        """Waves of the f16x2 forward kernel that clamped an activation to the fp16 range since the last reset;
        non-zero means this checkpoint / latent needs set_precision('fp32')."""
        import ctypes as C
        L, h = self._native()
        n = C.c_int64()
        N.check(L.surfd_decoder_saturation_count(h, int(reset), C.byref(n), N.stream()))
        return int(n.value)

    def sustained_clock_ghz(self, reset: bool = True) -> float:
        """Shader clock the chip held under the 8-wave forward kernel since the last reset (0.0: no launch) — the kernel is
        power-bound, so its reachable peak is the matrix peak times this over 2.4 GHz (bench.py roofline.sustained_clock_ghz)."""
        import ctypes as C
        L, h = self._native()
        g = C.c_double()
        N.check(L.surfd_decoder_sustained_clock(h, int(reset), C.byref(g), N.stream()))
        return float(g.value)

    def set_grid_blocks(self, blocks: int) -> None:
        """Persistent workgroups per decoder launch (0 = one per CU); fewer leaves CUs to other streams."""
        L, h = self._native()
        N.check(L.surfd_decoder_set_grid_blocks(h, int(blocks)))

    def bind_latents(self, latents: Tensor) -> None:
        """latents [S, D]: precompute the per-sample conditional-BN tables once."""
        L, h = self._native()
        lat = latents.detach().reshape(-1, self.cfg.latent_dim).float().contiguous()
        N.check(L.surfd_decoder_bind_latents(h, N.ptr(lat), lat.shape[0], N.stream()))
        self._latents_key = (lat.data_ptr(), lat._version, lat.shape[0], "explicit")
        self._keepalive = lat

    def _bind_single(self, lat: Tensor) -> int:
        # always rebind: writes by native kernels (a sampling loop's output, gather buffers) do not bump a tensor's
        # version counter, so (address, version) cannot prove the contents are the ones the tables were built from;
        # the table kernel is one tiny launch
        self.bind_latents(lat.reshape(1, -1))
        return 0

    # ---- kernels ---------------------------------------------------------------------------------
    def _logits_xyz(self, pts: Tensor, sample: int) -> Tensor:
        L, h = self._native()
        pts = pts.reshape(-1, 3).float().contiguous()
        out = torch.empty(pts.shape[0], device=pts.device, dtype=torch.float32)
        N.check(L.surfd_decoder_udf(h, sample, N.ptr(pts), pts.shape[0], None, N.ptr(out), N.stream()))
        return out

    def _dlogit_xyz(self, pts: Tensor, sample: int) -> Tensor:
        L, h = self._native()
        pts = pts.reshape(-1, 3).float().contiguous()
        out = torch.empty(pts.shape[0], 3, device=pts.device, dtype=torch.float32)
        N.check(L.surfd_decoder_udf_grad(h, sample, N.ptr(pts), pts.shape[0], None, None, N.ptr(out), N.stream()))
        return out

    def udf(self, pts: Tensor, sample: int = 0) -> Tensor:
        """udf_func semantics on bound latent `sample`: pts[n,3] -> udf[n]."""
        L, h = self._native()
        pts = pts.reshape(-1, 3).float().contiguous()
        out = torch.empty(pts.shape[0], device=pts.device, dtype=torch.float32)
        N.check(L.surfd_decoder_udf(h, sample, N.ptr(pts), pts.shape[0], N.ptr(out), None, N.stream()))
        return out

    def udf_and_ngrad(self, pts: Tensor, sample: int = 0):
        """(udf[n], -normalize(d udf/d p)[n,3]) — sample_grads semantics."""
        L, h = self._native()
        pts = pts.reshape(-1, 3).float().contiguous()
        udf = torch.empty(pts.shape[0], device=pts.device, dtype=torch.float32)
        ng = torch.empty(pts.shape[0], 3, device=pts.device, dtype=torch.float32)
        N.check(L.surfd_decoder_udf_grad(h, sample, N.ptr(pts), pts.shape[0], N.ptr(udf), N.ptr(ng), None, N.stream()))
        return udf, ng

    def forward(self, coords_emb, latent_codes: Tensor) -> Tensor:
        """coords_emb [B,n,63] (or the lazy result of CoordsEncoder.encode), latent [B,D] -> [B,n]."""
        if latent_codes.dim() != 2:
            raise NotImplementedError("per-point latents [B,n,D] are not on the sampling path (cbndec.py:131-132 broadcasts one latent)")
        B = latent_codes.shape[0]
        outs = []
        if B == 1:
            self._bind_single(latent_codes)
        else:
            self.bind_latents(latent_codes)
        for b in range(B):
            sample = b
            if isinstance(coords_emb, EncodedCoords):
                raw = coords_emb.raw[b]
                if raw.requires_grad and torch.is_grad_enabled():
                    outs.append(_DecoderFn.apply(raw, self, sample))
                else:
                    outs.append(self._logits_xyz(raw, sample))
            else:
                L, h = self._native()
                emb = coords_emb[b].detach().float().contiguous()
                out = torch.empty(emb.shape[0], device=emb.device, dtype=torch.float32)
                N.check(L.surfd_decoder_logits_emb(h, sample, N.ptr(emb), emb.shape[0], N.ptr(out), N.stream()))
                outs.append(out)
        return torch.stack(outs, 0)

    def __del__(self):
        try:
            if self._handle is not None:
                N.lib().surfd_decoder_destroy(self._handle)
        except Exception:
            pass


def make_udf_func(decoder: CbnDecoder, lat: Tensor, sample: Optional[int] = None) -> Callable[[Tensor], Tensor]:
    """The closure of sample/generate_uncond.py:96-101, fused: c[n,3] -> udf[n].

    The returned callable carries ``_surfd_native = (decoder, lat, sample)`` so GridFiller /
    get_mesh_from_udf can run the whole grid fill on the device without calling back."""
    lat = lat.detach().reshape(1, -1)

    def udf_func(c: Tensor) -> Tensor:
        s = decoder._bind_single(lat) if sample is None else sample
        if c.requires_grad and torch.is_grad_enabled():
            p = torch.sigmoid(_DecoderFn.apply(c, decoder, s))
            return (1 - p) * UDF_MAX_DIST
        return decoder.udf(c, s)

    udf_func._surfd_native = (decoder, lat, sample)
    return udf_func
