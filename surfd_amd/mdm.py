"""Drop-ins for the reference's denoiser modules, executing on libsurfd_hip.so.

  MDM                         <- models/mdm.py:9-113   (same ctor kwargs, state_dict keys "Unet.*", forward(x, timesteps, y))
  ClassifierFreeSampleModel   <- models/cfg_sampler.py:8-26
  create_model_and_diffusion / load_model_wo_clip / get_model_args  <- utils/model_util.py:6-30

``MDM`` is an ``nn.Module`` that only *holds* the parameters (identical names/shapes, so a
reference checkpoint loads with ``load_model_wo_clip``); ``forward`` hands device pointers to
the native UNet.  There is no torch implementation to fall back to.

The CLIP text tower is an input producer outside the hot path and its
weights are not available offline: in 'text' mode pass the 512-d embedding as ``y['context']``
(it is constant over the loop, so it is encoded once instead of the reference's once per step,
mdm.py:96-97).
"""
from __future__ import annotations

import ctypes as C
from copy import deepcopy
from typing import Optional

import torch
from torch import Tensor, nn

from . import _native as N
from .cbndec import _register
from .spec import UNetConfig, unet_param_spec


class MDM(nn.Module):
    def __init__(self, modeltype="", num_actions=9, dropout=0.1, activation="gelu", legacy=False,
                 dataset="deepfasion3d", clip_dim=512, arch="OpenUNet", clip_version=None, **kargs):
        super().__init__()
        if arch != "OpenUNet":
            raise NotImplementedError(f"arch {arch!r}: only 'OpenUNet' exists in the reference (mdm.py:30)")
        self.legacy, self.modeltype, self.num_actions, self.dataset = legacy, modeltype, num_actions, dataset
        self.dropout, self.activation, self.clip_dim, self.arch = dropout, activation, clip_dim, arch
        self.cond_mode = kargs.get("cond_mode", "no_cond")
        self.cond_mask_prob = kargs.get("cond_mask_prob", 0.0)
        self.clip_version = clip_version
        self.num_classes = self.num_actions if "category" in self.cond_mode else None
        self.cfg = kargs.get("unet_cfg") or UNetConfig(context_dim=clip_dim, num_classes=self.num_classes)
        g = torch.Generator().manual_seed(0)
        zero_init = (".out_layers.3.", ".proj_out.", "Unet.out.2.")      # zero_module sites (openaimodel.py:229,312,685)
        for key, shape in unet_param_spec(self.cfg):
            leaf = key.rsplit(".", 1)[-1]
            is_norm = ".in_layers.0." in key or ".out_layers.0." in key or ".norm." in key or key.startswith("Unet.out.0.")
            if is_norm:
                t = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
            elif any(z in key for z in zero_init):
                t = torch.zeros(shape)
            elif "label_emb" in key:
                t = torch.randn(shape, generator=g)
            else:
                fan_in = 1
                for s in (shape[1:] if len(shape) > 1 else (shape[0],)):
                    fan_in *= s
                t = (torch.rand(shape, generator=g) * 2 - 1) / fan_in ** 0.5
            _register(self, key, t, True)
        self._handle = None
        self._bound_key = None

    def parameters_wo_clip(self):
        return [p for name, p in self.named_parameters() if not name.startswith("clip_model.")]

    # ---- native handle -------------------------------------------------------------------------------
    def _state_key(self):
        return tuple((v.data_ptr(), v._version) for v in self.parameters())

    def _native(self):
        first = next(self.parameters())
        if not first.is_cuda:
            raise RuntimeError("MDM runs only on the GPU through libsurfd_hip.so (no CPU fallback); call .to('cuda') first")
        L = N.lib()
        if self._handle is None:
            c = self.cfg
            cfg = N.UNetCfg(c.in_channels, c.model_channels, c.out_channels, c.num_res_blocks, len(c.channel_mult),
                            (C.c_int * 8)(*c.channel_mult), len(c.attention_resolutions),
                            (C.c_int * 8)(*c.attention_resolutions), c.num_heads, c.context_dim or 0, c.num_classes or 0)
            h = C.c_void_p()
            N.check(L.surfd_unet_create(C.byref(cfg), C.byref(h)))
            self._handle = h
        key = self._state_key()
        if self._bound_key != key:
            st = N.stream()
            for k, v in self.state_dict(keep_vars=True).items():
                if not k.startswith("Unet."):
                    continue
                t = v.detach()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.float().contiguous()
                N.check(L.surfd_unet_set_param(self._handle, k.encode(), N.ptr(t), N.shape_arr(tuple(t.shape)), t.dim(), st))
            N.check(L.surfd_unet_finalize(self._handle, st))
            torch.cuda.current_stream().synchronize()
            self._bound_key = key
        return L, self._handle

    def replica(self) -> "MDM":
        """A second execution context over the SAME parameter tensors: its own native handle (workspace, embedding
        table, captured loop graph), so independent sampling loops can run concurrently on different streams
        (surfd_amd.parallel.BatchPipeline).  No reference counterpart (the reference samples one batch at a time)."""
        import copy
        other = copy.copy(self)              # shallow: _parameters / _buffers dicts are shared, not copied
        other._handle = None
        other._bound_key = None
        return other

    def set_precision(self, mode: str) -> None:
        """'f16x2' (default: split-fp16 products on the fp16 matrix pipe, fp32 accumulation, operands clamped to the
        fp16 range and counted) or 'fp32' (exact fp32 MFMA) for the denoiser's convolutions."""
        L, h = self._native()
        N.check(L.surfd_unet_set_precision(h, {"fp32": 0, "f16x2": 1}[mode]))

    def set_cu_budget(self, cus: int) -> None:
        """Tell this execution context how many CUs its launches can count on (256 = whole chip, the default): sizes
        the conv kernel's split-K.  Set by throughput pipelines that run several loops next to the decoder."""
        L, h = self._native()
        N.check(L.surfd_unet_set_cu_budget(h, int(cus)))

    def set_wide(self, design_batch: int) -> None:
        """Work decomposition of the f16x2 conv kernel: 0 = latency form (one narrow loop alone on the chip, the
        default), n > 0 = wide form designed for loops over about n latents — the operand staging is shared by four
        row tiles and the K split no longer depends on the batch, so a latent's result is bit-identical whatever
        batch it rides in.  Applies to this execution context (replicas have their own setting)."""
        L, h = self._native()
        N.check(L.surfd_unet_set_wide(h, int(design_batch)))

    def saturation_count(self, reset: bool = True) -> int:
        """Workgroups of the f16x2 conv kernel that clamped an operand to +-65504 since the last reset; a non-zero
        value means this checkpoint leaves the range the mode is exact for -> use set_precision('fp32')."""
        L, h = self._native()
        n = C.c_int64()
        N.check(L.surfd_unet_saturation_count(h, int(reset), C.byref(n), N.stream()))
        return int(n.value)

    # ---- conditioning dispatch (mdm.py:91-110) -----------------------------------------------------------
    def conditioning(self, y: Optional[dict], B: int):
        """-> (context[B,512] | None, labels[B] int64 | None) as contiguous device tensors."""
        y = y or {}
        dev = next(self.parameters()).device
        ctx = cls = None
        if "sketch" in self.cond_mode or "img" in self.cond_mode:
            ctx = y["context"]
        elif self.cond_mode == "no_cond":
            pass
        elif "text" in self.cond_mode:
            if "context" not in y:
                raise RuntimeError("text mode: pass the CLIP text embedding as y['context'] (CLIP weights are not "
                                   "available offline; the tower is outside the sampling hot path)")
            ctx = y["context"]
        else:
            cls = y["action_text"]
        if ctx is not None:
            assert ctx.shape[0] == B, (ctx.shape, B)
            ctx = ctx.to(device=dev, dtype=torch.float32).contiguous()
        if cls is not None:
            assert cls.shape == (B,), "must specify y if and only if the model is class-conditional"
            cls = cls.to(device=dev, dtype=torch.long).contiguous()
        return ctx, cls

    def forward(self, x: Tensor, timesteps: Tensor, y=None) -> Tensor:
        """x [B,1,L], timesteps [B] (int) -> [B,1,L]."""
        L_, h = self._native()
        B, L = x.shape[0], x.shape[-1]
        ctx, cls = self.conditioning(y, B)
        xin = x.detach().to(torch.float32).contiguous()
        t = timesteps.to(device=x.device, dtype=torch.long).contiguous()
        out = torch.empty_like(xin)
        N.check(L_.surfd_unet_forward(h, N.ptr(xin), N.ptr(t), N.ptr(ctx), N.ptr(cls), N.ptr(out), B, L, N.stream()))
        return out

    def train(self, *args, **kwargs):          # the reference's override returns None (mdm.py:112-113)
        super().train(*args, **kwargs)

    def __del__(self):
        try:
            if self._handle is not None:
                N.lib().surfd_unet_destroy(self._handle)
        except Exception:
            pass


class ClassifierFreeSampleModel(nn.Module):
    """Classifier-free guidance wrapper with the reference's semantics (models/cfg_sampler.py:8-26): the model is
    evaluated with `y` and with a copy of `y` flagged 'uncond', and the two are blended with y['scale'].  MDM.forward
    never reads the flag, so both evaluations are identical and the result equals the conditional output bit for
    bit; the fused loop relies on that identity (tests/test_gpu_unet.py proves it)."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.cond_mode, self.clip_version = model.cond_mode, model.clip_version

    def forward(self, x, timesteps, y=None):
        if self.model.cond_mode not in ("text", "action"):
            raise AssertionError(f"classifier-free sampling needs a text/action model, got {self.model.cond_mode!r}")
        guidance = y["scale"].view(-1, 1, 1)                 # KeyError when the caller forgot the scale, as upstream
        cond = self.model(x, timesteps, y)
        flagged = deepcopy(y)
        flagged["uncond"] = True
        uncond = self.model(x, timesteps, flagged)
        return uncond + guidance * (cond - uncond)


def get_model_args(args):
    return {"modeltype": "", "num_actions": args.num_actions, "dropout": 0.1, "activation": "gelu",
            "cond_mode": args.cond_mode, "arch": args.arch, "clip_version": "ViT-B/32", "dataset": args.dataset}


def create_model_and_diffusion(args, timestep_respacing=""):
    from .diffusion import create_gaussian_diffusion
    return MDM(**get_model_args(args)), create_gaussian_diffusion(args, timestep_respacing)


def load_model_wo_clip(model, state_dict):
    missing_keys, _ = model.load_state_dict(state_dict, strict=False)
    assert all(k.startswith("clip_model.") for k in missing_keys)
