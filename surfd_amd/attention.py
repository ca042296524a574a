"""CrossAttention over [b, n, c] token tensors on the MI355X path — SURVEY.md §8 row a19.

Drop-in for the reference's ``modules.attention.CrossAttention`` (modules/attention.py:152-193): same constructor,
same ``state_dict`` keys (``to_q.weight``, ``to_k.weight``, ``to_v.weight``, ``to_out.0.weight``, ``to_out.0.bias``),
same ``forward(x, context=None, mask=None)``.  No Surf-D configuration instantiates the module (the UNet is built with
``use_spatial_transformer=False``, models/mdm.py:34-57), so it is not part of the sampling loop; it exists so that a
checkpoint / configuration that turns the LDM transformer blocks on has its attention on the matrix pipe.
Inference only: fp32 device tensors in, fp32 out, no autograd, no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor, nn

from . import _native as N


class CrossAttention(nn.Module):
    def __init__(self, query_dim: int, context_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64, dropout: float = 0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.query_dim, self.context_dim, self.heads, self.dim_head = query_dim, context_dim, heads, dim_head
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self._handle = None
        self._bound_key = None

    def _native(self):
        L = N.lib()
        sd = {k: v for k, v in self.state_dict().items()}
        key = tuple((k, v.data_ptr(), v._version) for k, v in sd.items())
        if self._handle is None:
            h = C.c_void_p()
            N.check(L.surfd_xattn_create(self.query_dim, self.context_dim, self.heads, self.dim_head, C.byref(h)))
            self._handle = h
        if key != self._bound_key:
            for k, v in sd.items():
                if not v.is_cuda:
                    raise RuntimeError("surfd_amd.attention.CrossAttention runs on the GPU only: move the module with .cuda() (no CPU path)")
                t = v.detach().float().contiguous()
                shape = (C.c_int64 * 4)(*t.shape)
                N.check(L.surfd_xattn_set_param(self._handle, k.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(), N.stream()))
            self._bound_key = key
        return L, self._handle

    @torch.no_grad()
    def forward(self, x: Tensor, context: Optional[Tensor] = None, mask: Optional[Tensor] = None) -> Tensor:
        if self.training and self.to_out[1].p > 0:
            raise RuntimeError("surfd_amd.attention.CrossAttention is an inference op (dropout > 0 in training mode is not supported)")
        if not x.is_cuda:
            raise RuntimeError("surfd_amd.attention.CrossAttention runs on the GPU only (no CPU path)")
        if x.dim() != 3 or x.shape[-1] != self.query_dim:
            raise ValueError(f"x must be [b, n, {self.query_dim}]")
        L, h = self._native()
        B, n, _ = x.shape
        x = x.float().contiguous()
        m = n
        cptr = None
        if context is not None:
            if context.dim() != 3 or context.shape[0] != B or context.shape[-1] != self.context_dim:
                raise ValueError(f"context must be [b, m, {self.context_dim}]")
            context = context.float().contiguous()
            m = context.shape[1]
            cptr = C.c_void_p(context.data_ptr())
        mptr = None
        if mask is not None:
            mask = mask.reshape(B, -1).to(torch.uint8).contiguous()       # 'b ... -> b (...)'
            if mask.shape[1] != m:
                raise ValueError("mask must flatten to [b, m]")
            mptr = C.c_void_p(mask.data_ptr())
        out = torch.empty_like(x)
        N.check(L.surfd_xattn_forward(h, C.c_void_p(x.data_ptr()), cptr, mptr, C.c_void_p(out.data_ptr()), B, n, m, N.stream()))
        return out

    def __del__(self):
        try:
            if self._handle is not None:
                N.lib().surfd_xattn_destroy(self._handle)
        except Exception:                                   # interpreter shutdown
            pass


class GEGLU(nn.Module):
    """x -> a * gelu(g) with (a, g) = proj(x) split in halves (modules/attention.py:37-44)."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x: Tensor) -> Tensor:
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * torch.nn.functional.gelu(g)


class FeedForward(nn.Module):
    """modules/attention.py:47-64: (Linear + GELU | GEGLU) -> Dropout -> Linear, same `net.{0,2}` parameter names."""

    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, glu: bool = False, dropout: float = 0.0):
        super().__init__()
        inner = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        first = GEGLU(dim, inner) if glu else nn.Sequential(nn.Linear(dim, inner), nn.GELU())
        self.net = nn.Sequential(first, nn.Dropout(dropout), nn.Linear(inner, dim_out))

    def forward(self, x: Tensor) -> Tensor:
        return self.net(x)


class BasicTransformerBlock(nn.Module):
    """Pre-norm self-attention, cross-attention (self when no context) and gated feed-forward, each with a residual
    (modules/attention.py:196-216), the two attentions on the native MFMA op.  Same state_dict keys as the reference
    (attn1.*, attn2.*, ff.net.*, norm1-3.*).  The LayerNorms and the feed-forward are library ops: the block is not on
    the sampling path of any Surf-D configuration (use_spatial_transformer=False everywhere, models/mdm.py:34-57)."""

    def __init__(self, dim: int, n_heads: int, d_head: int, dropout: float = 0.0, context_dim: Optional[int] = None, gated_ff: bool = True,
                 checkpoint: bool = True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    @torch.no_grad()
    def forward(self, x: Tensor, context: Optional[Tensor] = None) -> Tensor:
        x = self.attn1(self.norm1(x).contiguous()) + x
        x = self.attn2(self.norm2(x).contiguous(), context=context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    """modules/attention.py:219-261: GroupNorm(32, eps 1e-6) -> 1x1 conv -> tokens [b, h*w, c] -> `depth` transformer
    blocks -> back to the map -> 1x1 conv -> + input.  Same constructor and state_dict keys as the reference."""

    def __init__(self, in_channels: int, n_heads: int, d_head: int, depth: int = 1, dropout: float = 0.0, context_dim: Optional[int] = None):
        super().__init__()
        inner = n_heads * d_head
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim)
                                                 for _ in range(depth)])
        self.proj_out = nn.Conv2d(inner, in_channels, kernel_size=1)
        nn.init.zeros_(self.proj_out.weight)
        nn.init.zeros_(self.proj_out.bias)

    @torch.no_grad()
    def forward(self, x: Tensor, context: Optional[Tensor] = None) -> Tensor:
        b, c, h, w = x.shape
        t = self.proj_in(self.norm(x)).flatten(2).transpose(1, 2).contiguous()          # b (h w) c
        for blk in self.transformer_blocks:
            t = blk(t, context=context)
        t = t.transpose(1, 2).reshape(b, -1, h, w)
        return self.proj_out(t) + x
