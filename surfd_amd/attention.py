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
