"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference's CrossAttention (modules/attention.py:152-193) as a
function of its state_dict.  Pinned by tests/golden/g14_cross_attention.npz, which tools/make_golden.py produced by
running the reference module itself.  Never imported by the product path."""
from __future__ import annotations

from typing import Dict, Optional

import torch


def cross_attention(sd: Dict[str, torch.Tensor], heads: int, x: torch.Tensor, context: Optional[torch.Tensor] = None,
                    mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [b,n,cq], context [b,m,cc] (None: x), mask [b,m] bool (None: all) -> [b,n,cq]."""
    ctx = x if context is None else context
    q = x @ sd["to_q.weight"].T                                   # :174
    k = ctx @ sd["to_k.weight"].T                                 # :176
    v = ctx @ sd["to_v.weight"].T                                 # :177
    b, n, inner = q.shape
    d = inner // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)     # 'b n (h d) -> (b h) n d' (:179)
    q, k, v = split(q), split(k), split(v)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)      # :181
    if mask is not None:
        keep = mask.reshape(b, 1, 1, -1).bool()
        sim = sim.masked_fill(~keep, -torch.finfo(sim.dtype).max)  # :183-187
    attn = sim.softmax(dim=-1)                                    # :190
    out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(b, n, inner)          # :192-193
    return out @ sd["to_out.0.weight"].T + sd["to_out.0.bias"]    # :194
