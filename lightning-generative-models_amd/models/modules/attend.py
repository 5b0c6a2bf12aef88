"""Path-compatible home of ``Attend`` (reference: models/modules/attend.py:97-126, non-flash branch, dropout 0).

Inside the UNet the attention core runs fused with its memory key/values straight from the qkv projection
(``Attention.fwd`` in models/generative/diffusion/ddpm.py -> ``lgm_attn_fwd``); this class offers the same HIP
kernel behind the reference's stand-alone call ``Attend()(q, k, v)`` with q ``[b, h, n, d]`` and k, v
``[b, h, n + M, d]`` (the reference concatenates M memory rows in front, ddpm.py:262-265).

Limits of the kernel, raised as NotImplementedError: d == 32, n <= 128, M <= 16, the M leading key/value rows equal
for every batch element (they are a broadcast parameter upstream), CUDA tensors.  Gradients flow for M == 0.
"""
from __future__ import annotations

import torch
from torch import nn

from lgm_hip import ops


class _AttendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        b, h, n, d = q.shape
        qkv = torch.cat((q, k, v), dim=1).permute(0, 2, 1, 3).reshape(b, n, 1, 3 * h * d).contiguous()   # [B,n,1,(q|k|v) x h x d]
        out = torch.empty((b, n, 1, h * d), device=q.device)
        mem = torch.zeros(16, device=q.device)                     # M = 0: never read
        lse = ops.attn_fwd(qkv, mem.data_ptr(), h, d, 0, out)
        ctx.save_for_backward(qkv, out, lse, mem)
        ctx.dims = (b, h, n, d)
        return out.reshape(b, n, h, d).permute(0, 2, 1, 3)

    @staticmethod
    def backward(ctx, gout):
        qkv, out, lse, mem = ctx.saved_tensors
        b, h, n, d = ctx.dims
        g = gout.permute(0, 2, 1, 3).reshape(b, n, 1, h * d).contiguous()
        gqkv = torch.empty_like(qkv)
        gmem = torch.zeros_like(mem)
        ops.attn_bwd(qkv, mem.data_ptr(), out, g, lse, h, d, 0, gqkv, gmem.data_ptr(), 0.0)
        gq, gk, gv = (t.permute(0, 2, 1, 3) for t in gqkv.reshape(b, n, 3 * h, d).split(h, dim=2))
        return gq, gk, gv


class Attend(nn.Module):
    def __init__(self, dropout: float = 0.0, flash: bool = False):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("HIP Attend: dropout is not part of the hot path (the reference UNet uses 0)")
        self.dropout, self.flash = dropout, flash

    def forward(self, q, k, v):
        b, h, n, d = q.shape
        M = k.shape[-2] - n
        if not q.is_cuda or d != 32 or n > 128 or M < 0 or M > 16 or k.shape != v.shape:
            raise NotImplementedError(f"HIP Attend: unsupported call q{tuple(q.shape)} k{tuple(k.shape)}")
        q, k, v = (t.float() for t in (q, k, v))
        if M == 0:
            return _AttendFn.apply(q, k, v)
        if torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v)):
            raise NotImplementedError("HIP Attend: gradients with memory rows go through models...ddpm.Attention")
        mk, mv = k[:, :, :M], v[:, :, :M]
        if b > 1 and not (torch.equal(mk, mk[:1].expand_as(mk)) and torch.equal(mv, mv[:1].expand_as(mv))):
            raise NotImplementedError("HIP Attend: the leading memory rows must be the same for every batch element")
        mem = torch.stack((mk[0], mv[0])).contiguous()                                   # [2, h, M, d]
        qkv = torch.cat((q, k[:, :, M:], v[:, :, M:]), dim=1).permute(0, 2, 1, 3).reshape(b, n, 1, 3 * h * d).contiguous()
        out = torch.empty((b, n, 1, h * d), device=q.device)
        ops.attn_fwd(qkv, mem.data_ptr(), h, d, M, out)
        return out.reshape(b, n, h, d).permute(0, 2, 1, 3)
