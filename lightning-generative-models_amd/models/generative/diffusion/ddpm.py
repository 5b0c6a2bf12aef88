"""DDPM / DDIM on the MI355X HIP engine — drop-in for the reference's
``models/generative/diffusion/ddpm.py`` (same class names, constructor arguments, state_dict
keys, ``training_step`` / ``configure_optimizers`` surface).

Nothing here calls ATen compute kernels on the hot path: the UNet forward AND backward are
explicit sequences of liblgm_hip.so launches over NHWC buffers (no autograd graph inside the
network); ``torch.autograd.Function`` is only the seam that lets ``loss.backward()`` of the
Lightning loop trigger the hand-written backward pass.

Reference line anchors are given per class.
"""
from __future__ import annotations

import math
import os
from collections import namedtuple
from typing import List, Optional, Tuple

import torch
from torch import nn

from lgm_hip import ops
from lgm_hip.flat import FlatParams, _r4
from lgm_hip.lightning import LightningModule, multi_rank
from lgm_hip.nn import Conv2d, GradCtx, GroupNorm, Linear, RMSNorm, param_kind
from lgm_hip.optim import EMA, FusedAdam


ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])     # reference :25


_NO_RES_FOLD = os.environ.get("LGM_NO_RES_FOLD") is not None       # A/B switch: the identity residual's gradient as its own axpby launch


def _chan(t: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
    return t[..., lo:hi]


# ----------------------------------------------------------------------------------------
# building blocks  (reference: Block :157-173, ResnetBlock :176-200)
# ----------------------------------------------------------------------------------------
class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.proj = Conv2d(dim, dim_out, 3, padding=1)
        self.norm = GroupNorm(groups, dim_out)


class ResnetBlock(nn.Module):
    """conv3x3 -> GN -> FiLM -> SiLU -> conv3x3 -> GN -> SiLU, + res_conv(x)."""

    def __init__(self, dim, dim_out, *, time_emb_dim, groups=8):
        super().__init__()
        self.dim, self.dim_out = dim, dim_out
        self.mlp = nn.Sequential(nn.Identity(), Linear(time_emb_dim, dim_out * 2))
        self.block1 = Block(dim, dim_out, groups)
        self.block2 = Block(dim_out, dim_out, groups)
        self.res_conv = Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def fwd(self, x, ss, out, save: bool):
        # a 3x3 convolution that splits its reduction (small maps, small batches) leaves partial planes; the
        # GroupNorm behind it sums them while it computes its statistics: no reducer launch, one round trip less
        G = self.block1.norm.groups
        u1, p1 = self.block1.proj.fwd_planes(x, G)
        h1, sv1 = self.block1.norm.fwd(u1, ss, True, None, planes=p1)
        u2, p2 = self.block2.proj.fwd_planes(h1, G)
        if isinstance(self.res_conv, Conv2d):
            h2, sv2 = self.block2.norm.fwd(u2, None, True, None, planes=p2)
            self.res_conv.fwd(x, out=out, res=h2)
        else:
            _, sv2 = self.block2.norm.fwd(u2, None, True, x, out=out, planes=p2)
        return (x, ss, u1, sv1, h1, u2, sv2) if save else None

    def bwd(self, gc: GradCtx, saved, gy, gss, gx, accumulate: bool):
        x, ss, u1, sv1, h1, u2, sv2 = saved
        # identity residual whose gradient joins a tensor that already holds another branch's (the skip connections of the
        # down path): gx += gy rides in block2's GroupNorm backward, which reads gy anyway (lgm_gn_bwd_add; was an axpby launch)
        fold_res = accumulate and not isinstance(self.res_conv, Conv2d) and not _NO_RES_FOLD
        gu2 = self.block2.norm.bwd(gc, u2, gy, None, True, sv2, None, add_gy_to=gx if fold_res else None)
        gh1, pg = self.block2.proj.bwd(gc, h1, gu2, planes_for_groups=self.block1.norm.groups)
        del gu2
        gu1 = self.block1.norm.bwd(gc, u1, gh1, ss, True, sv1, gss, gy_planes=pg)
        del gh1
        if isinstance(self.res_conv, Conv2d):
            self.block1.proj.bwd(gc, x, gu1, gx, accumulate)
            self.res_conv.bwd(gc, x, gy, gx, True)
        elif accumulate:
            if not fold_res:
                ops.axpby(gx, 1.0, gy, 1.0, gx)
            self.block1.proj.bwd(gc, x, gu1, gx, True)
        else:
            self.block1.proj.bwd(gc, x, gu1, gx, False, res=gy)
        return gx


class LinearAttention(nn.Module):
    """reference :203-239"""

    def __init__(self, dim, heads=4, dim_head=32, num_mem_kv=4):
        super().__init__()
        self.heads, self.dim_head, self.M = heads, dim_head, num_mem_kv
        hidden = heads * dim_head
        self.norm = RMSNorm(dim)
        self.mem_kv = nn.Parameter(torch.randn(2, heads, dim_head, num_mem_kv))
        self.to_qkv = Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(Conv2d(hidden, dim, 1), RMSNorm(dim))

    def fwd(self, x, out, save: bool):
        """out = attn(x) + x"""
        B, H, W, C = x.shape
        fp = self.mem_kv._lgm_flat
        r = ops.rms_qkv_fused(x, fp.ptr(self.norm.g), fp.ptr(self.to_qkv.weight), 3 * self.heads * self.dim_head)
        if r is not None:
            xn, qkv = r                                  # RMSNorm + to_qkv in one launch
        else:
            xn = self.norm.fwd(x)
            qkv = self.to_qkv.fwd(xn)
        ao = ops.new((B, H, W, self.heads * self.dim_head), x)
        conv, norm = self.to_out[0], self.to_out[1]
        if ops.linattn_fwd_fused_ok(self.heads, self.dim_head, C, qkv, x, fp.ptr(conv.weight), fp.ptr(conv.bias),
                                    fp.ptr(norm.g)) and out.data_ptr() % 16 == 0 and ops.pitch(out) % 4 == 0:
            # softmax_d(q) ctx -> to_out[0] -> RMSNorm -> + x in one launch behind the context launch
            o2 = ops.new((B, H, W, C), x)
            ctx, kstat = ops.linattn_fwd_fused(qkv, fp.ptr(self.mem_kv), self.heads, self.dim_head, self.M,
                                               fp.ptr(conv.weight), fp.ptr(conv.bias), fp.ptr(norm.g), x, ao, o2, out)
        else:
            ctx, kstat = ops.linattn_fwd(qkv, fp.ptr(self.mem_kv), self.heads, self.dim_head, self.M, ao)
            o2 = conv.fwd(ao)
            norm.fwd(o2, res=x, out=out)
        return (x, xn, qkv, ao, ctx, kstat, o2) if save else None

    def bwd(self, gc: GradCtx, saved, gy, gx, accumulate: bool):
        x, xn, qkv, ao, ctx, kstat, o2 = saved
        fp = gc.flat
        go2 = self.to_out[1].bwd(gc, o2, gy)
        gao = self.to_out[0].bwd(gc, ao, go2)
        del go2
        w = self.to_qkv.weight
        gqkv = None
        if ops.linattn_bwd_fused_ok(self.heads, self.dim_head, xn.shape[-1], qkv, gao, xn, fp.tptr(w), fp.gptr(w),
                                    fp.gptr(self.mem_kv)):
            # large maps: gq / gk / gv never reach memory - the kernel that computes them also multiplies them
            # with to_qkv's weight (input gradient) and with xn (weight gradient)
            dw, dm = gc.defer_for(w), gc.defer_for(self.mem_kv)
            bw, bm = gc.beta(w), gc.beta(self.mem_kv)
            gxn = ops.new(xn.shape, xn)
            ops.linattn_bwd_fused(qkv, fp.ptr(self.mem_kv), gao, ctx, kstat, xn, fp.tptr(w), self.heads, self.dim_head,
                                  self.M, gxn, fp.gptr(w), bw, dw, fp.gptr(self.mem_kv), bm, dm)
        else:
            gqkv = ops.new(qkv.shape, qkv)
            dm = gc.defer_for(self.mem_kv)
            ops.linattn_bwd(qkv, fp.ptr(self.mem_kv), gao, ctx, kstat, self.heads, self.dim_head, self.M, gqkv,
                            fp.gptr(self.mem_kv), gc.beta(self.mem_kv), defer=dm)
            gxn = self.to_qkv.bwd(gc, xn, gqkv)
        del gqkv, gao
        # gx (+)= d norm / dx + gy: the residual branch's gradient rides in the RMSNorm backward pass
        self.norm.bwd(gc, x, gxn, gx, accumulate, res=gy)
        return gx


class Attention(nn.Module):
    """reference :242-271 (+ modules/attend.py:97-126)"""

    def __init__(self, dim, heads=4, dim_head=32, num_mem_kv=4, flash=False):
        super().__init__()
        self.heads, self.dim_head, self.M = heads, dim_head, num_mem_kv
        hidden = heads * dim_head
        self.norm = RMSNorm(dim)
        self.mem_kv = nn.Parameter(torch.randn(2, heads, num_mem_kv, dim_head))
        self.to_qkv = Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = Conv2d(hidden, dim, 1)

    def fwd(self, x, out, save: bool):
        B, H, W, C = x.shape
        fp = self.mem_kv._lgm_flat
        r = ops.rms_qkv_fused(x, fp.ptr(self.norm.g), fp.ptr(self.to_qkv.weight), 3 * self.heads * self.dim_head)
        if r is not None:
            xn, qkv = r                                  # RMSNorm + to_qkv in one launch
        else:
            xn = self.norm.fwd(x)
            qkv = self.to_qkv.fwd(xn)
        ao = ops.new((B, H, W, self.heads * self.dim_head), x)
        lse = ops.attn_fwd(qkv, fp.ptr(self.mem_kv), self.heads, self.dim_head, self.M, ao)
        self.to_out.fwd(ao, out=out, res=x)
        return (x, xn, qkv, ao, lse) if save else None

    def bwd(self, gc: GradCtx, saved, gy, gx, accumulate: bool):
        x, xn, qkv, ao, lse = saved
        fp = gc.flat
        gao = self.to_out.bwd(gc, ao, gy)
        gqkv = ops.new(qkv.shape, qkv)
        dm = gc.defer_for(self.mem_kv)
        ops.attn_bwd(qkv, fp.ptr(self.mem_kv), ao, gao, lse, self.heads, self.dim_head, self.M, gqkv,
                     fp.gptr(self.mem_kv), gc.beta(self.mem_kv), defer=dm)
        gxn = self.to_qkv.bwd(gc, xn, gqkv)
        del gqkv, gao
        # gx (+)= d norm / dx + gy: the residual branch's gradient rides in the RMSNorm backward pass
        self.norm.bwd(gc, x, gxn, gx, accumulate, res=gy)
        return gx


class _DownConv(Conv2d):
    """The convolution of ``Downsample`` (:100-104: ``Rearrange('b c (h p1) (w p2) -> b (c p1 p2) h w')`` + ``Conv2d(4 C, N, 1)``)
    as what the two are together: a 2 x 2 / stride-2 convolution of the un-shuffled tensor (the index shuffle lives
    in the convolution's gather, nothing is materialised).  The parameter is held as ``[N, C, 2, 2]``; its row-major
    flattening IS the reference's ``[N, 4 C, 1, 1]`` weight (input channel ``c p1 p2`` = ``4 c + 2 p1 + p2``), same fan-in, same
    initialisation stream.  ``state_dict()`` / ``load_state_dict()`` and the optimizer's checkpoint state exchange the
    reference's shape (``ParamSlot.ref_shape``)."""

    def __init__(self, dim, dim_out):
        super().__init__(dim, dim_out, 2, stride=2, padding=0)

    @property
    def ref_shape(self):
        return (self.cout, 4 * self.cin, 1, 1)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        k = prefix + "weight"
        destination[k] = destination[k].reshape(self.ref_shape)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        k = prefix + "weight"
        w = state_dict.get(k)
        if w is not None and tuple(w.shape) == self.ref_shape:
            state_dict = {kk: v for kk, v in state_dict.items() if kk.startswith(prefix)}
            state_dict[k] = w.reshape(self.weight.shape)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


class _Down(nn.Module):
    """Downsample (:100-104): pixel-unshuffle + 1x1 conv = one 2x2 / stride-2 convolution (``_DownConv``); keys
    ``<idx>.1.weight`` / ``<idx>.1.bias``."""

    def __init__(self, dim, dim_out):
        super().__init__()
        self.add_module("0", nn.Identity())
        self.add_module("1", _DownConv(dim, dim_out))

    def fwd(self, x, out, save):
        self._modules["1"].fwd(x, out=out)
        return (x,) if save else None

    def bwd(self, gc, saved, gy, gx, accumulate):
        (x,) = saved
        self._modules["1"].bwd(gc, x, gy, gx, accumulate)
        return gx


class _Up(nn.Module):
    """Upsample (:93-97): nearest x2 + conv3x3; keys ``<idx>.1.weight``."""

    def __init__(self, dim, dim_out):
        super().__init__()
        self.add_module("0", nn.Identity())
        self.add_module("1", Conv2d(dim, dim_out, 3, padding=1))

    def fwd(self, x, out, save):
        B, H, W, C = x.shape
        hi = ops.new((B, 2 * H, 2 * W, C), x)
        ops.upsample2x_fwd(x, hi)
        self._modules["1"].fwd(hi, out=out)
        return (hi,) if save else None

    def bwd(self, gc, saved, gy, gx, accumulate):
        (hi,) = saved
        ghi = self._modules["1"].bwd(gc, hi, gy)
        ops.upsample2x_bwd(ghi, gx, accumulate)
        return gx


class _PlainConv(Conv2d):
    """Last-stage 3x3 conv used in place of Down/Up (:377, :413)."""

    def fwd_s(self, x, out, save):
        self.fwd(x, out=out)
        return (x,) if save else None

    def bwd_s(self, gc, saved, gy, gx, accumulate):
        (x,) = saved
        self.bwd(gc, x, gy, gx, accumulate)
        return gx


# ----------------------------------------------------------------------------------------
# UNet  (reference :275-471)
# ----------------------------------------------------------------------------------------
class Unet(nn.Module):
    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3,
                 self_condition=False, resnet_block_groups=8, learned_variance=False,
                 learned_sinusoidal_cond=False, random_fourier_features=False,
                 learned_sinusoidal_dim=16, sinusoidal_pos_emb_theta=10000, attn_dim_head=32,
                 attn_heads=4, full_attn=None, flash_attn=False):
        super().__init__()
        if self_condition or learned_variance or learned_sinusoidal_cond or random_fourier_features:
            raise NotImplementedError("HIP UNet covers the configuration the reference DDPM constructs "
                                      "(ddpm.py:984-987): no self-conditioning / learned variance / fourier features")
        if init_dim not in (None, dim):
            raise NotImplementedError("init_dim != dim")
        self.dim, self.channels = dim, channels
        self.self_condition = False
        self.random_or_learned_sinusoidal_cond = False
        self.theta = float(sinusoidal_pos_emb_theta)
        self.init_conv = Conv2d(channels, dim, 7, padding=3)
        dims = [dim, *[dim * m for m in dim_mults]]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.in_out = in_out
        time_dim = dim * 4
        self.time_dim = time_dim
        self.time_mlp = nn.Sequential(nn.Identity(), Linear(dim, time_dim), nn.Identity(), Linear(time_dim, time_dim))
        n = len(in_out)
        if not full_attn:
            full_attn = (*((False,) * (n - 1)), True)
        heads = attn_heads if isinstance(attn_heads, tuple) else (attn_heads,) * n
        dheads = attn_dim_head if isinstance(attn_dim_head, tuple) else (attn_dim_head,) * n
        rb = lambda a, b: ResnetBlock(a, b, time_emb_dim=time_dim, groups=resnet_block_groups)  # noqa: E731
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        for i, ((ci, co), fa, h, dh) in enumerate(zip(in_out, full_attn, heads, dheads)):
            last = i >= n - 1
            att = Attention(ci, heads=h, dim_head=dh) if fa else LinearAttention(ci, heads=h, dim_head=dh)
            self.downs.append(nn.ModuleList([rb(ci, ci), rb(ci, ci), att,
                                             _Down(ci, co) if not last else _PlainConv(ci, co, 3, padding=1)]))
        mid = dims[-1]
        self.mid_block1 = rb(mid, mid)
        self.mid_attn = Attention(mid, heads=heads[-1], dim_head=dheads[-1])
        self.mid_block2 = rb(mid, mid)
        for i, ((ci, co), fa, h, dh) in enumerate(zip(*map(reversed, (in_out, full_attn, heads, dheads)))):
            last = i == n - 1
            att = Attention(co, heads=h, dim_head=dh) if fa else LinearAttention(co, heads=h, dim_head=dh)
            self.ups.append(nn.ModuleList([rb(co + ci, co), rb(co + ci, co), att,
                                           _Up(co, ci) if not last else _PlainConv(co, ci, 3, padding=1)]))
        self.out_dim = out_dim if out_dim is not None else channels
        self.final_res_block = rb(dim * 2, dim)
        self.final_conv = Conv2d(dim, self.out_dim, 1)
        self._flat: Optional[FlatParams] = None

    # ---- flat storage ---------------------------------------------------------------------
    def resblocks(self) -> List[ResnetBlock]:
        out = []
        for b1, b2, _, _ in self.downs:
            out += [b1, b2]
        out += [self.mid_block1, self.mid_block2]
        for b1, b2, _, _ in self.ups:
            out += [b1, b2]
        out.append(self.final_res_block)
        return out

    def __deepcopy__(self, memo):
        # EMA shadow copies must get their own flat storage: copy with plain (unbound) parameters
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_flat":
                new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        for p in new.parameters():
            p.data = p.data.contiguous().clone()
        return new

    def prepare_hip(self, device) -> FlatParams:
        """Bind all parameters to flat device storage (idempotent)."""
        device = torch.device(device)
        if self._flat is not None and self._flat.device == device and self._flat.still_bound():
            return self._flat
        named = dict(self.named_parameters())
        rbs = self.resblocks()
        first: List[Tuple[str, nn.Parameter, str]] = []
        names = {id(p): n for n, p in named.items()}
        for rb in rbs:   # the 2C x time_dim FiLM projections become ONE [sum(2C), time_dim] GEMM operand
            first.append((names[id(rb.mlp[1].weight)], rb.mlp[1].weight, "weight"))
        for rb in rbs:
            first.append((names[id(rb.mlp[1].bias)], rb.mlp[1].bias, "vector"))
        for lin in (self.time_mlp[1], self.time_mlp[3]):      # their gradients are produced last, too
            first.append((names[id(lin.weight)], lin.weight, "weight"))
            first.append((names[id(lin.bias)], lin.bias, "vector"))
        taken = {id(p) for _, p, _ in first}
        rest = [(n, p, param_kind(n, p)) for n, p in named.items() if id(p) not in taken]
        self._flat = FlatParams(first + rest, device)
        for m in self.modules():
            if isinstance(m, _DownConv):       # checkpoints exchange this weight as the reference's [N, 4 C, 1, 1]
                self._flat.slot(m.weight).ref_shape = m.ref_shape
        if ops.B3:                     # opt-in split-precision 3x3 convolutions (LGM_CONV_MODE=bf16x3)
            self._flat.enable_b3()
        elif ops.WINO:                 # Winograd F(2x2,3x3) in exact fp32 arithmetic for the 3x3 layers
            self._flat.enable_wino()
        # gradient-exchange buckets in backward completion order: [ups, mid, final] -> [init_conv, downs]
        # -> [FiLM + time MLP]  (registration order of `rest`: init_conv, downs, ups, mid_*, final_*)
        slots = {s.name: s for s in self._flat.slots}
        self._head_end = slots[rest[0][0]].offset
        rest_names = [n for n, _, _ in rest]
        self._ups_start = min(slots[n].offset for n in rest_names if n.startswith("ups."))
        self._mid_start = min(slots[n].offset for n in rest_names if n.startswith("mid_"))
        self._final_start = min(slots[n].offset for n in rest_names if n.startswith("final_"))
        assert self._ups_start < self._mid_start < self._final_start
        assert all(self._ups_start <= slots[n].offset < self._mid_start for n in rest_names if n.startswith("ups."))
        assert all(self._mid_start <= slots[n].offset < self._final_start for n in rest_names if n.startswith("mid_"))
        assert all(slots[n].offset >= self._final_start for n in rest_names if n.startswith("final_"))
        assert all(slots[n].offset >= self._ups_start for n in rest_names
                   if n.startswith(("ups.", "mid_", "final_")))
        assert all(self._head_end <= slots[n].offset < self._ups_start for n in rest_names
                   if n.startswith(("downs.", "init_conv")))
        self.grad_sync = None
        self._ss_offsets = []
        off = 0
        for rb in rbs:
            self._ss_offsets.append(off)
            off += 2 * rb.dim_out
        self._ss_total = off
        self._mlp_geoms = {}
        return self._flat

    # ---- time embedding -------------------------------------------------------------------
    def _time_fwd(self, t, save):
        B = t.shape[0]
        fp = self._flat
        l1, l2 = self.time_mlp[1], self.time_mlp[3]
        pe = ops.new((B, self.dim), t)
        a1 = ops.new((B, self.time_dim), t)
        h = ops.new((B, self.time_dim), t)
        temb = ops.new((B, self.time_dim), t)
        st = ops.new((B, self.time_dim), t)
        if ops.time_mlp_ok(self.dim, self.time_dim, l1, l2):
            # posemb -> Linear -> GELU -> Linear -> SiLU in ONE launch (was six: csrc/elementwise.hip time_mlp_fwd_kernel)
            ops.time_mlp_fwd(t, self.dim, self.theta, fp.ptr(l1.weight), fp.ptr(l1.bias), fp.ptr(l2.weight),
                             fp.ptr(l2.bias), self.time_dim, pe, a1, h, temb, st)
        else:
            ops.posemb(t, self.dim, self.theta, pe)
            ops.conv_xy(l1.geom(B), pe, fp.ptr(l1.weight), fp.ptr(l1.bias), None, a1)
            ops.act_fwd(a1, None, None, h, ops.ACT_GELU)
            ops.conv_xy(l2.geom(B), h, fp.ptr(l2.weight), fp.ptr(l2.bias), None, temb)
            ops.act_fwd(temb, None, None, st, ops.ACT_SILU)
        g = self._mlp_geoms.get(B)
        if g is None:
            g = ops.make_geom(B, 1, 1, self.time_dim, self._ss_total, 1, 1, 1, 0)
            self._mlp_geoms[B] = g
        rb0 = self.resblocks()[0]
        ss_all = ops.new((B, self._ss_total), t)
        ops.conv_xy(g, st, fp.ptr(rb0.mlp[1].weight), fp.ptr(rb0.mlp[1].bias), None, ss_all)
        return ss_all, ((pe, a1, h, temb, st) if save else None)

    def _time_bwd(self, gc: GradCtx, saved, gss_all):
        pe, a1, h, temb, st = saved
        B = pe.shape[0]
        fp = gc.flat
        l1, l2 = self.time_mlp[1], self.time_mlp[3]
        rbs = self.resblocks()
        g = self._mlp_geoms[B]
        ops.conv_wgrad(g, gss_all, st, fp.gptr(rbs[0].mlp[1].weight), gc.beta0, fp.gptr(rbs[0].mlp[1].bias))
        for rb in rbs:
            gc.written.add(id(rb.mlp[1].weight))
            gc.written.add(id(rb.mlp[1].bias))
        gst = ops.new(st.shape, st)
        ops.conv_yx(g, gss_all, fp.ptr(rbs[0].mlp[1].weight), None, None, gst)
        gtemb = ops.new(st.shape, st)
        if ops.time_mlp_ok(self.dim, self.time_dim, l1, l2):
            # SiLU' -> (weight / bias gradient, input gradient) of the second linear -> GELU' -> weight / bias gradient of the
            # first: two launches (row-local chain, batch reductions in row order) instead of six
            bw = gc.beta(l2.weight)
            assert bw == gc.beta(l2.bias) == gc.beta(l1.weight) == gc.beta(l1.bias)
            ga1 = ops.new(a1.shape, a1)
            ops.time_mlp_bwd(gst, pe, a1, h, temb, fp.ptr(l2.weight), self.dim, self.time_dim, gtemb, ga1,
                             fp.gptr(l1.weight), fp.gptr(l1.bias), fp.gptr(l2.weight), fp.gptr(l2.bias), bw)
            return
        ops.act_bwd(temb, None, gst, gtemb, False, ops.ACT_SILU)
        gh = ops.new(h.shape, h)
        # weight gradient and input gradient of the second time-MLP linear in one launch (lgm_conv_bwd_pair)
        ops.conv_bwd_generic(l2.geom(B), gtemb, h, fp.ptr(l2.weight), fp.tptr(l2.weight), fp.gptr(l2.weight),
                             gc.beta(l2.weight), fp.gptr(l2.bias), None, None, gh)
        gc.beta(l2.bias)
        ga1 = ops.new(a1.shape, a1)
        ops.act_bwd(a1, None, gh, ga1, False, ops.ACT_GELU)
        ops.conv_wgrad(l1.geom(B), ga1, pe, fp.gptr(l1.weight), gc.beta(l1.weight), fp.gptr(l1.bias))
        gc.beta(l1.bias)

    # ---- network ----------------------------------------------------------------------------
    def forward_nhwc(self, x, t, save: bool, refresh_weights: bool = True):
        """x: [B, S, S, r4(channels)] NHWC (pad lanes zero), t: int64 [B].
        Returns (out [B,S,S,r4(out_dim)], tape).  ``refresh_weights=False``: the derived weight copies (Winograd /
        split-precision operands) are known to be current — a sampling chain refreshes them once, not per step."""
        B, S, _, _ = x.shape
        dim = self.dim
        n = len(self.in_out)
        assert S % (2 ** (n - 1)) == 0, f"input size {S} must be divisible by {2 ** (n - 1)}"
        if refresh_weights:
            self.refresh_derived_weights(save)
        ss_all, time_saved = self._time_fwd(t, save)
        ssl = [ss_all[:, o:o + 2 * rb.dim_out] for o, rb in zip(self._ss_offsets, self.resblocks())]
        k = 0  # running resblock index
        tape = []
        # final concat buffer (x, r): r = init_conv output lives in its upper half
        catF = ops.new((B, S, S, 2 * dim), x)
        r = _chan(catF, dim, 2 * dim)
        self.init_conv.fwd(x, out=r)
        cur = r
        res = S
        cats = []
        for s, (b1, b2, attn, down) in enumerate(self.downs):
            ci, co = self.in_out[s]
            cat1 = ops.new((B, res, res, co + ci), x)   # (x_up, h_b)
            cat2 = ops.new((B, res, res, co + ci), x)   # (x_up', h_a)
            cats.append((cat1, cat2))
            ha = _chan(cat2, co, co + ci)
            hb = _chan(cat1, co, co + ci)
            s1 = b1.fwd(cur, ssl[k], ha, save); k += 1
            mid = ops.new((B, res, res, ci), x)
            s2 = b2.fwd(ha, ssl[k], mid, save); k += 1
            s3 = attn.fwd(mid, hb, save)
            last = s == n - 1
            nres = res if last else res // 2
            nxt = ops.new((B, nres, nres, co), x)
            s4 = down.fwd_s(hb, nxt, save) if last else down.fwd(hb, nxt, save)
            tape.append((s1, s2, s3, s4))
            cur, res = nxt, nres
        m1 = ops.new(cur.shape, x)
        sm1 = self.mid_block1.fwd(cur, ssl[k], m1, save); k += 1
        m2 = ops.new(cur.shape, x)
        sm2 = self.mid_attn.fwd(m1, m2, save)
        # mid_block2 writes straight into the first up-stage concat buffer
        ups_tape = []
        for u, (b1, b2, attn, up) in enumerate(self.ups):
            s = n - 1 - u
            ci, co = self.in_out[s]
            cat1, cat2 = cats[s]
            xa = _chan(cat1, 0, co)
            if u == 0:
                sm3 = self.mid_block2.fwd(m2, ssl[k], xa, save); k += 1
            # (for u > 0 the previous up-stage already wrote xa)
            xb = _chan(cat2, 0, co)
            s1 = b1.fwd(cat1, ssl[k], xb, save); k += 1
            y2 = ops.new((B, res, res, co), x)
            s2 = b2.fwd(cat2, ssl[k], y2, save); k += 1
            y3 = ops.new((B, res, res, co), x)
            s3 = attn.fwd(y2, y3, save)
            last = u == n - 1
            if last:
                dst = _chan(catF, 0, dim)
                s4 = up.fwd_s(y3, dst, save)
            else:
                ci_n, co_n = self.in_out[s - 1]
                dst = _chan(cats[s - 1][0], 0, co_n)
                s4 = up.fwd(y3, dst, save)
                res *= 2
            ups_tape.append((s1, s2, s3, s4))
        fin = ops.new((B, S, S, dim), x)
        sf = self.final_res_block.fwd(catF, ssl[k], fin, save); k += 1
        out = self.final_conv.fwd(fin)
        if not save:
            return out, None
        return out, (time_saved, tape, sm1, sm2, sm3, ups_tape, sf, fin, x, [c[0].shape for c in cats])

    def refresh_derived_weights(self, for_backward: bool):
        if self._flat.b3:
            self._flat.refresh_split()      # bf16 planes of the current weights (one launch)
        if self._flat.wino:
            self._flat.refresh_wino(backward_operand=for_backward)   # U = G g G^T of the current weights (one launch)

    def backward_nhwc(self, tape_all, gout):
        """Hand-written backward pass: parameter gradients into the flat gradient buffer."""
        st = self.backward_phase1(tape_all, gout)
        self.backward_phase2(st)

    def _flush(self, gc: GradCtx, bucket: int):
        """End of an exchange bucket (0: ups + final, 1: mid, 2: init + downs, 3: FiLM + time): the deferred slab / row
        reductions of its layers.  Normally ONE batched launch right here.  While a step is being captured with
        ``_flush_collect`` set (lgm_hip.graph.GraphedDDPMStep) the descriptor rows are handed to the step object
        instead, which launches the reduction - and the bucket's all-reduce and Adam slice behind it - on a side stream
        next to the following backward phase: weight-sized, HBM-bound passes beside MFMA-bound convolutions."""
        col = getattr(self, "_flush_collect", None)
        gc.finish_pending()              # a large-map weight gradient still waiting for a partner (GradCtx.queue_wgrad)
        if col is None:
            gc.flush()
        else:
            col[bucket] = list(gc.deferred or [])
            if gc.deferred is not None:
                gc.deferred.clear()

    def bucket_ranges(self):
        """Flat-buffer slices of the four exchange buckets, in backward completion order."""
        t = self._flat.total
        return [[(self._ups_start, self._mid_start), (self._final_start, t)], [(self._mid_start, self._final_start)],
                [(self._head_end, self._ups_start)], [(0, self._head_end)]]

    def backward_phase1(self, tape_all, gout):
        """final conv -> final block -> up path -> middle.  After it the gradient slice
        [_ups_start, total) of the flat buffer is final (first exchange buckets)."""
        return self.backward_phase1b(self.backward_phase1a(tape_all, gout))

    def backward_phase1a(self, tape_all, gout):
        """final conv -> final block -> up path.  After it the slices [_ups_start, _mid_start) and
        [_final_start, total) of the flat gradient buffer are final: 18.6 M of the 35.7 M parameters, whose exchange
        overlaps everything that follows."""
        time_saved, tape, sm1, sm2, sm3, ups_tape, sf, fin, x_in, cat_shapes = tape_all
        fp = self._flat
        gc = GradCtx(fp, defer=True)     # weight-gradient slabs: one batched reduce per exchange bucket
        B, S = x_in.shape[0], x_in.shape[1]
        dim = self.dim
        n = len(self.in_out)
        rbs = self.resblocks()
        gss_all = ops.new((B, self._ss_total), x_in)
        gsl = [gss_all[:, o:o + 2 * rb.dim_out] for o, rb in zip(self._ss_offsets, rbs)]
        k = len(rbs) - 1
        gfin = self.final_conv.bwd(gc, fin, gout)
        gcatF = ops.new((B, S, S, 2 * dim), x_in)
        self.final_res_block.bwd(gc, sf, gfin, gsl[k], gcatF, False); k -= 1
        del gfin
        gcats = [None] * n
        g_next = _chan(gcatF, 0, dim)     # grad of the last up-stage output
        for u in range(n - 1, -1, -1):
            b1, b2, attn, up = self.ups[u]
            s = n - 1 - u
            ci, co = self.in_out[s]
            s1, s2, s3, s4 = ups_tape[u]
            shp = cat_shapes[s]
            res = shp[1]
            gy3 = ops.new((B, res, res, co), x_in)
            if u == n - 1:
                up.bwd_s(gc, s4, g_next, gy3, False)
            else:
                up.bwd(gc, s4, g_next, gy3, False)
            gy2 = ops.new((B, res, res, co), x_in)
            attn.bwd(gc, s3, gy3, gy2, False)
            del gy3
            gcat2 = ops.new(shp, x_in)
            b2.bwd(gc, s2, gy2, gsl[k], gcat2, False); k -= 1
            del gy2
            gcat1 = ops.new(shp, x_in)
            b1.bwd(gc, s1, _chan(gcat2, 0, co), gsl[k], gcat1, False); k -= 1
            gcats[s] = (gcat1, gcat2)
            g_next = _chan(gcat1, 0, co)
        self._flush(gc, 0)
        sync = getattr(self, "grad_sync", None)
        if sync is not None:
            sync.ready(self._ups_start, self._mid_start)
            sync.ready(self._final_start, fp.total)
        return dict(gc=gc, tape=tape, time_saved=time_saved, gss_all=gss_all, gsl=gsl, k=k, gcats=gcats,
                    gcatF=gcatF, g_next=g_next, x_in=x_in, mid=(sm1, sm2, sm3))

    def backward_phase1b(self, st):
        """middle blocks.  After it [_mid_start, _final_start) is final (10.2 M parameters)."""
        gc, gsl, k, g_next, x_in = (st[key] for key in ("gc", "gsl", "k", "g_next", "x_in"))
        sm1, sm2, sm3 = st["mid"]
        gm2 = ops.new(sm3[0].shape, x_in)
        self.mid_block2.bwd(gc, sm3, g_next, gsl[k], gm2, False); k -= 1
        gm1 = ops.new(gm2.shape, x_in)
        self.mid_attn.bwd(gc, sm2, gm2, gm1, False)
        del gm2
        gcur = ops.new(gm1.shape, x_in)
        self.mid_block1.bwd(gc, sm1, gm1, gsl[k], gcur, False); k -= 1
        del gm1
        self._flush(gc, 1)
        sync = getattr(self, "grad_sync", None)
        if sync is not None:
            sync.ready(self._mid_start, self._final_start)
        st = dict(st)
        st.update(k=k, gcur=gcur)
        return st

    def backward_phase2(self, st):
        """down path -> init conv -> time embedding / FiLM projections."""
        self.backward_phase2b(self.backward_phase2a(st))

    def backward_phase2a(self, st):
        """down path -> init conv.  After it [_head_end, _ups_start) is final."""
        gc, tape, gsl, k, gcats, gcatF, gcur, x_in = (st[key] for key in
                                                      ("gc", "tape", "gsl", "k", "gcats", "gcatF", "gcur", "x_in"))
        dim = self.dim
        n = len(self.in_out)
        sync = getattr(self, "grad_sync", None)
        for s in range(n - 1, -1, -1):
            b1, b2, attn, down = self.downs[s]
            ci, co = self.in_out[s]
            s1, s2, s3, s4 = tape[s]
            gcat1, gcat2 = gcats[s]
            ghb = _chan(gcat1, co, co + ci)   # already holds the skip-connection gradient
            gha = _chan(gcat2, co, co + ci)
            if s == n - 1:
                down.bwd_s(gc, s4, gcur, ghb, True)
            else:
                down.bwd(gc, s4, gcur, ghb, True)
            gmid = ops.new(s3[0].shape, x_in)
            attn.bwd(gc, s3, ghb, gmid, False)
            b2.bwd(gc, s2, gmid, gsl[k], gha, True); k -= 1
            del gmid
            if s == 0:
                gr = _chan(gcatF, dim, 2 * dim)   # init_conv output also feeds the final concat
                b1.bwd(gc, s1, gha, gsl[k], gr, True); k -= 1
                gcur = gr
            else:
                gprev = ops.new(s1[0].shape, x_in)
                b1.bwd(gc, s1, gha, gsl[k], gprev, False); k -= 1
                gcur = gprev
        assert k == -1
        self.init_conv.bwd(gc, x_in, gcur, need_gx=False)
        self._flush(gc, 2)
        if sync is not None:
            sync.ready(self._head_end, self._ups_start)
        return st

    def backward_phase2b(self, st):
        """time embedding / FiLM projections.  After it [0, _head_end) is final."""
        gc = st["gc"]
        sync = getattr(self, "grad_sync", None)
        self._time_bwd(gc, st["time_saved"], st["gss_all"])
        self._flush(gc, 3)
        if sync is not None:
            sync.ready(0, self._head_end)
        self._flat.bind_grad_views()

    def forward(self, x: torch.Tensor, time: torch.Tensor, x_self_cond=None) -> torch.Tensor:
        """NCHW in / NCHW out, like the reference Unet.forward (:428-471)."""
        return _UnetFn.apply(self._anchor(x.device), self, x, time)

    def _anchor(self, device):
        self.prepare_hip(device)
        a = getattr(self, "_anchor_t", None)
        if a is None or a.device != torch.device(device):
            a = torch.zeros(1, device=device, requires_grad=True)
            self._anchor_t = a
        return a

    def run_nchw(self, x, time, save):
        B, C, H, W = x.shape
        xin = ops.new((B, H, W, _r4(C)), x)
        ops.nchw_to_nhwc(x.contiguous(), xin)
        out, tape = self.forward_nhwc(xin, time, save)
        y = ops.new((B, self.out_dim, H, W), x)
        ops.nhwc_to_nchw(out, y)
        return y, tape, out


class _UnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, net: Unet, x, time):
        save = bool(ctx.needs_input_grad[0])   # grad mode on at apply() time and anchor requires grad
        y, tape, _ = net.run_nchw(x.detach().float(), time, save)
        ctx.net, ctx.tape = net, tape
        return y

    @staticmethod
    def backward(ctx, gy):
        net, tape = ctx.net, ctx.tape
        if tape is None:
            raise RuntimeError("Unet forward ran without saving activations")
        B, C, H, W = gy.shape
        g = ops.new((B, H, W, _r4(C)), gy)
        ops.nchw_to_nhwc(gy.contiguous(), g)
        net.backward_nhwc(tape, g)
        ctx.tape = None
        return None, None, None, None


# ----------------------------------------------------------------------------------------
# Gaussian diffusion  (reference :532-946)
# ----------------------------------------------------------------------------------------
def _sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1):
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    v0 = torch.tensor(start / tau).sigmoid()
    v1 = torch.tensor(end / tau).sigmoid()
    ac = (v1 - ((t * (end - start) + start) / tau).sigmoid()) / (v1 - v0)
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


def _linear_beta_schedule(timesteps):
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def _cosine_beta_schedule(timesteps, s=0.008):
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


class GaussianDiffusion(nn.Module):
    def __init__(self, model: Unet, *, img_size, timesteps=1000, sampling_timesteps=None, objective="pred_v",
                 beta_schedule="sigmoid", schedule_fn_kwargs=None, ddim_sampling_eta=0.0, auto_normalize=True,
                 offset_noise_strength=0.0, min_snr_loss_weight=False, min_snr_gamma=5):
        super().__init__()
        if objective != "pred_v" or offset_noise_strength != 0.0:
            raise NotImplementedError("HIP path implements the configuration DDPM constructs (pred_v, no offset noise)")
        self.model = model
        self.channels = model.channels
        self.self_condition = False
        self.img_size = img_size
        self.objective = objective
        fn = {"linear": _linear_beta_schedule, "cosine": _cosine_beta_schedule, "sigmoid": _sigmoid_beta_schedule}
        if beta_schedule not in fn:
            raise ValueError(f"unknown beta schedule {beta_schedule}")
        betas = fn[beta_schedule](timesteps, **(schedule_fn_kwargs or {}))
        alphas = 1.0 - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = torch.nn.functional.pad(ac[:-1], (1, 0), value=1.0)
        self.num_timesteps = int(betas.shape[0])
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else self.num_timesteps
        assert self.sampling_timesteps <= self.num_timesteps
        self.is_ddim_sampling = self.sampling_timesteps < self.num_timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        snr = ac / (1 - ac)
        clipped = snr.clone()
        if min_snr_loss_weight:
            clipped.clamp_(max=min_snr_gamma)
        reg = lambda n, v: self.register_buffer(n, v.to(torch.float32))  # noqa: E731
        reg("betas", betas)
        reg("alphas_cumprod", ac)
        reg("alphas_cumprod_prev", ac_prev)
        reg("sqrt_alphas_cumprod", torch.sqrt(ac))
        reg("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - ac))
        reg("log_one_minus_alphas_cumprod", torch.log(1.0 - ac))
        reg("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / ac))
        reg("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / ac - 1))
        reg("posterior_variance", post_var)
        reg("posterior_log_variance_clipped", torch.log(post_var.clamp(min=1e-20)))
        reg("posterior_mean_coef1", betas * torch.sqrt(ac_prev) / (1.0 - ac))
        reg("posterior_mean_coef2", (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac))
        reg("loss_weight", clipped / (snr + 1))
        self.auto_normalize = auto_normalize

    @property
    def device(self):
        return self.betas.device

    def ddim_time_pairs(self):
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    # -- training ---------------------------------------------------------------------------
    def p_losses(self, x_start, t, noise=None, offset_noise_strength=None, _normalize=False):
        """x_start already normalised unless _normalize (reference :878-925)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        anchor = self.model._anchor(x_start.device)
        return _PLossFn.apply(anchor, self, x_start, t, noise, _normalize)

    def forward(self, img, *args, **kwargs):
        b, c, h, w = img.shape
        assert h == self.img_size and w == self.img_size, f"height and width of image must be {self.img_size}"
        t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        return self.p_losses(img, t, *args, _normalize=self.auto_normalize, **kwargs)

    # -- the reference's per-sample-timestep algebra (:673-705, 869-876): one lgm_extract_axpby launch each ---------
    def normalize(self, img):
        return img * 2 - 1 if self.auto_normalize else img

    def unnormalize(self, t):
        return (t + 1) * 0.5 if self.auto_normalize else t

    def _axpby(self, ta, tb, td, t, x, y, sb=1.0, clip=False):
        """out = clamp?((ta[t] * x + sb * tb[t] * y) / td[t]) on dense NCHW tensors, t per sample ([B] int64)."""
        x = x.detach().float().contiguous()
        B = x.shape[0]
        assert t.shape == (B,), f"expected one timestep per sample, got {tuple(t.shape)}"
        t = t.to(device=x.device, dtype=torch.long).contiguous()
        if y is not None:
            y = y.detach().float().contiguous()
            assert y.shape == x.shape
        out = torch.empty_like(x)
        if x.numel() == 0:
            return out
        ptr = lambda a: None if a is None else a.data_ptr()  # noqa: E731
        ops.lib().lgm_extract_axpby(ptr(ta), ptr(tb), ptr(td), t.data_ptr(), x.data_ptr(), ptr(y), float(sb),
                                    1 if clip else 0, out.data_ptr(), B, x.numel() // B, self.num_timesteps,
                                    ops.stream())
        return out

    def predict_start_from_noise(self, x_t, t, noise):
        return self._axpby(self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, None, t, x_t, noise, -1.0)

    def predict_noise_from_start(self, x_t, t, x0):
        return self._axpby(self.sqrt_recip_alphas_cumprod, None, self.sqrt_recipm1_alphas_cumprod, t, x_t, x0, -1.0)

    def predict_v(self, x_start, t, noise):
        return self._axpby(self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, None, t, noise, x_start, -1.0)

    def predict_start_from_v(self, x_t, t, v):
        return self._axpby(self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, None, t, x_t, v, -1.0)

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        return self._axpby(self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, None, t, x_start, noise, 1.0)

    def _extract(self, a, t, ndim):
        return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))

    def q_posterior(self, x_start, x_t, t):
        mean = self._axpby(self.posterior_mean_coef1, self.posterior_mean_coef2, None, t, x_start, x_t, 1.0)
        return (mean, self._extract(self.posterior_variance, t, x_t.ndim),
                self._extract(self.posterior_log_variance_clipped, t, x_t.ndim))

    @torch.no_grad()
    def model_predictions(self, x, t, x_self_cond=None, clip_x_start=False, rederive_pred_noise=False):
        """-> ModelPrediction(pred_noise, pred_x_start), reference :707-734 (pred_v branch: ``rederive_pred_noise`` has no
        effect there, the noise is always derived from the possibly clipped x_start).  UNet forward on the HIP engine, then
        ONE launch for both results."""
        assert x_self_cond is None, "the network DDPM constructs is not self-conditioned"
        v = self.model(x, t, x_self_cond)
        x = x.detach().float().contiguous()
        B = x.shape[0]
        t = t.to(device=x.device, dtype=torch.long).contiguous()
        pred_noise, x_start = torch.empty_like(x), torch.empty_like(x)
        ops.lib().lgm_model_predictions(x.data_ptr(), v.data_ptr(), t.data_ptr(), self.sqrt_alphas_cumprod.data_ptr(),
                                        self.sqrt_one_minus_alphas_cumprod.data_ptr(),
                                        self.sqrt_recip_alphas_cumprod.data_ptr(),
                                        self.sqrt_recipm1_alphas_cumprod.data_ptr(), 1 if clip_x_start else 0,
                                        pred_noise.data_ptr(), x_start.data_ptr(), B, x.numel() // B,
                                        self.num_timesteps, ops.stream())
        return ModelPrediction(pred_noise, x_start)

    @torch.no_grad()
    def p_mean_variance(self, x, t, x_self_cond=None, clip_denoised=True):
        x_start = self.model_predictions(x, t, x_self_cond, clip_x_start=clip_denoised).pred_x_start
        mean, var, logvar = self.q_posterior(x_start=x_start, x_t=x, t=t)
        return mean, var, logvar, x_start

    # -- sampling: lgm_hip/sampler.py (one fused update kernel per step, graph replay for whole chains) -------------
    @torch.no_grad()
    def p_sample(self, x, t: int, x_self_cond=None, noise=None):
        """One ancestral step at the shared timestep ``t`` -> (pred_img, x_start), reference :748-757.  ``noise``
        (extension, for parity tests): the draw the reference takes from randn_like."""
        from lgm_hip import sampler
        assert x_self_cond is None
        chain = sampler._Chain(self, tuple(x.shape), x)
        if noise is None and t > 0:
            noise = torch.randn_like(x)
        sampler.p_sample_step(chain, int(t), noise)
        x0 = torch.empty(tuple(x.shape), device=chain.x.device)
        ops.nhwc_to_nchw(chain.x0, x0)
        return chain.image(False), x0

    @torch.no_grad()
    def p_sample_loop(self, shape, return_all_timesteps=False):
        from lgm_hip import sampler
        return sampler.p_sample_loop(self, tuple(shape), return_all_timesteps)

    @torch.no_grad()
    def ddim_sample(self, shape, return_all_timesteps=False):
        from lgm_hip import sampler
        return sampler.ddim_sample(self, tuple(shape), return_all_timesteps)

    @torch.no_grad()
    def sample(self, batch_size=16, return_all_timesteps=False):
        fn = self.ddim_sample if self.is_ddim_sampling else self.p_sample_loop
        return fn((batch_size, self.channels, self.img_size, self.img_size), return_all_timesteps=return_all_timesteps)

    @torch.no_grad()
    def interpolate(self, x1, x2, t=None, lam=0.5):
        """reference :847-867: noise both images to step t, blend, walk the ancestral chain back to 0 (no unnormalise)."""
        from lgm_hip import sampler
        b = x1.shape[0]
        t = self.num_timesteps - 1 if t is None else int(t)
        assert x1.shape == x2.shape
        tb = torch.full((b,), t, device=x1.device, dtype=torch.long)
        xt1, xt2 = self.q_sample(x1, tb), self.q_sample(x2, tb)
        img = (1 - lam) * xt1 + lam * xt2
        return sampler.p_sample_loop(self, tuple(img.shape), init_noise=img, start=t, unnormalize=False)


def hip_loss_forward(gd: "GaussianDiffusion", img, t, noise, normalize: bool, save: bool):
    """q_sample + UNet + v-target + weighted MSE on the HIP engine.  Returns (loss[1], ctx)."""
    net = gd.model
    B, C, H, W = img.shape
    Cp = _r4(C)
    img = img.detach().float().contiguous()
    noise = noise.detach().float().contiguous()
    t = t.contiguous()
    xt = ops.new((B, H, W, Cp), img)
    target = ops.new((B, H, W, Cp), img)
    L = ops.lib()
    st = ops.stream()
    L.lgm_qsample_target(img.data_ptr(), noise.data_ptr(), t.data_ptr(), gd.sqrt_alphas_cumprod.data_ptr(),
                         gd.sqrt_one_minus_alphas_cumprod.data_ptr(), 1 if normalize else 0, xt.data_ptr(),
                         target.data_ptr(), Cp, B, C, H * W, Cp, st)
    out, tape = net.forward_nhwc(xt, t, save)
    per = ops.new((B,), img)
    loss = ops.new((1,), img)
    L.lgm_weighted_mse_fwd(out.data_ptr(), target.data_ptr(), Cp, t.data_ptr(), gd.loss_weight.data_ptr(),
                           B, C, H * W, Cp, per.data_ptr(), loss.data_ptr(), st)
    return loss, (gd, tape, out, target, t, (B, C, H, W), img, noise)


def hip_loss_backward_phase1(ctx, gl):
    """Loss gradient + first half of the UNet backward.  ``gl``: device scalar [1] = dL/dloss."""
    return ctx[0].model.backward_phase1b(hip_loss_backward_phase1a(ctx, gl))


def hip_loss_backward_phase1a(ctx, gl):
    """Loss gradient + backward of the final block and the up path (see Unet.backward_phase1a)."""
    gd, tape, out, target, t, (B, C, H, W) = ctx[:6]
    if tape is None:
        raise RuntimeError("p_losses forward ran without saving activations")
    Cp = _r4(C)
    gout = ops.new(out.shape, out)
    ops.lib().lgm_weighted_mse_bwd(out.data_ptr(), target.data_ptr(), Cp, t.data_ptr(),
                                   gd.loss_weight.data_ptr(), gl.data_ptr(), B, C, H * W, Cp,
                                   gout.data_ptr(), ops.stream())
    return gd.model.backward_phase1a(tape, gout)


class _PLossFn(torch.autograd.Function):
    """q_sample + UNet + v-target + weighted MSE, forward and hand-written backward."""

    @staticmethod
    def forward(ctx, anchor, gd: GaussianDiffusion, img, t, noise, normalize):
        loss, ctx.stuff = hip_loss_forward(gd, img, t, noise, normalize, bool(ctx.needs_input_grad[0]))
        return loss.view(())

    @staticmethod
    def backward(ctx, gloss):
        gl = gloss.detach().float().reshape(1).contiguous()
        st = hip_loss_backward_phase1(ctx.stuff, gl)
        ctx.stuff[0].model.backward_phase2(st)
        ctx.stuff = None
        return None, None, None, None, None, None


# ----------------------------------------------------------------------------------------
# LightningModule  (reference :949-1094)
# ----------------------------------------------------------------------------------------
class DDPM(LightningModule):
    def __init__(self, img_channels: int = 3, img_size: int = 64, dim: int = 64, diffusion_timesteps: int = 1000,
                 sampling_timesteps: Optional[int] = None, lr: float = 2e-5, betas: Tuple[float, float] = (0.9, 0.99),
                 ema_update_every: int = 10, ema_decay: float = 0.995):
        super().__init__()
        self.save_hyperparameters()
        model = Unet(dim=dim, channels=img_channels)
        diffusion_model = GaussianDiffusion(model, img_size=img_size, timesteps=diffusion_timesteps,
                                            sampling_timesteps=sampling_timesteps)
        self.channels = img_channels
        self.img_size = img_size
        self.ema = EMA(diffusion_model, beta=ema_decay, update_every=ema_update_every)
        self.sample_every = 1000          # reference: every 1000 steps on rank 0 (:1025)
        self.last_samples = None

    def prepare_hip(self, device):
        self.ema.online_model.model.prepare_hip(device)
        self.ema.ema_model.model.prepare_hip(device)

    def ddp_buffers(self):
        """Buffers training writes (re-broadcast from rank 0 before every forward, lgm_hip.lightning.BufferSync): none —
        the 13 schedule tables are constants and the EMA shadow is updated identically on every rank."""
        return []

    def _common_step(self, batch, mode: str):
        assert mode in ["train", "val", "test"], f"Invalid mode: {mode}"
        data, _ = batch
        model = self.ema.model if self.training else self.ema.ema_model
        loss = model(data)
        self.log(f"{mode}_loss", loss, prog_bar=True, logger=True, sync_dist=multi_rank())
        if self.sample_every and self.global_step % self.sample_every == 0 and _is_master():
            self._log_sample()
        return loss

    @torch.no_grad()
    def _log_sample(self):
        self.ema.ema_model.eval()
        self.last_samples = self.ema.ema_model.sample(batch_size=64)
        logger = getattr(self, "logger", None)
        if logger is not None and hasattr(logger, "experiment"):
            try:  # W&B is optional
                import wandb  # type: ignore
                logger.experiment.log({"Random Generation": [wandb.Image(self.last_samples)]}, step=self.global_step)
            except Exception:  # noqa
                pass

    def training_step(self, batch):
        return self._common_step(batch, "train")

    def on_train_batch_end(self, outputs, batch, batch_idx):
        self.ema.update()

    def validation_step(self, batch):
        return self._common_step(batch, "val")

    def configure_optimizers(self):
        return FusedAdam(self.ema.model.parameters(), lr=self.hparams.lr, betas=self.hparams.betas)

    def make_fast_step(self, opt, world: int = 1, use_graph: bool = True):
        """The step object ``MiniTrainer.fit`` drives instead of training_step/backward/step: overlapped
        bucketed gradient exchange + HIP-graph replay (what bench.py times), eager fallback inside."""
        from lgm_hip.graph import DDPMFastStep
        return DDPMFastStep(self, opt, world, use_graph)


def _is_master() -> bool:
    import torch.distributed as dist
    return (not dist.is_available()) or (not dist.is_initialized()) or dist.get_rank() == 0
