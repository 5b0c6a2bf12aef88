"""Leaf layers of the HIP engine: nn.Modules that own reference-shaped parameters (so
state_dicts interchange with the reference) and expose explicit ``fwd`` / ``bwd`` methods over
NHWC tensors.  Gradients of parameters are written straight into the flat gradient buffer
(see flat.py); no autograd graph is built inside the engine.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from . import ops
from .flat import FlatParams, _r4


def _flat(p: nn.Parameter) -> FlatParams:
    fp = getattr(p, "_lgm_flat", None)
    if fp is None:
        raise RuntimeError("parameter is not bound to flat HIP storage; call prepare() on the network first")
    return fp


class GradCtx:
    """Per-backward bookkeeping: beta for the first gradient write of every parameter
    (0 = overwrite, 1 = accumulate), 1 afterwards (a parameter used twice in one pass)."""

    def __init__(self, flat: FlatParams, transposed: bool = True, defer: bool = False):
        self.flat = flat
        self.beta0 = flat.begin_backward()
        self.written = set()
        # defer=True: conv weight-gradient slabs are reduced by ONE batched launch per flush() instead of
        # one launch per layer; the owner of the context MUST call flush() before the gradients are read
        self.deferred = [] if defer else None
        self._pending = []
        self._pending1 = []
        self._keep = []            # operands of generic weight-gradient launches waiting in the library's queue
        if defer:
            ops.lib().lgm_wgrad_queue_enable(-1)     # nothing an abandoned pass left queued may ever be launched
        if transposed:
            flat.refresh_transposed()      # one launch per backward pass

    def flush(self):
        self.finish_pending()
        if self.deferred:
            ops.wgrad_reduce_batch(self.deferred, self.flat.device)

    # -- large-map 3x3 weight gradients wait for partners: up to four layers share ONE launch (ops.conv_wgrad_group) ---
    def queue_wgrad(self, g, gy, x, gw_ptr: int, beta: float, gb_ptr):
        """Deferred passes only.  An entry keeps ``gy`` and ``x`` alive until its launch; the group is issued when it is
        full, when a layer arrives that cannot join it, or at the end of the bucket (finish_pending)."""
        new = (g, gy, x, gw_ptr, beta, gb_ptr)
        if self._pending and not ops.wgrad_group_supported([e[0] for e in self._pending] + [g]):
            self._finish_pending3x3()
        self._pending.append(new)
        if len(self._pending) == ops.WGRAD_GROUP:
            self._finish_pending3x3()

    def _finish_pending3x3(self):
        pend, self._pending = self._pending, []
        if len(pend) == 1:
            e = pend[0]
            ops.conv_wgrad(e[0], e[1], e[2], e[3], e[4], e[5], defer=self.deferred)
        elif pend:
            ops.conv_wgrad_group(pend, self.deferred)

    def finish_pending(self):
        """End of a bucket (or of the pass): every weight gradient still waiting for partners is issued."""
        self._finish_pending3x3()
        self.finish_pending1x1()
        if self._keep:
            ops.wgrad_queue_flush()
            self._keep.clear()

    # -- 1x1 weight gradients on the streaming kernel wait for partners too: up to four share ONE launch ----------------
    def queue_wgrad1x1(self, g, gy, x, gw_ptr: int, beta: float, gb_ptr):
        """Deferred passes only; same rules as queue_wgrad (ops.conv_wgrad1x1_group)."""
        new = (g, gy, x, gw_ptr, beta, gb_ptr)
        # one queue per block tile (a launch runs one instantiation of the kernel): a layer joins the queue whose group
        # takes it, and a queue is issued when it is full or at the end of the bucket - layers of another tile in between
        # do not break it up (they did at first: 14 launches became 11 instead of 5)
        for q in self._pending1:
            if ops.wgrad1x1_group_supported([e[0] for e in q] + [g]):
                q.append(new)
                if len(q) == 4:
                    self._pending1.remove(q)
                    ops.conv_wgrad1x1_group(q, self.deferred)
                return
        self._pending1.append([new])

    def finish_pending1x1(self):
        queues, self._pending1 = self._pending1, []
        for pend in queues:
            if len(pend) == 1:
                e = pend[0]
                ops.conv_wgrad(e[0], e[1], e[2], e[3], e[4], e[5], defer=self.deferred)
            else:
                ops.conv_wgrad1x1_group(pend, self.deferred)

    def defer_for(self, p: nn.Parameter):
        """The deferred-reduction list for the FIRST gradient contribution of ``p`` in this pass, else None: the
        per-parameter slab / row workspaces hold one contribution at a time, so a parameter used twice
        (a shared module) reduces its later contributions immediately.  Call before ``beta(p)``."""
        return self.deferred if id(p) not in self.written else None

    def beta(self, p: nn.Parameter) -> float:
        k = id(p)
        if k in self.written:
            return 1.0
        self.written.add(k)
        return self.beta0


class Conv2d(nn.Module):
    """nn.Conv2d replacement (NHWC, implicit-GEMM MFMA kernels).  Parameter shapes as torch."""

    transposed = False

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding = cin, cout, k, stride, padding
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cin * k * k)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._geoms = {}

    # X side = input (cin), Y side = output (cout)
    def geom(self, B, H, W):
        key = (B, H, W)
        g = self._geoms.get(key)
        if g is None:
            g = ops.make_geom(B, H, W, _r4(self.cin), _r4(self.cout), self.k, self.k, self.stride, self.padding)
            self._geoms[key] = g
        return g

    def out_shape(self, x):
        B, H, W, _ = x.shape
        g = self.geom(B, H, W)
        return (B, g.Ho, g.Wo, _r4(self.cout))

    def fwd_planes(self, x, groups: int):
        """Forward for a consumer that is a GroupNorm(groups): returns (y, planes).  planes = None: y is complete;
        else the convolution split its reduction and left its result as partial planes (ops.conv_xy partial=True) that
        the GroupNorm sums itself (GroupNorm.fwd(..., planes=planes) also writes the finished y)."""
        B, H, W, C = x.shape
        assert C == _r4(self.cin), f"conv expects {_r4(self.cin)} (padded) channels, got {C}"
        g = self.geom(B, H, W)
        fp = _flat(self.weight)
        y = ops.new((B, g.Ho, g.Wo, _r4(self.cout)), x)
        if self.k == 3 and self.stride == 1:
            # maps whose GroupNorm needs two passes (64 x 64): the statistics ride in the convolution's epilogue instead
            st = ops.conv_xy_stats(g, x, fp.ptr(self.weight), fp.ptr(self.bias) if self.bias is not None else None, y,
                                   groups)
            if st is not None:
                return y, st
        ok = self.k == 3 and ops.gn_planes_ok(B, g.Ho * g.Wo, _r4(self.cout), groups)
        pl = ops.conv_xy(g, x, fp.ptr(self.weight), fp.ptr(self.bias) if self.bias is not None else None, None, y,
                         partial=ok)
        return y, pl

    def fwd(self, x, out=None, res=None, stats=False, act=0, slope=0.0):
        """stats=True (bias-free convolution feeding a train-mode BatchNorm): returns (y, (partials, tiles)) with the
        batch statistics of y left behind by the convolution's epilogue (tiles == 0: not for this geometry).
        act (ReLU / LeakyReLU): y = act(conv(x) + bias + res), applied by the convolution's epilogue."""
        B, H, W, C = x.shape
        assert C == _r4(self.cin), f"conv expects {_r4(self.cin)} (padded) channels, got {C}"
        g = self.geom(B, H, W)
        fp = _flat(self.weight)
        y = out if out is not None else ops.new((B, g.Ho, g.Wo, _r4(self.cout)), x)
        if stats and self.bias is None and res is None:
            return y, ops.conv_stats(0, g, x, fp.ptr(self.weight), y)
        ops.conv_xy(g, x, fp.ptr(self.weight), fp.ptr(self.bias) if self.bias is not None else None, res, y,
                    post=ops.make_post(act, slope))
        return (y, (None, 0)) if stats else y

    def bwd(self, gc: GradCtx, x, gy, gx=None, accumulate=False, need_gx=True, res=None, planes_for_groups=0,
            mask=None, mask_slope=0.0, bn_sums=None):
        """gW, gb into flat grads; returns gx = dgrad(gy) (+ res) (+ existing gx when accumulate).
        ``mask`` (a saved ReLU / LeakyReLU output): gx is multiplied by that activation's derivative in the input
        gradient's epilogue - the backward of an activation that sat in front of this layer's input.
        ``planes_for_groups`` = G > 0: the only reader of gx is the backward of a GroupNorm(G) - returns (gx, planes) as
        ``fwd_planes`` does (planes given: gx itself is NOT written).
        ``bn_sums`` (ops.BnSums): gx is the gradient arriving at a train-mode BatchNorm - ask the input gradient's epilogue
        for that BatchNorm's backward sums (bn_sums.tiles > 0 afterwards when it delivered)."""
        B, H, W, _ = x.shape
        g = self.geom(B, H, W)
        fp = gc.flat
        dfr = gc.defer_for(self.weight)
        bw = gc.beta(self.weight)
        gb = None
        if self.bias is not None:
            bb = gc.beta(self.bias)
            if bb == bw:
                gb = fp.gptr(self.bias)              # bias gradient fused into the wgrad kernel
            else:
                ops.colsum(gy, fp.gptr(self.bias), bb)
        if need_gx and self.k == 3 and mask is None and bn_sums is None:
            # 3x3 layers: input gradient and weight gradient side by side in ONE launch (ops.conv_bwd_pair)
            if gx is None:
                gx = ops.new(x.shape, x)
                accumulate = False
            pres = gx if accumulate else res
            assert not (accumulate and res is not None)
            ok_pl = bool(planes_for_groups) and pres is None and ops.gn_planes_ok(B, H * W, x.shape[-1], planes_for_groups)
            r = ops.conv_bwd_pair(g, gy, x, fp.ptr(self.weight), fp.gptr(self.weight), bw, gb, dfr, pres, gx, partial=ok_pl)
            if r is not False:
                return (gx, r) if planes_for_groups else gx
        if (need_gx and self.k == 1 and mask is None and bn_sums is None and not planes_for_groups and dfr is not None
                and ops.wgrad1x1_queueable(g, gy, x)):
            # a 1x1 layer whose weight gradient runs the streaming kernel on a launch of its own: it waits for up to three
            # partners (GradCtx.queue_wgrad1x1: one launch for four); the input gradient as before
            gc.queue_wgrad1x1(g, gy, x, fp.gptr(self.weight), bw, gb)
            if gx is None:
                gx = ops.new(x.shape, x)
                accumulate = False
            assert not (accumulate and res is not None)
            ops.conv_yx(g, gy, fp.ptr(self.weight), None, gx if accumulate else res, gx, fp.tptr(self.weight))
            return gx
        if need_gx and (self.k != 3 or mask is not None or bn_sums is not None) and not planes_for_groups and not ops.B3:
            # the other layers (1x1, 4x4 / stride 2, 7x7; a 3x3 layer the pair above did not take keeps its Winograd /
            # direct input gradient below): both gradients through lgm_conv_bwd_pair - one launch when the kernels can
            # share a grid
            if gx is None:
                gx = ops.new(x.shape, x)
                accumulate = False
            assert not (accumulate and res is not None)
            if ops._weng_take(g, fp.ptr(self.weight), gx if accumulate else res, False) is not None:
                # 4x4 / stride-2 layer registered with the non-fused Winograd engine: its input gradient runs there, the
                # weight gradient keeps the implicit-GEMM kernel (the one-launch pair would bypass the engine)
                ops.conv_wgrad(g, gy, x, fp.gptr(self.weight), bw, gb, defer=dfr)
                ops.conv_yx(g, gy, fp.ptr(self.weight), None, None, gx, fp.tptr(self.weight),
                            post=ops.make_post(0, 0.0, mask, mask_slope, bn=bn_sums), post_mask=mask)
                return gx
            if dfr is not None:
                gc._keep.append((gy, x))       # a stand-alone weight-gradient launch may wait in the library's queue
            ops.conv_bwd_generic(g, gy, x, fp.ptr(self.weight), fp.tptr(self.weight), fp.gptr(self.weight), bw, gb, dfr,
                                 gx if accumulate else res, gx, post=ops.make_post(0, 0.0, mask, mask_slope, bn=bn_sums),
                                 post_mask=mask, queue=dfr is not None)
            return gx
        if dfr is not None and self.k == 3 and ops.wgrad_queueable(g, gy, x):
            gc.queue_wgrad(g, gy, x, fp.gptr(self.weight), bw, gb)      # issued with the next such layer's (one launch for two)
        else:
            if dfr is not None:
                gc._keep.append((gy, x))
            ops.conv_wgrad(g, gy, x, fp.gptr(self.weight), bw, gb, defer=dfr, queue=dfr is not None)
        if not need_gx:
            return None
        if gx is None:
            gx = ops.new(x.shape, x)
            accumulate = False
        if accumulate:
            assert res is None
            res = gx
        if planes_for_groups:
            ok = self.k == 3 and res is None and ops.gn_planes_ok(B, H * W, x.shape[-1], planes_for_groups)
            return gx, ops.conv_yx(g, gy, fp.ptr(self.weight), None, res, gx, fp.tptr(self.weight), partial=ok)
        ops.conv_yx(g, gy, fp.ptr(self.weight), None, res, gx, fp.tptr(self.weight),
                    post=ops.make_post(0, 0.0, mask, mask_slope, bn=bn_sums), post_mask=mask)
        return gx


def _conv_linear(self, x, out=None):
    """W * x without bias (the convolution as a linear map; used by second-order passes)."""
    B, H, W, _ = x.shape
    g = self.geom(B, H, W)
    fp = _flat(self.weight)
    y = out if out is not None else ops.new((B, g.Ho, g.Wo, _r4(self.cout)), x)
    ops.conv_xy(g, x, fp.ptr(self.weight), None, None, y)
    return y


def _conv_dgrad(self, gy, in_shape, gx=None, accumulate=False, mask=None, mask_slope=0.0, bn_sums=None):
    """W^T * gy (input gradient only, no parameter gradients).  ``mask``: as Conv2d.bwd - the derivative of the
    activation that produced this layer's input, applied in the epilogue.  ``bn_sums``: as Conv2d.bwd."""
    B, H, W, _ = in_shape
    g = self.geom(B, H, W)
    fp = _flat(self.weight)
    if gx is None:
        gx = ops.new(tuple(in_shape), gy)
        accumulate = False
    ops.conv_yx(g, gy, fp.ptr(self.weight), None, gx if accumulate else None, gx,
                post=ops.make_post(0, 0.0, mask, mask_slope, bn=bn_sums), post_mask=mask)
    return gx


def _conv_wgrad(self, gc: GradCtx, y, x):
    """gW (+)= sum y (x) x  for an arbitrary (y, x) pair of this layer's geometry."""
    B, H, W, _ = x.shape
    ops.conv_wgrad(self.geom(B, H, W), y, x, gc.flat.gptr(self.weight), gc.beta(self.weight), None)


Conv2d.linear = _conv_linear
Conv2d.dgrad = _conv_dgrad
Conv2d.wgrad = _conv_wgrad


class ConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d replacement: forward is the Y->X pass of the equivalent convolution."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding = cin, cout, k, stride, padding
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cout * k * k)  # torch: fan_in of the transposed weight = size(1)*k*k
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._geoms = {}

    # equivalent conv: X side = convT OUTPUT (cout channels), Y side = convT INPUT (cin channels)
    def geom(self, B, Hin, Win):
        key = (B, Hin, Win)
        g = self._geoms.get(key)
        if g is None:
            H = (Hin - 1) * self.stride - 2 * self.padding + self.k
            W = (Win - 1) * self.stride - 2 * self.padding + self.k
            g = ops.make_geom(B, H, W, _r4(self.cout), _r4(self.cin), self.k, self.k, self.stride, self.padding)
            assert g.Ho == Hin and g.Wo == Win
            self._geoms[key] = g
        return g

    def fwd(self, x, out=None, res=None, stats=False, act=0, slope=0.0):
        """stats=True: as Conv2d.fwd - (y, (partials, tiles)) with the batch statistics of y from the epilogue.
        act: as Conv2d.fwd."""
        B, H, W, C = x.shape
        assert C == _r4(self.cin)
        g = self.geom(B, H, W)
        fp = _flat(self.weight)
        y = out if out is not None else ops.new((B, g.H, g.W, _r4(self.cout)), x)
        if stats and self.bias is None and res is None:
            return y, ops.conv_stats(1, g, x, fp.ptr(self.weight), y)
        ops.conv_yx(g, x, fp.ptr(self.weight), fp.ptr(self.bias) if self.bias is not None else None, res, y,
                    post=ops.make_post(act, slope))
        return (y, (None, 0)) if stats else y

    def bwd(self, gc: GradCtx, x, gy, gx=None, accumulate=False, need_gx=True, res=None, mask=None, mask_slope=0.0,
            bn_sums=None):
        B, H, W, _ = x.shape
        g = self.geom(B, H, W)
        fp = gc.flat
        dfr = gc.defer_for(self.weight)          # slab reduction batched with the other layers' when the pass defers
        ops.conv_wgrad(g, x, gy, fp.gptr(self.weight), gc.beta(self.weight), defer=dfr)  # Y side = input, X side = grad
        if self.bias is not None:
            dfb = gc.defer_for(self.bias)
            ops.colsum(gy, fp.gptr(self.bias), gc.beta(self.bias), defer=dfb)
        if not need_gx:
            return None
        if gx is None:
            gx = ops.new(x.shape, x)
            accumulate = False
        if accumulate:
            assert res is None
            res = gx
        ops.conv_xy(g, gy, fp.ptr(self.weight), None, res, gx, post=ops.make_post(0, 0.0, mask, mask_slope, bn=bn_sums),
                    post_mask=mask)
        return gx


class Linear(nn.Module):
    """nn.Linear replacement = 1x1 convolution over a [B, 1, 1, C] "image"."""

    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(cin)
        if bias:
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._geoms = {}

    def geom(self, B):
        g = self._geoms.get(B)
        if g is None:
            g = ops.make_geom(B, 1, 1, _r4(self.cin), _r4(self.cout), 1, 1, 1, 0)
            self._geoms[B] = g
        return g


def linear_fwd(g, x2d, w_ptr, b_ptr, y2d):
    ops.conv_xy(g, x2d, w_ptr, b_ptr, None, y2d)


class GroupNorm(nn.Module):
    """Parameters of nn.GroupNorm (weight, bias); the fused GN+FiLM+SiLU kernels are driven by
    the owning block."""

    def __init__(self, groups, channels, eps=1e-5):
        super().__init__()
        self.groups, self.channels, self.eps = groups, channels, eps
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))

    def fwd(self, x, ss, act, res, out=None, planes=None):
        """``planes`` (from Conv2d.fwd_planes): x arrives as split-K partial planes; summed, written to x and
        normalised in one pass."""
        fp = _flat(self.weight)
        y = out if out is not None else ops.new(x.shape, x)
        sv = ops.gn_fwd(x, self.groups, self.eps, fp.ptr(self.weight), fp.ptr(self.bias), ss, act, res, y, planes=planes)
        return y, sv

    def bwd(self, gc: GradCtx, x, gy, ss, act, sv, gss, gx=None, accumulate=False, gy_planes=None, add_gy_to=None):
        """``add_gy_to``: a tensor that also receives ``+= gy`` in this launch (ops.gn_bwd)."""
        fp = gc.flat
        if gx is None:
            gx = ops.new(x.shape, x)
            accumulate = False
        dfr = gc.defer_for(self.weight)
        ops.gn_bwd(x, gy, self.groups, fp.ptr(self.weight), fp.ptr(self.bias), ss, act, sv, gx, accumulate,
                   fp.gptr(self.weight), fp.gptr(self.bias), gc.beta(self.weight), gss, 0.0, defer=dfr,
                   gy_planes=gy_planes, add_gy_to=add_gy_to)
        gc.beta(self.bias)
        return gx


class RMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))

    def fwd(self, x, res=None, out=None):
        y = out if out is not None else ops.new(x.shape, x)
        ops.rmsnorm_fwd(x, _flat(self.g).ptr(self.g), res, y)
        return y

    def bwd(self, gc: GradCtx, x, gy, gx=None, accumulate=False, res=None):
        if gx is None:
            gx = ops.new(x.shape, x)
            accumulate = False
        dfr = gc.defer_for(self.g)
        ops.rmsnorm_bwd(x, gy, gc.flat.ptr(self.g), gx, accumulate, gc.flat.gptr(self.g), gc.beta(self.g),
                        defer=dfr, res=res)
        return gx


def param_kind(name: str, p: nn.Parameter) -> str:
    if name.endswith("weight") and p.dim() in (2, 4):
        return "weight"
    return "vector"
