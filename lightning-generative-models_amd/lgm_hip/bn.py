"""Train-mode BatchNorm2d on the HIP primitives (lgm_bn_*), including the adjoint pieces the
WGAN-GP double backward needs.  Notation (per channel, M = B*H*W rows):
    xhat = (a - mean) * rstd,   n = gamma*xhat + beta,   c = gamma*rstd
    T(v) = c * (v - mean(v) - xhat * mean(v*xhat))        (BatchNorm's backward operator)
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .nn import GradCtx, _flat


class BNSaved:
    __slots__ = ("a", "mean", "rstd")

    def __init__(self, a, mean, rstd):
        self.a, self.mean, self.rstd = a, mean, rstd


def _ws(a):
    rows, C = ops.rows(a), a.shape[-1]
    return ops.workspace(ops.lib().lgm_bn_workspace(rows, C), a.device)


# While a training step is being captured into a HIP graph (lgm_hip/graph.py) every train-mode forward appends its
# module here: a replay runs those forwards again without passing through Python, so the step object advances the
# host-side batch counters of exactly these modules after each replay.
CAPTURE_TRACE = None


def _count_forward(module):
    module._nbt_pending += 1
    if CAPTURE_TRACE is not None:
        CAPTURE_TRACE.append(module)


class BatchNorm2d(nn.Module):
    """Parameters/buffers named as nn.BatchNorm2d (weight, bias, running_mean, running_var,
    num_batches_tracked).  Only train-mode (batch statistics) is on the hot path; eval mode uses
    the running statistics through the same affine kernel."""

    def __init__(self, channels, eps=1e-5, momentum=0.1):
        super().__init__()
        self.channels, self.eps, self.momentum = channels, eps, momentum
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        # the counter is bookkeeping only (momentum is fixed): it is advanced on the host and written into the
        # buffer when a state dict is taken - `buffer += 1` was one 1-thread launch per BatchNorm forward
        # (13 per critic step)
        self._nbt_pending = 0
        self._register_state_dict_hook(BatchNorm2d._flush_counter_hook)
        self._register_load_state_dict_pre_hook(self._reset_counter_hook)

    @staticmethod
    def _flush_counter_hook(module, state_dict, prefix, local_metadata):
        if module._nbt_pending:
            module.num_batches_tracked += module._nbt_pending
            module._nbt_pending = 0
            state_dict[prefix + "num_batches_tracked"] = module.num_batches_tracked.detach()

    def _reset_counter_hook(self, *args, **kwargs):
        self._nbt_pending = 0

    # ---- forward: h = act(gamma*xhat + beta) ------------------------------------------------
    def fwd(self, a, act: int, slope: float = 0.0, training: bool = True, stats=None):
        """stats = (partials, tiles) from the producing convolution's epilogue (Conv2d.fwd(stats=True)): the
        statistics are finished from them and the read pass over ``a`` is skipped."""
        C = a.shape[-1]
        fp = _flat(self.weight)
        st = torch.empty((2, C), dtype=torch.float32, device=a.device)
        mean, rstd = st[0], st[1]
        L = ops.lib()
        if training and stats is not None and stats[1] > 0:
            L.lgm_bn_stats_from_tiles(stats[0].data_ptr(), stats[1], C, ops.rows(a), self.eps, self.momentum,
                                      mean.data_ptr(), rstd.data_ptr(), self.running_mean.data_ptr(),
                                      self.running_var.data_ptr(), ops.stream())
            _count_forward(self)
        elif training:
            L.lgm_bn_stats(a.data_ptr(), ops.pitch(a), ops.rows(a), C, self.eps, self.momentum, mean.data_ptr(),
                           rstd.data_ptr(), self.running_mean.data_ptr(), self.running_var.data_ptr(),
                           _ws(a).data_ptr(), ops.stream())
            _count_forward(self)
        else:
            mean.copy_(self.running_mean)
            rstd.copy_((self.running_var + self.eps).rsqrt())
        h = ops.new(a.shape, a)
        L.lgm_bn_affine3(None, 0, None, 0, a.data_ptr(), ops.pitch(a), mean.data_ptr(), rstd.data_ptr(), None, None,
                         fp.ptr(self.weight), fp.ptr(self.bias), h.data_ptr(), ops.pitch(h), 0, act, slope,
                         ops.rows(a), C, ops.stream())
        return h, BNSaved(a, mean, rstd)

    # ---- T(v), optionally with the parameter gradients of the forward node -------------------
    def sums_request(self, sv: BNSaved):
        """-> ops.BnSums for the convolution that is about to produce the gradient arriving at this BatchNorm (pass it as
        ``bn_sums=`` to that layer's bwd / dgrad, then as ``sums=`` to apply_T)."""
        return ops.BnSums(sv.a, sv.mean, sv.rstd)

    def apply_T(self, sv: BNSaved, v, gc: GradCtx = None, want_m: bool = False, out=None, accumulate=False, sums=None):
        """returns (T(v), mvec) where mvec = [mean v, mean v*xhat] (when want_m).
        ``sums`` (ops.BnSums with tiles > 0): the convolution that wrote ``v`` left the per-tile sums (sum v, sum v * xhat)
        behind - the reduction pass over (v, a) is skipped, the second stage runs on them."""
        a = sv.a
        C = a.shape[-1]
        rows = ops.rows(a)
        L = ops.lib()
        fp = _flat(self.weight)
        coef = torch.empty((4, C), dtype=torch.float32, device=a.device)
        mvec = torch.empty((2, C), dtype=torch.float32, device=a.device) if want_m else None
        gg = gb = None
        beta = 0.0
        if gc is not None:
            gg, gb = gc.flat.gptr(self.weight), gc.flat.gptr(self.bias)
            beta = gc.beta(self.weight)
            gc.beta(self.bias)
        if sums is not None and sums.tiles > 0:
            assert sums.a is sv.a
            L.lgm_bn_reduce3_coef_tiles(1, sums.partial.data_ptr(), sums.tiles, fp.ptr(self.weight), sv.rstd.data_ptr(),
                                        None, rows, C, coef.data_ptr(), gg, gb, beta,
                                        None if mvec is None else mvec.data_ptr(), None, 0.0, None, ops.stream())
        else:
            # sums (v, v*xhat) and the coefficients of T in two launches (the per-channel math runs in stage 2)
            L.lgm_bn_reduce3_coef(1, v.data_ptr(), ops.pitch(v), None, 0, a.data_ptr(), ops.pitch(a), sv.mean.data_ptr(),
                                  sv.rstd.data_ptr(), fp.ptr(self.weight), None, rows, C, coef.data_ptr(), gg, gb, beta,
                                  None if mvec is None else mvec.data_ptr(), None, 0.0, None, _ws(a).data_ptr(),
                                  ops.stream())
        if out is None:
            out = ops.new(a.shape, a)
            accumulate = False
        L.lgm_bn_affine3(v.data_ptr(), ops.pitch(v), None, 0, a.data_ptr(), ops.pitch(a), sv.mean.data_ptr(),
                         sv.rstd.data_ptr(), coef[0].data_ptr(), None, coef[2].data_ptr(), coef[3].data_ptr(),
                         out.data_ptr(), ops.pitch(out), 1 if accumulate else 0, 0, 0.0, rows, C, ops.stream())
        return out, mvec

    # ---- second-order pieces (gradient penalty) ----------------------------------------------
    def adjoint_T(self, sv: BNSaved, u, gn, mvec, gc: GradCtx):
        """Node ga = T(gn) of the first backward pass received the adjoint ``u`` (w.r.t. ga).
        Returns (adjoint w.r.t. gn, adjoint w.r.t. the forward activation a through the batch
        statistics) and accumulates the adjoint w.r.t. gamma."""
        a = sv.a
        C = a.shape[-1]
        rows = ops.rows(a)
        L = ops.lib()
        fp = _flat(self.weight)
        # (1) T is self-adjoint in its argument: adjoint w.r.t. gn = T(u)            (coefficient set 0)
        # (2) dependence of T on xhat / sigma (hence on a) and on gamma                (coefficient set 1)
        # one reduction for the sums of u both sets need, one pass over (u, gn, a) for both results
        coef = torch.empty((2, 4, C), dtype=torch.float32, device=a.device)
        L.lgm_bn_reduce3_coef(3, u.data_ptr(), ops.pitch(u), gn.data_ptr(), ops.pitch(gn), a.data_ptr(), ops.pitch(a),
                              sv.mean.data_ptr(), sv.rstd.data_ptr(), fp.ptr(self.weight), mvec.data_ptr(), rows, C,
                              coef.data_ptr(), None, None, 0.0, None, gc.flat.gptr(self.weight),
                              gc.beta(self.weight), None, _ws(a).data_ptr(), ops.stream())
        gn_bar = ops.new(a.shape, a)
        a_bar = ops.new(a.shape, a)
        L.lgm_bn_affine3x2(u.data_ptr(), ops.pitch(u), gn.data_ptr(), ops.pitch(gn), a.data_ptr(), ops.pitch(a),
                           sv.mean.data_ptr(), sv.rstd.data_ptr(), coef.data_ptr(), gn_bar.data_ptr(),
                           ops.pitch(gn_bar), a_bar.data_ptr(), ops.pitch(a_bar), rows, C, ops.stream())
        return gn_bar, a_bar
