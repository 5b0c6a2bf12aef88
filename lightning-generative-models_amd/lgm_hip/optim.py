"""Fused optimiser + EMA on flat HIP storage.

FusedAdam: torch.optim.Adam semantics (coupled L2 ``weight_decay``; reference call sites
ddpm.py:1053-1059, vqvae.py:207-214, wgan.py:183-195) as ONE streaming kernel per flat
buffer.  EMA: ema_pytorch.EMA surface used by ddpm.py:998,1014,1033,1048 (``.model``,
``.ema_model``, ``.update()``; state_dict keys ``online_model.*``, ``ema_model.*``,
``initted``, ``step``) — upstream is unpinned, the schedule follows its published algorithm
(see oracle/optim.py).
"""
from __future__ import annotations

import copy
from typing import Iterable, List

import torch
from torch import nn

from . import ops
from .flat import FlatParams


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, decoupled=False):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, decoupled=decoupled)
        super().__init__(params, defaults)
        self._flat_state = {}   # id(FlatParams) -> dict(m, v, step, flat)
        self.grad_scale = 1.0   # e.g. 1/world_size after a SUM all-reduce (folded into the kernel)

    def _flats(self, group) -> List[FlatParams]:
        seen, out = set(), []
        for p in group["params"]:
            fp = getattr(p, "_lgm_flat", None)
            if fp is None:
                raise RuntimeError("FusedAdam: parameter is not bound to flat HIP storage "
                                   "(call prepare_hip() on the model before configure_optimizers/step)")
            if id(fp) not in seen:
                seen.add(id(fp))
                out.append(fp)
        return out

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for fp in self._flats(group):
                st = self._flat_state.get(id(fp))
                if st is None:
                    st = dict(m=torch.zeros_like(fp.data), v=torch.zeros_like(fp.data), step=0, flat=fp)
                    self._flat_state[id(fp)] = st
                st["step"] += 1
                ops.adam_step(fp.data, fp.grad, st["m"], st["v"], fp.total, group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], st["step"], None, self.grad_scale, group["decoupled"])
        return loss

    # ---- one step issued slice by slice (lgm_hip.graph: a bucket's slice as soon as ITS gradients are final) --------
    def begin_step(self, fp: FlatParams):
        """Advance the step count of ``fp``'s state once; returns (group, state) for ``step_slice``."""
        for group in self.param_groups:
            if any(f is fp for f in self._flats(group)):
                st = self._flat_state.get(id(fp))
                if st is None:
                    st = dict(m=torch.zeros_like(fp.data), v=torch.zeros_like(fp.data), step=0, flat=fp)
                    self._flat_state[id(fp)] = st
                st["step"] += 1
                return group, st
        raise RuntimeError("begin_step: this optimizer does not own the flat buffer")

    @torch.no_grad()
    def step_slice(self, fp: FlatParams, group, st, lo: int, hi: int):
        """The Adam update of elements [lo, hi) (multiples of 4) - same kernel, same arithmetic as step()."""
        assert lo % 4 == 0 and hi % 4 == 0 and 0 <= lo < hi <= fp.total
        b1, b2 = group["betas"]
        ops.adam_step(fp.data[lo:hi], fp.grad[lo:hi], st["m"][lo:hi], st["v"][lo:hi], hi - lo, group["lr"], b1, b2,
                      group["eps"], group["weight_decay"], st["step"], None, self.grad_scale, group["decoupled"])

    def zero_grad(self, set_to_none: bool = True):
        for group in self.param_groups:
            for fp in self._flats(group):
                fp.zero_grad()

    # ---- checkpoint interchange: torch.optim.Adam's own state_dict layout -------------------------
    # (Lightning stores ``optimizer.state_dict()`` under "optimizer_states"; parameter order is the
    # registration order, which this package keeps identical to the reference's modules.)
    _STATE_KEYS = (("m", "exp_avg"), ("v", "exp_avg_sq"))

    def state_dict(self):
        from .flat import logical_view
        state, groups, idx = {}, [], 0
        for group in self.param_groups:
            ids = []
            for p in group["params"]:
                fp = p._lgm_flat
                st = self._flat_state.get(id(fp))
                if st is not None and st["step"] > 0:
                    sl = fp.slot(p)
                    ent = {"step": torch.tensor(float(st["step"]))}
                    for mine, theirs in self._STATE_KEYS:
                        ent[theirs] = logical_view(st[mine][sl.offset:sl.offset + sl.numel], p.shape, sl.kind,
                                                   sl.phys_shape).detach().clone().contiguous()
                        if sl.ref_shape is not None:          # the reference's shape (same row-major order)
                            ent[theirs] = ent[theirs].reshape(sl.ref_shape)
                    state[idx] = ent
                ids.append(idx)
                idx += 1
            g = {k: v for k, v in group.items() if k != "params"}
            g["params"] = ids
            groups.append(g)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        from .flat import logical_view
        idx = 0
        for group, saved in zip(self.param_groups, sd["param_groups"]):
            for k, v in saved.items():
                if k != "params" and k in group:
                    group[k] = tuple(v) if isinstance(group[k], tuple) else v
            for p in group["params"]:
                ent = sd["state"].get(idx, sd["state"].get(str(idx)))
                idx += 1
                if ent is None:
                    continue
                fp = p._lgm_flat
                st = self._flat_state.get(id(fp))
                if st is None:
                    st = {mine: torch.zeros_like(fp.data) for mine, _ in self._STATE_KEYS}
                    st.update(step=0, flat=fp)
                    self._flat_state[id(fp)] = st
                sl = fp.slot(p)
                for mine, theirs in self._STATE_KEYS:
                    logical_view(st[mine][sl.offset:sl.offset + sl.numel], p.shape, sl.kind, sl.phys_shape).copy_(
                        ent[theirs].to(fp.device, torch.float32).reshape(p.shape))
                st["step"] = int(float(ent["step"]))


class FusedRMSprop(FusedAdam):
    """torch.optim.RMSprop with its defaults (alpha 0.99, eps 1e-8, momentum 0, not centred) — the
    optimiser of the weight-clipping WGAN (reference wgan.py:171-181) — one kernel per flat buffer."""

    _STATE_KEYS = (("sq", "square_avg"),)

    def __init__(self, params: Iterable[nn.Parameter], lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        torch.optim.Optimizer.__init__(self, params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay,
                                                          momentum=0, centered=False))
        self._flat_state = {}
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for fp in self._flats(group):
                st = self._flat_state.get(id(fp))
                if st is None:
                    st = dict(sq=torch.zeros_like(fp.data), step=0, flat=fp)
                    self._flat_state[id(fp)] = st
                st["step"] += 1
                ops.rmsprop_step(fp.data, fp.grad, st["sq"], fp.total, group["lr"], group["alpha"], group["eps"],
                                 group["weight_decay"], self.grad_scale)
        return loss


class EMA(nn.Module):
    """Shadow copy of a network updated every ``update_every`` calls of update()."""

    def __init__(self, model: nn.Module, beta=0.9999, update_every=10, update_after_step=100,
                 inv_gamma=1.0, power=2.0 / 3.0, min_value=0.0):
        super().__init__()
        self.online_model = model
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.beta, self.update_every, self.update_after_step = beta, update_every, update_after_step
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("step", torch.tensor(0))
        self._step_py = 0          # host mirrors: no device sync on the hot path
        self._initted_py = False
        # a checkpoint restores the ``step`` / ``initted`` buffers: re-sync the host mirrors from them
        self.register_load_state_dict_post_hook(EMA._sync_host_state)

    @staticmethod
    def _sync_host_state(module, incompatible_keys):
        module._step_py = int(module.step.item())
        module._initted_py = bool(module.initted.item())

    @property
    def model(self):
        return self.online_model

    def forward(self, *a, **k):
        return self.ema_model(*a, **k)

    def current_decay(self) -> float:
        epoch = max(self._step_py - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        value = 1 - (1 + epoch / self.inv_gamma) ** (-self.power)
        return min(max(value, self.min_value), self.beta)

    def _pairs(self):
        on = dict(self.online_model.named_parameters())
        on.update(dict(self.online_model.named_buffers()))
        sh = dict(self.ema_model.named_parameters())
        sh.update(dict(self.ema_model.named_buffers()))
        for k, v in sh.items():
            yield v, on[k]

    @torch.no_grad()
    def _lerp_all(self, w: float):
        fo = _first_flat(self.online_model)
        fs = _first_flat(self.ema_model)
        if fo is not None and fs is not None and fo.total == fs.total:
            ops.ema_lerp(fs.data, fo.data, w)          # one kernel over the flat storage
            done = {id(s.param) for s in fs.slots}
        else:
            done = set()
        for sh, on in self._pairs():
            if id(sh) in done or not sh.dtype.is_floating_point:
                if id(sh) not in done and w == 1.0:
                    sh.copy_(on)
                continue
            if w == 1.0:
                sh.copy_(on)
            else:
                sh.lerp_(on, w)

    def update(self):
        step = self._step_py
        self._step_py += 1
        self.step += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step:
            self._lerp_all(1.0)
            return
        if not self._initted_py:
            self._lerp_all(1.0)
            self._initted_py = True
            self.initted.fill_(True)
        self._lerp_all(1.0 - self.current_decay())


def _first_flat(module: nn.Module):
    for p in module.parameters():
        return getattr(p, "_lgm_flat", None)
    return None
