"""Flat parameter / gradient storage for one network.

All parameters of a network live in ONE fp32 device buffer (and their gradients in a second
one) so that the optimiser is a single streaming kernel, DDP all-reduces one contiguous
tensor over RCCL, and the EMA shadow update is one lerp.  Each ``nn.Parameter`` keeps its
reference shape (state_dict compatible) but is re-pointed at a strided *view* of the flat
buffer whose physical layout is what the HIP kernels read directly:

  conv / linear weight  logical [N, C, KH, KW]   physical [Np][KH*KW][Cp]  (Np, Cp = N, C rounded up to 4;
                                                  = torch channels_last with zero padding lanes)
  vectors (bias, norm)  padded to a multiple of 4 floats

Padding lanes are zero and stay zero under Adam (zero grad -> zero update).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import weakref

import torch
from torch import nn

# every FlatParams alive in this process (weak): ops.forget_dead_flats() keeps the workspaces of the live ones
ALL_FLATS: "weakref.WeakSet" = weakref.WeakSet()


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


class ParamSlot:
    __slots__ = ("param", "name", "kind", "offset", "numel", "phys_shape", "ref_shape")

    def __init__(self, param, name, kind, offset, numel, phys_shape):
        self.param, self.name, self.kind = param, name, kind
        self.offset, self.numel, self.phys_shape = offset, numel, phys_shape
        # shape under which checkpoints exchange this parameter when it differs from the parameter's own (a layer held in
        # the form its kernel wants whose row-major flattening is the reference's tensor: ddpm._DownConv); None: the same
        self.ref_shape = None


def phys_numel(p: torch.Tensor, kind: str) -> Tuple[int, Tuple[int, ...]]:
    if kind == "weight":           # conv [N,C,KH,KW] / convT [Cin,Cout,KH,KW] / linear [N,C]
        if p.dim() == 4:
            N, C, KH, KW = p.shape
        else:
            (N, C), KH, KW = p.shape, 1, 1
        shp = (_r4(N), KH * KW, _r4(C))
        return shp[0] * shp[1] * shp[2], shp
    n = _r4(p.numel())
    return n, (n,)


def logical_view(buf: torch.Tensor, p_shape: Sequence[int], kind: str, phys_shape) -> torch.Tensor:
    if kind == "weight":
        Np, T, Cp = phys_shape
        if len(p_shape) == 4:
            N, C, KH, KW = p_shape
            return buf.view(Np, KH, KW, Cp)[:N, :, :, :C].permute(0, 3, 1, 2)
        N, C = p_shape
        return buf.view(Np, Cp)[:N, :C]
    n = 1
    for s in p_shape:
        n *= s
    return buf[:n].view(*p_shape)


class FlatParams:
    """Owns flat fp32 storage for ``named`` = [(name, param, kind)].  ``order`` is preserved, so
    callers can make groups of parameters adjacent (e.g. the 19 time-MLP weights of the UNet,
    which are then one [sum(2C), 256] GEMM operand)."""

    def __init__(self, named: List[Tuple[str, nn.Parameter, str]], device):
        self.device = torch.device(device)
        self.slots: List[ParamSlot] = []
        off = 0
        for name, p, kind in named:
            n, shp = phys_numel(p, kind)
            self.slots.append(ParamSlot(p, name, kind, off, n, shp))
            off += n
        self.total = off
        self.data = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        ALL_FLATS.add(self)
        self.by_param: Dict[int, ParamSlot] = {}
        self.fresh = True        # next backward overwrites (beta = 0) instead of accumulating
        self.written = False     # a backward pass wrote this buffer since the flag was last cleared
        for s in self.slots:
            view = logical_view(self.data[s.offset:s.offset + s.numel], s.param.shape, s.kind, s.phys_shape)
            with torch.no_grad():
                view.copy_(s.param.data.to(self.device, torch.float32))
            s.param.data = view
            s.param.grad = None
            s.param._lgm_flat = self           # noqa: back-reference used by FusedAdam / EMA
            self.by_param[id(s.param)] = s
        self._grad_views_bound = False
        self.data_t = None          # transposed copies of the weight slots (see refresh_transposed)
        self._t_table = None
        self._t_skip = set()
        self._wino_off = {}
        self._w4 = {}               # slot offset -> (Uf, Ub, Np, Cp) | False: F(4x4) operands, registered on first use
        self._w4_table = None
        self._w4_table_old = None
        # opt-in split-precision convolutions (LGM_CONV_MODE=bf16x3): three bf16 planes of the weights
        # (forward layout) and of their transposed copies, in MFMA fragment order
        self.wino = False           # Winograd-transformed 3x3 weights (see enable_wino)
        self.data_uf = None
        self.data_ub = None
        self.b3 = False
        self.planes = None
        self.planes_t = None
        self._b3_tables = None

    # -- transposed weights (input-gradient kernels read [Cw][T][Nw]) ---------------------------
    def refresh_transposed(self):
        """One launch: rewrite every conv/linear weight slot as [Cw][T][Nw] into ``data_t`` (same offsets)."""
        from . import ops
        if self._t_table is None:
            rows, blk = [], 0
            # slots whose input gradient runs on the Winograd operand (data_ub) need no transposed copy: 33 M of the
            # UNet's 35.7 M parameters.  ``tptr`` returns None for them, so a geometry the Winograd kernel does not
            # take falls back to the direct kernel's untransposed-weight mode instead of reading a stale copy.
            self._t_skip = set(self._wino_off) if (self.wino and ops.WINO and not self.b3) else set()
            for s in self.slots:
                if s.kind != "weight" or s.offset in self._t_skip:
                    continue
                Np, T, Cp = s.phys_shape
                rows.append([s.offset, Np, T, Cp, blk])
                blk += ((Np + 31) // 32) * ((Cp + 31) // 32) * T
            if not rows:
                return
            assert self.total < 2 ** 31
            self._t_table = torch.tensor(rows, dtype=torch.int32, device=self.device).contiguous()
            self._t_blocks = blk
            self.data_t = torch.zeros_like(self.data)
        ops.lib().lgm_transpose_weights(self.data.data_ptr(), self.data_t.data_ptr(), self._t_table.data_ptr(),
                                        self._t_table.shape[0], self._t_blocks, ops.stream())
        if self.b3:
            self._refresh_split_t()

    # -- Winograd-transformed 3x3 weights (csrc/winograd.hip) -----------------------------------
    def enable_wino(self):
        """Allocate the transformed copies U = G g G^T of every 3x3 weight slot whose (padded) channel counts
        are multiples of 32: forward operand and input-gradient operand, 16/9 of the slot each.
        ``refresh_wino`` rewrites them from the current weights in ONE launch."""
        from . import ops
        if self.wino:
            return
        off = 0
        self._wino_off = {}
        self._wino_dims = {}
        for s in self.slots:
            if s.kind != "weight":
                continue
            Np, T, Cp = s.phys_shape
            if T == 9 and Np % 32 == 0 and Cp % 32 == 0:
                self._wino_off[s.offset] = off
                self._wino_dims[s.offset] = (Np, Cp)
                off += Np * Cp * 16
        if not self._wino_off:
            return
        # the copies of ALL slots are allocated once (addresses baked into captured graphs never move); which of them
        # the refresh launch rewrites is decided by use (wino_u): a layer that runs on the F(4x4) kernel in both
        # directions never asks, and its 2 x 16/9 of the weights are not rewritten every step
        self._wino_used = {}
        self._wino_table = None
        self._wino_table_old = None
        self.data_uf = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.data_ub = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.wino = True
        ops.register_wino_flat(self)

    def refresh_wino(self, backward_operand: bool = True):
        """Transformed copies of the CURRENT weights (call once per forward pass: the optimizer moved them)."""
        from . import ops
        if not self.wino:
            return
        self._refresh_wino2(backward_operand)
        self._refresh_wino4(backward_operand)

    def wino_u(self, slot_offset: int, backward: bool):
        """Address of the F(2x2) operand (forward or input-gradient copy) of the 3x3 slot at ``slot_offset``, or None.
        The first request of a (slot, direction) adds it to the refresh launch's table and transforms the current
        weights at once; like ``wino4_u`` that is refused while a capture is running."""
        uo = self._wino_off.get(slot_offset) if self.wino else None
        if uo is None:
            return None
        used = self._wino_used.get(slot_offset)
        if used is None or not used[1 if backward else 0]:
            if torch.cuda.is_current_stream_capturing():
                return None
            if used is None:
                used = self._wino_used[slot_offset] = [False, False]
            from . import ops
            used[1 if backward else 0] = True
            self._wino_table = None
            # the new copy exists before its first use: a one-row launch now, the whole table from the next refresh on
            Np, Cp = self._wino_dims[slot_offset]
            row = torch.tensor([[slot_offset, Np, Cp, -1 if backward else uo, uo if backward else -1, 0]],
                               dtype=torch.int64, device=self.device)
            ops.lib().lgm_wino_weights(self.data.data_ptr(), self.data_uf.data_ptr(), self.data_ub.data_ptr(),
                                       row.data_ptr(), 1, (Np // 32) * (Cp // 32), ops.stream())
        return (self.data_ub if backward else self.data_uf).data_ptr() + 4 * uo

    def _refresh_wino2(self, backward_operand: bool = True):
        from . import ops
        if not self._wino_used:
            return
        if self._wino_table is None:
            rows, blk = [], 0
            for off in sorted(self._wino_used):
                f, b = self._wino_used[off]
                Np, Cp = self._wino_dims[off]
                uo = self._wino_off[off]
                rows.append([off, Np, Cp, uo if f else -1, uo if b else -1, blk])      # -1: that copy is not in use
                blk += (Np // 32) * (Cp // 32)
            if self._wino_table_old is not None:
                ops._WS_RETIRED.append(self._wino_table_old)   # a captured refresh launch may still read the old table
            self._wino_table = torch.tensor(rows, dtype=torch.int64, device=self.device).contiguous()
            self._wino_table_old = self._wino_table
            self._wino_blocks = blk
        ops.lib().lgm_wino_weights(self.data.data_ptr(), self.data_uf.data_ptr(),
                                   self.data_ub.data_ptr() if backward_operand else None, self._wino_table.data_ptr(),
                                   self._wino_table.shape[0], self._wino_blocks, ops.stream())

    # -- F(4x4, 3x3) operands (csrc/winograd4.hip): only for the slots a large-map layer actually asked for ----------
    def wino4_u(self, slot_offset: int, backward: bool):
        """Address of the F(4x4) operand U = G g G^T (36 values per weight) of the 3x3 slot at ``slot_offset``, or None.
        Slots register themselves on first use (the big-channel layers of the small maps never do: their operands
        would be 4x the weights, per direction, rewritten every step).  Each slot owns its two tensors and they never
        move, so addresses baked into captured graphs stay valid when another slot registers later; registration itself
        is refused while a capture is running (the caller then takes the F(2x2) kernel)."""
        ent = self._w4.get(slot_offset)
        if ent is None:
            if not self.wino or slot_offset not in self._wino_off or torch.cuda.is_current_stream_capturing():
                return None
            slot = next(s for s in self.slots if s.offset == slot_offset)
            Np, T, Cp = slot.phys_shape
            if T != 9 or Np % 64 or Cp % 64:
                self._w4[slot_offset] = False
                return None
            ent = (torch.zeros(Np * Cp * 36, dtype=torch.float32, device=self.device),
                   torch.zeros(Np * Cp * 36, dtype=torch.float32, device=self.device), Np, Cp)
            self._w4[slot_offset] = ent
            self._w4_table = None
            from . import ops
            # the new slot's operands exist before its first use: a one-row launch now, the whole table from the next
            # refresh on
            row = torch.tensor([[slot_offset, Np, Cp, 0, 0, 0]], dtype=torch.int64, device=self.device)
            ops.lib().lgm_wino4_weights(self.data.data_ptr(), ent[0].data_ptr(), ent[1].data_ptr(), row.data_ptr(), 1,
                                        (Np // 32) * (Cp // 32), ops.stream())
        return (ent[1] if backward else ent[0]).data_ptr() if ent else None

    def _refresh_wino4(self, backward_operand: bool = True):
        from . import ops
        live = [(off, e) for off, e in self._w4.items() if e]
        if not live:
            return
        if self._w4_table is None:
            # one table-driven launch for all registered slots: destination offsets are relative to the FIRST slot's
            # tensors (the others sit wherever the allocator put them: signed offsets)
            bf, bb = live[0][1][0].data_ptr(), live[0][1][1].data_ptr()
            rows, blk = [], 0
            for off, (uf, ub, Np, Cp) in live:
                assert (uf.data_ptr() - bf) % 4 == 0 and (ub.data_ptr() - bb) % 4 == 0
                rows.append([off, Np, Cp, (uf.data_ptr() - bf) // 4, (ub.data_ptr() - bb) // 4, blk])
                blk += (Np // 32) * (Cp // 32)
            if self._w4_table_old is not None:
                ops._WS_RETIRED.append(self._w4_table_old)     # a captured refresh launch may still read the old table
            self._w4_table = torch.tensor(rows, dtype=torch.int64, device=self.device).contiguous()
            self._w4_table_old = self._w4_table
            self._w4_blocks, self._w4_base = blk, (bf, bb)
        ops.lib().lgm_wino4_weights(self.data.data_ptr(), self._w4_base[0], self._w4_base[1] if backward_operand else None,
                                    self._w4_table.data_ptr(), self._w4_table.shape[0], self._w4_blocks, ops.stream())

    # -- split-precision planes ---------------------------------------------------------------
    def enable_b3(self):
        """Allocate the bf16 planes; ``refresh_split`` / ``refresh_transposed`` keep them current."""
        from . import ops
        if self.b3:
            return
        rows_f, rows_t, cf, ct = [], [], 0, 0
        for s in self.slots:
            if s.kind != "weight":
                continue
            Np, T, Cp = s.phys_shape
            if T == 9 and Np % 32 == 0 and Cp % 16 == 0:      # forward operand [Np][9][Cp]
                rows_f.append([s.offset, Np, T, Cp, cf])
                cf += Np * T * Cp // 8
            if T == 9 and Cp % 32 == 0 and Np % 16 == 0:      # transposed operand [Cp][9][Np]
                rows_t.append([s.offset, Cp, T, Np, ct])
                ct += Np * T * Cp // 8
        assert self.total < 2 ** 31
        mk = lambda r: torch.tensor(r, dtype=torch.int32, device=self.device).contiguous() if r else None  # noqa: E731
        self._b3_tables = (mk(rows_f), cf, mk(rows_t), ct)
        self._b3_slots = ({r[0] for r in rows_f}, {r[0] for r in rows_t})
        self.pstride = (self.total + 7) // 8 * 8       # elements per plane (16-byte aligned planes)
        self.planes = torch.zeros(3 * self.pstride, dtype=torch.int16, device=self.device)
        self.planes_t = torch.zeros(3 * self.pstride, dtype=torch.int16, device=self.device)
        self.b3 = True
        ops.register_b3_flat(self)

    def refresh_split(self):
        """bf16 planes of the CURRENT weights (call once per forward pass: the optimizer moved them)."""
        from . import ops
        tab, chunks, _, _ = self._b3_tables
        if tab is not None:
            ops.lib().lgm_split_bf16x3(self.data.data_ptr(), self.planes.data_ptr(), tab.data_ptr(), tab.shape[0], chunks,
                                       self.pstride, ops.stream())

    def _refresh_split_t(self):
        from . import ops
        _, _, tab, chunks = self._b3_tables
        if tab is not None:
            ops.lib().lgm_split_bf16x3(self.data_t.data_ptr(), self.planes_t.data_ptr(), tab.data_ptr(), tab.shape[0],
                                       chunks, self.pstride, ops.stream())

    def tptr(self, p: nn.Parameter):
        if self.data_t is None:
            return None
        off = self.by_param[id(p)].offset
        if off in self._t_skip:
            return None
        return self.data_t.data_ptr() + 4 * off

    # -- pointers --------------------------------------------------------------------------
    def ptr(self, p: nn.Parameter) -> int:
        return self.data.data_ptr() + 4 * self.by_param[id(p)].offset

    def gptr(self, p: nn.Parameter) -> int:
        return self.grad.data_ptr() + 4 * self.by_param[id(p)].offset

    def slot(self, p: nn.Parameter) -> ParamSlot:
        return self.by_param[id(p)]

    def still_bound(self) -> bool:
        """False when something (e.g. module.to()) replaced the parameter storage."""
        s = self.slots[0]
        return s.param.data.data_ptr() == self.data.data_ptr() + 4 * s.offset and s.param.device == self.device

    # -- gradients -------------------------------------------------------------------------
    def begin_backward(self) -> float:
        """Returns beta for gradient writes of this backward pass (0 = overwrite, 1 = accumulate)."""
        if self.slots[0].param.grad is None and self._grad_views_bound:
            # an external optimizer called zero_grad(set_to_none=True)
            self.fresh = True
            self._grad_views_bound = False
        beta = 0.0 if self.fresh else 1.0
        self.fresh = False
        self.written = True
        return beta

    def bind_grad_views(self):
        if self._grad_views_bound:
            return
        for s in self.slots:
            s.param.grad = logical_view(self.grad[s.offset:s.offset + s.numel], s.param.shape, s.kind, s.phys_shape)
        self._grad_views_bound = True

    def zero_grad(self):
        self.fresh = True

    def clone_storage(self) -> torch.Tensor:
        return self.data.clone()


def module_flat(module: nn.Module) -> Optional[FlatParams]:
    for p in module.parameters():
        return getattr(p, "_lgm_flat", None)
    return None
