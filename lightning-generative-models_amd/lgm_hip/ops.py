"""Thin tensor-level wrappers over the C-ABI (liblgm_hip.so).

torch is used for device memory and streams only.  Activations are NHWC fp32 tensors
``[B, H, W, C]`` (or ``[rows, C]`` matrices) whose last dimension is dense; they may be channel
slices of a wider buffer (pitch = stride of the pixel dimension).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ._lib import ConvGeom, lib

ACT_NONE, ACT_SILU, ACT_GELU, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3, 4, 5


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def pitch(t: torch.Tensor) -> int:
    """Distance in floats between consecutive pixels / rows."""
    assert t.stride(-1) == 1 or t.shape[-1] == 1, "last dim must be dense"
    if t.dim() == 1:
        return t.shape[0]
    pt = t.stride(-2)
    if t.dim() == 4:
        B, H, W, _ = t.shape
        assert (W == 1 or True) and (H == 1 or t.stride(1) == W * pt) and (B == 1 or t.stride(0) == H * W * pt), \
            f"not a pitched NHWC tensor: shape {tuple(t.shape)} strides {t.stride()}"
    elif t.dim() == 3:
        assert t.shape[0] == 1 or t.stride(0) == t.shape[1] * pt
    return pt


def rows(t: torch.Tensor) -> int:
    n = 1
    for s in t.shape[:-1]:
        n *= s
    return n


# ----------------------------------------------------------------------------------------
# workspace (one growing buffer per device; ops run in stream order so sharing is safe)
# ----------------------------------------------------------------------------------------
_WS = {}
_WS_RETIRED = []     # superseded buffers are NEVER freed: a captured HIP graph may have their address baked in


def workspace(nbytes: int, device) -> torch.Tensor:
    key = torch.device(device).index or 0
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        n = max(int(nbytes * 1.25) // 4 + 64, 1 << 20)
        if ws is not None:
            # a graph replay after this point still writes its scratch data into the OLD buffer: returning
            # it to the caching allocator would let that land in a live tensor (graph and eager launches
            # run in stream order, so the two buffers are never used concurrently)
            _WS_RETIRED.append(ws)
        ws = torch.empty(n, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def new(shape, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# ----------------------------------------------------------------------------------------
# convolution family
# ----------------------------------------------------------------------------------------
def make_geom(B, H, W, Cw, Nw, KH, KW, stride, pad) -> ConvGeom:
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    return ConvGeom(B, H, W, Cw, Ho, Wo, Nw, KH, KW, stride, pad)


class KernelTimer:
    """Optional per-launch timing of the convolution family with HIP events recorded on the
    launch stream (bench.py's roofline leg).  Disabled (None) on the normal path."""

    def __init__(self):
        self.records = []   # (name, flops, start_event, end_event)

    def begin(self, name, flops, nbytes=0.0):
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        self._cur = (name, flops, nbytes, s)

    def end(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        fam, flops, nbytes, s = self._cur
        k = lib()._dll.lgm_last_kernel()
        k = k.decode() if k else ""
        self.records.append((fam, k, flops, nbytes, s, e))

    def summary(self, by_kernel: bool = False):
        """per operator family (igemm_xy / igemm_yx / wgrad), or per primary kernel name as rocprofv3 prints it"""
        torch.cuda.synchronize()
        out = {}
        for fam, kern, flops, nbytes, s, e in self.records:
            d = out.setdefault((kern or fam) if by_kernel else fam, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += s.elapsed_time(e)
        return out


TIMER: Optional[KernelTimer] = None


def _conv_flops(g: ConvGeom) -> float:
    return 2.0 * g.B * g.Ho * g.Wo * g.Nw * g.KH * g.KW * g.Cw


def _conv_bytes(g: ConvGeom) -> float:
    """algorithmic bytes of one conv-family launch: X side + Y side + weights, each touched once"""
    return 4.0 * (g.B * g.H * g.W * g.Cw + g.B * g.Ho * g.Wo * g.Nw + g.Nw * g.KH * g.KW * g.Cw)


_CONV_WS_BYTES = {}

# OPT-IN split-precision 3x3 convolutions ("bf16x3 ... behind a flag"): LGM_CONV_MODE=bf16x3.
# Only flats that called enable_b3() take part; everything else, and the default, is exact fp32 MFMA.
import os as _os

B3 = _os.environ.get("LGM_CONV_MODE", "fp32") == "bf16x3"
_B3_FLATS = []
_B3_OK = {}


def register_b3_flat(fp):
    _B3_FLATS.append(fp)


def forget_dead_flats():
    """Drop the registrations of flat buffers that no longer exist and the per-weight workspaces (slab buffers keyed by
    gradient address, reducer tables) that belonged to them.  Workspaces of flat buffers that are still ALIVE stay: a
    captured HIP graph of a live model has their addresses baked in (deferred descriptors), so clearing them
    unconditionally would let a replay run against freed memory.  bench.py calls this between workloads."""
    from .flat import ALL_FLATS
    _WINO_FLATS[:] = [r for r in _WINO_FLATS if r() is not None]
    live = []
    for fp in list(ALL_FLATS):
        g = getattr(fp, "grad", None)
        if g is not None:
            live.append((g.data_ptr(), g.data_ptr() + 4 * g.numel()))

    def alive(addr) -> bool:
        return isinstance(addr, int) and any(lo <= addr < hi for lo, hi in live)

    for key in [k for k in _WGRAD_WS if not alive(k[0])]:
        del _WGRAD_WS[key]
    for key in [k for k in _WGRAD_TABLES if not any(alive(v) for row in k for v in row)]:
        del _WGRAD_TABLES[key]


def _b3_planes(ptr: Optional[int], transposed: bool):
    """-> (address of the slot's plane 0, plane stride in elements) when ``ptr`` is a 3x3 weight slot of a
    registered flat buffer, else None"""
    if ptr is None:
        return None
    for fp in _B3_FLATS:
        base = (fp.data_t if transposed else fp.data)
        if base is None:
            continue
        off = ptr - base.data_ptr()
        if 0 <= off < 4 * fp.total and (off // 4) in fp._b3_slots[1 if transposed else 0]:
            pl = fp.planes_t if transposed else fp.planes
            addr = pl.data_ptr() + (off // 4) * 2
            return (addr, fp.pstride) if addr % 16 == 0 else None     # fragment loads are 16 bytes wide
    return None


def _b3_supported(g: ConvGeom, mode: int, a_pitch: int) -> bool:
    key = (g.B, g.H, g.W, g.Cw, g.Nw, g.KH, g.stride, g.pad, mode, a_pitch)
    v = _B3_OK.get(key)
    if v is None:
        v = bool(g.KH == 3 and g.KW == 3 and lib().lgm_conv3x3_bf16x3_supported(ctypes.byref(g), mode, a_pitch))
        _B3_OK[key] = v
    return v


# Winograd F(2x2,3x3) for the 3x3 / stride 1 / pad 1 layers (exact fp32 arithmetic, 2.25x fewer MFMA FLOPs): the
# default for flats that called enable_wino(); LGM_NO_WINO=1 routes them through the direct fp32 MFMA kernels.
WINO = _os.environ.get("LGM_NO_WINO", "0") != "1"
_WINO_FLATS = []
_WINO_OK = {}
_WINO_WS = {}


def register_wino_flat(fp):
    import weakref as _wr
    _WINO_FLATS[:] = [r for r in _WINO_FLATS if r() is not None]     # weak: a model that goes away frees its copies
    _WINO_FLATS.append(_wr.ref(fp))


def _wino_u(w_ptr: Optional[int], backward: bool):
    """-> address of the transformed copy of the 3x3 weight slot at ``w_ptr`` (a registered flat buffer), or None"""
    if w_ptr is None:
        return None
    for ref in _WINO_FLATS:
        fp = ref()
        if fp is None:
            continue
        off = w_ptr - fp.data.data_ptr()
        if 0 <= off < 4 * fp.total:
            return fp.wino_u(off // 4, backward) if off % 4 == 0 else None
    return None


# Winograd F(4x4,3x3) on the large maps (csrc/winograd4.hip): taken before the F(2x2) kernel where the library's
# measured table prefers it (lgm_conv3x3_wino4_preferred); LGM_NO_WINO4=1 switches it off.
WINO4 = _os.environ.get("LGM_NO_WINO4", "0") != "1"
_WINO4_OK = {}


def _wino4_preferred(g: ConvGeom, yx: int) -> bool:
    key = (g.B, g.H, g.W, g.Cw, g.Nw, g.KH, g.KW, g.stride, g.pad, yx)
    v = _WINO4_OK.get(key)
    if v is None:
        v = bool(WINO4 and g.KH == 3 and g.KW == 3 and lib().lgm_conv3x3_wino4_preferred(ctypes.byref(g), yx))
        _WINO4_OK[key] = v
    return v


def _wino4_u(w_ptr: Optional[int], backward: bool):
    """-> address of the F(4x4) operand of the 3x3 weight slot at ``w_ptr`` (registered on first use), or None"""
    if w_ptr is None:
        return None
    for ref in _WINO_FLATS:
        fp = ref()
        if fp is None:
            continue
        off = w_ptr - fp.data.data_ptr()
        if 0 <= off < 4 * fp.total:
            return fp.wino4_u(off // 4, backward) if off % 4 == 0 else None
    return None


def _wino4_call(yx: int, g: ConvGeom, a, u_ptr: int, bias_ptr, res, out, partial: bool = False):
    """As _wino_call, through lgm_conv3x3_wino4[_partial]."""
    if a.data_ptr() % 16 or out.data_ptr() % 16 or pitch(a) % 4 or pitch(out) % 4 or (bias_ptr or 0) % 16:
        return False
    if res is not None and (res.data_ptr() % 16 or pitch(res) % 4):
        return False
    fkey = (g.B, g.H, g.W, pitch(a), pitch(out), pitch(res) if res is not None else 0)
    fits = _WINO_FITS.get(fkey)
    if fits is None:
        fits = bool(lib().lgm_conv3x3_wino_fits(ctypes.byref(g), fkey[3], fkey[4], fkey[5]))
        _WINO_FITS[fkey] = fits
    if not fits:
        return False
    key = (g.B, g.H, g.W, g.Cw, g.Nw, yx, "w4")
    n = _WINO_WS.get(key)
    if n is None:
        n = lib().lgm_conv3x3_wino4_workspace(ctypes.byref(g), yx)
        _WINO_WS[key] = n
    ws = workspace(n, a.device) if n > 0 else None
    if partial:
        assert res is None
        part = (ctypes.c_int64 * 2)()
        lib().lgm_conv3x3_wino4_partial(yx, ctypes.byref(g), a.data_ptr(), pitch(a), u_ptr, bias_ptr, out.data_ptr(),
                                        pitch(out), None if ws is None else ws.data_ptr(),
                                        0 if ws is None else ws.numel() * 4, ctypes.addressof(part), stream())
        return (ws.data_ptr(), int(part[1]), int(part[0])) if part[0] > 1 else True
    lib().lgm_conv3x3_wino4(yx, ctypes.byref(g), a.data_ptr(), pitch(a), u_ptr, bias_ptr, _p(res),
                            pitch(res) if res is not None else 0, out.data_ptr(), pitch(out),
                            None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, stream())
    return True


# GroupNorm statistics from the convolution's epilogue (lgm_conv3x3_wino4_stats + lgm_gn_fwd_stats): only where the
# GroupNorm forward would otherwise need two passes over x (the 64 x 64 maps); LGM_NO_GN_EPI_STATS=1 switches it off.
_EPI_STATS = {}


def conv_xy_stats(g: ConvGeom, x, w_ptr: Optional[int], bias_ptr, y, groups: int):
    """Forward 3x3 convolution whose consumer is GroupNorm(groups): -> ("stats", rows, rows per image, bias address) for
    gn_fwd(planes=...), or None when this layer does not take the path (the caller runs conv_xy)."""
    if not (WINO and WINO4 and _WINO_FLATS and not B3) or g.KH != 3 or g.KW != 3 or w_ptr is None:
        return None
    key = (g.B, g.H, g.W, g.Cw, g.Nw, groups)
    ent = _EPI_STATS.get(key)
    if ent is None:
        n, per = 0, ctypes.c_int(0)
        C = g.Nw
        if (_wino4_preferred(g, 0) and C % groups == 0 and 64 % (C // groups) == 0
                and not lib().lgm_gn_fwd_fused_supported(g.B, g.H * g.W, C, groups)):
            n = int(lib().lgm_conv3x3_wino4_stats_floats(ctypes.byref(g), ctypes.addressof(per)))
        ent = (n, int(per.value))
        _EPI_STATS[key] = ent
    if ent[0] <= 0:
        return None
    if x.data_ptr() % 16 or y.data_ptr() % 16 or pitch(x) % 4 or pitch(y) % 4 or (bias_ptr or 0) % 16:
        return None
    fkey = (g.B, g.H, g.W, pitch(x), pitch(y), 0)
    fits = _WINO_FITS.get(fkey)
    if fits is None:
        fits = bool(lib().lgm_conv3x3_wino_fits(ctypes.byref(g), fkey[3], fkey[4], fkey[5]))
        _WINO_FITS[fkey] = fits
    if not fits:
        return None
    u = _wino4_u(w_ptr, False)
    if u is None:
        return None
    st = torch.empty(ent[0], dtype=torch.float32, device=x.device)
    if TIMER is not None:
        TIMER.begin("igemm_xy", _conv_flops(g), _conv_bytes(g))
    lib().lgm_conv3x3_wino4_stats(ctypes.byref(g), x.data_ptr(), pitch(x), u, bias_ptr, y.data_ptr(), pitch(y),
                                  st.data_ptr(), ent[0], stream())
    if TIMER is not None:
        TIMER.end()
    return ("stats", st, ent[1], bias_ptr)


def _wino_supported(g: ConvGeom, yx: int) -> bool:
    key = (g.B, g.H, g.W, g.Cw, g.Nw, g.KH, g.KW, g.stride, g.pad, yx)
    v = _WINO_OK.get(key)
    if v is None:
        v = bool(g.KH == 3 and g.KW == 3 and lib().lgm_conv3x3_wino_supported(ctypes.byref(g), yx))
        _WINO_OK[key] = v
    return v


_WINO_FITS = {}


def _wino_call(yx: int, g: ConvGeom, a, u_ptr: int, bias_ptr, res, out, partial: bool = False):
    """Launch the Winograd kernel when the operands qualify (16-byte aligned, pitch % 4, 32-bit offsets); False = not
    taken (the caller falls back to the direct kernel).  ``partial``: the split-K planes are left for the consumer
    (GroupNorm) to sum - returns (planes address, plane stride in floats, planes) or, when the launch did not split,
    True (``out`` complete)."""
    if a.data_ptr() % 16 or out.data_ptr() % 16 or pitch(a) % 4 or pitch(out) % 4 or (bias_ptr or 0) % 16:
        return False
    if res is not None and (res.data_ptr() % 16 or pitch(res) % 4):
        return False
    fkey = (g.B, g.H, g.W, pitch(a), pitch(out), pitch(res) if res is not None else 0)
    fits = _WINO_FITS.get(fkey)
    if fits is None:
        fits = bool(lib().lgm_conv3x3_wino_fits(ctypes.byref(g), fkey[3], fkey[4], fkey[5]))
        _WINO_FITS[fkey] = fits
    if not fits:
        return False
    key = (g.B, g.H, g.W, g.Cw, g.Nw, yx, partial)
    n = _WINO_WS.get(key)
    if n is None:
        n = (lib().lgm_conv3x3_wino_workspace_partial if partial else lib().lgm_conv3x3_wino_workspace)(ctypes.byref(g), yx)
        _WINO_WS[key] = n
    ws = workspace(n, a.device) if n > 0 else None
    if partial:
        assert res is None
        part = (ctypes.c_int64 * 2)()
        lib().lgm_conv3x3_wino_partial(yx, ctypes.byref(g), a.data_ptr(), pitch(a), u_ptr, bias_ptr, out.data_ptr(),
                                       pitch(out), None if ws is None else ws.data_ptr(),
                                       0 if ws is None else ws.numel() * 4, ctypes.addressof(part), stream())
        return (ws.data_ptr(), int(part[1]), int(part[0])) if part[0] > 1 else True
    lib().lgm_conv3x3_wino(yx, ctypes.byref(g), a.data_ptr(), pitch(a), u_ptr, bias_ptr, _p(res),
                           pitch(res) if res is not None else 0, out.data_ptr(), pitch(out),
                           None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, stream())
    return True


def _conv_ws(g: ConvGeom, yx: int, device):
    key = (g.B, g.H, g.W, g.Cw, g.Nw, g.KH, g.stride, g.pad, yx)
    n = _CONV_WS_BYTES.get(key)
    if n is None:
        n = lib().lgm_conv_workspace(ctypes.byref(g), yx)
        _CONV_WS_BYTES[key] = n
    return workspace(n, device) if n > 0 else None


PLANES = _os.environ.get("LGM_NO_PLANES", "0") != "1"     # A/B switch: GroupNorm sums split-K partial planes itself
POSTOPS = _os.environ.get("LGM_NO_POSTOP", "0") != "1"    # A/B switch: activations / masks as separate launches


class PostOp(ctypes.Structure):
    """LgmPostOp (include/lgm_hip.h): out = act(conv + bias + res) * (mask > 0 ? 1 : mask_slope); optionally the
    BatchNorm-backward sums of ``out`` from the same epilogue (bn_*)"""
    _fields_ = [("act", ctypes.c_int32), ("slope", ctypes.c_float), ("mask", ctypes.c_void_p),
                ("mask_pitch", ctypes.c_int64), ("mask_slope", ctypes.c_float),
                ("bn_a", ctypes.c_void_p), ("bn_a_pitch", ctypes.c_int64), ("bn_mean", ctypes.c_void_p),
                ("bn_rstd", ctypes.c_void_p), ("bn_partial", ctypes.c_void_p), ("bn_partial_floats", ctypes.c_int64),
                ("bn_tiles", ctypes.c_void_p)]


# BatchNorm's backward reductions from the producing convolution's epilogue (LgmPostOp.bn_*): LGM_NO_BN_EPI=1 switches the
# request off (every BatchNorm backward then runs its own reduction pass, as before round 5)
BN_EPI = _os.environ.get("LGM_NO_BN_EPI", "0") != "1"


class BnSums:
    """A request for (sum gn, sum gn * xhat) per channel of the gradient ``gn`` a convolution is about to write, where
    xhat belongs to the train-mode BatchNorm that will consume gn (its saved input ``a`` and statistics).  After the
    convolution call ``tiles`` > 0 says the epilogue left ``partial[tile][3][C]``; 0: not on this path / geometry - the
    BatchNorm runs its own reduction (lgm_bn_reduce3_coef)."""
    __slots__ = ("a", "mean", "rstd", "partial", "_tiles")

    def __init__(self, a, mean, rstd):
        self.a, self.mean, self.rstd = a, mean, rstd
        r, C = rows(a), a.shape[-1]
        # at most one row tile per 64 rows (the smallest tile of the implicit-GEMM kernels)
        self.partial = torch.empty((r + 63) // 64 * 3 * C, dtype=torch.float32, device=a.device)
        self._tiles = ctypes.c_int32(0)

    @property
    def tiles(self) -> int:
        return int(self._tiles.value)


def make_post(act: int = 0, slope: float = 0.0, mask: Optional[torch.Tensor] = None, mask_slope: float = 0.0,
              bn: Optional[BnSums] = None):
    """None when there is nothing to do"""
    if bn is not None and not BN_EPI:
        bn = None
    if act == 0 and mask is None and bn is None:
        return None
    assert act in (0, ACT_RELU, ACT_LRELU)
    po = PostOp(act, slope, None if mask is None else mask.data_ptr(), 0 if mask is None else pitch(mask), mask_slope)
    if bn is not None:
        po.bn_a, po.bn_a_pitch = bn.a.data_ptr(), pitch(bn.a)
        po.bn_mean, po.bn_rstd = bn.mean.data_ptr(), bn.rstd.data_ptr()
        po.bn_partial, po.bn_partial_floats = bn.partial.data_ptr(), bn.partial.numel()
        po.bn_tiles = ctypes.addressof(bn._tiles)
    return po


def _apply_post_separately(post: PostOp, out, mask):
    """LGM_NO_POSTOP=1: the same arithmetic from the separate elementwise launches"""
    if post.act:
        act_fwd(out, None, None, out, post.act, post.slope)
    if mask is not None:
        act_bwd(mask, None, out, out, False, ACT_LRELU if post.mask_slope != 0.0 else ACT_RELU, post.mask_slope)


# ----------------------------------------------------------------------------------------
# Non-fused Winograd engine for the 4x4 / stride-2 / pad-1 layers (csrc/winograd_eng.hip; DCGAN generator / critic,
# reference dcgan.py:79-87, 150-158): F(4x4, 2x2) on the four pixel phases - input transform launch, ONE batched GEMM, output
# transform launch (bias, activation and backward mask in it) - for the forward AND the input-gradient direction of a
# registered layer.  LGM_WENG=1 switches it on, LGM_WENG_MIN_GFLOP (default 4) is the smallest layer it takes.  A layer takes
# part once its module registered the weight slot (weng_register) and keeps its transformed weights current (weng_refresh
# after every change of the weights, BEFORE the first convolution that uses them: models/generative/gan/dcgan.py).
# ----------------------------------------------------------------------------------------
# The 1x1 convolutions' forward through the engine's GEMM kernel used as a plain NT GEMM (lgm_weng_gemm_epi: ONE un-split
# launch, bias + residual in the epilogue).  Layer by layer and cold (tools/gemm1x1_bench.py) it is within +-10 % of
# lgm_conv_xy's dispatcher for 1024 <= rows <= 32768 with >= 128 reduction and output channels (two layers 1.3x faster; those
# launches split K and run a reducer), 0.6 - 0.7x on the 32 x 32 maps - and INSIDE the step it gains nothing: 9.81 / 9.81 vs 9.81 / 9.80 ms at B = 128, 6.61 vs 6.67 at B = 64, 4.38 vs
# 4.40 at B = 16 (profiles/r06_negative_results.txt).  Opt-in: LGM_GEMM1X1=1.
GEMM1X1 = _os.environ.get("LGM_GEMM1X1", "0") == "1"


def _gemm1x1_take(g: ConvGeom, x, y, res, w_ptr, bias_ptr) -> bool:
    if not (g.KH == 1 and g.KW == 1 and g.stride == 1 and g.pad == 0) or w_ptr is None:
        return False
    M = g.B * g.H * g.W
    if not (1024 <= M <= 32768 and g.Cw >= 128 and g.Nw >= 128):
        return False
    if x.data_ptr() % 16 or pitch(x) % 4 or w_ptr % 16 or g.Cw % 4:
        return False
    return M * max(pitch(x), g.Cw) * 4 < (1 << 31)


WENG = _os.environ.get("LGM_WENG", "0") == "1"
WENG_MIN_FLOP = float(_os.environ.get("LGM_WENG_MIN_GFLOP", "4")) * 1e9
_WENG_U = {}       # weight address -> (Nw, Cw, Uxy tensor [25][Nw][4 Cw], Uyx tensor [4][25][Cw][Nw])


def weng_register(w_ptr: int, Nw: int, Cw: int, device):
    """Allocate the transformed-weight tensors of the 4x4 / stride-2 layer whose physical weight [Nw][16][Cw] starts at
    ``w_ptr`` (idempotent; the tensors never move, so captured graphs stay valid)."""
    if not WENG or Cw % 4 or Nw % 4 or Cw < 32 or Nw < 32:
        return False
    ent = _WENG_U.get(w_ptr)
    if ent is None or ent[0] != Nw or ent[1] != Cw or ent[2].device != torch.device(device):
        _WENG_U[w_ptr] = (Nw, Cw, torch.zeros(25 * Nw * 4 * Cw, dtype=torch.float32, device=device),
                          torch.zeros(100 * Cw * Nw, dtype=torch.float32, device=device))
    return True


def weng_refresh(w_ptrs, xy: bool = True, yx: bool = True):
    """U = G g G^T of the current weights for the given registered layers (one launch per direction and layer)."""
    for wp in w_ptrs:
        ent = _WENG_U.get(wp)
        if ent is not None:
            lib().lgm_weng_f42_weights(wp, ent[0], ent[1], ent[2].data_ptr() if xy else None,
                                       ent[3].data_ptr() if yx else None, stream())


def _weng_take(g: ConvGeom, w_ptr, res, partial: bool):
    if not WENG or partial or res is not None or w_ptr is None:
        return None
    if not (g.KH == 4 and g.KW == 4 and g.stride == 2 and g.pad == 1 and g.H % 8 == 0 and g.W % 8 == 0):
        return None
    ent = _WENG_U.get(w_ptr)
    if ent is None or ent[0] != g.Nw or ent[1] != g.Cw or _conv_flops(g) < WENG_MIN_FLOP:
        return None
    return ent


def _weng_conv(yx: int, g: ConvGeom, a, ent, bias_ptr, out, post: Optional[PostOp]):
    """The three launches; ``a`` / ``out``: the Y side / X side tensors for yx = 1, X side / Y side for yx = 0."""
    L, st = lib(), stream()
    Nw, Cw, uxy, uyx = ent
    T = g.B * (g.Ho // 4) * (g.Wo // 4)
    act, slope, mask, mpitch, mslope = 0, 0.0, None, 0, 0.0
    if post is not None:
        act, slope, mask, mpitch, mslope = post.act, post.slope, post.mask, post.mask_pitch, post.mask_slope
    if yx == 0:
        K = 4 * Cw
        ws = workspace(4 * 25 * T * (K + Nw), a.device)
        V, M = ws.data_ptr(), ws.data_ptr() + 4 * 25 * T * K
        L.lgm_weng_f42_in_xy(a.data_ptr(), pitch(a), g.B, g.H, g.W, Cw, V, st)
        L.lgm_weng_gemm(V, uxy.data_ptr(), M, T, Nw, K, K, K, Nw, 25, T * K, Nw * K, T * Nw, st)
        L.lgm_weng_f42_out_xy_post(M, g.B, g.Ho, g.Wo, Nw, bias_ptr, out.data_ptr(), pitch(out), act, slope, mask, mpitch,
                                   mslope, st)
    else:
        ws = workspace(4 * 100 * T * (Nw + Cw), a.device)
        V, M = ws.data_ptr(), ws.data_ptr() + 4 * 100 * T * Nw
        L.lgm_weng_f42_in_yx(a.data_ptr(), pitch(a), g.B, g.Ho, g.Wo, Nw, V, st)
        L.lgm_weng_gemm(V, uyx.data_ptr(), M, T, Cw, Nw, Nw, Nw, Cw, 100, T * Nw, Cw * Nw, T * Cw, st)
        L.lgm_weng_f42_out_yx_post(M, g.B, g.Ho, g.Wo, Cw, bias_ptr, out.data_ptr(), pitch(out), act, slope, mask, mpitch,
                                   mslope, st)


def conv_xy(g: ConvGeom, x, w_ptr: int, bias_ptr: Optional[int], res, y, partial: bool = False, post: Optional[PostOp] = None,
            post_mask=None):
    """``partial=True`` (the consumer is a GroupNorm that can sum split-K planes, see gn_fwd): returns
    (planes address, plane stride, planes, bias address) when the convolution left its result in pieces - ``y`` is then
    NOT written and the bias NOT applied - else None (``y`` complete).
    ``post`` (make_post): activation / backward mask applied by the convolution's epilogue (lgm_conv_xy_post)."""
    if TIMER is not None:
        TIMER.begin("igemm_xy", _conv_flops(g), _conv_bytes(g))
    if GEMM1X1 and post is None and not partial and _gemm1x1_take(g, x, y, res, w_ptr, bias_ptr):
        # mid-sized 1x1 convolutions as ONE un-split GEMM launch of the engine's kernel (no split-K planes, no reducer)
        M = g.B * g.H * g.W
        lib().lgm_weng_gemm_epi(x.data_ptr(), w_ptr, y.data_ptr(), M, g.Nw, g.Cw, pitch(x), g.Cw, pitch(y), bias_ptr,
                                _p(res), pitch(res) if res is not None else 0, stream())
        if TIMER is not None:
            TIMER.end()
        return None
    ent = _weng_take(g, w_ptr, res, partial)
    if ent is not None:          # (a BatchNorm-sums request in ``post`` is not served here: its tile count stays 0)
        _weng_conv(0, g, x, ent, bias_ptr, y, post)
        if TIMER is not None:
            TIMER.end()
        return None
    if post is not None:
        ws = _conv_ws(g, 0, x.device)
        if POSTOPS:
            lib().lgm_conv_xy_post(ctypes.byref(g), x.data_ptr(), pitch(x), w_ptr, bias_ptr, _p(res),
                                   pitch(res) if res is not None else 0, y.data_ptr(), pitch(y),
                                   None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4,
                                   ctypes.byref(post), stream())
        else:
            lib().lgm_conv_xy(ctypes.byref(g), x.data_ptr(), pitch(x), w_ptr, bias_ptr, _p(res),
                              pitch(res) if res is not None else 0, y.data_ptr(), pitch(y),
                              None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, stream())
            _apply_post_separately(post, y, post_mask)
        if TIMER is not None:
            TIMER.end()
        return None
    if WINO and _WINO_FLATS and not B3 and _wino4_preferred(g, 0):
        u = _wino4_u(w_ptr, False)
        if u is not None:
            r = _wino4_call(0, g, x, u, bias_ptr, res, y, partial and PLANES and res is None)
            if r:
                if TIMER is not None:
                    TIMER.end()
                return None if r is True else r + (bias_ptr,)
    if WINO and _WINO_FLATS and not B3 and _wino_supported(g, 0):
        u = _wino_u(w_ptr, False)
        if u is not None:
            r = _wino_call(0, g, x, u, bias_ptr, res, y, partial and PLANES and res is None)
            if r:
                if TIMER is not None:
                    TIMER.end()
                return None if r is True else r + (bias_ptr,)
    ws = _conv_ws(g, 0, x.device)
    pl = _b3_planes(w_ptr, False) if (B3 and _b3_supported(g, 0, pitch(x))) else None
    if pl is not None:
        lib().lgm_conv3x3_bf16x3(0, ctypes.byref(g), x.data_ptr(), pitch(x), pl[0], pl[1], bias_ptr, _p(res),
                                 pitch(res) if res is not None else 0, y.data_ptr(), pitch(y),
                                 None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, stream())
    else:
        lib().lgm_conv_xy(ctypes.byref(g), x.data_ptr(), pitch(x), w_ptr, bias_ptr, _p(res),
                          pitch(res) if res is not None else 0, y.data_ptr(), pitch(y),
                          None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, stream())
    if TIMER is not None:
        TIMER.end()


def conv_stats(yx: int, g: ConvGeom, a, w_ptr: int, out, wt_ptr: Optional[int] = None):
    """conv_xy (yx = 0) / conv_yx (yx = 1) without bias and residual whose epilogue also leaves the BatchNorm
    statistics of ``out`` per row tile.  Returns (partials tensor, tiles) - tiles == 0: this geometry could not,
    run lgm_bn_stats on ``out``.  Only the generic implicit-GEMM kernels do this (the DCGAN 4x4 / stride-2 layers)."""
    L = lib()
    if TIMER is not None:
        TIMER.begin("igemm_yx" if yx else "igemm_xy", _conv_flops(g), _conv_bytes(g))
    ent = _weng_take(g, w_ptr, None, False)
    if ent is not None:          # the engine's output transform leaves no statistics: tiles = 0, the BatchNorm reduces itself
        _weng_conv(yx, g, a, ent, None, out, None)
        if TIMER is not None:
            TIMER.end()
        return None, 0
    ws = _conv_ws(g, yx, a.device)
    oc = g.Cw if yx else g.Nw
    stats = torch.empty(L.lgm_conv_stats_floats(ctypes.byref(g), yx), dtype=torch.float32, device=a.device)
    nt = ctypes.c_int(0)
    wsp, wsb = (None, 0) if ws is None else (ws.data_ptr(), ws.numel() * 4)
    if yx:
        L.lgm_conv_yx_stats(ctypes.byref(g), a.data_ptr(), pitch(a), w_ptr, wt_ptr, out.data_ptr(), pitch(out), wsp, wsb,
                            stats.data_ptr(), ctypes.addressof(nt), stream())
    else:
        L.lgm_conv_xy_stats(ctypes.byref(g), a.data_ptr(), pitch(a), w_ptr, out.data_ptr(), pitch(out), wsp, wsb,
                            stats.data_ptr(), ctypes.addressof(nt), stream())
    if TIMER is not None:
        TIMER.end()
    assert oc == out.shape[-1]
    return stats, int(nt.value)


def conv_yx(g: ConvGeom, y, w_ptr: int, bias_ptr: Optional[int], res, x, wt_ptr: Optional[int] = None,
            partial: bool = False, post: Optional[PostOp] = None, post_mask=None):
    """``partial`` / ``post``: as conv_xy."""
    if TIMER is not None:
        TIMER.begin("igemm_yx", _conv_flops(g), _conv_bytes(g))
    ent = _weng_take(g, w_ptr, res, partial)
    if ent is not None:
        _weng_conv(1, g, y, ent, bias_ptr, x, post)
        if TIMER is not None:
            TIMER.end()
        return None
    if post is not None:
        ws = _conv_ws(g, 1, y.device)
        if POSTOPS:
            lib().lgm_conv_yx_post(ctypes.byref(g), y.data_ptr(), pitch(y), w_ptr, wt_ptr, bias_ptr, _p(res),
                                   pitch(res) if res is not None else 0, x.data_ptr(), pitch(x),
                                   None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4,
                                   ctypes.byref(post), stream())
        else:
            lib().lgm_conv_yx(ctypes.byref(g), y.data_ptr(), pitch(y), w_ptr, wt_ptr, bias_ptr, _p(res),
                              pitch(res) if res is not None else 0, x.data_ptr(), pitch(x),
                              None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, stream())
            _apply_post_separately(post, x, post_mask)
        if TIMER is not None:
            TIMER.end()
        return None
    if WINO and _WINO_FLATS and not B3 and _wino4_preferred(g, 1):
        u = _wino4_u(w_ptr, True)
        if u is not None:
            r = _wino4_call(1, g, y, u, bias_ptr, res, x, partial and PLANES and res is None)
            if r:
                if TIMER is not None:
                    TIMER.end()
                return None if r is True else r + (bias_ptr,)
    if WINO and _WINO_FLATS and not B3 and _wino_supported(g, 1):
        u = _wino_u(w_ptr, True)
        if u is not None:
            r = _wino_call(1, g, y, u, bias_ptr, res, x, partial and PLANES and res is None)
            if r:
                if TIMER is not None:
                    TIMER.end()
                return None if r is True else r + (bias_ptr,)
    ws = _conv_ws(g, 1, y.device)
    pl = _b3_planes(wt_ptr, True) if (B3 and wt_ptr is not None and _b3_supported(g, 1, pitch(y))) else None
    if pl is not None:
        lib().lgm_conv3x3_bf16x3(1, ctypes.byref(g), y.data_ptr(), pitch(y), pl[0], pl[1], bias_ptr, _p(res),
                                 pitch(res) if res is not None else 0, x.data_ptr(), pitch(x),
                                 None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, stream())
        if TIMER is not None:
            TIMER.end()
        return
    lib().lgm_conv_yx(ctypes.byref(g), y.data_ptr(), pitch(y), w_ptr, wt_ptr, bias_ptr, _p(res),
                      pitch(res) if res is not None else 0, x.data_ptr(), pitch(x),
                      None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, stream())
    if TIMER is not None:
        TIMER.end()


_WGRAD_WS = {}        # deferred mode: one persistent slab workspace per weight (keyed by its gradient address)
_WGRAD_TABLES = {}    # tuple of descriptor rows -> (device table, blocks)


WGRAD_QUEUE = _os.environ.get("LGM_NO_WGRAD_QUEUE", "0") != "1"       # A/B switch: generic weight gradients launch at once


def wgrad_queue_flush():
    """Issue the generic weight-gradient launches still waiting for partners (lgm_wgrad_queue_*)."""
    if TIMER is not None:            # (their FLOPs were counted where they were queued)
        TIMER.begin("wgrad", 0.0, 0.0)
    lib().lgm_wgrad_queue_flush()
    if TIMER is not None:
        TIMER.end()


def conv_wgrad(g: ConvGeom, y, x, gw_ptr: int, beta: float, gbias_ptr: Optional[int] = None, defer=None, queue: bool = False):
    """``defer``: a list collecting slab descriptors; the caller must call ``wgrad_reduce_batch(defer)`` before
    the gradients are used (one reduce launch for many layers instead of one per layer).
    ``queue`` (with ``defer``): a stand-alone launch of the generic kernel may wait for partners - the caller keeps ``y`` /
    ``x`` alive and calls ``wgrad_queue_flush()`` before the reduction (GradCtx does)."""
    L = lib()
    queue = queue and defer is not None and WGRAD_QUEUE
    nbytes = L.lgm_conv_wgrad_workspace(ctypes.byref(g))
    if TIMER is not None:
        TIMER.begin("wgrad", _conv_flops(g), _conv_bytes(g))
    if defer is None:
        ws = workspace(nbytes, y.device)
        L.lgm_conv_wgrad(ctypes.byref(g), y.data_ptr(), pitch(y), x.data_ptr(), pitch(x), gw_ptr, gbias_ptr, beta,
                         ws.data_ptr(), ws.numel() * 4, stream())
    else:
        key = (gw_ptr, nbytes)
        ws = _WGRAD_WS.get(key)
        if ws is None:
            ws = torch.empty(max(nbytes // 4 + 4, 16), dtype=torch.float32, device=y.device)
            _WGRAD_WS[key] = ws
        desc = (ctypes.c_int64 * 8)()
        if queue:
            L.lgm_wgrad_queue_enable(1)
        try:
            L.lgm_conv_wgrad_deferred(ctypes.byref(g), y.data_ptr(), pitch(y), x.data_ptr(), pitch(x), gw_ptr, gbias_ptr,
                                      beta, ws.data_ptr(), ws.numel() * 4, ctypes.addressof(desc), stream())
        finally:
            if queue:
                L.lgm_wgrad_queue_enable(0)
        if desc[6] > 1:
            defer.append(tuple(desc))
    if TIMER is not None:
        TIMER.end()


_PAIR_OK = {}


# Up to four large-map layers' weight gradients in one launch (lgm_conv3x3_wino_wgradn): LGM_NO_WGRAD2=1 issues them
# singly, LGM_WGRAD_GROUP=n (2 ... 4) bounds the group.  Default 4 since the F(4x4) weight-gradient kernel takes up to four
# layers per launch (wino4_wgrad4_kernel): 10.03 / 10.02 / 10.01 ms per step at B = 128 and 6.98 / 6.95 / 6.93 at B = 64 for
# n = 2 / 3 / 4 (round 4's "groups of 3 and 4 measured the same" came through a dangling geometry array that refused every
# group above two; with only the F(2x2) kernel grouping, n = 4 was 10.11 vs 10.08).
WGRAD2 = _os.environ.get("LGM_NO_WGRAD2", "0") != "1"
WGRAD_GROUP = max(2, min(8, int(_os.environ.get("LGM_WGRAD_GROUP", "4"))))
_WG2_OK = {}
_WG2_WS = {}


class WgradItem(ctypes.Structure):
    """LgmWgradItem (include/lgm_hip.h)"""
    _fields_ = [("g", ctypes.c_void_p), ("y", ctypes.c_void_p), ("y_pitch", ctypes.c_int64), ("x", ctypes.c_void_p),
                ("x_pitch", ctypes.c_int64), ("gw", ctypes.c_void_p), ("gbias", ctypes.c_void_p), ("beta", ctypes.c_float),
                ("ws", ctypes.c_void_p), ("ws_bytes", ctypes.c_int64), ("desc", ctypes.c_void_p)]


def _gkey(g: ConvGeom):
    return (g.B, g.H, g.W, g.Cw, g.Nw, g.KH, g.KW, g.stride, g.pad)


def _geom_array(geoms):
    arr = (ctypes.c_void_p * len(geoms))()
    for i, g in enumerate(geoms):
        arr[i] = ctypes.addressof(g)
    return arr


def wgrad_group_supported(geoms) -> bool:
    key = tuple(_gkey(g) for g in geoms)
    v = _WG2_OK.get(key)
    if v is None:
        arr = _geom_array(geoms)        # bound to a local: the C side reads it during the call
        v = bool(lib().lgm_conv3x3_wino_wgradn_supported(len(geoms), ctypes.addressof(arr)))
        del arr
        _WG2_OK[key] = v
    return v


def wgrad2_supported(ga: ConvGeom, gb: ConvGeom) -> bool:
    return wgrad_group_supported([ga, gb])


def wgrad_queueable(g: ConvGeom, gy, x) -> bool:
    """The layers whose weight gradient waits for partners: 3x3 layers on the large maps whose input gradient runs
    apart (F(4x4)), with operands the grouped launch accepts."""
    if not (WGRAD2 and WINO and _WINO_FLATS and not B3 and _wino4_preferred(g, 1)):
        return False
    if gy.data_ptr() % 16 or x.data_ptr() % 16 or pitch(gy) % 4 or pitch(x) % 4:
        return False
    return wgrad_group_supported([g, g])


def conv_wgrad_group(entries, defer):
    """entries = 2 ... 4 of (geometry, gy, x, gw address, beta, gbias address): all weight gradients in ONE launch; their
    slab descriptors join ``defer`` (the bucket's batched reduction)."""
    L = lib()
    n = len(entries)
    geoms = [e[0] for e in entries]
    wkey = tuple(_gkey(g) for g in geoms)
    need = _WG2_WS.get(wkey)
    if need is None:
        out = (ctypes.c_int64 * n)()
        arr = _geom_array(geoms)        # bound to a local: the C side reads it during the call
        L.lgm_conv3x3_wino_wgradn_workspaces(n, ctypes.addressof(arr), ctypes.addressof(out))
        del arr
        # never smaller than the single-layer plan's need: the same slab buffer serves a layer whichever way it runs
        need = tuple(max(int(out[k]), int(L.lgm_conv_wgrad_workspace(ctypes.byref(g)))) for k, g in enumerate(geoms))
        _WG2_WS[wkey] = need
    items = (WgradItem * n)()
    descs = []
    flops = nbytes_alg = 0.0
    for k, (g, gy, x, gw_ptr, beta, gb_ptr) in enumerate(entries):
        key = (gw_ptr, need[k])
        ws = _WGRAD_WS.get(key)
        if ws is None:
            ws = torch.empty(max(need[k] // 4 + 4, 16), dtype=torch.float32, device=gy.device)
            _WGRAD_WS[key] = ws
        desc = (ctypes.c_int64 * 8)()
        descs.append(desc)
        items[k] = WgradItem(ctypes.addressof(g), gy.data_ptr(), pitch(gy), x.data_ptr(), pitch(x), gw_ptr, gb_ptr, beta,
                             ws.data_ptr(), ws.numel() * 4, ctypes.addressof(desc))
        flops += _conv_flops(g)
        nbytes_alg += _conv_bytes(g)
    if TIMER is not None:
        TIMER.begin("wgrad", flops, nbytes_alg)
    L.lgm_conv3x3_wino_wgradn(n, ctypes.addressof(items), stream())
    if TIMER is not None:
        TIMER.end()
    for desc in descs:
        if desc[6] > 1:
            defer.append(tuple(desc))


def conv_wgrad2(a, b, defer):
    conv_wgrad_group([a, b], defer)


# Up to four 1x1 layers' weight gradients in one launch of the streaming 1x1 kernel (lgm_wgrad1x1_group): a stand-alone
# launch is ~13 us of prologue / epilogue / ramp around a K loop at the MFMA rate, and nothing reads a weight gradient
# before the optimizer.  LGM_NO_WGRAD1X1_GROUP=1 issues them singly (A/B switch).
WGRAD1X1_GROUP = _os.environ.get("LGM_NO_WGRAD1X1_GROUP", "0") != "1"
_W1G_OK = {}
_W1G_WS = {}


def wgrad1x1_group_supported(geoms) -> bool:
    key = tuple(_gkey(g) for g in geoms)
    v = _W1G_OK.get(key)
    if v is None:
        arr = _geom_array(geoms)        # bound to a local: the C side reads it during the call
        v = bool(lib().lgm_wgrad1x1_group_supported(len(geoms), ctypes.addressof(arr)))
        del arr
        _W1G_OK[key] = v
    return v


def wgrad1x1_queueable(g: ConvGeom, gy, x) -> bool:
    """1x1 layers whose weight gradient takes the streaming kernel on its own launch today (never the one-launch
    gemm_bwd_pair): it can wait for partners."""
    if not WGRAD1X1_GROUP or B3 or g.KH != 1 or g.KW != 1 or g.stride != 1 or g.pad != 0:
        return False
    if gy.data_ptr() % 16 or x.data_ptr() % 16 or pitch(gy) % 4 or pitch(x) % 4:
        return False
    return wgrad1x1_group_supported([g, g])


def conv_wgrad1x1_group(entries, defer):
    """entries = 2 ... 4 of (geometry, gy, x, gw address, beta, gbias address), 1x1 layers: all weight gradients in ONE
    launch; slab descriptors of the layers that split join ``defer``."""
    L = lib()
    n = len(entries)
    geoms = [e[0] for e in entries]
    wkey = tuple(_gkey(g) for g in geoms)
    need = _W1G_WS.get(wkey)
    if need is None:
        out = (ctypes.c_int64 * n)()
        arr = _geom_array(geoms)
        L.lgm_wgrad1x1_group_workspaces(n, ctypes.addressof(arr), ctypes.addressof(out))
        del arr
        need = tuple(max(int(out[k]), int(L.lgm_conv_wgrad_workspace(ctypes.byref(g)))) for k, g in enumerate(geoms))
        _W1G_WS[wkey] = need
    items = (WgradItem * n)()
    descs = []
    flops = nbytes_alg = 0.0
    for k, (g, gy, x, gw_ptr, beta, gb_ptr) in enumerate(entries):
        key = (gw_ptr, need[k])
        ws = _WGRAD_WS.get(key)
        if ws is None:
            ws = torch.empty(max(need[k] // 4 + 4, 16), dtype=torch.float32, device=gy.device)
            _WGRAD_WS[key] = ws
        desc = (ctypes.c_int64 * 8)()
        descs.append(desc)
        items[k] = WgradItem(ctypes.addressof(g), gy.data_ptr(), pitch(gy), x.data_ptr(), pitch(x), gw_ptr, gb_ptr, beta,
                             ws.data_ptr(), ws.numel() * 4, ctypes.addressof(desc))
        flops += _conv_flops(g)
        nbytes_alg += _conv_bytes(g)
    if TIMER is not None:
        TIMER.begin("wgrad", flops, nbytes_alg)
    L.lgm_wgrad1x1_group(n, ctypes.addressof(items), stream())
    if TIMER is not None:
        TIMER.end()
    for desc in descs:
        if desc[6] > 1:
            defer.append(tuple(desc))


def conv_bwd_pair(g: ConvGeom, gy, x, w_ptr: int, gw_ptr: int, beta: float, gbias_ptr: Optional[int], defer, res, gx,
                  partial: bool = False):
    """Input gradient AND weight gradient of a 3x3 layer in ONE launch (lgm_conv3x3_wino_bwd): at small per-GPU batches
    each of the two fills a fraction of the chip and is latency-bound; side by side they take the time of one.
    Returns False when the pair kernel does not take this layer (the caller issues conv_wgrad + conv_yx), else None
    or, with ``partial``, the planes tuple of conv_yx(partial=True)."""
    if not (WINO and _WINO_FLATS and not B3) or g.KH != 3 or g.KW != 3:
        return False
    if _wino4_preferred(g, 1) and _wino4_u(w_ptr, True) is not None:
        return False        # large maps: F(4x4) input gradient (conv_yx) + the stand-alone Winograd weight gradient
    u = _wino_u(w_ptr, True)
    if u is None:
        return False
    if gy.data_ptr() % 16 or x.data_ptr() % 16 or gx.data_ptr() % 16 or (gbias_ptr or 0) % 16 or gw_ptr % 16:
        return False
    if res is not None and res.data_ptr() % 16:
        return False
    key = (g.B, g.H, g.W, g.Cw, g.Nw, pitch(gy), pitch(x), pitch(gx), pitch(res) if res is not None else 0)
    ok = _PAIR_OK.get(key)
    if ok is None:
        ok = bool(lib().lgm_conv3x3_wino_bwd_supported(ctypes.byref(g), key[5], key[6], key[7], key[8]))
        _PAIR_OK[key] = ok
    if not ok:
        return False
    L = lib()
    partial = bool(partial and PLANES and res is None)
    wkey = (g.B, g.H, g.W, g.Cw, g.Nw, "pair", partial)
    sizes = _WINO_WS.get(wkey)
    if sizes is None:
        two = (ctypes.c_int64 * 2)()
        L.lgm_conv3x3_wino_bwd_workspaces(ctypes.byref(g), 1 if partial else 0, ctypes.addressof(two))
        sizes = (int(two[0]), int(two[1]))
        _WINO_WS[wkey] = sizes
    n, nbytes = sizes
    dws = workspace(n, gy.device) if n > 0 else None
    if TIMER is not None:
        TIMER.begin("bwd_pair", 2.0 * _conv_flops(g), 2.0 * _conv_bytes(g))
    if defer is None:
        # the slabs must not share the generic workspace with the input gradient's split-K planes
        wws = _pair_slabs(nbytes, gy.device)
        desc = None
    else:
        k2 = (gw_ptr, nbytes)
        wws = _WGRAD_WS.get(k2)
        if wws is None:
            wws = torch.empty(max(nbytes // 4 + 4, 16), dtype=torch.float32, device=gy.device)
            _WGRAD_WS[k2] = wws
        desc = (ctypes.c_int64 * 8)()
    part = (ctypes.c_int64 * 2)() if partial else None
    L.lgm_conv3x3_wino_bwd(ctypes.byref(g), gy.data_ptr(), pitch(gy), x.data_ptr(), pitch(x), u, _p(res),
                           pitch(res) if res is not None else 0, gx.data_ptr(), pitch(gx),
                           None if dws is None else dws.data_ptr(), 0 if dws is None else dws.numel() * 4,
                           None if part is None else ctypes.addressof(part), gw_ptr, gbias_ptr, beta, wws.data_ptr(),
                           wws.numel() * 4, None if desc is None else ctypes.addressof(desc), stream())
    if TIMER is not None:
        TIMER.end()
    if desc is not None and desc[6] > 1:
        defer.append(tuple(desc))
    if part is not None and part[0] > 1:
        return (dws.data_ptr(), int(part[1]), int(part[0]), None)
    return None


def conv_bwd_generic(g: ConvGeom, gy, x, w_ptr: int, wt_ptr: Optional[int], gw_ptr: int, beta: float,
                     gbias_ptr: Optional[int], defer, res, gx, post=None, post_mask=None, queue: bool = False):
    """Weight / bias gradient and input gradient of any layer through lgm_conv_bwd_pair: ONE launch when the dispatchers
    pick the two kernels that can share a grid (the 1x1 convolutions and linears at small row counts), else exactly
    conv_wgrad + conv_yx."""
    L = lib()
    if TIMER is not None:
        TIMER.begin("bwd_pair", 2.0 * _conv_flops(g), 2.0 * _conv_bytes(g))
    dws = _conv_ws(g, 1, gy.device)
    nbytes = L.lgm_conv_wgrad_workspace(ctypes.byref(g))
    if defer is None:
        wws, desc = _pair_slabs(nbytes, gy.device), None
    else:
        k2 = (gw_ptr, nbytes)
        wws = _WGRAD_WS.get(k2)
        if wws is None:
            wws = torch.empty(max(nbytes // 4 + 4, 16), dtype=torch.float32, device=gy.device)
            _WGRAD_WS[k2] = wws
        desc = (ctypes.c_int64 * 8)()
    args = (ctypes.byref(g), gy.data_ptr(), pitch(gy), x.data_ptr(), pitch(x), w_ptr, wt_ptr, _p(res),
            pitch(res) if res is not None else 0, gx.data_ptr(), pitch(gx),
            None if dws is None else dws.data_ptr(), 0 if dws is None else dws.numel() * 4, gw_ptr, gbias_ptr,
            beta, wws.data_ptr(), wws.numel() * 4, None if desc is None else ctypes.addressof(desc))
    queue = queue and desc is not None and WGRAD_QUEUE      # (see conv_wgrad: a stand-alone weight-gradient launch may wait)
    if queue:
        L.lgm_wgrad_queue_enable(1)
    try:
        if post is None or not POSTOPS:
            L.lgm_conv_bwd_pair(*args, stream())
            if post is not None:
                _apply_post_separately(post, gx, post_mask)
        else:                # ``post_mask`` only keeps the mask tensor alive for the duration of the call
            L.lgm_conv_bwd_pair_post(*args, ctypes.byref(post), stream())
    finally:
        if queue:
            L.lgm_wgrad_queue_enable(0)
    if TIMER is not None:
        TIMER.end()
    if desc is not None and desc[6] > 1:
        defer.append(tuple(desc))


_PAIR_SLABS = {}


def _pair_slabs(nbytes: int, device) -> torch.Tensor:
    key = torch.device(device).index or 0
    ws = _PAIR_SLABS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _WS_RETIRED.append(ws)
        ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=device)
        _PAIR_SLABS[key] = ws
    return ws


def clear_plan_caches():
    """Forget every per-geometry answer of the library's planners (kernel choice, split counts, workspace sizes, statistics
    row counts).  Runs by itself after lgm_set_cu_margin / lgm_wino4_set_light (lgm_hip/_lib.py: on_selection_change):
    those knobs change what the queries return.  Buffers are kept (captured graphs have their addresses baked in); they are
    looked up by size, so a plan that now needs more gets a new one."""
    for d in (_CONV_WS_BYTES, _WINO_OK, _WINO_WS, _WINO4_OK, _EPI_STATS, _WINO_FITS, _PAIR_OK, _WG2_OK, _WG2_WS,
              _W1G_OK, _W1G_WS, _GN_PLANES_OK):
        d.clear()


def set_kernel_selection(cu_margin: Optional[int] = None, light: Optional[int] = None):
    """The kernel selection of a rank whose gradient exchange runs BESIDE its backward pass (FlatGradSync decides):
    ``cu_margin`` CUs left to the collective's workgroups by every launch plan, ``light`` = 1: light F(4x4) workgroups.
    None leaves a knob alone, -1 returns it to the library's default (environment, else one-GPU rules)."""
    if cu_margin is not None:
        lib().lgm_set_cu_margin(int(cu_margin))
    if light is not None:
        lib().lgm_wino4_set_light(int(light))


def wgrad_reduce_batch(rows, device):
    """One launch: fixed-order reduction of the partial slabs of every deferred weight gradient in ``rows``."""
    if not rows:
        return
    key = tuple(rows)
    ent = _WGRAD_TABLES.get(key)
    if ent is None:
        tab, blk = [], 0
        for r in rows:
            tab.append(list(r) + [blk])
            blk += (r[3] + r[5] + 255) // 256
        ent = (torch.tensor(tab, dtype=torch.int64, device=device).contiguous(), blk)
        _WGRAD_TABLES[key] = ent
    if TIMER is not None:
        TIMER.begin("wgrad", 0.0, 0.0)
    lib().lgm_wgrad_reduce_batch(ent[0].data_ptr(), len(rows), ent[1], stream())
    if TIMER is not None:
        TIMER.end()
    rows.clear()


def make_reducer(rows, device):
    """A reusable launcher for the batched slab / row reduction of ``rows`` (descriptor tuples collected by a deferred
    backward pass whose buffers have fixed addresses, i.e. under graph capture): (device table, rows, blocks), or None."""
    if not rows:
        return None
    tab, blk = [], 0
    for r in rows:
        tab.append(list(r) + [blk])
        blk += (r[3] + r[5] + 255) // 256
    return (torch.tensor(tab, dtype=torch.int64, device=device).contiguous(), len(rows), blk)


def launch_reducer(ent):
    if ent is not None:
        lib().lgm_wgrad_reduce_batch(ent[0].data_ptr(), ent[1], ent[2], stream())


def colsum(a, out_ptr: int, beta: float, defer=None):
    """out[c] = beta*out[c] + sum over all leading dims of a[..., c].  ``defer``: a deferred-reduction list - the second
    stage joins the caller's batched reduction (wgrad_reduce_batch)."""
    L = lib()
    r, c = rows(a), a.shape[-1]
    if defer is not None and c % 4 == 0 and out_ptr % 16 == 0:
        nb = L.lgm_colsum_workspace(r, c)
        ws = _persistent((out_ptr, nb, "colsum"), nb, a.device)
        desc = (ctypes.c_int64 * 8)()
        L.lgm_colsum_deferred(a.data_ptr(), pitch(a), r, c, out_ptr, beta, ws.data_ptr(), ctypes.addressof(desc), stream())
        defer.append(tuple(desc))
        return
    ws = workspace(L.lgm_colsum_workspace(r, c), a.device)
    L.lgm_colsum(a.data_ptr(), pitch(a), r, c, out_ptr, beta, ws.data_ptr(), stream())


RESSTACK_FUSED = _os.environ.get("LGM_NO_RESSTACK_FUSED") is None     # A/B switch: the VQ-VAE ResidualStack forward in one launch


def resstack_fwd(x, w3_ptrs, w1_ptrs, R: int):
    """The whole ResidualStack forward in one launch (lgm_resstack_fwd): returns ([y_l], [z_l]) or None when the
    geometry is not the one the kernel is built for."""
    B, H, W, C = x.shape
    n = len(w3_ptrs)
    if not RESSTACK_FUSED or not lib().lgm_resstack_fwd_supported(H, W, C, C, R, n) or x.data_ptr() % 16 or pitch(x) % 4 \
            or any(v % 16 for v in list(w3_ptrs) + list(w1_ptrs)):
        return None
    ys = [new((B, H, W, R), x) for _ in range(n)]
    zs = [new((B, H, W, C), x) for _ in range(n)]
    arr = ctypes.c_void_p * n
    lib().lgm_resstack_fwd(x.data_ptr(), pitch(x), B, H, W, C, C, R, n, arr(*w3_ptrs), arr(*w1_ptrs),
                           arr(*[t.data_ptr() for t in ys]), arr(*[t.data_ptr() for t in zs]), stream())
    return ys, zs


# ----------------------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------------------
class GNSaved:
    __slots__ = ("mean", "rstd", "A", "Bc")


_GN_PLANES_OK = {}


def gn_planes_ok(B, HW, C, G) -> bool:
    """The one-pass GroupNorm kernels (forward AND backward) exist for this shape: they can sum split-K planes."""
    key = (B, HW, C, G)
    v = _GN_PLANES_OK.get(key)
    if v is None:
        v = bool(lib().lgm_gn_planes_supported(B, HW, C, G))
        _GN_PLANES_OK[key] = v
    return v


_GN_WS = {}


def _gn_scratch(nfloats: int, device) -> torch.Tensor:
    """GroupNorm-backward scratch (S1, S2, P, Q, R rows).  NOT the shared workspace: the split-K planes of the
    producing convolution may still be sitting there when the backward kernel runs."""
    key = torch.device(device).index or 0
    ws = _GN_WS.get(key)
    if ws is None or ws.numel() < nfloats:
        if ws is not None:
            _WS_RETIRED.append(ws)          # a captured graph may have its address baked in
        ws = torch.empty(max(int(nfloats * 1.25) + 64, 1 << 16), dtype=torch.float32, device=device)
        _GN_WS[key] = ws
    return ws


def gn_fwd(x, G, eps, gamma_ptr, beta_ptr, ss, act: bool, res, y, planes=None) -> GNSaved:
    """``planes`` = (address, stride, count, conv bias address) from conv_xy(partial=True): x is summed from them,
    WRITTEN to ``x`` and normalised in the same pass; or ("stats", tensor, rows per image, conv bias address) from
    conv_xy_stats: the statistics come from the convolution's epilogue and x is read once."""
    B, H, W, C = x.shape
    sv = GNSaved()
    stats = new((2, B, G), x)
    coef = new((2, B, C), x)
    sv.mean, sv.rstd, sv.A, sv.Bc = stats[0], stats[1], coef[0], coef[1]
    if planes is not None and planes[0] == "stats":
        lib().lgm_gn_fwd_stats(planes[1].data_ptr(), planes[2], planes[3], x.data_ptr(), pitch(x), B, H * W, C, G, eps,
                               gamma_ptr, beta_ptr, _p(ss), pitch(ss) if ss is not None else 0, 1 if act else 0,
                               _p(res), pitch(res) if res is not None else 0, y.data_ptr(), pitch(y),
                               sv.mean.data_ptr(), sv.rstd.data_ptr(), sv.A.data_ptr(), sv.Bc.data_ptr(), stream())
        return sv
    if planes is not None:
        lib().lgm_gn_fwd_planes(planes[0], planes[1], planes[2], planes[3], x.data_ptr(), pitch(x), B, H * W, C, G, eps,
                                gamma_ptr, beta_ptr, _p(ss), pitch(ss) if ss is not None else 0, 1 if act else 0,
                                _p(res), pitch(res) if res is not None else 0, y.data_ptr(), pitch(y),
                                sv.mean.data_ptr(), sv.rstd.data_ptr(), sv.A.data_ptr(), sv.Bc.data_ptr(), stream())
        return sv
    lib().lgm_gn_fwd(x.data_ptr(), pitch(x), B, H * W, C, G, eps, gamma_ptr, beta_ptr, _p(ss),
                     pitch(ss) if ss is not None else 0, 1 if act else 0, _p(res),
                     pitch(res) if res is not None else 0, y.data_ptr(), pitch(y),
                     sv.mean.data_ptr(), sv.rstd.data_ptr(), sv.A.data_ptr(), sv.Bc.data_ptr(), stream())
    return sv


def gn_bwd(x, gy, G, gamma_ptr, beta_ptr, ss, act: bool, sv: GNSaved, gx, accumulate: bool,
           ggamma_ptr, gbeta_ptr, affine_beta: float, gss, gss_beta: float, defer=None, gy_planes=None, add_gy_to=None):
    """``gy_planes`` = (address, stride, count, _) from conv_yx(partial=True): gy is summed from them (``gy`` unused).
    ``add_gy_to``: a second tensor that receives ``+= gy`` in the same launch (lgm_gn_bwd_add: an identity residual's
    gradient beside this norm)."""
    B, H, W, C = x.shape
    ws = _gn_scratch(5 * B * C, x.device)
    if add_gy_to is not None:
        assert gy_planes is None and add_gy_to.shape == gy.shape
        rws, desc = None, None
        if defer is not None and ggamma_ptr % 16 == 0 and gbeta_ptr % 16 == 0:
            key = (ggamma_ptr, B * 2 * C)
            rws = _WGRAD_WS.get(key)
            if rws is None:
                rws = torch.empty(B * 2 * C + 4, dtype=torch.float32, device=x.device)
                _WGRAD_WS[key] = rws
            desc = (ctypes.c_int64 * 8)()
        lib().lgm_gn_bwd_add(x.data_ptr(), pitch(x), gy.data_ptr(), pitch(gy), B, H * W, C, G, gamma_ptr, beta_ptr,
                             _p(ss), pitch(ss) if ss is not None else 0, 1 if act else 0, sv.mean.data_ptr(),
                             sv.rstd.data_ptr(), sv.A.data_ptr(), sv.Bc.data_ptr(), gx.data_ptr(), pitch(gx),
                             1 if accumulate else 0, ggamma_ptr, gbeta_ptr, affine_beta, _p(gss),
                             pitch(gss) if gss is not None else 0, gss_beta, ws.data_ptr(),
                             None if rws is None else rws.data_ptr(), None if desc is None else ctypes.addressof(desc),
                             add_gy_to.data_ptr(), pitch(add_gy_to), stream())
        if desc is not None and desc[6] > 0:
            defer.append(tuple(desc))
        return
    if gy_planes is not None:
        rws, desc = None, None
        if defer is not None and ggamma_ptr % 16 == 0 and gbeta_ptr % 16 == 0:
            key = (ggamma_ptr, B * 2 * C)
            rws = _WGRAD_WS.get(key)
            if rws is None:
                rws = torch.empty(B * 2 * C + 4, dtype=torch.float32, device=x.device)
                _WGRAD_WS[key] = rws
            desc = (ctypes.c_int64 * 8)()
        lib().lgm_gn_bwd_planes(x.data_ptr(), pitch(x), gy_planes[0], gy_planes[1], gy_planes[2], B, H * W, C, G,
                                gamma_ptr, beta_ptr, _p(ss), pitch(ss) if ss is not None else 0, 1 if act else 0,
                                sv.mean.data_ptr(), sv.rstd.data_ptr(), sv.A.data_ptr(), sv.Bc.data_ptr(), gx.data_ptr(),
                                pitch(gx), 1 if accumulate else 0, ggamma_ptr, gbeta_ptr, affine_beta, _p(gss),
                                pitch(gss) if gss is not None else 0, gss_beta, ws.data_ptr(),
                                None if rws is None else rws.data_ptr(), None if desc is None else ctypes.addressof(desc),
                                stream())
        if desc is not None and desc[6] > 0:
            defer.append(tuple(desc))
        return
    if defer is not None and ggamma_ptr % 16 == 0 and gbeta_ptr % 16 == 0:
        key = (ggamma_ptr, B * 2 * C)
        rws = _WGRAD_WS.get(key)
        if rws is None:
            rws = torch.empty(B * 2 * C + 4, dtype=torch.float32, device=x.device)
            _WGRAD_WS[key] = rws
        desc = (ctypes.c_int64 * 8)()
        lib().lgm_gn_bwd_deferred(x.data_ptr(), pitch(x), gy.data_ptr(), pitch(gy), B, H * W, C, G, gamma_ptr, beta_ptr,
                                  _p(ss), pitch(ss) if ss is not None else 0, 1 if act else 0, sv.mean.data_ptr(),
                                  sv.rstd.data_ptr(), sv.A.data_ptr(), sv.Bc.data_ptr(), gx.data_ptr(), pitch(gx),
                                  1 if accumulate else 0, ggamma_ptr, gbeta_ptr, affine_beta, _p(gss),
                                  pitch(gss) if gss is not None else 0, gss_beta, ws.data_ptr(), rws.data_ptr(),
                                  ctypes.addressof(desc), stream())
        if desc[6] > 0:
            defer.append(tuple(desc))
        return
    lib().lgm_gn_bwd(x.data_ptr(), pitch(x), gy.data_ptr(), pitch(gy), B, H * W, C, G, gamma_ptr, beta_ptr,
                     _p(ss), pitch(ss) if ss is not None else 0, 1 if act else 0, sv.mean.data_ptr(),
                     sv.rstd.data_ptr(), sv.A.data_ptr(), sv.Bc.data_ptr(), gx.data_ptr(), pitch(gx),
                     1 if accumulate else 0, ggamma_ptr, gbeta_ptr, affine_beta, _p(gss),
                     pitch(gss) if gss is not None else 0, gss_beta, ws.data_ptr(), stream())


def rmsnorm_fwd(x, g_ptr, res, y):
    lib().lgm_rmsnorm_fwd(x.data_ptr(), pitch(x), g_ptr, _p(res), pitch(res) if res is not None else 0,
                          y.data_ptr(), pitch(y), rows(x), x.shape[-1], stream())


def rmsnorm_bwd(x, gy, g_ptr, gx, accumulate: bool, gg_ptr, gg_beta: float, defer=None, res=None):
    L = lib()
    rp, rpitch = (res.data_ptr(), pitch(res)) if res is not None else (None, 0)
    n, C = rows(x), x.shape[-1]
    nbytes = L.lgm_rmsnorm_bwd_workspace(n, C)
    if defer is None or gg_ptr % 16 != 0:
        ws = workspace(nbytes, x.device)
        L.lgm_rmsnorm_bwd(x.data_ptr(), pitch(x), gy.data_ptr(), pitch(gy), g_ptr, gx.data_ptr(), pitch(gx),
                          1 if accumulate else 0, rp, rpitch, gg_ptr, gg_beta, n, C, ws.data_ptr(), stream())
        return
    key = (gg_ptr, nbytes)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = torch.empty(max(nbytes // 4 + 4, 16), dtype=torch.float32, device=x.device)
        _WGRAD_WS[key] = ws
    desc = (ctypes.c_int64 * 8)()
    L.lgm_rmsnorm_bwd_deferred(x.data_ptr(), pitch(x), gy.data_ptr(), pitch(gy), g_ptr, gx.data_ptr(), pitch(gx),
                               1 if accumulate else 0, rp, rpitch, gg_ptr, gg_beta, n, C, ws.data_ptr(),
                               ctypes.addressof(desc),
                               stream())
    defer.append(tuple(desc))


# ----------------------------------------------------------------------------------------
# attention cores
# ----------------------------------------------------------------------------------------
def linattn_fwd(qkv, mem_ptr, heads, dim_head, M, out):
    B, H, W, _ = qkv.shape
    n = H * W
    ctx = new((B, heads, dim_head, dim_head), qkv)
    kstat = new((2, B, heads, dim_head), qkv)
    lib().lgm_linattn_fwd(qkv.data_ptr(), pitch(qkv), mem_ptr, B, n, heads, dim_head, M, out.data_ptr(),
                          pitch(out), ctx.data_ptr(), kstat[0].data_ptr(), kstat[1].data_ptr(), stream())
    return ctx, kstat


RMS_QKV_FUSED = _os.environ.get("LGM_NO_RMS_QKV_FUSED") is None     # A/B switch: RMSNorm + to_qkv in one launch


RMS_QKV_MIN_ROWS = int(_os.environ.get("LGM_RMS_QKV_MIN_ROWS", "65536"))


def rms_qkv_fused(x, g_ptr, w_ptr, N, any_size=False):
    """(xn, qkv) = (RMSNorm_g(x), to_qkv(xn)) in one launch (lgm_rms_qkv_fused), or None when the layer is not taken:
    by default only where it was measured faster (64 channels, at least RMS_QKV_MIN_ROWS pixel rows)."""
    C = x.shape[-1]
    lvl = lib().lgm_rms_qkv_fused_supported(C, N)
    if not RMS_QKV_FUSED or lvl < (1 if any_size else 2) or x.data_ptr() % 16 or pitch(x) % 4 or g_ptr % 16 or w_ptr % 16:
        return None
    if not any_size and rows(x) < RMS_QKV_MIN_ROWS:
        return None
    xn = new(x.shape, x)
    qkv = new(tuple(x.shape[:-1]) + (N,), x)
    lib().lgm_rms_qkv_fused(x.data_ptr(), pitch(x), g_ptr, w_ptr, C, N, rows(x), xn.data_ptr(), pitch(xn), qkv.data_ptr(),
                            pitch(qkv), stream())
    return xn, qkv


LA_FWD_FUSED = _os.environ.get("LGM_NO_LA_FWD_FUSED") is None     # A/B switch: the fused LinearAttention forward tail


def linattn_fwd_fused_ok(heads, dim_head, Cout, qkv, x, wout_ptr, bout_ptr, g_ptr, any_size=False) -> bool:
    """``any_size``: every layer the kernel is built for, not only those it was measured faster on."""
    if not LA_FWD_FUSED or lib().lgm_linattn_fwd_fused_supported(heads, dim_head, Cout) < (1 if any_size else 2):
        return False
    return not any(v % 16 for v in (qkv.data_ptr(), x.data_ptr(), wout_ptr, bout_ptr, g_ptr)) and \
        pitch(qkv) % 4 == 0 and pitch(x) % 4 == 0


def linattn_fwd_fused(qkv, mem_ptr, heads, dim_head, M, wout_ptr, bout_ptr, g_ptr, x, out, o2, y):
    """ctx launch + ONE launch for softmax_d(q) ctx -> to_out[0] -> RMSNorm -> + x (lgm_linattn_fwd_fused)."""
    B, H, W, _ = qkv.shape
    ctx = new((B, heads, dim_head, dim_head), qkv)
    kstat = new((2, B, heads, dim_head), qkv)
    lib().lgm_linattn_fwd_fused(qkv.data_ptr(), pitch(qkv), mem_ptr, B, H * W, heads, dim_head, M, wout_ptr, bout_ptr,
                                g_ptr, x.shape[-1], x.data_ptr(), pitch(x), out.data_ptr(), pitch(out), o2.data_ptr(),
                                pitch(o2), y.data_ptr(), pitch(y), ctx.data_ptr(), kstat[0].data_ptr(),
                                kstat[1].data_ptr(), stream())
    return ctx, kstat


def linattn_bwd(qkv, mem_ptr, gout, ctx, kstat, heads, dim_head, M, gqkv, gmem_ptr, gmem_beta, defer=None):
    """``defer`` (GradCtx.defer_for(mem_kv)): the mem_kv gradient's partial rows join the bucket's batched reduction."""
    L = lib()
    B, H, W, _ = qkv.shape
    ws = workspace(L.lgm_linattn_bwd_workspace(B, heads, dim_head, M), qkv.device)
    if defer is None or gmem_ptr % 16 or M <= 0:
        L.lgm_linattn_bwd(qkv.data_ptr(), pitch(qkv), mem_ptr, gout.data_ptr(), pitch(gout), ctx.data_ptr(),
                          kstat[0].data_ptr(), kstat[1].data_ptr(), B, H * W, heads, dim_head, M, gqkv.data_ptr(),
                          pitch(gqkv), gmem_ptr, gmem_beta, ws.data_ptr(), stream())
        return
    nb = B * 2 * heads * dim_head * M * 4
    part = _persistent((gmem_ptr, nb), nb, qkv.device)
    desc = (ctypes.c_int64 * 8)()
    L.lgm_linattn_bwd_deferred(qkv.data_ptr(), pitch(qkv), mem_ptr, gout.data_ptr(), pitch(gout), ctx.data_ptr(),
                               kstat[0].data_ptr(), kstat[1].data_ptr(), B, H * W, heads, dim_head, M, gqkv.data_ptr(),
                               pitch(gqkv), gmem_ptr, gmem_beta, part.data_ptr(), ctypes.addressof(desc), ws.data_ptr(),
                               stream())
    if desc[6] > 0:
        defer.append(tuple(desc))


def _persistent(key, nbytes, device):
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = torch.empty(max(nbytes // 4 + 4, 16), dtype=torch.float32, device=device)
        _WGRAD_WS[key] = ws
    return ws


# The fused LinearAttention backward tail (csrc/linattn_fused.hip) for layers with at least this many (image, 128-pixel
# tile) work items: one item is four heads in a row on ONE CU (55 us), so below a full round of the chip the three
# separate launches - which spread over more workgroups - are faster (measured).  LGM_NO_LA_FUSED=1: off.
LA_FUSED = _os.environ.get("LGM_NO_LA_FUSED") is None
LA_FUSED_MIN_ITEMS = int(_os.environ.get("LGM_LA_FUSED_MIN_ITEMS", "256"))


def linattn_bwd_fused_ok(heads, dim_head, C, qkv, gout, xn, wt_ptr, gw_ptr, gmem_ptr) -> bool:
    if not LA_FUSED or not wt_ptr or not lib().lgm_linattn_bwd_fused_supported(heads, dim_head, C):
        return False
    B, H, W, _ = qkv.shape
    if B * ((H * W + 127) // 128) < LA_FUSED_MIN_ITEMS:
        return False
    return not any(v % 16 for v in (qkv.data_ptr(), gout.data_ptr(), xn.data_ptr(), wt_ptr, gw_ptr, gmem_ptr)) and \
        pitch(qkv) % 4 == 0 and pitch(gout) % 4 == 0 and pitch(xn) % 4 == 0


def linattn_bwd_fused(qkv, mem_ptr, gout, ctx, kstat, xn, wt_ptr, heads, dim_head, M, gxn, gw_ptr, gw_beta, gw_defer,
                      gmem_ptr, gmem_beta, gmem_defer):
    """LinearAttention backward with to_qkv's backward folded in (lgm_linattn_bwd_fused): writes gxn and to_qkv's weight
    gradient (``wt_ptr``: the weight's transposed copy, [C][3 * hidden]).
    ``gw_defer`` / ``gmem_defer``: deferred-reduction lists (GradCtx.defer_for) or None = reduce now."""
    L = lib()
    B, H, W, _ = qkv.shape
    n, C = H * W, xn.shape[-1]
    dev = qkv.device
    wsb = L.lgm_linattn_bwd_fused_workspace(B, heads, dim_head)
    slab_bytes = L.lgm_linattn_bwd_fused_slabs(B, n, C)
    part_bytes = B * 2 * heads * dim_head * M * 4
    # scratch that dies with the call: gctx / r; the partial buffers too when their reduction is not deferred
    extra = (0 if gw_defer is not None else slab_bytes + 64) + (0 if gmem_defer is not None else part_bytes + 64)
    ws = workspace(wsb + extra + 64, dev)
    off = (wsb + 63) // 64 * 64
    if gmem_defer is not None:
        part_ptr = _persistent((gmem_ptr, part_bytes), part_bytes, dev).data_ptr()
    else:
        part_ptr = ws.data_ptr() + off
        off += (part_bytes + 63) // 64 * 64
    slab_ptr = _persistent((gw_ptr, slab_bytes), slab_bytes, dev).data_ptr() if gw_defer is not None \
        else ws.data_ptr() + off
    d_w = (ctypes.c_int64 * 8)() if gw_defer is not None else None
    d_m = (ctypes.c_int64 * 8)() if gmem_defer is not None else None
    L.lgm_linattn_bwd_fused(qkv.data_ptr(), pitch(qkv), mem_ptr, gout.data_ptr(), pitch(gout), ctx.data_ptr(),
                            kstat[0].data_ptr(), kstat[1].data_ptr(), xn.data_ptr(), pitch(xn), wt_ptr, C, B, n, heads,
                            dim_head, M, gxn.data_ptr(), pitch(gxn), gw_ptr, gw_beta, slab_ptr, slab_bytes,
                            None if d_w is None else ctypes.addressof(d_w), gmem_ptr, gmem_beta, part_ptr,
                            None if d_m is None else ctypes.addressof(d_m), ws.data_ptr(), stream())
    if d_w is not None and d_w[6] > 0:
        gw_defer.append(tuple(d_w))
    if d_m is not None and d_m[6] > 0:
        gmem_defer.append(tuple(d_m))


def attn_fwd(qkv, mem_ptr, heads, dim_head, M, out):
    B, H, W, _ = qkv.shape
    lse = new((B, heads, H * W), qkv)
    lib().lgm_attn_fwd(qkv.data_ptr(), pitch(qkv), mem_ptr, B, H * W, heads, dim_head, M, out.data_ptr(),
                       pitch(out), lse.data_ptr(), stream())
    return lse


def attn_bwd(qkv, mem_ptr, out, gout, lse, heads, dim_head, M, gqkv, gmem_ptr, gmem_beta, defer=None):
    L = lib()
    B, H, W, _ = qkv.shape
    if defer is None or gmem_ptr % 16 or M <= 0:
        ws = workspace(L.lgm_attn_bwd_workspace(B, heads, dim_head, M), qkv.device)
        L.lgm_attn_bwd(qkv.data_ptr(), pitch(qkv), mem_ptr, out.data_ptr(), pitch(out), gout.data_ptr(),
                       pitch(gout), lse.data_ptr(), B, H * W, heads, dim_head, M, gqkv.data_ptr(), pitch(gqkv),
                       gmem_ptr, gmem_beta, ws.data_ptr(), stream())
        return
    nb = B * 2 * heads * dim_head * M * 4
    part = _persistent((gmem_ptr, nb), nb, qkv.device)
    desc = (ctypes.c_int64 * 8)()
    L.lgm_attn_bwd_deferred(qkv.data_ptr(), pitch(qkv), mem_ptr, out.data_ptr(), pitch(out), gout.data_ptr(),
                            pitch(gout), lse.data_ptr(), B, H * W, heads, dim_head, M, gqkv.data_ptr(), pitch(gqkv),
                            gmem_ptr, gmem_beta, part.data_ptr(), ctypes.addressof(desc), stream())
    if desc[6] > 0:
        defer.append(tuple(desc))


# ----------------------------------------------------------------------------------------
# elementwise
# ----------------------------------------------------------------------------------------
_POSEMB_FREQS = {}


def posemb_freqs(dim, theta, device) -> torch.Tensor:
    """The reference's frequency table (ddpm.py:127-129), computed on the HOST with torch's CPU exp."""
    import math
    key = (dim, float(theta), str(device))
    f = _POSEMB_FREQS.get(key)
    if f is None:
        half = dim // 2
        f = torch.exp(torch.arange(half) * -(math.log(theta) / (half - 1))).to(device)
        _POSEMB_FREQS[key] = f
    return f


def posemb(t, dim, theta, out):
    lib().lgm_posemb(t.data_ptr(), t.shape[0], dim, posemb_freqs(dim, theta, t.device).data_ptr(), out.data_ptr(),
                     pitch(out), stream())


# The time embedding as ONE forward launch + two backward launches instead of 6 + 6 (lgm_time_mlp_fwd / _bwd): built, parity-
# tested, and measured SLOWER on the MI355X - opt-in (LGM_TIME_MLP=1).  Per step, graph replay, three versions of the kernels
# (profiles/r06_negative_results.txt): B = 128 9.86 -> 10.14 / 9.96 / 9.92 ms, B = 16 4.37 -> 4.55 / 4.42 / 4.38.  The chain
# is 3 dependent stages of 10 MFLOP; each stage of a plain FMA kernel pays a cold global round trip (1 - 2 us) and a 6-level
# ds_bpermute butterfly per weight row, while the six separate launches it would replace are pipelined MFMA GEMMs of 5 - 6 us
# each: 32.8 us fused vs 31.8 us for the six forward launches, 89 us vs 29 us backward.
TIME_MLP = _os.environ.get("LGM_TIME_MLP", "0") == "1"
_TIME_MLP_OK = {}


def time_mlp_ok(dim: int, time_dim: int, l1, l2) -> bool:
    """The fused time-embedding kernels take these widths (dense, unpadded Linear weights)."""
    key = (dim, time_dim)
    v = _TIME_MLP_OK.get(key)
    if v is None:
        v = bool(TIME_MLP and dim % 4 == 0 and time_dim % 4 == 0 and lib().lgm_time_mlp_supported(dim, time_dim))
        _TIME_MLP_OK[key] = v
    return v and l1.bias is not None and l2.bias is not None


def time_mlp_fwd(t, dim, theta, w1, b1, w2, b2, time_dim, pe, a1, h, temb, st):
    lib().lgm_time_mlp_fwd(t.data_ptr(), t.shape[0], dim, posemb_freqs(dim, theta, t.device).data_ptr(), w1, b1, w2, b2,
                           time_dim, pe.data_ptr(), a1.data_ptr(), h.data_ptr(), temb.data_ptr(), st.data_ptr(), stream())


def time_mlp_bwd(gst, pe, a1, h, temb, w2, dim, time_dim, gtemb, ga1, gw1, gb1, gw2, gb2, beta):
    lib().lgm_time_mlp_bwd(gst.data_ptr(), pe.data_ptr(), a1.data_ptr(), h.data_ptr(), temb.data_ptr(), w2, pe.shape[0],
                           dim, time_dim, gtemb.data_ptr(), ga1.data_ptr(), gw1, gb1, gw2, gb2, float(beta), stream())


def act_fwd(x, bias_ptr, res, y, act, slope=0.0):
    lib().lgm_act_fwd(x.data_ptr(), pitch(x), bias_ptr, _p(res), pitch(res) if res is not None else 0,
                      y.data_ptr(), pitch(y), rows(x), x.shape[-1], act, slope, stream())


def act_bwd(x, bias_ptr, gy, gx, accumulate, act, slope=0.0):
    lib().lgm_act_bwd(x.data_ptr(), pitch(x), bias_ptr, gy.data_ptr(), pitch(gy), gx.data_ptr(), pitch(gx),
                      1 if accumulate else 0, rows(x), x.shape[-1], act, slope, stream())


def axpby(a, alpha, b, beta, y):
    lib().lgm_axpby(a.data_ptr(), pitch(a), alpha, _p(b), pitch(b) if b is not None else 0, beta,
                    y.data_ptr(), pitch(y), rows(a), a.shape[-1], stream())


def upsample2x_fwd(x, y):
    B, H, W, C = x.shape
    lib().lgm_upsample2x_fwd(x.data_ptr(), pitch(x), y.data_ptr(), pitch(y), B, H, W, C, stream())


def upsample2x_bwd(gy, gx, accumulate):
    B, H, W, C = gx.shape
    lib().lgm_upsample2x_bwd(gy.data_ptr(), pitch(gy), gx.data_ptr(), pitch(gx), B, H, W, C,
                             1 if accumulate else 0, stream())


def pixel_unshuffle(hi, lo, inverse: bool, accumulate: bool = False):
    """hi: [B,2H,2W,C], lo: [B,H,W,4C].  inverse=False: hi -> lo; True: lo -> hi."""
    B, H, W, C4 = lo.shape
    src, dst = (lo, hi) if inverse else (hi, lo)
    lib().lgm_pixel_unshuffle(src.data_ptr(), pitch(src), dst.data_ptr(), pitch(dst), B, H, W, C4 // 4,
                              1 if inverse else 0, 1 if accumulate else 0, stream())


def nchw_to_nhwc(src, dst):
    B, C, H, W = src.shape
    assert src.is_contiguous()
    lib().lgm_nchw_to_nhwc(src.data_ptr(), dst.data_ptr(), pitch(dst), B, C, H * W, dst.shape[-1], stream())


def nhwc_to_nchw(src, dst):
    B, C, H, W = dst.shape
    assert dst.is_contiguous()
    lib().lgm_nhwc_to_nchw(src.data_ptr(), pitch(src), dst.data_ptr(), B, C, H * W, stream())


def adam_step(p, g, m, v, n, lr, b1, b2, eps, wd, step, step_dev=None, grad_scale=1.0, decoupled=False):
    lib().lgm_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, b1, b2, eps, wd,
                        float(step), _p(step_dev), grad_scale, 1 if decoupled else 0, stream())


def image_transform(src_u8: torch.Tensor, size: int, flip: Optional[torch.Tensor] = None, out=None):
    """u8 [B,H,W,C] decoded images -> fp32 NCHW [B,C,size,size] in [-1,1] (reference DataModule transforms)."""
    assert src_u8.dtype == torch.uint8 and src_u8.dim() == 4 and src_u8.is_contiguous()
    B, H, W, C = src_u8.shape
    if out is None:
        out = torch.empty((B, C, size, size), dtype=torch.float32, device=src_u8.device)
    if flip is not None:
        assert flip.dtype == torch.uint8 and flip.numel() == B
    lib().lgm_image_transform(src_u8.data_ptr(), B, H, W, C, _p(flip), out.data_ptr(), size, stream())
    return out


def rmsprop_step(p, g, sq, n, lr, alpha, eps, wd, grad_scale=1.0):
    lib().lgm_rmsprop_step(p.data_ptr(), g.data_ptr(), sq.data_ptr(), n, lr, alpha, eps, wd, grad_scale, stream())


def clamp_(x, lo, hi):
    lib().lgm_clamp(x.data_ptr(), x.numel(), lo, hi, stream())


def ema_lerp(shadow, online, w):
    lib().lgm_ema_lerp(shadow.data_ptr(), online.data_ptr(), shadow.numel(), w, stream())


def fill(x, val):
    lib().lgm_fill(x.data_ptr(), x.numel(), val, stream())


from ._lib import on_selection_change as _on_selection_change  # noqa: E402

_on_selection_change(clear_plan_caches)
