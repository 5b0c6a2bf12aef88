"""GPU: the OPT-IN split-precision (bf16x3) 3x3 convolution kernels — never on the default path.
Error is measured against an fp64 reference and compared with the error of the exact-fp32 MFMA kernel."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _planes(w_phys, dev):
    """w_phys: [rows][9][K] fp32 -> three fragment-major bf16 planes"""
    from lgm_hip import ops
    rows, T, K = w_phys.shape
    n = w_phys.numel()
    pl = torch.empty(3 * n, dtype=torch.int16, device=dev)
    table = torch.tensor([[0, rows, T, K, 0]], dtype=torch.int32, device=dev)
    ops.lib().lgm_split_bf16x3(w_phys.data_ptr(), pl.data_ptr(), table.data_ptr(), 1, n // 8, n, ops.stream())
    return pl, n


@pytest.mark.parametrize("shape", [(8, 32, 64, 64), (4, 16, 192, 128), (16, 8, 256, 256), (16, 4, 512, 512), (2, 32, 128, 64)])
def test_bf16x3_conv_forward_and_input_gradient_hold_fp32_level_error(dev, shape):
    from lgm_hip import ops
    B, S, ci, co = shape
    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn(B, ci, S, S, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    bias = torch.randn(co, generator=g)
    gy = torch.randn(B, co, S, S, generator=g)
    ref_y = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    ref_gx = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    geom = ops.make_geom(B, S, S, ci, co, 3, 3, 1, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = w.permute(0, 2, 3, 1).reshape(co, 9, ci).contiguous().to(dev)          # [Nw][9][Cw]
    wt = w.permute(1, 2, 3, 0).reshape(ci, 9, co).contiguous().to(dev)          # [Cw][9][Nw]
    bd = bias.to(dev)
    L = ops.lib()
    assert L.lgm_conv3x3_bf16x3_supported(ctypes.byref(geom), 0, ci) == 1
    ws = ops.workspace(max(L.lgm_conv_workspace(ctypes.byref(geom), 0), L.lgm_conv_workspace(ctypes.byref(geom), 1), 16), dev)

    def err(a, b):
        return float((a.double().cpu() - b).abs().max() / b.abs().max())

    # forward
    pl, n = _planes(wp, dev)
    y3 = torch.empty(B, S, S, co, device=dev)
    L.lgm_conv3x3_bf16x3(0, ctypes.byref(geom), xd.data_ptr(), ci, pl.data_ptr(), n, bd.data_ptr(), None, 0,
                         y3.data_ptr(), co, ws.data_ptr(), ws.numel() * 4, ops.stream())
    y32 = torch.empty_like(y3)
    ops.conv_xy(geom, xd, wp.data_ptr(), bd.data_ptr(), None, y32)
    e3, e32 = err(y3.permute(0, 3, 1, 2), ref_y), err(y32.permute(0, 3, 1, 2), ref_y)
    assert e3 < 2e-6 and e3 < 4 * e32 + 1e-7, (e3, e32)
    # input gradient on the transposed copy
    plt, nt = _planes(wt, dev)
    gx3 = torch.empty(B, S, S, ci, device=dev)
    L.lgm_conv3x3_bf16x3(1, ctypes.byref(geom), gyd.data_ptr(), co, plt.data_ptr(), nt, None, None, 0,
                         gx3.data_ptr(), ci, ws.data_ptr(), ws.numel() * 4, ops.stream())
    gx32 = torch.empty_like(gx3)
    ops.conv_yx(geom, gyd, wp.data_ptr(), None, None, gx32, wt.data_ptr())
    g3, g32 = err(gx3.permute(0, 3, 1, 2), ref_gx), err(gx32.permute(0, 3, 1, 2), ref_gx)
    assert g3 < 2e-6 and g3 < 4 * g32 + 1e-7, (g3, g32)
    print(f"shape {shape}: fwd err bf16x3 {e3:.2e} fp32 {e32:.2e}; dgrad err bf16x3 {g3:.2e} fp32 {g32:.2e}")
