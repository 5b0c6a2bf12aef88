"""GPU: the Winograd F(2x2,3x3) kernels (csrc/winograd.hip) against a float64 torch convolution on the CPU:
forward / input gradient / weight gradient at every map class (4x4, 8x8, >= 16x16), with split-K, bias, residual
and accumulation, on channel SLICES of wider buffers (pitch > channels), and their error measured beside the
direct fp32 MFMA kernels' (LGM_NO_WINO path of the same library).  Reference: Block.proj ddpm.py:160-171."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def maxerr(a, ref):
    a, ref = a.double().cpu(), ref.double()
    return float((a - ref).abs().max() / ref.abs().max())


def wino_weights(w):
    from lgm_hip import ops
    Np, _, Cp = w.shape
    uf = torch.empty(Np * Cp * 16, device=w.device)
    ub = torch.empty(Np * Cp * 16, device=w.device)
    tab = torch.tensor([[0, Np, Cp, 0, 0, 0]], dtype=torch.int64, device=w.device)
    ops.lib().lgm_wino_weights(w.data_ptr(), uf.data_ptr(), ub.data_ptr(), tab.data_ptr(), 1, (Np // 32) * (Cp // 32),
                               ops.stream())
    return uf, ub


def wino(yx, g, a, u, bias, res, out):
    from lgm_hip import ops
    L = ops.lib()
    n = L.lgm_conv3x3_wino_workspace(ctypes.byref(g), yx)
    ws = ops.workspace(n, a.device) if n > 0 else None
    L.lgm_conv3x3_wino(yx, ctypes.byref(g), a.data_ptr(), ops.pitch(a), u.data_ptr(), None if bias is None else bias.data_ptr(),
                       None if res is None else res.data_ptr(), 0 if res is None else ops.pitch(res), out.data_ptr(),
                       ops.pitch(out), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, ops.stream())
    return n


# (B, H, Cin, Cout): every unit class (G = 2 / 4 / 8), with and without split-K, several tile blocks and channel blocks
CASES = [(16, 4, 64, 64), (32, 4, 512, 128), (8, 8, 128, 64), (4, 8, 384, 256), (2, 16, 64, 64), (2, 16, 192, 128),
         (3, 32, 64, 64), (1, 32, 128, 64), (1, 64, 64, 64)]


@pytest.mark.parametrize("case", CASES)
def test_winograd_forward_and_input_gradient(dev, case, parity):
    from lgm_hip import ops
    B, hw, ci, co = case
    gen = torch.Generator().manual_seed(sum(case))
    # operands live in channel slices of wider buffers (the UNet's concat buffers): pitch > channels
    xbuf = torch.randn(B, hw, hw, ci + 32, generator=gen)
    ybuf = torch.randn(B, hw, hw, co + 64, generator=gen)
    x, y = xbuf[..., 32:], ybuf[..., :co]
    w = torch.randn(co, 9, ci, generator=gen) / (3 * ci ** 0.5)
    bias = torch.randn(co, generator=gen)
    res = torch.randn(B, hw, hw, co, generator=gen)
    w4 = w.reshape(co, 3, 3, ci).permute(0, 3, 1, 2).double()
    ref_xy = F.conv2d(x.permute(0, 3, 1, 2).double(), w4, bias.double(), padding=1).permute(0, 2, 3, 1) + res.double()
    ref_yx = F.conv_transpose2d(y.permute(0, 3, 1, 2).double(), w4, None, padding=1).permute(0, 2, 3, 1)
    xd_buf, yd_buf = xbuf.to(dev), ybuf.to(dev)
    xd, yd = xd_buf[..., 32:], yd_buf[..., :co]
    wd, bd, rd = w.to(dev), bias.to(dev), res.to(dev)
    uf, ub = wino_weights(wd)
    g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    assert ops.lib().lgm_conv3x3_wino_supported(ctypes.byref(g), 0) == 1
    obuf = torch.full((B, hw, hw, co + 16), 7.0, device=dev)
    out = obuf[..., 16:]
    nws = wino(0, g, xd, uf, bd, rd, out)
    parity(f"forward (+bias +residual, pitched operands, split-K workspace {nws} B)", maxerr(out, ref_xy), 2e-6)
    assert float((obuf[..., :16] - 7.0).abs().max()) == 0           # nothing written outside the channel slice
    gx = torch.empty(B, hw, hw, ci, device=dev)
    wino(1, g, yd, ub, None, None, gx)
    parity("input gradient", maxerr(gx, ref_yx), 2e-6)
    acc = torch.randn(B, hw, hw, ci, generator=gen).to(dev)
    gx2 = acc.clone()
    wino(1, g, yd, ub, None, gx2, gx2)                               # accumulate in place (res = out)
    parity("input gradient accumulated onto an existing gradient", maxerr(gx2, ref_yx + acc.double().cpu()), 2e-6)
    # the direct fp32 MFMA kernels on the same operands, for the record
    wt = torch.zeros_like(wd)
    tbl = torch.tensor([[0, co, 9, ci, 0]], dtype=torch.int32, device=dev)
    ops.lib().lgm_transpose_weights(wd.data_ptr(), wt.data_ptr(), tbl.data_ptr(), 1, ((co + 31) // 32) * ((ci + 31) // 32) * 9,
                                    ops.stream())
    o2 = torch.empty(B, hw, hw, co, device=dev)
    ws = ops._conv_ws(g, 0, dev)
    ops.lib().lgm_conv_xy(ctypes.byref(g), xd.data_ptr(), ops.pitch(xd), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(),
                          ops.pitch(rd), o2.data_ptr(), ops.pitch(o2), None if ws is None else ws.data_ptr(),
                          0 if ws is None else ws.numel() * 4, ops.stream())
    print(f"[parity] direct fp32 MFMA kernel on the same operands: {maxerr(o2, ref_xy):.2e}")
    # determinism: two launches, identical bits
    out2 = torch.empty(B, hw, hw, co, device=dev)
    out3 = torch.empty(B, hw, hw, co, device=dev)
    wino(0, g, xd, uf, bd, rd, out2)
    wino(0, g, xd, uf, bd, rd, out3)
    assert torch.equal(out2, out3) and torch.equal(out2, out.contiguous())


@pytest.mark.parametrize("case", CASES)
def test_winograd_weight_gradient(dev, case, parity):
    from lgm_hip import ops
    B, hw, ci, co = case
    if not (ci % 64 == 0 and co % 64 == 0):
        pytest.skip("weight gradient blocks are 64 x 64")
    gen = torch.Generator().manual_seed(sum(case) + 1)
    xbuf = torch.randn(B, hw, hw, ci + 32, generator=gen)
    x = xbuf[..., :ci]
    y = torch.randn(B, hw, hw, co, generator=gen)
    w0 = torch.zeros(co, ci, 3, 3, dtype=torch.double, requires_grad=True)
    out = F.conv2d(x.permute(0, 3, 1, 2).double(), w0, None, padding=1)
    gw_ref, = torch.autograd.grad(out, w0, y.permute(0, 3, 1, 2).double())
    gw_ref = gw_ref.permute(0, 2, 3, 1).reshape(co, 9, ci)
    gb_ref = y.double().sum((0, 1, 2))
    g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    xd, yd = xbuf.to(dev)[..., :ci], y.to(dev)
    gw = torch.full((co, 9, ci), float("nan"), device=dev)
    gb = torch.full((co,), float("nan"), device=dev)
    ops.conv_wgrad(g, yd, xd, gw.data_ptr(), 0.0, gb.data_ptr())
    k = ops.lib()._dll.lgm_last_kernel().decode()
    assert "wino_wgrad_kernel" in k, k                               # the Winograd kernel is the one that ran
    parity("weight gradient", maxerr(gw, gw_ref), 2e-6)
    parity("fused bias gradient", maxerr(gb, gb_ref), 2e-6)
    rows = []
    gw2 = torch.ones((co, 9, ci), device=dev)
    ops.conv_wgrad(g, yd, xd, gw2.data_ptr(), 1.0, None, defer=rows)
    assert rows, "deferred slab descriptors expected"
    ops.wgrad_reduce_batch(rows, dev)
    parity("deferred slabs + batched fixed-order reduction, accumulated (beta = 1)", maxerr(gw2 - 1.0, gw_ref), 2e-6)
    gw3 = torch.empty_like(gw)
    ops.conv_wgrad(g, yd, xd, gw3.data_ptr(), 0.0, gb.data_ptr())
    assert torch.equal(gw, gw3)                                      # deterministic


def test_winograd_weight_transform_matches_definition(dev):
    """U = G g G^T in the fragment layout [N/32][C/8][16][2][32][4] (forward) and with mirrored taps and swapped roles
    (input gradient), against the definition evaluated in float64."""
    gen = torch.Generator().manual_seed(5)
    co, ci = 64, 96
    w = torch.randn(co, 9, ci, generator=gen)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]], dtype=torch.double)
    g4 = w.reshape(co, 3, 3, ci).double()
    U = torch.einsum("ia,nabc,jb->ijnc", G, g4, G).reshape(16, co, ci)                  # [xi][n][c]
    Ub = torch.einsum("ia,nabc,jb->ijnc", G, g4.flip(1, 2), G).reshape(16, co, ci)      # mirrored taps
    uf, ub = wino_weights(w.to(dev))
    got_f = uf.cpu().double().view(co // 32, ci // 8, 16, 2, 32, 4)                     # [nb][chunk][xi][kk][n][j]
    exp_f = U.view(16, co // 32, 32, ci // 8, 2, 4).permute(1, 3, 0, 4, 2, 5)
    assert float((got_f - exp_f).abs().max()) < 1e-6
    got_b = ub.cpu().double().view(ci // 32, co // 8, 16, 2, 32, 4)                     # roles swapped: N = c, K = n
    exp_b = Ub.permute(0, 2, 1).reshape(16, ci // 32, 32, co // 8, 2, 4).permute(1, 3, 0, 4, 2, 5)
    assert float((got_b - exp_b).abs().max()) < 1e-6
