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


# (B, H, Cin, Cout): geometries whose partial launch splits (small maps / small batches) and one that does not
PLANE_CASES = [(16, 4, 512, 512), (16, 8, 256, 256), (4, 16, 128, 128), (16, 32, 64, 64), (128, 4, 256, 256), (64, 32, 64, 64)]


@pytest.mark.parametrize("case", PLANE_CASES)
def test_groupnorm_sums_the_split_k_planes(dev, case, parity):
    """Block.forward / its backward (ddpm.py:164-173) with the reducer folded into the consumer: the 3x3 convolution
    leaves its split-K partial planes (lgm_conv3x3_wino_partial), GroupNorm sums them while it computes its statistics
    (lgm_gn_fwd_planes: also writes the finished convolution output) / its backward sums the input-gradient planes
    (lgm_gn_bwd_planes).  Against the two-step path of the same library - convolution with its own reducer, then
    GroupNorm - every output must be IDENTICAL when the split counts agree and within fp32 reduction-order distance
    otherwise; and against float64."""
    from lgm_hip import ops
    B, hw, ci, co = case
    G = 8
    L = ops.lib()
    gen = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(B, hw, hw, ci, generator=gen).to(dev)
    w = (torch.randn(co, 9, ci, generator=gen) / (3 * ci ** 0.5)).to(dev)
    bias, gamma, beta = (torch.randn(co, generator=gen).to(dev) for _ in range(3))
    ss = torch.randn(B, 2 * co, generator=gen).to(dev) * 0.3
    res = torch.randn(B, hw, hw, co, generator=gen).to(dev)
    uf, ub = wino_weights(w)
    g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    assert L.lgm_gn_planes_supported(B, hw * hw, co, G) == 1

    def conv_partial(yx, a, u, b_, out):
        n = L.lgm_conv3x3_wino_workspace_partial(ctypes.byref(g), yx)
        ws = torch.empty(max(n // 4, 4), device=dev)
        part = (ctypes.c_int64 * 2)()
        L.lgm_conv3x3_wino_partial(yx, ctypes.byref(g), a.data_ptr(), ops.pitch(a), u.data_ptr(),
                                   None if b_ is None else b_.data_ptr(), out.data_ptr(), ops.pitch(out), ws.data_ptr(),
                                   ws.numel() * 4, ctypes.addressof(part), ops.stream())
        return ws, int(part[0]), int(part[1])

    # ---- forward: two-step reference of the same library
    u_ref = torch.empty(B, hw, hw, co, device=dev)
    wino(0, g, x, uf, bias, None, u_ref)
    y_ref = torch.empty_like(u_ref)
    sv_ref = ops.gn_fwd(u_ref, G, 1e-5, gamma.data_ptr(), beta.data_ptr(), ss, True, res, y_ref)
    # ---- forward: planes
    u_pl = torch.full((B, hw, hw, co), 3.0, device=dev)
    ws, splits, stride = conv_partial(0, x, uf, bias, u_pl)
    y_pl = torch.empty_like(u_ref)
    if splits > 1:
        sv = ops.gn_fwd(u_pl, G, 1e-5, gamma.data_ptr(), beta.data_ptr(), ss, True, res, y_pl,
                        planes=(ws.data_ptr(), stride, splits, bias.data_ptr()))
    else:
        sv = ops.gn_fwd(u_pl, G, 1e-5, gamma.data_ptr(), beta.data_ptr(), ss, True, res, y_pl)
    ref64 = F.conv2d(x.cpu().permute(0, 3, 1, 2).double(), w.cpu().reshape(co, 3, 3, ci).permute(0, 3, 1, 2).double(),
                     bias.cpu().double(), padding=1).permute(0, 2, 3, 1)
    parity(f"conv output written by GroupNorm ({splits} planes) vs float64", maxerr(u_pl, ref64), 2e-6)
    parity("conv output: planes path vs reducer path", maxerr(u_pl, u_ref.double().cpu()), 2e-6)
    parity("GroupNorm output: planes path vs reducer path", maxerr(y_pl, y_ref.double().cpu()), 5e-6)
    parity("GroupNorm mean / rstd: planes path vs reducer path",
           max(maxerr(sv.mean, sv_ref.mean.double().cpu()), maxerr(sv.rstd, sv_ref.rstd.double().cpu())), 5e-6)

    # ---- backward: gy of the GroupNorm = input gradient of a following convolution (co -> co here needs ci == co)
    if ci != co:
        return
    gz = torch.randn(B, hw, hw, co, generator=gen).to(dev)
    gy_ref = torch.empty(B, hw, hw, ci, device=dev)
    wino(1, g, gz, ub, None, None, gy_ref)
    gg = [torch.zeros(co, device=dev) for _ in range(4)]
    gss = [torch.zeros(B, 2 * co, device=dev) for _ in range(2)]
    gx_ref, gx_pl = torch.empty_like(u_ref), torch.empty_like(u_ref)
    ops.gn_bwd(u_ref, gy_ref, G, gamma.data_ptr(), beta.data_ptr(), ss, True, sv_ref, gx_ref, False, gg[0].data_ptr(),
               gg[1].data_ptr(), 0.0, gss[0], 0.0)
    gy_dummy = torch.full((B, hw, hw, ci), float("nan"), device=dev)     # never read, never written with planes
    ws, splits, stride = conv_partial(1, gz, ub, None, gy_dummy)
    if splits > 1:
        ops.gn_bwd(u_ref, gy_dummy, G, gamma.data_ptr(), beta.data_ptr(), ss, True, sv_ref, gx_pl, False, gg[2].data_ptr(),
                   gg[3].data_ptr(), 0.0, gss[1], 0.0, gy_planes=(ws.data_ptr(), stride, splits, None))
        assert bool(torch.isnan(gy_dummy).all())
    else:
        ops.gn_bwd(u_ref, gy_dummy, G, gamma.data_ptr(), beta.data_ptr(), ss, True, sv_ref, gx_pl, False, gg[2].data_ptr(),
                   gg[3].data_ptr(), 0.0, gss[1], 0.0)
    parity(f"GroupNorm input gradient from {splits} input-gradient planes vs reducer path", maxerr(gx_pl, gx_ref.double().cpu()), 5e-6)
    parity("gamma / beta gradients", max(maxerr(gg[2], gg[0].double().cpu()), maxerr(gg[3], gg[1].double().cpu())), 5e-6)
    parity("FiLM scale / shift gradients", maxerr(gss[1], gss[0].double().cpu()), 5e-6)


@pytest.mark.parametrize("case", [(16, 4, 512, 512), (8, 8, 128, 64), (2, 16, 192, 128), (16, 32, 64, 64), (3, 32, 128, 64)])
@pytest.mark.parametrize("variant", ["plain", "accumulate", "residual", "planes"])
def test_backward_pair_launch_matches_separate_launches(dev, case, variant, parity):
    """lgm_conv3x3_wino_bwd (input gradient + weight gradient of a 3x3 layer in ONE launch, the same kernels' code as
    two block ranges of one grid, split counts planned jointly) against lgm_conv3x3_wino(yx = 1) + lgm_conv_wgrad: equal
    within fp32 reduction-order distance for the input gradient
    (with a residual / accumulated into an existing tensor / left as partial planes), the weight gradient and the bias
    gradient - and against float64 autograd."""
    from lgm_hip import ops
    B, hw, ci, co = case
    L = ops.lib()
    gen = torch.Generator().manual_seed(sum(case) + 7)
    xbuf = torch.randn(B, hw, hw, ci + 32, generator=gen).to(dev)          # pitched operands, as in the concat buffers
    x = xbuf[..., 32:]
    gy = torch.randn(B, hw, hw, co, generator=gen).to(dev)
    w = (torch.randn(co, 9, ci, generator=gen) / (3 * ci ** 0.5)).to(dev)
    res = torch.randn(B, hw, hw, ci, generator=gen).to(dev)
    uf, ub = wino_weights(w)
    g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    assert L.lgm_conv3x3_wino_bwd_supported(ctypes.byref(g), co, ci + 32, ci, ci) == 1
    # ---- separate launches
    gx_ref = res.clone() if variant == "accumulate" else torch.empty(B, hw, hw, ci, device=dev)
    r_ref = gx_ref if variant == "accumulate" else (res if variant == "residual" else None)
    wino(1, g, gy, ub, None, r_ref, gx_ref)
    gw_ref, gb_ref = torch.full((co, 9, ci), 0.5, device=dev), torch.full((co,), 0.25, device=dev)
    nb = L.lgm_conv_wgrad_workspace(ctypes.byref(g))
    wws = torch.empty(nb // 4 + 16, device=dev)
    two = (ctypes.c_int64 * 2)()
    L.lgm_conv3x3_wino_bwd_workspaces(ctypes.byref(g), 1 if variant == "planes" else 0, ctypes.addressof(two))
    L.lgm_conv_wgrad(ctypes.byref(g), gy.data_ptr(), co, x.data_ptr(), ci + 32, gw_ref.data_ptr(), gb_ref.data_ptr(), 1.0,
                     wws.data_ptr(), wws.numel() * 4, ops.stream())
    # ---- the pair
    gx = res.clone() if variant == "accumulate" else torch.full((B, hw, hw, ci), float("nan"), device=dev)
    r = gx if variant == "accumulate" else (res if variant == "residual" else None)
    gw, gb = torch.full((co, 9, ci), 0.5, device=dev), torch.full((co,), 0.25, device=dev)
    partial = variant == "planes"
    dws = torch.empty(max(two[0] // 4, 4), device=dev)
    wws2 = torch.empty(two[1] // 4 + 16, device=dev)
    part = (ctypes.c_int64 * 2)()
    L.lgm_conv3x3_wino_bwd(ctypes.byref(g), gy.data_ptr(), co, x.data_ptr(), ci + 32, ub.data_ptr(),
                           None if r is None else r.data_ptr(), 0 if r is None else ci, gx.data_ptr(), ci, dws.data_ptr(),
                           dws.numel() * 4, ctypes.addressof(part) if partial else None, gw.data_ptr(), gb.data_ptr(), 1.0,
                           wws2.data_ptr(), wws2.numel() * 4, None, ops.stream())
    assert ops.lib()._dll.lgm_last_kernel().decode().startswith("lgmwino::wino_bwd_pair_kernel<")
    if partial and part[0] > 1:
        assert bool(torch.isnan(gx).all())                      # planes mode leaves gx alone ...
        planes = dws[: part[0] * part[1]].view(part[0], -1)[:, : B * hw * hw * ci]
        gx = planes.sum(0).view(B, hw, hw, ci)                 # ... its sum is the input gradient
        parity(f"input gradient from {part[0]} planes vs separate launch", maxerr(gx, gx_ref.double().cpu()), 2e-6)
    else:
        parity("input gradient vs separate launch (split counts may differ)", maxerr(gx, gx_ref.double().cpu()), 2e-6)
    # (the pair's slab count and the separate launch's may differ - they do under a CU margin - so this is a comparison of two
    # summation orders: 2.2e-6 observed with LGM_CU_MARGIN=16)
    parity("weight gradient vs separate launch", maxerr(gw, gw_ref.double().cpu()), 4e-6)
    parity("bias gradient vs separate launch", maxerr(gb, gb_ref.double().cpu()), 2e-6)
    w4 = w.cpu().reshape(co, 3, 3, ci).permute(0, 3, 1, 2).double().requires_grad_(True)
    x64 = x.cpu().permute(0, 3, 1, 2).double()
    y64 = F.conv2d(x64, w4, None, padding=1)
    (gw64,) = torch.autograd.grad(y64, w4, gy.cpu().permute(0, 3, 1, 2).double())
    parity("weight gradient vs float64 autograd", maxerr(gw - 0.5, gw64.permute(0, 2, 3, 1).reshape(co, 9, ci)), 5e-6)


# ---- Winograd F(4x4, 3x3) on the large maps (csrc/winograd4.hip) ------------------------------------------------------
def wino4_weights(w):
    from lgm_hip import ops
    Np, _, Cp = w.shape
    uf = torch.empty(Np * Cp * 36, device=w.device)
    # the input-gradient operand is laid out in blocks of 64 INPUT channels: a 32-channel layer has none (forward only)
    ub = torch.empty(Np * Cp * 36, device=w.device) if Cp % 64 == 0 else None
    tab = torch.tensor([[0, Np, Cp, 0, 0, 0]], dtype=torch.int64, device=w.device)
    ops.lib().lgm_wino4_weights(w.data_ptr(), uf.data_ptr(), None if ub is None else ub.data_ptr(), tab.data_ptr(), 1,
                                (Np // 32) * (Cp // 32), ops.stream())
    return uf, ub


def wino4(yx, g, a, u, bias, res, out, partial=False, light=False):
    from lgm_hip import ops
    L = ops.lib()
    if light:                                    # csrc/winograd4l.hip through its direct entry point
        n = L.lgm_conv3x3_wino4l_workspace(ctypes.byref(g), yx)
        ws = ops.workspace(n, a.device) if n > 0 else None
        wsp, wsb = (None, 0) if ws is None else (ws.data_ptr(), ws.numel() * 4)
        L.lgm_conv3x3_wino4l(yx, ctypes.byref(g), a.data_ptr(), ops.pitch(a), u.data_ptr(),
                             None if bias is None else bias.data_ptr(), None if res is None else res.data_ptr(),
                             0 if res is None else ops.pitch(res), out.data_ptr(), ops.pitch(out), wsp, wsb, ops.stream())
        assert "wino4l_conv_kernel" in L._dll.lgm_last_kernel().decode()
        return n
    n = L.lgm_conv3x3_wino4_workspace(ctypes.byref(g), yx)
    ws = ops.workspace(n, a.device) if n > 0 else None
    wsp, wsb = (None, 0) if ws is None else (ws.data_ptr(), ws.numel() * 4)
    if partial:
        part = (ctypes.c_int64 * 2)()
        L.lgm_conv3x3_wino4_partial(yx, ctypes.byref(g), a.data_ptr(), ops.pitch(a), u.data_ptr(),
                                    None if bias is None else bias.data_ptr(), out.data_ptr(), ops.pitch(out), wsp, wsb,
                                    ctypes.addressof(part), ops.stream())
        return ws, int(part[0]), int(part[1])
    L.lgm_conv3x3_wino4(yx, ctypes.byref(g), a.data_ptr(), ops.pitch(a), u.data_ptr(), None if bias is None else bias.data_ptr(),
                        None if res is None else res.data_ptr(), 0 if res is None else ops.pitch(res), out.data_ptr(),
                        ops.pitch(out), wsp, wsb, ops.stream())
    return n


# (B, H, W, Cin, Cout): the three unit classes (8 x 8 maps: eight images per unit; 16 x 16: two; >= 16 x 32: 4 x 8 tiles of one image), several
# tile blocks per image, several channel blocks, with and without split-K, a non-square map
CASES4 = [(2, 16, 16, 64, 64), (4, 16, 16, 192, 128), (3, 32, 32, 64, 64), (1, 32, 32, 128, 64), (1, 64, 64, 64, 64),
          (2, 16, 32, 64, 128), (130, 16, 16, 64, 64), (8, 8, 8, 128, 128), (16, 8, 8, 384, 256), (128, 8, 8, 256, 256)]
F4_TOL = 2e-5     # F(4x4,3x3) in fp32: 2e-6 ... 8e-6 of the output scale against float64 (F(2x2): 2e-7); the parity bar is 1e-4


@pytest.fixture(params=[0, 1], ids=["wg32tile", "wglight"])
def w4mode(request):
    """lgm_conv3x3_wino4* with the 32-tile workgroups only (winograd4.hip) / with the light workgroups wherever they take
    the geometry (winograd4l.hip, the default): both kernels stay under test whatever the default is."""
    from lgm_hip import ops
    ops.lib().lgm_wino4_set_light(request.param)
    yield request.param
    ops.lib().lgm_wino4_set_light(-1)


@pytest.mark.parametrize("case", CASES4)
def test_winograd4_forward_and_input_gradient(dev, case, parity, w4mode):
    """lgm_conv3x3_wino4 (reference op: Block.proj ddpm.py:157-173 and its input gradient) against a float64 convolution:
    bias + residual, operands in channel slices of wider buffers, nothing written outside the output slice."""
    from lgm_hip import ops
    B, H, W, ci, co = case
    gen = torch.Generator().manual_seed(sum(case))
    xbuf = torch.randn(B, H, W, ci + 32, generator=gen)
    ybuf = torch.randn(B, H, W, co + 64, generator=gen)
    x, y = xbuf[..., 32:], ybuf[..., :co]
    w = torch.randn(co, 9, ci, generator=gen) / (3 * ci ** 0.5)
    bias = torch.randn(co, generator=gen)
    res = torch.randn(B, H, W, co, generator=gen)
    w4 = w.reshape(co, 3, 3, ci).permute(0, 3, 1, 2).double()
    ref_xy = F.conv2d(x.permute(0, 3, 1, 2).double(), w4, bias.double(), padding=1).permute(0, 2, 3, 1) + res.double()
    ref_yx = F.conv_transpose2d(y.permute(0, 3, 1, 2).double(), w4, None, padding=1).permute(0, 2, 3, 1)
    xd_buf, yd_buf = xbuf.to(dev), ybuf.to(dev)
    xd, yd = xd_buf[..., 32:], yd_buf[..., :co]
    wd, bd, rd = w.to(dev), bias.to(dev), res.to(dev)
    uf, ub = wino4_weights(wd)
    g = ops.make_geom(B, H, W, ci, co, 3, 3, 1, 1)
    assert ops.lib().lgm_conv3x3_wino4_supported(ctypes.byref(g), 0) == 1
    obuf = torch.full((B, H, W, co + 16), 7.0, device=dev)
    out = obuf[..., 16:]
    nws = wino4(0, g, xd, uf, bd, rd, out)
    parity(f"F(4x4) forward (+bias +residual, pitched operands, split-K workspace {nws} B)", maxerr(out, ref_xy), F4_TOL)
    assert float((obuf[..., :16] - 7.0).abs().max()) == 0
    gx = torch.empty(B, H, W, ci, device=dev)
    wino4(1, g, yd, ub, None, None, gx)
    parity("F(4x4) input gradient", maxerr(gx, ref_yx), F4_TOL)
    # accumulate form (res == out): the skip-connection gradients of the up path
    gx2 = torch.randn(B, H, W, ci, generator=gen).to(dev)
    ref2 = ref_yx + gx2.double().cpu()
    wino4(1, g, yd, ub, None, gx2, gx2)
    parity("F(4x4) input gradient accumulated in place", maxerr(gx2, ref2), F4_TOL)
    # bit-reproducible (fixed summation orders everywhere)
    o2 = torch.empty_like(out)
    wino4(0, g, xd, uf, bd, rd, o2)
    o3 = torch.empty_like(out)
    wino4(0, g, xd, uf, bd, rd, o3)
    assert torch.equal(o2, o3)


# light workgroups (csrc/winograd4l.hip): the cases above that its unit classes take, plus maps only it takes (H = 8 with W a
# multiple of 32; 8 x 8 maps in groups of four images), split and unsplit reductions
CASES4L = [c for c in CASES4 if not (c[1] == 8 and c[0] % 4)] + [(2, 8, 32, 64, 64), (4, 8, 8, 128, 128), (5, 24, 64, 32, 64),
                                                                (64, 32, 32, 64, 64), (12, 8, 8, 512, 256)]


@pytest.mark.parametrize("case", CASES4L)
def test_winograd4_light_workgroups(dev, case, parity):
    """lgm_conv3x3_wino4l (reference op: Block.proj ddpm.py:157-173 and its input gradient): 16-tile units on 16x16x4 MFMAs,
    the same U operands as the 32-tile kernel.  Against a float64 convolution (bias + residual, pitched operands, nothing
    written outside the output slice, accumulate-in-place form), against the 32-tile kernel where that takes the geometry,
    and twice (bit-reproducible)."""
    from lgm_hip import ops
    B, H, W, ci, co = case
    gen = torch.Generator().manual_seed(sum(case) + 11)
    xbuf = torch.randn(B, H, W, ci + 32, generator=gen)
    ybuf = torch.randn(B, H, W, co + 64, generator=gen)
    x, y = xbuf[..., 32:], ybuf[..., :co]
    w = torch.randn(co, 9, ci, generator=gen) / (3 * ci ** 0.5)
    bias = torch.randn(co, generator=gen)
    res = torch.randn(B, H, W, co, generator=gen)
    w4 = w.reshape(co, 3, 3, ci).permute(0, 3, 1, 2).double()
    ref_xy = F.conv2d(x.permute(0, 3, 1, 2).double(), w4, bias.double(), padding=1).permute(0, 2, 3, 1) + res.double()
    ref_yx = F.conv_transpose2d(y.permute(0, 3, 1, 2).double(), w4, None, padding=1).permute(0, 2, 3, 1)
    xd_buf, yd_buf = xbuf.to(dev), ybuf.to(dev)
    xd, yd = xd_buf[..., 32:], yd_buf[..., :co]
    wd, bd, rd = w.to(dev), bias.to(dev), res.to(dev)
    uf, ub = wino4_weights(wd)
    g = ops.make_geom(B, H, W, ci, co, 3, 3, 1, 1)
    assert ops.lib().lgm_conv3x3_wino4l_supported(ctypes.byref(g), 0) == 1
    obuf = torch.full((B, H, W, co + 16), 7.0, device=dev)
    out = obuf[..., 16:]
    nws = wino4(0, g, xd, uf, bd, rd, out, light=True)
    parity(f"light F(4x4) forward (+bias +residual, pitched operands, split-K workspace {nws} B)", maxerr(out, ref_xy), F4_TOL)
    assert float((obuf[..., :16] - 7.0).abs().max()) == 0
    if ops.lib().lgm_conv3x3_wino4l_supported(ctypes.byref(g), 1) == 1:      # 32 input channels: forward only
        gx = torch.empty(B, H, W, ci, device=dev)
        wino4(1, g, yd, ub, None, None, gx, light=True)
        parity("light F(4x4) input gradient", maxerr(gx, ref_yx), F4_TOL)
        gx2 = torch.randn(B, H, W, ci, generator=gen).to(dev)
        ref2 = ref_yx + gx2.double().cpu()
        wino4(1, g, yd, ub, None, gx2, gx2, light=True)
        parity("light F(4x4) input gradient accumulated in place", maxerr(gx2, ref2), F4_TOL)
    else:
        assert ci % 64
    L = ops.lib()
    L.lgm_wino4_set_light(0)
    try:
        if L.lgm_conv3x3_wino4_supported(ctypes.byref(g), 0) == 1:
            big = torch.empty_like(out)
            wino4(0, g, xd, uf, bd, rd, big)
            assert "lgmwino4::wino4_conv_kernel" in L._dll.lgm_last_kernel().decode()
            parity("light vs 32-tile workgroups (same products, other summation order over the reduction)",
                   maxerr(out, big.double().cpu()), 1e-5)
    finally:
        L.lgm_wino4_set_light(-1)
    o2 = torch.empty_like(out)
    wino4(0, g, xd, uf, bd, rd, o2, light=True)
    assert torch.equal(o2, out.contiguous())


def test_winograd4_partial_planes_and_weight_transform(dev, parity, w4mode):
    """The split-K planes lgm_conv3x3_wino4_partial leaves sum (plane 0, 1, ..., + bias: the reducer's order) to exactly
    what the complete call writes; U = G g G^T matches its float64 definition."""
    from lgm_hip import ops
    B, H, W, ci, co = 4, 16, 16, 192, 128
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, ci, generator=gen).to(dev)
    w = (torch.randn(co, 9, ci, generator=gen) / (3 * ci ** 0.5)).to(dev)
    bias = torch.randn(co, generator=gen).to(dev)
    uf, ub = wino4_weights(w)
    g = ops.make_geom(B, H, W, ci, co, 3, 3, 1, 1)
    full = torch.empty(B, H, W, co, device=dev)
    wino4(0, g, x, uf, bias, None, full)
    out = torch.full((B, H, W, co), 3.0, device=dev)
    ws, planes, stride = wino4(0, g, x, uf, bias, None, out, partial=True)
    assert planes > 1, "this geometry is expected to split its reduction"
    acc = ws[:stride].clone()
    for k in range(1, planes):
        acc = acc + ws[k * stride:(k + 1) * stride]
    acc = acc.view(B, H, W, co) + bias
    assert torch.equal(acc, full) and float((out - 3.0).abs().max()) == 0      # `out` untouched in the partial form
    # weight transform against the definition, float64
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
    g3 = w.double().cpu().reshape(co, 3, 3, ci)
    U = torch.einsum("ia,nabc,jb->nijc", G, g3, G)                               # [n][6][6][c]
    # [n/64][c/8][xi][half = (n%32)/16][(c%8)/4][row = 16 ((n%64)/32) + n%16][c%4]: split row into (n%64)/32 and n%16
    uf_l = uf.cpu().view(co // 64, ci // 8, 36, 2, 2, 2, 16, 4)
    got = uf_l.permute(0, 5, 3, 6, 2, 1, 4, 7).reshape(co, 36, ci)               # [n/64][(n%64)/32][half][n%16] -> n; [xi]; c
    parity("F(4x4) forward operand vs G g G^T", maxerr(got, U.reshape(co, 36, ci)), 4e-7)
    Ub = torch.einsum("ia,nabc,jb->cijn", G, g3.flip(1, 2), G)                   # mirrored taps, roles swapped: [c][6][6][n]
    ub_l = ub.cpu().view(ci // 64, co // 8, 36, 2, 2, 2, 16, 4)
    gotb = ub_l.permute(0, 5, 3, 6, 2, 1, 4, 7).reshape(ci, 36, co)
    parity("F(4x4) input-gradient operand vs its definition", maxerr(gotb, Ub.reshape(ci, 36, co)), 4e-7)


@pytest.mark.parametrize("pair", [((3, 32, 64, 64), (3, 32, 128, 64)), ((4, 16, 192, 128), (4, 16, 128, 128)),
                                  ((2, 64, 64, 64), (2, 64, 64, 64))])
def test_two_layers_weight_gradients_in_one_launch(dev, pair, parity):
    """lgm_conv3x3_wino_wgrad2: the weight / bias gradients of two large-map 3x3 layers from ONE launch (the chip's
    workgroups shared by work), their slabs reduced by the bucket's batched fixed-order reducer - against float64
    autograd (reference op: the weight gradient of Block.proj, ddpm.py:157-173), with beta = 0 and beta = 1, twice
    (bit-reproducible)."""
    from lgm_hip import ops
    layers, refs = [], []
    for i, (B, hw, ci, co) in enumerate(pair):
        gen = torch.Generator().manual_seed(17 * i + sum(pair[i]))
        x = torch.randn(B, hw, hw, ci, generator=gen)
        y = torch.randn(B, hw, hw, co, generator=gen)
        w0 = torch.zeros(co, ci, 3, 3, dtype=torch.double, requires_grad=True)
        out = F.conv2d(x.permute(0, 3, 1, 2).double(), w0, None, padding=1)
        gw_ref, = torch.autograd.grad(out, w0, y.permute(0, 3, 1, 2).double())
        refs.append((gw_ref.permute(0, 2, 3, 1).reshape(co, 9, ci), y.double().sum((0, 1, 2))))
        layers.append((ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1), y.to(dev), x.to(dev)))
    assert ops.wgrad2_supported(layers[0][0], layers[1][0])
    results = []
    for rep in range(2):
        gws = [torch.ones((l[1].shape[-1], 9, l[2].shape[-1]), device=dev) for l in layers]
        gbs = [torch.ones((l[1].shape[-1],), device=dev) for l in layers]           # the fused bias gradient shares beta
        rows = []
        a = (layers[0][0], layers[0][1], layers[0][2], gws[0].data_ptr(), 1.0, gbs[0].data_ptr())     # accumulate
        b = (layers[1][0], layers[1][1], layers[1][2], gws[1].data_ptr(), 0.0, None)                  # overwrite, no bias
        ops.conv_wgrad2(a, b, rows)
        assert "wino_wgrad2_kernel" in ops.lib()._dll.lgm_last_kernel().decode()
        assert len(rows) == 2
        ops.wgrad_reduce_batch(rows, dev)
        results.append((gws[0].clone(), gws[1].clone(), gbs[0].clone()))
    parity("layer a, accumulated (beta = 1)", maxerr(results[0][0] - 1.0, refs[0][0]), 2e-6)
    parity("layer a, fused bias gradient (beta = 1)", maxerr(results[0][2] - 1.0, refs[0][1]), 2e-6)
    parity("layer b (beta = 0)", maxerr(results[0][1], refs[1][0]), 2e-6)
    assert all(torch.equal(p, q) for p, q in zip(results[0], results[1]))
    # four layers in one launch (the pair twice over): each layer against its reference
    ents, gws4 = [], []
    for k in range(4):
        gk, yk, xk = layers[k % 2]
        gws4.append(torch.full((yk.shape[-1], 9, xk.shape[-1]), float("nan"), device=dev))
        ents.append((gk, yk, xk, gws4[-1].data_ptr(), 0.0, None))
    # the geometries need <= 256 workgroups: the grouped launch MUST take them (an `if` here hid a
    # dangling-pointer bug in ops.wgrad_group_supported for two rounds)
    assert ops.wgrad_group_supported([e[0] for e in ents])
    rows = []
    ops.conv_wgrad_group(ents, rows)
    assert "wino_wgrad4_kernel" in ops.lib()._dll.lgm_last_kernel().decode() and len(rows) == 4
    ops.wgrad_reduce_batch(rows, dev)
    for k in range(4):
        parity(f"four layers in one launch, layer {k}", maxerr(gws4[k], refs[k % 2][0]), 2e-6)
    # three layers: the budget split for an odd group
    ents3, gws3 = [], []
    for k in range(3):
        gk, yk, xk = layers[(k + 1) % 2]
        gws3.append(torch.full((yk.shape[-1], 9, xk.shape[-1]), float("nan"), device=dev))
        ents3.append((gk, yk, xk, gws3[-1].data_ptr(), 0.0, None))
    assert ops.wgrad_group_supported([e[0] for e in ents3])
    rows = []
    ops.conv_wgrad_group(ents3, rows)
    assert "wino_wgrad4_kernel" in ops.lib()._dll.lgm_last_kernel().decode() and len(rows) == 3
    ops.wgrad_reduce_batch(rows, dev)
    for k in range(3):
        parity(f"three layers in one launch, layer {k}", maxerr(gws3[k], refs[(k + 1) % 2][0]), 2e-6)


# ---- Winograd F(4x4, 3x3) weight gradient (csrc/winograd4_wgrad.hip) -----------------------------------------------------
CASES4W = [(3, 32, 32, 64, 64), (2, 32, 32, 128, 64), (5, 16, 16, 192, 128), (1, 64, 64, 64, 64), (2, 16, 32, 64, 128),
           (128, 16, 16, 64, 64), (8, 8, 8, 128, 128), (24, 8, 8, 384, 256), (4, 8, 24, 64, 64)]


@pytest.mark.parametrize("case", CASES4W)
def test_winograd4_weight_gradient(dev, case, parity):
    """lgm_conv3x3_wino4_wgrad (reference op: autograd's weight / bias gradient of Block.proj, ddpm.py:157-173) against
    float64 autograd: dU = sum over tiles (A dY A^T) (.) (B^T x B), dw = G^T dU G, fused bias sums; operands in channel
    slices of wider buffers, slabs through the batched fixed-order reducer with beta = 0 and beta = 1, bit-reproducible."""
    from lgm_hip import ops
    B, H, W, ci, co = case
    gen = torch.Generator().manual_seed(sum(case) + 7)
    xbuf = torch.randn(B, H, W, ci + 32, generator=gen)
    ybuf = torch.randn(B, H, W, co + 64, generator=gen)
    x, y = xbuf[..., 32:], ybuf[..., :co]
    w0 = torch.zeros(co, ci, 3, 3, dtype=torch.double, requires_grad=True)
    out = F.conv2d(x.permute(0, 3, 1, 2).double(), w0, None, padding=1)
    gw_ref, = torch.autograd.grad(out, w0, y.permute(0, 3, 1, 2).double())
    gw_ref = gw_ref.permute(0, 2, 3, 1).reshape(co, 9, ci)
    gb_ref = y.double().sum((0, 1, 2))
    xd, yd = xbuf.to(dev)[..., 32:], ybuf.to(dev)[..., :co]
    g = ops.make_geom(B, H, W, ci, co, 3, 3, 1, 1)
    L = ops.lib()
    assert L.lgm_conv3x3_wino4_wgrad_supported(ctypes.byref(g)) == 1
    n = L.lgm_conv3x3_wino4_wgrad_workspace(ctypes.byref(g))
    ws = torch.empty(n // 4 + 16, device=dev)

    def run(gw, gb, beta):
        desc = (ctypes.c_int64 * 8)()
        L.lgm_conv3x3_wino4_wgrad(ctypes.byref(g), yd.data_ptr(), ops.pitch(yd), xd.data_ptr(), ops.pitch(xd), gw.data_ptr(),
                                  None if gb is None else gb.data_ptr(), beta, ws.data_ptr(), ws.numel() * 4,
                                  ctypes.addressof(desc), ops.stream())
        assert "wino4_wgrad_kernel" in L._dll.lgm_last_kernel().decode() and desc[6] >= 2
        ops.wgrad_reduce_batch([tuple(desc)], dev)
    gw = torch.full((co, 9, ci), float("nan"), device=dev)
    gb = torch.full((co,), float("nan"), device=dev)
    run(gw, gb, 0.0)
    parity("F(4x4) weight gradient", maxerr(gw, gw_ref), 1e-5)
    parity("F(4x4) fused bias gradient", maxerr(gb, gb_ref), 2e-6)
    gw2 = torch.ones((co, 9, ci), device=dev)
    run(gw2, None, 1.0)
    parity("F(4x4) weight gradient accumulated (beta = 1)", maxerr(gw2 - 1.0, gw_ref), 1e-5)
    gw3 = torch.empty_like(gw)
    run(gw3, gb, 0.0)
    assert torch.equal(gw, gw3)


@pytest.mark.parametrize("case", [(24, 64, 64, 64, 64), (12, 64, 64, 64, 128), (48, 32, 64, 64, 64)])
def test_winograd4_forward_leaves_the_groupnorm_statistics(dev, case, parity, w4mode):
    """Block.forward (ddpm.py:164-173) on maps whose GroupNorm needs two passes over x (64 x 64): lgm_conv3x3_wino4_stats
    writes the convolution output AND, per wave, (sum, sum of squares) of its pre-bias outputs; lgm_gn_fwd_stats combines
    them in float64 and normalises reading x once.  Against the two-step path of the same library (convolution, then
    lgm_gn_fwd) and against float64: outputs, mean / rstd, FiLM + SiLU + residual."""
    from lgm_hip import ops
    B, H, W, ci, co = case
    G = 8
    L = ops.lib()
    gen = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(B, H, W, ci, generator=gen).to(dev)
    w = (torch.randn(co, 9, ci, generator=gen) / (3 * ci ** 0.5)).to(dev)
    bias = (torch.randn(co, generator=gen) * 2.0).to(dev)          # a bias far from zero: the shift in the statistics matters
    gamma, beta = (torch.randn(co, generator=gen).to(dev) for _ in range(2))
    ss = torch.randn(B, 2 * co, generator=gen).to(dev) * 0.3
    res = torch.randn(B, H, W, co, generator=gen).to(dev)
    uf, _ = wino4_weights(w)
    g = ops.make_geom(B, H, W, ci, co, 3, 3, 1, 1)
    per = ctypes.c_int(0)
    n = L.lgm_conv3x3_wino4_stats_floats(ctypes.byref(g), ctypes.addressof(per))
    assert n == B * per.value * 2 * co and per.value == ((H // 8) if w4mode else (H // 16)) * (W // 32) * 4
    u_ref = torch.empty(B, H, W, co, device=dev)
    wino4(0, g, x, uf, bias, None, u_ref)
    y_ref = torch.empty_like(u_ref)
    sv_ref = ops.gn_fwd(u_ref, G, 1e-5, gamma.data_ptr(), beta.data_ptr(), ss, True, res, y_ref)
    u_st = torch.full((B, H, W, co), 3.0, device=dev)
    st = torch.full((n + 16,), float("nan"), device=dev)
    L.lgm_conv3x3_wino4_stats(ctypes.byref(g), x.data_ptr(), ops.pitch(x), uf.data_ptr(), bias.data_ptr(), u_st.data_ptr(),
                              ops.pitch(u_st), st.data_ptr(), n, ops.stream())
    assert ("wino4l_conv_kernel<0, true>" if w4mode else "wino4_conv_kernel<0, false, 0, true>") in L._dll.lgm_last_kernel().decode()
    assert torch.equal(u_st, u_ref)                                # the same convolution, bit for bit
    assert not bool(torch.isnan(st[:n]).any()) and bool(torch.isnan(st[n:]).all())     # every row written, nothing beyond
    y_st = torch.empty_like(u_ref)
    sv = ops.gn_fwd(u_st, G, 1e-5, gamma.data_ptr(), beta.data_ptr(), ss, True, res, y_st,
                    planes=("stats", st, per.value, bias.data_ptr()))
    u64 = F.conv2d(x.cpu().permute(0, 3, 1, 2).double(), w.cpu().reshape(co, 3, 3, ci).permute(0, 3, 1, 2).double(),
                   bias.cpu().double(), padding=1)
    v = u64.reshape(B, G, -1)
    m64, r64 = v.mean(-1), 1.0 / torch.sqrt(v.var(-1, unbiased=False) + 1e-5)
    parity("mean from the epilogue statistics vs float64", maxerr(sv.mean, m64), 2e-5)
    parity("rstd from the epilogue statistics vs float64", maxerr(sv.rstd, r64), 2e-5)
    parity("mean / rstd: epilogue statistics vs the two-pass kernel",
           max(maxerr(sv.mean, sv_ref.mean.double().cpu()), maxerr(sv.rstd, sv_ref.rstd.double().cpu())), 2e-5)
    parity("GroupNorm output (FiLM, SiLU, residual): epilogue statistics vs the two-pass kernel",
           maxerr(y_st, y_ref.double().cpu()), 2e-5)
    # run to run identical (fixed butterfly, fixed float64 order)
    st2 = torch.empty_like(st)
    L.lgm_conv3x3_wino4_stats(ctypes.byref(g), x.data_ptr(), ops.pitch(x), uf.data_ptr(), bias.data_ptr(), u_st.data_ptr(),
                              ops.pitch(u_st), st2.data_ptr(), n, ops.stream())
    assert torch.equal(st[:n], st2[:n])


def test_epilogue_statistics_survive_a_large_common_offset(dev, parity):
    """the statistics rows are fp32 sums of y and y^2, and E[y^2] - mean^2 cancels when |mean| >> std inside a
    group.  Activations with mean 30 and std 0.1 give pre-bias outputs whose per-channel mean is hundreds of standard
    deviations: lgm_gn_fwd_stats must detect that from the rows and measure the slice itself (float64, two passes).  mean /
    rstd against float64 statistics of the SAME convolution output, GroupNorm output against the two-pass kernel."""
    from lgm_hip import ops
    B, H, W, ci, co, G = 24, 64, 64, 64, 64, 8          # 192 units: the reduction is not split, the statistics path is taken
    L = ops.lib()
    gen = torch.Generator().manual_seed(77)
    x = (30.0 + 0.1 * torch.randn(B, H, W, ci, generator=gen)).to(dev)
    w = (torch.randn(co, 9, ci, generator=gen) / (3 * ci ** 0.5)).to(dev)
    bias = torch.randn(co, generator=gen).to(dev)
    gamma, beta = (torch.randn(co, generator=gen).to(dev) for _ in range(2))
    uf, _ = wino4_weights(w)
    g = ops.make_geom(B, H, W, ci, co, 3, 3, 1, 1)
    per = ctypes.c_int(0)
    n = L.lgm_conv3x3_wino4_stats_floats(ctypes.byref(g), ctypes.addressof(per))
    assert n > 0
    u = torch.empty(B, H, W, co, device=dev)
    st = torch.empty(n, device=dev)
    L.lgm_conv3x3_wino4_stats(ctypes.byref(g), x.data_ptr(), ops.pitch(x), uf.data_ptr(), bias.data_ptr(), u.data_ptr(),
                              ops.pitch(u), st.data_ptr(), n, ops.stream())
    # interior pixels only see the offset; keep a slice whose groups really are "large mean, small spread"
    v = u.double().cpu().reshape(B, H * W, G, co // G).permute(0, 2, 1, 3).reshape(B, G, -1)
    m64, var64 = v.mean(-1), v.var(-1, unbiased=False)
    r64 = 1.0 / torch.sqrt(var64 + 1e-5)
    y_st, y_ref = torch.empty_like(u), torch.empty_like(u)
    sv = ops.gn_fwd(u, G, 1e-5, gamma.data_ptr(), beta.data_ptr(), None, True, None, y_st,
                    planes=("stats", st, per.value, bias.data_ptr()))
    sv_ref = ops.gn_fwd(u, G, 1e-5, gamma.data_ptr(), beta.data_ptr(), None, True, None, y_ref)
    parity("large offset: mean from the statistics path vs float64 (|d mean| * rstd)",
           float(((sv.mean.double().cpu().reshape(B, G) - m64).abs() * r64).max()), 2e-5)
    parity("large offset: rstd from the statistics path vs float64 (relative)",
           float(((sv.rstd.double().cpu().reshape(B, G) - r64).abs() / r64).max()), 2e-5)
    parity("large offset: GroupNorm output, statistics path vs the two-pass kernel",
           float((y_st - y_ref).abs().max() / y_ref.abs().max()), 5e-5)


def test_winograd4_weight_gradients_of_two_layers_in_one_launch(dev, parity):
    """The grouped launch takes the F(4x4) kernel when both layers do (large maps at a chip-filling batch): against float64."""
    from lgm_hip import ops
    layers, refs = [], []
    for i, (B, hw, ci, co) in enumerate([(128, 32, 64, 64), (128, 32, 128, 64)]):
        gen = torch.Generator().manual_seed(31 + i)
        x = torch.randn(B, hw, hw, ci, generator=gen)
        y = torch.randn(B, hw, hw, co, generator=gen)
        w0 = torch.zeros(co, ci, 3, 3, dtype=torch.double, requires_grad=True)
        # the gradient is a sum over images: reference = sum of per-slice gradients (16 slices of 8 images, float64)
        acc = torch.zeros(co, 9, ci, dtype=torch.double)
        for k in range(0, B, 8):
            out = F.conv2d(x[k:k + 8].permute(0, 3, 1, 2).double(), w0, None, padding=1)
            gk, = torch.autograd.grad(out, w0, y[k:k + 8].permute(0, 3, 1, 2).double())
            acc += gk.permute(0, 2, 3, 1).reshape(co, 9, ci)
        refs.append(acc)
        layers.append((ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1), y.to(dev), x.to(dev)))
    gws = [torch.full((l[1].shape[-1], 9, l[2].shape[-1]), float("nan"), device=dev) for l in layers]
    rows = []
    ops.conv_wgrad_group([(l[0], l[1], l[2], gw.data_ptr(), 0.0, None) for l, gw in zip(layers, gws)], rows)
    k = ops.lib()._dll.lgm_last_kernel().decode()
    assert "wino4_wgrad2_kernel" in k, k
    ops.wgrad_reduce_batch(rows, dev)
    for i in range(2):
        # 131,072 pixels per sum in fp32: the error grows with the batch (5.7e-6 measured at B = 128, 2e-6 at B <= 5)
        parity(f"grouped F(4x4) weight gradient, layer {i}", maxerr(gws[i], refs[i]), 2e-5)
    # three and four layers in one launch of the F(4x4) kernel (wino4_wgrad4_kernel): the chip's workgroups shared by work,
    # half (a third) of the slabs per layer; each layer against its float64 reference, run to run identical
    # (more than two layers share a launch only while a workgroup's run stays under ~80 phases: three mixed layers, four of
    # the small one, seven of the small one at half the batch)
    half = (ops.make_geom(64, 32, 32, 64, 64, 3, 3, 1, 1), layers[0][1][:64].contiguous(), layers[0][2][:64].contiguous())
    w0 = torch.zeros(64, 64, 3, 3, dtype=torch.double, requires_grad=True)
    acc = torch.zeros(64, 9, 64, dtype=torch.double)
    for k in range(0, 64, 8):
        out = F.conv2d(half[2][k:k + 8].cpu().permute(0, 3, 1, 2).double(), w0, None, padding=1)
        gk, = torch.autograd.grad(out, w0, half[1][k:k + 8].cpu().permute(0, 3, 1, 2).double())
        acc += gk.permute(0, 2, 3, 1).reshape(64, 9, 64)
    assert not ops.wgrad_group_supported([layers[1][0]] * 4)        # 4 x 128 -> 64 @ 32 x 32, B = 128: 128 phases per workgroup
    for n, pick in ((3, lambda k: (layers[k % 2], refs[k % 2])), (4, lambda k: (layers[0], refs[0])), (7, lambda k: (half, acc))):
        ents, outs, want = [], [], []
        for k in range(n):
            l, r = pick(k)
            want.append(r)
            outs.append(torch.full((l[1].shape[-1], 9, l[2].shape[-1]), float("nan"), device=dev))
            ents.append((l[0], l[1], l[2], outs[-1].data_ptr(), 0.0, None))
        assert ops.wgrad_group_supported([e[0] for e in ents])
        rows = []
        ops.conv_wgrad_group(ents, rows)
        assert ("wino4_wgrad8_kernel" if n > 4 else "wino4_wgrad4_kernel") in ops.lib()._dll.lgm_last_kernel().decode()
        assert len(rows) == n
        ops.wgrad_reduce_batch(rows, dev)
        for k in range(n):
            parity(f"{n} layers in one F(4x4) launch, layer {k}", maxerr(outs[k], want[k]), 2e-5)
        again = [torch.empty_like(o) for o in outs]
        rows = []
        ops.conv_wgrad_group([(e[0], e[1], e[2], a.data_ptr(), 0.0, None) for e, a in zip(ents, again)], rows)
        ops.wgrad_reduce_batch(rows, dev)
        assert all(torch.equal(a, o) for a, o in zip(again, outs))


def test_grouped_weight_gradients_of_wide_layers_do_not_share_the_f4x4_launch(dev, parity):
    """Two 512 -> 512 layers at 8 x 8 (the 64 x 64 configuration's deepest large-map layers): each needs 256 workgroups'
    worth of 64 x 32-channel blocks for two slabs, so the pair cannot share the F(4x4) launch (a zero budget once divided
    by zero on the host).  The grouped entry point refuses the pair (GradCtx.queue_wgrad asks first) and each layer goes
    alone on the F(4x4) kernel - measured faster than the two together on the F(2x2) grouped launch; results against
    float64."""
    from lgm_hip import ops
    from lgm_hip._lib import LgmError
    B, hw, ci, co = 32, 8, 512, 512
    layers, refs = [], []
    for i in range(2):
        gen = torch.Generator().manual_seed(91 + i)
        x = torch.randn(B, hw, hw, ci, generator=gen)
        y = torch.randn(B, hw, hw, co, generator=gen)
        w0 = torch.zeros(co, ci, 3, 3, dtype=torch.double, requires_grad=True)
        out = F.conv2d(x.permute(0, 3, 1, 2).double(), w0, None, padding=1)
        gw_ref, = torch.autograd.grad(out, w0, y.permute(0, 3, 1, 2).double())
        refs.append(gw_ref.permute(0, 2, 3, 1).reshape(co, 9, ci))
        layers.append((ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1), y.to(dev), x.to(dev)))
    assert ops.lib().lgm_conv3x3_wino4_preferred(ctypes.byref(layers[0][0]), 1) == 1      # each alone takes the F(4x4) kernels
    assert not ops.wgrad_group_supported([l[0] for l in layers])
    gws = [torch.full((co, 9, ci), float("nan"), device=dev) for _ in layers]
    with pytest.raises(LgmError):                      # asked anyway: an error code, no launch
        ops.conv_wgrad_group([(l[0], l[1], l[2], gw.data_ptr(), 0.0, None) for l, gw in zip(layers, gws)], [])
    # what GradCtx does with them: one launch each through lgm_conv_wgrad - the F(4x4) kernel on 2 x 2-tile groups
    gc_rows = []
    for l, gw in zip(layers, gws):
        ops.conv_wgrad(l[0], l[1], l[2], gw.data_ptr(), 0.0, None, defer=gc_rows)
        assert "wino4_wgrad_kernel" in ops.lib()._dll.lgm_last_kernel().decode()
    ops.wgrad_reduce_batch(gc_rows, dev)
    for i in range(2):
        parity(f"wide layer {i} alone, F(4x4)", maxerr(gws[i], refs[i]), 1e-5)
