"""GPU: every HIP op (through the C-ABI) against a plain fp32 CPU reference of the same op.
Tolerances: fp32 results within 1e-4 relative (north_star); integer/index work bit-exact."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda", 0)


def nhwc(x, dev, cpad=None, extra=0, offset=0):
    """NCHW cpu -> NHWC cuda tensor, optionally as a channel slice [offset:offset+C] of a wider buffer."""
    B, C, H, W = x.shape
    Cp = cpad or ((C + 3) // 4 * 4)
    buf = torch.full((B, H, W, Cp + extra), float("nan"), device=dev)
    view = buf[..., offset:offset + Cp]
    view.zero_()
    view[..., :C] = x.permute(0, 2, 3, 1).to(dev)
    return view


def nchw(t, C=None):
    t = t.detach().cpu()
    if C is not None:
        t = t[..., :C]
    return t.permute(0, 3, 1, 2).contiguous()


def phys_weight(w, dev):
    """OIHW cpu -> physical [Np][T][Cp] cuda"""
    N, C, KH, KW = w.shape
    Np, Cp = (N + 3) // 4 * 4, (C + 3) // 4 * 4
    p = torch.zeros(Np, KH * KW, Cp, device=dev)
    p[:N, :, :C] = w.permute(0, 2, 3, 1).reshape(N, KH * KW, C).to(dev)
    return p


def vec(v, dev):
    n = (v.numel() + 3) // 4 * 4
    p = torch.zeros(n, device=dev)
    p[:v.numel()] = v.reshape(-1).to(dev)
    return p


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1),
    (3, 8, 8, 192, 128, 3, 1, 1),
    (2, 32, 32, 3, 64, 7, 1, 3),      # init_conv (padded input channels)
    (2, 16, 16, 64, 3, 1, 1, 0),      # final_conv (3 outputs)
    (2, 8, 8, 256, 128, 1, 1, 0),
    (2, 16, 16, 32, 64, 4, 2, 1),     # strided 4x4 (DCGAN D / VQ-VAE encoder)
    (5, 4, 4, 512, 1, 4, 1, 0),       # DCGAN critic head
    (1, 10, 6, 20, 36, 3, 1, 1),      # ragged sizes
    (4, 16, 16, 128, 256, 3, 1, 1),   # N > 64: 128-wide tiles
    (40, 32, 32, 64, 64, 3, 1, 1),    # enough rows for the 128x64 tile
    (24, 16, 16, 128, 128, 3, 1, 1),  # 128x128 tile
    (11, 4, 4, 256, 256, 3, 1, 1),    # 3x3 patch kernel: 8 images per tile, ragged batch
    (1, 64, 64, 64, 64, 3, 1, 1),     # 3x3 patch kernel: 64x64 images (two column tiles)
    (5, 8, 8, 96, 64, 3, 1, 1),       # 3x3 patch kernel with 32-channel chunks (C % 64 != 0)
    (3, 32, 32, 128, 64, 3, 1, 1),    # 3x3 patch kernel, two channel chunks
    (13, 32, 32, 64, 384, 1, 1, 0),   # resident-tile 1x1 kernel (to_qkv shape), 128-row tiles, ragged M
    (50, 16, 16, 384, 128, 1, 1, 0),  # resident-tile 1x1 kernel, 64-row tiles (K = 384)
    (16, 32, 32, 64, 384, 1, 1, 0),   # streaming 1x1 weight gradient, 128x64 block of gw, three n-tiles
    (20, 32, 32, 128, 64, 1, 1, 0),   # streaming 1x1 weight gradient, 64x128 block, ragged split
    (64, 16, 16, 128, 128, 1, 1, 0),  # streaming 1x1 weight gradient, 128x128 block
    (72, 16, 16, 192, 64, 1, 1, 0),   # streaming 1x1 weight gradient, 64x64 blocks, three k-tiles
    (64, 32, 32, 128, 128, 1, 1, 0),  # resident-weight streaming 1x1 conv, K = 128 (forward and input gradient)
    (48, 32, 32, 192, 128, 1, 1, 0),  # resident-weight streaming 1x1 conv, K = 192
    (64, 32, 32, 256, 64, 1, 1, 0),   # resident-weight streaming 1x1 conv, K = 256 forward, K = 64 input gradient
    (64, 32, 32, 128, 64, 1, 1, 0),   # resident-weight streaming 1x1 conv, 64-column slice (N = 64), K = 128
    (64, 32, 32, 64, 64, 1, 1, 0),    # resident-weight streaming 1x1 conv, 64-column slice, K = 64
    # DCGAN critic / generator shapes: uniform-tap chunks (raw buffer loads), residue-class input gradient,
    # shift/mask weight-gradient gather, image-end kernels
    (3, 64, 64, 3, 64, 4, 2, 1),      # image end: general decode forward, smalln input gradient with 16 lanes per pixel
    (2, 64, 64, 3, 128, 4, 2, 1),     # image end of the generator: smalln input gradient with 32 lanes per pixel
    (2, 32, 32, 64, 128, 4, 2, 1),
    (5, 16, 16, 128, 256, 4, 2, 1),   # ragged batch
    (1, 8, 8, 256, 512, 4, 2, 1),     # 16 output pixels: one partial pixel chunk in the weight gradient, split-K forward
    (3, 12, 20, 32, 64, 4, 2, 1),     # maps that are not powers of two: uniform taps yes, shift/mask gather no
    (2, 8, 8, 512, 1024, 4, 2, 1),    # 128-wide column tiles, long reduction
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case):
    _conv_fwd_dgrad_wgrad(dev, case)


def _clear_plan_caches():
    from lgm_hip import ops
    for n in ("_WINO4_OK", "_WINO_WS", "_EPI_STATS", "_PAIR_OK", "_WG2_OK", "_WG2_WS", "_WINO_OK"):
        if hasattr(ops, n):
            getattr(ops, n).clear()


@pytest.mark.parametrize("margin", [16, 40])
@pytest.mark.parametrize("case", [(40, 32, 32, 64, 64, 3, 1, 1), (24, 16, 16, 128, 128, 3, 1, 1), (11, 4, 4, 256, 256, 3, 1, 1),
                                  (3, 32, 32, 128, 64, 3, 1, 1), (64, 32, 32, 128, 128, 1, 1, 0), (20, 32, 32, 128, 64, 1, 1, 0),
                                  (64, 8, 8, 256, 256, 3, 1, 1), (32, 4, 4, 512, 512, 3, 1, 1)])
def test_conv_with_launch_plans_that_leave_cus_to_a_collective(dev, case, margin):
    """lgm_set_cu_margin (the default of a rank when WORLD_SIZE > 1 is 16): split-K, slab and persistent-range plans sized for
    256 - margin CUs give the same forward, input gradient and weight gradient."""
    from lgm_hip import ops
    _clear_plan_caches()
    base = ops.lib().lgm_cu_margin()                       # 0 on one GPU (LGM_CU_MARGIN / WORLD_SIZE > 1 set another default)
    ops.lib().lgm_set_cu_margin(margin)
    try:
        assert ops.lib().lgm_cu_margin() == margin
        _conv_fwd_dgrad_wgrad(dev, case)
    finally:
        ops.lib().lgm_set_cu_margin(-1)
        _clear_plan_caches()
    from lgm_hip._lib import LgmArgumentError
    with pytest.raises(LgmArgumentError):                 # more than half of the chip: refused
        ops.lib().lgm_set_cu_margin(200)
    assert ops.lib().lgm_cu_margin() == base


def _conv_fwd_dgrad_wgrad(dev, case):
    from lgm_hip import ops
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    y_ref = F.conv2d(x, w, b, stride=s, padding=p)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    Cp, Np = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    geom = ops.make_geom(B, H, W, Cp, Np, k, k, s, p)
    xd = nhwc(x.detach(), dev, extra=8, offset=4)          # exercise pitch != C
    wd, bd = phys_weight(w.detach(), dev), vec(b.detach(), dev)
    yd = torch.empty(B, geom.Ho, geom.Wo, Np + 4, device=dev)[..., :Np]
    ops.conv_xy(geom, xd, wd.data_ptr(), bd.data_ptr(), None, yd)
    assert rel(nchw(yd, Cout), y_ref) < RTOL
    # residual epilogue aliasing the output
    ops.conv_xy(geom, xd, wd.data_ptr(), None, yd, yd)
    assert rel(nchw(yd, Cout), 2 * y_ref - b.detach()[None, :, None, None]) < RTOL
    # input gradient
    gyd = nhwc(gy, dev, extra=4, offset=0)
    gxd = torch.empty(B, H, W, Cp, device=dev)
    ops.conv_yx(geom, gyd, wd.data_ptr(), None, None, gxd)
    assert rel(nchw(gxd, Cin), x.grad) < RTOL
    # input gradient again through the transposed-weight path ([Cw][T][Nw] copy made by one batched launch)
    wt = torch.zeros_like(wd)
    tbl = torch.tensor([[0, Np, k * k, Cp, 0]], dtype=torch.int32, device=dev)
    ops.lib().lgm_transpose_weights(wd.data_ptr(), wt.data_ptr(), tbl.data_ptr(), 1,
                                    ((Np + 31) // 32) * ((Cp + 31) // 32) * k * k, ops.stream())
    assert torch.equal(wt.view(Cp, k * k, Np), wd.permute(2, 1, 0).contiguous())
    gxt = torch.empty(B, H, W, Cp, device=dev)
    ops.conv_yx(geom, gyd, wd.data_ptr(), None, None, gxt, wt.data_ptr())
    assert rel(nchw(gxt, Cin), x.grad) < RTOL
    # weight gradient (+ accumulate with beta = 1) and bias gradient
    gw = torch.zeros_like(wd)
    ops.conv_wgrad(geom, gyd, xd, gw.data_ptr(), 0.0)
    gw_ref = phys_weight(w.grad, dev)
    assert rel(gw, gw_ref) < RTOL
    ops.conv_wgrad(geom, gyd, xd, gw.data_ptr(), 1.0)
    assert rel(gw, 2 * gw_ref) < RTOL
    gb = torch.zeros(Np, device=dev)
    ops.colsum(gyd, gb.data_ptr(), 0.0)
    assert rel(gb[:Cout], b.grad) < RTOL
    # bias gradient fused into the wgrad kernel (+ accumulate)
    gw_f, gb_f = torch.zeros_like(wd), torch.zeros(Np, device=dev)
    ops.conv_wgrad(geom, gyd, xd, gw_f.data_ptr(), 0.0, gb_f.data_ptr())
    assert rel(gw_f, gw_ref) < RTOL and rel(gb_f[:Cout], b.grad) < RTOL
    ops.conv_wgrad(geom, gyd, xd, gw_f.data_ptr(), 1.0, gb_f.data_ptr())
    assert rel(gw_f, 2 * gw_ref) < RTOL and rel(gb_f[:Cout], 2 * b.grad) < RTOL
    # wgrad determinism: two runs are bit-identical
    gw2 = torch.zeros_like(wd)
    ops.conv_wgrad(geom, gyd, xd, gw2.data_ptr(), 0.0)
    gw3 = torch.zeros_like(wd)
    ops.conv_wgrad(geom, gyd, xd, gw3.data_ptr(), 0.0)
    assert torch.equal(gw2, gw3)


@pytest.mark.parametrize("case", [(2, 4, 4, 100, 256, 4, 1, 0), (2, 8, 8, 64, 32, 4, 2, 1), (3, 7, 7, 128, 1, 4, 2, 1)])
def test_conv_transpose(dev, case):
    """ConvTranspose2d forward = Y->X pass; its backward = X->Y pass + wgrad with swapped roles."""
    from lgm_hip import ops
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cin, Cout, k, k, generator=g) / math.sqrt(Cin)).requires_grad_(True)
    y_ref = F.conv_transpose2d(x, w, None, stride=s, padding=p)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    Ho, Wo = y_ref.shape[-2:]
    Np, Cp = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    geom = ops.make_geom(B, Ho, Wo, Cp, Np, k, k, s, p)
    assert (geom.Ho, geom.Wo) == (H, W)
    xd, gyd = nhwc(x.detach(), dev), nhwc(gy, dev)
    wd = phys_weight(w.detach(), dev)     # [Cin][T][Cout]: same formula with N = Cin, C = Cout
    yd = torch.empty(B, Ho, Wo, Cp, device=dev)
    ops.conv_yx(geom, xd, wd.data_ptr(), None, None, yd)
    assert rel(nchw(yd, Cout), y_ref) < RTOL
    gxd = torch.empty(B, H, W, Np, device=dev)
    ops.conv_xy(geom, gyd, wd.data_ptr(), None, None, gxd)
    assert rel(nchw(gxd, Cin), x.grad) < RTOL
    gw = torch.zeros_like(wd)
    ops.conv_wgrad(geom, xd, gyd, gw.data_ptr(), 0.0)
    assert rel(gw, phys_weight(w.grad, dev)) < RTOL


@pytest.mark.parametrize("B,S", [(3, 16), (5, 32), (1, 8), (7, 5)])
def test_narrow_1x1_convolution(dev, B, S):
    """final_conv (reference ddpm.py:422: Conv2d(64, 3, 1)) forward and input gradient on the streaming dot-product kernels
    (narrow1x1_*: 16 lanes of a DPP row per pixel) instead of an MFMA tile that idles 15/16 of its columns; ragged pixel
    counts included.  Against float64 and against the implicit-GEMM path (LGM_NO_NARROW1X1 is read once: compared by value)."""
    from lgm_hip import ops
    g = torch.Generator().manual_seed(B * 100 + S)
    x = torch.randn(B, 64, S, S, generator=g)
    w = torch.randn(3, 64, 1, 1, generator=g) * 0.2
    b = torch.randn(3, generator=g)
    gy = torch.randn(B, 3, S, S, generator=g)
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, w.double(), b.double())
    yr.backward(gy.double())
    xd, gyd = nhwc(x, dev, extra=4), nhwc(gy, dev)
    wd, bd = phys_weight(w, dev), vec(b, dev)
    geom = ops.make_geom(B, S, S, 64, 4, 1, 1, 1, 0)
    yd = torch.full((B, S, S, 4), 9.0, device=dev)
    ops.conv_xy(geom, xd, wd.data_ptr(), bd.data_ptr(), None, yd)
    assert ops.lib()._dll.lgm_last_kernel().decode().startswith("narrow1x1_fwd_kernel")
    assert rel(nchw(yd, 3), yr) < RTOL and float(yd[..., 3].abs().max()) == 0.0        # the pad lane stays zero
    gxd = torch.full((B, S, S, 64), 9.0, device=dev)
    ops.conv_yx(geom, gyd, wd.data_ptr(), None, None, gxd)
    assert ops.lib()._dll.lgm_last_kernel().decode().startswith("narrow1x1_dgrad_kernel")
    assert rel(nchw(gxd), xr.grad) < RTOL


@pytest.mark.parametrize("layers", [
    [(64, 32, 128, 64, True), (64, 32, 128, 64, False), (128, 32, 128, 64, True)],             # tile 64 x 128, three layers
    [(128, 16, 128, 256, True), (128, 16, 256, 256, True)],                                    # tile 128 x 128, a pair
    [(64, 32, 64, 64, True), (64, 32, 64, 64, True), (128, 32, 64, 64, False), (96, 32, 64, 64, True)]])   # tile 64 x 64, four
def test_grouped_1x1_weight_gradients(dev, layers):
    """lgm_wgrad1x1_group: the weight / bias gradients of 2 ... 4 1x1 convolutions in ONE launch (block ranges of one grid,
    every layer on its share of the chip) against float64, overwrite and accumulate, through the batched slab reducer."""
    from lgm_hip import ops
    g = torch.Generator().manual_seed(len(layers))
    ents, refs, keep = [], [], []
    for (B, S, C, N, bias) in layers:
        x = torch.randn(B, C, S, S, generator=g)
        gy = torch.randn(B, N, S, S, generator=g)
        xd, gyd = nhwc(x, dev, extra=4), nhwc(gy, dev)
        geom = ops.make_geom(B, S, S, C, N, 1, 1, 1, 0)
        gw = torch.full((N, 1, C), 3.0, device=dev)
        gb = torch.full((N,), -1.0, device=dev) if bias else None
        ents.append((geom, gyd, xd, gw.data_ptr(), 0.0, gb.data_ptr() if bias else None))
        refs.append((torch.einsum("bnhw,bchw->nc", gy.double(), x.double()), gy.double().sum((0, 2, 3))))
        keep.append((gw, gb, xd, gyd, geom))
    assert ops.wgrad1x1_group_supported([e[0] for e in ents])
    rows = []
    ops.conv_wgrad1x1_group(ents, rows)
    assert ops.lib()._dll.lgm_last_kernel().decode().startswith("wgrad1x1_group_kernel")
    ops.wgrad_reduce_batch(rows, dev)
    for (gw, gb, _, _, _), (rw, rb) in zip(keep, refs):
        assert rel(gw[:, 0, :], rw) < RTOL
        if gb is not None:
            assert rel(gb, rb) < RTOL
    # beta = 1: accumulates
    rows = []
    ops.conv_wgrad1x1_group([(e[0], e[1], e[2], e[3], 1.0, e[5]) for e in ents], rows)
    ops.wgrad_reduce_batch(rows, dev)
    for (gw, gb, _, _, _), (rw, rb) in zip(keep, refs):
        assert rel(gw[:, 0, :], 2 * rw) < RTOL
        if gb is not None:
            assert rel(gb, 2 * rb) < RTOL
    # the same layers one by one give the same numbers to rounding
    for (geom, gyd, xd, _, _, gbp), (gw, gb, _, _, _) in zip(ents, keep):
        gw1 = torch.zeros_like(gw)
        ops.conv_wgrad(geom, gyd, xd, gw1.data_ptr(), 0.0, None)
        assert rel(gw, 2 * gw1) < 1e-5


@pytest.mark.parametrize("shape", [(3, 64, 16, 16, 8), (2, 128, 8, 8, 8), (2, 512, 4, 4, 8), (2, 16, 8, 8, 8), (5, 256, 4, 4, 8),
                                   (2, 64, 32, 32, 8),     # one-pass kernels, 1024-thread blocks
                                   (1, 64, 64, 64, 8)])    # slice too large for the registers: two-pass kernels
@pytest.mark.parametrize("film", [True, False])
def test_groupnorm_film_silu(dev, shape, film):
    from lgm_hip import ops
    B, C, H, W, G = shape
    g = torch.Generator().manual_seed(C + H + int(film))
    x = (torch.randn(B, C, H, W, generator=g) * 1.7 + 0.5).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    ss = (0.5 * torch.randn(B, 2 * C, generator=g)).requires_grad_(True)
    res = torch.randn(B, C, H, W, generator=g)
    z = F.group_norm(x, G, gamma, beta, 1e-5)
    if film:
        sc, sh = ss[:, :C, None, None], ss[:, C:, None, None]
        z = z * (sc + 1) + sh
    y_ref = F.silu(z) + res
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd, resd, gyd = nhwc(x.detach(), dev, extra=4), nhwc(res, dev), nhwc(gy, dev, extra=8, offset=8)
    gd, bd = vec(gamma.detach(), dev), vec(beta.detach(), dev)
    ssd = torch.zeros(B, 2 * C + 12, device=dev)[:, 4:4 + 2 * C]
    ssd.copy_(ss.detach())
    yd = torch.empty(B, H, W, C, device=dev)
    sv = ops.gn_fwd(xd, G, 1e-5, gd.data_ptr(), bd.data_ptr(), ssd if film else None, True, resd, yd)
    assert rel(nchw(yd), y_ref) < RTOL
    gxd = torch.zeros(B, H, W, C, device=dev)
    gg, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    gss = torch.zeros(B, 2 * C, device=dev)
    ops.gn_bwd(xd, gyd, G, gd.data_ptr(), bd.data_ptr(), ssd if film else None, True, sv, gxd, False,
               gg.data_ptr(), gb.data_ptr(), 0.0, gss if film else None, 0.0)
    assert rel(nchw(gxd), x.grad) < RTOL
    assert rel(gg, gamma.grad) < RTOL and rel(gb, beta.grad) < RTOL
    if film:
        assert rel(gss, ss.grad) < RTOL
    # accumulate into gx
    ops.gn_bwd(xd, gyd, G, gd.data_ptr(), bd.data_ptr(), ssd if film else None, True, sv, gxd, True,
               gg.data_ptr(), gb.data_ptr(), 1.0, None, 0.0)
    assert rel(nchw(gxd), 2 * x.grad) < RTOL and rel(gg, 2 * gamma.grad) < RTOL
    # deferred form: gamma/beta gradients through the batched slab reducer (or complete at once when the
    # two-pass kernels ran and nothing was deferred)
    rows = []
    gx2 = torch.zeros(B, H, W, C, device=dev)
    gg2, gb2 = torch.full((C,), 3.0, device=dev), torch.full((C,), -2.0, device=dev)
    ops.gn_bwd(xd, gyd, G, gd.data_ptr(), bd.data_ptr(), ssd if film else None, True, sv, gx2, False,
               gg2.data_ptr(), gb2.data_ptr(), 0.0, None, 0.0, defer=rows)
    ops.wgrad_reduce_batch(rows, dev)
    assert rel(nchw(gx2), x.grad) < RTOL and rel(gg2, gamma.grad) < RTOL and rel(gb2, beta.grad) < RTOL
    # lgm_gn_bwd_add: the same results, and a second tensor (a channel slice of a wider one) receives += gy in the launch -
    # bit for bit what a separate addition gives; immediate and deferred forms
    for deferred in (False, True):
        rows = []
        wide = torch.randn(B, H, W, C + 8, device=dev)
        tgt = wide[..., 4:4 + C]
        want = tgt + gyd
        keep = wide.clone()
        gx3 = torch.zeros(B, H, W, C, device=dev)
        gg3, gb3 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.gn_bwd(xd, gyd, G, gd.data_ptr(), bd.data_ptr(), ssd if film else None, True, sv, gx3, False,
                   gg3.data_ptr(), gb3.data_ptr(), 0.0, None, 0.0, defer=rows if deferred else None, add_gy_to=tgt)
        if deferred:
            ops.wgrad_reduce_batch(rows, dev)
        assert torch.equal(gx3, gx2) and rel(gg3, gamma.grad) < RTOL and rel(gb3, beta.grad) < RTOL
        assert torch.equal(tgt, want)
        assert torch.equal(wide[..., :4], keep[..., :4]) and torch.equal(wide[..., 4 + C:], keep[..., 4 + C:])


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (3, 128, 8, 8), (2, 512, 4, 4), (2, 16, 8, 8), (130, 256, 4, 4)])
def test_rmsnorm(dev, shape):
    from lgm_hip import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    gp = (1 + 0.2 * torch.randn(1, C, 1, 1, generator=g)).requires_grad_(True)
    res = torch.randn(B, C, H, W, generator=g)
    y_ref = F.normalize(x, dim=1) * gp * (C ** 0.5) + res
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd, resd, gyd = nhwc(x.detach(), dev, extra=4, offset=4), nhwc(res, dev), nhwc(gy, dev)
    gpd = vec(gp.detach(), dev)
    yd = torch.empty(B, H, W, C, device=dev)
    ops.rmsnorm_fwd(xd, gpd.data_ptr(), resd, yd)
    assert rel(nchw(yd), y_ref) < RTOL
    gxd = torch.empty(B, H, W, C, device=dev)
    gg = torch.zeros(C, device=dev)
    ops.rmsnorm_bwd(xd, gyd, gpd.data_ptr(), gxd, False, gg.data_ptr(), 0.0)
    assert rel(nchw(gxd), x.grad) < RTOL
    assert rel(gg, gp.grad.reshape(-1)) < RTOL
    # residual-branch gradient added in the same pass (+ accumulate into gx)
    rg = torch.randn(B, C, H, W, generator=g)
    ops.rmsnorm_bwd(xd, gyd, gpd.data_ptr(), gxd, True, gg.data_ptr(), 1.0, res=nhwc(rg, dev, extra=4))
    assert rel(nchw(gxd), 2 * x.grad + rg) < RTOL and rel(gg, 2 * gp.grad.reshape(-1)) < RTOL


def _lin_attn_core(q, k, v, mem, scale):
    """q,k,v: [B, h, d, n]; mem: [2, h, d, M] (reference ddpm.py:227-237)."""
    b = q.shape[0]
    mk, mv = (m.unsqueeze(0).expand(b, -1, -1, -1) for m in mem)
    k = torch.cat((mk, k), dim=-1)
    v = torch.cat((mv, v), dim=-1)
    q = q.softmax(dim=-2) * scale
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    return torch.einsum("bhde,bhdn->bhen", ctx, q)


@pytest.mark.parametrize("shape", [(2, 4, 16, 16), (3, 4, 8, 8), (2, 2, 5, 7), (1, 4, 32, 32)])
def test_linear_attention_core(dev, shape):
    from lgm_hip import ops
    B, heads, H, W = shape
    d, M, n = 32, 4, H * W
    hidden = heads * d
    g = torch.Generator().manual_seed(n)
    qkv = (torch.randn(B, 3 * hidden, H, W, generator=g) * 1.5).requires_grad_(True)
    mem = torch.randn(2, heads, d, M, generator=g, requires_grad=True)
    q, k, v = (t.reshape(B, heads, d, n) for t in qkv.chunk(3, dim=1))
    out_ref = _lin_attn_core(q, k, v, mem, d ** -0.5).reshape(B, hidden, H, W)
    gout = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gout)
    qd, god = nhwc(qkv.detach(), dev), nhwc(gout, dev)
    memd = vec(mem.detach(), dev)
    od = torch.empty(B, H, W, hidden, device=dev)
    ctx, kstat = ops.linattn_fwd(qd, memd.data_ptr(), heads, d, M, od)
    assert rel(nchw(od), out_ref) < RTOL
    gq = torch.empty(B, H, W, 3 * hidden, device=dev)
    gm = torch.zeros(mem.numel(), device=dev)
    ops.linattn_bwd(qd, memd.data_ptr(), god, ctx, kstat, heads, d, M, gq, gm.data_ptr(), 0.0)
    assert rel(nchw(gq), qkv.grad) < RTOL
    assert rel(gm, mem.grad.reshape(-1)) < RTOL


@pytest.mark.parametrize("case", [(2, 64, 16, 16), (3, 64, 5, 7), (2, 128, 8, 8), (1, 256, 4, 4), (40, 64, 32, 32),
                                  (300, 256, 4, 4)])
def test_rmsnorm_and_to_qkv_in_one_launch(dev, case):
    """lgm_rms_qkv_fused against F.normalize(x, dim=1) * g * sqrt(C) -> bias-free 1x1 convolution (ddpm.py:115-121,
    :224-225)."""
    from lgm_hip import ops
    B, C, H, W = case
    N = 384
    g = torch.Generator().manual_seed(31 * C + H * W + B)
    x = torch.randn(B, C, H, W, generator=g) * 2.0
    gn = 1.0 + 0.3 * torch.randn(C, generator=g)
    w = torch.randn(N, C, generator=g) / math.sqrt(C)
    xn_ref = F.normalize(x, dim=1) * gn.view(1, C, 1, 1) * C ** 0.5
    qkv_ref = torch.einsum("oc,bchw->bohw", w, xn_ref)
    xd, gd, wd = nhwc(x, dev, extra=4), vec(gn, dev), w.contiguous().to(dev)
    r = ops.rms_qkv_fused(xd, gd.data_ptr(), wd.data_ptr(), N, any_size=True)
    assert r is not None
    xn, qkv = r
    assert rel(nchw(xn), xn_ref) < RTOL
    assert rel(nchw(qkv), qkv_ref) < RTOL
    # against the separate RMSNorm launch: the same arithmetic, only the channel sum of squares in another order
    y = torch.empty_like(xn)
    ops.rmsnorm_fwd(xd, gd.data_ptr(), None, y)
    assert rel(xn, y) < 1e-6


@pytest.mark.parametrize("case", [(2, 64, 16, 16), (3, 64, 20, 20), (2, 128, 8, 8), (1, 256, 5, 7), (5, 64, 32, 32),
                                  (70, 128, 16, 16)])
def test_linear_attention_forward_with_fused_tail(dev, case):
    """lgm_linattn_fwd_fused: softmax_d(q) ctx -> to_out[0] -> RMSNorm -> + x in one launch, against the reference's
    arithmetic (ddpm.py:229-239: to_out = Sequential(Conv2d(hidden, dim, 1), RMSNorm(dim)); RMSNorm :115-121)."""
    from lgm_hip import ops
    B, C, H, W = case
    heads, d, M, n = 4, 32, 4, H * W
    hidden = heads * d
    g = torch.Generator().manual_seed(77 * C + n + B)
    qkv = torch.randn(B, 3 * hidden, H, W, generator=g) * 1.5
    mem = torch.randn(2, heads, d, M, generator=g)
    x = torch.randn(B, C, H, W, generator=g)
    wout = torch.randn(C, hidden, generator=g) / math.sqrt(hidden)
    bout = torch.randn(C, generator=g) * 0.1
    gn = 1.0 + 0.2 * torch.randn(C, generator=g)
    q, k, v = (t.reshape(B, heads, d, n) for t in qkv.chunk(3, dim=1))
    out_ref = _lin_attn_core(q, k, v, mem, d ** -0.5).reshape(B, hidden, H, W)
    o2_ref = torch.einsum("ck,bkhw->bchw", wout, out_ref) + bout.view(1, C, 1, 1)
    y_ref = F.normalize(o2_ref, dim=1) * gn.view(1, C, 1, 1) * C ** 0.5 + x
    qd, xd = nhwc(qkv, dev), nhwc(x, dev, extra=4)
    memd, wd, bd, gd = vec(mem, dev), wout.contiguous().to(dev), vec(bout, dev), vec(gn, dev)
    assert ops.linattn_fwd_fused_ok(heads, d, C, qd, xd, wd.data_ptr(), bd.data_ptr(), gd.data_ptr(), any_size=True)
    ao = torch.full((B, H, W, hidden), float("nan"), device=dev)
    o2 = torch.full((B, H, W, C), float("nan"), device=dev)
    y = torch.full((B, H, W, C + 4), float("nan"), device=dev)[..., :C]
    ctx, kstat = ops.linattn_fwd_fused(qd, memd.data_ptr(), heads, d, M, wd.data_ptr(), bd.data_ptr(), gd.data_ptr(), xd,
                                       ao, o2, y)
    assert rel(nchw(ao), out_ref) < RTOL
    assert rel(nchw(o2), o2_ref) < RTOL
    assert rel(nchw(y), y_ref) < RTOL
    # the saved context is what the unfused forward leaves (the backward pass reads it)
    od = torch.empty(B, H, W, hidden, device=dev)
    ctx2, kstat2 = ops.linattn_fwd(qd, memd.data_ptr(), heads, d, M, od)
    assert torch.equal(ctx, ctx2) and torch.equal(kstat, kstat2)


@pytest.mark.parametrize("case", [(2, 64, 16, 16, False), (3, 64, 20, 20, True), (2, 64, 8, 8, True), (5, 64, 32, 32, True),
                                  (1, 64, 20, 12, False), (130, 64, 16, 16, True)])
def test_linear_attention_backward_with_fused_tail(dev, case, monkeypatch):
    """lgm_linattn_bwd_fused (gq / gk / gv stay on the chip; to_qkv's input gradient, and for 64 channels its weight
    gradient, come out of the same launch) against autograd through to_qkv + the attention core (ddpm.py:214-239)."""
    from lgm_hip import ops
    monkeypatch.setattr(ops, "LA_FUSED", True)
    monkeypatch.setattr(ops, "LA_FUSED_MIN_ITEMS", 0)   # the size gate is a performance choice, not a limit
    B, C, H, W, deferred = case
    heads, d, M, n = 4, 32, 4, H * W
    hidden = heads * d
    g = torch.Generator().manual_seed(1000 * C + n + B)
    xn = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    w = (torch.randn(3 * hidden, C, generator=g) * (1.5 / math.sqrt(C))).requires_grad_(True)
    mem = torch.randn(2, heads, d, M, generator=g, requires_grad=True)
    qkv = torch.einsum("oc,bchw->bohw", w, xn)
    q, k, v = (t.reshape(B, heads, d, n) for t in qkv.chunk(3, dim=1))
    out_ref = _lin_attn_core(q, k, v, mem, d ** -0.5).reshape(B, hidden, H, W)
    gout = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gout)
    qd, god, xd = nhwc(qkv.detach(), dev), nhwc(gout, dev, extra=4), nhwc(xn.detach(), dev)
    memd, wd = vec(mem.detach(), dev), w.detach().t().contiguous().to(dev)      # transposed copy [C][3 * hidden]
    od = torch.empty(B, H, W, hidden, device=dev)
    ctx, kstat = ops.linattn_fwd(qd, memd.data_ptr(), heads, d, M, od)
    assert ops.linattn_bwd_fused_ok(heads, d, C, qd, god, xd, wd.data_ptr(), wd.data_ptr(), memd.data_ptr())
    gxn = torch.full((B, H, W, C), float("nan"), device=dev)
    gw = torch.full((3 * hidden, C), 0.5, device=dev)          # beta = 1 on top of existing content
    gm = torch.full((mem.numel(),), 0.25, device=dev)
    dw, dm = ([], []) if deferred else (None, None)
    ops.linattn_bwd_fused(qd, memd.data_ptr(), god, ctx, kstat, xd, wd.data_ptr(), heads, d, M, gxn, gw.data_ptr(), 1.0, dw,
                          gm.data_ptr(), 1.0, dm)
    if deferred:
        rows = dw + dm
        assert len(rows) == 2
        ops.wgrad_reduce_batch(rows, dev)
    assert rel(nchw(gxn), xn.grad) < RTOL
    assert rel(gm - 0.25, mem.grad.reshape(-1)) < RTOL
    assert rel(gw - 0.5, w.grad) < RTOL


@pytest.mark.parametrize("shape", [(2, 4, 4, 4), (3, 4, 8, 8), (2, 2, 3, 5), (1, 4, 8, 16)])
def test_full_attention_core(dev, shape):
    from lgm_hip import ops
    B, heads, H, W = shape
    d, M, n = 32, 4, H * W
    hidden = heads * d
    g = torch.Generator().manual_seed(n + 1)
    qkv = torch.randn(B, 3 * hidden, H, W, generator=g, requires_grad=True)
    mem = torch.randn(2, heads, M, d, generator=g, requires_grad=True)
    q, k, v = (t.reshape(B, heads, d, n).transpose(-1, -2) for t in qkv.chunk(3, dim=1))
    mk, mv = (m.unsqueeze(0).expand(B, -1, -1, -1) for m in mem)
    k = torch.cat((mk, k), dim=-2)
    v = torch.cat((mv, v), dim=-2)
    attn = (torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5).softmax(dim=-1)
    out_ref = torch.einsum("bhij,bhjd->bhid", attn, v).transpose(-1, -2).reshape(B, hidden, H, W)
    gout = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gout)
    qd, god = nhwc(qkv.detach(), dev), nhwc(gout, dev)
    memd = vec(mem.detach(), dev)
    od = torch.empty(B, H, W, hidden, device=dev)
    lse = ops.attn_fwd(qd, memd.data_ptr(), heads, d, M, od)
    assert rel(nchw(od), out_ref) < RTOL
    gq = torch.empty(B, H, W, 3 * hidden, device=dev)
    gm = torch.zeros(mem.numel(), device=dev)
    ops.attn_bwd(qd, memd.data_ptr(), od, god, lse, heads, d, M, gq, gm.data_ptr(), 0.0)
    assert rel(nchw(gq), qkv.grad) < RTOL
    assert rel(gm, mem.grad.reshape(-1)) < RTOL


def test_elementwise_kernels(dev):
    from lgm_hip import ops
    g = torch.Generator().manual_seed(5)
    # sinusoidal embedding: known answers from the reference and the oracle formula
    t = torch.tensor([0, 1, 17, 999])
    pe = torch.empty(4, 64, device=dev)
    ops.posemb(t.to(dev), 64, 10000.0, pe)
    half = 32
    f = torch.exp(torch.arange(half) * -(math.log(10000.0) / (half - 1)))
    ref = torch.cat(((t[:, None] * f).sin(), (t[:, None] * f).cos()), dim=-1)
    err = float((pe.cpu() - ref).abs().max())
    print(f"posemb max abs err vs host formula: {err:.2e}")
    assert err < 1e-6        # host-computed frequency table: only sin/cos rounding is left
    assert abs(pe[1, 0].item() - 0.84147096) < 1e-6 and abs(pe[1, 32].item() - 0.54030234) < 1e-6
    # activations
    x = torch.randn(37, 64, generator=g, requires_grad=True)
    gy = torch.randn(37, 64, generator=g)
    for act, fn in ((ops.ACT_SILU, F.silu), (ops.ACT_GELU, F.gelu), (ops.ACT_RELU, F.relu),
                    (ops.ACT_LRELU, lambda v: F.leaky_relu(v, 0.2)), (ops.ACT_TANH, torch.tanh)):
        x.grad = None
        y = fn(x)
        y.backward(gy)
        xd, gyd = x.detach().to(dev), gy.to(dev)
        yd, gxd = torch.empty_like(xd), torch.empty_like(xd)
        ops.act_fwd(xd, None, None, yd, act, 0.2)
        ops.act_bwd(xd, None, gyd, gxd, False, act, 0.2)
        assert rel(yd, y) < 1e-5 and rel(gxd, x.grad) < 1e-5
    # upsample / unshuffle round trips against torch / einops-equivalent formulas
    a = torch.randn(2, 8, 5, 6, generator=g)
    ad = nhwc(a, dev)
    up = torch.empty(2, 10, 12, 8, device=dev)
    ops.upsample2x_fwd(ad, up)
    assert torch.equal(nchw(up), F.interpolate(a, scale_factor=2, mode="nearest"))
    gup = torch.randn(2, 8, 10, 12, generator=g)
    gad = torch.empty(2, 5, 6, 8, device=dev)
    ops.upsample2x_bwd(nhwc(gup, dev), gad, False)
    assert rel(nchw(gad), F.avg_pool2d(gup, 2) * 4) < 1e-6
    hi = torch.randn(2, 8, 6, 4, generator=g)
    lo = torch.empty(2, 3, 2, 32, device=dev)
    ops.pixel_unshuffle(nhwc(hi, dev), lo, inverse=False)
    ref = hi.reshape(2, 8, 3, 2, 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(2, 32, 3, 2)
    assert torch.equal(nchw(lo), ref)
    back = torch.empty(2, 6, 4, 8, device=dev)
    ops.pixel_unshuffle(back, lo, inverse=True)
    assert torch.equal(nchw(back), hi)


def test_qsample_loss_and_adam(dev):
    from lgm_hip import ops
    from oracle import diffusion as OD
    from oracle import optim as OO
    g = torch.Generator().manual_seed(9)
    bufs = OD.diffusion_buffers(1000)
    B, C, S = 5, 3, 8
    img = torch.rand(B, C, S, S, generator=g)
    noise = torch.randn(B, C, S, S, generator=g)
    t = torch.tensor([0, 17, 500, 998, 999])
    x0 = img * 2 - 1
    xt_ref, v_ref = OD.q_sample(bufs, x0, t, noise), OD.predict_v(bufs, x0, t, noise)
    L = ops.lib()
    xt = torch.empty(B, S, S, 4, device=dev)
    tg = torch.empty(B, S, S, 4, device=dev)
    sa, sb, lw = (bufs[k].to(dev) for k in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "loss_weight"))
    td = t.to(dev)
    imgd, noised = img.to(dev), noise.to(dev)     # keep the device copies alive across the launch
    L.lgm_qsample_target(imgd.data_ptr(), noised.data_ptr(), td.data_ptr(), sa.data_ptr(), sb.data_ptr(),
                         1, xt.data_ptr(), tg.data_ptr(), 4, B, C, S * S, 4, ops.stream())
    assert rel(nchw(xt, 3), xt_ref) < 1e-6 and rel(nchw(tg, 3), v_ref) < 1e-6
    assert float(xt[..., 3].abs().max()) == 0.0
    out = torch.randn(B, C, S, S, generator=g, requires_grad=True)
    loss_ref = (F.mse_loss(out, v_ref, reduction="none").reshape(B, -1).mean(1) * bufs["loss_weight"][t]).mean()
    loss_ref.backward()
    outd = nhwc(out.detach(), dev)
    per, loss = torch.empty(B, device=dev), torch.empty(1, device=dev)
    L.lgm_weighted_mse_fwd(outd.data_ptr(), tg.data_ptr(), 4, td.data_ptr(), lw.data_ptr(), B, C, S * S, 4,
                           per.data_ptr(), loss.data_ptr(), ops.stream())
    assert rel(loss, loss_ref.reshape(1)) < 1e-5
    gout = torch.empty(B, S, S, 4, device=dev)
    one = torch.ones(1, device=dev)
    L.lgm_weighted_mse_bwd(outd.data_ptr(), tg.data_ptr(), 4, td.data_ptr(), lw.data_ptr(), one.data_ptr(), B, C,
                           S * S, 4, gout.data_ptr(), ops.stream())
    assert rel(nchw(gout, 3), out.grad) < 1e-5
    # fused Adam against torch.optim.Adam and the oracle restatement, 5 steps, with coupled L2
    n = 1003
    p = torch.randn(n, generator=g)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-3, betas=(0.9, 0.99), weight_decay=1e-2)
    pd = torch.zeros(1004, device=dev)
    pd[:n] = p.to(dev)
    md, vd = torch.zeros_like(pd), torch.zeros_like(pd)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        gd_ = torch.zeros(1004, device=dev)
        gd_[:n] = gr.to(dev)
        ops.adam_step(pd, gd_, md, vd, n, 2e-3, 0.9, 0.99, 1e-8, 1e-2, step)
    assert rel(pd[:n], ref) < 1e-6
    # EMA lerp
    sh, on = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    shd, ond = sh.to(dev), on.to(dev)
    ops.ema_lerp(shd, ond, 0.25)
    assert rel(shd, OO.ema_apply(sh, on, "lerp", 0.25)) < 1e-6


@pytest.mark.parametrize("shape", [(28, 28, 1, 28), (32, 32, 3, 32), (218, 178, 3, 64), (100, 160, 3, 64), (40, 40, 3, 64)])
def test_device_image_transforms_match_reference_stack(dev, shape):
    """ToTensor + Normalize + CenterCropMinXY + antialiased Resize + flip in one kernel vs the
    torch restatement of the reference transform stack (identity, down- and up-scaling cases)."""
    from data.transforms import DeviceTransforms
    from oracle import data as OD
    H, W, C, S = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    imgs = torch.randint(0, 256, (5, H, W, C), dtype=torch.uint8, generator=g)
    flip = torch.tensor([0, 1, 0, 1, 1], dtype=torch.uint8)
    out = DeviceTransforms(S, train=True)(imgs.to(dev), flip.to(dev)).cpu()
    ref = torch.stack([OD.transform(imgs[i], S, bool(flip[i])) for i in range(5)])
    assert out.shape == ref.shape == (5, C, S, S)
    assert float((out - ref).abs().max()) < 2e-5
    ev = DeviceTransforms(S, train=False)(imgs.to(dev)).cpu()
    ref0 = torch.stack([OD.transform(imgs[i], S, False) for i in range(5)])
    assert float((ev - ref0).abs().max()) < 2e-5


def test_attend_module_path(dev):
    """models/modules/attend.py (reference models/modules/attend.py:97-126): stand-alone Attend()(q, k, v) on the HIP
    attention kernel, with and without leading memory rows, against the reference's einsum / softmax formula."""
    from models.modules.attend import Attend
    g = torch.Generator().manual_seed(11)
    b, h, n, d, M = 3, 4, 16, 32, 4
    q = torch.randn(b, h, n, d, generator=g)
    k = torch.randn(b, h, n + M, d, generator=g)
    v = torch.randn(b, h, n + M, d, generator=g)
    k[:, :, :M] = k[:1, :, :M]
    v[:, :, :M] = v[:1, :, :M]

    def ref(q, k, v):
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
        return torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    att = Attend()
    with torch.no_grad():
        out = att(q.to(dev), k.to(dev), v.to(dev))
    assert rel(out, ref(q, k, v)) < RTOL
    q2, k2, v2 = (t[:, :, :n].clone().requires_grad_(True) for t in (q, k[:, :, M:], v[:, :, M:]))
    w = torch.randn(b, h, n, d, generator=g)
    (ref(q2, k2, v2) * w).sum().backward()
    qd, kd, vd = (t.detach().to(dev).requires_grad_(True) for t in (q2, k2, v2))
    o = att(qd, kd, vd)
    assert rel(o, ref(q2, k2, v2)) < RTOL
    (o * w.to(dev)).sum().backward()
    for a, r in ((qd, q2), (kd, k2), (vd, v2)):
        assert rel(a.grad, r.grad) < RTOL


# (B, Cin, Cout, H, k, stride, pad): the paths behind lgm_conv_xy_post / lgm_conv_yx_post - implicit GEMM with and
# without split-K, strided (phase-decomposed) input gradient, the 1x1 fast paths and the direct 3x3 kernel (no epilogue
# hook there: the entry point finishes with the elementwise launch), ragged channel counts
POST_CASES = [(8, 3, 32, 32, 4, 2, 1), (16, 32, 64, 16, 4, 2, 1), (64, 128, 128, 4, 3, 1, 1), (64, 128, 32, 4, 3, 1, 1),
              (64, 32, 128, 4, 1, 1, 0), (4, 64, 128, 8, 4, 2, 1), (2, 20, 12, 8, 3, 1, 1)]


@pytest.mark.parametrize("case", POST_CASES)
@pytest.mark.parametrize("act", ["relu", "lrelu"])
def test_conv_epilogue_activation_and_mask(dev, case, act):
    """Conv2d -> ReLU / LeakyReLU as ONE launch (reference vqvae.py:36-51,74-85, residual.py:14-20, dcgan.py) and its
    autograd mirror image - the activation's backward as a mask in the epilogue of the input gradient that precedes
    it: y = act(conv(x) + b + res);  gx = (dgrad(gy) + res') * act'(saved output).  Against torch on the CPU."""
    from lgm_hip import ops
    B, ci, co, hw, k, st, pd = case
    gen = torch.Generator().manual_seed(sum(case))
    slope = 0.2 if act == "lrelu" else 0.0
    code = ops.ACT_LRELU if act == "lrelu" else ops.ACT_RELU
    x = torch.randn(B, ci, hw, hw, generator=gen)
    w = torch.randn(co, ci, k, k, generator=gen) / math.sqrt(ci * k * k)
    b = torch.randn(co, generator=gen)
    ho = (hw + 2 * pd - k) // st + 1
    res = torch.randn(B, co, ho, ho, generator=gen)
    f = (lambda t: F.leaky_relu(t, slope)) if act == "lrelu" else F.relu
    y_ref = f(F.conv2d(x, w, b, stride=st, padding=pd) + res)
    xd, wd, bd, rd = nhwc(x, dev), phys_weight(w, dev), vec(b, dev), nhwc(res, dev)
    Cp, Np = xd.shape[-1], rd.shape[-1]
    g = ops.make_geom(B, hw, hw, Cp, Np, k, k, st, pd)
    y = torch.empty(B, ho, ho, Np, device=dev)
    ops.conv_xy(g, xd, wd.data_ptr(), bd.data_ptr(), rd, y, post=ops.make_post(code, slope))
    assert rel(nchw(y, co), y_ref) < 1e-5
    # backward: gradient w.r.t. an input that is itself an activation output `xin` (saved), with a residual-path term
    xin = f(torch.randn(B, ci, hw, hw, generator=gen))
    gy = torch.randn(B, co, ho, ho, generator=gen)
    gres = torch.randn(B, ci, hw, hw, generator=gen)
    dg = torch.nn.grad.conv2d_input(x.shape, w, gy, stride=st, padding=pd)
    gx_ref = (dg + gres) * torch.where(xin > 0, torch.ones_like(xin), torch.full_like(xin, slope))
    gyd, xind = nhwc(gy, dev), nhwc(xin, dev)
    gx = nhwc(gres, dev).contiguous()
    wt = wd.permute(2, 1, 0).contiguous()
    ops.conv_yx(g, gyd, wd.data_ptr(), None, gx, gx, wt.data_ptr(), post=ops.make_post(0, 0.0, xind, slope), post_mask=xind)
    assert rel(nchw(gx, ci), gx_ref) < 1e-5
    assert float(gx[..., ci:].abs().max()) == 0 if Cp > ci else True      # padding lanes stay zero


# (B, Cin, Cout, H, k): 1x1 layers of the UNet at small per-GPU batches (the pairable kernels), one with a residual, a
# large one (other kernels are picked: the call must still equal the two separate calls) and a 3x3 without Winograd
PAIR_CASES = [(16, 512, 384, 4, 1), (16, 128, 512, 4, 1), (16, 1024, 512, 4, 1), (8, 256, 384, 8, 1), (2, 96, 64, 16, 1),
              (128, 64, 384, 32, 1), (4, 128, 32, 4, 3)]


@pytest.mark.parametrize("case", PAIR_CASES)
@pytest.mark.parametrize("variant", ["plain", "accumulate", "deferred"])
def test_generic_backward_pair_equals_separate_calls(dev, case, variant):
    """lgm_conv_bwd_pair (weight / bias gradient + input gradient of a layer, ONE launch when the dispatchers pick the two
    kernels that can share a grid) against lgm_conv_wgrad + lgm_conv_yx: torch.equal - the same kernel bodies run either
    way - and against torch on the CPU."""
    import ctypes
    from lgm_hip import ops
    B, ci, co, hw, k = case
    L = ops.lib()
    gen = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(B, ci, hw, hw, generator=gen)
    gy = torch.randn(B, co, hw, hw, generator=gen)
    w = torch.randn(co, ci, k, k, generator=gen) / math.sqrt(ci * k * k)
    xd, gyd, wd = nhwc(x, dev), nhwc(gy, dev), phys_weight(w, dev)
    wt = wd.permute(2, 1, 0).contiguous()
    g = ops.make_geom(B, hw, hw, xd.shape[-1], gyd.shape[-1], k, k, 1, k // 2)
    base = torch.randn(B, hw, hw, xd.shape[-1], generator=gen).to(dev)
    # ---- separate calls
    gw_ref, gb_ref = torch.zeros_like(wd), torch.zeros(gyd.shape[-1], device=dev)
    ops.conv_wgrad(g, gyd, xd, gw_ref.data_ptr(), 0.0, gb_ref.data_ptr())
    gx_ref = base.clone()
    ops.conv_yx(g, gyd, wd.data_ptr(), None, gx_ref if variant == "accumulate" else None, gx_ref, wt.data_ptr())
    # ---- one call
    gw, gb = torch.zeros_like(wd), torch.zeros(gyd.shape[-1], device=dev)
    gx = base.clone()
    defer = [] if variant == "deferred" else None
    ops.conv_bwd_generic(g, gyd, xd, wd.data_ptr(), wt.data_ptr(), gw.data_ptr(), 0.0, gb.data_ptr(), defer,
                         gx if variant == "accumulate" else None, gx)
    kern = L._dll.lgm_last_kernel().decode()
    if defer:
        ops.wgrad_reduce_batch(defer, dev)
    assert torch.equal(gx, gx_ref), kern
    assert torch.equal(gw, gw_ref) and torch.equal(gb, gb_ref), kern
    if k == 1 and B * hw * hw <= 1024 and ci % 32 == 0 and co % 32 == 0:
        assert kern == "gemm_bwd_pair_kernel", kern          # the small 1x1 layers do share a launch
    dg = torch.nn.grad.conv2d_input(x.shape, w, gy, padding=k // 2)
    ref = dg + (nchw(base, ci) if variant == "accumulate" else 0)
    assert rel(nchw(gx, ci), ref) < 1e-5
    gw_oihw = gw[:co, :, :ci].reshape(co, k, k, ci).permute(0, 3, 1, 2)          # physical [N][T][C] -> OIHW
    assert rel(gw_oihw, torch.nn.grad.conv2d_weight(x, w.shape, gy, padding=k // 2)) < 1e-5


@pytest.mark.parametrize("B,layers", [(5, 2), (256, 2), (2, 4)])
def test_vqvae_residual_stack_forward_in_one_launch(dev, B, layers):
    """lgm_resstack_fwd against the reference arithmetic (residual.py:5-43): per layer y = relu(conv3x3(cur)),
    cur' = relu(conv1x1(y) + cur), with cur = relu(x) on entry."""
    from lgm_hip import ops
    C, R, H, W = 128, 32, 4, 4
    g = torch.Generator().manual_seed(B + layers)
    x = torch.relu(torch.randn(B, C, H, W, generator=g))
    w3 = [torch.randn(R, C, 3, 3, generator=g) / math.sqrt(9 * C) for _ in range(layers)]
    w1 = [torch.randn(C, R, 1, 1, generator=g) / math.sqrt(R) for _ in range(layers)]
    cur = x.double()
    ys, zs = [], []
    for a, b in zip(w3, w1):
        y = torch.relu(F.conv2d(cur, a.double(), padding=1))
        cur = torch.relu(F.conv2d(y, b.double()) + cur)
        ys.append(y)
        zs.append(cur)
    xd = nhwc(x, dev, extra=4)
    w3d = [a.permute(0, 2, 3, 1).reshape(R, 9, C).contiguous().to(dev) for a in w3]       # [Np][taps][Cp]
    w1d = [b.reshape(C, R).contiguous().to(dev) for b in w1]
    r = ops.resstack_fwd(xd, [t.data_ptr() for t in w3d], [t.data_ptr() for t in w1d], R)
    assert r is not None
    for l in range(layers):
        assert rel(nchw(r[0][l]), ys[l]) < 2e-5, l
        assert rel(nchw(r[1][l]), zs[l]) < 2e-5, l


@pytest.mark.parametrize("dim,B", [(16, 2), (32, 5), (64, 128), (64, 16), (64, 1)])
def test_time_embedding_in_one_launch(dev, dim, B, parity):
    """lgm_time_mlp_fwd / _bwd (SinusoidalPosEmb -> Linear -> GELU -> Linear -> SiLU, reference ddpm.py:119-132, 328-333,
    181-183; opt-in, LGM_TIME_MLP=1: measured slower than the six launches it replaces) against the same chain in torch
    float64 autograd; a row's results are bit-identical whatever batch it is in."""
    from lgm_hip import ops
    td = 4 * dim
    g = torch.Generator().manual_seed(dim + B)
    t = torch.randint(0, 1000, (B,), generator=g)
    w1, b1 = torch.randn(td, dim, generator=g) * dim ** -0.5, torch.randn(td, generator=g) * 0.1
    w2, b2 = torch.randn(td, td, generator=g) * td ** -0.5, torch.randn(td, generator=g) * 0.1
    gst = torch.randn(B, td, generator=g)
    assert ops.lib().lgm_time_mlp_supported(dim, td) == 1
    d = lambda x: x.to(dev).contiguous()  # noqa: E731
    W1, B1, W2, B2, T, GST = d(w1), d(b1), d(w2), d(b2), d(t), d(gst)
    outs = [torch.empty(B, n, device=dev) for n in (dim, td, td, td, td)]
    ops.time_mlp_fwd(T, dim, 10000.0, W1.data_ptr(), B1.data_ptr(), W2.data_ptr(), B2.data_ptr(), td, *outs)
    pe, a1, h, temb, st = outs
    # float64 reference with the reference's own frequency table (fp32 exp on the host, as ops.posemb_freqs)
    fr = ops.posemb_freqs(dim, 10000.0, "cpu").double()
    arg = (t.float()[:, None] * fr.float()[None, :]).double()          # the product is formed in fp32 (ddpm.py:130)
    pe_r = torch.cat([arg.sin(), arg.cos()], 1)
    W1r, W2r = w1.double().requires_grad_(True), w2.double().requires_grad_(True)
    B1r, B2r = b1.double().requires_grad_(True), b2.double().requires_grad_(True)
    a1_r = pe_r @ W1r.T + B1r
    h_r = F.gelu(a1_r)
    temb_r = h_r @ W2r.T + B2r
    st_r = F.silu(temb_r)
    st_r.backward(gst.double())
    for name, got, want in (("pe", pe, pe_r), ("a1", a1, a1_r), ("h", h, h_r), ("temb", temb, temb_r), ("st", st, st_r)):
        parity(f"time embedding {name} (dim {dim}, B {B})", rel(got, want.detach()), 1e-5)
    gtemb, ga1 = torch.empty(B, td, device=dev), torch.empty(B, td, device=dev)
    for beta in (0.0, 1.0):
        gw1, gb1 = torch.full((td, dim), 0.5, device=dev), torch.full((td,), 0.5, device=dev)
        gw2, gb2 = torch.full((td, td), 0.5, device=dev), torch.full((td,), 0.5, device=dev)
        ops.time_mlp_bwd(GST, pe, a1, h, temb, W2.data_ptr(), dim, td, gtemb, ga1, gw1.data_ptr(), gb1.data_ptr(),
                         gw2.data_ptr(), gb2.data_ptr(), beta)
        for name, got, want in (("gw1", gw1, W1r.grad), ("gb1", gb1, B1r.grad), ("gw2", gw2, W2r.grad), ("gb2", gb2, B2r.grad)):
            parity(f"time embedding {name} (beta {beta})", rel(got.double().cpu() - 0.5 * beta, want), 2e-5)
    if B >= 5:      # rows 1..3 alone == the same rows inside the batch, bit for bit (2 ranks x B/2 == 1 rank x B)
        sub = [torch.empty(3, n, device=dev) for n in (dim, td, td, td, td)]
        ops.time_mlp_fwd(T[1:4].contiguous(), dim, 10000.0, W1.data_ptr(), B1.data_ptr(), W2.data_ptr(), B2.data_ptr(), td, *sub)
        for a, b in zip(sub, outs):
            assert torch.equal(a, b[1:4])
