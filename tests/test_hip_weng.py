"""GPU: the non-fused Winograd engine (csrc/winograd_eng.hip, lgm_hip/weng.py) - F(4x4,3x3) on small maps and F(4x4,2x2) on the
pixel phases of the 4x4 / stride-2 layers, both directions - against float64 convolutions of the same operands (reference
arithmetic: nn.Conv2d ddpm.py:160, dcgan.py:150-158; nn.ConvTranspose2d dcgan.py:79-87)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("M,N,K,batch", [(128, 512, 512, 36), (16, 512, 768, 36), (2048, 128, 256, 25), (100, 36, 20, 3),
                                         (64, 128, 32, 1), (300, 260, 100, 5)])
def test_batched_gemm(dev, M, N, K, batch, parity):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(batch, M, K, generator=g).to(dev)
    Bm = torch.randn(batch, N, K, generator=g).to(dev)
    C = torch.full((batch, M, N), float("nan"), device=dev)
    from lgm_hip import ops
    ops.lib().lgm_weng_gemm(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, K, K, N, batch, M * K, N * K, M * N,
                            ops.stream())
    ref = torch.einsum("bmk,bnk->bmn", A.double(), Bm.double())
    parity(f"batched NT GEMM {batch} x [{M} x {K}] [{N} x {K}]^T", rel(C, ref), 2e-6)


@pytest.mark.parametrize("B,C,N,H", [(128, 512, 512, 4), (16, 256, 512, 4), (3, 64, 128, 8), (2, 32, 32, 16)])
def test_f43_small_maps(dev, B, C, N, H, parity):
    from lgm_hip import weng
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    w = (torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    y = weng.conv3x3_f43(nhwc(x), weng.f43_weights(w), b)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    parity(f"F(4x4,3x3) engine {C}->{N} @{H}, B={B}", rel(y, nhwc(ref)), 2e-5)


@pytest.mark.parametrize("B,C,N,H", [(128, 64, 128, 32), (4, 128, 256, 16), (5, 256, 512, 8), (2, 8, 12, 16)])
def test_f42_stride2_both_directions(dev, B, C, N, H, parity):
    """Conv2d k4 s2 p1 X -> Y and its input gradient (= ConvTranspose2d k4 s2 p1 forward) through the phase decomposition."""
    from lgm_hip import weng
    g = torch.Generator().manual_seed(B + C + N)
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    w = (torch.randn(N, C, 4, 4, generator=g) * (16 * C) ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    y = weng.conv4x4s2_xy(nhwc(x), weng.f42_weights_xy(w), b)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    parity(f"F(4x4,2x2)-phase engine, X->Y {C}->{N} @{H}, B={B}", rel(y, nhwc(ref)), 2e-5)
    dy = torch.randn(B, N, H // 2, H // 2, generator=g).to(dev)
    bc = torch.randn(C, generator=g).to(dev)
    dx = weng.conv4x4s2_yx(nhwc(dy), weng.f42_weights_yx(w), bc)
    refx = F.conv_transpose2d(dy.double(), w.double(), bc.double(), stride=2, padding=1)
    parity(f"F(4x4,2x2)-phase engine, Y->X {N}->{C} @{H // 2}->{H}, B={B}", rel(dx, nhwc(refx)), 2e-5)


@pytest.mark.parametrize("M,K,N,xp,yp,rp", [(2048, 512, 384, 512, 384, 0), (8192, 256, 384, 256, 384, 0),
                                            (2048, 768, 512, 768, 1024, 1024), (32768, 192, 128, 192, 256, 128),
                                            (1000, 132, 130, 136, 132, 140)])
def test_gemm_with_bias_and_residual_epilogue(dev, M, K, N, xp, yp, rp, parity):
    """lgm_weng_gemm_epi = a 1x1 convolution (y = x W^T + b + res) with pitched operands: a channel slice of a concat buffer in,
    a channel slice out, the residual from a third pitch; rp == 0: no residual."""
    from lgm_hip import ops
    g = torch.Generator().manual_seed(M + K)
    xb = torch.randn(M, xp, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    yb = torch.full((M, yp), float("nan"), device=dev)
    rb = torch.randn(M, rp, generator=g).to(dev) if rp else None
    ops.lib().lgm_weng_gemm_epi(xb.data_ptr(), w.data_ptr(), yb.data_ptr(), M, N, K, xp, K, yp, b.data_ptr(),
                                None if rb is None else rb.data_ptr(), rp, ops.stream())
    ref = xb[:, :K].double() @ w.double().T + b.double()
    if rb is not None:
        ref = ref + rb[:, :N].double()
    parity(f"1x1 GEMM with epilogue M={M} K={K} N={N}", rel(yb[:, :N], ref), 2e-6)
    assert torch.isnan(yb[:, N:]).all()                        # nothing written past the N columns of a wider row
