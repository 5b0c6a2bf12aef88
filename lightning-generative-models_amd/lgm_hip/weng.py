"""Host side of the non-fused Winograd engine (csrc/winograd_eng.hip): weight transforms U = G g G^T (float64 on the device,
rounded once) and the three-launch convolutions.  Matrices: Toom-Cook with points {0, 1, -1, 2, -2, inf} (F(4,3)) and
{0, 1, -1, 2, inf} (F(4,2)); tools/weng_matrices.py derives them in exact arithmetic and checks them.

Round-6 status: measured prototypes for VERDICT r5 items 3 and 4 (tools/weng_proto.py, tests/test_hip_weng.py); the product
models do not route through here yet (profiles/r06_weng_proto.txt has the numbers the decision rests on)."""
from __future__ import annotations

import torch

from . import ops

_G43 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                     [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
_G42 = torch.tensor([[1 / 2, 0], [-1 / 2, -1 / 2], [-1 / 6, 1 / 6], [1 / 6, 1 / 3], [0, 1]], dtype=torch.float64)


def f43_weights(w: torch.Tensor) -> torch.Tensor:
    """w [N][C][3][3] (Conv2d weight) -> U [36][N][C] fp32"""
    G = _G43.to(w.device)
    u = torch.einsum("ia,ncab,jb->ijnc", G, w.double(), G)
    return u.reshape(36, w.shape[0], w.shape[1]).float().contiguous()


def f42_weights_xy(w: torch.Tensor) -> torch.Tensor:
    """w [N][C][4][4] (Conv2d, stride 2, pad 1) -> U [25][N][4 C], k = (2 p + q) C + c, taps g_pq[a][b] = w[2a + p][2b + q]"""
    G = _G42.to(w.device)
    N, C = w.shape[:2]
    g = w.double().reshape(N, C, 2, 2, 2, 2)             # [n][c][a][p][b][q]
    u = torch.einsum("ia,ncapbq,jb->ijnpqc", G, g, G)
    return u.reshape(25, N, 4 * C).float().contiguous()


def f42_weights_yx(w: torch.Tensor) -> torch.Tensor:
    """The same layer's Y -> X pass: w [N][C][4][4] -> U [4][25][C][N]; phase (p, q) correlates the padded Y side with the
    flipped taps gf_pq[a'][b'] = w[2 (1 - a') + p][2 (1 - b') + q]"""
    G = _G42.to(w.device)
    N, C = w.shape[:2]
    g = w.double().reshape(N, C, 2, 2, 2, 2).flip(2).flip(4)     # flip a and b
    u = torch.einsum("ia,ncapbq,jb->pqijcn", G, g, G)
    return u.reshape(4, 25, C, N).float().contiguous()


def _scratch(n: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(n, dtype=torch.float32, device=like.device)


def conv3x3_f43(x: torch.Tensor, U: torch.Tensor, bias, out=None, work=None):
    """x [B][H][W][C] NHWC dense -> y [B][H][W][N]: three launches.  ``work`` = (V, M) scratch tensors or None."""
    B, H, W, C = x.shape
    N = U.shape[1]
    T = B * (H // 4) * (W // 4)
    V, M = work if work is not None else (_scratch(36 * T * C, x), _scratch(36 * T * N, x))
    y = out if out is not None else torch.empty(B, H, W, N, device=x.device)
    L, st = ops.lib(), ops.stream()
    L.lgm_weng_f43_in(x.data_ptr(), ops.pitch(x), B, H, W, C, V.data_ptr(), st)
    L.lgm_weng_gemm(V.data_ptr(), U.data_ptr(), M.data_ptr(), T, N, C, C, C, N, 36, T * C, N * C, T * N, st)
    L.lgm_weng_f43_out(M.data_ptr(), B, H, W, N, None if bias is None else bias.data_ptr(), y.data_ptr(), ops.pitch(y), st)
    return y


def conv4x4s2_xy(x: torch.Tensor, U: torch.Tensor, bias, out=None, work=None):
    """x [B][H][W][C] -> y [B][H/2][W/2][N] (Conv2d k4 s2 p1)"""
    B, H, W, C = x.shape
    N = U.shape[1]
    T = B * (H // 8) * (W // 8)
    K = 4 * C
    V, M = work if work is not None else (_scratch(25 * T * K, x), _scratch(25 * T * N, x))
    y = out if out is not None else torch.empty(B, H // 2, W // 2, N, device=x.device)
    L, st = ops.lib(), ops.stream()
    L.lgm_weng_f42_in_xy(x.data_ptr(), ops.pitch(x), B, H, W, C, V.data_ptr(), st)
    L.lgm_weng_gemm(V.data_ptr(), U.data_ptr(), M.data_ptr(), T, N, K, K, K, N, 25, T * K, N * K, T * N, st)
    L.lgm_weng_f42_out_xy(M.data_ptr(), B, H // 2, W // 2, N, None if bias is None else bias.data_ptr(), y.data_ptr(),
                          ops.pitch(y), st)
    return y


def conv4x4s2_yx(dy: torch.Tensor, U: torch.Tensor, bias, out=None, work=None):
    """dy [B][Ho][Wo][Ny] -> dx [B][2 Ho][2 Wo][C] (input gradient of Conv2d k4 s2 p1 = ConvTranspose2d k4 s2 p1 forward)"""
    B, Ho, Wo, Ny = dy.shape
    C = U.shape[2]
    T = B * (Ho // 4) * (Wo // 4)
    V, M = work if work is not None else (_scratch(100 * T * Ny, dy), _scratch(100 * T * C, dy))
    dx = out if out is not None else torch.empty(B, 2 * Ho, 2 * Wo, C, device=dy.device)
    L, st = ops.lib(), ops.stream()
    L.lgm_weng_f42_in_yx(dy.data_ptr(), ops.pitch(dy), B, Ho, Wo, Ny, V.data_ptr(), st)
    L.lgm_weng_gemm(V.data_ptr(), U.data_ptr(), M.data_ptr(), T, C, Ny, Ny, Ny, C, 100, T * Ny, C * Ny, T * C, st)
    L.lgm_weng_f42_out_yx(M.data_ptr(), B, Ho, Wo, C, None if bias is None else bias.data_ptr(), dx.data_ptr(),
                          ops.pitch(dx), st)
    return dx
