"""lgm_weng_gemm alone: TFLOP/s (fp32 MFMA peak 157.3) at asymptotic and at the engine's shapes; LGM_WENG_TILE=<bm>x<bn> pins the tile.
usage (GPU box): python tools/weng_gemm_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402
from wino_bench import timeit  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [("square 4096^3", 4096, 4096, 4096, 1), ("square 2048^3 x 4", 2048, 2048, 2048, 4),
          ("unet 512->512 @4 B128", 128, 512, 512, 36), ("unet 768->512 @4 B128", 128, 512, 768, 36),
          ("D 64->128 xy", 2048, 128, 256, 25), ("D 128->256 xy", 512, 256, 512, 25), ("D 256->512 xy", 128, 512, 1024, 25),
          ("G 512->1024 xy", 128, 1024, 2048, 25), ("G 256->512 xy", 512, 512, 1024, 25), ("G 128->256 xy", 2048, 256, 512, 25),
          ("G 1024->512 yx", 128, 512, 1024, 100), ("G 512->256 yx", 512, 256, 512, 100), ("G 256->128 yx", 2048, 128, 256, 100)]
print(f"tile pin: {os.environ.get('LGM_WENG_TILE', 'auto')}")
for name, M, N, K, b in SHAPES:
    A = torch.randn(b, M, K, device=dev)
    Bm = torch.randn(b, N, K, device=dev)
    C = torch.empty(b, M, N, device=dev)
    fn = lambda: ops.lib().lgm_weng_gemm(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, K, K, N, b, M * K, N * K, M * N,  # noqa: E731
                                         ops.stream())
    ms = timeit(fn, iters=30)
    fl = 2.0 * M * N * K * b
    print(f"  {name:26s} M={M:5d} N={N:5d} K={K:5d} b={b:3d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s  ({fl / ms / 1e9 / 157.3:.2f})", flush=True)
