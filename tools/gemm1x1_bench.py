"""The UNet's 1x1 convolutions / linears at B = 128 (and the per-rank batches): the product's dispatcher (ops.conv_xy: igemm / gemm_stream /
gemm_rows + split-K reducer) against the engine's batched GEMM used as a plain NT GEMM (lgm_weng_gemm, batch 1), warm and cold.
usage (GPU box): python tools/gemm1x1_bench.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402
from cold_launch import cold  # noqa: E402
from wino_bench import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
big = torch.empty(128 << 20, device=dev)
flush = lambda: big.fill_(1.0)  # noqa: E731
# (name, H, Cin, Cout)
LAYERS = [("down 256->64 @16", 16, 256, 64), ("to_qkv 64->384 @16", 16, 64, 384), ("down 256->128 @8", 8, 256, 128),
          ("to_qkv 128->384 @8", 8, 128, 384), ("down 512->256 @4", 4, 512, 256), ("to_qkv 256->384 @4", 4, 256, 384),
          ("to_out 128->256 @4", 4, 128, 256), ("to_qkv 512->384 @4", 4, 512, 384), ("to_out 128->512 @4", 4, 128, 512),
          ("res 768->512 @4", 4, 768, 512), ("res 384->256 @8", 8, 384, 256), ("to_qkv 256->384 @8", 8, 256, 384),
          ("res 192->128 @16", 16, 192, 128), ("to_qkv 128->384 @16", 16, 128, 384), ("res 128->64 @32", 32, 128, 64),
          ("to_qkv 64->384 @32", 32, 64, 384), ("to_out 128->64 @32", 32, 128, 64)]
print(f"B = {B}; us warm / cold;  event overhead (cold) {cold(lambda: None, flush):.1f}")
tot = [0.0, 0.0]
for name, h, ci, co in LAYERS:
    g = ops.make_geom(B, h, h, ci, co, 1, 1, 1, 0)
    x = torch.randn(B, h, h, ci, device=dev)
    w = torch.randn(co, 1, ci, device=dev) * ci ** -0.5
    bias = torch.randn(co, device=dev)
    y = torch.empty(B, h, h, co, device=dev)
    y2 = torch.empty_like(y)
    M = B * h * h
    a = (timeit(lambda: ops.conv_xy(g, x, w.data_ptr(), bias.data_ptr(), None, y)) * 1e3,
         cold(lambda: ops.conv_xy(g, x, w.data_ptr(), bias.data_ptr(), None, y), flush))
    k = ops.lib()._dll.lgm_last_kernel().decode()
    fn = lambda: ops.lib().lgm_weng_gemm(x.data_ptr(), w.data_ptr(), y2.data_ptr(), M, co, ci, ci, ci, co, 1, 0, 0, 0, ops.stream())  # noqa: E731
    b = (timeit(fn) * 1e3, cold(fn, flush))
    fl = 2.0 * M * ci * co
    by = 4.0 * (M * ci + M * co + ci * co)
    ideal = max(fl / 157.3e12, by / 5e12) * 1e6
    tot[0] += a[1]
    tot[1] += b[1]
    print(f"  {name:22s} M={M:6d}: product {a[0]:6.1f} / {a[1]:6.1f} ({k[:34]:34s})  weng_gemm {b[0]:6.1f} / {b[1]:6.1f}   ideal {ideal:5.1f}"
          f"   cold ratio {a[1] / b[1]:.2f}", flush=True)
print(f"  sum cold: product {tot[0]:.0f} us, weng_gemm {tot[1]:.0f} us (no bias / residual epilogue in weng_gemm yet)")
