"""lgm_bn_stats against float64 on random activations: worst |d mean| * rstd and |d rstd| / rstd.
usage (GPU box): python tools/bn_stats_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402

dev = torch.device("cuda", 0)
L = ops.lib()
for rows, C, shift in ((64, 512, 0.0), (64, 512, 5.0), (4096, 128, 0.0), (131072, 64, 0.0), (131072, 64, 3.0)):
    torch.manual_seed(0)
    a = (torch.randn(rows, C, device=dev) * 1.7 + shift).contiguous()
    mean = torch.empty(C, device=dev)
    rstd = torch.empty(C, device=dev)
    ws = torch.empty(L.lgm_bn_workspace(rows, C) // 4 + 16, device=dev)
    L.lgm_bn_stats(a.data_ptr(), C, rows, C, 1e-5, 0.1, mean.data_ptr(), rstd.data_ptr(), None, None, ws.data_ptr(),
                   ops.stream())
    a64 = a.double()
    m64 = a64.mean(0)
    r64 = (a64.var(0, unbiased=False) + 1e-5).rsqrt()
    m32 = a.mean(0)
    print(f"rows {rows:7d} C {C:4d} shift {shift}: |dmean|*rstd {float(((mean.double() - m64).abs() * r64).max()):.2e} "
          f"(torch fp32 {float(((m32.double() - m64).abs() * r64).max()):.2e})  |drstd|/rstd "
          f"{float(((rstd.double() - r64).abs() / r64).max()):.2e}")
