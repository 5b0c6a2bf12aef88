"""Latency of the GroupNorm kernels at small batches (GPU box): python tools/gn_bench.py [B]
Times back-to-back launches with HIP events; variants: plain / planes (2, 4) / residual / FiLM / activation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lightning-generative-models_amd"))
import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ONLY = sys.argv[2] if len(sys.argv) > 2 else None     # one variant only (under rocprofv3: kernel durations per variant)
dev = torch.device("cuda", 0)


def timeit(fn, n=200):
    if ONLY is not None and ONLY != timeit.current:
        return float("nan")
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for (hw, C) in ((32, 64), (16, 128), (8, 256), (4, 512)):
    x = torch.randn(B, hw, hw, C, device=dev)
    y = torch.empty_like(x)
    res = torch.randn_like(x)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    ss = torch.randn(B, 2 * C, device=dev)
    planes = torch.randn(4, B * hw * hw * C, device=dev)
    bias = torch.randn(C, device=dev)
    row = {}

    def t(name, fn):
        timeit.current = name
        row[name] = timeit(fn)
    t("plain", lambda: ops.gn_fwd(x, 8, 1e-5, gamma.data_ptr(), beta.data_ptr(), None, False, None, y))
    t("act", lambda: ops.gn_fwd(x, 8, 1e-5, gamma.data_ptr(), beta.data_ptr(), None, True, None, y))
    t("film+act", lambda: ops.gn_fwd(x, 8, 1e-5, gamma.data_ptr(), beta.data_ptr(), ss, True, None, y))
    t("act+res", lambda: ops.gn_fwd(x, 8, 1e-5, gamma.data_ptr(), beta.data_ptr(), None, True, res, y))
    for sp in (2, 4):
        t(f"planes{sp}", lambda: ops.gn_fwd(x, 8, 1e-5, gamma.data_ptr(), beta.data_ptr(), ss, True, None, y,
                                                      planes=(planes.data_ptr(), planes.shape[1], sp, bias.data_ptr())))
    sv = ops.gn_fwd(x, 8, 1e-5, gamma.data_ptr(), beta.data_ptr(), ss, True, None, y)
    gx = torch.empty_like(x)
    gg, gb_, gss = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(B, 2 * C, device=dev)
    t("bwd", lambda: ops.gn_bwd(x, res, 8, gamma.data_ptr(), beta.data_ptr(), ss, True, sv, gx, False,
                                           gg.data_ptr(), gb_.data_ptr(), 0.0, gss, 0.0))
    t("empty", lambda: ops.fill(gg, 0.0))
    t("rms", lambda: ops.rmsnorm_fwd(x, gamma.data_ptr(), None, y))
    print(f"B={B} {hw}x{hw} C={C}: " + "  ".join(f"{k} {v:.1f}us" for k, v in row.items()), flush=True)

