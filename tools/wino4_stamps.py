"""Cycle stamps of the F(4x4,3x3) convolution kernel (diagnostic build): where a workgroup's time goes.
usage (GPU box): python tools/wino4_stamps.py [Cin Cout HW B]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402
from wino4_bench import wino4, wino4_weights  # noqa: E402


def main():
    ci, co, hw, B = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 64, 32, 128)
    exps = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) >= 6 else [0]
    dev = torch.device("cuda", 0)
    w = torch.randn(co, 9, ci, device=dev) * 0.05
    uf, ub = wino4_weights(w)
    x = torch.randn(B, hw, hw, ci, device=dev)
    y = torch.empty(B, hw, hw, co, device=dev)
    bias = torch.randn(co, device=dev)
    g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    for _ in range(200):                      # clocks up
        wino4(0, g, x, uf, bias, None, y)
    torch.cuda.synchronize()
    cold = len(sys.argv) >= 7 and sys.argv[6] == "cold"      # every stamped launch behind a 512 MB fill, as inside the step
    big = torch.empty(128 << 20, device=dev) if cold else None
    for exp in exps:
        one(exp, g, x, uf, bias, y, dev, big)


def one(exp, g, x, uf, bias, y, dev, big=None):
    nwg = 4096
    dbg = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
    ops.lib().lgm_wino4_set_debug_buffer(dbg.data_ptr(), exp)
    for _ in range(3):
        if big is not None:
            big.fill_(1.0)
        wino4(0, g, x, uf, bias, None, y)
    torch.cuda.synchronize()
    ops.lib().lgm_wino4_set_debug_buffer(None, 0)
    print(f"---- EXP = {exp} (bit 0: no transform, 1: no fetch / commit, 2: no U loads, 3: no V reads)")
    d = dbg.view(nwg, 32).cpu()
    used = d[:, 0] > 0
    d = d[used]
    n = int(d[0, 0])
    st = d[:, 1:1 + n].double()
    dt = (st[:, 1:] - st[:, :-1])
    names = ["setup", "zero+addr", "land+commit", "barrier", "transform", "barrier"] + [f"phase{i}" for i in range(n - 9)] + ["epi0", "epi1"]
    print(f"{d.shape[0]} workgroups, {n} stamps; cycles (median over workgroups):")
    for i, nm in enumerate(names):
        print(f"  {nm:10s} {dt[:, i].median():9.0f}   (min {dt[:, i].min():.0f} max {dt[:, i].max():.0f})")
    tot = st[:, -1] - st[:, 0]
    print(f"  total      {tot.median():9.0f}")
    # launch ramp and tail: entry stamps (s_memtime: shader clock, compared across workgroups - meaningful if the XCDs'
    # counters run together) and exit stamps (s_memrealtime, 100 MHz, one clock for the chip)
    ent = st[:, 0] - st[:, 0].min()
    ext = (d[:, 31].double() - d[:, 31].double().min()) * 10.0          # ns
    q = torch.tensor([0.1, 0.5, 0.9, 1.0], dtype=torch.float64)
    print("  entry stamp - earliest entry (cycles), 10 / 50 / 90 / 100 %:", [int(v) for v in torch.quantile(ent, q)])
    print("  exit time - earliest exit (ns), 10 / 50 / 90 / 100 %:", [int(v) for v in torch.quantile(ext, q)])


if __name__ == "__main__":
    main()
