"""Cycle stamps of the F(4x4,3x3) convolution kernel's launches INSIDE a training step (eager issue, diagnostic build of
every wino4_conv_kernel launch): where a workgroup's time goes when the launch finds the chip as the step leaves it.
usage (GPU box): python tools/wino4_step_stamps.py [B]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from lgm_hip import ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(no_graph=True)
    step, eager_step, info = bench.setup_ddpm(args, dev, 1, 0, 32, B)
    for i in range(6):
        eager_step(i)
    torch.cuda.synchronize()
    slots = 64
    dbg = torch.zeros(slots * 4096 * 32, dtype=torch.int64, device=dev)
    ops.lib().lgm_wino4_set_debug_buffer(dbg.data_ptr(), slots << 16)
    eager_step(6)
    torch.cuda.synchronize()
    ops.lib().lgm_wino4_set_debug_buffer(None, 0)
    d = dbg.view(slots, 4096, 32).cpu()
    print("launch  cls C->N @H splits grid res | cycles (median over workgroups): setup zero+addr land+commit barrier transform "
          "barrier | phases (first four, median of the rest) | epi0 epi1 | total | entry spread us (50/100 %) exit spread us (50/100 %) "
          "| first entry -> last exit us")
    for s in range(slots):
        meta = d[s, 4095]
        if meta[7] == 0:
            continue
        C, N, H, W, Bq, splits, cls, grid, res, tns = (int(v) for v in meta[:10])
        rows = d[s, :grid]
        n = int(rows[0, 0])
        m = min(n, 30)
        st = rows[:, 1:1 + m].double()
        dt = st[:, 1:] - st[:, :-1]
        med = dt.median(dim=0).values
        nph = n - 9
        ph = med[6:6 + min(nph, m - 1 - 6)]
        pro = [int(v) for v in med[:6]]
        epi = [int(v) for v in med[6 + nph:6 + nph + 2]] if n <= 30 else []
        tot = int((st[:, -1] - st[:, 0]).median()) if n <= 30 else -1
        line = f"{s:3d}  <{cls}> {C:3d}->{N:3d} @{H:2d} s{splits} g{grid:4d} r{res} | " + " ".join(f"{v:5d}" for v in pro) + " | "
        line += " ".join(f"{int(v):5d}" for v in ph[:4]) + (f" ~{int(ph[4:].median()):5d}" if len(ph) > 4 else "") + " | "
        line += " ".join(f"{v:5d}" for v in epi) + f" | {tot:6d}"
        if n <= 29:
            ent = rows[:, 30].double()
            ext = rows[:, 31].double()
            e0 = ent.min()
            q = torch.tensor([0.5, 1.0], dtype=torch.float64)
            a = torch.quantile(ent - e0, q) * 0.01
            b = torch.quantile(ext - ext.min(), q) * 0.01
            line += f" | {a[0]:.2f} {a[1]:.2f}  {b[0]:.2f} {b[1]:.2f} | {(ext.max() - e0) * 0.01:.2f}"
        print(line)


if __name__ == "__main__":
    main()
