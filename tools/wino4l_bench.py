"""Light (16-tile, 256-thread) vs whole-CU (32-tile, 512-thread) F(4x4,3x3) workgroups and the F(2x2) kernel on the UNet's
large-map layers, per-rank batches, alone and beside resident foreign workgroups (tools/cu_hog.hip: what a collective's
kernel does to a chip-filling launch).
usage (GPU box): hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libcu_hog.so tools/cu_hog.hip && \
                 python tools/wino4l_bench.py [B ...]           (HOG=k: k resident workgroups, default 0 and 16)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402
from wino_bench import timeit, wino, wino_weights  # noqa: E402
from wino4_bench import wino4, wino4_weights  # noqa: E402

SHAPES = [("64->64 @32", 64, 64, 32, 8), ("128->64 @32", 128, 64, 32, 4), ("64->64 @16", 64, 64, 16, 4),
          ("192->128 @16", 192, 128, 16, 2), ("128->128 @16", 128, 128, 16, 2), ("256->128 @16", 256, 128, 16, 1),
          ("128->128 @8", 128, 128, 8, 4), ("384->256 @8", 384, 256, 8, 2), ("256->256 @8", 256, 256, 8, 2),
          ("512->256 @8", 512, 256, 8, 1)]


def wino4l(yx, g, a, u, bias, res, out):
    L = ops.lib()
    n = L.lgm_conv3x3_wino4l_workspace(ctypes.byref(g), yx)
    ws = ops.workspace(n, a.device) if n > 0 else None
    L.lgm_conv3x3_wino4l(yx, ctypes.byref(g), a.data_ptr(), ops.pitch(a), u.data_ptr(), None if bias is None else bias.data_ptr(),
                         None if res is None else res.data_ptr(), 0 if res is None else ops.pitch(res), out.data_ptr(),
                         ops.pitch(out), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, ops.stream())


def main():
    batches = [int(b) for b in sys.argv[1:]] or [128, 64, 32, 16]
    hogs = [int(h) for h in os.environ.get("HOG", "0,16").split(",")]
    dev = torch.device("cuda", 0)
    hog = None
    if any(hogs):
        hog = ctypes.CDLL(os.path.join(ROOT, "tools", "libcu_hog.so"))
        hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(4, device=dev)
    side = torch.cuda.Stream()
    for B in batches:
        for k in hogs:
            print(f"--- B = {B}, {k} resident foreign workgroups ---  us per launch: xy F(2x2) | F(4x4) 32-tile | F(4x4) light || yx ...")
            tot = [0.0] * 6
            for name, ci, co, hw, cnt in SHAPES:
                g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
                x = torch.randn(B, hw, hw, ci, device=dev)
                y = torch.randn(B, hw, hw, co, device=dev)
                gx = torch.empty_like(x)
                w = torch.randn(co, 9, ci, device=dev) * (1.0 / (3 * ci ** 0.5))
                bd = torch.randn(co, device=dev)
                uf, ub = wino_weights(w)
                uf4, ub4 = wino4_weights(w)
                big_ok = ops.lib().lgm_conv3x3_wino4_supported(ctypes.byref(g), 0) == 1
                light_ok = ops.lib().lgm_conv3x3_wino4l_supported(ctypes.byref(g), 0) == 1
                fns = [lambda: wino(0, g, x, uf, bd, None, y), (lambda: wino4(0, g, x, uf4, bd, None, y)) if big_ok else None,
                       (lambda: wino4l(0, g, x, uf4, bd, None, y)) if light_ok else None,
                       lambda: wino(1, g, y, ub, None, None, gx), (lambda: wino4(1, g, y, ub4, None, None, gx)) if big_ok else None,
                       (lambda: wino4l(1, g, y, ub4, None, None, gx)) if light_ok else None]
                ts = []
                for fn in fns:
                    if fn is None:
                        ts.append(float("nan"))
                        continue
                    torch.cuda.synchronize()
                    if k:      # ~25 ms of spinning covers the 130 launches of one timing
                        hog.cu_hog(k, 256, int(os.environ.get("HOG_REGS", "96")), ctypes.c_longlong(int(0.025 * 2.1e9)),
                                   sink.data_ptr(), side.cuda_stream)
                    ts.append(timeit(fn) * 1e3)
                    torch.cuda.synchronize()
                for i, t in enumerate(ts):
                    tot[i] += cnt * t
                print(f"{name:14s} | {ts[0]:7.1f} {ts[1]:7.1f} {ts[2]:7.1f} || {ts[3]:7.1f} {ts[4]:7.1f} {ts[5]:7.1f}", flush=True)
            print("per step (us, weighted by layer count): " + "  ".join(f"{t:8.0f}" for t in tot), flush=True)


if __name__ == "__main__":
    main()
