"""VERDICT r5 items 3 and 4, measured: the non-fused Winograd engine (transform launch -> batched GEMM -> transform launch)
against the kernels the product runs today, warm (back to back) and cold (behind a 512 MB fill, as inside the step).
  item 3: 3x3 layers of the 4 x 4 maps (F(4x4,3x3), one tile per image) vs wino_conv_kernel<2> (fused F(2x2)), B = 128 and 16
  item 4: 4x4 / stride-2 layers of the DCGAN critic / generator (F(4x4,2x2) on pixel phases) vs igemm_kernel, both directions
usage (GPU box): python tools/weng_proto.py > gpurun_out/r06_weng_proto.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from lgm_hip import ops, weng  # noqa: E402
from cold_launch import cold  # noqa: E402
from wino_bench import timeit, wino, wino_weights  # noqa: E402

dev = torch.device("cuda", 0)
big = torch.empty(128 << 20, device=dev)


def flush():
    big.fill_(1.0)


def both(fn):
    return timeit(fn) * 1e3, cold(fn, flush)


def stages(fns):
    """cold time of each launch of a sequence, run alone behind the flush (upper bound per stage)"""
    return [cold(f, flush) for f in fns]


def item3():
    print("== item 3: 3x3 on 4 x 4 maps, forward.  us per layer, warm / cold")
    L = ops.lib()
    for B in (128, 16):
        for ci, co in ((512, 512), (256, 256), (768, 512), (256, 512)):
            g = ops.make_geom(B, 4, 4, ci, co, 3, 3, 1, 1)
            x = torch.randn(B, 4, 4, ci, device=dev)
            y = torch.empty(B, 4, 4, co, device=dev)
            w = torch.randn(co, 9, ci, device=dev) * (1.0 / (3 * ci ** 0.5))
            bd = torch.randn(co, device=dev)
            uf, _ = wino_weights(w)
            w4 = w.reshape(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
            U = weng.f43_weights(w4)
            T = B
            V = torch.empty(36 * T * ci, device=dev)
            M = torch.empty(36 * T * co, device=dev)
            y2 = torch.empty_like(y)
            a = both(lambda: wino(0, g, x, uf, bd, None, y))
            b = both(lambda: weng.conv3x3_f43(x, U, bd, out=y2, work=(V, M)))
            st = ops.stream()
            s = stages([lambda: L.lgm_weng_f43_in(x.data_ptr(), ci, B, 4, 4, ci, V.data_ptr(), st),
                        lambda: L.lgm_weng_gemm(V.data_ptr(), U.data_ptr(), M.data_ptr(), T, co, ci, ci, ci, co, 36, T * ci,
                                                co * ci, T * co, st),
                        lambda: L.lgm_weng_f43_out(M.data_ptr(), B, 4, 4, co, bd.data_ptr(), y2.data_ptr(), co, st)])
            wino(0, g, x, uf, bd, None, y)
            err = float((y2 - y).abs().max() / y.abs().max())
            fl = 2.0 * B * 16 * ci * co * 9
            print(f"  B={B:3d} {ci:4d}->{co:4d}: fused F(2x2) {a[0]:6.1f} / {a[1]:6.1f}   engine F(4x4) {b[0]:6.1f} / {b[1]:6.1f}"
                  f"   (stages cold: in {s[0]:.1f} gemm {s[1]:.1f} out {s[2]:.1f}; gemm executes {fl / 4 / 1e9:.2f} GFLOP)"
                  f"   ratio cold {a[1] / b[1]:.2f}x   max diff vs fused {err:.1e}", flush=True)


def item4():
    print("== item 4: 4x4 / stride-2 layers (DCGAN), B = 128.  us per layer, warm / cold; TFLOP/s algorithmic (cold)")
    L = ops.lib()
    B = 128
    for name, ci, co, h in (("D 64->128 @32->16", 64, 128, 32), ("D 128->256 @16->8", 128, 256, 16),
                            ("D 256->512 @8->4", 256, 512, 8), ("G 1024->512 @4->8 (conv 512->1024 @8)", 512, 1024, 8),
                            ("G 512->256 @8->16 (conv 256->512 @16)", 256, 512, 16),
                            ("G 256->128 @16->32 (conv 128->256 @32)", 128, 256, 32)):
        g = ops.make_geom(B, h, h, ci, co, 4, 4, 2, 1)
        x = torch.randn(B, h, h, ci, device=dev)
        y = torch.randn(B, h // 2, h // 2, co, device=dev)
        w = torch.randn(co, 16, ci, device=dev) * 0.05
        w4 = w.reshape(co, 4, 4, ci).permute(0, 3, 1, 2).contiguous()
        fl = 2.0 * B * (h // 2) ** 2 * ci * co * 16
        Ux, Uy = weng.f42_weights_xy(w4), weng.f42_weights_yx(w4)
        T = B * (h // 8) ** 2
        Vx, Mx = torch.empty(25 * T * 4 * ci, device=dev), torch.empty(25 * T * co, device=dev)
        T2 = B * (h // 8) ** 2
        Vy, My = torch.empty(100 * T2 * co, device=dev), torch.empty(100 * T2 * ci, device=dev)
        yo, xo = torch.empty_like(y), torch.empty_like(x)
        yr, xr = torch.empty_like(y), torch.empty_like(x)
        a = both(lambda: ops.conv_xy(g, x, w.data_ptr(), None, None, yr))
        b = both(lambda: weng.conv4x4s2_xy(x, Ux, None, out=yo, work=(Vx, Mx)))
        c = both(lambda: ops.conv_yx(g, y, w.data_ptr(), None, None, xr))
        d = both(lambda: weng.conv4x4s2_yx(y, Uy, None, out=xo, work=(Vy, My)))
        st = ops.stream()
        K = 4 * ci
        s = stages([lambda: L.lgm_weng_f42_in_xy(x.data_ptr(), ci, B, h, h, ci, Vx.data_ptr(), st),
                    lambda: L.lgm_weng_gemm(Vx.data_ptr(), Ux.data_ptr(), Mx.data_ptr(), T, co, K, K, K, co, 25, T * K, co * K,
                                            T * co, st),
                    lambda: L.lgm_weng_f42_out_xy(Mx.data_ptr(), B, h // 2, h // 2, co, None, yo.data_ptr(), co, st)])
        ex = float((yo - yr).abs().max() / yr.abs().max())
        ey = float((xo - xr).abs().max() / xr.abs().max())
        tf = lambda us: fl / (us * 1e-6) / 1e12  # noqa: E731
        print(f"  {name:40s} {fl / 1e9:6.2f} GFLOP | X->Y igemm {a[0]:6.1f} / {a[1]:6.1f} ({tf(a[1]):5.1f} TF)  engine {b[0]:6.1f} / "
              f"{b[1]:6.1f} ({tf(b[1]):5.1f} TF alg, {tf(b[1]) / 2.56:5.1f} exec)  {a[1] / b[1]:.2f}x"
              f" | Y->X igemm {c[0]:6.1f} / {c[1]:6.1f}  engine {d[0]:6.1f} / {d[1]:6.1f}  {c[1] / d[1]:.2f}x"
              f" | stages X->Y cold: in {s[0]:.1f} gemm {s[1]:.1f} out {s[2]:.1f} | diff {ex:.1e} / {ey:.1e}", flush=True)


if __name__ == "__main__":
    print(f"event overhead alone (cold): {cold(lambda: None, flush):.1f} us")
    item3()
    item4()
