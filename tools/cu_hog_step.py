"""What a resident collective kernel costs the training step (1 GPU emulation of the N > 1 overlap): `k` workgroups of 256
threads spin on a side stream while the captured DDPM step replays on the main stream.
usage (GPU box): hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libcu_hog.so tools/cu_hog.hip && python tools/cu_hog_step.py [batch ...]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    batches = [int(b) for b in sys.argv[1:]] or [128, 64, 16]
    hog = ctypes.CDLL(os.path.join(ROOT, "tools", "libcu_hog.so"))
    hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    hog.cu_hog_quiet.argtypes = hog.cu_hog.argtypes
    dev = torch.device("cuda", 0)
    sink = torch.zeros(4, device=dev)
    side = torch.cuda.Stream()
    for B in batches:
        torch.manual_seed(10)
        args = type("A", (), {"no_graph": False})()
        step, _, info = bench.setup_ddpm(args, dev, 1, 0, 32, B)
        for i in range(8):
            step(i)
        torch.cuda.synchronize()

        def timed(k, regs, n=20):
            torch.cuda.synchronize()
            if k:
                # ~60 ms of spinning at ~2.1 GHz: covers the 20 timed steps at every batch here
                (hog.cu_hog_quiet if os.environ.get("HOG_QUIET") else hog.cu_hog)(k, 256, regs, ctypes.c_longlong(int(0.35 * 2.1e9)), sink.data_ptr(), side.cuda_stream)
            t0 = time.perf_counter()
            for i in range(n):
                step(100 + i)
            torch.cuda.current_stream().synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
            torch.cuda.synchronize()
            return dt
        base = timed(0, 32)
        print(f"B={B}: alone {base:.3f} ms/step", flush=True)
        for k in (1, 16):
            quiet = ", quiet" if os.environ.get("HOG_QUIET") else ""
            for regs in [int(v) for v in os.environ.get("HOG_REGS", "96").split(",")]:
                print(f"   {k:3d} resident workgroups x 256 threads ({regs} VGPR class{quiet}): {timed(k, regs):.3f} ms/step", flush=True)
        del step, info
        bench._release()


if __name__ == "__main__":
    main()
