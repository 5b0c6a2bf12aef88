"""Per-kernel view of tools/cu_hog_step.py: N replays of the captured DDPM step with (argv[2] = k > 0) or without a resident
side-stream kernel; run under `rocprofv3 --kernel-trace --stats` and compare the two summaries.
usage: python tools/cu_hog_profile.py <batch> <k>"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402

B, k = int(sys.argv[1]), int(sys.argv[2])
hog = ctypes.CDLL(os.path.join(ROOT, "tools", "libcu_hog.so"))
hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
sink = torch.zeros(4, device=dev)
side = torch.cuda.Stream()
torch.manual_seed(10)
args = type("A", (), {"no_graph": False})()
step, _, info = bench.setup_ddpm(args, dev, 1, 0, 32, B)
for i in range(8):
    step(i)
torch.cuda.synchronize()
if k:
    hog.cu_hog_quiet.argtypes = hog.cu_hog.argtypes
    regs = int(os.environ.get("HOG_REGS", "96"))           # 96 / 240; HOG_QUIET=1: the guest mostly waits
    (hog.cu_hog_quiet if os.environ.get("HOG_QUIET") else hog.cu_hog)(k, 256, regs, ctypes.c_longlong(int(0.5 * 2.1e9)),
                                                                       sink.data_ptr(), side.cuda_stream)
for i in range(20):
    step(100 + i)
torch.cuda.synchronize()
