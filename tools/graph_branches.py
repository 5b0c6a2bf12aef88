"""Does a captured HIP graph run independent branches concurrently on this runtime?  Two chains of small kernels
(each kernel fills a fraction of the chip), captured on two streams with an event fork / join, against the same work on
one stream."""
import time
import torch

dev = torch.device("cuda", 0)
N = 40


def chain(x, w, n):
    for _ in range(n):
        x = torch.mm(x, w)          # 512 x 512 x 512: a few workgroups, latency-bound
    return x


def measure(two_streams, size=512):
    a = torch.randn(size, size, device=dev) * 0.01
    b = torch.randn(size, size, device=dev) * 0.01
    w = torch.eye(size, device=dev)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(a, w, 3); chain(b, w, 3)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            if two_streams:
                side.wait_stream(s)
                with torch.cuda.stream(side):
                    rb = chain(b, w, N)
                ra = chain(a, w, N)
                s.wait_stream(side)
            else:
                ra = chain(a, w, N)
                rb = chain(b, w, N)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


for size in (256, 512, 1024):
    print(f"size {size}: one stream {measure(False, size):.3f} ms, two streams {measure(True, size):.3f} ms")
