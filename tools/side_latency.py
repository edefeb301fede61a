#!/usr/bin/env python3
"""Round-trip latency of a small kernel + event wait on a side stream while the main stream is saturated with GEMMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)


def make_graph(M, N, K, reps):
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ops.linear(A, W, out, None, 0)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                ops.linear(A, W, out, None, 0)
    return gr, (A, W, out)


def probe(side, work, n=40):
    ev = torch.cuda.Event()
    lat = []
    for _ in range(n):
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            work()
            ev.record()
        ev.synchronize()
        lat.append(1e6 * (time.perf_counter() - t0))
        time.sleep(0.0005)
    lat.sort()
    return lat[len(lat) // 2], lat[-1]


def main():
    x = torch.zeros(1, device=DEV)
    big = torch.zeros(6 * 196608, 3, device=DEV)
    small = lambda: x.add_(1)
    large = lambda: big.add_(1.0)
    for name, (M, N, K, reps) in {"enc fc1 15360x4096x1024": (15360, 4096, 1024, 60), "dec 3076x768x768": (3076, 768, 768, 600)}.items():
        gr, keep = make_graph(M, N, K, reps)
        for prio in (0, -1):
            side = torch.cuda.Stream(priority=prio)
            torch.cuda.synchronize()
            idle_s, idle_l = probe(side, small)[0], probe(side, large)[0]
            res = []
            for work in (small, large):
                gr.replay(); gr.replay(); gr.replay()
                res.append(probe(side, work, n=30))
                torch.cuda.synchronize()
            print(f"{name:26s} prio {prio:2d}: idle small {idle_s:6.0f} us, large {idle_l:6.0f} us | busy small med {res[0][0]:7.0f} max {res[0][1]:7.0f} us, "
                  f"large med {res[1][0]:7.0f} max {res[1][1]:7.0f} us")


if __name__ == "__main__":
    main()
