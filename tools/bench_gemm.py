#!/usr/bin/env python3
"""Micro-benchmark of the GEMM family on the shapes of the CUT3R stack (run on the GPU box)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops

DEV = "cuda:0"
SHAPES = [  # (M, N, K, label)
    (15360, 3072, 1024, "enc qkv B20"), (15360, 1024, 1024, "enc proj B20"), (15360, 4096, 1024, "enc fc1 B20"), (15360, 1024, 4096, "enc fc2 B20"),
    (3076, 2304, 768, "dec qkv W4"), (3076, 768, 768, "dec proj W4"), (3076, 1536, 768, "dec kv W4"), (3076, 3072, 768, "dec fc1 W4"),
    (3076, 768, 3072, "dec fc2 W4"), (1024, 4608, 1536, "mem qkv W4"), (1024, 1536, 1536, "mem proj W4"), (1024, 1536, 6144, "mem fc2 W4"),
    (6152, 768, 768, "dec proj W8"), (6152, 768, 3072, "dec fc2 W8"), (6152, 2304, 768, "dec qkv W8"), (6152, 3072, 768, "dec fc1 W8"),
]


def timeit(fn, reps=50, graph=True):
    """us per call; graph=True replays `reps` captured launches (GPU-side time, no host launch cost)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    if graph:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(reps):
                    fn()
        gr.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            gr.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / (3 * reps) * 1e3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us


def main():
    g = torch.Generator().manual_seed(0)
    CFG = [(128, 9), (128, 13), (256, 0), (128, 9), (128, 13), (256, 0)]
    print(f"{'shape':34s} " + " ".join(f"t{t}s{s:>1d}".rjust(12) for t, s in CFG))
    for M, N, K, label in SHAPES:
        A = torch.randn(M, K, generator=g).half().to(DEV)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        row = []
        for tile, st in CFG:
            ops.GEMM_STAGES = st
            us = timeit(lambda: ops.linear(A, W, out, b, 0, tile=tile))
            row.append(f"{us:6.1f}us/{2.0*M*N*K/us/1e6:5.0f}T")
        ops.GEMM_STAGES = 0
        print(f"{label:12s} {M:5d}x{N:5d}x{K:5d}  " + " ".join(r.rjust(12) for r in row))
    # convs of the DPT head (B=6)
    for (B, H, W_, Cin, Cout, label) in [(8, 192, 256, 256, 256, "rcu @192x256"), (8, 384, 512, 128, 128, "head.2 @384x512"),
                                         (6, 96, 128, 256, 256, "rcu @96x128"), (6, 192, 256, 256, 128, "head.0")]:
        x = torch.randn(B, H, W_, Cin, generator=g).half().to(DEV)
        wk = (torch.randn(Cout, 9 * Cin, generator=g) / (9 * Cin) ** 0.5).half().to(DEV)
        out = torch.empty(B, H, W_, Cout, dtype=torch.float16, device=DEV)
        row = []
        for tile, st in [(128, 9), (128, 13), (256, 0)]:
            ops.GEMM_STAGES = st
            us = timeit(lambda: ops.conv3x3_nhwc(x, wk, out, None, tile=tile), reps=10)
            row.append(f"t{tile}s{st} {us:8.1f}us/{2.0*B*H*W_*Cout*9*Cin/us/1e6:5.0f}T")
        ops.GEMM_STAGES = 0
        print(f"{label:20s} " + "  ".join(row))


if __name__ == "__main__":
    main()
