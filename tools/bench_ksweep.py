#!/usr/bin/env python3
"""time vs K at fixed (M, N): slope = cost per 64-deep K-tile, intercept = fixed launch/prologue/epilogue cost"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops
from tools.bench_gemm import timeit

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
CFG = [(128, 9), (128, 10), (128, 13), (64, 3), (64, 2)]
for (M, N) in [(3076, 768), (3076, 2304), (15360, 1024)]:
    print(f"M={M} N={N}   " + " ".join(f"t{t}s{s}".rjust(10) for t, s in CFG))
    for K in [64, 128, 256, 512, 768, 1536, 3072]:
        A = torch.randn(M, K, generator=g).half().to(DEV)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        row = []
        for tile, st in CFG:
            ops.GEMM_STAGES = st
            row.append(timeit(lambda: ops.linear(A, W, out, b, 0, tile=tile), reps=100))
        ops.GEMM_STAGES = 0
        print(f"  K={K:5d}  " + " ".join(f"{u:8.1f}us" for u in row))
