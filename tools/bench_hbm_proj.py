#!/usr/bin/env python3
"""fp32 + residual projections (the HBM-bound launches of the step) on every tile kernel: us, TFLOP/s, algorithmic TB/s.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops
from tools.bench_gemm import timeit
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for M, N, K, label in [(21532, 768, 768, "dec proj+res"), (107520, 1024, 1024, "enc proj+res"), (21532, 768, 3072, "dec fc2+res")]:
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    byts = M * K * 2 + 2 * M * N * 4
    for tile in (256, 128, 192, 64):
        try:
            us = min(timeit(lambda: ops.linear(A, W, out, b, 0, res1=res, tile=tile), reps=20) for _ in range(3))
            print(f"{label:14s} {M}x{N}x{K} tile {tile:3d}: {us:8.1f} us  {2*M*N*K/us/1e6:6.0f} TF/s  {byts/us/1e6:6.2f} TB/s", flush=True)
        except Exception as e:
            print(label, tile, "error", str(e)[:100])
