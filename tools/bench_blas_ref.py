#!/usr/bin/env python3
"""Known-good reference on the same hardware: torch.nn.functional.linear (hipBLASLt / rocBLAS) on the stack's GEMM shapes,
next to this repo's kernels.  Measurement only -- the product never calls a BLAS library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops
from tools.bench_gemm import timeit

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
print(f"{'shape':26s} {'torch (BLAS)':>16s} {'ours auto':>16s} {'ours t128':>16s} {'ours t256':>16s} {'ours t192x128':>16s}")
for (M, N, K) in [(30720, 3072, 1024), (30720, 1024, 1024), (30720, 4096, 1024), (30720, 1024, 4096), (6152, 2304, 768), (6152, 768, 768),
                  (6152, 1536, 768), (6152, 3072, 768), (6152, 768, 3072), (98304, 256, 2304), (8192, 8192, 8192)]:
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).half().to(DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    bf = b.float()
    row = [timeit(lambda: torch.nn.functional.linear(A, W, b), reps=20)]
    for tile in (0, 128, 256, 192128):
        row.append(timeit(lambda: ops.linear(A, W, out, bf, 0, tile=tile), reps=20))
    print(f"{M:6d}x{N:5d}x{K:5d}       " + " ".join(f"{u:7.1f}us/{2.0 * M * N * K / u / 1e6:5.0f}T" for u in row))
