#!/usr/bin/env python3
"""A few launches of the two large-tile GEMM kernels on one encoder and one decoder shape (target of rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(30720, 4096, 1024), (6152, 768, 768), (6152, 3072, 768)]:
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    for tile in (128, 256):
        for _ in range(5):
            ops.linear(A, W, out, None, 0, tile=tile)
    torch.cuda.synchronize()
