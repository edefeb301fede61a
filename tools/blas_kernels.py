#!/usr/bin/env python3
"""Which hipBLASLt / rocBLAS kernels torch picks for the stack's GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(6152, 768, 768), (6152, 768, 3072), (6152, 2304, 768), (6152, 3072, 768), (30720, 4096, 1024), (30720, 1024, 4096)]:
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = torch.randn(N, K, generator=g).half().to(DEV)
    b = torch.randn(N, generator=g).half().to(DEV)
    for _ in range(3):
        torch.nn.functional.linear(A, W, b)
    torch.cuda.synchronize()
