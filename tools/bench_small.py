#!/usr/bin/env python3
"""Micro-benchmark of the memory-bound / latency-bound kernels at the shapes of the batched decoder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops
from tools.bench_gemm import timeit
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for M, C in [(3076, 768), (3072, 768), (15360, 1024), (769, 768), (1024, 1536)]:
    x = torch.randn(M, C, generator=g).to(DEV); w = torch.randn(C).to(DEV); b = torch.randn(C).to(DEV)
    o16 = torch.empty(M, C, dtype=torch.float16, device=DEV)
    us = timeit(lambda: ops.layernorm(x, w, b, 1e-6, o16, None))
    print(f"layernorm {M}x{C}: {us:6.1f} us  {(M*C*6)/us/1e6:6.2f} TB/s")
for B, N, H, D in [(4, 769, 12, 64), (4, 768, 16, 48), (20, 768, 16, 64), (1, 769, 12, 64)]:
    qkv = torch.randn(B, N, 3, H, D, generator=g).half().to(DEV)
    pos = torch.randint(-1, 32, (B, N, 2), generator=g).to(DEV)
    us = timeit(lambda: ops.rope_2d_qk(qkv[:, :, 0], qkv[:, :, 1], pos, 100.0, 1.0))
    print(f"rope_qk B{B} N{N} H{H} D{D}: {us:6.1f} us  {(B*N*H*D*2*2*2)/us/1e6:6.2f} TB/s")
    out = torch.empty(B, N, H, D, dtype=torch.float16, device=DEV)
    us = timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out, D ** -0.5))
    print(f"attention B{B} N{N} H{H} D{D}: {us:6.1f} us  {4.0*B*H*N*N*D/us/1e6:6.1f} TFLOP/s")
x = torch.randn(768, 1024, generator=g).to(DEV); y = torch.empty(1024, device=DEV)
print(f"colmean 768x1024: {timeit(lambda: ops.colmean(x, y)):6.1f} us")
