#!/usr/bin/env python3
"""Is the fp32 + residual epilogue of the 256 x 256 kernel bound per CU or by the chip?  ONE round of tiles on 64 / 128 / 192 / 256 CUs
(K = 1024 and K = 64: the second is almost pure epilogue): if a tile takes as long on 64 CUs as on 256, the limit is per CU."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops
from tools.bench_gemm import timeit

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for K in (1024, 64):
    for tiles in (64, 128, 192, 256, 512):
        M, N = 256 * tiles, 256
        A = torch.randn(M, K, generator=g).half().to(DEV)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        out16 = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        t3 = min(timeit(lambda: ops.linear(A, W, out, b, 0, out, tile=256), reps=20) for _ in range(3))
        t1 = min(timeit(lambda: ops.linear(A, W, out16, b, 0, None, tile=256), reps=20) for _ in range(3))
        print(f"K={K:5d} tiles={tiles:4d}: fp32+residual {t3:7.1f} us ({512 * tiles / t3 / 1e3:6.2f} TB/s of epilogue traffic) | fp16 {t1:7.1f} us", flush=True)
