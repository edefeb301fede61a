#!/usr/bin/env python3
"""bit-equality of one GEMM across tile kernels, for every epilogue the model uses (debug / regression aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for (M, N, K) in ((1536, 1024, 1024), (2000, 768, 768), (3840, 3072, 1024)):
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r32 = torch.randn(M, N, generator=g).to(DEV)
    for act, res, odt in ((0, None, torch.float16), (1, None, torch.float16), (0, r32, torch.float32), (0, r32.half(), torch.float16), (2, None, torch.float32), (0, None, torch.float32)):
        outs = {}
        for tile in (64, 128, 256, 192128, 128192, 256128):
            o = torch.zeros(M, N, dtype=odt, device=DEV)
            try:
                ops.linear(A, W, o, b, act, res, tile=tile)
            except Exception as e:
                continue
            torch.cuda.synchronize()
            outs[tile] = o
        ref = outs[128]
        bad = {t: int((o != ref).sum()) for t, o in outs.items() if not torch.equal(o, ref)}
        print(f"M{M} N{N} K{K} act{act} res{None if res is None else res.dtype} out{odt}: tiles {sorted(outs)} mismatching elements vs 128: {bad}")
