#!/usr/bin/env python3
"""A/B of the 128^2 and 256^2 GEMM kernels on the large shapes of the stack, with the epilogues the model uses."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops
from tools.bench_gemm import timeit

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
CASES = [  # M, N, K, act, residual(fp32, in place), out dtype, label
    (15360, 3072, 1024, 0, False, torch.float16, "enc qkv"), (15360, 1024, 1024, 0, True, torch.float32, "enc proj+res"),
    (15360, 4096, 1024, 1, False, torch.float16, "enc fc1 gelu"), (15360, 1024, 4096, 0, True, torch.float32, "enc fc2+res"),
    (6152, 2304, 768, 0, False, torch.float16, "dec qkv W8"), (6152, 3072, 768, 1, False, torch.float16, "dec fc1 W8"),
    (6152, 768, 3072, 0, True, torch.float32, "dec fc2 W8"), (12304, 768, 768, 0, True, torch.float32, "dec proj W16"),
    (12304, 768, 3072, 0, True, torch.float32, "dec fc2 W16"), (12304, 3072, 768, 1, False, torch.float16, "dec fc1 W16"),
]
CFG = [(128, 0), (256, 0), (128, 0), (256, 0)]
print(f"{'case':16s} {'shape':22s} " + " ".join(f"t{t}".rjust(14) for t, _ in CFG))
for M, N, K, act, use_res, odt, label in CASES:
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    out = torch.zeros(M, N, dtype=odt, device=DEV)
    row = []
    for tile, st in CFG:
        us = timeit(lambda: ops.linear(A, W, out, b, act, out if use_res else None, tile=tile))
        row.append(f"{us:7.1f}us/{2.0*M*N*K/us/1e6:5.0f}T")
    print(f"{label:16s} {M:6d}x{N:5d}x{K:5d}  " + " ".join(r.rjust(14) for r in row))
for (B, H, W_, Cin, Cout, relu, label) in [(24, 192, 256, 256, 256, True, "rcu 192x256 B24"), (24, 96, 128, 256, 256, True, "rcu 96x128 B24"),
                                            (24, 48, 64, 256, 256, True, "rcu 48x64 B24")]:
    x = torch.randn(B, H, W_, Cin, generator=g).half().to(DEV)
    wk = (torch.randn(Cout, 9 * Cin, generator=g) / (9 * Cin) ** 0.5).half().to(DEV)
    out = torch.empty(B, H, W_, Cout, dtype=torch.float16, device=DEV)
    row = []
    for tile, st in CFG:
        us = timeit(lambda: ops.conv3x3_nhwc(x, wk, out, None, 1, relu, 0, res1=x, tile=tile), reps=10)
        row.append(f"{us:7.1f}us/{2.0*B*H*W_*Cout*9*Cin/us/1e6:5.0f}T")
    print(f"{label:39s}  " + " ".join(r.rjust(14) for r in row))
