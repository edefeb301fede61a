#!/usr/bin/env python3
"""The 256^2 GEMM kernel on the 28-window shapes of the step (M = 107520 encoder rows, 21532 decoder rows), epilogues as the model uses them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import _lib
if len(sys.argv) > 2 and sys.argv[1] == "--lib":        # A/B arm: another build of the library
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from cut3r_slam_amd import ops
from tools.bench_gemm import timeit

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
ZEROS = "--zeros" in sys.argv          # all-zero operands: the same instruction stream at the clock the chip holds without data toggling
print("library:", _lib.LIB_PATH, "| operands:", "zeros" if ZEROS else "random")
CASES = [  # M, N, K, act, residual(fp32, in place), out dtype, label
    (107520, 4096, 1024, 1, False, torch.float16, "enc fc1 gelu"), (107520, 1024, 4096, 0, True, torch.float32, "enc fc2+res"),
    (107520, 3072, 1024, 0, False, torch.float16, "enc qkv"), (107520, 1024, 1024, 0, True, torch.float32, "enc proj+res"),
    (21532, 3072, 768, 1, False, torch.float16, "dec fc1 gelu"), (21532, 2304, 768, 0, False, torch.float16, "dec qkv"),
    (21532, 768, 3072, 0, True, torch.float32, "dec fc2+res"), (21532, 768, 768, 0, True, torch.float32, "dec proj+res"),
]
for M, N, K, act, use_res, odt, label in CASES:
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    if ZEROS:
        A.zero_(); W.zero_(); b.zero_()
    out = torch.zeros(M, N, dtype=odt, device=DEV)
    us = min(timeit(lambda: ops.linear(A, W, out, b, act, out if use_res else None, tile=256), reps=20) for _ in range(3))
    print(f"{label:16s} {M:6d}x{N:5d}x{K:5d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:6.0f} TF/s", flush=True)
    if "qkv" in label:          # with the fused 2-D RoPE epilogue on q and k (64-wide heads; tokens of 24 x 32 patch grids)
        n = torch.arange(M, device=DEV)
        pos = torch.stack([(n // 32) % 24, n % 32], 1).contiguous()
        us = min(timeit(lambda: ops.linear(A, W, out, b, 0, None, tile=256, rope=(pos, 2 * N // 3, 100.0)), reps=20) for _ in range(3))
        print(f"{label + ' + rope':16s} {M:6d}x{N:5d}x{K:5d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:6.0f} TF/s", flush=True)
