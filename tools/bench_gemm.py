#!/usr/bin/env python3
"""Micro-benchmark of the GEMM family on the shapes of the CUT3R stack (run on the GPU box)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops

DEV = "cuda:0"
SHAPES = [  # (M, N, K, label)
    (4608, 3072, 1024, "enc qkv B6"), (4608, 1024, 1024, "enc proj B6"), (4608, 4096, 1024, "enc fc1 B6"),
    (4608, 1024, 4096, "enc fc2 B6"), (3840, 3072, 1024, "enc qkv B5"), (3840, 1024, 1024, "enc proj B5"), (3840, 1024, 4096, "enc fc2 B5"),
    (768, 3072, 1024, "enc qkv B1"), (768, 1024, 4096, "enc fc2 B1"), (769, 768, 768, "dec proj"), (769, 2304, 768, "dec qkv"), (769, 768, 3072, "dec fc2 "),
    (769, 2304, 768, "dec qkv"), (769, 768, 768, "dec proj"), (769, 3072, 768, "dec fc1"), (769, 768, 3072, "dec fc2"),
    (768, 1536, 768, "dec kv"), (256, 4608, 1536, "mem qkv"), (1, 1536, 1536, "mem M=1"),
]


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us


def main():
    g = torch.Generator().manual_seed(0)
    CFG = [(64, 3), (64, 8), (128, 4), (128, 9), (128, 10), (256128, 2)]
    print(f"{'shape':34s} " + " ".join(f"t{t}s{s:>1d}".rjust(12) for t, s in CFG))
    for M, N, K, label in SHAPES:
        A = torch.randn(M, K, generator=g).half().to(DEV)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        row = []
        for tile, st in CFG:
            ops.GEMM_STAGES = st
            us = timeit(lambda: ops.linear(A, W, out, b, 0, tile=tile))
            row.append(f"{us:6.1f}us/{2.0*M*N*K/us/1e6:5.0f}T")
        ops.GEMM_STAGES = 0
        print(f"{label:12s} {M:5d}x{N:5d}x{K:5d}  " + " ".join(r.rjust(12) for r in row))
    print("--- z-batched small GEMMs (co-residency of independent problems): us per launch / us per problem")
    for M, N, K, label in [(769, 768, 768, "dec proj"), (769, 2304, 768, "dec qkv"), (769, 768, 3072, "dec fc2"), (1, 1536, 1536, "mem M=1")]:
        row = []
        for Z in (1, 2, 4, 8):
            A = torch.randn(Z, M, K, generator=g).half().to(DEV)
            W = (torch.randn(Z, N, K, generator=g) / K ** 0.5).half().to(DEV)
            out = torch.empty(Z, M, N, dtype=torch.float16, device=DEV)
            us = timeit(lambda: ops.linear_batched(A, W, out, tile=64))
            row.append(f"Z{Z}: {us:6.1f}/{us/Z:5.1f}")
        print(f"{label:10s} {M}x{N}x{K}  " + "   ".join(row))
    # two streams
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    M, N, K = 769, 768, 768
    A = torch.randn(M, K, generator=g).half().to(DEV); W = torch.randn(N, K, generator=g).half().to(DEV)
    o1 = torch.empty(M, N, dtype=torch.float16, device=DEV); o2 = torch.empty_like(o1)
    def two():
        with torch.cuda.stream(s1):
            ops.linear(A, W, o1, tile=64)
        with torch.cuda.stream(s2):
            ops.linear(A, W, o2, tile=64)
    torch.cuda.synchronize()
    import time
    for _ in range(5): two()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): two()
    torch.cuda.synchronize(); print(f"2 streams x dec proj: {(time.perf_counter()-t0)/200*1e6:.1f} us per pair (wall)")
    t0 = time.perf_counter()
    for _ in range(400): ops.linear(A, W, o1, tile=64)
    torch.cuda.synchronize(); print(f"1 stream dec proj: {(time.perf_counter()-t0)/400*1e6:.1f} us per launch (wall, incl. host)")
    # convs of the DPT head (B=6)
    for (B, H, W_, Cin, Cout, label) in [(6, 192, 256, 256, 256, "rcu @192x256"), (6, 384, 512, 128, 128, "head.2 @384x512"),
                                         (6, 96, 128, 256, 256, "rcu @96x128"), (6, 192, 256, 256, 128, "head.0")]:
        x = torch.randn(B, H, W_, Cin, generator=g).half().to(DEV)
        wk = (torch.randn(Cout, 9 * Cin, generator=g) / (9 * Cin) ** 0.5).half().to(DEV)
        out = torch.empty(B, H, W_, Cout, dtype=torch.float16, device=DEV)
        row = []
        for tile, st in [(128, 9), (128, 10), (256128, 2)]:
            ops.GEMM_STAGES = st
            us = timeit(lambda: ops.conv3x3_nhwc(x, wk, out, None, tile=tile), reps=10)
            row.append(f"t{tile}s{st} {us:8.1f}us/{2.0*B*H*W_*Cout*9*Cin/us/1e6:5.0f}T")
        ops.GEMM_STAGES = 0
        print(f"{label:20s} " + "  ".join(row))


if __name__ == "__main__":
    main()
