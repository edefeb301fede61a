#!/usr/bin/env python3
"""Round-2 GEMM A/B on the shapes that dominate a tracking step (window batch 8: encoder M = 40 x 768, decoder M = 8 x 768|769),
with the epilogues the model uses (fp16 out / GELU / fp32 out + fp32 residual in place).

Method (cdna_hip_programming.md rule 24): every variant of a case is captured into its own hipGraph of `reps` launches; the graphs
are replayed round-robin for several rounds in ONE process and the median round is reported.  Operands are random.

usage: python tools/bench_gemm_r2.py [--lib path/to/other/libcut3r_hip.so] [variant ...]
       variant = tile[:stages], e.g. 0 128 256 128:10 192128          (0 = the library's own choice)
"""
import argparse
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None, help="alternative build of the C-ABI library (only the GEMM entry points are bound)")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--wb", type=int, default=8, help="window batch: decoder M = wb x 768|769")
ap.add_argument("--cases", default="", help="comma-separated substrings selecting cases")
ap.add_argument("variants", nargs="*")
args = ap.parse_args()

from cut3r_slam_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in ("cut3r_abi_version", "cut3r_gemm_f16", "cut3r_gemm_tile_for")}
from cut3r_slam_amd import ops  # noqa: E402

DEV = "cuda:0"
F16, F32 = torch.float16, torch.float32
CASES = [  # M, N, K, act, residual, out dtype, label
    (30720, 3072, 1024, 0, False, F16, "enc qkv"), (30720, 1024, 1024, 0, True, F32, "enc proj+res"),
    (30720, 4096, 1024, 1, False, F16, "enc fc1 gelu"), (30720, 1024, 4096, 0, True, F32, "enc fc2+res"),
    (args.wb * 769, 768, 768, 0, True, F32, "dec proj+res"), (args.wb * 769, 768, 768, 0, False, F16, "dec projq"),
    (args.wb * 768, 1536, 768, 0, False, F16, "dec projkv"), (args.wb * 769, 2304, 768, 0, False, F16, "dec qkv"),
    (args.wb * 769, 3072, 768, 1, False, F16, "dec fc1 gelu"), (args.wb * 769, 768, 3072, 0, True, F32, "dec fc2+res"),
    (2048, 1536, 1536, 0, True, F32, "mem proj+res"), (2048, 4608, 1536, 0, False, F16, "mem qkv"),
    (2048, 6144, 1536, 1, False, F16, "mem fc1"), (2048, 1536, 6144, 0, True, F32, "mem fc2+res"),
    (768, 768, 768, 0, True, F32, "dec proj W1"), (769, 3072, 768, 1, False, F16, "dec fc1 W1"), (769, 768, 3072, 0, True, F32, "dec fc2 W1"),
]
CONVS = [(8, 384, 512, 128, 128, False, False, "head.2 384x512 B8"), (8, 192, 256, 256, 256, True, True, "rcu 192x256 B8"),
         (8, 192, 256, 256, 128, False, False, "head.0 192x256 B8"), (8, 96, 128, 256, 256, True, True, "rcu 96x128 B8")]
variants = []
for v in (args.variants or ["0", "128", "256", "192128"]):
    t, _, s = v.partition(":")
    variants.append((int(t), int(s or 0)))
sel = [c for c in args.cases.split(",") if c]


def graph_of(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn()
    gr.replay()
    torch.cuda.synchronize()
    return gr


def run_case(label, flops, fns, reps):
    graphs = []
    for fn in fns:
        try:
            graphs.append(graph_of(fn, reps))
        except Exception:
            graphs.append(None)
    times = [[] for _ in fns]
    for _ in range(args.rounds):
        for i, gr in enumerate(graphs):
            if gr is None:
                continue
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            gr.replay()
            e.record()
            torch.cuda.synchronize()
            times[i].append(s.elapsed_time(e) / reps * 1e3)
    cells = []
    for t in times:
        if not t:
            cells.append("n/a")
        else:
            us = statistics.median(t)
            cells.append(f"{us:7.1f}us/{flops / us / 1e6:5.0f}T")
    print(f"{label:34s} " + " ".join(c.rjust(15) for c in cells), flush=True)


def with_variant(tile, stages, fn):
    def go():
        ops.GEMM_STAGES = stages
        fn(tile)
        ops.GEMM_STAGES = 0
    return go


g = torch.Generator().manual_seed(0)
print(f"library: {_lib.LIB_PATH}")
print(f"{'case':34s} " + " ".join((f"t{t}" + (f":s{s}" if s else "")).rjust(15) for t, s in variants))
for M, N, K, act, use_res, odt, label in CASES:
    if sel and not any(c in label for c in sel):
        continue
    A = torch.randn(M, K, generator=g).half().to(DEV)
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    out = torch.zeros(M, N, dtype=odt, device=DEV)
    fns = [with_variant(t, s, lambda tile: ops.linear(A, Wt, out, b, act, out if use_res else None, tile=tile)) for t, s in variants]
    run_case(f"{label:14s} {M:6d}x{N:5d}x{K:5d}", 2.0 * M * N * K, fns, 30)
for (B, H, W_, Cin, Cout, relu, res, label) in CONVS:
    if sel and not any(c in label for c in sel):
        continue
    x = torch.randn(B, H, W_, Cin, generator=g).half().to(DEV)
    wk = (torch.randn(Cout, 9 * Cin, generator=g) / (9 * Cin) ** 0.5).half().to(DEV)
    out = torch.empty(B, H, W_, Cout, dtype=F16, device=DEV)
    fns = [with_variant(t, s, lambda tile: ops.conv3x3_nhwc(x, wk, out, None, 1, relu, 0, res1=x if res and Cin == Cout else None, tile=tile))
           for t, s in variants]
    run_case(label, 2.0 * B * H * W_ * Cout * 9 * Cin, fns, 8)
