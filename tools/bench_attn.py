#!/usr/bin/env python3
"""Fused-attention micro-benchmark on the shapes of a tracking step (window batch 8), median of interleaved graph replays.
usage: python tools/bench_attn.py [--lib other/libcut3r_hip.so]"""
import argparse
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--zeros", action="store_true", help="all-zero q, k, v: the same instruction stream without data toggling (clock check)")
args = ap.parse_args()
from cut3r_slam_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in ("cut3r_abi_version", "cut3r_attention_f16")}
from cut3r_slam_amd import ops  # noqa: E402

DEV = "cuda:0"
CASES = [(140, 16, 768, 768, 64, "enc self 140 kf"), (28, 12, 769, 769, 64, "dec img self W28"), (28, 12, 769, 768, 64, "dec img cross W28"),
         (28, 16, 768, 768, 48, "dec state self W28"), (28, 16, 768, 769, 48, "dec state cross W28"), (28, 12, 256, 256, 128, "mem write self W28"),
         (40, 16, 768, 768, 64, "enc self"), (8, 12, 769, 769, 64, "dec img self"), (8, 12, 769, 768, 64, "dec img cross"),
         (8, 16, 768, 768, 48, "dec state self"), (8, 16, 768, 769, 48, "dec state cross"), (8, 12, 256, 256, 128, "mem write self"),
         (1, 12, 769, 769, 64, "dec img self W1"), (1, 16, 768, 769, 48, "dec state cross W1")]
g = torch.Generator().manual_seed(0)
print(f"library: {_lib.LIB_PATH}")
for B, H, Nq, Nk, D, label in CASES:
    q = torch.randn(B, Nq, H, D, generator=g).half().to(DEV)
    k = torch.randn(B, Nk, H, D, generator=g).half().to(DEV)
    v = torch.randn(B, Nk, H, D, generator=g).half().to(DEV)
    if args.zeros:
        q.zero_(); k.zero_(); v.zero_()
    o = torch.empty(B, Nq, H, D, dtype=torch.float16, device=DEV)
    fn = lambda: ops.attention(q, k, v, o, D ** -0.5)
    reps = 20
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
    ts = []
    for _ in range(args.rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        gr.replay()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / reps * 1e3)
    us = statistics.median(ts)
    fl = 4.0 * B * H * Nq * Nk * D
    # fp32 reference on a slice
    ref = torch.nn.functional.scaled_dot_product_attention(q[:1].float().transpose(1, 2), k[:1].float().transpose(1, 2), v[:1].float().transpose(1, 2))
    err = (o[:1].float().transpose(1, 2) - ref).abs().max().item()
    print(f"{label:20s} [{B},{H},{Nq}x{Nk},{D}] {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s  max|err| vs fp32 sdpa {err:.2e}", flush=True)
