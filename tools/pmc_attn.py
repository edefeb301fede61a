#!/usr/bin/env python3
"""A few launches of the attention kernel on the stack's shapes (target of rocprofv3 --pmc / --kernel-trace runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for (B, H, Nq, Nk, D) in [(40, 16, 768, 768, 64), (8, 12, 769, 769, 64), (8, 12, 769, 768, 64), (8, 16, 768, 768, 48), (8, 16, 768, 769, 48)]:
    q = torch.randn(B, Nq, H, D, generator=g).half().to(DEV)
    k = torch.randn(B, Nk, H, D, generator=g).half().to(DEV)
    v = torch.randn(B, Nk, H, D, generator=g).half().to(DEV)
    o = torch.empty_like(q)
    for _ in range(5):
        ops.attention(q, k, v, o, D ** -0.5)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.attention(q, k, v, o, D ** -0.5)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"B{B} H{H} Nq{Nq} Nk{Nk} D{D}: {us:7.1f} us  {4.0 * B * H * Nq * Nk * D / us / 1e6:6.0f} TF/s")
