#!/usr/bin/env python3
"""isolated timing of the memory-bound launches of a decoder block at window batch WB: RoPE (table-driven pair launch) and LayerNorm"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops

WB = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for H, D, N in ((12, 64, 769), (16, 48, 768)):
    qkv = torch.randn(WB, N, 3, H, D, device=dev).half()
    pos = torch.randint(-1, 32, (WB, N, 2), device=dev)
    us = timeit(lambda: ops.rope_2d_pair(qkv[:, :, 0], pos, qkv[:, :, 1], pos, 100.0, 1.0))
    mb = 2 * 2 * WB * N * H * D * 2 / 1e6
    print(f"rope pair   [{WB},{N},{H},{D}] x2: {us:7.1f} us  {mb / us:6.2f} TB/s ({mb:.0f} MB)")
for C in (768, 1024):
    M = WB * 769 if C == 768 else 5 * WB * 768
    x = torch.randn(M, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    o = torch.empty(M, C, device=dev, dtype=torch.float16)
    us = timeit(lambda: ops.layernorm(x, g, b, 1e-6, o, None))
    mb = M * C * 6 / 1e6
    print(f"layernorm   [{M},{C}] fp32 -> fp16: {us:7.1f} us  {mb / us:6.2f} TB/s ({mb:.0f} MB)")
