#!/usr/bin/env python3
"""Loop-closure optimiser (cut3r_lc_optimize: fused Adam over per-submap se(3) corrections, track_backend.py:256-299): iterations/s
and the fraction of the HBM roof for B submaps of N = 192 x 256 points.  Algorithmic traffic per iteration: 2 * B * N * 12
bytes (the last pointmap of submap b and the first of submap b+1), read once by the fused accumulate kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops

dev = "cuda:0"
h, w = 192, 256
N = h * w
iters = 1000
for B in (20, 80, 200):
    g = torch.Generator().manual_seed(B)
    sub = (torch.randn(B, 6, h, w, 3, generator=g) * 0.5 + 2.0).to(dev).contiguous()
    sub[1:, 0] = sub[:-1, 5] + 0.01 * torch.randn(B - 1, h, w, 3, generator=g).to(dev)       # consecutive submaps share a keyframe
    cur = sub[-1, 5].reshape(-1, 3).contiguous()
    cur_lc = (cur + 0.02).contiguous()
    ops.lc_optimize(sub, None, cur, cur_lc, 10)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.lc_optimize(sub, None, cur, cur_lc, iters)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    byt = 2.0 * B * N * 12
    print(f"B={B:4d}: {iters} iterations in {ms:8.1f} ms = {iters / ms * 1e3:8.0f} iterations/s, {ms * 1e3 / iters:6.1f} us per iteration, "
          f"{byt / 1e6:6.1f} MB per iteration -> {byt * iters / ms / 1e9:6.2f} TB/s = {byt * iters / ms / 1e9 / 8.0:.3f} of 8 TB/s")
