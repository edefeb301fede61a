#!/usr/bin/env python3
"""isolated timing of the covisibility counting kernels (cut3r_overlap_fwd: one full-resolution pointmap against B cameras;
cut3r_overlap_bwd: B stride-2 pointmaps against one camera), optionally against another build of the library (--lib)"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
args = ap.parse_args()
from cut3r_slam_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in ("cut3r_abi_version", "cut3r_overlap_fwd", "cut3r_overlap_bwd")}
from cut3r_slam_amd import ops  # noqa: E402

print("library:", _lib.LIB_PATH)
dev = "cuda:0"
H, W = 384, 512
g = torch.Generator().manual_seed(0)
K4 = [256.0, 211.8, 255.8, 191.6]
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
d = 2 + torch.rand(H, W, generator=g)
pm = torch.stack([(xs - K4[2]) / K4[0] * d, (ys - K4[3]) / K4[1] * d, d], -1).contiguous().to(dev)
for B in (200, 1000):
    w2c = torch.eye(4)[:3].reshape(1, 12).repeat(B, 1)
    w2c[:, 3] = torch.randn(B, generator=g) * 0.5          # cameras displaced sideways: partial overlap
    w2c[:, 7] = torch.randn(B, generator=g) * 0.3
    w2c = w2c.contiguous().to(dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.overlap_fwd(pm, w2c, K4, W, H, cnt)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.overlap_fwd(pm, w2c, K4, W, H, cnt)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"overlap_fwd N={H * W} B={B}: {us:8.1f} us  {H * W * B / us / 1e3:7.1f} G proj/s  mean ratio {cnt.float().mean().item() / (H * W):.3f}")
    store = pm.view(H, W, 3)[::2, ::2].contiguous().view(1, -1, 3).repeat(B, 1, 1).contiguous()
    cb = torch.zeros(B, dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.overlap_bwd(store, w2c[0].contiguous(), K4, W // 2, H // 2, cb, B=B, N=store.shape[1])
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        ops.overlap_bwd(store, w2c[0].contiguous(), K4, W // 2, H // 2, cb, B=B, N=store.shape[1])
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"overlap_bwd N={store.shape[1]} B={B}: {us:8.1f} us  {store.shape[1] * B / us / 1e3:7.1f} G proj/s  {store.numel() * 4 / us / 1e6:6.2f} TB/s")
