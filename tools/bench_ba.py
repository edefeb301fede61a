#!/usr/bin/env python3
"""Dense-BA step micro-benchmark (csrc/ba.hip; hislam2/geom/ba.py:32-107 `BA`): a DROID-sized problem -- P frames at 1/8 resolution
(48x64), every frame linked to its neighbours within 3 -- one Gauss-Newton step = per-edge Jacobian / Hessian assembly, Schur
complement, in-LDS Cholesky, back-substitution.  HIP events around the step; run under `rocprofv3 --kernel-trace --stats` for the kernels.
usage: python tools/bench_ba.py [--frames 12 24] [--size 48 64]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cut3r_slam_amd.ba import BA
from cut3r_slam_amd.lietorch import SE3

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, nargs="+", default=[12, 24])
ap.add_argument("--size", type=int, nargs=2, default=[48, 64])
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
DEV = "cuda:0"
ht, wd = args.size
for P in args.frames:
    g = torch.Generator().manual_seed(P)
    tw = torch.zeros(P, 6)
    tw[:, 0] = torch.linspace(0, 0.05 * P, P)
    tw[:, 1:] = 0.02 * torch.randn(P, 5, generator=g)
    poses = SE3.exp(tw.to(DEV))
    disps = (0.3 + 0.3 * torch.rand(P, ht, wd, generator=g)).to(DEV)
    intr = torch.tensor([0.8 * wd, 0.8 * wd, wd / 2 - 0.5, ht / 2 - 0.5]).repeat(P, 1).to(DEV)
    ii, jj = zip(*[(i, j) for i in range(P) for j in range(P) if i != j and abs(i - j) <= 3])
    ii, jj = torch.tensor(ii), torch.tensor(jj)
    N = len(ii)
    ys, xs = torch.meshgrid(torch.arange(ht).float(), torch.arange(wd).float(), indexing="ij")
    tgt = (torch.stack([xs, ys], -1)[None] + 2.0 * torch.randn(N, ht, wd, 2, generator=g)).to(DEV)
    wgt = (0.2 + 0.8 * torch.rand(N, ht, wd, 2, generator=g)).to(DEV)
    eta = (1e-3 + 9e-3 * torch.rand(len(set(ii.tolist())), ht, wd, generator=g)).to(DEV)

    def step():
        return BA(tgt[None], wgt[None], eta, SE3(poses.data[None]), disps[None], intr[None], ii, jj, fixedp=2)
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    assert int(out[2]["failed"].item()) == 0
    # algorithmic bytes of the assembly: per edge and pixel target 8 + weight 8 + disparity 4 in, the 12 floats of the depth/pose coupling out
    by = N * ht * wd * (8 + 8 + 4 + 48)
    print(f"P={P:3d} frames, {N:4d} edges, {wd}x{ht}: BA step {ts[len(ts) // 2]:7.3f} ms (median of {args.iters}; assembly streams ~{by / 1e6:.1f} MB)")
