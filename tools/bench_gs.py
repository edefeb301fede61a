#!/usr/bin/env python3
"""Gaussian rasteriser micro-benchmark (csrc/gs.hip): forward and forward+backward time of one 640x480 view of a synthetic room of
P Gaussians (SH degree 0, as the GS mapper runs it), HIP events on the current stream.
usage: python tools/bench_gs.py [--points 50000 200000 800000] [--size 480 640]"""
import argparse
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cut3r_slam_amd.gaussian_rasterizer import GaussianRasterizationSettings, GaussianRasterizer, distCUDA2

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, nargs="+", default=[50000, 200000, 800000])
ap.add_argument("--size", type=int, nargs=2, default=[480, 640])
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
DEV = "cuda:0"
H, W = args.size
fovx = 2 * math.atan(W / (2 * 600.0))
fovy = 2 * math.atan(H / (2 * 600.0))
tanx, tany = math.tan(fovx / 2), math.tan(fovy / 2)
znear, zfar = 0.01, 100.0
Pm = torch.zeros(4, 4)
Pm[0, 0], Pm[1, 1], Pm[3, 2], Pm[2, 2], Pm[2, 3] = 1 / tanx, 1 / tany, 1.0, zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)
view = torch.eye(4)
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tanx, tanfovy=tany, kernel_size=0.0, bg=torch.zeros(3, device=DEV),
                                   scale_modifier=1.0, viewmatrix=view.to(DEV), projmatrix=(view @ Pm.T).to(DEV), sh_degree=0,
                                   campos=torch.zeros(3, device=DEV), prefiltered=False, require_depth=True, require_coord=True, debug=False)
rast = GaussianRasterizer(st)
g = torch.Generator().manual_seed(0)
for P in args.points:
    # points on the walls of a 6 x 3 x 6 m room seen from inside, scales from the 3-NN distance as gaussian_model.py:189-195 does
    u = torch.rand(P, 3, generator=g)
    face = torch.randint(0, 5, (P,), generator=g)
    pts = torch.stack([(u[:, 0] - 0.5) * 6, (u[:, 1] - 0.5) * 3, u[:, 2] * 5 + 0.5], -1)
    pts[face == 0, 2] = 5.5
    pts[face == 1, 0] = -3.0
    pts[face == 2, 0] = 3.0
    pts[face == 3, 1] = -1.5
    pts[face == 4, 1] = 1.5
    pts = pts.to(DEV)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    distCUDA2(pts[:1000])                                             # (code object load, allocator warm-up)
    torch.cuda.synchronize()
    e0.record()
    d2 = distCUDA2(pts)
    e1.record()
    torch.cuda.synchronize()
    t_knn = e0.elapsed_time(e1)
    scales = torch.sqrt(d2.clamp_min(1e-7))[:, None].repeat(1, 3).requires_grad_(True)
    means = pts.clone().requires_grad_(True)
    rots = torch.tensor([[1.0, 0, 0, 0]], device=DEV).repeat(P, 1).requires_grad_(True)
    opac = torch.full((P, 1), 0.5, device=DEV, requires_grad=True)
    shs = (torch.rand(P, 1, 3, generator=g).to(DEV) - 0.5).requires_grad_(True)
    m2d = torch.zeros(P, 3, device=DEV, requires_grad=True)

    def fwd():
        return rast(means3D=means, means2D=m2d, opacities=opac, shs=shs, scales=scales, rotations=rots)

    outs = fwd()
    (outs[0].sum() + outs[4].sum() + outs[7].sum()).backward()
    torch.cuda.synchronize()
    n_inst = int(rast and outs[1].gt(0).sum())
    from cut3r_slam_amd.gaussian_rasterizer import _forward as _raw_forward
    with torch.no_grad():
        _, _, _buf = _raw_forward(means, shs, None, opac, scales, rots, st)
    pairs = _buf.n_inst * 256                                         # (pixel, Gaussian) pairs the render kernels walk at most
    tf, tb = [], []
    for _ in range(args.iters):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record()
        outs = fwd()
        b.record()
        (outs[0].sum() + outs[4].sum() + outs[7].sum()).backward()
        c.record()
        torch.cuda.synchronize()
        tf.append(a.elapsed_time(b))
        tb.append(b.elapsed_time(c))
    tf.sort(), tb.sort()
    cover = float((outs[6] > 0.5).float().mean())
    print(f"P={P:8d} {W}x{H}: visible {n_inst:8d}, alpha>0.5 on {100 * cover:5.1f} % of the image, 3-NN scales {t_knn:8.2f} ms, "
          f"forward {tf[len(tf) // 2]:7.2f} ms, backward {tb[len(tb) // 2]:7.2f} ms (median of {args.iters}, incl. the host-side read of the instance count); "
          f"{_buf.n_inst} tile instances = {pairs / 1e6:.0f} M pixel-Gaussian pairs per pass")

# ---- the mapper on a synthetic wall at the tracking resolution (384x512): one 6-keyframe window through GSMapper.run with the
#      reference's iteration counts (the same leg bench.py reports as operating_points.gs_mapper_synthetic_window)
if os.environ.get("CUT3R_BENCH_GS_MAPPER", "1") == "1":
    import time
    from cut3r_slam_amd import synth, gs_mapper as GM
    r = synth.gs_mapper_window_leg(384, 512, DEV)
    print(f"mapper, one 6-keyframe window at 512x384: {r['seconds']:.2f} s = {r['ms_per_keyframe']:.0f} ms per keyframe, about {r['render_iterations']} "
          f"forward+backward renders ({r['ms_per_render_iteration']:.2f} ms each incl. losses and optimiser), {r['gaussians']} Gaussians, "
          f"PSNR {r['psnr_db']:.1f} dB")
    # a long loop over a fixed set of Gaussians (a final refinement): eager vs one captured iteration replayed
    packet, imgs, cfg = synth.gs_wall_window(384, 512, device=DEV)
    for graphs in (False, True):
        m = GM.GSMapper(cfg, 440.0, 440.0, 256.0, 192.0, downsample_ratio=2, device=DEV)
        m.run(packet, iterations=5, init_iters=20, gba_per_view=1)
        torch.cuda.synchronize()
        t0 = time.time()
        loss = m.optimization(400, optimize_pose=True, current_window=[0, 1, 2], graph=graphs)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"mapper, 400 iterations over 3 views, {'captured and replayed' if graphs else 'eager'}: {dt:.2f} s = {1e3 * dt / 1200:.2f} ms per render "
              f"iteration, loss {loss:.4f}")
