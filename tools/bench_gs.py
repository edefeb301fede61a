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
          f"forward {tf[len(tf) // 2]:7.2f} ms, backward {tb[len(tb) // 2]:7.2f} ms (median of {args.iters}, incl. the host-side read of the instance count)")

# ---- the mapper on a synthetic wall at the tracking resolution (384x512): one 6-keyframe window through GSMapper.run with the
#      reference's iteration counts (100 initial, per new keyframe 50 pose-refine + 20 window + 50 single-view, 10 per view global)
if os.environ.get("CUT3R_BENCH_GS_MAPPER", "1") == "1":
    import time
    from cut3r_slam_amd import gs_mapper as GM
    from cut3r_slam_amd.lietorch import SE3
    Hm, Wm, F = 384, 512, 440.0
    cfg = {"Training": {"lambda_depth": 10.0, "lambda_normal": 0.1, "lambda_iso": 10.0, "gaussian_th": 0.05, "gaussian_extent": 1.0, "size_threshold": 20,
                        "window_size": 10},
           "opt_params": {"pose_lr": 0.0001, "position_lr_init": 0.0005, "feature_lr": 0.005, "opacity_lr": 0.05, "scaling_lr": 0.001,
                          "rotation_lr": 0.001, "percent_dense": 0.01, "densify_grad_threshold": 0.0005}}
    ys, xs = torch.meshgrid(torch.linspace(-1.7, 1.7, 384), torch.linspace(-2.3, 2.3, 512), indexing="ij")
    z = 3.0 + 0.2 * torch.sin(xs) * torch.cos(1.3 * ys)
    col = torch.stack([0.5 + 0.4 * torch.sin(3 * xs), 0.5 + 0.4 * torch.cos(2.5 * ys), 0.5 + 0.4 * torch.sin(2 * xs + 3 * ys)], -1)
    truth = GM.GaussianMap(cfg["opt_params"], DEV)
    truth.extend_from_pcd_seq(0, rgb=col.reshape(-1, 3), pointmap=torch.stack([xs, ys, z], -1).reshape(-1, 3))
    with torch.no_grad():
        truth.p["opacity"].fill_(2.2)
        truth.p["scaling"] += 0.26
    poses = [SE3.exp(torch.tensor([[0.05 * k, 0.01 * (k % 2), 0.0, 0.0, -0.01 * k, 0.0]], device=DEV)).data[0].cpu() for k in range(6)]
    imgs, depths, pms = [], [], []
    yy, xx = torch.meshgrid(torch.arange(Hm, device=DEV).float(), torch.arange(Wm, device=DEV).float(), indexing="ij")
    for p in poses:
        T = GM.pose_vec_to_matrix(p[None].to(DEV))[0]
        cam = GM.Camera(0, torch.zeros(3, Hm, Wm), torch.ones(Hm, Wm), torch.inverse(T), F, F, Wm / 2, Hm / 2, device=DEV)
        with torch.no_grad():
            pkg = GM.render(cam, truth, torch.zeros(3, device=DEV))
        d = pkg["depth"][0]
        imgs.append((pkg["render"].clamp(0, 1) * 255).round().to(torch.uint8))
        depths.append(d)
        pc = torch.stack([(xx - Wm / 2) / F * d, (yy - Hm / 2) / F * d, d], -1)
        pms.append((pc @ T[:3, :3].T + T[:3, 3])[::2, ::2])
    packet = {"viz_idx": list(range(6)), "submap_idx": 0, "tstamp": torch.arange(6).float(), "poses": torch.stack(poses), "images": torch.stack(imgs),
              "pointmaps": torch.stack(pms), "confs": torch.ones(6, Hm // 2, Wm // 2, device=DEV), "depths": torch.stack(depths),
              "intrinsics": torch.tensor([F, F, Wm / 2, Hm / 2])}
    mapper = GM.GSMapper(cfg, F, F, Wm / 2, Hm / 2, downsample_ratio=2, device=DEV)
    torch.cuda.synchronize()
    t0 = time.time()
    mapper.run(packet, iterations=100)
    torch.cuda.synchronize()
    dt = time.time() - t0
    with torch.no_grad():
        ps = []
        for k in range(6):
            r = GM.render(mapper.viewpoints[k], mapper.gaussians, torch.zeros(3, device=DEV))["render"]
            ps.append(float(-10 * torch.log10(((r - imgs[k].float() / 255) ** 2).mean())))
    renders = 100 + 5 * (50 + 50) + sum(20 * min(k + 1, 10) for k in range(1, 6)) + 10 * 6
    print(f"mapper, one 6-keyframe window at {Wm}x{Hm}: {dt:.2f} s = {1e3 * dt / 6:.0f} ms per keyframe, about {renders} forward+backward renders "
          f"({1e3 * dt / renders:.2f} ms each incl. losses and optimiser), {len(mapper.gaussians)} Gaussians, PSNR {sum(ps) / 6:.1f} dB")
