"""Gaussian rasteriser (csrc/gs.hip, cut3r_slam_amd/gaussian_rasterizer.py) against the fp64 restatement oracle/gs_oracle.py of
thirdparty/diff-gaussian-rasterization (forward.cu:23-692, rasterizer_impl.cu:70-176).  The rasteriser takes hard decisions per
pixel and Gaussian (alpha >= 1/255, T < 1e-4, T > 0.5 for the median outputs, the ceil() of the radius): an fp32 evaluation flips a
few of them against fp64, so the comparison bounds the FRACTION of pixels beyond the tolerance as well as the typical error."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu

from cut3r_slam_amd.gaussian_rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from oracle import gs_oracle as GO  # noqa: E402

DEV = "cuda:0"


def _scene(P, seed, spread=0.8, smin=0.03, smax=0.25, behind=4):
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(P, 3, generator=g, dtype=torch.float64) * spread + torch.tensor([0.0, 0.0, 3.0], dtype=torch.float64)
    means[:behind, 2] = -1.0                                              # behind the camera: culled (auxiliary.h:170)
    scales = torch.rand(P, 3, generator=g, dtype=torch.float64) * (smax - smin) + smin
    scales[behind:behind + 3, 2] = 1e-5                                   # flat Gaussians: the ill-conditioned branch (forward.cu:138-150)
    q = torch.randn(P, 4, generator=g, dtype=torch.float64)
    q = q / q.norm(dim=-1, keepdim=True)
    op = torch.rand(P, 1, generator=g, dtype=torch.float64) * 0.85 + 0.1
    shs = torch.randn(P, 16, 3, generator=g, dtype=torch.float64) * 0.3
    return means, scales, q, op, shs


def _w2c(rx, ry, t):
    cx, sx, cy, sy = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry)
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
    M = torch.eye(4, dtype=torch.float64)
    M[:3, :3] = Rx @ Ry
    M[:3, 3] = torch.tensor(t, dtype=torch.float64)
    return M


def _settings(st):
    f = lambda t: t.float().to(DEV)
    return GaussianRasterizationSettings(image_height=st["image_height"], image_width=st["image_width"], tanfovx=st["tanfovx"], tanfovy=st["tanfovy"],
                                         kernel_size=st["kernel_size"], bg=f(st["bg"]), scale_modifier=st["scale_modifier"],
                                         viewmatrix=f(st["viewmatrix"]), projmatrix=f(st["projmatrix"]), sh_degree=st["sh_degree"],
                                         campos=f(st["campos"]), prefiltered=False, require_depth=True, require_coord=True, debug=False)


def _compare(got, ref, name, tol, max_bad):
    g, r = got.detach().double().cpu().numpy(), ref.detach().numpy()
    err = np.abs(g - r)
    bad = (err > tol * (1.0 + np.abs(r))).mean()
    print(f"[gs] {name}: median |err| {np.median(err):.2e}, 99th pct {np.percentile(err, 99):.2e}, beyond tolerance {100 * bad:.3f} %")
    assert np.median(err) < tol * 0.1 + 1e-7, name
    assert bad <= max_bad, f"{name}: {bad}"


@pytest.mark.parametrize("H,W,P,deg,ks,cam,precomp", [(40, 56, 80, 3, 0.0, (0.0, 0.0, (0.0, 0.0, 0.0)), False),
                                                      (64, 80, 300, 1, 0.1, (0.15, -0.2, (0.1, -0.05, 0.3)), False),
                                                      (33, 47, 150, 0, 0.0, (-0.1, 0.1, (0.0, 0.1, 0.0)), True),
                                                      (96, 128, 1200, 2, 0.0, (0.05, 0.05, (0.0, 0.0, 0.5)), False)])
def test_forward_outputs_match_restatement(H, W, P, deg, ks, cam, precomp):
    means, scales, q, op, shs = _scene(P, P + deg)
    st = GO.camera_settings(H, W, 1.0, 1.0 * H / W, _w2c(*cam), bg=(0.1, 0.2, 0.3), sh_degree=deg, kernel_size=ks)
    colors = torch.rand(P, 3, dtype=torch.float64) if precomp else None
    ref = GO.rasterize(means, op, scales, q, st, shs=None if precomp else shs, colors_precomp=colors)
    f = lambda t: t.float().to(DEV)
    rast = GaussianRasterizer(_settings(st))
    color, radii, coord, mcoord, depth, mdepth, alpha, normal = rast(
        means3D=f(means), means2D=torch.zeros(P, 3, device=DEV), opacities=f(op), shs=None if precomp else f(shs),
        colors_precomp=f(colors) if precomp else None, scales=f(scales), rotations=f(q))
    torch.cuda.synchronize()
    assert color.shape == (3, H, W) and depth.shape == (1, H, W) and alpha.shape == (1, H, W) and normal.shape == (3, H, W)
    rr, gr = ref["radii"].numpy(), radii.cpu().numpy()
    assert ((rr > 0) == (gr > 0)).mean() > 0.995 and (np.abs(rr - gr) <= 1).all()        # (ceil of an fp32 vs fp64 value)
    assert (gr[:4] == 0).all() and gr.max() > 0
    cover = float((ref["alpha"] > 0.5).double().mean())
    assert cover > 0.3, cover                                              # the scene covers a good part of the image
    _compare(color, ref["color"], "color", 1e-5, 0.004)
    _compare(alpha, ref["alpha"], "alpha", 1e-5, 0.004)
    _compare(depth, ref["depth"], "depth", 2e-5, 0.004)
    _compare(coord, ref["coord"], "coord", 2e-5, 0.004)
    _compare(normal, ref["normal"], "normal", 2e-5, 0.006)
    _compare(mdepth, ref["mdepth"], "mdepth", 2e-5, 0.006)                # (the T > 0.5 switch)
    _compare(mcoord, ref["mcoord"], "mcoord", 2e-5, 0.006)


def test_precomputed_covariances_render_like_scales_and_rotations():
    """GaussianRasterizer.forward(cov3D_precomp=...) (forward.cu:363-371): Sigma = R diag(s^2) R^T of a scene with anisotropic scales and
    general rotations, handed over as six numbers per Gaussian, renders the same seven images; a gradient reaches the covariances"""
    from cut3r_slam_amd.gaussian_rasterizer import cov3d_to_scale_rotation
    H, W, P = 64, 80, 300
    means, scales, q, op, shs = _scene(P, P + 1)
    st = GO.camera_settings(H, W, 1.0, 1.0 * H / W, _w2c(0.15, -0.2, (0.1, -0.05, 0.3)), bg=(0.1, 0.2, 0.3), sh_degree=1)
    f = lambda t: t.float().to(DEV)
    qn = q / q.norm(dim=-1, keepdim=True)
    r, x, y, z = qn.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(P, 3, 3)
    Sig = R @ torch.diag_embed(scales ** 2) @ R.transpose(1, 2)
    cov6 = torch.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], -1)
    s_, q_ = cov3d_to_scale_rotation(f(cov6))
    assert float((s_.sort(-1).values.cpu().double() - scales.sort(-1).values).abs().max() / scales.max()) < 1e-4
    rast = GaussianRasterizer(_settings(st))
    kw = dict(means3D=f(means), means2D=torch.zeros(P, 3, device=DEV), opacities=f(op), shs=f(shs))
    a = rast(scales=f(scales), rotations=f(q), **kw)
    c6 = f(cov6).requires_grad_(True)
    b = rast(cov3D_precomp=c6, **kw)
    torch.cuda.synchronize()
    for i, name in ((0, "color"), (2, "coord"), (4, "depth"), (6, "alpha"), (7, "normal")):
        err = float((a[i] - b[i]).abs().max() / a[i].abs().max().clamp_min(1e-9))
        bad = float(((a[i] - b[i]).abs() > 2e-3 * a[i].abs().max()).float().mean())
        assert bad < 0.004, (name, err, bad)                 # (a radius that rounds differently moves a few tile-border pixels)
    b[0].sum().backward()
    assert c6.grad is not None and bool(torch.isfinite(c6.grad).all()) and float(c6.grad.abs().max()) > 0


def test_argument_checks_and_empty_scene():
    st = GO.camera_settings(32, 32, 1.0, 1.0, torch.eye(4, dtype=torch.float64), bg=(0.5, 0.25, 0.0))
    rast = GaussianRasterizer(_settings(st))
    z = torch.zeros(4, 3, device=DEV)
    with pytest.raises(Exception, match="SHs or precomputed"):
        rast(means3D=z, means2D=z, opacities=z[:, :1], scales=z, rotations=torch.zeros(4, 4, device=DEV))
    with pytest.raises(Exception, match="scale/rotation"):
        rast(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z)
    # every Gaussian behind the camera: background everywhere, zero alpha, radii 0
    means = torch.tensor([[0.0, 0.0, -2.0]] * 4, device=DEV)
    rot = torch.tensor([[1.0, 0, 0, 0]] * 4, device=DEV)
    color, radii, coord, mcoord, depth, mdepth, alpha, normal = rast(means3D=means, means2D=z, opacities=torch.ones(4, 1, device=DEV),
                                                                     colors_precomp=torch.ones(4, 3, device=DEV), scales=torch.full((4, 3), 0.1, device=DEV),
                                                                     rotations=rot)
    assert (radii == 0).all() and float(alpha.abs().max()) == 0.0 and float(depth.abs().max()) == 0.0
    np.testing.assert_allclose(color.cpu().numpy(), np.broadcast_to(np.array([0.5, 0.25, 0.0], np.float32)[:, None, None], (3, 32, 32)))
    vis = rast.markVisible(torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 0.1], [0.0, 0.0, -1.0]], device=DEV))
    assert vis.tolist() == [True, False, False]


def _grad_compare(got, ref, name, tol, max_bad):
    g, r = got.double().cpu().numpy(), ref.numpy()
    scale = np.abs(r).mean() + 1e-12
    err = np.abs(g - r)
    bad = (err > tol * (np.abs(r) + scale)).mean()
    print(f"[gs bwd] {name}: mean |ref| {scale:.3e}, median |err| {np.median(err):.2e}, max |err| / scale {err.max() / scale:.2e}, "
          f"beyond tolerance {100 * bad:.3f} %")
    assert bad <= max_bad, f"{name}: {bad}"
    assert np.abs(g.sum() - r.sum()) <= 5e-3 * np.abs(r).sum() + 1e-9, name


@pytest.mark.parametrize("H,W,P,deg,ks,cam,precomp", [(40, 56, 60, 3, 0.0, (0.0, 0.0, (0.0, 0.0, 0.0)), False),
                                                      (48, 64, 200, 1, 0.1, (0.15, -0.2, (0.1, -0.05, 0.3)), False),
                                                      (33, 47, 120, 0, 0.0, (-0.1, 0.1, (0.0, 0.1, 0.0)), True),
                                                      (33, 47, 900, 0, 0.0, (0.0, 0.0, (0.0, 0.0, 0.3)), True)])      # > 256 Gaussians per tile: several staging rounds
def test_backward_matches_autograd_of_the_restatement(H, W, P, deg, ks, cam, precomp):
    """every output image gets a random cotangent; the gradients of means, scales, rotations, opacities and SH / colours are compared
    with torch autograd through the fp64 restatement (hard per-pixel decisions flip for a few pixels in fp32: bounded fraction)"""
    means, scales, q, op, shs = _scene(P, 7 * P + deg)
    st = GO.camera_settings(H, W, 1.0, 1.0 * H / W, _w2c(*cam), bg=(0.1, 0.2, 0.3), sh_degree=deg, kernel_size=ks)
    colors = torch.rand(P, 3, dtype=torch.float64) if precomp else None
    g = torch.Generator().manual_seed(99)
    names = ("color", "coord", "mcoord", "depth", "mdepth", "alpha", "normal")
    chans = (3, 3, 3, 1, 1, 1, 3)
    cot = {n: torch.randn(c, H, W, generator=g, dtype=torch.float64) for n, c in zip(names, chans)}
    # ---- oracle
    leaves = [t.clone().requires_grad_(True) for t in (means, scales, q, op)] + [(colors if precomp else shs).clone().requires_grad_(True)]
    ref = GO.rasterize(leaves[0], leaves[3], leaves[1], leaves[2], st, shs=None if precomp else leaves[4], colors_precomp=leaves[4] if precomp else None)
    sum(float(1.0) * (ref[n] * cot[n]).sum() for n in names).backward()
    # ---- HIP
    f = lambda t: t.float().to(DEV)
    hm, hs, hq, ho = (f(t).requires_grad_(True) for t in (means, scales, q, op))
    hc = f(colors if precomp else shs).requires_grad_(True)
    m2d = torch.zeros(P, 3, device=DEV, requires_grad=True)
    rast = GaussianRasterizer(_settings(st))
    outs = rast(means3D=hm, means2D=m2d, opacities=ho, shs=None if precomp else hc, colors_precomp=hc if precomp else None, scales=hs, rotations=hq)
    color, radii, coord, mcoord, depth, mdepth, alpha, normal = outs
    loss = sum((o * f(cot[n])).sum() for n, o in zip(names, (color, coord, mcoord, depth, mdepth, alpha, normal)))
    loss.backward()
    torch.cuda.synchronize()
    _grad_compare(hm.grad, leaves[0].grad, "means3D", 2e-4, 0.01)
    _grad_compare(hs.grad, leaves[1].grad, "scales", 2e-4, 0.01)
    _grad_compare(hq.grad, leaves[2].grad, "rotations", 5e-4, 0.01)
    _grad_compare(ho.grad, leaves[3].grad, "opacities", 2e-4, 0.01)
    _grad_compare(hc.grad, leaves[4].grad, "colors" if precomp else "shs", 2e-4, 0.01)
    assert m2d.grad.shape == (P, 3) and float(m2d.grad[:, 2].min()) >= 0.0 and float(m2d.grad.abs().sum()) > 0
    assert float(hm.grad[:4].abs().max()) == 0.0                          # culled Gaussians get no gradient


def test_knn_mean_distance_matches_kdtree_and_alias_imports():
    """simple_knn.distCUDA2 (gaussian_model.py:191): mean squared distance to the 3 nearest other points, against scipy's cKDTree in
    fp64; duplicates count as neighbours at distance 0; the alias packages resolve to the same objects"""
    from scipy.spatial import cKDTree
    from cut3r_slam_amd.gaussian_rasterizer import distCUDA2
    g = torch.Generator().manual_seed(5)
    for P in (4, 257, 5000):
        pts = torch.randn(P, 3, generator=g) * torch.tensor([2.0, 1.0, 0.3])
        if P > 100:
            pts[10] = pts[11]                                              # an exact duplicate
        got = distCUDA2(pts.to(DEV)).cpu().numpy()
        d, _ = cKDTree(pts.double().numpy()).query(pts.double().numpy(), k=4)
        ref = (d[:, 1:] ** 2).mean(axis=1)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-7)
    with pytest.raises(Exception):
        distCUDA2(torch.zeros(3, 3, device=DEV))
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cut3r_slam_amd", "compat")
    sys.path.insert(0, compat)
    try:
        import diff_gaussian_rasterization as dgr
        from simple_knn._C import distCUDA2 as d2
        assert dgr.GaussianRasterizer is GaussianRasterizer and dgr.GaussianRasterizationSettings is GaussianRasterizationSettings
        assert d2 is distCUDA2
    finally:
        sys.path.remove(compat)


def test_grid_knn_is_exact_against_the_exhaustive_search_and_a_kdtree():
    """the uniform-grid 3-NN (cut3r_knn3_grid_mean_dist2: large maps, P >= 2e5 by default) returns the SAME three nearest squared
    distances as the exhaustive kernel -- on a pointmap-like surface, on heavy clustering with exact duplicates, on a degenerate plane
    (one empty grid axis), on a tiny cloud (G = 8) -- and agrees with scipy's cKDTree in fp64"""
    import time
    from scipy.spatial import cKDTree
    from cut3r_slam_amd import gaussian_rasterizer as GR
    g = torch.Generator().manual_seed(9)

    def both(pts):
        was = GR.KNN_GRID_MIN
        try:
            GR.KNN_GRID_MIN = 1 << 30
            ex = GR.distCUDA2(pts)
            GR.KNN_GRID_MIN = 4
            torch.cuda.synchronize()
            t0 = time.time()
            gr = GR.distCUDA2(pts)
            torch.cuda.synchronize()
            return ex, gr, time.time() - t0
        finally:
            GR.KNN_GRID_MIN = was

    # a stride-2 pointmap-like surface of 250 k points with noise
    u, v = torch.meshgrid(torch.linspace(-2, 2, 500), torch.linspace(-1.5, 1.5, 500), indexing="ij")
    surf = torch.stack([u, v, 3 + 0.3 * torch.sin(2 * u) * torch.cos(3 * v)], -1).reshape(-1, 3) + 1e-3 * torch.randn(250000, 3, generator=g)
    clus = torch.cat([torch.randn(3000, 3, generator=g) * 0.01 + c for c in torch.randn(70, 3, generator=g)], 0)
    clus[100:200] = clus[0:100]                                            # exact duplicates
    plane = torch.cat([torch.rand(50000, 2, generator=g), torch.zeros(50000, 1)], 1)
    tiny = torch.randn(9, 3, generator=g)
    # a few far outliers (sky / far-depth pixels that pass conf > 0) must not stretch the grid: 200 k surface points + 10 points
    # at 100 x the extent.  The grid is built on the robust box (mean +- 3 sigma per axis), outliers are clamped into border cells
    outl = torch.cat([surf[:200000], 300.0 * torch.randn(10, 3, generator=g)], 0)
    times = {}
    for name, pts in (("surface 250k", surf), ("clusters 210k + duplicates", clus), ("plane 50k", plane), ("tiny 9", tiny),
                      ("surface 200k + 10 outliers at 100 x the extent", outl)):
        ex, gr, dt = both(pts.to(DEV).contiguous())
        err = float((ex - gr).abs().max() / ex.abs().max().clamp_min(1e-30))
        print(f"[gs knn grid] {name}: max |grid - exhaustive| / max = {err:.2e}, grid search {1e3 * dt:.2f} ms")
        assert err < 1e-6, (name, err)
        times[name] = dt
    # with the bounding box of ALL points the surface fell into a handful of cells and the query degenerated to O(P^2) from global memory
    # (measured: 0.9 s; 0.33 s with the robust box alone -- the ten outlier QUERIES walked every shell of the grid; 22 ms with their list scan)
    assert times["surface 200k + 10 outliers at 100 x the extent"] < 0.1, times
    d, _ = cKDTree(surf.double().numpy()).query(surf[:2000].double().numpy(), k=4)
    ref = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(both(surf.to(DEV).contiguous())[1][:2000].cpu().numpy(), ref, rtol=2e-4, atol=1e-9)
    assert GR.KNN_GRID_MIN == 200_000


@pytest.mark.parametrize("C,H,W", [(3, 96, 128), (3, 37, 50), (1, 16, 16)])
def test_fused_ssim_matches_convolutions_forward_and_backward(C, H, W):
    """cut3r_ssim_forward/backward vs the conv2d formulation of loss_utils.py:140-170 (fp64): value and gradient w.r.t. the first image"""
    from cut3r_slam_amd.gaussian_rasterizer import fused_ssim
    from cut3r_slam_amd.gs_mapper import ssim_torch
    g = torch.Generator().manual_seed(C * H + W)
    a = torch.rand(C, H, W, generator=g)
    b = (a + 0.2 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    a64 = a.double().requires_grad_(True)
    ref = ssim_torch(a64, b.double())
    ref.backward()
    ag = a.to(DEV).requires_grad_(True)
    got = fused_ssim(ag, b.to(DEV))
    (3.0 * got).backward()
    assert abs(float(got.detach()) - float(ref.detach())) < 2e-6
    np.testing.assert_allclose(ag.grad.cpu().numpy() / 3.0, a64.grad.numpy(), atol=2e-6 * float(a64.grad.abs().max()) + 1e-10)


def test_scale_modifier_and_opacity_cap_paths():
    """scale_modifier (forward.cu:273-276) and the 0.99 alpha cap (forward.cu:545): a scene of large, nearly opaque Gaussians"""
    H, W, P = 40, 56, 70
    means, scales, q, op, shs = _scene(P, 77, smin=0.05, smax=0.3)
    op = torch.full_like(op, 0.999)
    st = GO.camera_settings(H, W, 1.0, 1.0 * H / W, _w2c(0.0, 0.0, (0.0, 0.0, 0.0)), bg=(0.0, 0.0, 0.0), sh_degree=0)
    st["scale_modifier"] = 1.7
    col = torch.rand(P, 3, dtype=torch.float64)
    lm, lo = means.clone().requires_grad_(True), op.clone().requires_grad_(True)
    ref = GO.rasterize(lm, lo, scales, q, st, colors_precomp=col)
    (ref["color"].sum() + ref["depth"].sum()).backward()
    f = lambda t: t.float().to(DEV)
    hm, ho = f(means).requires_grad_(True), f(op).requires_grad_(True)
    outs = GaussianRasterizer(_settings(st))(means3D=hm, means2D=torch.zeros(P, 3, device=DEV), opacities=ho, colors_precomp=f(col), scales=f(scales),
                                             rotations=f(q))
    (outs[0].sum() + outs[4].sum()).backward()
    assert float((ref["alpha"] > 0.98).double().mean()) > 0.3            # the cap is active on a good part of the image
    _compare(outs[0], ref["color"], "color (scale_modifier 1.7)", 1e-5, 0.004)
    _compare(outs[4], ref["depth"], "depth (scale_modifier 1.7)", 2e-5, 0.004)
    _grad_compare(hm.grad, lm.grad, "means3D (capped alphas)", 5e-4, 0.02)
    _grad_compare(ho.grad, lo.grad, "opacities (capped alphas)", 5e-4, 0.02)


def test_capacity_mode_equals_the_counted_pass_and_flags_overflow():
    """cut3r_gs_bin with a capacity instead of the read-back instance count (for callers that must not stop the host): same images and
    gradients bit for bit when the capacity suffices; the device flag is raised when it does not"""
    from cut3r_slam_amd import gaussian_rasterizer as GR
    H, W, P = 64, 80, 300
    means, scales, q, op, shs = _scene(P, 5)
    st = GO.camera_settings(H, W, 1.0, 1.0 * H / W, _w2c(0.1, -0.1, (0.0, 0.0, 0.2)), bg=(0.1, 0.2, 0.3), sh_degree=1)
    f = lambda t: t.float().to(DEV)
    rast = GaussianRasterizer(_settings(st))

    def run():
        leaves = [f(t).requires_grad_(True) for t in (means, op, shs, scales, q)]
        outs = rast(means3D=leaves[0], means2D=torch.zeros(P, 3, device=DEV), opacities=leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
        (outs[0].sum() + 2.0 * outs[4].sum() + outs[7].sum()).backward()
        return [o.detach().clone() for o in outs], [l.grad.clone() for l in leaves]
    GR.LAST_INSTANCES[0] = 0
    o0, g0 = run()
    n = GR.LAST_INSTANCES[0]
    assert n > 1000
    GR.overflow_flag(torch.device(DEV)).zero_()
    with GR.fixed_capacity(int(1.5 * n) + 7):
        o1, g1 = run()
    assert int(GR.overflow_flag(torch.device(DEV))) == 0
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    for a, b in zip(g0[2:], g1[2:]):                                     # (atomics: means / opacities may differ in the last bits)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    with GR.fixed_capacity(n // 2):
        run()
    assert int(GR.overflow_flag(torch.device(DEV))) == 1
    GR.overflow_flag(torch.device(DEV)).zero_()


@pytest.mark.parametrize("H,W", [(48, 64), (37, 50)])
def test_fused_pixel_losses_match_the_tensor_formulation(H, W):
    """cut3r_pixel_loss_forward/backward vs the tensor formulation of the mapper's colour-L1 / inverse-depth / depth-normal terms
    (gs_backend_per_frame.py:516-531) in fp64: value, and gradients w.r.t. the rendered image and depth"""
    from types import SimpleNamespace
    from cut3r_slam_amd.gaussian_rasterizer import pixel_losses
    from cut3r_slam_amd.gs_mapper import depth_to_normal
    g = torch.Generator().manual_seed(H)
    K = (60.0, 62.0, W / 2 - 0.3, H / 2 + 0.2)
    ys, xs = torch.meshgrid(torch.arange(H).double(), torch.arange(W).double(), indexing="ij")
    gt_depth = 2.0 + 0.3 * torch.sin(xs / 7) + 0.2 * torch.cos(ys / 5)
    gt_depth[3:6, 4:9] = 0.0                                               # invalid keyframe depth: masked out
    depth = (gt_depth + 0.05 * torch.randn(H, W, generator=g, dtype=torch.float64)).clamp_min(0.0)
    depth[10:12, 20:25] = 0.0                                              # no rendered depth there
    image, gt_image = torch.rand(3, H, W, generator=g, dtype=torch.float64), torch.rand(3, H, W, generator=g, dtype=torch.float64)
    cam = SimpleNamespace(fx=K[0], fy=K[1], cx=K[2], cy=K[3])
    gn = depth_to_normal(cam, gt_depth[None])
    w = (0.8, 10.0, 0.1)
    im64, d64 = image.clone().requires_grad_(True), depth[None].clone().requires_grad_(True)
    dmask = (gt_depth[None] > 0.001) & (d64 > 0.001)
    nd = dmask.sum().clamp_min(1)
    one = torch.ones_like(d64)
    ref = (w[0] * torch.abs(gt_image - im64).mean() + w[1] * (torch.abs(1 / torch.where(dmask, d64, one) - 1 / torch.where(dmask, gt_depth[None], one)) * dmask).sum() / nd
           + w[2] * ((1 - (depth_to_normal(SimpleNamespace(fx=K[0], fy=K[1], cx=K[2], cy=K[3]), d64) * gn).sum(0, keepdim=True)) * dmask).sum() / nd)
    ref.backward()
    f = lambda t: t.float().to(DEV)
    im, dd = f(image).requires_grad_(True), f(depth[None]).requires_grad_(True)
    got = pixel_losses(im, dd, f(gt_image), f(gt_depth), f(gn), K, *w)
    (2.0 * got).backward()
    assert abs(float(got.detach()) - float(ref.detach())) < 2e-5 * abs(float(ref.detach()))
    np.testing.assert_allclose(im.grad.cpu().numpy() / 2.0, im64.grad.numpy(), atol=1e-9)
    gd, rd = dd.grad.cpu().numpy()[0] / 2.0, d64.grad.numpy()[0]
    err = np.abs(gd - rd)
    print(f"[gs] pixel losses: value {float(got.detach()):.6f} vs {float(ref.detach()):.6f}, depth-gradient max |err| {err.max():.2e} (scale {np.abs(rd).max():.2e})")
    assert err.max() < 2e-4 * np.abs(rd).max()


def test_fused_refine_losses_match_the_tensor_formulation():
    """cut3r_refine_loss_forward/backward vs the tensor formulation of the pose-refinement terms (gs_backend_per_frame.py:240-262), fp64"""
    from cut3r_slam_amd.gaussian_rasterizer import refine_losses
    H, W = 41, 53
    g = torch.Generator().manual_seed(9)
    image, gt_image = torch.rand(3, H, W, generator=g, dtype=torch.float64), torch.rand(3, H, W, generator=g, dtype=torch.float64)
    gt_depth = 1.5 + torch.rand(H, W, generator=g, dtype=torch.float64)
    gt_depth[2:5, 3:9] = 0.0
    depth = (gt_depth * (1 + 0.1 * torch.randn(H, W, generator=g, dtype=torch.float64))).clamp_min(0.0)[None]
    depth[0, 20:23, 10:14] = 0.0
    alpha = torch.rand(1, H, W, generator=g, dtype=torch.float64)
    th = 0.4
    im64, d64 = image.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    amask = alpha > th
    ratio = amask.sum() / amask.numel()
    dmask = (gt_depth[None] > 0.001) & (d64 > 0.001) & amask
    rgb = torch.abs((gt_image - im64)[:, amask[0]]).mean()
    diff = torch.log(d64[dmask]) - torch.log(gt_depth[None][dmask])
    var = (diff ** 2).mean() - diff.mean() ** 2
    (5 * ratio * rgb + ratio * var).backward()
    f = lambda t: t.float().to(DEV)
    im, dd = f(image).requires_grad_(True), f(depth).requires_grad_(True)
    r_rgb, r_var, r = refine_losses(im, dd, f(gt_image), f(gt_depth), f(alpha), th)
    (5 * r_rgb + r_var).backward()
    assert (abs(float(r.detach()) - float(ratio)) < 1e-6 and abs(float(r_rgb.detach()) - float((ratio * rgb).detach())) < 1e-6
            and abs(float(r_var.detach()) - float((ratio * var).detach())) < 1e-6)
    np.testing.assert_allclose(im.grad.cpu().numpy(), im64.grad.numpy(), atol=1e-9)
    np.testing.assert_allclose(dd.grad.cpu().numpy(), d64.grad.numpy(), atol=2e-9, rtol=2e-4)
