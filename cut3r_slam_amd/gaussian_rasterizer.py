"""`diff_gaussian_rasterization` on gfx950: the interface of thirdparty/diff-gaussian-rasterization/diff_gaussian_rasterization/
__init__.py (GaussianRasterizationSettings :190-205, GaussianRasterizer :207-249, _RasterizeGaussians :42-188) over the HIP
rasteriser in csrc/gs.hip.  Call site: hislam2/gaussian/renderer/__init__.py:101-140 (`render`).

Same argument meaning, same outputs `(color, radii, coord, mcoord, depth, mdepth, alpha, normal)`, same exceptions for the
either/or argument pairs.  Forward and backward.  The backward pass returns the exact derivatives of the forward function (forward-mode duals through
the per-Gaussian projection); `means2D` receives the screen-space gradient in the reference's units (x, y in NDC, z = |.| sum).
`cov3D_precomp` is accepted (round 4): the packed covariance is factored into R and S on the host side of the call (symmetric
eigendecomposition, differentiable), because the ray-space plane of this rasteriser is written from R and S directly.  Not built, by
decision: `integrate` (mesh extraction; no file of the SLAM path calls it) raises NotImplementedError.
There is no CPU fallback: without the HIP library every call raises."""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import check

GS_REC = 32
_ZERO_IMAGES = {}
_CAPACITY = [0]                   # > 0: capacity mode of cut3r_gs_bin (see fixed_capacity)
LAST_INSTANCES = [0]              # largest instance count read back since it was last reset (sizes a capacity)
_OVERFLOW = {}


def overflow_flag(device):
    """device int32 [1]: set to 1 by a capacity-mode pass whose scene needed more instances than the capacity"""
    t = _OVERFLOW.get(device)
    if t is None:
        t = _OVERFLOW[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


_WS_BYTES = {}


def workspace_bytes(P, n_instances):
    """scan / sort scratch size for P Gaussians and n instances (cached: the size query itself must not run inside a stream capture)"""
    key = (int(P), int(n_instances))
    v = _WS_BYTES.get(key)
    if v is None:
        if len(_WS_BYTES) > 4096:
            _WS_BYTES.clear()
        v = _WS_BYTES[key] = int(_lib.load().cut3r_gs_workspace_bytes(key[0], key[1]))
    return v


class fixed_capacity:
    """with fixed_capacity(n): every rasteriser pass inside sizes its binning buffers for n instances WITHOUT reading the count back
    (no host stop: the pass can be captured in a graph).  The caller checks `overflow_flag(device)` afterwards."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev = _CAPACITY[0]
        _CAPACITY[0] = self.n
        return self

    def __exit__(self, *a):
        _CAPACITY[0] = self.prev
        return False

KNN_GRID_MIN = 200_000           # distCUDA2: from this many points on, the grid search (exact) replaces the exhaustive one
MAX_INSTANCES = 1 << 27          # sort buffers are sized from the data: refuse sizes that only a diverged map produces (3.2 GB at the limit)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    require_depth: bool
    require_coord: bool
    debug: bool


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _host16(t, n):
    a = (C.c_float * n)(*[float(v) for v in t.detach().reshape(-1).cpu().tolist()[:n]])
    return a


class _Buffers:
    """what rasterize_gaussians keeps between forward and backward (geomBuffer / binningBuffer / imgBuffer of the reference)"""
    __slots__ = ("geom", "point_list", "ranges", "n_contrib", "aux", "n_inst", "alpha", "coord", "depth", "normal")


def _forward(means3D, sh, colors_precomp, opacities, scales, rotations, st):
    lib = _lib.load()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("GaussianRasterizer: tensors must live on the GPU")
    P = means3D.shape[0]
    H, W = int(st.image_height), int(st.image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    means3D, scales, rotations, opacities = (t.detach().contiguous().float() for t in (means3D, scales, rotations, opacities))
    use_sh = colors_precomp is None or colors_precomp.numel() == 0
    sh = sh.detach().contiguous().float() if use_sh else None
    colors = None if use_sh else colors_precomp.detach().contiguous().float()
    K = sh.shape[1] if use_sh else 0
    # the kernels write every pixel / every Gaussian: no fills (an empty scene is the one case that needs zeros)
    alloc = torch.zeros if P == 0 else torch.empty
    out = {k: alloc(c, H, W, **f32) for k, c in (("color", 3), ("coord", 3), ("mcoord", 3), ("depth", 1), ("mdepth", 1), ("alpha", 1), ("normal", 3))}
    radii = alloc(P, dtype=torch.int32, device=dev)
    buf = _Buffers()
    buf.n_contrib = alloc(2, H, W, dtype=torch.int32, device=dev)
    buf.aux = alloc(2, H, W, **f32)
    bg = _host16(st.bg, 3)
    if P == 0:
        out["color"] += torch.as_tensor(list(bg), **f32)[:, None, None]
        buf.geom, buf.point_list, buf.ranges, buf.n_inst = None, None, None, 0
        return out, radii, buf
    buf.geom = torch.empty(P, GS_REC, **f32)
    tiles = torch.empty(P, dtype=torch.int32, device=dev)
    offsets = torch.empty(P, dtype=torch.int32, device=dev)
    ws = torch.empty(workspace_bytes(P, 0), dtype=torch.uint8, device=dev)
    view, proj, campos = _host16(st.viewmatrix, 16), _host16(st.projmatrix, 16), _host16(st.campos, 3)
    check(lib.cut3r_gs_preprocess(P, _p(means3D), _p(scales), _p(rotations), _p(opacities), _p(sh), int(st.sh_degree), K, _p(colors), view, proj,
                                  campos, W, H, float(st.tanfovx), float(st.tanfovy), float(st.kernel_size), float(st.scale_modifier), _p(buf.geom),
                                  _p(radii), _p(tiles), _p(offsets), _p(ws), ws.numel(), _s()), "gs_preprocess")
    overflow = None
    if _CAPACITY[0]:
        # capacity mode (fixed_capacity()): the instance count is NOT read back -- buffers of the given size, padded, overflow flagged
        n_inst = int(_CAPACITY[0])
        overflow = overflow_flag(dev)
    else:
        n_inst = int(offsets[-1].item()) & 0xffffffff                 # the one host read of the pass (rasterizer_impl.cu:346-354)
        LAST_INSTANCES[0] = max(LAST_INSTANCES[0], n_inst)
    if n_inst > MAX_INSTANCES:
        raise RuntimeError(f"GaussianRasterizer: {n_inst} Gaussian/tile instances (limit {MAX_INSTANCES}, 24 bytes each): degenerate scales or a "
                           "diverged map; raise gaussian_rasterizer.MAX_INSTANCES if this is intended")
    gx, gy = (W + 15) // 16, (H + 15) // 16
    buf.ranges = torch.empty(gy * gx, 2, dtype=torch.int32, device=dev)
    buf.n_inst = n_inst
    keys_tmp = torch.empty(max(1, n_inst), dtype=torch.int64, device=dev)
    keys_sorted = torch.empty_like(keys_tmp)
    vals_tmp = torch.empty(max(1, n_inst), dtype=torch.int32, device=dev)
    buf.point_list = torch.empty_like(vals_tmp)
    ws2 = torch.empty(workspace_bytes(P, n_inst), dtype=torch.uint8, device=dev)
    check(lib.cut3r_gs_bin(P, _p(buf.geom), _p(offsets), n_inst, W, H, _p(keys_tmp), _p(vals_tmp), _p(keys_sorted), _p(buf.point_list),
                           _p(buf.ranges), _p(ws2), ws2.numel(), _p(overflow), _s()), "gs_bin")
    check(lib.cut3r_gs_render_forward(_p(buf.ranges), _p(buf.point_list), _p(buf.geom), W, H, float(st.tanfovx), float(st.tanfovy), bg,
                                      _p(out["color"]), _p(out["coord"]), _p(out["mcoord"]), _p(out["depth"]), _p(out["mdepth"]), _p(out["alpha"]),
                                      _p(out["normal"]), _p(buf.n_contrib), _p(buf.aux), _s()), "gs_render_forward")
    # detached aliases: the returned images become outputs of the autograd node that owns `buf`; holding them directly would tie the node
    # to itself (only the cyclic collector would free the graph -- and keep its gradient accumulators alive until then)
    buf.alpha, buf.coord, buf.depth, buf.normal = (out[k].detach() for k in ("alpha", "coord", "depth", "normal"))
    return out, radii, buf


def _backward(buf, st, means3D, sh, colors_precomp, opacities, scales, rotations, grads):
    """grads: (color, coord, mcoord, depth, mdepth, alpha, normal) images or None -> the gradients of the Gaussian parameters"""
    lib = _lib.load()
    dev = means3D.device
    P = means3D.shape[0]
    H, W = int(st.image_height), int(st.image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    use_sh = colors_precomp is None or colors_precomp.numel() == 0
    flat = torch.zeros(P * 14, **f32)                                  # one fill: culled Gaussians are not written by the kernel
    d = {"means": flat[0:3 * P].view(P, 3), "scales": flat[3 * P:6 * P].view(P, 3), "rots": flat[6 * P:10 * P].view(P, 4),
         "opac": flat[10 * P:11 * P].view(P, 1), "means2D": flat[11 * P:14 * P].view(P, 3)}
    d["shs"] = torch.zeros(sh.shape, **f32) if use_sh else None
    d["colors"] = None if use_sh else torch.zeros(P, 3, **f32)
    if P == 0 or buf.geom is None:
        return d
    chans = (3, 3, 3, 1, 1, 1, 3)
    zero = _ZERO_IMAGES.get((H, W, dev))
    if zero is None:
        zero = _ZERO_IMAGES[(H, W, dev)] = torch.zeros(3, H, W, **f32)  # read-only stand-in for the images the loss did not use
    g = [zero[:c] if t is None else t.detach().contiguous().float() for t, c in zip(grads, chans)]
    means3D, scales, rotations, opacities = (t.detach().contiguous().float() for t in (means3D, scales, rotations, opacities))
    sh_c = sh.detach().contiguous().float() if use_sh else None
    K = sh_c.shape[1] if use_sh else 0
    dgeom = torch.empty(P, GS_REC, **f32)
    bg = _host16(st.bg, 3)
    check(lib.cut3r_gs_render_backward(_p(buf.ranges), _p(buf.point_list), _p(buf.geom), P, W, H, float(st.tanfovx), float(st.tanfovy), bg,
                                       _p(buf.n_contrib), _p(buf.aux), _p(buf.alpha), _p(buf.coord), _p(buf.depth), _p(buf.normal), _p(g[0]),
                                       _p(g[1]), _p(g[2]), _p(g[3]), _p(g[4]), _p(g[5]), _p(g[6]), _p(dgeom), _s()), "gs_render_backward")
    view, proj, campos = _host16(st.viewmatrix, 16), _host16(st.projmatrix, 16), _host16(st.campos, 3)
    check(lib.cut3r_gs_preprocess_backward(P, _p(means3D), _p(scales), _p(rotations), _p(opacities), _p(sh_c), int(st.sh_degree), K, view, proj,
                                           campos, W, H, float(st.tanfovx), float(st.tanfovy), float(st.kernel_size), float(st.scale_modifier),
                                           _p(buf.geom), _p(dgeom), _p(d["means"]), _p(d["scales"]), _p(d["rots"]), _p(d["opac"]), _p(d["shs"]),
                                           _p(d["colors"]), _p(d["means2D"]), _s()), "gs_preprocess_backward")
    return d


class _RasterizeGaussians(torch.autograd.Function):
    """diff_gaussian_rasterization/__init__.py:42-188"""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        if cov3Ds_precomp is not None and cov3Ds_precomp.numel() > 0:
            raise NotImplementedError("rasterize_gaussians: pass scales and rotations (GaussianRasterizer.forward converts cov3D_precomp)")
        out, radii, buf = _forward(means3D, sh, colors_precomp, opacities, scales, rotations, raster_settings)
        ctx.raster_settings = raster_settings
        ctx.buf = buf
        ctx.save_for_backward(means3D, sh, colors_precomp, opacities, scales, rotations)
        ctx.mark_non_differentiable(radii)
        return out["color"], radii, out["coord"], out["mcoord"], out["depth"], out["mdepth"], out["alpha"], out["normal"]

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_coord, grad_mcoord, grad_depth, grad_mdepth, grad_alpha, grad_normal):
        means3D, sh, colors_precomp, opacities, scales, rotations = ctx.saved_tensors
        d = _backward(ctx.buf, ctx.raster_settings, means3D, sh, colors_precomp, opacities, scales, rotations,
                      (grad_color, grad_coord, grad_mcoord, grad_depth, grad_mdepth, grad_alpha, grad_normal))
        # (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)
        return d["means"], d["means2D"], d["shs"], d["colors"], d["opac"].reshape(opacities.shape), d["scales"], d["rots"], None, None


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """auxiliary.h:155-180 / rasterizer_impl.cu:54-66: in front of the near plane (view z > 0.2)"""
        with torch.no_grad():
            V = self.raster_settings.viewmatrix.to(positions.device, torch.float32)
            z = positions.float() @ V[:3, 2] + V[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if cov3D_precomp is not None:
            # precomputed 3-D covariances (forward.cu:363-371 takes them instead of computeCov3D; unused by the live mapper,
            # gaussian/renderer/__init__.py:128-143): the kernels work from the factors R, S (the ray-space plane and the inverse covariance
            # are written from them, no eigen-solver on the device), so Sigma = R diag(s^2) R^T is factored here -- exact forward; gradients
            # reach cov3D_precomp through torch.linalg.eigh (undefined where two eigenvalues coincide).  computeCov3D's scale_modifier does
            # not apply to precomputed covariances (forward.cu:363-366): the call runs with modifier 1.
            scales, rotations = cov3d_to_scale_rotation(cov3D_precomp)
            st = self.raster_settings
            settings = st._replace(scale_modifier=1.0) if hasattr(st, "_replace") else st
            empty = torch.Tensor([])
            return rasterize_gaussians(means3D, means2D, shs if shs is not None else empty, colors_precomp if colors_precomp is not None else empty,
                                       opacities, scales, rotations, empty, settings)
        empty = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs if shs is not None else empty, colors_precomp if colors_precomp is not None else empty,
                                   opacities, scales if scales is not None else empty, rotations if rotations is not None else empty,
                                   cov3D_precomp if cov3D_precomp is not None else empty, self.raster_settings)

    def integrate(self, *args, **kwargs):
        """RaDe-GS's `integrate` evaluates the Gaussians' opacity field at arbitrary 3-D points for mesh extraction (marching tetrahedra,
        diff_gaussian_rasterization/__init__.py:251-298).  DECISION (round 4): not built -- no file of the reference's SLAM path calls it
        (hislam2/gaussian/renderer/__init__.py:128-143 renders through forward() only; mesh extraction lives in the upstream RaDe-GS
        scripts, outside this repository's scope table)."""
        raise NotImplementedError("GaussianRasterizer.integrate (mesh extraction) is not on the SLAM path and is not built")


def cov3d_to_scale_rotation(cov6):
    """[P,6] upper-triangular 3-D covariances (xx, xy, xz, yy, yz, zz: computeCov3D's layout, forward.cu:113-150) -> (scales [P,3],
    rotations [P,4] real-first unit quaternions) with Sigma = R diag(s^2) R^T.  Differentiable (torch.linalg.eigh)."""
    c = cov6.float()
    S = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], -1).reshape(-1, 3, 3)
    w, V = torch.linalg.eigh(S)
    V = V * torch.where(torch.linalg.det(V) < 0, -1.0, 1.0)[:, None, None]          # a proper rotation (flipping all three axes keeps R diag R^T)
    scales = w.clamp_min(0).sqrt()
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [V[:, i, j] for i in range(3) for j in range(3)]
    # quaternion (r, x, y, z) from a rotation matrix: the candidate with the largest denominator per row
    q_abs = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1).clamp_min(0).sqrt()
    cand = torch.stack([torch.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
                        torch.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], -1),
                        torch.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], -1),
                        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(-1)
    q = cand[torch.arange(cand.shape[0], device=cand.device), best]
    return scales, q / q.norm(dim=-1, keepdim=True)


def distCUDA2(points):
    """simple_knn._C.distCUDA2 (hislam2/gaussian/scene/gaussian_model.py:18,191,313): points [P,3] float32 on the GPU -> [P] mean squared
    distance to the 3 nearest other points (exhaustive search on the GPU)."""
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2: points must live on the GPU")
    pts = points.detach().contiguous().float()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("distCUDA2: points must be [P,3]")
    lib = _lib.load()
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P >= KNN_GRID_MIN:          # large maps: exact search through a uniform grid instead of P^2 pairs
        nb = int(lib.cut3r_knn3_grid_workspace_bytes(P))
        ws = torch.empty(nb, dtype=torch.uint8, device=pts.device)
        check(lib.cut3r_knn3_grid_mean_dist2(_p(pts), P, _p(out), _p(ws), nb, _s()), "knn3_grid_mean_dist2")
        return out
    ws = torch.empty(max(1, lib.cut3r_knn3_chunks(P)) * P * 3, dtype=torch.float32, device=pts.device)
    check(lib.cut3r_knn3_mean_dist2(_p(pts), P, _p(out), _p(ws), _s()), "knn3_mean_dist2")
    return out


class _FusedSSIM(torch.autograd.Function):
    """mean SSIM of a rendered image against a fixed image on the fused HIP kernels (`cut3r_ssim_forward/backward`)"""

    @staticmethod
    def forward(ctx, img, target):
        a, b = img.detach().contiguous().float(), target.detach().contiguous().float()
        Cn, H, W = a.shape
        smap, d1, d2, d3 = (torch.empty_like(a) for _ in range(4))
        check(_lib.load().cut3r_ssim_forward(_p(a), _p(b), Cn, H, W, _p(smap), _p(d1), _p(d2), _p(d3), _s()), "ssim_forward")
        ctx.save_for_backward(a, b, d1, d2, d3)
        return smap.mean()

    @staticmethod
    def backward(ctx, g):
        a, b, d1, d2, d3 = ctx.saved_tensors
        Cn, H, W = a.shape
        scale = (g.detach().float() / a.numel()).reshape(1).contiguous()
        grad = torch.empty_like(a)
        check(_lib.load().cut3r_ssim_backward(_p(a), _p(b), _p(d1), _p(d2), _p(d3), Cn, H, W, _p(scale), _p(grad), _s()), "ssim_backward")
        return grad, None


def fused_ssim(img, target):
    """loss_utils.py:129-170 `ssim(img1, img2)` with size_average=True; img [C,H,W] (gradients flow to it), target [C,H,W] constant"""
    if img.device.type != "cuda":
        raise RuntimeError("fused_ssim: tensors must live on the GPU")
    return _FusedSSIM.apply(img, target)


class _PixelLosses(torch.autograd.Function):
    """colour L1 + inverse-depth L1 + depth-normal agreement in one kernel each way (`cut3r_pixel_loss_forward/backward`)"""

    @staticmethod
    def forward(ctx, image, depth, gt_image, gt_depth, gt_normal, K, w_rgb, w_depth, w_normal):
        img, d = image.detach().contiguous().float(), depth.detach().contiguous().float()
        H, W = img.shape[-2:]
        sums = torch.empty(4, dtype=torch.float32, device=img.device)
        check(_lib.load().cut3r_pixel_loss_forward(_p(img), _p(gt_image), _p(d), _p(gt_depth), _p(gt_normal), H, W, K[0], K[1], K[2], K[3], _p(sums),
                                                   _s()), "pixel_loss_forward")
        nd = sums[3].clamp_min(1.0)
        ctx.save_for_backward(img, d, gt_image, gt_depth, gt_normal, nd)
        ctx.K, ctx.w = K, (w_rgb, w_depth, w_normal)
        return w_rgb * sums[0] / (3 * H * W) + (w_depth * sums[1] + w_normal * sums[2]) / nd

    @staticmethod
    def backward(ctx, g):
        img, d, gt_image, gt_depth, gt_normal, nd = ctx.saved_tensors
        H, W = img.shape[-2:]
        w_rgb, w_depth, w_normal = ctx.w
        coef = torch.stack([g * (w_rgb / (3 * H * W)), g * w_depth / nd, g * w_normal / nd]).float().contiguous()
        g_img, g_d = torch.empty_like(img), torch.empty_like(d)
        K = ctx.K
        check(_lib.load().cut3r_pixel_loss_backward(_p(img), _p(gt_image), _p(d), _p(gt_depth), _p(gt_normal), H, W, K[0], K[1], K[2], K[3], _p(coef),
                                                    _p(g_img), _p(g_d), _s()), "pixel_loss_backward")
        return g_img, g_d, None, None, None, None, None, None, None


def pixel_losses(image, depth, gt_image, gt_depth, gt_normal, K, w_rgb, w_depth, w_normal):
    """w_rgb * mean|gt - image| + w_depth * mean_mask |1/depth - 1/gt_depth| + w_normal * mean_mask (1 - n(depth) . gt_normal), mask =
    (gt_depth > 0.001) & (depth > 0.001)  (hislam2/gs_backend_per_frame.py:516-531).  image [3,H,W] and depth [1,H,W] or [H,W] receive
    gradients; gt_image [3,H,W], gt_depth [H,W], gt_normal [3,H,W] are constants (contiguous float32); K = (fx, fy, cx, cy)."""
    if image.device.type != "cuda":
        raise RuntimeError("pixel_losses: tensors must live on the GPU")
    return _PixelLosses.apply(image, depth, gt_image.contiguous(), gt_depth.contiguous(), gt_normal.contiguous(), tuple(float(v) for v in K),
                              float(w_rgb), float(w_depth), float(w_normal))


class _RefineLosses(torch.autograd.Function):
    """(ratio * rgb, ratio * log-depth variance, ratio) of the pose refinement on `cut3r_refine_loss_forward/backward`"""

    @staticmethod
    def forward(ctx, image, depth, gt_image, gt_depth, alpha, alpha_th):
        img, d, a = image.detach().contiguous().float(), depth.detach().contiguous().float(), alpha.detach().contiguous().float()
        H, W = img.shape[-2:]
        sums = torch.empty(5, dtype=torch.float32, device=img.device)
        check(_lib.load().cut3r_refine_loss_forward(_p(img), _p(gt_image), _p(d), _p(gt_depth), _p(a), float(alpha_th), H, W, _p(sums), _s()),
              "refine_loss_forward")
        na, nm = sums[1].clamp_min(1.0), sums[4].clamp_min(1.0)
        ratio = sums[1] / (H * W)
        mean = sums[2] / nm
        ctx.save_for_backward(img, d, gt_image, gt_depth, a, ratio, na, nm, mean)
        ctx.alpha_th = float(alpha_th)
        return ratio * sums[0] / (3 * na), ratio * (sums[3] / nm - mean * mean), ratio

    @staticmethod
    def backward(ctx, g_rgb, g_var, g_ratio):
        img, d, gt_image, gt_depth, a, ratio, na, nm, mean = ctx.saved_tensors
        H, W = img.shape[-2:]
        coef = torch.stack([g_rgb * ratio / (3 * na), g_var * ratio / nm, mean]).float().contiguous()
        g_img, g_d = torch.empty_like(img), torch.empty_like(d)
        check(_lib.load().cut3r_refine_loss_backward(_p(img), _p(gt_image), _p(d), _p(gt_depth), _p(a), ctx.alpha_th, H, W, _p(coef), _p(g_img), _p(g_d),
                                                     _s()), "refine_loss_backward")
        return g_img, g_d, None, None, None, None


def refine_losses(image, depth, gt_image, gt_depth, alpha, alpha_th):
    """hislam2/gs_backend_per_frame.py:240-262: with a = alpha > alpha_th (treated as constant) and ratio = |a| / HW returns
    (ratio * mean_a |gt - image|, ratio * var_m (log depth - log gt_depth), ratio), m = a & depth > 0.001 & gt_depth > 0.001."""
    if image.device.type != "cuda":
        raise RuntimeError("refine_losses: tensors must live on the GPU")
    return _RefineLosses.apply(image, depth, gt_image.contiguous(), gt_depth.contiguous(), alpha, float(alpha_th))
