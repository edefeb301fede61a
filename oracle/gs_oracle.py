"""TEST INFRASTRUCTURE ONLY -- float64 torch-CPU restatement of the Gaussian rasteriser the GS mapper calls
(/root/reference/hislam2/gaussian/renderer/__init__.py:89-152 -> thirdparty/diff-gaussian-rasterization, RaDe-GS flavour):
  preprocess      cuda_rasterizer/forward.cu:308-427 (cull, projection, 3D covariance :270-305, EWA 2D covariance + ray-space
                  plane / normal :77-265, conic, radius, tile rectangle auxiliary.h:62-72, SH colour :23-74)
  binning         cuda_rasterizer/rasterizer_impl.cu:70-112 (one instance per Gaussian and covered 16x16 tile, ordered by view depth)
  render          cuda_rasterizer/forward.cu:429-692 (front-to-back alpha compositing of colour, ray distance, camera-space
                  coordinate, normal; median values at T > 0.5)
Differentiable (torch autograd), so the same function is the oracle of the backward pass: the gradients it gives are the
derivatives of THIS forward function.  PARITY UNPINNED vs the reference: its rasteriser is CUDA (cannot be built or run here) and
the tree holds no fixture of its outputs; the restatement follows the source line by line as cited.
Matrix conventions as the reference passes them: viewmatrix / projmatrix are the TRANSPOSED 4x4s (row vector times matrix:
p_view = [p, 1] @ viewmatrix), quaternions are (r, x, y, z) and are NOT normalised by the rasteriser."""
import math

import torch

DT = torch.float64
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]
BLOCK = 16


def quat_to_rot(q):
    """forward.cu:281-291 read as a standard matrix: R_std of the (r, x, y, z) quaternion, no normalisation"""
    r, x, y, z = q.unbind(-1)
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def sh_colour(deg, means, campos, shs):
    """forward.cu:23-74; shs [P,K,3]"""
    d = means - campos
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
               + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10] + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
               + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13]
               + SH_C3[5] * z * (xx - yy) * shs[:, 14] + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp(res + 0.5, min=0.0)


def preprocess(means3D, opacities, scales, rotations, shs, colors_precomp, st):
    """Per-Gaussian screen-space quantities.  Returns a dict of [P,...] tensors; `visible` [P] bool (radius > 0)."""
    P = means3D.shape[0]
    H, W = st["image_height"], st["image_width"]
    V = st["viewmatrix"].to(DT)
    PM = st["projmatrix"].to(DT)
    fx, fy = W / (2.0 * st["tanfovx"]), H / (2.0 * st["tanfovy"])
    ks = float(st["kernel_size"])
    ones = torch.ones(P, 1, dtype=DT)
    p_view = torch.cat([means3D, ones], 1) @ V[:, :3]                          # auxiliary.h transformPoint4x3
    p_hom = torch.cat([means3D, ones], 1) @ PM
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :3] * p_w[:, None]
    in_front = p_view[:, 2] > 0.2                                               # auxiliary.h:170
    # ---- 3D covariance (forward.cu:270-305)
    R = quat_to_rot(rotations)
    S2 = (st["scale_modifier"] * scales) ** 2
    Sigma = R @ torch.diag_embed(S2) @ R.transpose(1, 2)
    # ---- computeCov2D (forward.cu:77-265)
    tz = torch.where(in_front, p_view[:, 2], torch.ones_like(p_view[:, 2]))     # culled rows: keep the arithmetic finite
    limx, limy = 1.3 * st["tanfovx"], 1.3 * st["tanfovy"]
    txtz = torch.clamp(p_view[:, 0] / tz, -limx, limx)
    tytz = torch.clamp(p_view[:, 1] / tz, -limy, limy)
    tx, ty = txtz * tz, tytz * tz
    zero = torch.zeros_like(tz)
    Jstd = torch.stack([torch.stack([fx / tz, zero, -fx * tx / (tz * tz)], -1),
                        torch.stack([zero, fy / tz, -fy * ty / (tz * tz)], -1)], -2)         # [P,2,3]
    Rw2c = V[:3, :3].T
    A = Jstd @ Rw2c
    cov = A @ Sigma @ A.transpose(1, 2)                                          # [P,2,2]
    c00, c01, c11 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    a, b, c = c00 + ks, c01, c11 + ks
    det0r, det1r = c00 * c11 - c01 * c01, a * c - c01 * c01
    det_0, det_1 = torch.clamp(det0r, min=1e-6), torch.clamp(det1r, min=1e-6)
    coef = torch.sqrt(det_0 / (det_1 + 1e-6) + 1e-6)
    coef = torch.where((det_0 <= 1e-6) | (det_1 <= 1e-6), torch.zeros_like(coef), coef)
    # inverse covariance in the camera frame; the eigen-decomposition of forward.cu:130-150 is that of R S^2 R^T: eigenvalues S2,
    # eigenvectors the columns of R
    lam_min, k_min = S2.min(dim=1)
    well = lam_min > 1e-8
    Sig_inv = R @ torch.diag_embed(1.0 / torch.where(well[:, None], S2, torch.ones_like(S2))) @ R.transpose(1, 2)
    e_min = torch.gather(R, 2, k_min[:, None, None].expand(P, 3, 1))[:, :, 0]
    Sig_inv = torch.where(well[:, None, None], Sig_inv, e_min[:, :, None] * e_min[:, None, :])
    cam_inv = Rw2c @ Sig_inv @ Rw2c.T
    uvh = torch.stack([txtz, tytz, torch.ones_like(txtz)], -1)
    uvh_m = (cam_inv @ uvh[:, :, None])[:, :, 0]
    uvh_mn = uvh_m / uvh_m.norm(dim=-1, keepdim=True)
    u2, v2, uv = txtz * txtz, tytz * tytz, txtz * tytz
    l = torch.sqrt(tx * tx + ty * ty + tz * tz)
    vbn = (uvh_mn * uvh).sum(-1)
    aa = uvh_mn / torch.clamp(vbn, min=1e-7)[:, None]
    plane0 = (v2 + 1) * aa[:, 0] - uv * aa[:, 1] - txtz * aa[:, 2]
    plane1 = -uv * aa[:, 0] + (u2 + 1) * aa[:, 1] - tytz * aa[:, 2]
    nl = u2 + v2 + 1
    camera_plane = torch.stack([(-(v2 + 1) * tz + plane0 * tx) / nl / fx, (uv * tz + plane1 * tx) / nl / fy,
                                (uv * tz + plane0 * ty) / nl / fx, (-(u2 + 1) * tz + plane1 * ty) / nl / fy,
                                (tx + plane0 * tz) / nl / fx, (ty + plane1 * tz) / nl / fy], -1)
    ray_plane = torch.stack([plane0 * l / nl / fx, plane1 * l / nl / fy], -1)
    fn = l / nl
    rn = torch.stack([-plane0 * fn, -plane1 * fn, -torch.ones_like(fn)], -1)
    cam_n = torch.stack([rn[:, 0] / tz + rn[:, 2] * tx / l, rn[:, 1] / tz + rn[:, 2] * ty / l,
                         -rn[:, 0] * tx / (tz * tz) - rn[:, 1] * ty / (tz * tz) + rn[:, 2] * tz / l], -1)
    normal = cam_n / cam_n.norm(dim=-1, keepdim=True)
    bad = torch.isnan(uvh_mn[:, 0])
    camera_plane = torch.where(bad[:, None], torch.zeros_like(camera_plane), camera_plane)
    ray_plane = torch.where(bad[:, None], torch.zeros_like(ray_plane), ray_plane)
    normal = torch.where(bad[:, None], torch.zeros_like(normal), normal)
    # ---- rest of preprocessCUDA (forward.cu:377-421)
    ts = p_view.norm(dim=-1)
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], -1)
    mid = 0.5 * (a + c)
    root = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + root, mid - root))).detach()
    xy = torch.stack([((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5, ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5], -1)
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    xd, yd = xy[:, 0].detach(), xy[:, 1].detach()

    def tile(v, g):          # auxiliary.h:62-72: (int) truncates towards zero
        return torch.clamp(torch.trunc(v / BLOCK), 0, g).long()
    rect = torch.stack([tile(xd - radius, gx), tile(yd - radius, gy), tile(xd + radius + BLOCK - 1, gx), tile(yd + radius + BLOCK - 1, gy)], -1)
    tiles = (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])
    visible = in_front & (det != 0) & (tiles > 0)
    colour = colors_precomp if colors_precomp is not None else sh_colour(st["sh_degree"], means3D, st["campos"].to(DT), shs)
    return {"visible": visible, "radii": torch.where(visible, radius, torch.zeros_like(radius)).long(), "xy": xy, "depth": p_view[:, 2],
            "view_point": p_view, "ts": ts, "conic": conic, "opacity": opacities.reshape(-1) * coef, "camera_plane": camera_plane,
            "ray_plane": ray_plane, "normal": normal, "colour": colour, "rect": rect, "fx": fx, "fy": fy}


def rasterize(means3D, opacities, scales, rotations, st, shs=None, colors_precomp=None):
    """-> dict(color [3,H,W], radii [P], coord [3,H,W], mcoord [3,H,W], depth [1,H,W], mdepth [1,H,W], alpha [1,H,W], normal [3,H,W])
    in the order / shapes of _RasterizeGaussians.forward (diff_gaussian_rasterization/__init__.py:88)."""
    means3D, opacities, scales, rotations = (t.to(DT) for t in (means3D, opacities, scales, rotations))
    shs = shs.to(DT) if shs is not None else None
    colors_precomp = colors_precomp.to(DT) if colors_precomp is not None else None
    g = preprocess(means3D, opacities, scales, rotations, shs, colors_precomp, st)
    H, W = st["image_height"], st["image_width"]
    fx, fy = g["fx"], g["fy"]
    bg = st["bg"].to(DT)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=DT), torch.arange(W, dtype=DT), indexing="ij")
    px, py = xs.reshape(-1), ys.reshape(-1)
    tile_x, tile_y = (px / BLOCK).long(), (py / BLOCK).long()
    ln = torch.sqrt(((px - W / 2.0) / fx) ** 2 + ((py - H / 2.0) / fy) ** 2 + 1)
    n = H * W
    T = torch.ones(n, dtype=DT)
    done = torch.zeros(n, dtype=torch.bool)
    C = torch.zeros(n, 3, dtype=DT)
    Coord, mCoord, Normal = torch.zeros(n, 3, dtype=DT), torch.zeros(n, 3, dtype=DT), torch.zeros(n, 3, dtype=DT)
    Depth, mDepth, weight = torch.zeros(n, dtype=DT), torch.zeros(n, dtype=DT), torch.zeros(n, dtype=DT)
    any_contrib = torch.zeros(n, dtype=torch.bool)
    idx = torch.nonzero(g["visible"]).reshape(-1)
    # key order: view depth as the fp32 bit pattern of a positive float == numeric order; the radix sort is stable (index order on ties)
    order = idx[torch.argsort(g["depth"][idx].float(), stable=True)]
    for k in order.tolist():
        r = g["rect"][k]
        member = (tile_x >= r[0]) & (tile_x < r[2]) & (tile_y >= r[1]) & (tile_y < r[3])
        dx, dy = g["xy"][k, 0] - px, g["xy"][k, 1] - py
        con = g["conic"][k]
        power = -0.5 * (con[0] * dx * dx + con[2] * dy * dy) - con[1] * dx * dy
        alpha = torch.clamp(g["opacity"][k] * torch.exp(power), max=0.99)
        ok = member & ~done & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        stop = ok & (test_T < 1e-4)
        done = done | stop
        ok = ok & ~stop
        aT = torch.where(ok, alpha * T, torch.zeros_like(T))
        before = ok & (T > 0.5)
        C = C + aT[:, None] * g["colour"][k][None]
        cp = g["camera_plane"][k]
        coord = torch.stack([g["view_point"][k, 0] + cp[0] * dx + cp[1] * dy, g["view_point"][k, 1] + cp[2] * dx + cp[3] * dy,
                             g["view_point"][k, 2] + cp[4] * dx + cp[5] * dy], -1)
        Coord = Coord + aT[:, None] * coord
        mCoord = torch.where(before[:, None], coord, mCoord)
        t = g["ts"][k] + g["ray_plane"][k, 0] * dx + g["ray_plane"][k, 1] * dy
        Depth = Depth + aT * t
        mDepth = torch.where(before, t, mDepth)
        Normal = Normal + aT[:, None] * g["normal"][k][None]
        weight = weight + aT
        T = torch.where(ok, test_T, T)
        any_contrib = any_contrib | ok
    color = C + T[:, None] * bg[None]
    safe_w = torch.where(any_contrib, weight, torch.ones_like(weight))
    coord_o = torch.where(any_contrib[:, None], Coord / safe_w[:, None], torch.zeros_like(Coord))
    depth_o = torch.where(any_contrib, Depth / ln / safe_w, torch.zeros_like(Depth))
    nlen = torch.sqrt((Normal * Normal).sum(-1) + (~any_contrib).to(DT))                # (no contribution: avoid sqrt'(0) in autograd)
    normal_o = torch.where(any_contrib[:, None], Normal / torch.clamp(nlen, min=1e-12)[:, None], torch.zeros_like(Normal))
    sh = lambda t, ch: t.reshape(H, W, ch).permute(2, 0, 1)
    return {"color": sh(color, 3), "radii": g["radii"], "coord": sh(coord_o, 3), "mcoord": sh(mCoord, 3), "depth": sh(depth_o[:, None], 1),
            "mdepth": sh((mDepth / ln)[:, None], 1), "alpha": sh(weight[:, None], 1), "normal": sh(normal_o, 3)}


def camera_settings(H, W, fovx, fovy, w2c, bg=(0.0, 0.0, 0.0), sh_degree=0, kernel_size=0.0, znear=0.01, zfar=100.0):
    """GaussianRasterizationSettings as a dict, built the way the reference's cameras do (gaussian/utils/graphics_utils.py
    getProjectionMatrix / scene cameras: viewmatrix = W2C^T, projmatrix = viewmatrix @ P^T)."""
    tanx, tany = math.tan(fovx * 0.5), math.tan(fovy * 0.5)
    top, right = tany * znear, tanx * znear
    Pm = torch.zeros(4, 4, dtype=DT)
    Pm[0, 0] = 2.0 * znear / (2 * right)
    Pm[1, 1] = 2.0 * znear / (2 * top)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    view = torch.as_tensor(w2c, dtype=DT).T.contiguous()
    return {"image_height": H, "image_width": W, "tanfovx": tanx, "tanfovy": tany, "kernel_size": kernel_size, "bg": torch.tensor(bg, dtype=DT),
            "scale_modifier": 1.0, "viewmatrix": view, "projmatrix": view @ Pm.T, "sh_degree": sh_degree,
            "campos": torch.linalg.inv(torch.as_tensor(w2c, dtype=DT))[:3, 3]}
