"""TEST INFRASTRUCTURE ONLY -- fp64 restatement of ONE iteration of the Gaussian mapper's loss (hislam2/gs_backend_per_frame.py:451-587
`optimization`) on top of oracle/gs_oracle.py, differentiable by torch autograd: the independent reference for the tape-free trainer
(cut3r_slam_amd/gs_step.py) and for the tensor-op formulation in gs_mapper.py.

  render           hislam2/gaussian/renderer/__init__.py:89-152 with utils/slam_utils.py:93-102 (pose = exp([tau, phi]) * T_w2c applied to the
                   Gaussians, identity view matrix), activations of scene/gaussian_model.py:77-101
  colour           0.8 * mean |gt - image| + 0.2 * (1 - SSIM)           (loss_utils.py:129-170: 11x11 window, sigma 1.5, zero padding)
  depth            lambda_depth * mean_mask |1/depth - 1/gt_depth|, mask = gt_depth > 0.001 and depth > 0.001
  normal           lambda_normal * mean_mask (1 - n(depth) . n(gt_depth)), n = camera-frame normal from central differences of the
                   back-projected neighbours, zero on the image border
  isotropy         lambda_iso * sum_visible |s - mean s| / max(3 n_visible, 1)

PARITY UNPINNED against the reference (its CUDA rasteriser cannot run here); this file pins the two product formulations to an
independent reading of the cited lines."""
import math

import torch
import torch.nn.functional as F

from . import gs_oracle as GO

DT = torch.float64
SH_C0 = 0.28209479177387814


def hat(v):
    z = torch.zeros((), dtype=DT)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def se3_exp_matrix(tau, phi):
    """4x4 exp of the twist (tau, phi) by torch.matrix_exp (differentiable)"""
    M = torch.zeros(4, 4, dtype=DT)
    M = M.clone()
    M[:3, :3] = hat(phi)
    M[:3, 3] = tau
    return torch.matrix_exp(M)


def quat_rxyz_to_rot(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                        2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def rot_to_quat_rxyz(R):
    """unit quaternion (r, x, y, z) of a rotation matrix (trace branch: the tests use small rotations)"""
    r = torch.sqrt(torch.clamp(1 + R[0, 0] + R[1, 1] + R[2, 2], min=1e-12)) / 2
    return torch.stack([r, (R[2, 1] - R[1, 2]) / (4 * r), (R[0, 2] - R[2, 0]) / (4 * r), (R[1, 0] - R[0, 1]) / (4 * r)])


def quat_mul_rxyz(a, b):
    r1, x1, y1, z1 = a.unbind(-1)
    r2, x2, y2, z2 = b.unbind(-1)
    return torch.stack([r1 * r2 - x1 * x2 - y1 * y2 - z1 * z2, r1 * x2 + x1 * r2 + y1 * z2 - z1 * y2, r1 * y2 - x1 * z2 + y1 * r2 + z1 * x2,
                        r1 * z2 + x1 * y2 - y1 * x2 + z1 * r2], -1)


def depth_to_normal(depth, K):
    """[H,W] -> [3,H,W]: normalised cross product of the central differences of the back-projected points, zero on the border"""
    fx, fy, cx, cy = K
    H, W = depth.shape
    y, x = torch.meshgrid(torch.arange(H, dtype=DT), torch.arange(W, dtype=DT), indexing="ij")
    pts = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(x)], 0) * depth[None]
    dx = pts[:, 1:-1, 2:] - pts[:, 1:-1, :-2]
    dy = pts[:, 2:, 1:-1] - pts[:, :-2, 1:-1]
    n = F.normalize(torch.cross(dx, dy, dim=0), dim=0)
    return F.pad(n, (1, 1, 1, 1))


def ssim(a, b, window=11, sigma=1.5):
    """gaussian/utils/loss_utils.py:51-68,129-170.  The reference builds its window in fp32 (torch.Tensor of Python floats, normalised and
    multiplied out in fp32) and only then casts it to the image type: the same here, so that the fp64 value equals the reference's
    (tests/golden/gs_utils.npz) to rounding."""
    g = torch.tensor([math.exp(-((x - window // 2) ** 2) / float(2 * sigma ** 2)) for x in range(window)], dtype=torch.float32)
    g = (g / g.sum())[:, None]
    w = (g @ g.T)[None, None].expand(a.shape[0], 1, window, window).contiguous().to(a.dtype)
    f = lambda t: F.conv2d(t[None], w, padding=window // 2, groups=a.shape[0])[0]
    mu1, mu2 = f(a), f(b)
    s11, s22, s12 = f(a * a) - mu1 * mu1, f(b * b) - mu2 * mu2, f(a * b) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))).mean()


def render(theta, w2c, tau, phi, H, W, K):
    """theta [P,14] (xyz | DC colour | opacity logit | log scale | quaternion rxyz), w2c [4,4], increments tau / phi [3] -> rasteriser dict"""
    fx, fy, cx, cy = K
    T = se3_exp_matrix(tau, phi) @ w2c
    xyz = theta[:, 0:3] @ T[:3, :3].T + T[:3, 3]
    q = quat_mul_rxyz(rot_to_quat_rxyz(T[:3, :3])[None], F.normalize(theta[:, 10:14], dim=-1))
    st = GO.camera_settings(H, W, 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy)), torch.eye(4, dtype=DT))
    assert abs(cx - W / 2) < 1e-9 and abs(cy - H / 2) < 1e-9, "the restatement's camera has a centred principal point"
    return GO.rasterize(xyz, torch.sigmoid(theta[:, 6:7]), torch.exp(theta[:, 7:10]), q, st, shs=theta[:, None, 3:6])


def mapping_loss(theta, w2c, tau, phi, gt_image, gt_depth, K, lambda_depth, lambda_normal, lambda_iso):
    H, W = gt_depth.shape
    out = render(theta, w2c, tau, phi, H, W, K)
    image, depth = out["color"], out["depth"][0]
    l_rgb = (gt_image - image).abs().mean()
    mask = (gt_depth > 0.001) & (depth > 0.001)
    nmask = mask.sum().clamp_min(1)
    l_depth = ((1.0 / depth.clamp_min(1e-12) - 1.0 / gt_depth.clamp_min(1e-12)).abs() * mask).sum() / nmask
    n_r, n_g = depth_to_normal(depth, K), depth_to_normal(gt_depth, K)
    l_normal = ((1 - (n_r * n_g).sum(0)) * mask).sum() / nmask
    vis = out["radii"] > 0
    sc = torch.exp(theta[:, 7:10])
    iso = ((sc - sc.mean(dim=1, keepdim=True)).abs() * vis[:, None]).sum() / (3 * vis.sum()).clamp_min(1)
    return 0.8 * l_rgb + lambda_depth * l_depth + lambda_normal * l_normal + 0.2 * (1.0 - ssim(image, gt_image)) + lambda_iso * iso
