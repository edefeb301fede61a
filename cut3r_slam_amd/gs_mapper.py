"""Gaussian-splatting mapper on the gfx950 rasteriser: the part of the reference's GS backend that sits between the tracker and the
rasteriser.  Mirrors, with the same names, argument meaning and loss terms:

  GaussianMap            hislam2/gaussian/scene/gaussian_model.py:34-105,146-216,336-431,547-790 (parameters and activations, new
                         Gaussians from a keyframe's pointmap with 3-NN scales, Adam groups, prune / clone / split)
  Camera, get_pose,      hislam2/gaussian/utils/camera_utils.py, slam_utils.py:62-102 (world->camera R, T plus the tangent-space
  update_pose            deltas the optimiser moves; new_w2c = exp([trans, rot]) @ T_w2c)
  render                 hislam2/gaussian/renderer/__init__.py:89-152 (Gaussians moved into the camera frame, identity view matrix)
  GSMapper.pose_refine   hislam2/gs_backend_per_frame.py:202-326
  GSMapper.optimization  hislam2/gs_backend_per_frame.py:451-587
  GSMapper.add_new_view  hislam2/gs_backend_per_frame.py:87-121
  GSMapper.gaussian_reinit, finalize, save / load, eval_rendering_kf   :865-944, :1067-1102
  GSMapper.global_BA     hislam2/gs_backend_per_frame.py:946-1062 (all keyframes, poses and Gaussians together, one random keyframe per
                         iteration, densification / opacity-reset / learning-rate schedule)

The reference's backend cannot run here (CUDA rasteriser, open3d, munch): PARITY UNPINNED, covered by functional tests on a
synthetic scene (tests/test_gs_mapper_gpu.py).  The rasteriser and the 3-NN search are the HIP kernels of csrc/gs.hip; the loss
terms and the Adam updates are plain torch tensor arithmetic on the GPU.  Not built: GUI, ply export, LPIPS and the TSDF / mesh evaluation utilities (PSNR / SSIM of the keyframes and a safetensors checkpoint are).
`gaussain_update` (the map correction after a loop closure, :701-774) composes rotations consistently by default; see its docstring."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .gaussian_rasterizer import GaussianRasterizationSettings, GaussianRasterizer, distCUDA2, fused_ssim, pixel_losses, refine_losses
from .lietorch import SE3, SO3

SH_C0 = 0.28209479177387814


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def pose_vec_to_matrix(p):
    """[.., 7] (t, q_xyzw) camera->world -> [.., 4, 4]"""
    return SE3(torch.as_tensor(p, dtype=torch.float32)).matrix()


class GaussianMap:
    """Parameters of all Gaussians in ONE [P,14] block (xyz 3 | DC colour 3 | opacity logit 1 | log scale 3 | quaternion 4) with its Adam
    moments beside it and ONE step count (torch.optim.Adam keeps one `step` per parameter group and the reference's
    cat_tensors_to_optimizer / _prune_optimizer / replace_tensor_to_optimizer, gaussian_model.py:510-599, zero-pad or cut the moments
    but keep that count: appended Gaussians get no bias correction of their own): growth and pruning are row operations on three
    tensors, the optimiser step is one pass over the block with a per-column learning rate, `p[name]` are column views."""
    GROUPS = ("xyz", "f_dc", "opacity", "scaling", "rotation")
    COLS = {"xyz": (0, 3), "f_dc": (3, 6), "opacity": (6, 7), "scaling": (7, 10), "rotation": (10, 14)}

    def __init__(self, opt, device="cuda:0", isotropic=False):
        self.device, self.isotropic = torch.device(device), isotropic
        z = lambda *s: torch.zeros(*s, device=self.device)
        self.theta = z(0, 14).requires_grad_(True)
        self.m, self.v = z(0, 14), z(0, 14)
        lr = {"xyz": opt["position_lr_init"], "f_dc": opt["feature_lr"], "opacity": opt["opacity_lr"], "scaling": opt["scaling_lr"],
              "rotation": opt["rotation_lr"]}
        self.lr = z(1, 14)
        for k, (a, b) in self.COLS.items():
            self.lr[0, a:b] = lr[k]
        self.percent_dense = opt.get("percent_dense", 0.01)
        self.step_count = z(1)                                            # device scalar: a captured iteration increments it in place
        self.steps = 0                                                    # the same count on the host (the fused trainer's bias corrections)
        self._steps_dev_stale = False                                     # the fused trainer advanced `steps` without touching the device scalar
        self.kf_id = torch.zeros(0, dtype=torch.int32, device=self.device)
        self.max_radii2D = z(0)
        self.grad_accum, self.grad_accum_abs, self.denom = z(0, 1), z(0, 1), z(0, 1)

    @property
    def p(self):
        return {k: self.theta[:, a:b] for k, (a, b) in self.COLS.items()}

    # ---- activations (gaussian_model.py:77-101)
    def __len__(self):
        return self.theta.shape[0]

    @property
    def get_xyz(self):
        return self.theta[:, 0:3]

    @property
    def get_scaling(self):
        return torch.exp(self.theta[:, 7:10])

    @property
    def get_rotation(self):
        return F.normalize(self.theta[:, 10:14], dim=-1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self.theta[:, 6:7])

    @property
    def get_features(self):
        return self.theta[:, None, 3:6]                                   # [P,1,3]: SH degree 0, as the backend runs the model

    # ---- growth
    def _append(self, new, kf_id):
        rows = torch.cat([new[k].to(self.device).float().reshape(-1, b - a) for k, (a, b) in self.COLS.items()], 1)
        n = rows.shape[0]
        z = lambda *s: torch.zeros(*s, device=self.device)
        self.theta = torch.cat([self.theta.detach(), rows], 0).requires_grad_(True)
        self.m, self.v = torch.cat([self.m, z(n, 14)], 0), torch.cat([self.v, z(n, 14)], 0)
        self.kf_id = torch.cat([self.kf_id, kf_id.to(self.device, torch.int32)], 0)
        self.max_radii2D = torch.cat([self.max_radii2D, z(n)], 0)
        self.grad_accum, self.denom = torch.cat([self.grad_accum, z(n, 1)], 0), torch.cat([self.denom, z(n, 1)], 0)
        self.grad_accum_abs = torch.cat([self.grad_accum_abs, z(n, 1)], 0)

    def extend_from_pcd_seq(self, submap_idx=-1, rgb=None, pointmap=None, conf=None, point_size=1.0):
        """gaussian_model.py:150-216,363-372: one Gaussian per pointmap pixel with conf > 0; scale = sqrt of the mean squared 3-NN distance,
        identity rotation, opacity 0.1, colour as the DC SH coefficient.  rgb [.., 3] in [0,1], pointmap [.., 3], conf [..] or None."""
        pts = torch.as_tensor(pointmap, dtype=torch.float32, device=self.device).reshape(-1, 3)
        col = torch.as_tensor(rgb, dtype=torch.float32, device=self.device).reshape(-1, 3)
        if conf is not None:
            keep = torch.as_tensor(conf, device=self.device).reshape(-1) > 0.0
            pts, col = pts[keep], col[keep]
        if pts.shape[0] < 5:
            return 0
        d2 = torch.clamp_min(distCUDA2(pts), 1e-7) * point_size
        scales = torch.log(torch.sqrt(d2))[:, None].repeat(1, 3)
        rots = torch.zeros(pts.shape[0], 4, device=self.device)
        rots[:, 0] = 1
        new = {"xyz": pts, "f_dc": (col - 0.5) / SH_C0, "opacity": inverse_sigmoid(0.1 * torch.ones(pts.shape[0], 1, device=self.device)),
               "scaling": scales, "rotation": rots}
        self._append(new, torch.full((pts.shape[0],), int(submap_idx)))
        return pts.shape[0]

    def prune_points(self, mask):
        keep = ~mask
        self.theta = self.theta.detach()[keep].requires_grad_(True)
        self.m, self.v = self.m[keep], self.v[keep]
        self.kf_id, self.max_radii2D = self.kf_id[keep], self.max_radii2D[keep]
        self.grad_accum, self.grad_accum_abs, self.denom = self.grad_accum[keep], self.grad_accum_abs[keep], self.denom[keep]

    def reset_opacity(self):
        """gaussian_model.py:483-486: every opacity back to 0.15, its optimiser state cleared"""
        with torch.no_grad():
            self.theta[:, 6:7] = float(inverse_sigmoid(torch.tensor(0.15)))
            self.m[:, 6:7] = 0
            self.v[:, 6:7] = 0

    def reset_moments(self, rows):
        """optimiser moments of the given Gaussians back to those of new ones (what the reference's prune + re-append does; the step
        count of the group is kept, as there)"""
        self.m[rows] = 0
        self.v[rows] = 0

    def add_densification_stats(self, viewspace_grad, update_filter):
        """gaussian_model.py:779-783: per visible Gaussian the norm of the screen-space gradient (x, y) and -- the RaDe-GS / AbsGS flavour the
        reference vendors -- the absolute-gradient statistic the rasteriser's backward leaves in the third channel of means2D.grad"""
        f = update_filter[:, None]
        self.grad_accum += torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * f
        if viewspace_grad.shape[1] > 2:
            self.grad_accum_abs += torch.norm(viewspace_grad[:, 2:], dim=-1, keepdim=True) * f
        self.denom += f.float()

    def _split_noise(self, n):
        """standard-normal draws of densify_and_split (gaussian_model.py:652-654 torch.normal(0, stds) = stds * these); its own method so
        that a test can replay the reference's draws"""
        return torch.randn(n, 3, device=self.device)

    def reset_densification_stats(self):
        self.grad_accum.zero_()
        self.grad_accum_abs.zero_()
        self.denom.zero_()
        self.max_radii2D.zero_()

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        """gaussian_model.py:639-777.  A Gaussian is selected when its mean screen gradient reaches `max_grad` OR its mean absolute-gradient
        statistic reaches the (1 - ratio) quantile of that statistic, ratio = the share selected by the first rule (:751-756); the small ones
        (largest scale <= percent_dense * extent) are cloned, the large ones replaced by two samples of themselves with scales / 1.6; then the
        transparent (opacity < min_opacity), the oversized in the world (> 0.1 extent, only with `max_screen_size`) and the degenerate
        (largest scale < 5e-4, :767-768) are pruned.  densification_postfix (:629-633) zeroes the statistics INCLUDING max_radii2D before the
        prune mask is formed, so the reference's screen-size rule never fires after a densification: reproduced."""
        if len(self) == 0:
            return
        grads = self.grad_accum / self.denom
        grads[grads.isnan()] = 0.0
        grads_abs = self.grad_accum_abs / self.denom
        grads_abs[grads_abs.isnan()] = 0.0
        hot = grads[:, 0] >= max_grad
        ratio = hot.float().mean()
        flat = grads_abs.reshape(-1)
        if flat.numel() <= (1 << 24):
            Q = torch.quantile(flat, 1 - ratio)
        else:                                   # torch.quantile refuses more than 2^24 elements: the same linear interpolation on a sort
            srt = torch.sort(flat).values
            pos = (1 - ratio) * (flat.numel() - 1)
            lo = pos.floor().long().clamp(0, flat.numel() - 1)
            hi = pos.ceil().long().clamp(0, flat.numel() - 1)
            Q = srt[lo] + (srt[hi] - srt[lo]) * (pos - lo.float())
        hot = hot | (grads_abs[:, 0] >= Q)
        big = self.get_scaling.detach().max(dim=1).values > self.percent_dense * extent
        clone, split = hot & ~big, hot & big
        th = self.theta.detach()
        new = []
        if clone.any():
            new.append((th[clone].clone(), self.kf_id[clone]))
        if split.any():
            n = int(split.sum())
            std = self.get_scaling.detach()[split].repeat(2, 1)
            R = SO3_matrix(self.get_rotation.detach()[split]).repeat(2, 1, 1)
            rows = th[split].repeat(2, 1)
            rows[:, 0:3] = (R @ (self._split_noise(2 * n) * std)[:, :, None])[:, :, 0] + rows[:, 0:3]
            rows[:, 7:10] = torch.log(std / (0.8 * 2))
            new.append((rows, self.kf_id[split].repeat(2)))
        n_before = len(self)
        for rows, ids in new:
            self._append({k: rows[:, a:b] for k, (a, b) in self.COLS.items()}, ids)
        self.reset_densification_stats()
        drop = torch.zeros(len(self), dtype=torch.bool, device=self.device)
        drop[:n_before] = split
        smax = self.get_scaling.detach().max(dim=1).values
        drop |= (self.get_opacity.detach() < min_opacity)[:, 0]
        if max_screen_size:
            drop |= (self.max_radii2D > max_screen_size) | (smax > 0.1 * extent)
        drop |= smax < 5e-4
        self.prune_points(drop)

    # ---- Adam (torch.optim.Adam(lr per group, eps=1e-15) of gaussian_model.py:374-417; one step count for the block, see the class
    #      docstring)
    def zero_grad(self):
        self.theta.grad = None

    @torch.no_grad()
    def step_like_reference(self, densified=False, reset=False, **kw):
        """One optimiser step in the reference's order (gs_backend_per_frame.py:425-438, 564-577, 920-935, 1025-1041): densify_and_prune /
        reset_opacity run BEFORE `optimizer.step()`, and both re-create `nn.Parameter`s (cat_tensors_to_optimizer, _prune_optimizer,
        replace_tensor_to_optimizer: gaussian_model.py:488-560) whose `.grad` is then None -- so after a densification Adam skips EVERY group
        (no update, no moment update, no step increment) and after an opacity reset it skips the opacity group."""
        if densified:
            return
        if reset:
            keep = self.theta.detach()[:, 6:7].clone()
            self.step(**kw)
            with torch.no_grad():
                self.theta[:, 6:7] = keep
                self.m[:, 6:7] = 0
                self.v[:, 6:7] = 0
            return
        self.step(**kw)

    def step(self, b1=0.9, b2=0.999, eps=1e-15):
        g = self.theta.grad
        if g is None:
            return
        if self._steps_dev_stale:
            self.step_count.fill_(float(self.steps))
            self._steps_dev_stale = False
        self.step_count += 1
        self.steps += 1
        self.m.mul_(b1).add_(g, alpha=1 - b1)
        self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
        self.theta.sub_(self.lr * (self.m / bc1) / ((self.v / bc2).sqrt() + eps))


def position_lr(op, iteration):
    """general_utils.py:41-56 `get_expon_lr_func` as gaussian_model.py:395-431 sets it up: log-linear from position_lr_init to
    position_lr_final over position_lr_max_steps + 1000 steps (no delay), clamped at the end"""
    t = min(max(iteration / (op.get("position_lr_max_steps", 20000) + 1000), 0.0), 1.0)
    return math.exp(math.log(op["position_lr_init"]) * (1 - t) + math.log(op["position_lr_final"]) * t)


def SO3_matrix(q):
    """(r, x, y, z) unit quaternions [n,4] -> rotation matrices [n,3,3]"""
    r, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                        2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


class Camera:
    """camera_utils.py Camera.init_from_tracking: image [3,H,W] in [0,1], depth [H,W], world->camera w2c [4,4], pinhole K"""
    half_pixel_center = False

    def __init__(self, uid, image, depth, w2c, fx, fy, cx, cy, tstamp=None, device="cuda:0"):
        dev = torch.device(device)
        self.uid, self.tstamp, self.device = uid, tstamp, dev
        # dense row-major copies: the loss kernels of the tape-free trainer read these through raw pointers (a permuted HWC view handed in by
        # a caller would otherwise be read with the wrong strides)
        self.original_image = image.to(dev, torch.float32).contiguous()
        self.depth = depth.to(dev, torch.float32).contiguous()
        self.image_height, self.image_width = image.shape[-2:]
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.FoVx, self.FoVy = 2 * math.atan(self.image_width / (2 * self.fx)), 2 * math.atan(self.image_height / (2 * self.fy))
        w2c = torch.as_tensor(w2c, dtype=torch.float32, device=dev)
        self.R, self.T = w2c[:3, :3].clone(), w2c[:3, 3].clone()
        self.w2c_data = SE3_from_matrix(w2c).detach()                   # (t, q_xyzw): the pose the lie kernels compose the deltas with
        self.cam_rot_delta = torch.zeros(3, device=dev, requires_grad=True)
        self.cam_trans_delta = torch.zeros(3, device=dev, requires_grad=True)
        # per-view affine colour model (camera_utils.py exposure_a / exposure_b; used when Training.compensate_exposure)
        self.exposure_a = torch.eye(3, device=dev).requires_grad_(True)
        self.exposure_b = torch.zeros(3, device=dev, requires_grad=True)
        # graphics_utils.getProjectionMatrix2 (:72-93: left = cx - W, right = cx in units of znear / fx, so P[0,2] = 2 cx / W - 1),
        # stored transposed like the reference's cameras.  Through the rasteriser's ndc2Pix (auxiliary.h:57-60, pixel u = ((ndc + 1) W -
        # 1) / 2) a point lands at u = fx X/Z + cx - 0.5: maps trained here and in the reference render alike.  Camera.half_pixel_center
        # = True (a declared deviation, off by default) puts it at u = fx X/Z + cx exactly.
        znear, zfar, W, H = 0.01, 100.0, self.image_width, self.image_height
        hp = 1.0 if Camera.half_pixel_center else 0.0
        P = torch.zeros(4, 4)
        P[0, 0], P[1, 1] = 2 * self.fx / W, 2 * self.fy / H
        P[0, 2], P[1, 2] = (2 * self.cx + hp) / W - 1, (2 * self.cy + hp) / H - 1
        P[3, 2], P[2, 2], P[2, 3] = 1.0, zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)
        self.projection_matrix = P.T.contiguous().to(dev)
        self.projection_matrix_host = P.T.contiguous()                  # the rasteriser reads its matrices on the host: no copy back per render

    def update_RT(self, R, T, data=None):
        """in place: a captured iteration (GSMapper.optimization(graph=True)) writes the new pose through these tensors on every replay"""
        self.R.copy_(R.detach())
        self.T.copy_(T.detach())
        if data is None:
            M = torch.eye(4, device=self.device)
            M[:3, :3], M[:3, 3] = self.R, self.T
            data = SE3_from_matrix(M)
        self.w2c_data.copy_(data.detach())

    @property
    def camera_center(self):
        return -(self.R.T @ self.T)


def get_pose_se3(camera):
    """slam_utils.py:93-102 on the lie kernels: exp([trans delta, rot delta]) * T_w2c as an SE3 element (autograd to the deltas)"""
    tau = torch.cat([camera.cam_trans_delta, camera.cam_rot_delta], 0)
    return SE3.exp(tau[None]) * SE3(camera.w2c_data[None])


def get_pose(camera):
    """slam_utils.py:93-102: the 4x4 world->camera matrix"""
    return get_pose_se3(camera).matrix()[0]


@torch.no_grad()
def update_pose(camera):
    """slam_utils.py:77-91"""
    new = get_pose_se3(camera)
    M = new.matrix()[0]
    camera.update_RT(M[:3, :3], M[:3, 3], data=new.data[0])
    camera.cam_rot_delta.data.fill_(0)
    camera.cam_trans_delta.data.fill_(0)


def _rotmat_to_quat(R):
    """renderer/__init__.py:165-195 as tensor selects (no host round trip): the four largest-diagonal candidates are formed and the
    reference's branch order picks one; (r, x, y, z), differentiable"""
    m00, m11, m22 = R[0, 0], R[1, 1], R[2, 2]
    r = torch.sqrt(torch.clamp(1 + m00 + m11 + m22, min=1e-12)) / 2
    x = torch.sqrt(torch.clamp(1 + m00 - m11 - m22, min=1e-12)) / 2
    y = torch.sqrt(torch.clamp(1 - m00 + m11 - m22, min=1e-12)) / 2
    z = torch.sqrt(torch.clamp(1 - m00 - m11 + m22, min=1e-12)) / 2
    c0 = torch.stack([r, (R[2, 1] - R[1, 2]) / (4 * r), (R[0, 2] - R[2, 0]) / (4 * r), (R[1, 0] - R[0, 1]) / (4 * r)])
    c1 = torch.stack([(R[2, 1] - R[1, 2]) / (4 * x), x, (R[0, 1] + R[1, 0]) / (4 * x), (R[0, 2] + R[2, 0]) / (4 * x)])
    c2 = torch.stack([(R[0, 2] - R[2, 0]) / (4 * y), (R[0, 1] + R[1, 0]) / (4 * y), y, (R[1, 2] + R[2, 1]) / (4 * y)])
    c3 = torch.stack([(R[1, 0] - R[0, 1]) / (4 * z), (R[0, 2] + R[2, 0]) / (4 * z), (R[1, 2] + R[2, 1]) / (4 * z), z])
    with torch.no_grad():
        t0, t1, t2 = (m00 + m11 + m22) > 0, (m00 > m11) & (m00 > m22), m11 > m22
    return torch.where(t0, c0, torch.where(t1, c1, torch.where(t2, c2, c3)))


def _quat_mult(a, b):
    r1, x1, y1, z1 = a.unbind(-1)
    r2, x2, y2, z2 = b.unbind(-1)
    return torch.stack([r1 * r2 - x1 * x2 - y1 * y2 - z1 * z2, r1 * x2 + x1 * r2 + y1 * z2 - z1 * y2, r1 * y2 - x1 * z2 + y1 * r2 + z1 * x2,
                        r1 * z2 + x1 * y2 - y1 * x2 + z1 * r2], -1)


def _quat_mult_xyzw(a, b):
    """Hamilton product of (x, y, z, w) quaternions (lietorch SO3 data order)"""
    r = _quat_mult(torch.cat([a[:, 3:], a[:, :3]], -1), torch.cat([b[:, 3:], b[:, :3]], -1))
    return torch.cat([r[:, 1:], r[:, :1]], -1)


def render(viewpoint, pc, bg_color, scaling_modifier=1.0):
    """renderer/__init__.py:89-152: the Gaussians are moved into the camera frame (so the pose receives gradients through means and
    rotations) and rasterised with an identity view matrix; returns the reference's dict"""
    pose = get_pose_se3(viewpoint)
    xyz = pose.act(pc.get_xyz)                                            # (a [P,3] x [3,3] product through rocBLAS cost 0.14 ms)
    # camera rotation x Gaussian rotation on the SO3 kernel (its data order is (x, y, z, w); the Gaussians keep (r, x, y, z))
    gq = pc.get_rotation
    rq = (SO3(pose.data[:, 3:]) * SO3(torch.cat([gq[:, 1:], gq[:, :1]], -1))).data
    rot = torch.cat([rq[:, 3:], rq[:, :3]], -1)
    screenspace_points = torch.zeros_like(xyz, requires_grad=True)
    bg_host = getattr(pc, "_bg_host", None)
    if bg_host is None or bg_host[0] is not bg_color:
        bg_host = pc._bg_host = (bg_color, bg_color.detach().cpu())
    st = GaussianRasterizationSettings(image_height=int(viewpoint.image_height), image_width=int(viewpoint.image_width),
                                       tanfovx=math.tan(viewpoint.FoVx * 0.5), tanfovy=math.tan(viewpoint.FoVy * 0.5), kernel_size=0.0,
                                       bg=bg_host[1], scale_modifier=scaling_modifier, viewmatrix=torch.eye(4),
                                       projmatrix=viewpoint.projection_matrix_host, sh_degree=0, campos=torch.zeros(3), prefiltered=False,
                                       require_coord=True, require_depth=True, debug=False)
    color, radii, coord, mcoord, depth, mdepth, alpha, normal = GaussianRasterizer(st)(
        means3D=xyz, means2D=screenspace_points, opacities=pc.get_opacity, shs=pc.get_features, scales=pc.get_scaling, rotations=rot)
    return {"render": color, "mask": alpha, "expected_coord": coord, "median_coord": mcoord, "depth": depth, "median_depth": mdepth,
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii, "normal": normal, "n_touched": None}


def project2world(c2w, depths, fx, fy, cx, cy):
    """slam_utils.py:108-140: depth maps [N,H,W] -> world points [N,H,W,3]"""
    N, H, W = depths.shape
    y, x = torch.meshgrid(torch.arange(H, device=depths.device).float(), torch.arange(W, device=depths.device).float(), indexing="ij")
    pc = torch.stack([(x - cx) / fx * depths, (y - cy) / fy * depths, depths], -1)
    return pc @ c2w[:, None, :3, :3].transpose(-1, -2) + c2w[:, None, None, :3, 3]


def depth_to_normal(viewpoint, depth):
    """camera-frame normals from a depth map [1,H,W] (cross product of the back-projected neighbours; borders zero) -> [3,H,W]"""
    rays = getattr(viewpoint, "_rays", None)
    if rays is None or rays.shape[-2:] != depth.shape[-2:]:               # ((x - cx) / fx, (y - cy) / fy, 1) per pixel: fixed per camera
        H, W = depth.shape[-2:]
        y, x = torch.meshgrid(torch.arange(H, device=depth.device).float(), torch.arange(W, device=depth.device).float(), indexing="ij")
        rays = viewpoint._rays = torch.stack([(x - viewpoint.cx) / viewpoint.fx, (y - viewpoint.cy) / viewpoint.fy, torch.ones_like(x)], 0)
    pts = rays * depth
    dx = pts[:, 1:-1, 2:] - pts[:, 1:-1, :-2]
    dy = pts[:, 2:, 1:-1] - pts[:, :-2, 1:-1]
    n = F.normalize(torch.cross(dx, dy, dim=0), dim=0)
    return F.pad(n, (1, 1, 1, 1))


def ssim(a, b):
    """gaussian/utils/loss_utils.py:129-170 ssim on the fused kernels: a = the rendering (receives gradients), b = the keyframe image"""
    return fused_ssim(a, b)


def ssim_torch(a, b, window=11, sigma=1.5):
    """the same quantity in plain tensor operations (tests compare the fused kernels with it)"""
    g = torch.exp(-((torch.arange(window, device=a.device).to(a.dtype) - window // 2) ** 2) / (2 * sigma * sigma))
    g = (g / g.sum())[:, None]
    w = (g @ g.T)[None, None].expand(a.shape[0], 1, window, window).contiguous()
    f = lambda t: F.conv2d(t[None], w, padding=window // 2, groups=a.shape[0])[0]
    mu1, mu2 = f(a), f(b)
    s11, s22, s12 = f(a * a) - mu1 * mu1, f(b * b) - mu2 * mu2, f(a * b) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))).mean()


class GSMapper:
    """the mapping half of hislam2/gs_backend_per_frame.py:GSBackEnd"""

    def __init__(self, config, fx, fy, cx, cy, downsample_ratio=2, device="cuda:0"):
        self.config, self.device, self.downsample_ratio = config, torch.device(device), downsample_ratio
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        tr = config["Training"]
        self.lambda_depth, self.lambda_normal, self.lambda_iso = tr["lambda_depth"], tr["lambda_normal"], tr["lambda_iso"]
        self.gaussian_th, self.size_threshold = tr["gaussian_th"], tr["size_threshold"]
        self.gaussian_extent = 6.0 * tr["gaussian_extent"]                # cameras_extent * gaussian_extent (:49,71)
        self.gaussians = GaussianMap(config["opt_params"], device)
        self.background = torch.zeros(3, device=self.device)
        self.viewpoints = {}
        self.iteration_count = 0
        # optimization() / pose_refine(): capture one iteration as a hipGraph and replay it.  Capturing costs tens of eager iterations
        # (collection, three eager iterations on the capture stream, instantiation), a replayed iteration about 40 % less than an eager one
        # (tools/bench_gs.py: 2.30 -> 1.41 ms per render iteration over 400 iterations): it pays for long loops over a fixed set of Gaussians (a final refinement), not for the 20-100 iteration calls of run()
        self.use_graphs = False
        self.graph_min_iters = 150
        self.graph_capacity = (1.25, 8192)       # tile-instance capacity of captured iterations: factor on the largest count seen + margin
        # optimization() / pose_refine() on the tape-free trainer of gs_step.py (direct C-ABI calls, ~4x fewer launches); False: the
        # tensor-op formulation below (the tests compare the two)
        self.fused = True
        self._trainer = None

    def _fused_trainer(self):
        if self._trainer is None:
            from .gs_step import FusedTrainer
            self._trainer = FusedTrainer(self)
        return self._trainer

    def _pose_optimizer(self, views, exposure=False, capturable=False):
        lr = self.config["opt_params"]["pose_lr"]
        groups = []
        for v in views:
            groups += [{"params": [v.cam_rot_delta], "lr": lr * 2}, {"params": [v.cam_trans_delta], "lr": lr * 10}]
            if exposure:                                                  # gs_backend_per_frame.py:467-475
                elr = self.config["opt_params"].get("exposure_lr", 0.0005)
                groups += [{"params": [v.exposure_a], "lr": elr}, {"params": [v.exposure_b], "lr": elr}]
        return torch.optim.Adam(groups, capturable=capturable)

    def _loop(self, one_iteration, iters, use_graph, opt):
        """run `one_iteration(it, eager)` iters times; with use_graph: three eager iterations, then ONE captured as a hipGraph and replayed.
        Returns the last loss (a device tensor)."""
        last = None
        if not use_graph:
            for it in range(iters):
                last = one_iteration(it, True)                            # (read back once, by the caller)
            return last
        import gc
        from . import gaussian_rasterizer as GR
        warm = 3
        GR.LAST_INSTANCES[0] = 0
        gc.collect()                                                      # no autograd graph of an earlier pass may survive into the capture
        # the eager iterations run on the stream the capture will use: autograd's gradient accumulators remember the stream of their
        # first use, and a backward pass inside the capture that has to synchronise with another stream ends the process
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for it in range(warm):
                last = one_iteration(it, True)
            cap = int(self.graph_capacity[0] * GR.LAST_INSTANCES[0]) + int(self.graph_capacity[1])
            flag = GR.overflow_flag(self.device)
            flag.zero_()
            GR.workspace_bytes(len(self.gaussians), 0), GR.workspace_bytes(len(self.gaussians), cap)      # (size queries outside the capture)
            self.gaussians.zero_grad()
            if opt is not None:
                opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        was = gc.isenabled()
        gc.disable()                                                      # (a collection during capture may free tensors of the graph's pool)
        try:
            with GR.fixed_capacity(cap), torch.cuda.graph(g, stream=side):
                static_loss = one_iteration(warm, False)
        finally:
            if was:
                gc.enable()
        # state at the start of the replays: if a replayed iteration overflows the capacity its tile lists were truncated, and NO such
        # iteration may be kept -- the replays are then undone and redone eagerly with exact instance counts
        gm = self.gaussians
        snap = (gm.theta.detach().clone(), gm.m.clone(), gm.v.clone(), gm.step_count.clone(), gm.steps - 1,      # (- 1: the capture pass's host increment)
                [(v.R.clone(), v.T.clone(), v.w2c_data.clone(), v.cam_rot_delta.detach().clone(), v.cam_trans_delta.detach().clone())
                 for v in self.viewpoints.values()],
                [[{k: (t.clone() if torch.is_tensor(t) else t) for k, t in opt.state[p].items()} for p in grp["params"]] for grp in opt.param_groups]
                if opt is not None else None)
        for _ in range(iters - warm):
            g.replay()
        gm.steps += iters - warm - 1                                      # (the capture pass advanced the host count once, every replay the device's)
        if int(flag):                                                     # one read after all replays
            import warnings
            warnings.warn(f"GSMapper: a captured iteration needed more than {cap} tile instances; the {iters - warm} replayed iterations are "
                          "redone eagerly from their starting state (and this mapper stops capturing)", RuntimeWarning)
            self.use_graphs = False
            flag.zero_()
            with torch.no_grad():
                gm.theta.data.copy_(snap[0]); gm.m.copy_(snap[1]); gm.v.copy_(snap[2]); gm.step_count.copy_(snap[3])
                gm.steps = snap[4]
                for v, (R, T, data, dr, dt) in zip(self.viewpoints.values(), snap[5]):
                    v.update_RT(R, T, data=data)
                    v.cam_rot_delta.data.copy_(dr); v.cam_trans_delta.data.copy_(dt)
                if opt is not None:
                    for grp, saved in zip(opt.param_groups, snap[6]):
                        for p, st in zip(grp["params"], saved):
                            for k, t in st.items():
                                if torch.is_tensor(t):
                                    opt.state[p][k].copy_(t)
            last = None
            with torch.cuda.stream(side):
                for it in range(warm, iters):
                    last = one_iteration(it, True)
            torch.cuda.current_stream().wait_stream(side)
            return last
        return static_loss

    def pose_refine(self, BA_window, iters=50, return_args=True, alpha_th=0.5, graph=None):
        """gs_backend_per_frame.py:202-326: the Gaussians stay fixed, the poses of the window move; photometric L1 on the covered pixels,
        scale-invariant log-depth variance, a small pull to the starting pose.  Returns (pointmaps_ds, valid_ds) of the refined poses."""
        views = [self.viewpoints[k] for k in BA_window]
        B = len(views)
        if len(self.gaussians) > 0 and self.fused and not graph and iters > 0:
            self._fused_trainer().pose_refine(views, iters, alpha_th)
        elif len(self.gaussians) > 0:
            use_graph = (graph if graph is not None else (self.use_graphs and iters >= self.graph_min_iters)) and iters >= 8
            opt = self._pose_optimizer(views, capturable=use_graph)

            def one_iteration(it, eager):
                rgb_all = depth_all = pose_all = 0.0
                for v in views:
                    pkg = render(v, self.gaussians, self.background)
                    image, depth, alpha = pkg["render"], pkg["depth"], pkg["mask"]
                    # :240-262 -- colour L1 over the covered pixels, variance of the log-depth difference, both weighted by the covered share
                    r_rgb, r_var, ratio = refine_losses(image, depth, v.original_image, v.depth, alpha, alpha_th)
                    pl = (v.cam_rot_delta ** 2).sum() + (v.cam_trans_delta ** 2).sum()
                    rgb_all, depth_all, pose_all = rgb_all + r_rgb, depth_all + r_var, pose_all + (2 - ratio) * pl
                loss = (5 * rgb_all + depth_all + 0.05 * pose_all) / B
                opt.zero_grad(set_to_none=True)
                self.gaussians.zero_grad()
                loss.backward()
                opt.step()
                return loss.detach()
            self._loop(one_iteration, iters, use_graph, opt)
            self.gaussians.zero_grad()
            for v in views:
                update_pose(v)
        if not return_args:
            return None
        with torch.no_grad():
            w2c_all, depth_all, valid_all = [], [], []
            for v in views:
                gt_depth = v.depth[None]
                if len(self.gaussians) > 0:
                    pkg = render(v, self.gaussians, self.background)
                    depth, alpha = pkg["depth"], pkg["mask"]
                else:
                    depth, alpha = torch.zeros_like(gt_depth), torch.zeros_like(gt_depth)
                valid = (alpha <= alpha_th) & (gt_depth > 0.001)
                amask = alpha > alpha_th
                dmask = (gt_depth > 0.001) & (depth > 0.001) & amask
                if float(amask.float().mean()) > 0.3 and dmask.any():
                    scale = torch.exp((torch.log(depth[dmask]) - torch.log(gt_depth[dmask])).mean()).clamp(0.95, 1.05)
                    gt_depth = scale * gt_depth
                w2c_all.append(get_pose(v))
                depth_all.append(gt_depth)
                valid_all.append(valid)
            pm = project2world(torch.inverse(torch.stack(w2c_all)), torch.cat(depth_all), self.fx, self.fy, self.cx, self.cy)
            ds = self.downsample_ratio
            return pm[:, ::ds, ::ds], torch.cat(valid_all).float()[:, ::ds, ::ds]

    def add_new_view(self, new_img, new_pose, new_depth, new_tstamp=None, kf_sub_idx=0, iters=50):
        """gs_backend_per_frame.py:87-121: register the keyframe, refine its pose against the map, add Gaussians where the map does not
        cover it yet.  new_img u8 or float [3,H,W]; new_pose [7] camera->world (t, q_xyzw); new_depth [H,W]."""
        img = new_img.to(self.device).float()
        img = img / 255.0 if img.max() > 1.5 else img
        w2c = torch.inverse(pose_vec_to_matrix(torch.as_tensor(new_pose)[None].to(self.device))[0])
        idx = len(self.viewpoints)
        self.viewpoints[idx] = Camera(idx, img, new_depth, w2c, self.fx, self.fy, self.cx, self.cy, new_tstamp, self.device)
        pointmap, valid = self.pose_refine([idx], iters=iters)
        ds = self.downsample_ratio
        rgb = img[:, ::ds, ::ds].permute(1, 2, 0)
        self.gaussians.extend_from_pcd_seq(submap_idx=kf_sub_idx, rgb=rgb, pointmap=pointmap[0], conf=valid[0])
        return idx

    def optimization(self, iters, optimize_pose=True, current_window=None, densify=False, graph=None):
        """gs_backend_per_frame.py:451-587: L1 + SSIM colour, inverse-depth L1, depth-normal agreement with the keyframe depth, isotropy;
        Gaussians (and optionally the window's poses) step together; clone / split / prune at iters/4 and iters/2 when densifying.
        graph (default `self.use_graphs`): when the set of Gaussians stays fixed (no densification) and there are enough iterations,
        ONE iteration -- render, losses, backward, both optimiser steps, pose update -- is captured as a hipGraph after three eager
        iterations and replayed: the ~300 small launches of an iteration stop costing host time.  The rasteriser runs in capacity mode
        inside the graph (no instance-count read); its overflow flag is checked once after the replays."""
        views = [self.viewpoints[k] for k in current_window]
        N = len(views)
        exposure = bool(self.config["Training"].get("compensate_exposure", False))
        if self.fused and not graph and not densify and not exposure and iters > 0 and len(self.gaussians) > 0 and N > 0:
            return self._fused_trainer().optimization(views, iters, optimize_pose)
        use_graph = ((graph if graph is not None else (self.use_graphs and iters >= self.graph_min_iters)) and not densify and iters >= 8
                     and len(self.gaussians) > 0)
        opt = self._pose_optimizer(views, exposure, capturable=use_graph) if optimize_pose else None

        def one_iteration(it, eager):
            loss = 0.0
            stats = []
            for v in views:
                pkg = render(v, self.gaussians, self.background)
                image, depth = pkg["render"], pkg["depth"]
                if exposure:                                              # :513: colours through the view's affine exposure model
                    image = (image.permute(1, 2, 0) @ v.exposure_a + v.exposure_b).permute(2, 0, 1).contiguous()
                gt_image, gt_depth = v.original_image, v.depth[None]
                gcache = getattr(v, "_gt_normal", None)
                if gcache is None or gcache[0] is not v.depth:               # the keyframe's own depth normals change only with its depth
                    gcache = v._gt_normal = (v.depth, depth_to_normal(v, gt_depth).detach().contiguous())
                # :516-531 -- 0.8 L1 + 0.2 (1 - SSIM) colour, lambda_depth inverse-depth L1, lambda_normal depth-normal agreement: the three
                # per-pixel terms in one fused kernel each way, SSIM in its own
                pix = pixel_losses(image, depth, gt_image, v.depth, gcache[1], (v.fx, v.fy, v.cx, v.cy), 0.8, self.lambda_depth, self.lambda_normal)
                rgb_dl_nl = pix + 0.2 * (1.0 - ssim(image, gt_image))
                vis = pkg["visibility_filter"]
                sc = self.gaussians.get_scaling
                iso = (torch.abs(sc - sc.mean(dim=1, keepdim=True)) * vis[:, None]).sum() / (3 * vis.sum()).clamp_min(1)
                loss = loss + rgb_dl_nl + self.lambda_iso * iso
                stats.append((pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]))
            loss = loss / N
            self.gaussians.zero_grad()
            if opt is not None:
                opt.zero_grad(set_to_none=True)
            loss.backward()
            with torch.no_grad():
                if densify and eager:
                    for vs, vis, radii in stats:
                        self.gaussians.max_radii2D = torch.max(self.gaussians.max_radii2D, radii.float() * vis)
                        self.gaussians.add_densification_stats(vs.grad, vis)
                dens = bool(densify and eager and it in (iters // 4, iters // 2))
                if dens:
                    self.gaussians.densify_and_prune(self.config["opt_params"]["densify_grad_threshold"], self.gaussian_th, self.gaussian_extent,
                                                     self.size_threshold)
                self.gaussians.step_like_reference(densified=dens)
            if opt is not None:
                opt.step()
                for v in views:
                    update_pose(v)
            return loss.detach()

        last = self._loop(one_iteration, iters, use_graph, opt)
        self.gaussians.zero_grad()
        return float(last) if last is not None else None

    def global_BA(self, iteration_total, densify=True, densify_every=None, opacity_reset=True, seed=0):
        """gs_backend_per_frame.py:946-1058: Gaussians and ALL keyframe poses (and exposures) together, one randomly drawn keyframe per
        iteration; colour, inverse-depth (only with densify_every), agreement of the rendered normal with the normal of the rendered
        depth, and of that with the keyframe depth's normal; clone / split / prune at half time (densify_every) or every
        Training.gaussian_update_every iterations, opacity reset every Training.gaussian_reset iterations, position learning-rate decay."""
        import random
        views = list(self.viewpoints.values())
        if not views or len(self.gaussians) == 0:
            return None
        exposure = bool(self.config["Training"].get("compensate_exposure", False))
        if self.fused and not exposure and iteration_total > 0:
            return self._fused_trainer().global_BA(iteration_total, densify, densify_every, opacity_reset, seed)
        opt = self._pose_optimizer(views, exposure)
        rng = random.Random(seed)
        tr, op = self.config["Training"], self.config["opt_params"]
        update_every, reset_every = tr.get("gaussian_update_every", 200), tr.get("gaussian_reset", 3001)
        last = None
        for iteration in range(iteration_total):
            self.iteration_count += 1
            v = views[rng.randint(0, len(views) - 1)]
            pkg = render(v, self.gaussians, self.background)
            image, depth = pkg["render"], pkg["depth"]
            if exposure:
                image = (image.permute(1, 2, 0) @ v.exposure_a + v.exposure_b).permute(2, 0, 1).contiguous()
            gt_image, gt_depth = v.original_image, v.depth[None]
            gcache = getattr(v, "_gt_normal", None)
            if gcache is None or gcache[0] is not v.depth:
                gcache = v._gt_normal = (v.depth, depth_to_normal(v, gt_depth).detach().contiguous())
            nl = (1 - (pkg["normal"] * depth_to_normal(v, depth)).sum(0)).mean()          # rendered normal vs normal of the rendered depth
            # :1003-1006: with densify_every  rgb + lambda_depth/10 depth + lambda_normal (nl + gt normal);  else  rgb + lambda_normal/2 (nl + gt normal)
            w_d, w_n = (self.lambda_depth / 10, self.lambda_normal) if densify_every is not None else (0.0, self.lambda_normal / 2)
            loss = (pixel_losses(image, depth, gt_image, v.depth, gcache[1], (v.fx, v.fy, v.cx, v.cy), 0.8, w_d, w_n)
                    + 0.2 * (1.0 - ssim(image, gt_image)) + w_n * nl)
            self.gaussians.zero_grad()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            with torch.no_grad():
                if iteration < 10000 and densify:
                    vis = pkg["visibility_filter"]
                    self.gaussians.max_radii2D = torch.max(self.gaussians.max_radii2D, pkg["radii"].float() * vis)
                    self.gaussians.add_densification_stats(pkg["viewspace_points"].grad, vis)
                do_densify = do_reset = False
                if iteration < 10000 and densify:
                    do_densify = (iteration == iteration_total // 2) if densify_every is not None else ((self.iteration_count + 1) % update_every == 0)
                    if do_densify:
                        self.gaussians.densify_and_prune(op["densify_grad_threshold"], self.gaussian_th, self.gaussian_extent, self.size_threshold)
                    do_reset = bool((self.iteration_count + 1) % reset_every == 0 and opacity_reset)
                    if do_reset:
                        self.gaussians.reset_opacity()
                self.gaussians.step_like_reference(densified=do_densify, reset=do_reset)
                if densify and "position_lr_final" in op:
                    # gs_backend_per_frame.py:1043-1044 calls update_learning_rate(iteration) on EVERY iteration of a densifying run (only
                    # the statistics / densify / reset above stop at 10000); gaussian_model.py:419-431, general_utils.py:41-56
                    self.gaussians.lr[0, 0:3] = position_lr(op, iteration)
            opt.step()
            update_pose(v)
            last = loss.detach()
        self.gaussians.zero_grad()
        return float(last) if last is not None else None

    def reset(self):
        """gs_backend_per_frame.py:79-85: forget the map (the keyframe views stay)"""
        self.iteration_count = 0
        self.current_window, self.initialized = [], False
        self.gaussians = GaussianMap(self.config["opt_params"], self.device)

    def gaussian_reinit(self, rgbs, pointmaps, iteration_total=3000, seed=0):
        """gs_backend_per_frame.py:865-944 (Hi2.terminate(gaussian_retrain=True)): a fresh map from every keyframe's stride-2 pointmap
        (rgbs u8 [n,3,H,W], pointmaps [n,h,w,3] world), then `iteration_total` single-view iterations with fixed poses -- colour, inverse
        depth and the depth-normal term -- densifying every Training.gaussian_update_every iterations after the first 1000"""
        import random
        self.reset()
        self.initialized = True
        pm = torch.as_tensor(pointmaps, dtype=torch.float32, device=self.device)
        h1, w1 = pm.shape[1:3]
        rgb = F.interpolate(torch.as_tensor(rgbs).to(self.device).float() / 255.0, size=(h1, w1), mode="bilinear", align_corners=False)
        self.gaussians.extend_from_pcd_seq(submap_idx=0, rgb=rgb[..., ::2, ::2].permute(0, 2, 3, 1), pointmap=pm[:, ::2, ::2])
        views = list(self.viewpoints.values())
        if not views or len(self.gaussians) == 0:
            return None
        if self.fused and iteration_total > 0:
            return self._fused_trainer().reinit_loop(iteration_total, seed)
        rng = random.Random(seed)
        update_every = self.config["Training"].get("gaussian_update_every", 200)
        last = None
        for iteration in range(iteration_total):
            v = views[rng.randint(0, len(views) - 1)]
            pkg = render(v, self.gaussians, self.background)
            image, depth = pkg["render"], pkg["depth"]
            gcache = getattr(v, "_gt_normal", None)
            if gcache is None or gcache[0] is not v.depth:
                gcache = v._gt_normal = (v.depth, depth_to_normal(v, v.depth[None]).detach().contiguous())
            # (the reference averages the normal term over all pixels here, not over the depth mask: where either depth is missing the
            #  term is the constant 1 -- same gradient as the masked mean up to the normaliser; kept masked, as in optimization())
            loss = (pixel_losses(image, depth, v.original_image, v.depth, gcache[1], (v.fx, v.fy, v.cx, v.cy), 0.8, self.lambda_depth,
                                 self.lambda_normal) + 0.2 * (1.0 - ssim(image, v.original_image)))
            self.gaussians.zero_grad()
            loss.backward()
            with torch.no_grad():
                if iteration > 1000:
                    vis = pkg["visibility_filter"]
                    self.gaussians.max_radii2D = torch.max(self.gaussians.max_radii2D, pkg["radii"].float() * vis)
                    self.gaussians.add_densification_stats(pkg["viewspace_points"].grad, vis)
                dens = iteration > 1000 and (iteration + 1) % update_every == 0
                if dens:
                    self.gaussians.densify_and_prune(self.config["opt_params"]["densify_grad_threshold"], self.gaussian_th, self.gaussian_extent,
                                                     self.size_threshold)
                self.gaussians.step_like_reference(densified=dens)
            last = loss.detach()
        self.gaussians.zero_grad()
        return float(last) if last is not None else None

    def finalize(self, iteration_total=None, path=None):
        """gs_backend_per_frame.py:1067-1086: the closing global BA (the reference runs `max_steps` = position_lr_max_steps iterations),
        an optional checkpoint, and the keyframe poses camera->world [n,7] (t, q_xyzw) in keyframe order (the views added by
        terminate() come after the tracker's keyframes, as in the reference's dict order)"""
        self.iteration_count = 0
        if iteration_total is None:
            iteration_total = int(self.config["opt_params"].get("position_lr_max_steps", 20000))
        if iteration_total > 0:
            self.global_BA(iteration_total=iteration_total)
        if path is not None:
            self.save(path)
        with torch.no_grad():
            return torch.stack([SE3_from_matrix(torch.inverse(get_pose(v))) for v in self.viewpoints.values()])

    def global_pose_refine(self, iters=5):
        """gs_backend_per_frame.py:1060-1062"""
        last = None
        for _ in range(iters):
            last = self.global_BA(iteration_total=5 * len(self.viewpoints), densify=True, opacity_reset=False)
        return last

    # ---- the tracker-facing entry points (hi2.py:56-99 calls run() once per tracked window)
    def run(self, packet, iterations=100, init_iters=100, gba_per_view=10):
        """gs_backend_per_frame.py:776-862.  packet: viz_idx (keyframe indices of the window), submap_idx, tstamp [n], poses [n,7]
        camera->world, images [n,3,H,W] u8, pointmaps [n,h,w,3] world, confs [n,h,w], depths [n,h',w'], intrinsics [4].  New keyframes
        are chained to the mapper's own estimate of the previous one, refined against the map, mapped; then a global pass.  Returns
        (updated packet, keyframe indices) as data_update()."""
        with torch.no_grad():
            H, W = packet["images"].shape[-2:]
            self.fx, self.fy, self.cx, self.cy = (float(v) for v in list(packet["intrinsics"])[:4])
            viz_idx, submap_idx = list(packet["viz_idx"]), int(packet["submap_idx"])
            imgs = packet["images"].to(self.device).float() / 255.0
            ds = self.downsample_ratio
            pointmaps = F.interpolate(packet["pointmaps"].to(self.device).permute(0, 3, 1, 2), size=(H // ds, W // ds), mode="bilinear",
                                      align_corners=False).permute(0, 2, 3, 1)
            self.h, self.w = packet["pointmaps"].shape[1:3]
            depths = F.interpolate(packet["depths"].to(self.device)[None], size=(H, W), mode="bilinear", align_corners=False)[0].clone()
            confs = F.interpolate(packet["confs"].to(self.device)[None], size=(H, W), mode="bilinear", align_corners=False)[0]
            depths[confs < 0.0] = 0.0
            w2c = torch.inverse(pose_vec_to_matrix(packet["poses"].to(self.device)))
        if not hasattr(self, "current_window"):
            self.current_window, self.initialized = [], False
        window_size = self.config["Training"].get("window_size", 10)
        for i, idx in enumerate(viz_idx):
            cur = w2c[i]
            if i > 0 and viz_idx[i - 1] in self.viewpoints:
                with torch.no_grad():
                    cur = (cur @ torch.inverse(w2c[i - 1])) @ get_pose(self.viewpoints[viz_idx[i - 1]]).detach()
            if idx in self.viewpoints:
                continue
            self.viewpoints[idx] = Camera(idx, imgs[i], depths[i], cur, self.fx, self.fy, self.cx, self.cy, float(packet["tstamp"][i]), self.device)
            if not self.initialized:
                self.gaussians.extend_from_pcd_seq(submap_idx=0, rgb=imgs[i, :, ::ds, ::ds].permute(1, 2, 0), pointmap=pointmaps[i])
                self.current_window = [idx]
                self.optimization(init_iters, current_window=self.current_window)
                self.initialized = True
            else:
                self.current_window = (self.current_window + [idx])[-window_size:]
                pm, valid = self.pose_refine([idx], iters=50)
                self.gaussians.extend_from_pcd_seq(submap_idx=submap_idx, rgb=imgs[i, :, ::ds, ::ds].permute(1, 2, 0), pointmap=pm[0], conf=valid[0])
                self.optimization(min(20, iterations), current_window=self.current_window)
                self.optimization(min(50, iterations), current_window=[idx], optimize_pose=False)
        gba_iters = gba_per_view * len(self.viewpoints)
        self.global_BA(iteration_total=gba_iters, densify=True, densify_every=gba_iters // 2, opacity_reset=False)      # :860-861
        return self.data_update(self.h, self.w, self.current_window)

    def gaussain_update(self, packet, reference_quat_order=False, refine_iters=50):
        """gs_backend_per_frame.py:701-774 (the reference's spelling): after a loop closure the tracker hands over the corrected poses of
        the affected keyframes (`camera_idx`, `camera_pose` [n,7] camera->world) and one SE3 correction per submap (`submap_idx`,
        `pose_updates` [m,7]); every Gaussian born in such a submap moves with its correction (position and orientation, optimiser state
        reset as the reference's prune + re-append does), each updated keyframe is re-refined against the moved map, and the window data
        goes back as in data_update().
        Orientation: the reference multiplies `SO3(update[:, 3:]) * SO3(get_rotation)`, i.e. it reads the Gaussians' (r, x, y, z)
        quaternions as lietorch's (x, y, z, w).  That is exact only for isotropic Gaussians; the default here composes the rotations in
        one convention (a rigid correction of the whole scene then leaves every rendering unchanged, which the test checks);
        reference_quat_order=True reproduces the reference's arithmetic."""
        with torch.no_grad():
            w2cs = torch.inverse(pose_vec_to_matrix(torch.as_tensor(packet["camera_pose"]).float().to(self.device)))
            update_idx = []
            for i, k in enumerate(packet["camera_idx"]):
                if k in self.viewpoints and i < w2cs.shape[0]:
                    update_idx.append(k)
                    self.viewpoints[k].update_RT(w2cs[i, :3, :3], w2cs[i, :3, 3])
            sub = torch.as_tensor(list(packet["submap_idx"]), device=self.device, dtype=torch.int32)
            upd = torch.as_tensor(packet["pose_updates"]).float().to(self.device)
            hit = self.gaussians.kf_id[:, None] == sub[None, :]
            gi = hit.any(dim=1).nonzero()[:, 0]
            if gi.numel():
                row = hit[gi].float().argmax(dim=1)
                T = SE3(upd[row]).matrix()
                gm = self.gaussians
                xyz = (T[:, :3, :3] @ gm.p["xyz"].detach()[gi][:, :, None])[:, :, 0] + T[:, :3, 3]
                rot = gm.get_rotation.detach()[gi]
                q_u = upd[row][:, 3:]                                        # (x, y, z, w)
                if reference_quat_order:
                    new_rot = _quat_mult_xyzw(q_u, rot)
                else:
                    new_rot = _quat_mult(torch.cat([q_u[:, 3:], q_u[:, :3]], -1), rot)
                gm.theta.data[gi, 0:3] = xyz
                gm.theta.data[gi, 10:14] = new_rot
                gm.reset_moments(gi)
                gm.reset_densification_stats()
        for k in update_idx:
            if refine_iters > 0:
                self.pose_refine([k], iters=refine_iters, return_args=False, alpha_th=0.0)
        return self.data_update(self.h, self.w, update_idx)

    @torch.no_grad()
    def data_update(self, h, w, current_window):
        """gs_backend_per_frame.py:649-699: poses (camera->world, t + q_xyzw), scale-corrected depths and their world pointmaps of the
        window's keyframes, resampled to (h, w) * downsample_ratio"""
        poses, depths, pms = [], [], []
        if not len(current_window):
            e = torch.zeros(0, device=self.device)
            return {"pointmaps": e.reshape(0, 0, 0, 3), "depths": e.reshape(0, 0, 0), "poses": e.reshape(0, 7)}, []
        for k in current_window:
            v = self.viewpoints[k]
            gt = v.depth[None]
            pkg = render(v, self.gaussians, self.background)
            ok = (pkg["depth"] > 0.001) & (gt > 0.001) & (pkg["mask"] > 0.9)
            if ok.any():
                gt = gt * torch.exp((torch.log(pkg["depth"][ok]) - torch.log(gt[ok])).mean()).clamp(0.95, 1.05)
            v.depth = gt[0]
            c2w = torch.inverse(get_pose(v))
            pms.append(project2world(c2w[None], gt, self.fx, self.fy, self.cx, self.cy)[0])
            depths.append(gt[0])
            poses.append(SE3_from_matrix(c2w))
        ds = self.downsample_ratio
        size = (h * ds, w * ds)
        pms = F.interpolate(torch.stack(pms).permute(0, 3, 1, 2), size=size, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        depths = F.interpolate(torch.stack(depths)[None], size=size, mode="bilinear", align_corners=False)[0]
        return {"pointmaps": pms, "depths": depths, "poses": torch.stack(poses)}, list(current_window)


    # ---- checkpoint / evaluation (gs_backend_per_frame.py:1088-1096; gaussian/utils/eval_utils.py:110-150 without LPIPS, whose
    #      pretrained network is not available offline)
    def save(self, path):
        """the Gaussian map (parameters, optimiser moments, bookkeeping) as a safetensors file: nothing executable in the file"""
        from safetensors.torch import save_file
        g = self.gaussians
        if g._steps_dev_stale:
            g.step_count.fill_(float(g.steps))
            g._steps_dev_stale = False
        save_file({"theta": g.theta.detach().contiguous(), "m": g.m.contiguous(), "v": g.v.contiguous(), "step_count": g.step_count.contiguous(),
                   "kf_id": g.kf_id.contiguous(), "max_radii2D": g.max_radii2D.contiguous(), "grad_accum": g.grad_accum.contiguous(),
                   "grad_accum_abs": g.grad_accum_abs.contiguous(), "denom": g.denom.contiguous()}, path)

    def load(self, path):
        from safetensors.torch import load_file
        t = load_file(path, device=str(self.device))
        g = self.gaussians
        g.theta = t["theta"].requires_grad_(True)
        g.m, g.v, g.kf_id = t["m"], t["v"], t["kf_id"]
        # one Adam step count per block (a device scalar).  Older files carry one count per Gaussian ([P,1]) and an empty map carries none:
        # take the largest count (0 for an empty map) and keep the one-element form the captured iteration increments in place
        sc = t["step_count"].reshape(-1)
        g.steps = int(sc.max()) if sc.numel() > 0 else 0
        g.step_count = torch.full((1,), float(g.steps), dtype=g.m.dtype if g.m.is_floating_point() else torch.float32, device=g.m.device)
        g._steps_dev_stale = False
        g.max_radii2D, g.grad_accum, g.denom = t["max_radii2D"], t["grad_accum"], t["denom"]
        g.grad_accum_abs = t["grad_accum_abs"] if "grad_accum_abs" in t else torch.zeros_like(g.grad_accum)

    @torch.no_grad()
    def eval_rendering_kf(self):
        """eval_utils.py:110-150 over the mapper's own keyframes: PSNR on the pixels with a ground-truth colour (gt > 0) and SSIM per
        view -> dict(mean_psnr, mean_ssim, per_view)"""
        rows = []
        for k in sorted(self.viewpoints):
            v = self.viewpoints[k]
            img = torch.clamp(render(v, self.gaussians, self.background)["render"], 0.0, 1.0)
            gt = v.original_image
            mask = gt > 0
            mse = ((img[mask] - gt[mask]) ** 2).mean() if mask.any() else torch.zeros((), device=self.device)
            rows.append((k, float(20 * torch.log10(1.0 / torch.sqrt(mse.clamp_min(1e-12)))), float(ssim(img, gt))))
        n = max(1, len(rows))
        return {"mean_psnr": sum(r[1] for r in rows) / n, "mean_ssim": sum(r[2] for r in rows) / n, "per_view": rows}

    @torch.no_grad()
    def trajectory(self):
        """camera->world [n,4,4] of the keyframes as refined by the mapper"""
        return torch.stack([torch.inverse(get_pose(self.viewpoints[k])) for k in sorted(self.viewpoints)])


def SE3_from_matrix(T):
    """[4,4] rigid transform -> [7] (t, q_xyzw), quaternion by the largest-diagonal rule (scipy Rotation.from_matrix / as_quat order)"""
    q = _rotmat_to_quat(T[:3, :3])
    q = q / q.norm()
    return torch.cat([T[:3, 3], q[1:], q[:1]])
