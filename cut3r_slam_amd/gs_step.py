"""The Gaussian mapper's optimisation loops without an autograd tape: `GSMapper.optimization` and `GSMapper.pose_refine`
(/root/reference/hislam2/gs_backend_per_frame.py:451-587, :202-326) as direct calls into the C ABI -- about 25 launches per rendered view
instead of the ~100 launches + autograd bookkeeping of the tensor-op formulation in gs_mapper.py, which stays as the second
implementation the tests compare this one with (same losses, same Adam arithmetic, same pose update).

Per view and iteration:  cut3r_gs_activate -> cut3r_gs_preprocess / _bin / _render_forward -> loss kernels (+ cut3r_gs_map_coef /
_refine_coef for the scalar algebra between their two passes) -> cut3r_gs_render_backward / _preprocess_backward ->
cut3r_gs_activate_backward (gradient of theta accumulated over the views, 16 pose sums per view) -> cut3r_gs_pose_step; then ONE
cut3r_gs_adam over all Gaussian parameters.

The rasteriser's one host read per pass (the instance count) is taken in the FIRST iteration of a call only: later iterations run in
capacity mode (1.5 x the largest count seen + a margin, overflow flagged on the device).  The flag is read once at the end of the call; if
it is set, parameters, moments and poses are restored from the snapshot taken at the start and the whole call is redone with exact
counts -- no truncated iteration is ever kept.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from . import gaussian_rasterizer as GR
from ._lib import check

PS = 32            # floats of pose state per view (include/cut3r_hip.h)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _arr(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


_IDENTITY16 = _arr([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1])
_ZERO3 = _arr([0, 0, 0])


class FusedTrainer:
    """workspaces + the two loops; one instance per GSMapper (buffers are re-sized when the number of Gaussians or the image size changes)"""

    def __init__(self, mapper):
        self.mp = mapper
        self.dev = mapper.device
        self.lib = _lib.load()
        self._shape = None
        self._cap = 0
        self.capacity = (1.5, 16384)             # capacity-mode iterations: factor on the largest instance count of iteration 0 + margin
        self.redone = 0                          # calls that overflowed their capacity and were redone from the snapshot

    # ------------------------------------------------------------------ buffers
    def _buffers(self, P, H, W):
        if self._shape == (P, H, W):
            return
        dev = self.dev
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        i = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
        self.means, self.scales, self.rots, self.opac, self.shs = f(P, 3), f(P, 3), f(P, 4), f(P), f(P, 1, 3)
        self.geom, self.dgeom = f(P, GR.GS_REC), f(P, GR.GS_REC)
        self.radii, self.tiles, self.offsets = i(P), i(P), i(P)
        self.scan_ws = torch.empty(GR.workspace_bytes(P, 0), dtype=torch.uint8, device=dev)
        self.flat = f(P * 14 + P * 3)                  # d_means | d_scales | d_rots | d_opac | d_means2D | d_shs: zeroed together
        self.d_means, self.d_scales = self.flat[0:3 * P].view(P, 3), self.flat[3 * P:6 * P].view(P, 3)
        self.d_rots, self.d_opac = self.flat[6 * P:10 * P].view(P, 4), self.flat[10 * P:11 * P]
        self.d_means2D, self.d_shs = self.flat[11 * P:14 * P].view(P, 3), self.flat[14 * P:17 * P].view(P, 1, 3)
        self.gtheta = f(P, 14)
        if self._shape is None or self._shape[1:] != (H, W):
            self.img = {k: f(c, H, W) for k, c in (("color", 3), ("coord", 3), ("mcoord", 3), ("depth", 1), ("mdepth", 1), ("alpha", 1), ("normal", 3))}
            self.n_contrib, self.aux = i(2, H, W), f(2, H, W)
            self.smap, self.sd1, self.sd2, self.sd3 = f(3, H, W), f(3, H, W), f(3, H, W), f(3, H, W)
            self.g_img, self.g_ssim, self.g_depth = f(3, H, W), f(3, H, W), f(1, H, W)
            self.zero_img = torch.zeros(3, H, W, dtype=torch.float32, device=dev)
            gx, gy = (W + 15) // 16, (H + 15) // 16
            self.ranges = i(gy * gx, 2)
            self.sums, self.coef, self.ratio = f(8), f(4), f(1)
            self.nvis, self.loss_acc, self.ssim_scale = f(1), torch.zeros(1, device=dev), f(1)
        self._shape = (P, H, W)
        self._cap = 0

    def _sort_buffers(self, n):
        if n <= self._cap:
            return
        dev = self.dev
        self._cap = int(n)
        self.keys_tmp = torch.empty(self._cap, dtype=torch.int64, device=dev)
        self.keys_sorted = torch.empty(self._cap, dtype=torch.int64, device=dev)
        self.vals_tmp = torch.empty(self._cap, dtype=torch.int32, device=dev)
        self.point_list = torch.empty(self._cap, dtype=torch.int32, device=dev)
        self.sort_ws = torch.empty(GR.workspace_bytes(self._shape[0], self._cap), dtype=torch.uint8, device=dev)

    @staticmethod
    def _image_size(views):
        """(H, W) shared by the views; the image buffers and the kernels' raw pointers assume ONE size and dense fp32 observations"""
        H, W = int(views[0].image_height), int(views[0].image_width)
        for v in views:
            if (int(v.image_height), int(v.image_width)) != (H, W):
                raise ValueError(f"tape-free GS trainer: views of different sizes ({v.image_height}x{v.image_width} vs {H}x{W})")
            if (tuple(v.original_image.shape) != (3, H, W) or tuple(v.depth.shape) != (H, W) or not v.original_image.is_contiguous()
                    or not v.depth.is_contiguous() or v.original_image.dtype != torch.float32 or v.depth.dtype != torch.float32):
                raise ValueError("tape-free GS trainer: a view's image / depth must be dense fp32 [3,H,W] / [H,W]")
        return H, W

    # ------------------------------------------------------------------ per-view pieces
    @staticmethod
    def _cam(v):
        c = getattr(v, "_fused_cam", None)
        if c is None:
            c = v._fused_cam = {"proj": _arr(v.projection_matrix_host.reshape(-1).tolist()), "tanx": math.tan(v.FoVx * 0.5),
                                "tany": math.tan(v.FoVy * 0.5), "K": (float(v.fx), float(v.fy), float(v.cx), float(v.cy))}
        return c

    def _render(self, v, ps_v, exact):
        """activations + rasteriser forward of one view into the shared buffers; returns the instance count (exact mode) or None"""
        lib, P = self.lib, self._shape[0]
        H, W = self._shape[1:]
        cam = self._cam(v)
        theta = self.mp.gaussians.theta
        check(lib.cut3r_gs_activate(P, _p(theta), _p(ps_v), _p(self.means), _p(self.scales), _p(self.rots), _p(self.opac), _p(self.shs), _s()),
              "gs_activate")
        check(lib.cut3r_gs_preprocess(P, _p(self.means), _p(self.scales), _p(self.rots), _p(self.opac), _p(self.shs), 0, 1, None, _IDENTITY16,
                                      cam["proj"], _ZERO3, W, H, cam["tanx"], cam["tany"], 0.0, 1.0, _p(self.geom), _p(self.radii), _p(self.tiles),
                                      _p(self.offsets), _p(self.scan_ws), self.scan_ws.numel(), _s()), "gs_preprocess")
        n_read = None
        if exact:
            n_read = int(self.offsets[-1].item()) & 0xffffffff
            if n_read > GR.MAX_INSTANCES:
                raise RuntimeError(f"GaussianRasterizer: {n_read} Gaussian/tile instances (limit {GR.MAX_INSTANCES}): degenerate scales or a diverged map")
            self._sort_buffers(max(1, n_read))
            n_inst, overflow = n_read, None
        else:
            n_inst, overflow = self._n_cap, GR.overflow_flag(self.dev)
        check(lib.cut3r_gs_bin(P, _p(self.geom), _p(self.offsets), n_inst, W, H, _p(self.keys_tmp), _p(self.vals_tmp), _p(self.keys_sorted),
                               _p(self.point_list), _p(self.ranges), _p(self.sort_ws), self.sort_ws.numel(), _p(overflow), _s()), "gs_bin")
        im = self.img
        check(lib.cut3r_gs_render_forward(_p(self.ranges), _p(self.point_list), _p(self.geom), W, H, cam["tanx"], cam["tany"], _ZERO3, _p(im["color"]),
                                          _p(im["coord"]), _p(im["mcoord"]), _p(im["depth"]), _p(im["mdepth"]), _p(im["alpha"]), _p(im["normal"]),
                                          _p(self.n_contrib), _p(self.aux), _s()), "gs_render_forward")
        return n_read

    def _backward(self, v, ps_v, g_color, g_depth, iso_coef, gtheta, sums_v, g_normal=None):
        """rasteriser backward + activation chain of one view (its forward buffers are still the current ones)"""
        lib, P = self.lib, self._shape[0]
        H, W = self._shape[1:]
        cam, im, z = self._cam(v), self.img, self.zero_img
        self.flat.zero_()
        check(lib.cut3r_gs_render_backward(_p(self.ranges), _p(self.point_list), _p(self.geom), P, W, H, cam["tanx"], cam["tany"], _ZERO3,
                                           _p(self.n_contrib), _p(self.aux), _p(im["alpha"]), _p(im["coord"]), _p(im["depth"]), _p(im["normal"]),
                                           _p(g_color), _p(z), _p(z), _p(g_depth), _p(z), _p(z), _p(g_normal if g_normal is not None else z), _p(self.dgeom),
                                           _s()), "gs_render_backward")
        check(lib.cut3r_gs_preprocess_backward(P, _p(self.means), _p(self.scales), _p(self.rots), _p(self.opac), _p(self.shs), 0, 1, _IDENTITY16,
                                               cam["proj"], _ZERO3, W, H, cam["tanx"], cam["tany"], 0.0, 1.0, _p(self.geom), _p(self.dgeom),
                                               _p(self.d_means), _p(self.d_scales), _p(self.d_rots), _p(self.d_opac), _p(self.d_shs), None,
                                               _p(self.d_means2D), _s()), "gs_preprocess_backward")
        check(lib.cut3r_gs_activate_backward(P, _p(self.mp.gaussians.theta), _p(ps_v), _p(self.d_means), _p(self.d_scales), _p(self.d_rots),
                                             _p(self.d_opac), _p(self.d_shs), _p(self.radii), float(iso_coef), _p(self.nvis), _p(gtheta), _p(sums_v),
                                             _s()), "gs_activate_backward")

    # ------------------------------------------------------------------ pose state <-> cameras
    def _load_poses(self, views):
        ps = torch.zeros(len(views), PS, dtype=torch.float32, device=self.dev)
        for k, v in enumerate(views):
            ps[k, 0:7] = v.w2c_data
            ps[k, 7:10] = v.cam_trans_delta.detach()
            ps[k, 10:13] = v.cam_rot_delta.detach()
        return ps

    def _store_poses(self, views, ps):
        from .lietorch import SE3
        M = SE3(ps[:, 0:7].contiguous()).matrix()
        for k, v in enumerate(views):
            v.update_RT(M[k, :3, :3], M[k, :3, 3], data=ps[k, 0:7])
            v.cam_trans_delta.data.copy_(ps[k, 7:10])
            v.cam_rot_delta.data.copy_(ps[k, 10:13])

    # ------------------------------------------------------------------ the two loops
    def _run(self, body, iters, views, snapshot):
        """iteration 0 with exact instance counts, the others in capacity mode; one overflow read at the end, redo from the snapshot if set"""
        flag = GR.overflow_flag(self.dev)
        flag.zero_()
        counts = body(0, True)
        self._n_cap = int(self.capacity[0] * max(counts)) + int(self.capacity[1])
        self._sort_buffers(self._n_cap)
        for it in range(1, iters):
            body(it, False)
        if iters > 1 and int(flag.item()):
            flag.zero_()
            self.redone += 1
            snapshot(restore=True)
            for it in range(iters):
                body(it, True)

    def optimization(self, views, iters, optimize_pose=True):
        """GSMapper.optimization without densification / exposure compensation; returns the loss of the last iteration (float)"""
        mp, gm, lib = self.mp, self.mp.gaussians, self.lib
        P = len(gm)
        H, W = self._image_size(views)
        self._buffers(P, H, W)
        N = len(views)
        g = 1.0 / N
        ps = self._load_poses(views)
        sums = torch.zeros(N, 16, dtype=torch.float32, device=self.dev) if optimize_pose else None
        lr = mp.config["opt_params"]["pose_lr"]
        self.ssim_scale.fill_(-0.2 * g / (3 * H * W))
        saved = {}

        def snapshot(restore=False):
            if restore:
                gm.theta.data.copy_(saved["theta"]); gm.m.copy_(saved["m"]); gm.v.copy_(saved["v"]); ps.copy_(saved["ps"])
                gm.steps = saved["steps"]
            else:
                saved.update(theta=gm.theta.detach().clone(), m=gm.m.clone(), v=gm.v.clone(), ps=ps.clone(), steps=gm.steps)
        snapshot()
        for v in views:                                   # the keyframe's own depth normals (constant while its depth stays)
            gc = getattr(v, "_gt_normal", None)
            if gc is None or gc[0] is not v.depth:
                from .gs_mapper import depth_to_normal
                v._gt_normal = (v.depth, depth_to_normal(v, v.depth[None]).detach().contiguous())
        last = [None]

        def body(it, exact):
            final = it == iters - 1
            self.gtheta.zero_()
            if sums is not None:
                sums.zero_()
            if final:
                self.loss_acc.zero_()
            counts = []
            extra = 0.0
            for k, v in enumerate(views):
                cam, im = self._cam(v), self.img
                counts.append(self._render(v, ps[k], exact))
                K = cam["K"]
                check(lib.cut3r_pixel_loss_forward(_p(im["color"]), _p(v.original_image), _p(im["depth"]), _p(v.depth), _p(v._gt_normal[1]), H, W,
                                                   K[0], K[1], K[2], K[3], _p(self.sums), _s()), "pixel_loss_forward")
                check(lib.cut3r_gs_map_coef(_p(self.sums), 0.8, float(mp.lambda_depth), float(mp.lambda_normal), g, H, W, _p(self.coef),
                                            _p(self.loss_acc) if final else None, _s()), "gs_map_coef")
                check(lib.cut3r_pixel_loss_backward(_p(im["color"]), _p(v.original_image), _p(im["depth"]), _p(v.depth), _p(v._gt_normal[1]), H, W,
                                                    K[0], K[1], K[2], K[3], _p(self.coef), _p(self.g_img), _p(self.g_depth), _s()), "pixel_loss_backward")
                check(lib.cut3r_ssim_forward(_p(im["color"]), _p(v.original_image), 3, H, W, _p(self.smap), _p(self.sd1), _p(self.sd2), _p(self.sd3),
                                             _s()), "ssim_forward")
                check(lib.cut3r_ssim_backward(_p(im["color"]), _p(v.original_image), _p(self.sd1), _p(self.sd2), _p(self.sd3), 3, H, W,
                                              _p(self.ssim_scale), _p(self.g_ssim), _s()), "ssim_backward")
                self.g_img.add_(self.g_ssim)
                if final:                                 # the reported loss: the two terms that have no sums kernel, in tensor operations
                    vis = self.radii > 0
                    sc = self.scales
                    iso = (torch.abs(sc - sc.mean(dim=1, keepdim=True)) * vis[:, None]).sum() / (3 * vis.sum()).clamp_min(1)
                    extra = extra + g * (0.2 * (1.0 - self.smap.mean()) + mp.lambda_iso * iso)
                self._backward(v, ps[k], self.g_img, self.g_depth, g * mp.lambda_iso, self.gtheta, sums[k] if sums is not None else None)
            gm.steps += 1
            b1, b2 = 0.9, 0.999
            check(lib.cut3r_gs_adam(P * 14, _p(gm.theta), _p(gm.m), _p(gm.v), _p(self.gtheta), _p(gm.lr), b1, b2, 1 - b1 ** gm.steps,
                                    1 - b2 ** gm.steps, 1e-15, _s()), "gs_adam")
            if sums is not None:
                for k in range(N):
                    check(lib.cut3r_gs_pose_step(_p(ps[k]), _p(sums[k]), 0.0, None, lr * 2, lr * 10, 1, _s()), "gs_pose_step")
            if final:
                last[0] = self.loss_acc[0] + extra
            return counts

        self._run(body, iters, views, snapshot)
        gm._steps_dev_stale = True
        if optimize_pose:
            self._store_poses(views, ps)
        return float(last[0]) if last[0] is not None else None

    def pose_refine(self, views, iters, alpha_th=0.5):
        """the optimisation part of GSMapper.pose_refine: the Gaussians stay fixed, the increments of the views' poses move (not folded
        until the end, as the reference's update_pose after the loop)"""
        mp, gm, lib = self.mp, self.mp.gaussians, self.lib
        P = len(gm)
        H, W = self._image_size(views)
        self._buffers(P, H, W)
        B = len(views)
        ps = self._load_poses(views)
        sums = torch.zeros(B, 16, dtype=torch.float32, device=self.dev)
        ratios = torch.zeros(B, 1, dtype=torch.float32, device=self.dev)
        lr = mp.config["opt_params"]["pose_lr"]
        saved = {}

        def snapshot(restore=False):
            if restore:
                ps.copy_(saved["ps"])
            else:
                saved["ps"] = ps.clone()
        snapshot()

        def body(it, exact):
            sums.zero_()
            counts = []
            for k, v in enumerate(views):
                im = self.img
                counts.append(self._render(v, ps[k], exact))
                check(lib.cut3r_refine_loss_forward(_p(im["color"]), _p(v.original_image), _p(im["depth"]), _p(v.depth), _p(im["alpha"]),
                                                    float(alpha_th), H, W, _p(self.sums), _s()), "refine_loss_forward")
                check(lib.cut3r_gs_refine_coef(_p(self.sums), 5.0 / B, 1.0 / B, H, W, _p(self.coef), _p(ratios[k]), None, _s()), "gs_refine_coef")
                check(lib.cut3r_refine_loss_backward(_p(im["color"]), _p(v.original_image), _p(im["depth"]), _p(v.depth), _p(im["alpha"]),
                                                     float(alpha_th), H, W, _p(self.coef), _p(self.g_img), _p(self.g_depth), _s()), "refine_loss_backward")
                self._backward(v, ps[k], self.g_img, self.g_depth, 0.0, None, sums[k])
            for k in range(B):
                check(lib.cut3r_gs_pose_step(_p(ps[k]), _p(sums[k]), 0.05 / B, _p(ratios[k]), lr * 2, lr * 10, 0, _s()), "gs_pose_step")
            return counts

        self._run(body, iters, views, snapshot)
        for k in range(B):                                # update_pose: T <- exp(delta) T, delta <- 0
            check(lib.cut3r_gs_pose_step(_p(ps[k]), _p(sums[k]), 0.0, None, 0.0, 0.0, 2, _s()), "gs_pose_step (fold)")
        self._store_poses(views, ps)

    def global_BA(self, iteration_total, densify=True, densify_every=None, opacity_reset=True, seed=0):
        """GSMapper.global_BA (gs_backend_per_frame.py:946-1058) without exposure compensation: one randomly drawn keyframe per iteration,
        colour + (inverse depth) + depth-normal agreement + the rendered-normal term (cut3r_normal_agree_*), densification statistics
        (cut3r_gs_densify_stats), clone / split / prune and opacity resets at the reference's iterations, position learning-rate decay.
        The instance count is read every iteration (the set of Gaussians changes under densification)."""
        import random
        from .gs_mapper import depth_to_normal, position_lr
        mp, lib = self.mp, self.lib
        gm = mp.gaussians
        views = list(mp.viewpoints.values())
        H, W = self._image_size(views)
        ps = self._load_poses(views)
        sums = torch.zeros(len(views), 16, dtype=torch.float32, device=self.dev)
        rng = random.Random(seed)
        tr, op = mp.config["Training"], mp.config["opt_params"]
        update_every, reset_every = tr.get("gaussian_update_every", 200), tr.get("gaussian_reset", 3001)
        lr = op["pose_lr"]
        w_d, w_n = (mp.lambda_depth / 10, mp.lambda_normal) if densify_every is not None else (0.0, mp.lambda_normal / 2)
        self._buffers(len(gm), H, W)
        self.ssim_scale.fill_(-0.2 / (3 * H * W))
        g_nrm = torch.empty(3, H, W, dtype=torch.float32, device=self.dev)
        last = None
        for iteration in range(iteration_total):
            mp.iteration_count += 1
            gm = mp.gaussians
            P = len(gm)
            self._buffers(P, H, W)
            k = rng.randint(0, len(views) - 1)
            v = views[k]
            gc = getattr(v, "_gt_normal", None)
            if gc is None or gc[0] is not v.depth:
                v._gt_normal = (v.depth, depth_to_normal(v, v.depth[None]).detach().contiguous())
            final = iteration == iteration_total - 1
            cam, im = self._cam(v), self.img
            K = cam["K"]
            self.gtheta.zero_()
            sums[k].zero_()
            if final:
                self.loss_acc.zero_()
            self._render(v, ps[k], True)
            check(lib.cut3r_pixel_loss_forward(_p(im["color"]), _p(v.original_image), _p(im["depth"]), _p(v.depth), _p(v._gt_normal[1]), H, W, K[0], K[1],
                                               K[2], K[3], _p(self.sums), _s()), "pixel_loss_forward")
            check(lib.cut3r_gs_map_coef(_p(self.sums), 0.8, float(w_d), float(w_n), 1.0, H, W, _p(self.coef), _p(self.loss_acc) if final else None, _s()),
                  "gs_map_coef")
            check(lib.cut3r_pixel_loss_backward(_p(im["color"]), _p(v.original_image), _p(im["depth"]), _p(v.depth), _p(v._gt_normal[1]), H, W, K[0], K[1],
                                                K[2], K[3], _p(self.coef), _p(self.g_img), _p(self.g_depth), _s()), "pixel_loss_backward")
            check(lib.cut3r_ssim_forward(_p(im["color"]), _p(v.original_image), 3, H, W, _p(self.smap), _p(self.sd1), _p(self.sd2), _p(self.sd3), _s()),
                  "ssim_forward")
            check(lib.cut3r_ssim_backward(_p(im["color"]), _p(v.original_image), _p(self.sd1), _p(self.sd2), _p(self.sd3), 3, H, W, _p(self.ssim_scale),
                                          _p(self.g_ssim), _s()), "ssim_backward")
            self.g_img.add_(self.g_ssim)
            if final:
                check(lib.cut3r_normal_agree_forward(_p(im["normal"]), _p(im["depth"]), H, W, K[0], K[1], K[2], K[3], _p(self.nvis), _s()), "normal_agree_forward")
                last = self.loss_acc[0] + 0.2 * (1.0 - self.smap.mean()) + w_n * self.nvis[0] / (H * W)
            check(lib.cut3r_normal_agree_backward(_p(im["normal"]), _p(im["depth"]), H, W, K[0], K[1], K[2], K[3], float(w_n) / (H * W), _p(g_nrm),
                                                  _p(self.g_depth), _s()), "normal_agree_backward")
            self._backward(v, ps[k], self.g_img, self.g_depth, 0.0, self.gtheta, sums[k], g_normal=g_nrm)
            if iteration < 10000 and densify:
                check(lib.cut3r_gs_densify_stats(P, _p(self.radii), _p(self.d_means2D), _p(gm.max_radii2D), _p(gm.grad_accum), _p(gm.grad_accum_abs), _p(gm.denom), _s()),
                      "gs_densify_stats")
            # the reference's order (gs_backend_per_frame.py:1025-1041): densify_and_prune / reset_opacity come BEFORE optimizer.step() and
            # re-create the parameters, whose .grad is then None: Adam skips every group after a densification (no update, no moment
            # update, no step increment) and the opacity group after a reset (GaussianMap.step_like_reference)
            do_densify = do_reset = False
            if iteration < 10000 and densify:
                do_densify = (iteration == iteration_total // 2) if densify_every is not None else ((mp.iteration_count + 1) % update_every == 0)
                if do_densify:
                    gm.densify_and_prune(op["densify_grad_threshold"], mp.gaussian_th, mp.gaussian_extent, mp.size_threshold)
                do_reset = bool((mp.iteration_count + 1) % reset_every == 0 and opacity_reset)
                if do_reset:
                    gm.reset_opacity()
            if not do_densify:
                keep = gm.theta.detach()[:, 6:7].clone() if do_reset else None
                gm.steps += 1
                gm._steps_dev_stale = True
                b1, b2 = 0.9, 0.999
                check(lib.cut3r_gs_adam(P * 14, _p(gm.theta), _p(gm.m), _p(gm.v), _p(self.gtheta), _p(gm.lr), b1, b2, 1 - b1 ** gm.steps, 1 - b2 ** gm.steps,
                                        1e-15, _s()), "gs_adam")
                if do_reset:
                    with torch.no_grad():
                        gm.theta[:, 6:7] = keep
                        gm.m[:, 6:7] = 0
                        gm.v[:, 6:7] = 0
            if densify and "position_lr_final" in op:
                gm.lr[0, 0:3] = position_lr(op, iteration)
            check(lib.cut3r_gs_pose_step(_p(ps[k]), _p(sums[k]), 0.0, None, lr * 2, lr * 10, 1, _s()), "gs_pose_step")
        self._store_poses(views, ps)
        return float(last) if last is not None else None

    def reinit_loop(self, iteration_total, seed=0):
        """the training loop of GSMapper.gaussian_reinit (gs_backend_per_frame.py:865-944): single-view iterations with FIXED poses --
        colour, inverse depth and the depth-normal term -- densification statistics and clone / split / prune every
        Training.gaussian_update_every iterations after the first 1000"""
        import random
        from .gs_mapper import depth_to_normal
        mp, lib = self.mp, self.lib
        views = list(mp.viewpoints.values())
        H, W = self._image_size(views)
        ps = self._load_poses(views)
        rng = random.Random(seed)
        update_every = mp.config["Training"].get("gaussian_update_every", 200)
        op = mp.config["opt_params"]
        self._buffers(len(mp.gaussians), H, W)
        self.ssim_scale.fill_(-0.2 / (3 * H * W))
        last = None
        for iteration in range(iteration_total):
            gm = mp.gaussians
            P = len(gm)
            self._buffers(P, H, W)
            k = rng.randint(0, len(views) - 1)
            v = views[k]
            gc = getattr(v, "_gt_normal", None)
            if gc is None or gc[0] is not v.depth:
                v._gt_normal = (v.depth, depth_to_normal(v, v.depth[None]).detach().contiguous())
            final = iteration == iteration_total - 1
            cam, im = self._cam(v), self.img
            K = cam["K"]
            self.gtheta.zero_()
            if final:
                self.loss_acc.zero_()
            self._render(v, ps[k], True)
            check(lib.cut3r_pixel_loss_forward(_p(im["color"]), _p(v.original_image), _p(im["depth"]), _p(v.depth), _p(v._gt_normal[1]), H, W, K[0], K[1],
                                               K[2], K[3], _p(self.sums), _s()), "pixel_loss_forward")
            check(lib.cut3r_gs_map_coef(_p(self.sums), 0.8, float(mp.lambda_depth), float(mp.lambda_normal), 1.0, H, W, _p(self.coef),
                                        _p(self.loss_acc) if final else None, _s()), "gs_map_coef")
            check(lib.cut3r_pixel_loss_backward(_p(im["color"]), _p(v.original_image), _p(im["depth"]), _p(v.depth), _p(v._gt_normal[1]), H, W, K[0], K[1],
                                                K[2], K[3], _p(self.coef), _p(self.g_img), _p(self.g_depth), _s()), "pixel_loss_backward")
            check(lib.cut3r_ssim_forward(_p(im["color"]), _p(v.original_image), 3, H, W, _p(self.smap), _p(self.sd1), _p(self.sd2), _p(self.sd3), _s()),
                  "ssim_forward")
            check(lib.cut3r_ssim_backward(_p(im["color"]), _p(v.original_image), _p(self.sd1), _p(self.sd2), _p(self.sd3), 3, H, W, _p(self.ssim_scale),
                                          _p(self.g_ssim), _s()), "ssim_backward")
            self.g_img.add_(self.g_ssim)
            if final:
                last = self.loss_acc[0] + 0.2 * (1.0 - self.smap.mean())
            self._backward(v, ps[k], self.g_img, self.g_depth, 0.0, self.gtheta, None)
            if iteration > 1000:
                check(lib.cut3r_gs_densify_stats(P, _p(self.radii), _p(self.d_means2D), _p(gm.max_radii2D), _p(gm.grad_accum), _p(gm.grad_accum_abs), _p(gm.denom), _s()),
                      "gs_densify_stats")
            if iteration > 1000 and (iteration + 1) % update_every == 0:
                # (gs_backend_per_frame.py:920-935: densification first, then an optimizer.step() that finds no gradients)
                gm.densify_and_prune(op["densify_grad_threshold"], mp.gaussian_th, mp.gaussian_extent, mp.size_threshold)
            else:
                gm.steps += 1
                gm._steps_dev_stale = True
                b1, b2 = 0.9, 0.999
                check(lib.cut3r_gs_adam(P * 14, _p(gm.theta), _p(gm.m), _p(gm.v), _p(self.gtheta), _p(gm.lr), b1, b2, 1 - b1 ** gm.steps, 1 - b2 ** gm.steps,
                                        1e-15, _s()), "gs_adam")
        return float(last) if last is not None else None
