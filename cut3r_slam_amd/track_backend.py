"""Loop closure backend with the reference's control flow (/root/reference/hislam2/track_backend.py:527-586 `run`,
:137-217 `track`, :220-358 `loop_closure_init`, :361-524 `loop_closure`) on HBM-resident submaps.

  detection : covisible keyframes farther than 8 frames apart (FactorGraph.detect_loop)
  NMS       : 0.8 * mean(bidirectional reprojection overlap) + 0.2 * patch-feature overlap, accept if > 0.4
  re-track  : 6-view inference over [5 keyframes of the matched submap, current keyframe], chained to the anchor
  optimise  : first loop  -> fused HIP Adam over per-submap se(3) (ops.lc_optimize: 2 launches / iteration)
              later loops -> chain + matched-submap + current-vs-lc terms, the same fused optimiser over a term list
                             (ops.lc_optimize_terms), Adam over the submap AND the lc-submap corrections
  rewrite   : every pointmap of every corrected submap in place (ops.transform_submaps) + the 7-float poses
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import geom_host as gh
from . import ops
from .lietorch import SE3


class TrackBackend:
    def __init__(self, slam, keyframes, config, device="cuda:0"):
        self.device = device
        self.keyframes = keyframes
        self.model = slam.model
        self.graph = slam.graph
        self.loop_iters = int(config.get("iteration", 0))
        self._chain_ws = None
        self.downsample_ratio = slam.downsample_ratio
        self.conf_th = 0.05
        self.lc_initialized = False
        self.closed_loop = {"idx_current": [], "idx_matched": [], "pointmaps_lc": []}
        self._lsum = torch.zeros(1, dtype=torch.float64, device=device)
        self._count = torch.zeros(1, dtype=torch.int32, device=device)
        self._ws = None

    # ------------------------------------------------------------------ NMS (factor_graph.py:561-582)
    def _feat_overlap(self, f0, f1s, thr=0.7):
        N, C = f0.shape
        need = 2 * (N - 1) * C + N
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device)
        B = len(f1s)
        if B == 1:
            counts = torch.zeros(1, dtype=torch.int32, device=self.device)
            ops.patch_overlap_count(f0.contiguous(), f1s[0].contiguous(), thr, self._ws, counts)
            return counts.float() / float(N - 1)
        # every candidate against the SAME current keyframe in one call: the look-ahead chain of the motion filter with a ratio threshold
        # nobody passes (no candidate ever becomes the "last keyframe"): rows normalised once, two launches per candidate issued from C,
        # bit-identical counts (tests/test_backend_golden_gpu.py: the reference's values and the per-pair launches)
        feats = torch.stack([f.contiguous() for f in f1s], 0)
        need = (B + 1) * (N - 1) * C + N
        if self._chain_ws is None or self._chain_ws[0].numel() < need or self._chain_ws[1].numel() < 2 * B + 1:
            self._chain_ws = (torch.empty(need, device=self.device), torch.zeros(2 * B + 1, dtype=torch.int32, device=self.device))
        ws, ints = self._chain_ws
        ops.patch_overlap_chain(f0.contiguous(), feats, thr, -1.0, None, ws, ints[2 * B:2 * B + 1], ints[:B], ints[B:2 * B])
        return ints[:B].float() / float(N - 1)

    def nms_scores(self, ids_matched, idx_current, K4):
        """scores [B] of the loop candidates `ids_matched` for keyframe idx_current (factor_graph.py:561-577):
        0.8 * mean(overlap of the candidates' maps in the current camera, overlap of the current map in the candidates'
        cameras) + 0.2 * patch-feature overlap (compute_feature_overlap_batch, :328-341).  Returned on the host."""
        kf = self.keyframes
        h, w = kf.submap_ds.shape[2], kf.submap_ds.shape[3]
        if kf.feat_rows:
            raise RuntimeError("loop closure needs the full keyframe feature store (KeyFrame feat_buffer = 0)")
        ids = torch.as_tensor(np.asarray(ids_matched), dtype=torch.long)
        pm_matched = kf.submap_ds[ids // 5, ids % 5].contiguous()                       # [B,h,w,3]
        pm_cur = kf.submap_ds[idx_current // 5, idx_current % 5].contiguous()
        B = len(ids)
        a2c = torch.empty(B, dtype=torch.int32, device=self.device)
        c2a = torch.empty(B, dtype=torch.int32, device=self.device)
        ops.overlap_bwd(pm_matched, kf.w2c[idx_current].contiguous(), K4, w, h, a2c)
        ops.overlap_fwd(pm_cur, kf.w2c[ids.to(kf.w2c.device)].contiguous(), K4, w, h, c2a, clamp_z=False)
        feat = self._feat_overlap(kf.featI[idx_current], [kf.featI[int(i)] for i in ids])
        overlap = (a2c.float() / (h * w) + c2a.float() / (h * w)) / 2
        return (0.8 * overlap + 0.2 * feat).cpu()

    def nms(self, ids_matched, idx_current, K4, th=0.4):
        scores = self.nms_scores(ids_matched, idx_current, K4)
        if float(scores.max()) > th:
            return int(torch.argmax(scores))
        return None

    # ------------------------------------------------------------------ re-tracking against the anchor submap
    def track(self, selected_idx, anchor_sub_num):
        kf, ds = self.keyframes, self.downsample_ratio
        sel = torch.as_tensor(np.asarray(selected_idx), dtype=torch.long)
        sel_d = sel.to(kf.image.device)
        miss = [int(i) for i in sel if not kf.feat_valid[int(i)]]
        if miss:
            mi = torch.as_tensor(miss, device=kf.image.device)
            kf.featI[mi] = self.model.encode_batch(kf.image[mi])
            for i in miss:
                kf.feat_valid[i] = True
        preds, _ = self.model.decode_window(kf.featI[sel_d].contiguous(), kf.ht, kf.wd)
        pts = torch.cat([p["pts3d_in_self_view"] for p in preds], 0).contiguous()
        conf = torch.cat([p["conf_self"] for p in preds], 0).contiguous()
        enc = torch.cat([p["camera_pose"] for p in preds], 0)
        V, H, W, _ = pts.shape
        a = anchor_sub_num * 5
        ops.logdepth_sum(kf.depth[a], pts[0], self._lsum)
        host = enc.detach().cpu().numpy()
        align_s = np.float32(math.exp(np.float32(float(self._lsum.item()) / (H * W))))
        poses = gh.pose_encoding_to_camera(host)
        first_w2c = gh.inv4(poses[0])
        prev = gh.pose_vec_to_matrix(kf.pose[a].numpy()[None])[0]
        pm = torch.empty(V, H // ds, W // ds, 3, device=self.device)
        cf = torch.empty(V, H // ds, W // ds, device=self.device)
        dp = torch.empty(H, W, device=self.device)
        out_poses = []
        for i in range(V):
            pose = gh.chain_pose(first_w2c, poses[i], prev[:3, :3], prev[:3, 3], align_s)
            ops.align_view(pts[i], conf[i], pose[:3, :4].reshape(-1), float(align_s), ds, pm[i], cf[i], dp)
            out_poses.append(gh.matrix_to_pose_vec(pose))
        return pm, cf, np.stack(out_poses)

    # ------------------------------------------------------------------ optimisation + rewrite
    def _rewrite(self, sub0, sub1, T34, include_last=True):
        """submaps sub0..sub1 <- T_b applied; poses of their keyframes likewise (track_backend.py:301-346)."""
        kf = self.keyframes
        B = sub1 - sub0 + 1
        block = kf.submap_ds[sub0:sub1 + 1]
        ops.transform_submaps(block, T34.reshape(B, 12).contiguous())
        Th = T34.detach().cpu().numpy()
        # the keyframes of submaps sub0..sub1 are the contiguous range [5 sub0, 5 (sub1 + 1)) (+ the first keyframe of the next submap, moved
        # by the last transform): one batched host pass -- per keyframe the same arithmetic as the former loop (4x4 product per keyframe,
        # scipy's Rotation on the stack) -- and ONE upload of the world->camera rows
        i0, n = sub0 * 5, B * 5 + (1 if include_last else 0)
        c2w = gh.pose_vec_to_matrix(kf.pose[i0:i0 + n].numpy())
        out = np.empty_like(c2w)
        for j in range(n):
            Ts = np.eye(4, dtype=np.float32)
            Ts[:3, :4] = Th[min(j // 5, B - 1)]
            out[j] = Ts @ c2w[j]
        new = gh.matrices_to_pose_vecs(out)
        kf.set_poses(i0, new)
        return new

    def loop_closure_init(self, pointmap_current_lc, idx_matched, idx_current, return_loss=False):
        kf = self.keyframes
        sub0, sub1 = 0, idx_current // 5
        block = kf.submap_ds[sub0:sub1 + 1]                                               # [B,6,h,w,3] view, resident
        B = block.shape[0]
        mask = (kf.conf_ds[sub0:sub1, 5] > 0).reshape(B - 1, -1) if B > 1 else None         # conf of each submap's last map
        cur = kf.submap_ds[sub1, idx_current % 5]
        res = ops.lc_optimize(block.contiguous(), mask, cur, pointmap_current_lc, self.loop_iters, 5e-4, return_loss)
        xi, T = res[0], res[1]
        se3 = SE3.exp(xi)
        new_pose = self._rewrite(sub0, sub1, T)
        updates = {"pose_updates": se3.data, "submap_idx": range(sub0, sub1 + 1),
                   "camera_idx": range(sub0 * 5, (sub1 + 1) * 5 + 1), "camera_pose": torch.from_numpy(new_pose)}
        if return_loss:
            updates["loss"] = res[2]
        return updates

    def loop_closure(self, pointmaps_lc, idx_matched, idx_current, return_loss=False):
        """Second and later loops (track_backend.py:361-524): Adam over the per-submap corrections `_align_lie` [B-1,6] AND one
        se(3) per re-tracked "lc" submap (`matched_lie` [Bc,6]) on three L1 terms -- the chain of first/last maps (:447), each
        lc submap's first map against the first map of the submap it matched (:449) and each loop's current map against
        the last map of its lc submap (:451).  One fused HIP optimiser call (ops.lc_optimize_terms: two launches per iteration)
        over the resident stores; returns (aligned lc submap of THIS loop [6,h,w,3], updates) like the reference."""
        kf = self.keyframes
        sub1 = idx_current // 5
        block = kf.submap_ds[0:sub1 + 1]                                                 # [B,6,h,w,3] view of the resident store
        B, N6, h, w, _ = block.shape
        N = h * w
        prev_cur = np.array(self.closed_loop["idx_current"], dtype=np.int64)
        sub_cur_all = np.append(prev_cur // 5, sub1)
        pm_cur = torch.cat([kf.submap_ds[torch.as_tensor(prev_cur // 5), torch.as_tensor(prev_cur % 5)],
                            kf.submap_ds[sub1, idx_current % 5][None]], 0).contiguous()                     # [Bc,h,w,3] global frame
        lc_all = torch.cat([torch.stack(self.closed_loop["pointmaps_lc"], 0), pointmaps_lc[None]], 0).float().contiguous()   # [Bc,6,h,w,3]
        Bc = lc_all.shape[0]
        sub_matched_all = np.append(np.array(self.closed_loop["idx_matched"], dtype=np.int64) // 5, idx_matched // 5)
        # transform table: 0..B-1 = submap corrections (0 fixed = lie_0), B..B+Bc-1 = matched_lie
        terms = []
        w_fl = 1.0 / (3.0 * (B - 1) * N) if B > 1 else 0.0
        w_c = 1.0 / (3.0 * Bc * N)
        for p in range(B - 1):
            terms.append((block[p, N6 - 1], p, block[p + 1, 0], p + 1, w_fl, None))
        for k in range(Bc):
            terms.append((lc_all[k, 0], B + k, block[int(sub_matched_all[k]), 0], int(sub_matched_all[k]), w_c, None))
        for k in range(Bc):
            terms.append((pm_cur[k], int(sub_cur_all[k]), lc_all[k, N6 - 1], B + k, w_c, None))
        res = ops.lc_optimize_terms(terms, B + Bc, N, self.loop_iters, 5e-4, return_loss)
        xi, T = res[0], res[1]
        se3 = SE3.exp(xi[:B].contiguous())
        new_pose = self._rewrite(0, sub1, T[:B].contiguous())
        ops.transform_submaps(lc_all, T[B:].reshape(Bc, 12).contiguous())              # lc submaps moved by their matched transform (:513-516)
        for i in range(Bc - 1):
            self.closed_loop["pointmaps_lc"][i] = lc_all[i]
        updates = {"pose_updates": se3.data, "submap_idx": range(0, sub1 + 1), "camera_idx": range(0, (sub1 + 1) * 5 + 1),
                   "camera_pose": torch.from_numpy(new_pose)}
        if return_loss:
            updates["loss"] = res[2]
        return lc_all[-1], updates

    # ------------------------------------------------------------------ entry point (track_backend.py:527-586)
    def run(self, t1=None):
        kf = self.keyframes
        intr = kf.intrinsic[0].numpy() / self.downsample_ratio
        K4 = [float(intr[0]), float(intr[1]), float(intr[2]), float(intr[3])]
        t1 = kf.counter.value - 1 if t1 is None else int(t1)
        t0 = t1 - 6
        ids_matched, idx_current = None, None
        for idx_current in range(t0, t1 - 1):
            ids_matched = self.graph.detect_loop(idx_current, small_loop_candidates=False)
            if ids_matched is not None:
                break
        if ids_matched is None:
            return False, None
        k_th = self.nms(ids_matched, idx_current, K4)
        if k_th is None:
            return False, None
        idx_matched = int(ids_matched[k_th])
        anchor = idx_matched // 5
        selected = list(range(anchor * 5, (anchor + 1) * 5)) + [idx_current]
        pm_lc, conf_lc, poses_lc = self.track(selected, anchor)
        return True, self.close_loop(pm_lc, idx_matched, idx_current)

    def close_loop(self, pm_lc, idx_matched, idx_current):
        """submap-level optimisation + bookkeeping of one accepted loop (track_backend.py:559-575).  pm_lc [6,h,w,3]: the
        re-tracked submap [5 keyframes of the matched submap, current keyframe] in the matched submap's frame."""
        if not self.lc_initialized:
            updates = self.loop_closure_init(pm_lc[-1], idx_matched, idx_current)
            self.lc_initialized = True
            stored = pm_lc
        else:
            # later loops keep the lc submap as aligned by its matched transform (track_backend.py:566-575 rebinds
            # pointmaps_lc to pointmaps_lc_aligned[-1] before appending it)
            stored, updates = self.loop_closure(pm_lc, idx_matched, idx_current)
        self.closed_loop["idx_current"].append(idx_current)
        self.closed_loop["idx_matched"].append(idx_matched)
        self.closed_loop["pointmaps_lc"].append(stored)
        return updates
