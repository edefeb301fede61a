"""Window tracker: same control flow, state updates and graph calls as the reference `TrackFrontend`
(/root/reference/hislam2/track_frontend.py:17-330), re-hosted on HBM-resident buffers and HIP kernels.

Per window the reference copies every predicted map to the CPU, does the chaining with CPU tensor ops and
re-uploads all previous pointmaps for every keyframe.  Here the only device->host traffic per window is the V x 7
camera poses + one fp64 scalar (the log-depth sum); pointmaps, confidences and depths never leave HBM.
"""
from __future__ import annotations

import math
import time
import warnings

import numpy as np
import torch

from . import geom_host as gh
from . import ops
from .factor_graph import AlignedPoints, SubmapStore


def _check_scale(t0, s, stats=None):
    """The reference chains windows by align_s = exp(mean(log depth_stored[t0] - log depth_window[view 0])) (track_frontend.py:203-222)
    and multiplies the window's points and translations by it, silently.  The same arithmetic here, plus book-keeping and a warning
    when the scale leaves the finite fp32 range.  Two things can do that: a predicted depth <= 0 in the shared keyframe (log of it
    is NaN / -inf), or -- what an uncut stream through a network whose view-0 and view-5 depths of the same image differ by a
    constant factor produces -- geometric growth of the chained scale, log s_k = log s_{k-1} + delta, until exp() overflows (the
    camera centres overflow the squared-distance gate of the covisibility test, factor_graph.py:148-160, well before that).
    `stats` (TrackFrontend.scale_stats) records the drift so that a caller can tell the two apart."""
    s = float(s)
    if stats is not None:
        stats["windows"] += 1
        if math.isfinite(s) and s > 0:
            ls = math.log(s)
            stats["log_scale_last"] = ls
            stats["log_scale_absmax"] = max(stats["log_scale_absmax"], abs(ls))
    if not (math.isfinite(s) and s > 0):
        if stats is not None:
            stats["nonfinite_windows"] += 1
            if stats["first_nonfinite_keyframe"] is None:
                stats["first_nonfinite_keyframe"] = t0
        grew = stats is not None and stats["log_scale_absmax"] > 40.0
        warnings.warn(f"window at keyframe {t0}: chained log-depth scale is {s} ("
                      + ("the chained scale grew geometrically along the stream: |log s| reached "
                         f"{stats['log_scale_absmax']:.1f} before leaving the fp32 range" if grew else
                         "a non-positive or non-finite predicted depth in the shared keyframe, or an overflowing chained scale")
                      + "); poses from this window on are not finite", RuntimeWarning, stacklevel=3)


def new_scale_stats():
    return {"windows": 0, "nonfinite_windows": 0, "first_nonfinite_keyframe": None, "log_scale_last": 0.0, "log_scale_absmax": 0.0}


def make_views(model, images_u8):
    """View dicts exactly as the reference builds them (track_frontend.py:47-75), for the `inference()` entry point."""
    images = model.normalize(images_u8.float())
    views = []
    for i in range(len(images)):
        views.append({
            "img": images[i][None],
            "ray_map": torch.full((1, 6, images[i].shape[-2], images[i].shape[-1]), torch.nan),
            "true_shape": torch.from_numpy(np.int32([images[i].shape[-2], images[i].shape[-1]])),
            "idx": i, "instance": str(i),
            "camera_pose": torch.from_numpy(np.eye(4, dtype=np.float32)).unsqueeze(0),
            "img_mask": torch.tensor(True).unsqueeze(0), "ray_mask": torch.tensor(False).unsqueeze(0),
            "update": torch.tensor(True).unsqueeze(0), "reset": torch.tensor(False).unsqueeze(0)})
    return views


TIMING = {"sync1_s": 0.0, "sync2_s": 0.0, "windows": 0}     # host wall-clock of the two device round trips per window


class TrackFrontend:
    def __init__(self, slam, keyframes, config, device="cuda:0"):
        self.device = device
        self.keyframes = keyframes
        self.model = slam.model
        self.graph = slam.graph
        self.t1 = 0
        self.warmup = 6
        self.frontend_nms = config.get("frontend_nms", 1)
        self.keyframe_thresh = config.get("keyframe_thresh", 4.0)
        self.frontend_window = config.get("frontend_window", 25)
        self.frontend_thresh = config.get("frontend_thresh", 16.0)
        self.frontend_radius = config.get("frontend_radius", 2)
        self.keyframes.mono_depth_alpha = config.get("mono_depth_alpha", 0.01)
        self.downsample_ratio = slam.downsample_ratio
        # throughput knob (not in the reference, default 1 = reference behaviour): wait for `window_batch` windows of new
        # keyframes and push them through the decoder TOGETHER (windows are independent network evaluations; the chaining
        # below stays sequential).  Trades latency (window_batch*5 keyframes) for MFMA-bound decoder GEMMs.
        self.window_batch = int(config.get("window_batch", 1))
        # throughput knob (not in the reference, default 0 = one endless sequence): cut the stream into consecutive SEQUENCES of
        # `sequence_windows` windows.  The cut keyframe (view 0 of the first window of a sequence = last keyframe of the previous
        # one) becomes keyframe 0 of a new run: that window is handled exactly like the reference's initialisation window
        # (track_frontend.py:181-200: no chaining, scale 1), the covisibility graph starts empty and the overlap tests only see
        # the keyframes of the current sequence -- the results of a sequence equal a fresh run over its frames (tested).  A
        # Replica-shaped 2000-frame sequence at kf_every = 10 is 40 windows.
        self.sequence_windows = int(config.get("sequence_windows", 0))
        self._lsum = torch.zeros(1, dtype=torch.float64, device=device)     # kept at 0 between windows (window_update resets it)
        self._lsum_host = torch.zeros(1, dtype=torch.float64).pin_memory()
        self._counts = None
        self._counts_host = None
        self._pose_host = torch.zeros(8, 7, dtype=torch.float32).pin_memory()
        self._ev = None
        self.scale_stats = new_scale_stats()     # chained-scale book-keeping of every steady-state window (see _check_scale)

    def prepare_input(self, images):
        return make_views(self.model, images)

    # ------------------------------------------------------------------ one window
    def window_features(self, t0, t1):
        """encoder features of keyframes t0..t1-1: cached ones are reused, missing ones are encoded in one batch and
        written back to keyframes.featI (so each keyframe goes through the ViT-L encoder exactly once)."""
        kf = self.keyframes
        missing = [i for i in range(t0, t1) if not kf.feat_valid[i]]
        if missing:
            # runs of consecutive keyframes -> slices of the resident image store (no index tensor: a host->device copy
            # of pageable memory would block until the stream drains)
            runs, a = [], missing[0]
            for p, q in zip(missing, missing[1:] + [None]):
                if q != p + 1:
                    runs.append((a, p + 1))
                    a = q
            imgs = kf.image[runs[0][0]:runs[0][1]] if len(runs) == 1 else torch.cat([kf.image[a:b] for a, b in runs], 0)
            feats = self.model.encode_batch(imgs)
            o = 0
            for a, b in runs:
                kf.feat_store(a, b, feats[o:o + b - a])
                o += b - a
        return kf.feat_slice(t0, t1) if (not kf.feat_rows or t1 - t0 <= 6) else None

    def infer(self, imgs_u8=None, t0=None, t1=None):
        """model outputs consumed by SLAM (track_frontend.py:81-100 keeps only these three)."""
        if t0 is not None:
            H, W = self.keyframes.ht, self.keyframes.wd
            preds, _ = self.model.decode_window(self.window_features(t0, t1), H, W)
        else:
            preds, _ = self.model.forward_window(imgs_u8)
        pts = torch.cat([p["pts3d_in_self_view"] for p in preds], 0)       # [V,H,W,3]
        conf = torch.cat([p["conf_self"] for p in preds], 0)               # [V,H,W]
        pose_enc = torch.cat([p["camera_pose"] for p in preds], 0)         # [V,7] (t, q wxyz)
        return pts.contiguous(), conf.contiguous(), pose_enc

    def predict(self, new_img, kf_img, kf_pose, kf_depth, kf_pointmap=None, outputs=None):
        """track_frontend.py:102-162: pose / depth / stride-ds pointmap of a NON-keyframe `new_img` from a 2-view inference
        [keyframe, new frame], chained to the keyframe by the log-depth scale and the keyframe pose (hi2.py:203 uses it to
        densify the map between distant keyframes).  new_img, kf_img u8 [3,H,W]; kf_pose [7] c2w (t, q_xyzw); kf_depth
        [H,W].  Returns (new_pose [7] host, new_depth [H,W], new_pointmap [h,w,3], new_conf [h,w]) -- the last three on the
        GPU; new_conf is the raw confidence of view 0 as in the reference (:114,157).  kf_pointmap is only read by the
        reference's disabled Umeyama branch."""
        ds = self.downsample_ratio
        if outputs is None:
            outputs = self.infer(torch.stack([kf_img, new_img], 0).to(self.device))
        pts, conf, pose_enc = outputs
        _, H, W, _ = pts.shape
        kf_depth = torch.as_tensor(kf_depth).to(self.device, torch.float32).contiguous()
        lsum = torch.zeros(1, dtype=torch.float64, device=self.device)
        ops.logdepth_sum(kf_depth, pts[0].contiguous(), lsum)
        packed = torch.cat([pose_enc.detach().reshape(-1).double(), lsum]).cpu().numpy()
        host, lsum = packed[:-1].astype(np.float32).reshape(2, 7), float(packed[-1])
        poses = gh.pose_encoding_to_camera(host)
        align_s = np.float32(math.exp(np.float32(lsum / (H * W))))
        _check_scale("of predict()", align_s)
        prev_c2w = gh.pose_vec_to_matrix(np.asarray(torch.as_tensor(kf_pose).cpu(), np.float32)[None])[0]
        pose = gh.chain_pose(gh.inv4(poses[0]), poses[1], prev_c2w[:3, :3], prev_c2w[:3, 3], align_s)
        pm = torch.empty(H // ds, W // ds, 3, device=self.device)
        cf = torch.empty(H // ds, W // ds, device=self.device)
        depth = torch.empty(H, W, device=self.device)
        ops.align_view(pts[1].contiguous(), conf[1].contiguous(), pose[:3, :4].reshape(-1), float(align_s), ds, pm, cf, depth)
        new_pose = torch.from_numpy(gh.matrix_to_pose_vec(pose))
        return new_pose, depth, pm, conf[0][::ds, ::ds].contiguous()

    def track(self, t0, t1, init=False, outputs=None):
        """track_frontend.py:166-262.  `outputs` = (pts, conf, pose_enc) lets callers (tests, the multi-GPU
        driver) supply precomputed network outputs."""
        if outputs is None:
            outputs = self.infer(t0=t0, t1=t1)
        self.track_many([(t0, t1)], [outputs], init=init)

    def seq_base(self, t0):
        """first keyframe of the sequence the window starting at keyframe t0 belongs to (0 without sequence cuts)"""
        S = 5 * self.sequence_windows
        return (t0 // S) * S if S > 0 else 0

    def _decide(self, t0, t1, init, done, base=0):
        """decision half of a window, in the reference's order: neighbourhood factors of keyframe i, then its overlap
        factors (track_frontend.py:246-261 -> factor_graph.py:109-117, 170-197).  Host only.  t0, t1 and the tickets are
        relative to `base`, the first keyframe of the window's sequence."""
        graph = self.graph
        if init:
            if base != graph.base:
                graph.begin_sequence(base)
            graph.add_neighborhood_factors(0, 3, r=3)
        for i in range(t0, t1):
            if not init:
                graph.add_neighborhood_factors(i - 3, i + 1, r=3)
            if i in done:
                graph.add_finish(*done[i])

    def prefetch_logdepth(self, t0, pts, pose_enc):
        """Issue (no wait) the log-depth reduction + pose read-back of the first window of a coming track_many call, so a
        pipelined driver can queue it AHEAD of the next network pass; pass the returned event as `first_event`."""
        kf = self.keyframes
        if pose_enc.is_cuda:
            self._pose_host[:pose_enc.shape[0]].copy_(pose_enc.detach(), non_blocking=True)
        ops.logdepth_accum(kf.depth[t0], pts[0], self._lsum)
        self._lsum_host.copy_(self._lsum, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def track_many(self, ranges, outputs, init=False, first_event=None, count_mask=None, exchange=None, defer_decisions=False):
        """The sequential part of tracking for consecutive windows `ranges` = [(t0, t1), ...] with their network outputs
        [(pts [V,H,W,3], conf [V,H,W], pose_enc [V,7] device or host), ...]: log-depth scale + pose chaining
        (track_frontend.py:193-245), keyframe store update and covisibility-graph update (:246-261).

        ONE device round trip per window: the fused window update of window k is followed on the stream by the log-depth
        reduction of window k+1 (it needs only the depth window k has just stored), both results come back together, and
        the host-only graph decisions of window k run while the device works on window k+1.

        Multi-GPU (`count_mask`, `exchange`): the O(#keyframes) overlap counting of a window is done only where
        count_mask[k] is true (by the rank that owns the window; every rank still chains and stores every window, so the
        stores stay replicated); `exchange(int32 tensor [n_windows, 6, 2, L])` then sums the owners' counts over the ranks
        (one small all-reduce per call) and the decisions of all windows are taken afterwards, in order.  With
        `defer_decisions` that host-only last part is returned as a callable instead of being run, so the caller can launch
        device work first."""
        kf, graph, ds = self.keyframes, self.graph, self.downsample_ratio
        deferred = [] if exchange is not None else None     # (t0, t1, groups, centres) per window, decisions after the exchange
        L = ((max(t1 - self.seq_base(t0) for t0, t1 in ranges) + 63) // 64) * 64      # widest count row: keyframes of a window's sequence
        all_counts = torch.zeros(len(ranges), 6, 2, L, dtype=torch.int32) if exchange is not None else None
        if self._ev is None:
            self._ev = torch.cuda.Event()
        pending = None                    # (t0, t1, init, done, base) of the previous window: its decisions are still to be made
        lsum_next = None                  # log-depth sum of the coming window when the previous round trip brought it back
        h, w = kf.submap_ds.shape[2:4]
        for k, ((t0, t1), (pts, conf, pose_enc)) in enumerate(zip(ranges, outputs)):
            V, H, W, _ = pts.shape
            b0 = self.seq_base(t0)                      # first keyframe of this window's sequence
            init_k = init or (b0 == t0 and t0 > 0)      # a cut: this window starts a sequence like an initialisation window
            if k == 0 and first_event is not None and not init_k:
                tic = time.perf_counter()
                first_event.synchronize()                                 # issued by prefetch_logdepth before the network pass
                lsum_next = float(self._lsum_host[0])
                TIMING["sync1_s"] += time.perf_counter() - tic
            if pose_enc.is_cuda and lsum_next is None:
                self._pose_host[:V].copy_(pose_enc.detach(), non_blocking=True)
            if not init_k and lsum_next is None:
                tic = time.perf_counter()
                ops.logdepth_accum(kf.depth[t0], pts[0], self._lsum)     # window k's view 0 == previous window's last KF
                self._lsum_host.copy_(self._lsum, non_blocking=True)
                self._ev.record()
                self._ev.synchronize()                                    # extra round trip: first window of a call only
                lsum_next = float(self._lsum_host[0])
                TIMING["sync1_s"] += time.perf_counter() - tic
            elif init_k and pose_enc.is_cuda and lsum_next is None:
                self._ev.record()
                self._ev.synchronize()
            host = (self._pose_host[:V] if pose_enc.is_cuda else pose_enc).numpy().copy()
            lsum, lsum_next = lsum_next, None
            poses = gh.pose_encoding_to_camera(host)
            first_w2c = gh.inv4(poses[0])
            sub_num = t0 // 5
            align = None
            if not init_k:
                align_s = np.float32(math.exp(np.float32(lsum / (H * W))))
                _check_scale(t0, align_s, self.scale_stats)
                prev_c2w = gh.pose_vec_to_matrix(kf.pose[t0].numpy()[None])[0]
                align = (prev_c2w[:3, :3], prev_c2w[:3, 3], align_s)
            # host: chained pose of every view, one vectorised matrix->quaternion conversion (the reference converts and
            # stores per keyframe, track_frontend.py:236-245); the world->camera rows travel as kernel arguments
            if init_k:
                chained, s_win = [gh.chain_pose(first_w2c, poses[v]) for v in range(V)], np.float32(1.0)
            else:
                chained, s_win = [gh.chain_pose(first_w2c, poses[v], *align) for v in range(V)], align[2]
            w2c_rows = kf.set_poses(t0, gh.matrices_to_pose_vecs(np.stack(chained)), upload=False)
            centres = kf.pose[b0:t1, :3].numpy()                            # zero-copy view of the host pose table
            # device: ONE fused call stores every view (downsampled chained pointmap, confidence, depth) and counts the
            # forward / backward overlaps of every keyframe of the window (the reference re-uploads all previous pointmaps
            # and reads ratios back per keyframe, track_frontend.py:248-259).  Indices are relative to the sequence's first
            # keyframe b0 (a multiple of 5: submap slots stay aligned): the tests see the cameras / pointmaps b0..i-1
            if self._counts is None or self._counts.shape[-1] < t1 - b0:
                self._counts = torch.zeros(6, 2, max(256, 2 * (t1 - b0)), dtype=torch.int32, device=self.device)
                self._counts_host = torch.zeros(6, 2, self._counts.shape[-1], dtype=torch.int32).pin_memory()
            intr = kf.intrinsic[t0:t1].numpy()
            # one call per run of keyframes with equal intrinsics (a sequence has ONE calibration: normally one call)
            groups, g0 = [], 0
            for v in range(1, V + 1):
                if v == V or not np.array_equal(intr[v], intr[g0]):
                    groups.append((g0, v))
                    g0 = v
            tic = time.perf_counter()
            counting = count_mask is None or bool(count_mask[k])
            for (a, b) in groups:
                ops.window_update(pts[a:b], conf[a:b], np.concatenate([c[:3, :4].reshape(-1) for c in chained[a:b]]), float(s_win), ds,
                                  kf.submap_ds[sub_num, a:b], kf.conf_ds[sub_num, a:b], kf.depth[t0 + a:t0 + b], kf.submap_ds[b0 // 5:],
                                  kf.w2c[b0:], t0 + a - b0, 3 if counting else (1 << 30), [float(x) for x in intr[a]], self._counts[a:b],
                                  w2c_new=w2c_rows[a:b].reshape(-1), lsum_reset=self._lsum)
            if counting:
                self._counts_host[:V].copy_(self._counts[:V], non_blocking=True)
            prefetch = k + 1 < len(ranges)
            if prefetch:                 # the next window's log-depth sum (and poses) ride on the same round trip
                nt0, (npts, _, npose) = ranges[k + 1][0], outputs[k + 1]
                ops.logdepth_accum(kf.depth[nt0], npts[0], self._lsum)
                self._lsum_host.copy_(self._lsum, non_blocking=True)
                if npose.is_cuda:
                    self._pose_host[:npose.shape[0]].copy_(npose.detach(), non_blocking=True)
            self._ev.record()
            if pending is not None:      # host-only decisions of the previous window while the device works on this one
                self._decide(*pending)
                pending = None
            self._ev.synchronize()                                        # THE round trip of the window
            TIMING["sync2_s"] += time.perf_counter() - tic
            TIMING["windows"] += 1
            if prefetch:
                lsum_next = float(self._lsum_host[0])
            r0, r1 = t0 - b0, t1 - b0                                     # the window in its sequence's own numbering
            if deferred is not None:
                if counting:
                    all_counts[k, :V, :, :r1] = self._counts_host[:V, :, :r1]
                deferred.append((r0, r1, groups, centres[:r1].copy(), H * W, init_k, b0))
                continue
            done = {}
            host_counts = self._counts_host.numpy()
            for (a, b) in groups:
                if r0 + b - 1 >= 3:
                    for tk, cf, cb in graph.window_tickets(r0 + a, r0 + b, centres, host_counts[a:b], H * W, h * w):
                        done[tk["idx"]] = (tk, cf.copy(), cb.copy())
            pending = (r0, r1, init_k, done, b0)
        if pending is not None:
            self._decide(*pending)
        if deferred is not None:
            total = exchange(all_counts).numpy()

            def finish():
                for k, (r0, r1, groups, centres, npix, init_k, b0) in enumerate(deferred):
                    done = {}
                    for (a, b) in groups:
                        if r0 + b - 1 >= 3:
                            for tk, cf, cb in graph.window_tickets(r0 + a, r0 + b, centres, total[k, a:b], npix, h * w):
                                done[tk["idx"]] = (tk, cf.copy(), cb.copy())
                    self._decide(r0, r1, init_k, done, b0)
            if defer_decisions:
                return finish
            finish()
        return None

    # ------------------------------------------------------------------ multi-GPU: scan form of the chaining
    # The chain is a scan: window k's scale is exp(mean(log depth_stored[t0] - log d_k[view 0])) with depth_stored[t0] =
    # s_{k-1} * d_{k-1}[view 5] (track_frontend.py:216-222, 235), i.e.
    #       log s_k = log s_{k-1} + ( sum log d_{k-1}[view 5] - sum log d_k[view 0] ) / (H W),
    # and the poses compose through the 7-float pose of the shared keyframe.  So a rank only needs TWO fp64 sums and the 6 raw
    # poses of every other rank's windows (352 bytes per window) to know every scale and every chained pose; it then aligns and
    # stores ITS OWN windows, the stride-2 stores are all-gathered in place, it counts overlaps for its own windows against the
    # now complete store, and one small all-reduce sums the counts.  Per rank and step: O(window_batch) device work + O(world *
    # window_batch) host 4x4 math.  (log(fl32(s d)) vs log s + log d differ at the 1e-7 level: results agree
    # with the sequential form to fp32 rounding, and are bit-identical across ranks and across world sizes.)
    def window_scalars(self, outs):
        """device fp64 [n, 44] per window: (sum log z of view 0, sum log z of the last view, the V x 7 raw pose encodings)"""
        n = len(outs)
        V, H, W, _ = outs[0][0].shape
        if getattr(self, "_ones_hw", None) is None or self._ones_hw.shape != (H, W):
            self._ones_hw = torch.ones(H, W, device=self.device)
            self._ones_pts = torch.ones(H, W, 3, device=self.device)
        acc = torch.zeros(n, 2, dtype=torch.float64, device=self.device)
        for j, (pts, _, _) in enumerate(outs):
            ops.logdepth_accum(self._ones_hw, pts[0], acc[j, 0:1])          # sum(log 1 - log z) = -sum log z
            ops.logdepth_accum(self._ones_hw, pts[V - 1], acc[j, 1:2])
        poses = torch.stack([o[2].detach().reshape(-1) for o in outs], 0).to(self.device, torch.float64)
        return torch.cat([-acc, poses], 1)

    def chain_state(self, t0):
        """scan state in front of the window that starts at keyframe t0: log-scale 0 with the STORED depth of that keyframe"""
        kf = self.keyframes
        H, W = kf.ht, kf.wd
        if getattr(self, "_ones_hw", None) is None or self._ones_hw.shape != (H, W):
            self._ones_hw = torch.ones(H, W, device=self.device)
            self._ones_pts = torch.ones(H, W, 3, device=self.device)
        acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        ops.logdepth_accum(kf.depth[t0], self._ones_pts, acc)                 # sum(log stored - log 1)
        return {"log_s": 0.0, "L5": float(acc.item())}

    def track_sharded(self, ranges, own, outs_own, scal, chain, gather_store, exchange_counts, defer_decisions=False):
        """One step of the multi-GPU tracker.  ranges: ALL windows of the step, in order; own = (i0, i1): the ones this rank
        inferred, with outputs outs_own [(pts, conf, pose_enc)]; scal: host float64 [len(ranges), 44] from every rank's
        window_scalars; chain: scan state (updated in place).  gather_store(sub0, n, phase) completes the stride-2 stores (phase 0) /
        the depth rows (phase 1) of submaps sub0..sub0+n-1 on every rank; exchange_counts sums the owners' int32 counts."""
        kf, graph, ds = self.keyframes, self.graph, self.downsample_ratio
        H, W = kf.ht, kf.wd
        h, w = kf.submap_ds.shape[2:4]
        n = len(ranges)
        i0, i1 = own
        # ---- host scan: scale and chained poses of every window of the step (geom_host.chain_windows: one pose composition and
        #      one quaternion conversion per window in sequence, the other views batched)
        V = ranges[0][1] - ranges[0][0]
        if any(t1 - t0 != V for t0, t1 in ranges):
            raise NotImplementedError("sharded tracking: windows of equal length")
        scal = np.asarray(scal, np.float64)

        bases = [self.seq_base(t0) for t0, _ in ranges]
        cut = [b0 == t0 and t0 > 0 for b0, (t0, _) in zip(bases, ranges)]      # window starts a new sequence: identity pose, scale 1

        def scale_of(k):
            if cut[k]:
                chain["log_s"], chain["L5"] = 0.0, float(scal[k, 1])
                return np.float32(1.0)
            lsum = chain["log_s"] * (H * W) + chain["L5"] - float(scal[k, 0])
            align_s = np.float32(math.exp(np.float32(lsum / (H * W))))
            chain["log_s"], chain["L5"] = math.log(float(align_s)), float(scal[k, 1])
            return align_s

        encs = scal[:, 2:2 + 7 * V].reshape(n, V, 7).astype(np.float32)
        chained_all, s_all, vecs, rows_all = gh.chain_windows(encs, scale_of, kf.pose[ranges[0][0]].numpy(), reset=cut)
        for k in range(n):
            if not cut[k]:
                _check_scale(ranges[k][0], s_all[k], self.scale_stats)
        cent = []
        for k, (t0, t1) in enumerate(ranges):                 # later windows overwrite the shared keyframe, as track() does
            kf.pose[t0:t1] = torch.from_numpy(vecs[k])
            cent.append(kf.pose[bases[k]:t1, :3].numpy().copy())      # camera centres of the window's sequence as track() sees them
        per = [(chained_all[k], s_all[k], rows_all[k]) for k in range(n)]
        # device mirror of every new world->camera row (other ranks' windows included: the forward counts need all cameras)
        ta, tb = ranges[0][0], ranges[-1][1]
        allrows = np.zeros((tb - ta, 12), np.float32)
        for (t0, t1), (_, _, rows) in zip(ranges, per):
            allrows[t0 - ta:t1 - ta] = rows
        if getattr(self, "_rows_pinned", None) is None or self._rows_pinned.shape[0] < tb - ta:
            self._rows_pinned = torch.zeros(max(64, tb - ta), 12).pin_memory()
        self._rows_pinned[:tb - ta].copy_(torch.from_numpy(allrows))
        kf.w2c[ta:tb].copy_(self._rows_pinned[:tb - ta], non_blocking=True)
        L = ((max(t1 - b0 for (t0, t1), b0 in zip(ranges, bases)) + 63) // 64) * 64      # widest count row (sequence-relative)
        nown = i1 - i0
        cm = getattr(self, "_counts_many", None)
        if cm is None or cm.shape[0] < nown or cm.shape[-1] < L:
            self._counts_many = torch.zeros(nown, 6, 2, max(256, 2 * L), dtype=torch.int32, device=self.device)
            self._counts_many_host = torch.zeros(tuple(self._counts_many.shape), dtype=torch.int32).pin_memory()
        intr_all = kf.intrinsic[ta:tb].numpy()

        def update(k, counting):
            (t0, t1), (pts, conf, _), (chained, s_win, rows) = ranges[k], outs_own[k - i0], per[k]
            V = t1 - t0
            intr = intr_all[t0 - ta:t1 - ta]
            if not all(np.array_equal(intr[v], intr[0]) for v in range(V)):
                raise NotImplementedError("sharded tracking assumes one calibration per window")
            sub, b0 = t0 // 5, bases[k]
            ops.window_update(pts, conf, np.concatenate([c[:3, :4].reshape(-1) for c in chained]), float(s_win), ds,
                              kf.submap_ds[sub, :V], kf.conf_ds[sub, :V], kf.depth[t0:t1], kf.submap_ds[b0 // 5:], kf.w2c[b0:], t0 - b0,
                              3 if counting else (1 << 30), [float(x) for x in intr[0]], self._counts_many[k - i0, :V], w2c_new=rows.reshape(-1))

        # ---- phase A: align + store the own windows; phase B: complete the stores; phase C: count for the own windows
        for k in range(i0, i1):
            update(k, False)
        gather_store(ranges[0][0] // 5, n, 0)          # stride-2 pointmaps + confidences of every window of the step
        all_counts = torch.zeros(n, 6, 2, L, dtype=torch.int32)
        if self._ev is None:
            self._ev = torch.cuda.Event()
        for k in range(i0, i1):
            update(k, True)
        # a keyframe shared by two windows keeps the LATER window's pose (sequential order of track()): window_update wrote each
        # own window's rows for its counts; the table of the whole step goes in again so that every rank ends with the same mirror
        kf.w2c[ta:tb].copy_(self._rows_pinned[:tb - ta], non_blocking=True)
        gather_store(ranges[0][0] // 5, n, 1)          # (optional) depth rows, after the last local write to them
        self._counts_many_host[:nown].copy_(self._counts_many[:nown], non_blocking=True)
        self._ev.record()
        self._ev.synchronize()                                    # THE round trip of the step's replay
        for k in range(i0, i1):
            t0, t1 = ranges[k]
            all_counts[k, :t1 - t0, :, :t1 - bases[k]] = self._counts_many_host[k - i0, :t1 - t0, :, :t1 - bases[k]]
        total = exchange_counts(all_counts).numpy()

        def finish():
            for k, (t0, t1) in enumerate(ranges):
                b0 = bases[k]
                if cut[k]:
                    if b0 != graph.base:
                        graph.begin_sequence(b0)
                    graph.add_neighborhood_factors(0, 3, r=3)
                graph.window_decide(t0 - b0, t1 - b0, cent[k], total[k, :t1 - t0], H * W, h * w, cut[k])
        if defer_decisions:
            return finish
        finish()
        return None

    def track_batch(self, ranges, after_window=None):
        """several consecutive 6-keyframe windows: ONE batched decoder/head inference, then the reference's sequential
        chaining + graph update window by window (identical results to calling track() per window).
        after_window(t0, t1): called after EACH window's sequential part with the tracker at that window (`self.t1 = t1`) -- the place the
        per-frame loop runs the loop-closure backend (hi2.py:112-121).  A closure rewrites stored pointmaps and poses; the network
        outputs of the later windows of the batch do not depend on them (every window re-initialises state and memory), their
        chaining does and runs afterwards: the results are those of one window at a time."""
        kf = self.keyframes
        self.window_features(ranges[0][0], ranges[-1][1])                                 # ONE batched encode of every new keyframe
        feats = torch.stack([self.window_features(a, b) for a, b in ranges], 0)          # [Wb,6,N,E] (all cached now)
        res = self.model.decode_windows(feats, kf.ht, kf.wd)
        V = ranges[0][1] - ranges[0][0]
        outs = [(res["pts3d_in_self_view"][j * V:(j + 1) * V], res["conf_self"][j * V:(j + 1) * V], res["camera_pose"][j * V:(j + 1) * V])
                for j in range(len(ranges))]
        if after_window is None:
            self.track_many(ranges, outs)
            self.t1 = ranges[-1][1]
            return
        for rng, out in zip(ranges, outs):
            self.track_many([rng], [out])
            self.t1 = rng[1]
            after_window(rng[0], rng[1])

    # ------------------------------------------------------------------ scheduling (track_frontend.py:285-330)
    def run(self, tstamp, last_frame=False, after_window=None):
        kf = self.keyframes
        if not kf.is_initialized and kf.counter.value - 1 == self.warmup:
            t1 = kf.counter.value - 1
            self.track(0, t1, init=True)
            kf.is_initialized = True
            self.t1 = t1
            return False, range(0, t1), 0
        elif kf.is_initialized and self.window_batch > 1:
            Wb = self.window_batch
            if self.t1 < kf.counter.value - 5 * Wb:
                first = self.t1 - 1
                self.track_batch([(first + 5 * j, first + 5 * j + 6) for j in range(Wb)], after_window=after_window)
                t1 = self.t1
                # (with the per-window hook the caller has already had its turn after every window of the batch)
                return (t1 > 10) and after_window is None, range(first, t1), (t1 - 6) // 5
            if last_frame:
                while self.t1 < kf.counter.value - 1:
                    t0 = self.t1 - 1
                    t1 = min(t0 + 6, kf.counter.value - 1)
                    self.track(t0, t1)
                    self.t1 = t1
                    # (the one-window loop tracked -- and handed to the caller -- every full window whose keyframes existed BEFORE the last frame)
                    if after_window is not None and t1 == t0 + 6 and t1 <= kf.counter.value - 2:
                        after_window(t0, t1)
                return False, None, None
            return False, None, None
        elif kf.is_initialized and self.t1 < kf.counter.value - 5:
            t0 = self.t1 - 1
            t1 = kf.counter.value - 1
            self.track(t0, t1)
            self.t1 = t1
            return (t1 > 10), range(t0, t1), t0 // 5
        elif last_frame and kf.is_initialized:
            t0 = self.t1 - 1
            t1 = kf.counter.value - 1
            if t1 > t0:
                self.track(t0, t1)
                self.t1 = t1
            return False, range(t0, t1), t0 // 5
        return False, None, None
