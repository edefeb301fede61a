"""TEST INFRASTRUCTURE ONLY -- end-to-end CPU restatement of the reference tracking loop, frame in -> trajectory out.
Only tests/, __graft_entry__.smoke() and bench.py's bounded `cpu_baseline` / trajectory-parity legs may import it.

It composes the other oracle pieces into the per-frame loop of the reference WITHOUT touching any product code:

  Hi2.run                 /root/reference/hislam2/hi2.py:101-133 (GS mapper off; loop closure when `iteration` > 0)
  TrackBackend.run/track  /root/reference/hislam2/track_backend.py:527-586, 137-217 (detect_loop = factor_graph.py:503-543, NMS = :561-582
                          with :328-341 and cal_overlap_bi :286-315; the two optimisers and the store rewrite = oracle.lc_oracle.LoopCloser,
                          fp64 autograd with matrix_exp in place of the absent lietorch: parity unpinned, see its header)
  MotionFilter.kfFilter   /root/reference/hislam2/motion_filter.py:70-135  (encoder = oracle.cut3r_oracle.encode_image,
                          overlap ratio = hislam2/util/utils.py:726-736 restated below in fp32)
  TrackFrontend.run/track /root/reference/hislam2/track_frontend.py:285-330, 166-262 (network = cut3r_oracle.forward_views,
                          chaining = oracle.slam_oracle.track_window)
  FactorGraph             /root/reference/hislam2/factor_graph.py:29-39, 59-81, 109-117, 148-197, 255-315 as `RefGraph`
                          (ordered python edge lists; reprojection counts from oracle_geom.c, which is pinned to the
                          reference's cal_overlap_* by tests/golden/graph.npz)
  save_trajectory         /root/reference/demo_s.py:97-109 (keyframe poses up to counter-1)

Pinned pieces: the network (model_tiny_*.npz), the overlap counts and the edge bookkeeping (graph.npz, checked for
RefGraph in tests/test_graph_cpu.py).  The tracker drivers themselves hard-code 'cuda' in the reference and cannot run
here: that composition is read-faithful, parity unpinned.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import cut3r_oracle as O
from . import geom as G
from . import lc_oracle as LC
from . import slam_oracle as SO


# ---------------------------------------------------------------------------------------------------------------- graph
class RefGraph:
    """factor_graph.py:17-197 with python lists for ii / jj / age (the reference keeps int64 device tensors)."""

    def __init__(self):
        self.ii, self.jj, self.age = [], [], []
        self.ratios = {}                    # (i, j) -> (forward ratio, backward ratio | None) of every tested pair, for diagnostics

    def add_factors(self, ii, jj):
        """:59-81 -- pairs already in the graph are dropped (:29-39; duplicates INSIDE the new batch are kept), the rest
        appended in order with age 0.  (`remove` defaults to False at every call site, so max_factors never trims.)"""
        eset = set(zip(self.ii, self.jj))
        for i, j in zip(ii, jj):
            if (int(i), int(j)) not in eset:
                self.ii.append(int(i))
                self.jj.append(int(j))
                self.age.append(0)

    def add_neighborhood_factors(self, t0, t1, r=3):
        """:109-117, meshgrid(indexing='ij') order"""
        ii, jj = [], []
        for i in range(t0, t1):
            for j in range(t0, t1):
                if 0 < abs(i - j) <= r:
                    ii.append(i)
                    jj.append(j)
        self.add_factors(ii, jj)

    def add(self, current_idx, all_c2w, all_pointmaps, current_c2w, current_pointmap, K4):
        """:148-197.  all_c2w [i,4,4], all_pointmaps [i,h,w,3], current_pointmap [H,W,3] (numpy fp32), K4 = fx,fy,cx,cy."""
        all_c2w = np.asarray(all_c2w, np.float32)
        cur = np.asarray(current_c2w, np.float32)
        d = all_c2w[:, :3, 3] - cur[None, :3, 3]
        dists = np.sqrt((d * d).sum(axis=1, dtype=np.float32), dtype=np.float32)      # torch.norm in fp32
        cond1 = dists <= np.float32(1.0)
        H, W, _ = current_pointmap.shape
        w2c_all = G.w2c_rows(all_c2w)
        for cond, bidir in ((cond1, False), (~cond1, True)):
            idx = np.nonzero(cond)[0]
            if idx.size == 0:
                continue
            ratio = G.overlap_fwd(current_pointmap, w2c_all[idx], K4, W, H).astype(np.float32) / np.float32(H * W)
            mask = ratio > np.float32(0.3)
            rb = None
            if bidir:
                hb, wb = all_pointmaps.shape[1:3]
                rb = G.overlap_bwd(all_pointmaps[idx], G.w2c_rows(cur[None])[0], K4, wb, hb).astype(np.float32) / np.float32(hb * wb)
                mask = mask | (rb > np.float32(0.3))
            for n, j in enumerate(idx):
                self.ratios[(current_idx, int(j))] = (float(ratio[n]), float(rb[n]) if rb is not None else None)
            jj = idx[mask]
            if jj.size:
                ii = [current_idx] * len(jj)
                self.add_factors(ii, jj.tolist())
                self.add_factors(jj.tolist(), ii)
        self.age = [a + 1 for a in self.age]

    def detect_loop(self, current_idx, temporal_window=8):
        """:503-543 (the live part): keyframes covisible with `current_idx` and more than `temporal_window` keyframes away"""
        cov = set(j for i, j in zip(self.ii, self.jj) if i == current_idx)
        cand = [i for i in cov if abs(i - current_idx) > temporal_window]
        return np.array(cand) if cand else None

    def edges_numpy(self):
        return np.asarray(self.ii, np.int64), np.asarray(self.jj, np.int64), np.asarray(self.age, np.int64)


# ---------------------------------------------------------------------------------------------------------------- filter
def patch_overlap_ratio_f32(feat0: torch.Tensor, feat1: torch.Tensor, threshold: float = 0.7) -> float:
    """hislam2/util/utils.py:726-736, torch CPU fp32 (the reference evaluates it under fp16 autocast on the GPU)."""
    f0 = torch.nn.functional.normalize(feat0[1:], dim=1)
    f1 = torch.nn.functional.normalize(feat1[1:], dim=1)
    sim = f0 @ f1.T
    return float((sim.max(dim=1)[0] > threshold).float().mean().item())


# ---------------------------------------------------------------------------------------------------------------- driver
class SlamOracle:
    """Frame-by-frame CPU tracker.  `run(t, image_u8, intr4, second_last_frame, last_frame)` mirrors Hi2.run."""

    def __init__(self, cfg, sd, image_size, buffer, motion_filter, intrinsics_ds=2, precision="fp32", iteration=0, lc_dtype=torch.float64):
        self.cfg, self.sd = cfg, sd
        H, W = image_size
        self.H, self.W, self.ds = H, W, 2
        self.thresh = motion_filter["thresh"]
        self.skip = motion_filter.get("skip", 1)
        self.kf_every = motion_filter.get("kf_every", -1)
        self.precision = precision
        nsub = buffer // 5 + 1
        self.state = {"pose": torch.zeros(buffer, 7), "depth": torch.ones(buffer, H, W),
                      "submap_ds": torch.ones(nsub, 6, H // 2, W // 2, 3), "conf_ds": torch.zeros(nsub, 6, H // 2, W // 2)}
        self.state["pose"][:, 6] = 1
        self.tstamp = np.zeros(buffer)
        self.image = torch.zeros(buffer, 3, H, W, dtype=torch.uint8)
        self.intrinsic = np.zeros((buffer, 4), np.float32)
        self.featI = [None] * buffer
        self.counter = 0
        self.is_initialized = False
        self.t1 = 0
        self.warmup = 6
        self.graph = RefGraph()
        self.windows = []                   # (t0, t1, init) in execution order
        self.ratios = []                    # (tstamp, overlap ratio) of every tested frame
        self.timing = {"encode_s": 0.0, "window_s": 0.0, "align_graph_s": 0.0, "encodes": 0, "windows": 0, "graph_adds": 0}
        # loop closure (hi2.py:44-49,112-121: the backend runs when Tracking.frontend.iteration > 0, every other eligible window)
        self.iteration = int(iteration)
        self.lc_dtype = lc_dtype
        self.freeze_counter = 0
        self.closer = None                  # oracle.lc_oracle.LoopCloser, created at the first closure
        self.closures = []                  # dicts: idx_current, idx_matched, candidates, scores, at_keyframe, poses / submaps after it

    # -- keyframe store (keyframe.py:43-77,105-107)
    def _append(self, tstamp, image, intr, feat):
        i = self.counter
        self.tstamp[i] = tstamp
        self.image[i] = image
        self.intrinsic[i] = np.asarray(intr, np.float32).reshape(-1)[:4] if intr is not None else self.intrinsic[0]
        self.featI[i] = feat
        self.counter += 1

    def _encode(self, image_u8):
        tic = time.perf_counter()
        with O.matmul_precision(self.precision):
            feat, _ = O.encode_image(self.cfg, self.sd, O.normalize(image_u8[None]))
        self.timing["encode_s"] += time.perf_counter() - tic
        self.timing["encodes"] += 1
        return feat[0]

    def kf_filter(self, tstamp, image_u8, intr, second_last_frame=False, last_frame=False):
        """motion_filter.py:70-135.  Returns True when the frame became a keyframe."""
        compute_overlap = not (self.kf_every > 0)
        if self.counter == 0 or last_frame or second_last_frame:
            self._append(tstamp, image_u8, intr, self._encode(image_u8))
            return True
        ratio, feat1 = 1.0, None
        if compute_overlap and tstamp % self.skip == 0:
            feat1 = self._encode(image_u8)
            ratio = patch_overlap_ratio_f32(self.featI[self.counter - 1], feat1)
            self.ratios.append((tstamp, ratio))
        elif not compute_overlap and tstamp % self.kf_every == 0:
            feat1 = self._encode(image_u8)
        if (compute_overlap and ratio < self.thresh) or (not compute_overlap and tstamp % self.kf_every == 0):
            self._append(tstamp, image_u8, intr, feat1)
            return True
        return False

    # -- track_frontend.py:166-262
    def track(self, t0, t1, init=False):
        tic = time.perf_counter()
        with O.matmul_precision(self.precision):
            preds = O.forward_views(self.cfg, self.sd, O.normalize(self.image[t0:t1]), minimal=True)
        self.timing["window_s"] += time.perf_counter() - tic
        self.timing["windows"] += 1
        tic = time.perf_counter()
        pts = torch.cat([p["pts3d_in_self_view"] for p in preds], 0)
        conf = torch.cat([p["conf_self"] for p in preds], 0)
        enc = torch.cat([p["camera_pose"] for p in preds], 0)
        st = self.state
        if init:
            self.graph.add_neighborhood_factors(0, 3, r=3)
        # per-view results of the chaining; the stores are then written view by view so that graph.add(i) sees exactly the
        # stores the reference sees at that point of its loop (:246-258: keyframes < i only)
        tmp = {k: v.clone() for k, v in st.items()}
        full = SO.track_window(tmp, t0, t1, pts, conf, enc, init, self.ds)
        sub = t0 // 5
        for i in range(t0, t1):
            if not init:
                self.graph.add_neighborhood_factors(i - 3, i + 1, r=3)
            v = i - t0
            st["submap_ds"][sub, v] = tmp["submap_ds"][sub, v]
            st["conf_ds"][sub, v] = tmp["conf_ds"][sub, v]
            st["pose"][i] = tmp["pose"][i]
            st["depth"][i] = tmp["depth"][i]
            if i > 2:
                all_c2w = SO.pose_vec_to_matrix(st["pose"][:i]).numpy()
                cur_c2w = SO.pose_vec_to_matrix(st["pose"][i][None])[0].numpy()
                if sub > 0:
                    all_pm = torch.cat([st["submap_ds"][:sub, :-1].reshape(-1, self.H // 2, self.W // 2, 3), st["submap_ds"][sub, :v]], 0)
                else:
                    all_pm = st["submap_ds"][sub, :v]
                self.graph.add(i, all_c2w, all_pm.numpy(), cur_c2w, full[v][2].numpy(), self.intrinsic[i])
                self.timing["graph_adds"] += 1
        self.timing["align_graph_s"] += time.perf_counter() - tic
        self.windows.append((t0, t1, init))

    # -- track_frontend.py:285-330 (returns its `run_backend` flag: a steady-state window beyond keyframe 10 was tracked)
    def tracker_run(self, last_frame=False):
        if not self.is_initialized and self.counter - 1 == self.warmup:
            t1 = self.counter - 1
            self.track(0, t1, init=True)
            self.is_initialized = True
            self.t1 = t1
        elif self.is_initialized and self.t1 < self.counter - 5:
            t0, t1 = self.t1 - 1, self.counter - 1
            self.track(t0, t1)
            self.t1 = t1
            return t1 > 10
        elif last_frame:
            # (the reference calls track() here even when nothing is pending or nothing was initialised; both cases
            # raise there -- an empty window / a never-initialised tracker -- so they are not part of the compared runs)
            t0, t1 = self.t1 - 1, self.counter - 1
            if self.is_initialized and t1 > t0:
                self.track(t0, t1)
                self.t1 = t1

        return False

    # -- track_backend.py:561-582 + 328-341 + 286-315
    def nms_scores(self, ids_matched, idx_current, K4):
        st = self.state
        h, w = st["submap_ds"].shape[2:4]
        ids = np.asarray(ids_matched)
        pm_m = st["submap_ds"][ids // 5, ids % 5].numpy()
        pm_c = st["submap_ds"][idx_current // 5, idx_current % 5].numpy()
        c2w_m = SO.pose_vec_to_matrix(st["pose"][ids]).numpy()
        c2w_c = SO.pose_vec_to_matrix(st["pose"][idx_current][None]).numpy()
        a2c = G.overlap_bwd(pm_m, G.w2c_rows(c2w_c)[0], K4, w, h).astype(np.float32) / np.float32(h * w)
        c2a = G.overlap_fwd(pm_c, G.w2c_rows(c2w_m), K4, w, h, clamp_z=False).astype(np.float32) / np.float32(h * w)
        feat = np.array([patch_overlap_ratio_f32(self.featI[idx_current], self.featI[int(i)]) for i in ids], np.float32)
        return np.float32(0.8) * ((a2c + c2a) / np.float32(2)) + np.float32(0.2) * feat

    # -- track_backend.py:137-217: the submap [5 keyframes of the matched submap, current keyframe] chained to the anchor keyframe
    def backend_track(self, selected, anchor_sub):
        with O.matmul_precision(self.precision):
            preds = O.forward_views(self.cfg, self.sd, O.normalize(self.image[selected]), minimal=True)
        pts = torch.cat([p["pts3d_in_self_view"] for p in preds], 0)
        conf = torch.cat([p["conf_self"] for p in preds], 0)
        enc = torch.cat([p["camera_pose"] for p in preds], 0)
        st, a = self.state, anchor_sub * 5
        tmp = {"pose": st["pose"][a:a + 1].clone().repeat(6, 1), "depth": st["depth"][a:a + 1].clone().repeat(6, 1, 1),
               "submap_ds": torch.zeros(1, 6, self.H // 2, self.W // 2, 3), "conf_ds": torch.zeros(1, 6, self.H // 2, self.W // 2)}
        SO.track_window(tmp, 0, len(selected), pts, conf, enc, False, self.ds)      # same chaining arithmetic as the front end (:166-199)
        return tmp["submap_ds"][0], tmp["conf_ds"][0], tmp["pose"]

    # -- track_backend.py:527-586
    def backend_run(self):
        K4 = (self.intrinsic[0] / np.float32(self.ds)).astype(np.float32)
        t1 = self.counter - 1
        t0 = t1 - 6
        ids, idx_current = None, None
        for idx_current in range(t0, t1 - 1):
            ids = self.graph.detect_loop(idx_current)
            if ids is not None:
                break
        if ids is None:
            return False
        scores = self.nms_scores(ids, idx_current, K4)
        if not float(scores.max()) > 0.4:
            return False
        idx_matched = int(ids[int(np.argmax(scores))])
        anchor = idx_matched // 5
        selected = list(range(anchor * 5, anchor * 5 + 5)) + [idx_current]
        pm_lc, _, _ = self.backend_track(selected, anchor)
        st = self.state
        if self.closer is None:
            self.closer = LC.LoopCloser(st["submap_ds"], st["conf_ds"], st["pose"], self.iteration, dtype=self.lc_dtype)
        lc = self.closer
        lc.sub, lc.conf, lc.pose = st["submap_ds"].double(), st["conf_ds"].double(), st["pose"].double()
        lc.close(pm_lc, idx_matched, idx_current)
        st["submap_ds"], st["pose"] = lc.sub.float(), lc.pose.float()
        n_sub = idx_current // 5 + 1
        self.closures.append({"idx_current": idx_current, "idx_matched": idx_matched, "candidates": np.asarray(ids).copy(), "scores": scores.copy(),
                              "at_keyframe": self.counter, "pose": st["pose"][:n_sub * 5 + 1].clone(), "submap_ds": st["submap_ds"][:n_sub].clone(),
                              "loss": lc.losses[-1]})
        return True

    def run(self, tstamp, image_u8, intr, second_last_frame=False, last_frame=False):
        """hi2.py:101-121"""
        self.kf_filter(tstamp, image_u8, intr, second_last_frame, last_frame)
        run_backend = self.tracker_run(last_frame)
        if run_backend and not last_frame and self.iteration > 0:
            if self.freeze_counter > 0:
                if self.backend_run():
                    self.freeze_counter = 0
            else:
                self.freeze_counter += 1

    def trajectory(self):
        """demo_s.py:97-100: rows [tstamp, tx,ty,tz, qx,qy,qz,qw] of keyframes 0..counter-2"""
        t = self.counter - 1
        return np.concatenate([self.tstamp[:t, None], self.state["pose"][:t].numpy().astype(np.float64)], 1)


def run_stream(cfg, sd, frames_u8, intr, motion_filter, precision="fp32", buffer=None, mark_tail=True, iteration=0, lc_dtype=torch.float64):
    """Drive SlamOracle over frames_u8 [n,3,H,W] like demo_s.py:151-160 (second-last / last frame flags) and return it."""
    n = frames_u8.shape[0]
    H, W = frames_u8.shape[2:]
    so = SlamOracle(cfg, sd, (H, W), buffer or (n + 8), motion_filter, precision=precision, iteration=iteration, lc_dtype=lc_dtype)
    for t in range(n):
        so.run(t, frames_u8[t], intr, second_last_frame=mark_tail and t == n - 2, last_frame=mark_tail and t == n - 1)
    return so
