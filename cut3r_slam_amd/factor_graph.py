"""Covisibility graph of the tracker, same interface and edge ORDER as the reference `FactorGraph`
(/root/reference/hislam2/factor_graph.py:17-117, 148-197, 255-341, 503-582).

Design differences (results identical, topology bit-exact):
  * the edge list lives on the host (a Python set + ordered lists): the reference's duplicate filter does two
    `.item()` device syncs per existing and per new edge (factor_graph.py:29-39); here `add()` costs ONE small
    device->host copy (the int32 overlap counts of the new keyframe).
  * the reprojection tests are two HIP kernels over pointmaps that already sit in HBM (ops.overlap_fwd / overlap_bwd);
    nothing is re-uploaded and no [B,N,4] intermediates are materialised.
`ii`, `jj`, `age` are exposed as int64 tensors on `device`, rebuilt lazily.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from . import geom_host as gh
from . import ops


class AlignedPoints:
    """A chained full-resolution pointmap P*(s*pts) that is never materialised: `pts` [H,W,3] are the network's
    camera-frame points, P the 3x4 chained c2w (12 floats), s the window scale (track_frontend.py:234,259)."""

    def __init__(self, pts, P12, s):
        self.pts, self.P12, self.s = pts, [float(v) for v in np.asarray(P12).reshape(-1)], float(s)
        self.shape = tuple(pts.shape)


class SubmapStore:
    """Keyframe-ordered view of the resident [submap][6 slots][h][w][3] store: keyframe j < n lives at slot
    (j//5)*6 + j%5 (track_frontend.py:251-255) -- what the reference gathers into `all_pointmaps` with a copy."""

    def __init__(self, submap_ds, n):
        self.store, self.n = submap_ds, int(n)
        self.shape = (self.n,) + tuple(submap_ds.shape[2:])


class HipOverlapBackend:
    """Default (and only product) backend: counts from the gfx950 kernels.  Raises if the library is missing."""

    def __init__(self, device):
        self.device = torch.device(device)

    def fwd(self, pointmap, w2c_rows, K4, W, H):
        B = w2c_rows.shape[0]
        cnt = torch.empty(B, dtype=torch.int32, device=self.device)
        if isinstance(pointmap, AlignedPoints):
            ops.overlap_fwd(pointmap.pts.contiguous(), w2c_rows, K4, W, H, cnt, pointmap.P12, pointmap.s)
        else:
            ops.overlap_fwd(pointmap.contiguous(), w2c_rows, K4, W, H, cnt)
        return cnt

    def bwd(self, pointmaps, w2c_row, K4, W, H):
        if isinstance(pointmaps, SubmapStore):
            B = pointmaps.n
            cnt = torch.empty(B, dtype=torch.int32, device=self.device)
            ops.overlap_bwd(pointmaps.store, w2c_row, K4, W, H, cnt, B=B, N=W * H, grp=5, grp_stride=6)
        else:
            B = pointmaps.shape[0]
            cnt = torch.empty(B, dtype=torch.int32, device=self.device)
            ops.overlap_bwd(pointmaps.contiguous(), w2c_row, K4, W, H, cnt)
        return cnt


class FactorGraph:
    def __init__(self, keyframes, device="cuda:0", max_factors=-1, backend=None):
        self.keyframes = keyframes
        self.device = device
        self.max_factors = max_factors
        self._ii, self._jj, self._born = [], [], []     # age of an edge = adds since its insertion = _epoch - _born (O(1) ageing)
        self._epoch = 0
        self._eset = set()
        self._cache = None
        self.base, self.closed = 0, []                  # sequence cuts (begin_sequence); base 0 and no cut = the reference's graph
        self.backend = backend if backend is not None else HipOverlapBackend(device)

    @property
    def _age(self):
        """ages as the reference keeps them (age += 1 on every edge per add(), factor_graph.py:197)"""
        return (self._epoch - np.asarray(self._born, dtype=np.int64)).tolist()

    # ------------------------------------------------------------------ tensor views (reference attribute names)
    def _tensors(self):
        if self._cache is None:
            mk = lambda v: torch.as_tensor(np.asarray(v, dtype=np.int64), dtype=torch.long, device=self.device)
            self._cache = (mk(self._ii), mk(self._jj), mk(self._age))
        return self._cache

    @property
    def ii(self):
        return self._tensors()[0]

    @property
    def jj(self):
        return self._tensors()[1]

    @property
    def age(self):
        return self._tensors()[2]

    def edges_numpy(self):
        return (np.asarray(self._ii, np.int64), np.asarray(self._jj, np.int64), np.asarray(self._age, np.int64))

    # ------------------------------------------------------------------ edge maintenance
    @staticmethod
    def _as_list(x):
        if isinstance(x, list):                      # (the graph's own callers pass int lists: no numpy round trip)
            return x
        if isinstance(x, torch.Tensor):
            return [int(v) for v in x.reshape(-1).tolist()]
        if isinstance(x, np.ndarray) and x.dtype.kind in "iu":
            return x.reshape(-1).tolist()
        return [int(v) for v in np.asarray(x).reshape(-1).tolist()]

    def add_factors(self, ii, jj, remove=False):
        """factor_graph.py:59-81: drop pairs already present (NOT duplicates inside the new batch), append in order."""
        ii, jj = self._as_list(ii), self._as_list(jj)
        new = [(i, j) for i, j in zip(ii, jj) if (i, j) not in self._eset]
        if not new:
            return
        if self.max_factors > 0 and len(self._ii) + len(new) > self.max_factors and remove:
            order = np.argsort(np.asarray(self._age), kind="stable")
            ix = np.arange(len(self._age))[order]
            self.rm_factors(ix >= self.max_factors - len(new))
        for i, j in new:
            self._ii.append(i)
            self._jj.append(j)
            self._born.append(self._epoch)
            self._eset.add((i, j))
        self._cache = None

    def rm_factors(self, mask, store=False):
        m = mask.cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
        keep = ~m.astype(bool)
        self._ii = [v for v, k in zip(self._ii, keep) if k]
        self._jj = [v for v, k in zip(self._jj, keep) if k]
        self._born = [v for v, k in zip(self._born, keep) if k]
        self._eset = set(zip(self._ii, self._jj))
        self._cache = None

    def clear_edges(self):
        self.rm_factors(np.ones(len(self._ii), bool))

    def begin_sequence(self, base):
        """Throughput drivers that cut one long stream into consecutive sequences (TrackFrontend.sequence_windows): the edges of the
        finished sequence are archived with ABSOLUTE keyframe indices and the graph starts empty; from here on this graph's indices
        are relative to keyframe `base` (the cut keyframe = index 0 of the new sequence)."""
        ii, jj, age = self.edges_numpy()
        self.closed.append((self.base, ii + self.base, jj + self.base, age))
        self._ii, self._jj, self._born = [], [], []
        self._eset = set()
        self._epoch = 0
        self._cache = None
        self.base = int(base)

    def edges_absolute(self):
        """(ii, jj) of every sequence so far, absolute keyframe indices, in insertion order"""
        ii, jj, _ = self.edges_numpy()
        parts_i = [c[1] for c in self.closed] + [ii + self.base]
        parts_j = [c[2] for c in self.closed] + [jj + self.base]
        return np.concatenate(parts_i), np.concatenate(parts_j)

    def add_neighborhood_factors(self, t0, t1, r=3):
        """factor_graph.py:109-117 (row-major meshgrid order)."""
        ii, jj = [], []
        for i in range(t0, t1):
            for j in range(t0, t1):
                if 0 < abs(i - j) <= r:
                    ii.append(i)
                    jj.append(j)
        self.add_factors(ii, jj)

    # ------------------------------------------------------------------ reprojection overlap
    @staticmethod
    def _K4(K):
        K = np.asarray(K, dtype=np.float64)
        if K.shape == (3, 3):
            return [float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])]
        return [float(v) for v in K.reshape(-1)[:4]]

    def cal_overlap_batch(self, pointmap_i, T_j_batch, K, w2c_rows=None):
        """factor_graph.py:255-282 -> fp32 ratios [B] (device).  T_j_batch: c2w [B,4,4] (tensor/ndarray)."""
        H, W, _ = pointmap_i.shape
        if w2c_rows is None:
            c2w = T_j_batch.detach().cpu().numpy() if isinstance(T_j_batch, torch.Tensor) else np.asarray(T_j_batch)
            w2c_rows = torch.from_numpy(gh.w2c_rows(c2w)).to(self.device)
        cnt = self.backend.fwd(pointmap_i, w2c_rows.contiguous(), self._K4(K), W, H)
        return cnt.float() / float(H * W)

    def cal_overlap_bi(self, pointmap_i, T_j_batch, K, w2c_row=None):
        """factor_graph.py:284-315 with B2 == 1 (its only call shape) -> fp32 ratios [B1,1] (device)."""
        B1, H, W, _ = pointmap_i.shape
        if w2c_row is None:
            c2w = T_j_batch.detach().cpu().numpy() if isinstance(T_j_batch, torch.Tensor) else np.asarray(T_j_batch)
            if c2w.reshape(-1, 4, 4).shape[0] != 1:
                raise NotImplementedError("cal_overlap_bi: one target camera (the reference never passes more)")
            w2c_row = torch.from_numpy(gh.w2c_rows(c2w)[0]).to(self.device)
        cnt = self.backend.bwd(pointmap_i, w2c_row.contiguous(), self._K4(K), W, H)
        return (cnt.float() / float(H * W)).reshape(B1, 1)

    def add_begin(self, current_idx, all_poses, all_pointmaps, current_pose, current_pointmap, K,
                  all_w2c_rows=None, current_w2c_row=None):
        """Launch half of `add`: host distance classes + the two counting kernels, nothing read back.  Returns the
        ticket `add_finish` consumes.  Lets a caller issue the launches of several keyframes before ONE read-back."""
        c2w = all_poses.detach().cpu().numpy() if isinstance(all_poses, torch.Tensor) else np.asarray(all_poses)
        cur = current_pose.detach().cpu().numpy() if isinstance(current_pose, torch.Tensor) else np.asarray(current_pose)
        c2w, cur = c2w.astype(np.float32, copy=False), cur.astype(np.float32, copy=False)
        if c2w.ndim == 2 and c2w.shape[1] == 3:
            # camera centres only ([n,3] / [3]): the caller supplies the world->camera rows, nothing else is needed here
            if all_w2c_rows is None or current_w2c_row is None:
                raise ValueError("add_begin: camera centres alone need all_w2c_rows and current_w2c_row")
            centres, cur_c = c2w, cur.reshape(3)
        else:
            c2w = c2w.reshape(-1, 4, 4)
            cur = cur.reshape(4, 4)
            centres, cur_c = c2w[:, :3, 3], cur[:3, 3]
        d = centres - cur_c[None]
        dists = np.sqrt((d * d).sum(axis=1, dtype=np.float32), dtype=np.float32)
        cond1 = dists <= np.float32(1.0)
        n = centres.shape[0]
        if all_w2c_rows is None:
            all_w2c_rows = torch.from_numpy(gh.w2c_rows(c2w)).to(self.device)
        if current_w2c_row is None:
            current_w2c_row = torch.from_numpy(gh.w2c_rows(cur[None])[0]).to(self.device)
        H, W, _ = current_pointmap.shape
        K4 = self._K4(K)
        # ONE forward launch over every previous camera and ONE backward stream over every previous pointmap (the
        # reference launches per distance class and gathers copies; counts are per camera, so the decisions are the same)
        cnt_f = self.backend.fwd(current_pointmap, all_w2c_rows[:n].contiguous(), K4, W, H)
        idx2 = np.nonzero(~cond1)[0]
        cnt_b = None
        hb, wb = all_pointmaps.shape[1], all_pointmaps.shape[2]
        if idx2.size:
            cnt_b = self.backend.bwd(all_pointmaps, current_w2c_row.contiguous(), K4, wb, hb)
        return {"idx": current_idx, "n": n, "cnt_f": cnt_f, "cnt_b": cnt_b, "idx1": np.nonzero(cond1)[0], "idx2": idx2,
                "npix_f": H * W, "npix_b": hb * wb}

    def window_tickets(self, t0, t1, centres, counts_host, npix_f, npix_b, first=3):
        """tickets + counts of keyframes t0..t1-1 from ONE cut3r_window_update call (ops.window_update): the host half
        of add_begin (distance classes from the camera centres [>= t1, 3]) and the slices of counts_host [V,2,ldc]."""
        out = []
        for i in range(max(t0, first), t1):
            d = centres[:i] - centres[i][None]
            dists = np.sqrt((d * d).sum(axis=1, dtype=np.float32), dtype=np.float32)
            cond1 = dists <= np.float32(1.0)
            tk = {"idx": i, "n": i, "cnt_f": None, "cnt_b": None, "idx1": np.nonzero(cond1)[0], "idx2": np.nonzero(~cond1)[0],
                  "npix_f": npix_f, "npix_b": npix_b}
            out.append((tk, counts_host[i - t0, 0, :i], counts_host[i - t0, 1, :i]))
        return out

    def window_decide(self, r0, r1, centres, counts, npix_f, npix_b, init, first=3):
        """The decisions of a whole window in one pass: what `window_tickets` + `add_neighborhood_factors` + `add_finish` do keyframe
        by keyframe (factor_graph.py:109-117, 170-197), with the distance classes and ratio tests of the window's keyframes evaluated
        as ONE set of array operations (the multi-GPU replay decides hundreds of windows per step on the host).  Same float32
        operations per element, same insertion order, same ages.  counts: int32 [V, 2, >= r1] (forward | backward rows)."""
        V = r1 - r0
        c = np.asarray(centres[:r1], np.float32)
        d = c[None, :, :] - c[r0:r1][:, None, :]
        dists = np.sqrt((d * d).sum(axis=2, dtype=np.float32), dtype=np.float32)
        cond1 = dists <= np.float32(1.0)
        hit_f = (counts[:V, 0, :r1].astype(np.float32) / np.float32(npix_f)) > np.float32(0.3)
        hit_b = (counts[:V, 1, :r1].astype(np.float32) / np.float32(npix_b)) > np.float32(0.3)
        sel1 = cond1 & hit_f
        sel2 = ~cond1 & (hit_f | hit_b)
        for v in range(V):
            i = r0 + v
            if not init:
                self.add_neighborhood_factors(i - 3, i + 1, r=3)
            if i >= first:
                for sel in (sel1, sel2):
                    jj = np.nonzero(sel[v, :i])[0]
                    if jj.size:
                        jl = jj.tolist()
                        il = [i] * len(jl)
                        self.add_factors(il, jl)
                        self.add_factors(jl, il)
                self._epoch += 1
                self._cache = None

    @staticmethod
    def read_counts(tickets):
        """single device->host hop for the counts of any number of tickets -> [(cf, cb|None)]"""
        parts = []
        for tk in tickets:
            parts.append(tk["cnt_f"].reshape(-1))
            if tk["cnt_b"] is not None:
                parts.append(tk["cnt_b"].reshape(-1))
        if not parts:
            return []
        host = (torch.cat(parts) if len(parts) > 1 else parts[0]).cpu().numpy()
        res, o = [], 0
        for tk in tickets:
            n = tk["n"]
            cf = host[o:o + n]
            o += n
            cb = None
            if tk["cnt_b"] is not None:
                cb = host[o:o + n]
                o += n
            res.append((cf, cb))
        return res

    def add_finish(self, tk, cf, cb):
        """decision half of `add` (factor_graph.py:170-197), host only"""
        current_idx, idx1, idx2 = tk["idx"], tk["idx1"], tk["idx2"]
        ratio_f = cf.astype(np.float32) / np.float32(tk["npix_f"])
        if idx1.size:
            jj = idx1[ratio_f[idx1] > np.float32(0.3)]
            if jj.size:
                ii = np.full_like(jj, current_idx)
                self.add_factors(ii, jj)
                self.add_factors(jj, ii)
        if idx2.size:
            ratio_b = cb[idx2].astype(np.float32) / np.float32(tk["npix_b"])
            mask = (ratio_f[idx2] > np.float32(0.3)) | (ratio_b > np.float32(0.3))
            jj = idx2[mask]
            if jj.size:
                ii = np.full_like(jj, current_idx)
                self.add_factors(ii, jj)
                self.add_factors(jj, ii)
        self._epoch += 1
        self._cache = None

    def add(self, current_idx, all_poses, all_pointmaps, current_pose, current_pointmap, K,
            all_w2c_rows=None, current_w2c_row=None):
        """factor_graph.py:148-197.  all_poses [i,4,4] c2w, all_pointmaps [i,h,w,3], current_pose [4,4],
        current_pointmap [H,W,3] (tensors, or the AlignedPoints / SubmapStore descriptors of resident data);
        optional precomputed world->camera rows avoid any host inverse."""
        tk = self.add_begin(current_idx, all_poses, all_pointmaps, current_pose, current_pointmap, K,
                            all_w2c_rows, current_w2c_row)
        (cf, cb), = self.read_counts([tk])
        self.add_finish(tk, cf, cb)

    # ------------------------------------------------------------------ loop detection (factor_graph.py:503-543)
    def detect_loop(self, current_idx, current_featI=None, all_featI=None, temporal_window=8, feat_th=0.7,
                    small_loop_candidates=False):
        covisible = set(j for i, j in zip(self._ii, self._jj) if i == current_idx)
        cand = [i for i in covisible if abs(i - current_idx) > temporal_window]
        if cand:
            return min(cand) if small_loop_candidates else np.array(cand)
        return None
