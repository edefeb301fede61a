"""Sim(3)-aligned absolute trajectory error of a `traj_kf.txt` against a TUM-format ground truth -- the number the
reference's run scripts read from `evo_ape tum <gt> traj_kf.txt -vas` (/root/reference/scripts/run_scannet.py:34-36).
evo is not a dependency here: timestamps are associated by nearest neighbour (evo's default max difference 0.01 s),
positions aligned with Umeyama's closed form (rotation, translation, scale), RMSE over the translation residuals.
Host-side numpy, O(#keyframes)."""
from __future__ import annotations

import numpy as np


def load_tum(path: str) -> np.ndarray:
    """rows `stamp tx ty tz qx qy qz qw` (comments with #) -> [n,8]"""
    rows = [list(map(float, line.split()[:8])) for line in open(path) if line.strip() and not line.startswith("#")]
    return np.asarray(rows, np.float64).reshape(-1, 8)


def associate(est: np.ndarray, gt: np.ndarray, max_diff: float = 0.01):
    """indices (i_est, i_gt) of stamp pairs closer than max_diff, each ground-truth stamp used once"""
    order = np.argsort(gt[:, 0])
    gts = gt[order, 0]
    ie, ig, used = [], [], set()
    for i, t in enumerate(est[:, 0]):
        j = int(np.searchsorted(gts, t))
        best = None
        for c in (j - 1, j):
            if 0 <= c < len(gts) and abs(gts[c] - t) <= max_diff and (best is None or abs(gts[c] - t) < abs(gts[best] - t)):
                best = c
        if best is not None and best not in used:
            used.add(best)
            ie.append(i)
            ig.append(int(order[best]))
    return np.asarray(ie, int), np.asarray(ig, int)


def umeyama(src: np.ndarray, dst: np.ndarray, with_scale: bool = True):
    """least-squares (s, R, t) with dst ~ s R src + t  (Umeyama 1991)"""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / (xs ** 2).sum() * len(src)) if with_scale else 1.0
    return s, R, mu_d - s * R @ mu_s


def ate_rmse(est: np.ndarray, gt: np.ndarray, max_diff: float = 0.01, with_scale: bool = True):
    """-> dict(rmse, mean, median, max, n, scale) of the aligned translation error"""
    ie, ig = associate(est, gt, max_diff)
    if len(ie) < 3:
        raise ValueError(f"only {len(ie)} associated poses (max_diff {max_diff})")
    p, q = est[ie, 1:4], gt[ig, 1:4]
    s, R, t = umeyama(p, q, with_scale)
    err = np.linalg.norm((s * (R @ p.T)).T + t - q, axis=1)
    return {"rmse": float(np.sqrt((err ** 2).mean())), "mean": float(err.mean()), "median": float(np.median(err)),
            "max": float(err.max()), "n": int(len(err)), "scale": s}


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser(description="Sim(3)-aligned ATE of traj_kf.txt against a TUM ground truth (evo_ape tum -vas)")
    ap.add_argument("gt")
    ap.add_argument("est")
    ap.add_argument("--max-diff", type=float, default=0.01)
    ap.add_argument("--no-scale", action="store_true", help="SE(3) alignment (evo_ape -va)")
    a = ap.parse_args()
    print(json.dumps(ate_rmse(load_tum(a.est), load_tum(a.gt), a.max_diff, not a.no_scale)))
