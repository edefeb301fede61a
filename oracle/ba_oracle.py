"""TEST INFRASTRUCTURE ONLY -- float64 torch-CPU restatement of the legacy dense-BA math
(/root/reference/hislam2/geom/projective_ops.py:15-74, pinhole.py:6-53, ba.py:32-107, chol.py:47-78,
modules/corr.py:23-39).  PARITY UNPINNED vs the reference: `droid_backends` sources and lietorch are absent and
`geom/ba.py` cannot execute (undefined scatter_sum).  This oracle is pinned by its own finite-difference checks
(tests/test_ba_gpu.py) and solves the UN-REDUCED normal equations densely, so it also checks the Schur path."""
import numpy as np
import torch

from . import lie_oracle as LO

DT = torch.float64


def se3_matrix(data):
    """[n,7] (t, q_xyzw) -> [n,4,4]"""
    return torch.from_numpy(np.stack([LO.data_to_matrix(1, d) for d in np.asarray(data, np.float64)]))


def project(Gij_M, disp, intr_i, intr_j, ht, wd):
    """coords [HW,2], valid [HW], X1 for one edge; Gij_M [4,4] fp64"""
    y, x = torch.meshgrid(torch.arange(ht, dtype=DT), torch.arange(wd, dtype=DT), indexing="ij")
    fx, fy, cx, cy = [float(v) for v in intr_i]
    X0 = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(x), disp.reshape(ht, wd).to(DT)], -1).reshape(-1, 4)
    X1 = X0 @ Gij_M.T
    Z = torch.where(X1[:, 2] < 0.1, torch.ones_like(X1[:, 2]), X1[:, 2])
    d = 1.0 / Z
    fxj, fyj, cxj, cyj = [float(v) for v in intr_j]
    coords = torch.stack([fxj * X1[:, 0] * d + cxj, fyj * X1[:, 1] * d + cyj], -1)
    valid = (X1[:, 2] > 0.2).to(DT)
    return coords, valid


def ba_dense(target, weight, eta, poses7, disps, intr, ii, jj, fixedp, ep=0.1, lm=1e-4, eps=1e-6):
    """Numerical-Jacobian Gauss-Newton step on the same normal equations, solved WITHOUT the Schur reduction.
    Left perturbation of world->camera poses: G <- exp(a) G (retr, ba.py:29).  Returns dx [P-fixedp,6], dz [M,HW]."""
    P, ht, wd = disps.shape
    HW = ht * wd
    N = len(ii)
    G = se3_matrix(poses7)
    kx = np.unique(ii)
    M = len(kx)
    kk = {int(k): m for m, k in enumerate(kx)}
    Pf = P - fixedp
    nx = Pf * 6 + M * HW
    H = torch.zeros(nx, nx, dtype=DT)
    g = torch.zeros(nx, dtype=DT)

    def resid(e, Gi, Gj, d_i):
        Gij = Gj @ torch.linalg.inv(Gi)
        c, valid = project(Gij, d_i, intr[ii[e]], intr[jj[e]], ht, wd)
        return c, valid

    for e in range(N):
        i, j = int(ii[e]), int(jj[e])
        c0, valid = resid(e, G[i], G[j], disps[i])
        r = (target[e].reshape(HW, 2).to(DT) - c0)
        w = 0.001 * valid[:, None] * weight[e].reshape(HW, 2).to(DT)
        # numerical Jacobians of coords wrt left perturbations of G_i, G_j and wrt the disparity of each pixel
        Ji = torch.zeros(HW, 2, 6, dtype=DT)
        Jj = torch.zeros(HW, 2, 6, dtype=DT)
        for a in range(6):
            da = np.zeros(6); da[a] = eps
            Ep, Em = torch.from_numpy(LO.exp_matrix(1, da)), torch.from_numpy(LO.exp_matrix(1, -da))
            Ji[:, :, a] = (resid(e, Ep @ G[i], G[j], disps[i])[0] - resid(e, Em @ G[i], G[j], disps[i])[0]) / (2 * eps)
            Jj[:, :, a] = (resid(e, G[i], Ep @ G[j], disps[i])[0] - resid(e, G[i], Em @ G[j], disps[i])[0]) / (2 * eps)
        Jz = (resid(e, G[i], G[j], disps[i] + eps)[0] - resid(e, G[i], G[j], disps[i] - eps)[0]) / (2 * eps)   # [HW,2]
        m = kk[i]
        zi = Pf * 6 + m * HW + torch.arange(HW)
        ip, jp = i - fixedp, j - fixedp
        blocks = []
        if ip >= 0:
            blocks.append((ip * 6, Ji))
        if jp >= 0:
            blocks.append((jp * 6, Jj))
        for (o1, J1) in blocks:
            g[o1:o1 + 6] += torch.einsum("kc,kca,kc->a", w, J1, r)
            for (o2, J2) in blocks:
                H[o1:o1 + 6, o2:o2 + 6] += torch.einsum("kc,kca,kcb->ab", w, J1, J2)
            Ez = torch.einsum("kc,kca,kc->ka", w, J1, Jz)            # [HW,6]
            H[o1:o1 + 6][:, zi] += Ez.T
            H[zi[:, None], torch.arange(o1, o1 + 6)[None]] += Ez
        H[zi, zi] += (w * Jz * Jz).sum(-1)
        g[zi] += (w * r * Jz).sum(-1)
    # damping exactly as chol.py:56-57 on the pose block, eta + 1e-7 on the depth diagonal (ba.py:92)
    idx = torch.arange(Pf * 6)
    H[idx, idx] = H[idx, idx] + (ep + lm * H[idx, idx])
    zi = torch.arange(Pf * 6, nx)
    H[zi, zi] += eta.reshape(-1).to(DT) + 1e-7
    sol = torch.linalg.solve(H, g)
    return sol[:Pf * 6].reshape(Pf, 6), sol[Pf * 6:].reshape(M, HW), kx


def corr_lookup(volume, coords, r):
    """grid_sample statement of the (2r+1)^2 window lookup on an all-pairs volume (corr.py:23-39 + DROID kernel)."""
    import torch.nn.functional as F
    BN, h1, w1, h2, w2 = volume.shape
    rd = 2 * r + 1
    out = torch.zeros(BN, rd, rd, h1, w1, dtype=volume.dtype)
    v = volume.reshape(BN * h1 * w1, 1, h2, w2)
    x0 = coords[:, 0].reshape(-1)
    y0 = coords[:, 1].reshape(-1)
    for i in range(rd):
        for j in range(rd):
            xs, ys = x0 - r + i, y0 - r + j
            grid = torch.stack([2 * xs / (w2 - 1) - 1, 2 * ys / (h2 - 1) - 1], -1).reshape(-1, 1, 1, 2)
            s = F.grid_sample(v, grid.to(v.dtype), mode="bilinear", padding_mode="zeros", align_corners=True)
            out[:, i, j] = s.reshape(BN, h1, w1)
    return out


# ------------------------------------------------------------------------------------------------ round 2: MoBA / JDSA / alt-corr
def _edge_jacobians(G, disps, intr, ii, jj, e, ht, wd, eps=1e-6):
    """numerical Jacobians of the projected coordinates of edge e w.r.t. left perturbations of G_i, G_j and the disparity"""
    HW = ht * wd
    i, j = int(ii[e]), int(jj[e])

    def coords(Gi, Gj, d):
        return project(Gj @ torch.linalg.inv(Gi), d, intr[i], intr[j], ht, wd)

    c0, valid = coords(G[i], G[j], disps[i])
    Ji = torch.zeros(HW, 2, 6, dtype=DT)
    Jj = torch.zeros(HW, 2, 6, dtype=DT)
    for a in range(6):
        da = np.zeros(6); da[a] = eps
        Ep, Em = torch.from_numpy(LO.exp_matrix(1, da)), torch.from_numpy(LO.exp_matrix(1, -da))
        Ji[:, :, a] = (coords(Ep @ G[i], G[j], disps[i])[0] - coords(Em @ G[i], G[j], disps[i])[0]) / (2 * eps)
        Jj[:, :, a] = (coords(G[i], Ep @ G[j], disps[i])[0] - coords(G[i], Em @ G[j], disps[i])[0]) / (2 * eps)
    Jz = (coords(G[i], G[j], disps[i] + eps)[0] - coords(G[i], G[j], disps[i] - eps)[0]) / (2 * eps)
    return c0, valid, Ji, Jj, Jz


def moba_dense(target, weight, poses7, disps, intr, ii, jj, fixedp, ep=0.1, lm=1e-4):
    """Motion-only step (geom/ba.py:110-158 + chol.py:32-45): pose block only, damping H + (ep + lm H) I, dense fp64 solve."""
    P, ht, wd = disps.shape
    HW = ht * wd
    G = se3_matrix(poses7)
    Pf = P - fixedp
    H = torch.zeros(Pf * 6, Pf * 6, dtype=DT)
    g = torch.zeros(Pf * 6, dtype=DT)
    for e in range(len(ii)):
        c0, valid, Ji, Jj, _ = _edge_jacobians(G, disps, intr, ii, jj, e, ht, wd)
        r = target[e].reshape(HW, 2).to(DT) - c0
        w = 0.001 * valid[:, None] * weight[e].reshape(HW, 2).to(DT)
        blocks = [(int(ii[e]) - fixedp, Ji), (int(jj[e]) - fixedp, Jj)]
        for (p1, J1) in blocks:
            if p1 < 0:
                continue
            g[p1 * 6:p1 * 6 + 6] += torch.einsum("kc,kca,kc->a", w, J1, r)
            for (p2, J2) in blocks:
                if p2 >= 0:
                    H[p1 * 6:p1 * 6 + 6, p2 * 6:p2 * 6 + 6] += torch.einsum("kc,kca,kcb->ab", w, J1, J2)
    idx = torch.arange(Pf * 6)
    H[idx, idx] = H[idx, idx] + (ep + lm * H[idx, idx])
    return torch.linalg.solve(H, g).reshape(Pf, 6)


def proj_trans_dense(target, weight, poses7, disps, intr, ii, jj):
    """C [M,HW] = sum_e w Jz^2, w [M,HW] = sum_e w r Jz per source frame (the quantities droid_backends.proj_trans returns)"""
    P, ht, wd = disps.shape
    HW = ht * wd
    G = se3_matrix(poses7)
    kx = np.unique(ii)
    kk = {int(k): m for m, k in enumerate(kx)}
    C = torch.zeros(len(kx), HW, dtype=DT)
    wv = torch.zeros(len(kx), HW, dtype=DT)
    for e in range(len(ii)):
        c0, valid, _, _, Jz = _edge_jacobians(G, disps, intr, ii, jj, e, ht, wd)
        r = target[e].reshape(HW, 2).to(DT) - c0
        w = 0.001 * valid[:, None] * weight[e].reshape(HW, 2).to(DT)
        m = kk[int(ii[e])]
        C[m] += (w * Jz * Jz).sum(-1)
        wv[m] += (w * r * Jz).sum(-1)
    return C, wv, kx


def bi_inter_ref(scales, grid):
    """bilinear interpolation of scales [M,hs,ws] at grid [M,ht,wd,2] (x,y) and its dense Jacobian, by autograd (fp64)"""
    import torch.nn.functional as F
    M, hs, ws = scales.shape
    s = scales.clone().to(DT).requires_grad_(True)
    gx = 2 * grid[..., 0].to(DT) / max(ws - 1, 1) - 1
    gy = 2 * grid[..., 1].to(DT) / max(hs - 1, 1) - 1
    v = F.grid_sample(s[:, None], torch.stack([gx, gy], -1), mode="bilinear", padding_mode="zeros", align_corners=True)[:, 0]
    J = torch.zeros(M, v.shape[1], v.shape[2], hs * ws, dtype=DT)
    for m in range(M):
        for y in range(v.shape[1]):
            for x in range(v.shape[2]):
                (g,) = torch.autograd.grad(v[m, y, x], s, retain_graph=True)
                J[m, y, x] = g[m].reshape(-1)
    return v.detach(), J


def jdsa_dense(C, w, eta, disps_src, prior, scales, alpha, ep=0.1, lm=1e-4):
    """Un-reduced solve of the JDSA normal equations (geom/ba.py:172-241 + chol.py:80-107): unknowns = the source frames'
    disparities [M,HW] and their scale-grid nodes [M,D].  C, w from proj_trans; disps_src, prior [M,ht,wd]; scales [M,hs,ws].
    Returns (dz [M,HW], dso [M,D])."""
    M, ht, wd = disps_src.shape
    HW = ht * wd
    hs, ws = scales.shape[-2:]
    D = hs * ws
    yy, xx = torch.meshgrid(torch.linspace(0, hs - 1 - 1e-6, ht), torch.linspace(0, ws - 1 - 1e-6, wd), indexing="ij")
    grid = torch.stack((xx, yy), -1)[None].expand(M, -1, -1, -1)
    sbi, Jbi = bi_inter_ref(scales, grid)
    m = (prior > 0).to(DT).reshape(M, HW)
    rd = (disps_src.to(DT) - prior.to(DT) * sbi).reshape(M, HW)
    Jso = -m[..., None] * prior.to(DT).reshape(M, HW, 1) * Jbi.reshape(M, HW, D)           # d rd / d scale nodes
    n = M * HW + M * D
    A = torch.zeros(n, n, dtype=DT)
    b = torch.zeros(n, dtype=DT)
    for k in range(M):
        zi = torch.arange(k * HW, (k + 1) * HW)
        si = torch.arange(M * HW + k * D, M * HW + (k + 1) * D)
        Ck = C[k].to(DT) + m[k] * alpha + (1 - m[k]) * eta[k].reshape(-1).to(DT)
        wk = w[k].to(DT) - m[k] * alpha * rd[k]
        A[zi, zi] += Ck
        b[zi] += wk
        Hs = alpha * Jso[k].T @ Jso[k]
        Hs = Hs + (ep + lm * Hs) * torch.eye(D, dtype=DT)
        A[si[:, None], si[None]] += Hs
        Es = alpha * Jso[k].T                                                                  # [D,HW] (Jd = 1)
        A[si[:, None], zi[None]] += Es
        A[zi[:, None], si[None]] += Es.T
        b[si] += -alpha * Jso[k].T @ rd[k]
    sol = torch.linalg.solve(A, b)
    return sol[:M * HW].reshape(M, HW), sol[M * HW:].reshape(M, D)


def altcorr_ref(fmap1, fmap2, coords, r):
    """on-the-fly correlation == the lookup in the all-pairs volume (modules/corr.py:62-71 `CorrBlock.corr` without the /4, which
    AltCorrBlock applies to the feature maps): fmap1 [BN,H,W,C], fmap2 [BN,H2,W2,C], coords [BN,S,H,W,2] -> [BN,S,(2r+1)^2,H,W]"""
    BN, H, W, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    vol = torch.einsum("nyxc,nuvc->nyxuv", fmap1, fmap2)
    S = coords.shape[1]
    rd = 2 * r + 1
    outs = []
    for s in range(S):
        c = coords[:, s].permute(0, 3, 1, 2)                       # [BN,2,H,W] (x,y)
        outs.append(corr_lookup(vol, c, r).reshape(BN, rd * rd, H, W))
    return torch.stack(outs, 1)


def depth_filter_ref(poses7, disps, intr, inds, thresh):
    """droid_backends.depth_filter (call site /root/reference/hislam2/util/droid_visualization.py:100), restated in fp64 from the
    published DROID-SLAM kernel (src/droid_kernels.cu `depth_filter_kernel`; the extension is absent from the reference tree:
    PARITY UNPINNED).  Returns (count [M,ht,wd], margin [M,ht,wd]): margin = the smallest | |1/dj - 1/d| - thresh | met, so a test
    can leave out pixels whose decision sits on an fp32 rounding."""
    disps = np.asarray(disps, np.float64)
    n, ht, wd = disps.shape
    G = se3_matrix(poses7).numpy()
    fx, fy, cx, cy = [float(v) for v in np.asarray(intr).reshape(-1)[:4]]
    y, x = np.meshgrid(np.arange(ht, dtype=np.float64), np.arange(wd, dtype=np.float64), indexing="ij")
    M = len(inds)
    count = np.zeros((M, ht, wd))
    margin = np.full((M, ht, wd), np.inf)
    for m, ix in enumerate(int(v) for v in inds):
        Xi = np.stack([(x - cx) / fx, (y - cy) / fy, np.ones_like(x), disps[ix]], -1).reshape(-1, 4)
        for neigh in range(6):
            jx = ix - neigh - 1 if neigh < 3 else ix + neigh
            if jx < 0 or jx >= n:
                continue
            Xj = Xi @ (G[jx] @ np.linalg.inv(G[ix])).T
            with np.errstate(divide="ignore", invalid="ignore"):
                uj, vj, dj = fx * Xj[:, 0] / Xj[:, 2] + cx, fy * Xj[:, 1] / Xj[:, 2] + cy, Xj[:, 3] / Xj[:, 2]
                u0, v0 = np.floor(uj), np.floor(vj)
                inside = (u0 >= 0) & (v0 >= 0) & (u0 < wd - 1) & (v0 < ht - 1)
                # a landing point within 1e-4 px of a pixel edge may floor differently in fp32: flag it through the margin
                edge = np.minimum(np.abs(uj - np.round(uj)), np.abs(vj - np.round(vj)))
                u0c, v0c = np.clip(np.nan_to_num(u0), 0, wd - 2).astype(int), np.clip(np.nan_to_num(v0), 0, ht - 2).astype(int)
                hit = np.zeros(ht * wd, bool)
                mg = np.full(ht * wd, np.inf)
                for dv in (0, 1):
                    for du in (0, 1):
                        e = np.abs(1.0 / dj - 1.0 / disps[jx][v0c + dv, u0c + du])
                        hit |= e < thresh[m]
                        mg = np.minimum(mg, np.abs(e - thresh[m]))
            hit &= inside
            mg = np.where(inside, mg, np.inf)
            mg = np.where(edge < 1e-4, 0.0, mg)
            count[m] += hit.reshape(ht, wd)
            margin[m] = np.minimum(margin[m], mg.reshape(ht, wd))
    return count, margin
