"""Dense bundle adjustment with the reference's `geom.ba.BA` signature (/root/reference/hislam2/geom/ba.py:32-107),
`geom.projective_ops.projective_transform` semantics (:44-74) and `geom.chol.schur_solve` damping (:47-78), executed by
the fused gfx950 operators in csrc/ba.hip (one Gauss-Newton step = 5 launches + the two Lie kernels that form G_ij).

The reference version cannot run (it calls an undefined `scatter_sum` and `droid_backends` is absent), so parity is
pinned by oracle/ba_oracle.py (finite-difference Jacobians + dense solve of the un-reduced system), not by the reference.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .lietorch import SE3


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def edge_structure(ii, jj, P, fixedp):
    """host bookkeeping: kx = unique(ii), CSR of edges by source, presence mask of the E blocks (ba.py:76-90)."""
    ii_h, jj_h = np.asarray(ii.cpu(), np.int64), np.asarray(jj.cpu(), np.int64)
    kx, kk = np.unique(ii_h, return_inverse=True)
    M = len(kx)
    order = np.argsort(kk, kind="stable")
    src_ptr = np.zeros(M + 1, np.int32)
    np.add.at(src_ptr, kk + 1, 1)
    src_ptr = np.cumsum(src_ptr).astype(np.int32)
    Pf = P - fixedp
    present = np.zeros((Pf, M), np.uint8)
    for e in range(len(ii_h)):
        i, j, m = ii_h[e] - fixedp, jj_h[e] - fixedp, kk[e]
        if i >= 0:
            present[i, m] = 1
        if j >= 0:
            present[j, m] = 1
    return kx, kk, order.astype(np.int32), src_ptr, present


_EDGE_TABLES = {}


def _prepare(target, weight, eta, poses, disps, intrinsics, ii, jj, fixedp):
    dev = disps.device
    B, P, ht, wd = disps.shape
    if B != 1:
        raise NotImplementedError("batch 1 (the only use in the reference)")
    N = ii.shape[0]
    HW = ht * wd
    # the graph's index tables (host analysis + six small uploads) are kept per topology: Gauss-Newton runs several steps on one graph
    ii_h, jj_h = np.ascontiguousarray(np.asarray(ii.cpu())), np.ascontiguousarray(np.asarray(jj.cpu()))
    key = (ii_h.tobytes(), jj_h.tobytes(), P, fixedp, str(dev))
    tab = _EDGE_TABLES.get(key)
    if tab is None:
        if len(_EDGE_TABLES) >= 16:
            _EDGE_TABLES.clear()
        kx, kk, order, src_ptr, present = edge_structure(ii, jj, P, fixedp)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        tab = _EDGE_TABLES[key] = dict(kx=kx, ii=ii.to(dev, torch.int32).contiguous(), jj=jj.to(dev, torch.int32).contiguous(),
                                       ii_l=ii.to(dev), jj_l=jj.to(dev), src_ptr=t(src_ptr, torch.int32), order=t(order, torch.int32),
                                       kx_d=t(kx, torch.int32), present=t(present, torch.uint8))
    kx = tab["kx"]
    M = len(kx)
    ii_d, jj_d = tab["ii"], tab["jj"]
    G = poses[0] if isinstance(poses, SE3) else SE3(poses[0])
    Gij = (G[tab["jj_l"]] * G[tab["ii_l"]].inv()).data.contiguous().float()               # projective_ops.py:51
    intr = intrinsics.reshape(-1, 4).float()
    if intr.shape[0] == 1:
        intr = intr.expand(P, 4)
    intr = intr.contiguous()
    eta_d = None
    if eta is not None:
        eta_d = eta.to(dev).float().expand(M, ht, wd).reshape(M, HW).contiguous() if eta.numel() != M * HW else eta.reshape(M, HW).float().contiguous()
    return dict(dev=dev, P=P, ht=ht, wd=wd, N=N, HW=HW, M=M, kx=kx, G=G, Gij=Gij, intr=intr, eta=eta_d, ii=ii_d, jj=jj_d,
                tgt=target.reshape(N, HW, 2).float().contiguous(), wgt=weight.reshape(N, HW, 2).float().contiguous(),
                dsp=disps.reshape(P, HW).float().contiguous(),
                src_ptr=tab["src_ptr"], order=tab["order"], kx_d=tab["kx_d"], present=tab["present"])


def _assemble(c, fixedp, motion_only=False):
    """undamped reduced system of the edges in `c`: S [n,n], vS [n], diag(H) [n] (cut3r_ba_assemble) + the workspace holding E, C, w"""
    lib = _lib.load()
    dev, n = c["dev"], (c["P"] - fixedp) * 6
    ws = torch.empty(int(lib.cut3r_ba_workspace_floats(c["P"], c["ht"], c["wd"], c["N"], c["M"], fixedp)), device=dev)
    S, vS, hd = torch.empty(n, n, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
    eta = c["eta"] if c["eta"] is not None else torch.zeros(c["M"], c["HW"], device=dev)
    check(lib.cut3r_ba_assemble(_p(c["Gij"]), _p(c["dsp"]), _p(c["intr"]), _p(c["tgt"]), _p(c["wgt"]), _p(eta), _p(c["ii"]), _p(c["jj"]),
                                _p(c["src_ptr"]), _p(c["order"]), _p(c["kx_d"]), _p(c["present"]), c["P"], c["ht"], c["wd"], c["N"], c["M"],
                                fixedp, int(motion_only), _p(ws), _p(S), _p(vS), _p(hd), _stream()), "cut3r_ba_assemble")
    return ws, S, vS, hd


def _solve(S, vS, hd, ep, lm):
    lib = _lib.load()
    n = S.shape[0]
    dx = torch.empty(n // 6, 6, device=S.device)
    flag = torch.zeros(1, dtype=torch.int32, device=S.device)
    scratch = torch.empty(n * n, device=S.device)
    check(lib.cut3r_ba_solve(_p(S), _p(vS), _p(hd), n, float(ep), float(lm), _p(scratch), _p(dx), _p(flag), _stream()), "cut3r_ba_solve")
    return dx, flag


def BA(target, weight, eta, poses, disps, intrinsics, ii, jj, fixedp=1, ep=0.1, lm=1e-4, group=None, edge_mask=None):
    """One full-BA Gauss-Newton step.  target/weight [1,N,ht,wd,2]; eta [M,ht,wd] (or broadcastable); poses SE3 [1,P];
    disps [1,P,ht,wd]; intrinsics [1,P,4] (or [1,4]); ii,jj LongTensor [N].  Returns (poses, disps, info).

    Multi-GPU (north_star: per-edge BA sharded with an all-reduce of the normal-equation blocks): pass the torch.distributed
    `group` and this rank's `edge_mask` (bool [N]) -- edges MUST be split by source frame `ii` so that a source frame's depth
    blocks stay on one rank (`shard_edges_by_source`).  Each rank assembles the undamped reduced system of its edges, S / vS /
    diag(H) are summed with ONE all-reduce of (6(P-fixedp))^2 + 12(P-fixedp) floats, every rank solves the same damped system and
    updates the disparities of its own source frames; the disparity increments are then summed (disjoint supports)."""
    dev = disps.device
    B, P, ht, wd = disps.shape
    if edge_mask is not None:
        keep = torch.as_tensor(edge_mask, dtype=torch.bool)
        if int(keep.sum()) == 0:
            raise ValueError("this rank holds no edge: give every rank at least one source frame")
        sel = keep.nonzero().reshape(-1)
        target, weight = target[:, sel.to(target.device)], weight[:, sel.to(weight.device)]
        ii_l, jj_l = ii[sel.to(ii.device)], jj[sel.to(jj.device)]
        kx_all = np.unique(np.asarray(ii.cpu()))
        if eta.numel() == len(kx_all) * ht * wd:                      # eta is indexed by source frame: keep this rank's rows
            rows = np.searchsorted(kx_all, np.unique(np.asarray(ii_l.cpu())))
            eta = eta.reshape(len(kx_all), ht, wd)[torch.as_tensor(rows, device=eta.device)]
    else:
        ii_l, jj_l = ii, jj
    c = _prepare(target, weight, eta, poses, disps, intrinsics, ii_l, jj_l, fixedp)
    ws, S, vS, hd = _assemble(c, fixedp)
    if group is not None:
        import torch.distributed as dist
        packed = torch.cat([S.reshape(-1), vS, hd])
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(packed, group=group)
        else:
            host = packed.cpu()
            dist.all_reduce(host, group=group)
            packed = host.to(dev)
        n = S.shape[0]
        S, vS, hd = packed[:n * n].reshape(n, n).contiguous(), packed[n * n:n * n + n].contiguous(), packed[n * n + n:].contiguous()
    dx, flag = _solve(S, vS, hd, ep, lm)
    M, HW, kx = c["M"], c["HW"], c["kx"]
    dz = torch.empty(M, HW, device=dev)
    check(_lib.load().cut3r_ba_backsub(_p(ws), _p(dx), _p(c["present"]), P, ht, wd, c["N"], M, fixedp, _p(dz), _stream()), "cut3r_ba_backsub")
    # retraction (ba.py:100-105)
    full_dx = torch.zeros(P, 6, device=dev)
    full_dx[fixedp:] = dx
    new_poses = SE3(c["G"].data[None]).retr(full_dx[None])
    inc = torch.zeros(P, ht, wd, device=dev)
    inc[torch.as_tensor(kx, device=dev)] = dz.view(M, ht, wd)
    if group is not None:
        import torch.distributed as dist
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(inc, group=group)
        else:
            host = inc.cpu()
            dist.all_reduce(host, group=group)
            inc = host.to(dev)
    new_disps = disps.clone()
    new_disps[0] += inc
    new_disps = torch.where(new_disps > 10, torch.zeros_like(new_disps), new_disps).clamp(min=0.001)
    return new_poses, new_disps, {"dx": dx, "dz": dz, "failed": flag, "kx": kx, "S": S, "vS": vS}


def shard_edges_by_source(ii, world, rank):
    """bool mask [N]: edges whose source frame belongs to `rank` (source frames dealt round-robin in sorted order) -- keeps
    C_k, w_k, E_.k of a frame on one rank"""
    ii_h = np.asarray(torch.as_tensor(ii).cpu(), np.int64)
    kx = np.unique(ii_h)
    owner = {int(k): n % world for n, k in enumerate(kx)}
    return torch.as_tensor([owner[int(i)] == rank for i in ii_h])


def MoBA(target, weight, eta, poses, disps, intrinsics, ii, jj, fixedp=1, rig=1, ep=0.1, lm=1e-4):
    """Motion-only bundle adjustment (geom/ba.py:110-158): the pose block of the normal equations with block_solve's damping
    (geom/chol.py:32-45); disparities stay fixed.  rig = 1 (the reference's only use)."""
    if rig != 1:
        raise NotImplementedError("rig > 1 (multi-camera rigs) is not used by the reference")
    B, P, ht, wd = disps.shape
    c = _prepare(target, weight, eta, poses, disps, intrinsics, ii, jj, fixedp)
    _, S, vS, hd = _assemble(c, fixedp, motion_only=True)
    dx, flag = _solve(S, vS, hd, ep, lm)
    full_dx = torch.zeros(P, 6, device=disps.device)
    full_dx[fixedp:] = dx
    return SE3(c["G"].data[None]).retr(full_dx[None])


def _mono_prior_solve(C, w, H, E, v, ep, lm, dzcov):
    """cut3r_schur_mono_prior on dense H [n,n], E [n,cols], v [n], C / w [cols] (fp32, contiguous, on the GPU)"""
    lib = _lib.load()
    n, cols = int(H.shape[0]), int(E.shape[1])
    dev = H.device
    ws = torch.empty(int(lib.cut3r_schur_mono_prior_workspace_floats(n, cols)), device=dev)
    dso, dz = torch.empty(n, device=dev), torch.empty(cols, device=dev)
    cov = torch.empty(cols, device=dev) if dzcov else None
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.cut3r_schur_mono_prior(_p(C), _p(w), _p(H), _p(E), _p(v), n, cols, float(ep), float(lm), _p(ws), _p(dso), _p(dz), _p(cov), _p(flag),
                                     _stream()), "cut3r_schur_mono_prior")
    return dso, dz, cov, flag


CHOL_MAXN = 192          # the in-LDS Cholesky of csrc/ba.hip (cut3r_schur_mono_prior / cut3r_ba_solve) holds systems up to this size


def _schur_mono_prior_large(C, w, Hs, Es, vs, ep, lm, dzcov):
    """The same solve for reduced systems LARGER than the in-LDS Cholesky (M*D > 192: ADVICE r3): dense tensor ops on the GPU, the
    formulation of geom/chol.py:80-107 term by term (damping H + (ep + lm H) I, S = H - E C^-1 E^T, CholeskySolver's zeros on failure)."""
    D = Hs.shape[-1]
    B, M, HW = C.shape
    Q = (1.0 / C).view(B, M * HW, 1)
    w = w.reshape(B, M * HW, 1)
    H = Hs.permute(0, 1, 3, 2, 4).reshape(B, M * D, M * D)
    E = Es.permute(0, 1, 3, 2, 4).reshape(B, M * D, M * HW)
    v = vs.reshape(B, M * D, 1)
    H = H + (ep + lm * H) * torch.eye(M * D, device=H.device)
    Et = E.transpose(1, 2)
    S = H - torch.matmul(E, Q * Et)
    v = v - torch.matmul(E, Q * w)
    L, info = torch.linalg.cholesky_ex(S)
    ok = int(info.max()) == 0
    dso = torch.cholesky_solve(v, L) if ok else torch.zeros_like(v)
    dz = (Q * (w - Et @ dso)).reshape(B, M, HW)
    cov = None
    if dzcov:
        Fm = torch.linalg.solve_triangular(L, E * Q[..., 0], upper=False) if ok else torch.zeros_like(E)
        cov = (torch.sum(torch.square(Fm), dim=1) + Q[..., 0]).reshape(M, HW)
    return dso.reshape(B, M, D), dz, cov


def schur_solve_mono_prior(C, w, Hs, Es, vs, ep=0.1, lm=1e-4, dzcov=False):
    """geom/chol.py:80-107 on the HIP kernels of csrc/ba.hip (reduction S = H + damping - E C^-1 E^T, in-LDS Cholesky, back-substitution,
    column-wise covariance): the reduced system is M*D x M*D with D = hs*ws scale-grid nodes (M*D <= 192; larger systems: _schur_mono_prior_large).
    C, w [1,M,HW]; Hs [1,M,M,D,D]; Es [1,M,M,D,HW]; vs [1,M,D].  Returns (dso [1,M,D], dz [1,M,HW], dzcov [M,HW] or None)."""
    D = Hs.shape[-1]
    B, M, HW = C.shape
    if B != 1:
        raise NotImplementedError("batch 1 (the only use in the reference)")
    if M * D > CHOL_MAXN:
        return _schur_mono_prior_large(C, w, Hs, Es, vs, ep, lm, dzcov)
    H = Hs.permute(0, 1, 3, 2, 4).reshape(M * D, M * D).float().contiguous()         # the reference's layout change, chol.py:88-89
    E = Es.permute(0, 1, 3, 2, 4).reshape(M * D, M * HW).float().contiguous()
    dso, dz, cov, _ = _mono_prior_solve(C.reshape(-1).float().contiguous(), w.reshape(-1).float().contiguous(), H, E,
                                        vs.reshape(-1).float().contiguous(), ep, lm, dzcov)
    return dso.reshape(1, M, D), dz.reshape(1, M, HW), (cov.reshape(M, HW) if cov is not None else None)


def get_prior_depth_aligned(depth_prior, scales):
    """geom/ba.py:160-170"""
    from . import droid_backends
    M, ht, wd = depth_prior.shape
    hs, ws = scales.shape[-2:]
    meshx, meshy = torch.meshgrid(torch.linspace(0, hs - 1 - 1e-6, ht), torch.linspace(0, ws - 1 - 1e-6, wd), indexing="ij")
    grid = torch.stack((meshy, meshx), -1).to(depth_prior.device)
    grid = grid.unsqueeze(0).expand(M, -1, -1, -1).contiguous()
    mscales_bi, Jbi = droid_backends.bi_inter(scales, grid)
    return depth_prior * mscales_bi, Jbi


def JDSA(target, weight, eta, poses, disps, intrinsics, disps_prior, dscales, ii, jj, alpha, ep=0.1, lm=1e-4):
    """Joint depth and scale adjustment (geom/ba.py:172-241): disparities of the source frames + one bilinear scale grid per
    frame that aligns its monocular prior; poses fixed.  disps [1,P,ht,wd], disps_prior [P,ht,wd], dscales [P,hs,ws]
    (updated in place like the reference).  Returns (disps, dscales, dzcov)."""
    from . import droid_backends
    B, P, ht, wd = disps.shape
    dev = disps.device
    HW = ht * wd
    G = poses[0] if isinstance(poses, SE3) else SE3(poses[0])
    Cm, wv = droid_backends.proj_trans(G.data, disps[0], intrinsics.reshape(-1, 4)[0], target, weight, ii, jj)
    kx = torch.unique(ii.to(dev))
    M = kx.shape[0]
    prior = disps_prior[kx]
    m = (prior > 0).to(torch.float).view(-1, HW)
    hs, ws = dscales.shape[-2:]
    disps_bi, Jbi = get_prior_depth_aligned(prior, dscales[kx])
    rd = (disps[0, kx] - disps_bi).reshape(-1, HW).float().contiguous()
    D = hs * ws
    n = M * D
    C = Cm.reshape(M, HW) + m * float(alpha) + (1 - m) * eta.reshape(M, HW)                     # Jd = 1
    w = wv.reshape(M, HW) - m * float(alpha) * rd
    if n > CHOL_MAXN:
        # more than 192 / D source frames: H and E are block diagonal (frame k's blocks sit at (kx, kx): ba.py:213-228), and so
        # is the damping, so the reduced system falls apart into M independent D x D systems -- one per frame, E restricted to that frame's
        # HW columns: no dense [n, n] / [n, M*HW] buffers, no size limit.  Same kernels, same arithmetic per frame.
        lib = _lib.load()
        dso, dz, dzcov = torch.empty(M, D, device=dev), torch.empty(M, HW, device=dev), torch.empty(M, HW, device=dev)
        pr, jb = prior.reshape(M, HW).float().contiguous(), Jbi.reshape(M, HW, D).float().contiguous()
        Hk, Ek, vk = torch.empty(D, D, device=dev), torch.empty(D, HW, device=dev), torch.empty(D, device=dev)
        for k in range(M):
            check(lib.cut3r_jdsa_blocks(_p(pr[k]), _p(jb[k]), _p(rd[k]), float(alpha), 1, HW, D, _p(Hk), _p(Ek), _p(vk), _stream()), "cut3r_jdsa_blocks")
            a_, b_, c_, _ = _mono_prior_solve(C[k].contiguous(), w[k].contiguous(), Hk, Ek, vk, ep, lm, True)
            dso[k], dz[k], dzcov[k] = a_, b_, c_
        new_disps = disps.clone()
        new_disps[0, kx] += dz.view(M, ht, wd)
        dscales[kx] += dso.view(-1, hs, ws)
        new_disps = torch.where(new_disps > 10, torch.zeros_like(new_disps), new_disps).clamp(min=0.001)
        return new_disps, dscales, dzcov
    # the scatter of the reference puts frame k's blocks on the diagonal (kx, kx): block-diagonal H [n,n] / E [n,M*HW], written by one
    # kernel (Jso = -m prior Jbi, H_k = alpha Jso^T Jso, E_k = alpha Jso^T (Jd = 1), v_k = -alpha Jso^T rd, ba.py:213-228)
    Hd, Ed, vd = torch.empty(n, n, device=dev), torch.empty(n, M * HW, device=dev), torch.empty(n, device=dev)
    check(_lib.load().cut3r_jdsa_blocks(_p(prior.reshape(M, HW).float().contiguous()), _p(Jbi.reshape(M, HW, D).float().contiguous()), _p(rd),
                                        float(alpha), M, HW, D, _p(Hd), _p(Ed), _p(vd), _stream()), "cut3r_jdsa_blocks")
    dso, dz, dzcov, _ = _mono_prior_solve(C.reshape(-1).contiguous(), w.reshape(-1).contiguous(), Hd, Ed, vd, ep, lm, True)
    dzcov = dzcov.reshape(M, HW)
    new_disps = disps.clone()
    new_disps[0, kx] += dz.view(M, ht, wd)
    dscales[kx] += dso.view(-1, hs, ws)
    new_disps = torch.where(new_disps > 10, torch.zeros_like(new_disps), new_disps).clamp(min=0.001)
    return new_disps, dscales, dzcov
