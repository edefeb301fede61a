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


def BA(target, weight, eta, poses, disps, intrinsics, ii, jj, fixedp=1, ep=0.1, lm=1e-4):
    """One full-BA Gauss-Newton step.  target/weight [1,N,ht,wd,2]; eta [M,ht,wd] (or broadcastable); poses SE3 [1,P];
    disps [1,P,ht,wd]; intrinsics [1,P,4] (or [1,4]); ii,jj LongTensor [N].  Returns (poses, disps, info)."""
    dev = disps.device
    B, P, ht, wd = disps.shape
    if B != 1:
        raise NotImplementedError("batch 1 (the only use in the reference)")
    N = ii.shape[0]
    HW = ht * wd
    kx, kk, order, src_ptr, present = edge_structure(ii, jj, P, fixedp)
    M = len(kx)
    Pf = P - fixedp
    ii_d = ii.to(dev, torch.int32).contiguous()
    jj_d = jj.to(dev, torch.int32).contiguous()
    G = poses[0] if isinstance(poses, SE3) else SE3(poses[0])
    Gij = (G[jj.to(dev)] * G[ii.to(dev)].inv()).data.contiguous().float()               # projective_ops.py:51
    intr = intrinsics.reshape(-1, 4).float()
    if intr.shape[0] == 1:
        intr = intr.expand(P, 4)
    intr = intr.contiguous()
    eta_d = eta.to(dev).float().expand(M, ht, wd).reshape(M, HW).contiguous() if eta.numel() != M * HW else eta.reshape(M, HW).float().contiguous()
    tgt = target.reshape(N, HW, 2).float().contiguous()
    wgt = weight.reshape(N, HW, 2).float().contiguous()
    dsp = disps.reshape(P, HW).float().contiguous()
    lib = _lib.load()
    ws = torch.empty(int(lib.cut3r_ba_workspace_floats(P, ht, wd, N, M, fixedp)), device=dev)
    dx = torch.empty(Pf, 6, device=dev)
    dz = torch.empty(M, HW, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    # keep the index tensors alive until the launches are enqueued (a temporary would be recycled by the allocator)
    src_ptr_d, order_d, kx_d, present_d = t(src_ptr, torch.int32), t(order, torch.int32), t(kx, torch.int32), t(present, torch.uint8)
    check(lib.cut3r_ba_step(_p(Gij), _p(dsp), _p(intr), _p(tgt), _p(wgt), _p(eta_d), _p(ii_d), _p(jj_d),
                            _p(src_ptr_d), _p(order_d), _p(kx_d), _p(present_d),
                            P, ht, wd, N, M, fixedp, float(ep), float(lm), _p(ws), _p(dx), _p(dz), _p(flag), _stream()), "cut3r_ba_step")
    # retraction (ba.py:100-105)
    full_dx = torch.zeros(P, 6, device=dev)
    full_dx[fixedp:] = dx
    new_poses = SE3(G.data[None]).retr(full_dx[None])
    new_disps = disps.clone()
    new_disps[0, torch.as_tensor(kx, device=dev)] += dz.view(M, ht, wd)
    new_disps = torch.where(new_disps > 10, torch.zeros_like(new_disps), new_disps).clamp(min=0.001)
    return new_poses, new_disps, {"dx": dx, "dz": dz, "failed": flag, "kx": kx}
