"""`droid_backends` module surface (the reference's setup.py target whose CUDA sources are absent: SURVEY F5) on the
gfx950 kernels.  Call sites: hislam2/modules/corr.py:12,19 (corr_index_forward/backward).

Implemented: corr_index_forward, corr_index_backward, iproj (via the Lie kernels).  The remaining names the reference
mentions (altcorr_forward/backward, bi_inter, proj_trans, depth_filter) raise NotImplementedError with a pointer to
DESIGN.md -- they belong to code paths (AltCorrBlock, JDSA, the Open3D viewer) that nothing in the reference reaches.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check


def _p(t):
    return C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def corr_index_forward(volume, coords, radius):
    """volume [BN,h1,w1,h2,w2], coords [BN,2,h1,w1] -> (corr [BN,2r+1,2r+1,h1,w1],)"""
    if not volume.is_cuda:
        raise RuntimeError("droid_backends: GPU tensors only")
    BN, h1, w1, h2, w2 = volume.shape
    volume, coords = volume.contiguous().float(), coords.contiguous().float()
    rd = 2 * radius + 1
    out = torch.empty(BN, rd, rd, h1, w1, device=volume.device)
    check(_lib.load().cut3r_corr_index_forward(_p(volume), _p(coords), _p(out), BN, h1, w1, h2, w2, int(radius), _s()), "corr_index_forward")
    return (out,)


def corr_index_backward(volume, coords, grad, radius):
    BN, h1, w1, h2, w2 = volume.shape
    coords, grad = coords.contiguous().float(), grad.contiguous().float()
    gv = torch.empty(BN, h1, w1, h2, w2, device=volume.device)
    check(_lib.load().cut3r_corr_index_backward(_p(coords), _p(grad), _p(gv), BN, h1, w1, h2, w2, int(radius), _s()), "corr_index_backward")
    return (gv,)


def iproj(poses_inv, disps, intrinsics):
    """camera-to-world back-projection: poses_inv [n,7] (c2w SE3 data), disps [n,h,w], intrinsics [4] -> points [n,h,w,3]"""
    from .lietorch import SE3
    n, h, w = disps.shape
    fx, fy, cx, cy = [float(v) for v in intrinsics.reshape(-1)[:4]]
    y, x = torch.meshgrid(torch.arange(h, device=disps.device).float(), torch.arange(w, device=disps.device).float(), indexing="ij")
    X = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(x)], -1)[None].expand(n, -1, -1, -1)
    P4 = torch.cat([X, disps[..., None]], -1)
    out = SE3(poses_inv)[:, None, None].act(P4)
    return out[..., :3] / out[..., 3:].clamp(min=1e-6)


def _missing(name):
    def f(*a, **k):
        raise NotImplementedError(f"droid_backends.{name}: not reachable from any live or BA path of the reference; see DESIGN.md section 7")
    return f


altcorr_forward = _missing("altcorr_forward")
altcorr_backward = _missing("altcorr_backward")
bi_inter = _missing("bi_inter")
proj_trans = _missing("proj_trans")
depth_filter = _missing("depth_filter")
