"""`droid_backends` module surface (the reference's setup.py target whose CUDA sources are absent: SURVEY F5) on the
gfx950 kernels.  Call sites: hislam2/modules/corr.py:12,19 (corr_index_forward/backward).

Implemented: corr_index_forward / backward, altcorr_forward / backward (+ AltCorrBlock), bi_inter, proj_trans, iproj,
depth_filter (the point-cloud filter of the Open3D viewer, hislam2/util/droid_visualization.py:100; the viewer itself is out
of scope).
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


def bi_inter(scales, grid):
    """scales [M,hs,ws], grid [M,ht,wd,2] (x,y) -> (vals [M,ht,wd], J [M,ht,wd,hs*ws])   (call site geom/ba.py:167)"""
    M, hs, ws = scales.shape
    _, ht, wd, _ = grid.shape
    scales, grid = scales.contiguous().float(), grid.contiguous().float()
    vals = torch.empty(M, ht, wd, device=scales.device)
    J = torch.empty(M, ht, wd, hs * ws, device=scales.device)
    check(_lib.load().cut3r_bi_inter(_p(scales), _p(grid), M, hs, ws, ht, wd, _p(vals), _p(J), _s()), "bi_inter")
    return vals, J


def proj_trans(poses, disps, intrinsics, target, weight, ii, jj):
    """poses [P,7] (world->camera SE3 data), disps [P,ht,wd], intrinsics [4], target/weight [1,N,ht,wd,2] | [N,ht*wd,2], ii, jj
    -> (C [M,ht*wd], w [M,ht*wd]) of the source frames unique(ii)   (call site geom/ba.py:200)"""
    import numpy as np
    from .ba import edge_structure
    from .lietorch import SE3
    dev = disps.device
    P, ht, wd = disps.shape
    N, HW = ii.shape[0], ht * wd
    kx, kk, order, src_ptr, _ = edge_structure(ii, jj, P, P)
    M = len(kx)
    G = SE3(poses.reshape(P, 7).to(dev).float())
    Gij = (G[jj.to(dev)] * G[ii.to(dev)].inv()).data.contiguous().float()
    intr = intrinsics.reshape(-1, 4)[:1].float().expand(P, 4).contiguous().to(dev)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    src_ptr_d, order_d, kx_d = t(src_ptr, torch.int32), t(order, torch.int32), t(kx, torch.int32)
    ii_d, jj_d = ii.to(dev, torch.int32).contiguous(), jj.to(dev, torch.int32).contiguous()
    tgt = target.reshape(N, HW, 2).float().contiguous()
    wgt = weight.reshape(N, HW, 2).float().contiguous()
    dsp = disps.reshape(P, HW).float().contiguous()
    ws = torch.empty(N * ((HW + 255) // 256) * 120, device=dev)
    Cm, wv = torch.empty(M, HW, device=dev), torch.empty(M, HW, device=dev)
    check(_lib.load().cut3r_ba_proj_trans(_p(Gij), _p(dsp), _p(intr), _p(tgt), _p(wgt), _p(ii_d), _p(jj_d), _p(src_ptr_d), _p(order_d), _p(kx_d),
                                          P, ht, wd, N, M, _p(ws), _p(Cm), _p(wv), _s()), "proj_trans")
    return Cm, wv


def altcorr_forward(fmap1, fmap2, coords, radius):
    """fmap1 [BN,H,W,C], fmap2 [BN,H2,W2,C], coords [BN,S,H,W,2] -> (corr [BN,S,(2r+1)^2,H,W],)   (call site corr.py:79)"""
    BN, H, W, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    fmap1, fmap2, coords = fmap1.contiguous().float(), fmap2.contiguous().float(), coords.contiguous().float()
    rd = 2 * radius + 1
    corr = torch.empty(BN, S, rd * rd, H, W, device=fmap1.device)
    check(_lib.load().cut3r_altcorr_forward(_p(fmap1), _p(fmap2), _p(coords), BN, S, H, W, H2, W2, Cc, int(radius), _p(corr), _s()), "altcorr_forward")
    return (corr,)


def altcorr_backward(fmap1, fmap2, coords, grad, radius):
    """-> (grad_fmap1, grad_fmap2, grad_coords = zeros)   (call site corr.py:87)"""
    BN, H, W, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    fmap1, fmap2, coords, grad = fmap1.contiguous().float(), fmap2.contiguous().float(), coords.contiguous().float(), grad.contiguous().float()
    g1, g2 = torch.empty_like(fmap1), torch.empty_like(fmap2)
    check(_lib.load().cut3r_altcorr_backward(_p(fmap1), _p(fmap2), _p(coords), _p(grad), BN, S, H, W, H2, W2, Cc, int(radius), _p(g1), _p(g2), _s()),
          "altcorr_backward")
    return g1, g2, torch.zeros_like(coords)


class CorrLayer(torch.autograd.Function):
    """modules/corr.py:74-90"""

    @staticmethod
    def forward(ctx, fmap1, fmap2, coords, r):
        ctx.r = r
        ctx.save_for_backward(fmap1, fmap2, coords)
        corr, = altcorr_forward(fmap1, fmap2, coords, r)
        return corr

    @staticmethod
    def backward(ctx, grad_corr):
        fmap1, fmap2, coords = ctx.saved_tensors
        g1, g2, gc = altcorr_backward(fmap1, fmap2, coords, grad_corr.contiguous(), ctx.r)
        return g1, g2, gc, None


class AltCorrBlock:
    """modules/corr.py:93-139: correlation pyramid evaluated on the fly (no all-pairs volume)"""

    def __init__(self, fmaps, num_levels=4, radius=3):
        import torch.nn.functional as F
        self.num_levels, self.radius = num_levels, radius
        B, N, Cc, H, W = fmaps.shape
        fmaps = fmaps.view(B * N, Cc, H, W) / 4.0
        self.pyramid = []
        for i in range(num_levels):
            sz = (B, N, H // 2 ** i, W // 2 ** i, Cc)
            self.pyramid.append(fmaps.permute(0, 2, 3, 1).contiguous().view(*sz))
            fmaps = F.avg_pool2d(fmaps, 2, stride=2)

    def corr_fn(self, coords, ii, jj):
        B, N, H, W, S, _ = coords.shape
        coords = coords.permute(0, 1, 4, 2, 3, 5)
        out = []
        for i in range(self.num_levels):
            f1 = self.pyramid[0][:, ii]
            f2 = self.pyramid[i][:, jj]
            ci = (coords / 2 ** i).reshape(B * N, S, H, W, 2).contiguous()
            f1 = f1.reshape((B * N,) + f1.shape[2:])
            f2 = f2.reshape((B * N,) + f2.shape[2:])
            corr = CorrLayer.apply(f1.float(), f2.float(), ci, self.radius)
            out.append(corr.view(B, N, S, -1, H, W).permute(0, 1, 3, 4, 5, 2))
        return torch.cat(out, dim=2)

    def __call__(self, coords, ii, jj):
        squeeze = False
        if len(coords.shape) == 5:
            coords = coords.unsqueeze(dim=-2)
            squeeze = True
        corr = self.corr_fn(coords, ii, jj)
        if squeeze:
            corr = corr.squeeze(dim=-1)
        return corr.contiguous()


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """poses [n,7] (world->camera SE3 data), disps [n,ht,wd], intrinsics [4], ix [M] int64 frame indices, thresh [M] ->
    count [M,ht,wd] float: the number of neighbour frames {ix-1, ix-2, ix-3, ix+3, ix+4, ix+5} that confirm each pixel's depth
    (call site hislam2/util/droid_visualization.py:98-104: `count >= 2` keeps a point)."""
    n, ht, wd = disps.shape
    poses, disps = poses[:n].contiguous().float(), disps.contiguous().float()
    ix = ix.to(disps.device, torch.int64).contiguous()
    M = ix.numel()
    count = torch.zeros(M, ht, wd, device=disps.device)
    if M == 0:
        return count
    if int(ix.min()) < 0 or int(ix.max()) >= n:
        raise IndexError("depth_filter: frame index out of range")
    intr = intrinsics.reshape(-1)[:4].to(disps.device, torch.float32).contiguous()
    thresh = thresh.to(disps.device, torch.float32).contiguous()
    check(_lib.load().cut3r_depth_filter(_p(poses), _p(disps), _p(intr), _p(ix), _p(thresh), n, M, ht, wd, _p(count), _s()), "depth_filter")
    return count
