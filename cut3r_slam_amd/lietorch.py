"""`lietorch`-compatible SE3 / SO3 / Sim3 on the gfx950 Lie kernels (csrc/lie.hip).

Class API as used by the reference (its dependency princeton-vl/lietorch 0.2 is an empty, un-vendored submodule:
/root/reference/thirdparty/lietorch): construction from a data tensor, `.exp(tangent)` (differentiable), `.log()`,
`.inv()`, `*` (group*group, group*points [..,3|4]), `.matrix()`, `.retr(a)`, `.adj(a)`, `.adjT(a)`, `[...]` indexing,
`.shape`, `.data`, `.manifold_dim`, `.Identity(...)`.
Call sites: hislam2/track_backend.py:269-270,298-299,418-425,458-459; hislam2/gs_backend_per_frame.py:721-731;
hislam2/geom/projective_ops.py:17,51,67,69; hislam2/geom/ba.py:29,37; hislam2/pgo_buffer.py:28-48.

Autograd: every op is a torch.autograd.Function whose backward is the matching HIP vector-Jacobian kernel; gradients
w.r.t. group elements are carried on the Euclidean components of `.data` (so `SE3.exp(xi).matrix()` followed by any
torch expression back-propagates to `xi` exactly as the reference's optimiser needs).  GPU tensors only.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check

_OP_EXP, _OP_LOG, _OP_INV, _OP_MATRIX = 0, 1, 2, 3
_TDIM = {0: 3, 1: 6, 2: 7}
_DDIM = {0: 4, 1: 7, 2: 8}


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _prep(t):
    if not t.is_cuda:
        raise RuntimeError("cut3r_slam_amd.lietorch: GPU tensors only (HIP kernels; no CPU path in the product)")
    return t.contiguous().float()


class _Unary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group, op):
        x = _prep(x)
        nin = _TDIM[group] if op == _OP_EXP else _DDIM[group]
        nout = {_OP_EXP: _DDIM[group], _OP_LOG: _TDIM[group], _OP_INV: _DDIM[group], _OP_MATRIX: 16}[op]
        if x.shape[-1] != nin:
            raise ValueError(f"expected last dim {nin}, got {tuple(x.shape)}")
        n = x.numel() // nin
        out = torch.empty(x.shape[:-1] + (nout,), dtype=torch.float32, device=x.device)
        if n:
            check(_lib.load().cut3r_lie_unary(group, op, _p(x), _p(out), n, _s()), "cut3r_lie_unary")
        ctx.save_for_backward(x)
        ctx.group, ctx.op = group, op
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _prep(g)
        gin = torch.empty_like(x)
        n = x.numel() // x.shape[-1]
        if n:
            check(_lib.load().cut3r_lie_unary_bwd(ctx.group, ctx.op, _p(x), _p(g), _p(gin), n, _s()), "cut3r_lie_unary_bwd")
        return gin, None, None


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, group):
        x, y = torch.broadcast_tensors(x, y)
        x, y = _prep(x), _prep(y)
        out = torch.empty_like(x)
        n = x.numel() // _DDIM[group]
        if n:
            check(_lib.load().cut3r_lie_mul(group, _p(x), _p(y), _p(out), n, _s()), "cut3r_lie_mul")
        ctx.save_for_backward(x, y)
        ctx.group = group
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = _prep(g)
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        n = x.numel() // _DDIM[ctx.group]
        if n:
            check(_lib.load().cut3r_lie_mul_bwd(ctx.group, _p(x), _p(y), _p(g), _p(gx), _p(gy), n, _s()), "cut3r_lie_mul_bwd")
        return gx, gy, None


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, group):
        """x [..., D], p [..., P, pd] with matching leading dims (already broadcast)"""
        x, p = _prep(x), _prep(p)
        pd = p.shape[-1]
        P = p.shape[-2]
        n = x.numel() // _DDIM[group]
        out = torch.empty_like(p)
        if n:
            check(_lib.load().cut3r_lie_act(group, _p(x), _p(p), _p(out), n, P, pd, _s()), "cut3r_lie_act")
        ctx.save_for_backward(x, p)
        ctx.group = group
        return out

    @staticmethod
    def backward(ctx, g):
        x, p = ctx.saved_tensors
        g = _prep(g)
        gx, gp = torch.empty_like(x), torch.empty_like(p)
        n = x.numel() // _DDIM[ctx.group]
        if n:
            check(_lib.load().cut3r_lie_act_bwd(ctx.group, _p(x), _p(p), _p(g), _p(gx), _p(gp), n, p.shape[-2], p.shape[-1], _s()),
                  "cut3r_lie_act_bwd")
        return gx, gp, None


class LieGroup:
    group_id = -1
    manifold_dim = 0
    embedded_dim = 0

    def __init__(self, data):
        if isinstance(data, LieGroup):
            data = data.data
        if data.shape[-1] != self.embedded_dim:
            raise ValueError(f"{type(self).__name__}: data must end with {self.embedded_dim}, got {tuple(data.shape)}")
        self.data = data

    # ---- construction
    @classmethod
    def exp(cls, a):
        return cls(_Unary.apply(a, cls.group_id, _OP_EXP))

    @classmethod
    def Identity(cls, *batch, device="cuda:0", dtype=torch.float32, **kw):
        d = torch.zeros(*batch, cls.embedded_dim, device=device, dtype=dtype)
        qw = 3 if cls.group_id == 0 else 6
        d[..., qw] = 1.0
        if cls.group_id == 2:
            d[..., 7] = 1.0
        return cls(d)

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(*G.shape, device=G.data.device)

    # ---- tensor-ish plumbing
    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    def __getitem__(self, index):
        return type(self)(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def view(self, *dims):
        return type(self)(self.data.view(*dims, self.embedded_dim))

    def to(self, *a, **k):
        return type(self)(self.data.to(*a, **k))

    def detach(self):
        return type(self)(self.data.detach())

    def vec(self):
        return self.data

    def tensor(self):
        return self.data

    # ---- group operations
    def log(self):
        return _Unary.apply(self.data, self.group_id, _OP_LOG)

    def inv(self):
        return type(self)(_Unary.apply(self.data, self.group_id, _OP_INV))

    def matrix(self):
        m = _Unary.apply(self.data, self.group_id, _OP_MATRIX)
        return m.view(*m.shape[:-1], 4, 4)

    def mul(self, other):
        return type(self)(_Mul.apply(self.data, other.data, self.group_id))

    def act(self, p):
        """points [..., 3] or homogeneous [..., 4]; leading dims broadcast against the group's batch shape"""
        pd = p.shape[-1]
        if pd not in (3, 4):
            raise ValueError("points must end with 3 or 4")
        lead = torch.broadcast_shapes(self.shape, p.shape[:-1])
        x = self.data.expand(lead + (self.embedded_dim,))
        q = p.expand(lead + (pd,))
        out = _Act.apply(x.reshape(-1, self.embedded_dim), q.reshape(-1, 1, pd), self.group_id)
        return out.view(lead + (pd,))

    def __mul__(self, other):
        if isinstance(other, LieGroup):
            return self.mul(other)
        return self.act(other)

    def retr(self, a):
        """exp(a) * X   (hislam2/geom/ba.py:29,37)"""
        return type(self).exp(a).mul(self)

    def adj(self, a):
        return self._adj(a, 0)

    def adjT(self, a):
        return self._adj(a, 1)

    def _adj(self, a, transpose):
        lead = torch.broadcast_shapes(self.shape, a.shape[:-1])
        x = _prep(self.data.expand(lead + (self.embedded_dim,)))
        v = _prep(a.expand(lead + (self.manifold_dim,)))
        out = torch.empty_like(v)
        n = v.numel() // self.manifold_dim
        if n:
            check(_lib.load().cut3r_lie_adj(self.group_id, _p(x), _p(v), _p(out), n, transpose, _s()), "cut3r_lie_adj")
        return out

    def translation(self):
        return self.data[..., 0:3] if self.group_id else torch.zeros_like(self.data[..., 0:3])

    def __repr__(self):
        return f"{type(self).__name__}: size={tuple(self.shape)}, device={self.device}"


class SO3(LieGroup):
    group_id, manifold_dim, embedded_dim = 0, 3, 4


class SE3(LieGroup):
    group_id, manifold_dim, embedded_dim = 1, 6, 7


class Sim3(LieGroup):
    group_id, manifold_dim, embedded_dim = 2, 7, 8


def cat(group_list, dim):
    return type(group_list[0])(torch.cat([g.data for g in group_list], dim=dim))


def stack(group_list, dim):
    return type(group_list[0])(torch.stack([g.data for g in group_list], dim=dim))
