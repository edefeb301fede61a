"""Thin torch-tensor front ends over the C ABI (include/cut3r_hip.h).  torch is used only for device memory and
the current HIP stream; every computation is a hand-written gfx950 kernel in libcut3r_hip.so.

All functions validate shapes/dtypes/strides on the host BEFORE launching (a faulting kernel can reset the GPU).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import GemmDesc, check

F16, F32 = torch.float16, torch.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _req(cond, msg):
    if not cond:
        raise ValueError(msg)


def _cuda(*ts):
    for t in ts:
        if t is not None:
            _req(t.is_cuda, "tensor must live on the GPU (no CPU fallback in the product path)")


# ------------------------------------------------------------------------------------------------ RoPE
def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """Drop-in for curope.rope_2d (reference src/croco/models/curope/curope.cpp:49-65): in place on a (B,N,H,D)
    view; same checks/errors as the reference's TORCH_CHECKs (RuntimeError on violation)."""
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise RuntimeError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise RuntimeError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    if tokens.is_cuda != positions.is_cuda:
        raise RuntimeError("tokens and positions are not on the same device")
    if not tokens.is_cuda:
        raise RuntimeError("cut3r_slam_amd.rope_2d: GPU tensors only (HIP kernel; no CPU path in the product)")
    B, N, H, D = tokens.shape
    if tokens.stride(3) != 1 or tokens.stride(2) != D:
        raise RuntimeError("tokens are not contiguous")
    if not positions.is_contiguous():
        raise RuntimeError("positions are not contiguous")
    if D % 4 != 0:
        raise RuntimeError("token dim must be multiple of 4")
    if D % 16 != 0:
        raise RuntimeError("cut3r_slam_amd.rope_2d: head dims that are multiples of 16 only (CUT3R uses 16/32/48/64)")
    if positions.dtype != torch.int64:
        raise RuntimeError("positions must be int64")
    dt = {F32: 0, F16: 1}.get(tokens.dtype)
    if dt is None:
        raise RuntimeError(f"unsupported token dtype {tokens.dtype}")
    lib = _lib.load()
    check(lib.cut3r_rope2d(_p(tokens), dt, _p(positions), B, N, H, D, tokens.stride(0), tokens.stride(1),
                           tokens.stride(2), float(base), float(fwd), _stream()), "cut3r_rope2d")


def rope_2d_qk(q, k, positions, base, fwd):
    """rope_2d(q) and rope_2d(k) with shared positions in one launch (self-attention).  q,k: (B,N,H,D) views, head stride D."""
    for t in (q, k):
        if t.dim() != 4 or t.stride(3) != 1 or t.stride(2) != t.size(3) or not t.is_cuda:
            raise RuntimeError("tokens are not contiguous")
    B, N, H, D = q.shape
    if k.shape != q.shape or positions.shape != (B, N, 2) or positions.dtype != torch.int64 or not positions.is_contiguous():
        raise RuntimeError("rope_2d_qk: shape/dtype mismatch")
    dt = {F32: 0, F16: 1}.get(q.dtype)
    if dt is None or k.dtype != q.dtype:
        raise RuntimeError(f"unsupported token dtype {q.dtype}")
    lib = _lib.load()
    check(lib.cut3r_rope2d_qk(_p(q), _p(k), dt, _p(positions), B, N, H, D, q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                              float(base), float(fwd), _stream()), "cut3r_rope2d_qk")


_ROPE_TABLE_LAUNCH = os.environ.get("CUT3R_ROPE_TABLE", "1") != "0"


def _rope_seg(t, pos):
    """(B,N,H,D) fp16 view with head stride D and a uniform token stride -> (ptr, pos ptr, tokens, token stride)"""
    if t.dim() != 4 or t.dtype != F16 or not t.is_cuda or t.stride(3) != 1 or t.stride(2) != t.size(3):
        raise RuntimeError("tokens are not contiguous")
    B, N = t.shape[:2]
    if B > 1 and t.stride(0) != N * t.stride(1):
        raise RuntimeError("rope_2d_pair: the token stride must be uniform over the batch")
    if pos.shape != (B, N, 2) or pos.dtype != torch.int64 or not pos.is_contiguous() or not pos.is_cuda:
        raise RuntimeError("positions must be contiguous int64 [B,N,2]")
    return _p(t), _p(pos), B * N, t.stride(1)


def rope_2d_pair(t0, pos0, t1, pos1, base, fwd=1.0):
    """rope_2d(t0, pos0) and rope_2d(t1, pos1) in ONE table-driven launch (fp16; results are bit-identical to rope_2d): q and k of a
    self-attention (pos1 is pos0) or of a cross-attention (two token streams).  t1 may be None."""
    H, D = t0.shape[2:]
    if D % 16 != 0:
        raise RuntimeError("cut3r_slam_amd.rope_2d: head dims that are multiples of 16 only (CUT3R uses 16/32/48/64)")
    if not _ROPE_TABLE_LAUNCH:            # A/B knob: the one-workgroup-per-token kernel (same bits)
        if t1 is not None and pos1 is pos0 and t1.shape == t0.shape:
            return rope_2d_qk(t0, t1, pos0, base, fwd)
        rope_2d(t0, pos0, base, fwd)
        if t1 is not None:
            rope_2d(t1, pos1, base, fwd)
        return
    if t1 is not None and tuple(t1.shape[2:]) != (H, D):
        raise RuntimeError("rope_2d_pair: both tensors must have the same heads and head dim")
    a = _rope_seg(t0, pos0)
    b = _rope_seg(t1, pos1) if t1 is not None else (None, None, 0, 0)
    tab = rope_table(t0.device, base, fwd, D)
    lib = _lib.load()
    check(lib.cut3r_rope2d_tab(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], H, D, _p(tab), ROPE_PMIN, ROPE_NPOS, float(base), float(fwd),
                               _stream()), "cut3r_rope2d_tab")


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm(x, gamma, beta, eps=1e-6, out16=None, out32=None, mod_scale=None, mod_shift=None):
    _cuda(x, gamma, beta, out16, out32, mod_scale, mod_shift)
    _req(x.dtype == F32 and x.dim() == 2 and x.stride(1) == 1, "x must be fp32 [M,C] with unit inner stride")
    M, Cc = x.shape
    _req(gamma.dtype == F32 and beta.dtype == F32 and gamma.numel() == Cc and beta.numel() == Cc, "gamma/beta [C] fp32")
    _req(gamma.is_contiguous() and beta.is_contiguous(), "gamma/beta contiguous")
    for o, dt in ((out16, F16), (out32, F32)):
        if o is not None:
            _req(o.dtype == dt and o.shape == x.shape and o.stride(1) == 1, "bad LN output tensor")
    for m in (mod_scale, mod_shift):
        if m is not None:
            _req(m.dtype == F32 and m.numel() == Cc and m.is_contiguous(), "modulation vectors must be fp32 [C]")
    lib = _lib.load()
    check(lib.cut3r_layernorm(_p(x), x.stride(0), _p(gamma), _p(beta), float(eps), M, Cc,
                              _p(out16), out16.stride(0) if out16 is not None else 0,
                              _p(out32), out32.stride(0) if out32 is not None else 0,
                              _p(mod_scale), _p(mod_shift), _stream()), "cut3r_layernorm")


def layernorm_dual(x, g1, b1, out1, g2, b2, out2, eps=1e-6):
    """out1 = LN(x; g1, b1), out2 = LN(x; g2, b2), both fp16, one pass over x (C in {768, 1024, 1536})"""
    _cuda(x, g1, b1, out1, g2, b2, out2)
    M, Cc = x.shape
    _req(x.dtype == F32 and x.stride(1) == 1 and Cc in (768, 1024, 1536), "x fp32 [M,C], C in {768,1024,1536}")
    for o in (out1, out2):
        _req(o.dtype == F16 and o.shape == (M, Cc) and o.stride(1) == 1, "outputs fp16 [M,C]")
    for t in (g1, b1, g2, b2):
        _req(t.dtype == F32 and t.numel() == Cc and t.is_contiguous(), "gamma/beta fp32 [C]")
    lib = _lib.load()
    check(lib.cut3r_layernorm_dual(_p(x), x.stride(0), _p(g1), _p(b1), _p(out1), out1.stride(0), _p(g2), _p(b2), _p(out2), out2.stride(0),
                                   float(eps), M, Cc, _stream()), "cut3r_layernorm_dual")


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_STAGES = int(os.environ.get("CUT3R_GEMM_STAGES", "0"))     # tuning override (0 = kernel default)


def _fill_common(d, A, Bw, out, bias, res1, res2, act, tile):
    d.A, d.B, d.C = A.data_ptr(), Bw.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.act = act
    d.out_f16 = 1 if out.dtype == F16 else 0
    d.batch = 1
    d.tile = tile
    d.stages = GEMM_STAGES
    for i, r in ((1, res1), (2, res2)):
        if r is not None:
            _req(r.dtype in (F16, F32) and r.stride(-1) == 1, "residual must be fp16/fp32 with unit inner stride")
            setattr(d, f"res{i}", r.data_ptr())
            setattr(d, f"ldr{i}", r.stride(-2))
            setattr(d, f"res{i}_f16", 1 if r.dtype == F16 else 0)


_ROPE_TABLES = {}
ROPE_PMIN, ROPE_NPOS = -1, 258        # positions -1 (pose token) .. 256


def rope_table(device, base, fwd=1.0, head_dim=64):
    """cos|sin table [2, ROPE_NPOS, head_dim/4] of the RoPE angles (cut3r_rope2d_table), cached per device"""
    key = (str(device), float(base), float(fwd), int(head_dim))
    t = _ROPE_TABLES.get(key)
    if t is None:
        t = torch.empty(2, ROPE_NPOS, head_dim // 4, dtype=F32, device=device)
        lib = _lib.load()
        check(lib.cut3r_rope2d_table(_p(t), ROPE_PMIN, ROPE_NPOS, head_dim // 4, float(base), float(fwd), _stream()), "cut3r_rope2d_table")
        _ROPE_TABLES[key] = t
    return t


def linear(A, W, out, bias=None, act=0, res1=None, res2=None, tile=0, rope=None, ln=None, emit=None):
    """out[M,N] = act(A[M,K] @ W[N,K]^T + bias) (+res1)(+res2).  A,W fp16; out fp16|fp32 (any row stride).
    rope = (positions int64 [M,2] contiguous, cols, base[, head_dim = 64 | 48]): 2-D RoPE fused on the first `cols` columns
    (48-wide heads use the 128 x 192 tile).
    LayerNorm fold (include/cut3r_hip.h, cut3r_gemm_desc):
      ln = (stats fp32 [K/64, M, 2] (slab-major), colsum fp32 [N], eps): A holds the UN-normalised rows (fp16 copy of the residual stream), W the
           gamma-folded panel, bias the folded d; the epilogue normalises per row from the slab statistics;
      emit = (stats_out fp32 [N/64, M, 2], out16 fp16 [M, N]): an fp32 + fp32-residual GEMM also writes the fp16 copy of its
           output and the slab statistics the next consumer needs."""
    _cuda(A, W, out, bias, res1, res2)
    _req(A.dtype == F16 and W.dtype == F16 and A.dim() == 2 and W.dim() == 2, "A,W must be 2-D fp16")
    M, K = A.shape
    N = W.shape[0]
    _req(W.shape[1] == K and out.shape == (M, N), f"shape mismatch A{tuple(A.shape)} W{tuple(W.shape)} out{tuple(out.shape)}")
    _req(A.stride(1) == 1 and W.stride(1) == 1 and out.stride(1) == 1, "unit inner strides required")
    _req(out.dtype in (F16, F32), "out must be fp16 or fp32")
    if bias is not None:
        _req(bias.dtype == F32 and bias.numel() == N and bias.is_contiguous(), "bias fp32 [N]")
    for r in (res1, res2):
        if r is not None:
            _req(r.shape == (M, N), "residual shape")
    if rope is not None:
        pos, cols, base = rope[:3]
        hd = rope[3] if len(rope) > 3 else 64
        _req(pos.dtype == torch.int64 and pos.is_contiguous() and pos.numel() == 2 * M and pos.is_cuda, "rope positions int64 [M,2]")
        _req(out.dtype == F16 and act == 0 and res1 is None and res2 is None and hd in (48, 64) and cols % hd == 0 and 0 < cols <= N
             and N % hd == 0 and tile != 16, "fused rope: fp16 output, no activation / residual, whole heads of 64 or 48")
    if tile == 16 and M > 64:           # skinny kernel: 64 rows per launch (row results do not depend on the chunking)
        for m0 in range(0, M, 64):
            sl = slice(m0, min(M, m0 + 64))
            linear(A[sl], W, out[sl], bias, act, None if res1 is None else res1[sl], None if res2 is None else res2[sl], tile=16)
        return out
    d = GemmDesc()
    _fill_common(d, A, W, out, bias, res1, res2, act, tile)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, A.stride(0), W.stride(0), out.stride(0)
    if rope is not None:
        tab = rope_table(out.device, base, 1.0, hd)
        d.rope_pos, d.rope_table, d.rope_cols, d.rope_pmin, d.rope_npos = pos.data_ptr(), tab.data_ptr(), int(cols), ROPE_PMIN, ROPE_NPOS
        d.rope_d = hd
    if ln is not None:
        st, cs, eps = ln
        _cuda(st, cs)
        _req(K % 64 == 0 and st.dtype == F32 and st.is_contiguous() and st.numel() == M * (K // 64) * 2, f"ln stats fp32 [K/64={K // 64}, M={M}, 2]")
        _req(cs.dtype == F32 and cs.is_contiguous() and cs.numel() == N and bias is not None and tile != 16, "ln colsum fp32 [N], folded bias required")
        d.ln_stats, d.ln_colsum, d.ln_nslab, d.ln_eps = st.data_ptr(), cs.data_ptr(), K // 64, float(eps)
    if emit is not None:
        so, o16 = emit
        _cuda(so, o16)
        _req(N % 64 == 0 and so.dtype == F32 and so.is_contiguous() and so.numel() == M * (N // 64) * 2, f"stats_out fp32 [N/64={N // 64}, M={M}, 2]")
        _req(o16.dtype == F16 and o16.shape == (M, N) and o16.stride(1) == 1, "out16 fp16 [M,N]")
        _req(out.dtype == F32 and res1 is not None and res1.dtype == F32 and res2 is None and act == 0 and tile != 16, "emit: fp32 output with an fp32 residual")
        d.stats_out, d.out16, d.ld16 = so.data_ptr(), o16.data_ptr(), o16.stride(0)
    lib = _lib.load()
    check(lib.cut3r_gemm_f16(C.byref(d), _stream()), f"cut3r_gemm_f16 M={M} N={N} K={K}")
    return out


def _linear_desc(A, W, out, bias, act, res1, tile, rope=None, ln=None, emit=None):
    _cuda(A, W, out, bias, res1)
    _req(A.dtype == F16 and W.dtype == F16 and A.dim() == 2 and W.dim() == 2, "A,W must be 2-D fp16")
    M, K = A.shape
    N = W.shape[0]
    _req(W.shape[1] == K and out.shape == (M, N), f"shape mismatch A{tuple(A.shape)} W{tuple(W.shape)} out{tuple(out.shape)}")
    _req(A.stride(1) == 1 and W.stride(1) == 1 and out.stride(1) == 1, "unit inner strides required")
    _req(out.dtype in (F16, F32), "out must be fp16 or fp32")
    if bias is not None:
        _req(bias.dtype == F32 and bias.numel() == N and bias.is_contiguous(), "bias fp32 [N]")
    if res1 is not None:
        _req(res1.shape == (M, N), "residual shape")
    d = GemmDesc()
    _fill_common(d, A, W, out, bias, res1, None, act, tile)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, A.stride(0), W.stride(0), out.stride(0)
    if rope is not None:
        pos, cols, base = rope[:3]
        hd = rope[3] if len(rope) > 3 else 64
        _req(pos.dtype == torch.int64 and pos.is_contiguous() and pos.numel() == 2 * M and pos.is_cuda and hd == 64 and out.dtype == F16, "pair rope: heads of 64")
        tab = rope_table(out.device, base, 1.0, hd)
        d.rope_pos, d.rope_table, d.rope_cols, d.rope_pmin, d.rope_npos, d.rope_d = pos.data_ptr(), tab.data_ptr(), int(cols), ROPE_PMIN, ROPE_NPOS, hd
    if ln is not None:
        st, cs, eps = ln
        _cuda(st, cs)
        _req(K % 64 == 0 and st.dtype == F32 and st.is_contiguous() and st.numel() == M * (K // 64) * 2 and cs.dtype == F32 and cs.numel() == N and bias is not None,
             "ln: stats fp32 [K/64, M, 2], colsum fp32 [N], folded bias")
        d.ln_stats, d.ln_colsum, d.ln_nslab, d.ln_eps = st.data_ptr(), cs.data_ptr(), K // 64, float(eps)
    if emit is not None:
        so, o16 = emit
        _cuda(so, o16)
        _req(N % 64 == 0 and so.dtype == F32 and so.is_contiguous() and so.numel() == M * (N // 64) * 2 and o16.dtype == F16 and o16.shape == (M, N)
             and out.dtype == F32 and res1 is not None and res1.dtype == F32 and act == 0, "emit: stats_out fp32 [N/64, M, 2], out16 fp16 [M,N], fp32 out + residual")
        d.stats_out, d.out16, d.ld16 = so.data_ptr(), o16.data_ptr(), o16.stride(0)
    return d


def linear_pair(p0, p1, act=0, tile=0):
    """Two independent linears in ONE launch (cut3r_gemm_f16_pair): p = (A [M,K], W [N,K], out [M,N], bias | None, res1 | None[, extras]) with
    the same N and K; rows are bit-identical to ops.linear.  The decoder runs its state-side and image-side projection of a layer this
    way.  extras: dict with any of rope / ln / emit as in ops.linear (per problem)."""
    x0 = p0[5] if len(p0) > 5 and p0[5] else {}
    x1 = p1[5] if len(p1) > 5 and p1[5] else {}
    d0 = _linear_desc(p0[0], p0[1], p0[2], p0[3], act, p0[4], tile, **x0)
    d1 = _linear_desc(p1[0], p1[1], p1[2], p1[3], act, p1[4], tile, **x1)
    _req(d0.N == d1.N and d0.K == d1.K, "pair: same N and K")
    lib = _lib.load()
    check(lib.cut3r_gemm_f16_pair(C.byref(d0), C.byref(d1), _stream()), f"cut3r_gemm_f16_pair M={d0.M}+{d1.M} N={d0.N} K={d0.K}")


def linear_batched(A, W, out, bias=None, act=0, res1=None, tile=0):
    """Z independent problems in ONE launch (blockIdx.z): A [Z,M,K], W [Z,N,K] fp16; out [Z,M,N] fp16|fp32;
    bias [Z,N] fp32; res1 [Z,M,N].  Used to run the state-side and image-side decoder GEMMs of a layer together."""
    _cuda(A, W, out, bias, res1)
    _req(A.dtype == F16 and W.dtype == F16 and A.dim() == 3 and W.dim() == 3 and out.dim() == 3, "3-D fp16 operands")
    Z, M, K = A.shape
    N = W.shape[1]
    _req(W.shape == (Z, N, K) and out.shape == (Z, M, N), "batched shapes")
    _req(A.stride(2) == 1 and W.stride(2) == 1 and out.stride(2) == 1, "unit inner strides")
    d = GemmDesc()
    _fill_common(d, A, W, out, bias, None, None, act, tile)
    if bias is not None:
        _req(bias.dtype == F32 and bias.shape == (Z, N) and bias.stride(1) == 1, "bias [Z,N]")
        d.strideBias = bias.stride(0)
    if res1 is not None:
        _req(res1.shape == (Z, M, N) and res1.stride(2) == 1 and res1.dtype in (F16, F32), "res1 [Z,M,N]")
        d.res1, d.ldr1, d.res1_f16, d.strideR1 = res1.data_ptr(), res1.stride(1), int(res1.dtype == F16), res1.stride(0)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, A.stride(1), W.stride(1), out.stride(1)
    d.batch, d.strideA, d.strideB, d.strideC = Z, A.stride(0), W.stride(0), out.stride(0)
    lib = _lib.load()
    check(lib.cut3r_gemm_f16(C.byref(d), _stream()), f"cut3r_gemm_f16 batched Z={Z} M={M} N={N} K={K}")
    return out


def conv3x3_nhwc(x, Wk, out, bias=None, stride=1, relu_in=False, act=0, res1=None, res2=None, tile=0):
    """3x3 / pad 1 convolution as implicit GEMM.  x fp16 [B,H,W,Cin] contiguous; Wk fp16 [Cout, 9*Cin] ordered
    (ky,kx,ci); out fp16 [B,Ho,Wo,Cout] contiguous."""
    _cuda(x, Wk, out, bias, res1, res2)
    _req(x.dtype == F16 and x.dim() == 4 and x.is_contiguous(), "x must be contiguous NHWC fp16")
    Bn, H, Wd, Cin = x.shape
    Cout = Wk.shape[0]
    Ho, Wo = (H + 2 - 3) // stride + 1, (Wd + 2 - 3) // stride + 1
    _req(Wk.dtype == F16 and Wk.shape == (Cout, 9 * Cin) and Wk.is_contiguous(), "Wk must be fp16 [Cout, 9*Cin]")
    _req(out.shape == (Bn, Ho, Wo, Cout) and out.is_contiguous() and out.dtype in (F16, F32), "bad conv output")
    M = Bn * Ho * Wo
    r1 = res1.reshape(M, Cout) if res1 is not None else None
    r2 = res2.reshape(M, Cout) if res2 is not None else None
    d = GemmDesc()
    _fill_common(d, x, Wk, out, bias, r1, r2, act, tile)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, Cout, 9 * Cin, Cin, 9 * Cin, Cout
    d.conv_k, d.H, d.W, d.Cin, d.conv_stride, d.Ho, d.Wo, d.relu_in = 3, H, Wd, Cin, stride, Ho, Wo, int(relu_in)
    lib = _lib.load()
    check(lib.cut3r_gemm_f16(C.byref(d), _stream()), f"cut3r_gemm_f16(conv3x3) M={M} N={Cout} K={9*Cin}")
    return out


def conv_transpose_nhwc(x, Wt, out, bias, s):
    """ConvTranspose2d(kernel == stride == s).  x fp16 [B,H,W,Cin]; Wt fp16 [s*s*Cout, Cin] ordered (i,j,co);
    out fp16 [B,H*s,W*s,Cout]."""
    _cuda(x, Wt, out, bias)
    Bn, H, Wd, Cin = x.shape
    Cout = Wt.shape[0] // (s * s)
    _req(x.dtype == F16 and x.is_contiguous() and Wt.dtype == F16 and Wt.is_contiguous(), "fp16 contiguous inputs")
    _req(Wt.shape == (s * s * Cout, Cin) and out.shape == (Bn, H * s, Wd * s, Cout) and out.is_contiguous(), "bad shapes")
    _req(bias is not None and bias.numel() == Cout, "bias [Cout]")
    d = GemmDesc()
    _fill_common(d, x, Wt, out, bias, None, None, 0, 0)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = Bn * H * Wd, s * s * Cout, Cin, Cin, Cin, Cout
    d.shuf, d.shuf_cout, d.shuf_Hin, d.shuf_Win = s, Cout, H, Wd
    lib = _lib.load()
    check(lib.cut3r_gemm_f16(C.byref(d), _stream()), "cut3r_gemm_f16(convT)")
    return out


def gemv(X, W, out, bias=None, act=0, res=None, silu_in=False):
    """out[M,N] = act(X[M,K] @ W[N,K]^T + bias) (+res); X,out fp32, W fp16, M <= 64."""
    _cuda(X, W, out, bias, res)
    _req(X.dtype == F32 and W.dtype == F16 and out.dtype == F32, "dtypes: X fp32, W fp16, out fp32")
    M, K = X.shape
    N = W.shape[0]
    _req(W.shape[1] == K and out.shape == (M, N) and X.stride(1) == 1 and W.stride(1) == 1 and out.stride(1) == 1, "shapes")
    if bias is not None:
        _req(bias.dtype == F32 and bias.numel() == N, "bias")
    if res is not None:
        _req(res.dtype == F32 and res.shape == (M, N) and res.stride(1) == 1, "res")
    lib = _lib.load()
    check(lib.cut3r_gemv_f16w(_p(X), X.stride(0), _p(W), W.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, act,
                              _p(res), res.stride(0) if res is not None else 0, int(silu_in), _stream()), "cut3r_gemv_f16w")
    return out


# ------------------------------------------------------------------------------------------------ attention
def attention(q, k, v, out, scale):
    """q [B,Nq,H,D], k/v [B,Nk,H,D] fp16 views with unit d-stride and head stride D; out [B,Nq,H,D] fp16."""
    _cuda(q, k, v, out)
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    for t in (q, k, v, out):
        _req(t.dtype == F16 and t.dim() == 4 and t.stride(3) == 1 and t.stride(2) == D, "attention operands: fp16 (B,N,H,D) views")
    _req(k.shape == (B, Nk, H, D) and v.shape == (B, Nk, H, D) and out.shape == (B, Nq, H, D), "attention shapes")
    _req(D in (16, 32, 48, 64, 128), f"unsupported head dim {D}")
    lib = _lib.load()
    check(lib.cut3r_attention_f16(_p(q), _p(k), _p(v), _p(out), B, H, Nq, Nk, D, q.stride(0), q.stride(1), k.stride(0),
                                  k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1), float(scale),
                                  _stream()), f"cut3r_attention_f16 B={B} H={H} Nq={Nq} Nk={Nk} D={D}")
    return out


# ------------------------------------------------------------------------------------------------ helpers
def im2col_patch(img, P, out):
    _cuda(img, out)
    _req(img.dim() == 4 and img.is_contiguous() and img.dtype in (F32, torch.uint8), "img [B,C,H,W] fp32|u8 contiguous")
    B, Cc, H, W = img.shape
    _req(out.dtype == F16 and out.is_contiguous() and out.shape == (B * (H // P) * (W // P), Cc * P * P), "im2col out shape")
    lib = _lib.load()
    check(lib.cut3r_im2col_patch(_p(img), int(img.dtype == torch.uint8), B, Cc, H, W, P, _p(out), _stream()), "cut3r_im2col_patch")
    return out


def cast_f16(x, out):
    _cuda(x, out)
    _req(x.dtype == F32 and out.dtype == F16 and x.shape == out.shape and x.dim() == 2, "cast: 2-D fp32 -> fp16")
    _req(x.stride(1) == 1 and out.stride(1) == 1, "unit inner stride")
    lib = _lib.load()
    check(lib.cut3r_cast_f32_f16(_p(x), x.stride(0), _p(out), out.stride(0), x.shape[0], x.shape[1], _stream()), "cut3r_cast_f32_f16")
    return out


def colmean(x, out):
    _cuda(x, out)
    _req(x.dtype == F32 and x.dim() == 2 and x.stride(1) == 1 and out.dtype == F32 and out.numel() == x.shape[1], "colmean")
    lib = _lib.load()
    check(lib.cut3r_colmean(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), _stream()), "cut3r_colmean")
    return out


def colmean_batched(x, out):
    """x fp32 [B,M,C] (any batch / row stride, unit inner stride) -> out fp32 [B,C]: per-matrix column means, one launch"""
    _cuda(x, out)
    _req(x.dtype == F32 and x.dim() == 3 and x.stride(2) == 1 and out.dtype == F32 and out.shape == (x.shape[0], x.shape[2]) and out.stride(1) == 1,
         "colmean_batched")
    lib = _lib.load()
    check(lib.cut3r_colmean_batched(_p(x), x.shape[0], x.stride(0), x.stride(1), x.shape[1], x.shape[2], _p(out), out.stride(0), _stream()),
          "cut3r_colmean_batched")
    return out


def upsample2x(x, out):
    _cuda(x, out)
    B, H, W, Cc = x.shape
    _req(x.dtype == F16 and x.is_contiguous() and out.dtype == F16 and out.is_contiguous() and out.shape == (B, 2 * H, 2 * W, Cc), "upsample2x")
    lib = _lib.load()
    check(lib.cut3r_upsample2x_nhwc(_p(x), _p(out), B, H, W, Cc, _stream()), "cut3r_upsample2x_nhwc")
    return out


def dpt_final(x, w, b, mode, pts, conf=None):
    _cuda(x, w, b, pts, conf)
    P, Cin = x.shape
    nout = 4 if mode == 0 else 3
    _req(x.dtype == F16 and x.is_contiguous() and w.dtype == F32 and w.shape == (nout, Cin) and w.is_contiguous(), "dpt_final in/w")
    _req(b.dtype == F32 and b.numel() == nout and pts.dtype == F32 and pts.numel() == 3 * P and pts.is_contiguous(), "dpt_final b/pts")
    if mode == 0:
        _req(conf is not None and conf.dtype == F32 and conf.numel() == P and conf.is_contiguous(), "conf")
    lib = _lib.load()
    check(lib.cut3r_dpt_final(_p(x), P, Cin, _p(w), _p(b), mode, _p(pts), _p(conf), _stream()), "cut3r_dpt_final")


def postprocess_pts(raw, pos_z, pts, conf=None):
    _cuda(raw, pts, conf)
    P, nch = raw.shape
    _req(raw.dtype == F32 and raw.is_contiguous() and pts.dtype == F32 and pts.numel() == 3 * P and pts.is_contiguous(), "postprocess_pts")
    if nch == 4:
        _req(conf is not None and conf.numel() == P and conf.dtype == F32 and conf.is_contiguous(), "conf")
    lib = _lib.load()
    check(lib.cut3r_postprocess_pts(_p(raw), P, nch, int(pos_z), _p(pts), _p(conf), _stream()), "cut3r_postprocess_pts")


def postprocess_pose(raw, out):
    _cuda(raw, out)
    _req(raw.dtype == F32 and raw.is_contiguous() and raw.shape[-1] == 7 and out.shape == raw.shape and out.is_contiguous(), "pose")
    lib = _lib.load()
    check(lib.cut3r_postprocess_pose(_p(raw), raw.numel() // 7, _p(out), _stream()), "cut3r_postprocess_pose")
    return out


# ------------------------------------------------------------------------------------------------ geometry
def patch_overlap_count(feat0, feat1, thr, ws, count):
    _cuda(feat0, feat1, ws, count)
    N, Cc = feat0.shape
    _req(feat0.dtype == F32 and feat1.dtype == F32 and feat1.shape == (N, Cc) and feat0.is_contiguous() and feat1.is_contiguous(), "features fp32 [N,C]")
    _req(ws.dtype == F32 and ws.numel() >= 2 * (N - 1) * Cc + (N - 1) and count.dtype == torch.int32, "workspace/count")
    lib = _lib.load()
    check(lib.cut3r_patch_overlap(_p(feat0), _p(feat1), N, Cc, float(thr), _p(ws), _p(count), _stream()), "cut3r_patch_overlap")


def patch_overlap_chain(feat_last, feats, thr_sim, thr_ratio, forced, ws, state, counts, decisions):
    """Keyframe decisions of B consecutive tested frames on the device (motion_filter.py:98-124 without a host round trip per
    frame): feat_last [N,C], feats [B,N,C] fp32; forced: list of B bools (always-keyframe frames) or None;
    state int32[1], counts / decisions int32[B] on the device."""
    _cuda(feat_last, feats, ws, state, counts, decisions)
    B, N, Cc = feats.shape
    _req(feats.dtype == F32 and feat_last.dtype == F32 and feat_last.shape == (N, Cc) and feats.is_contiguous() and feat_last.is_contiguous(),
         "features fp32 [B,N,C] / [N,C]")
    _req(ws.dtype == F32 and ws.numel() >= (B + 1) * (N - 1) * Cc + (N - 1), "workspace of (B+1)*(N-1)*C + N-1 floats")
    _req(state.dtype == torch.int32 and counts.dtype == torch.int32 and decisions.dtype == torch.int32 and counts.numel() >= B and decisions.numel() >= B,
         "state / counts / decisions int32")
    arr = (C.c_int32 * B)(*[1 if f else 0 for f in forced]) if forced is not None else None
    lib = _lib.load()
    check(lib.cut3r_patch_overlap_chain(_p(feat_last), _p(feats), B, N, Cc, float(thr_sim), float(thr_ratio), arr, _p(ws), _p(state), _p(counts),
                                        _p(decisions), _stream()), "cut3r_patch_overlap_chain")


def overlap_fwd(pm, w2c, K4, W, H, counts, P12=None, s_align=1.0, clamp_z=True):
    """counts[b] = #points of pm (optionally mapped p <- P*(s*p) first) that land inside camera b's W x H image."""
    _cuda(pm, w2c, counts)
    N = pm.numel() // 3
    B = w2c.shape[0]
    _req(pm.dtype == F32 and pm.is_contiguous() and w2c.dtype == F32 and w2c.shape == (B, 12) and w2c.is_contiguous(), "overlap_fwd inputs")
    _req(counts.dtype == torch.int32 and counts.numel() >= B, "counts int32[B]")
    arr = (C.c_float * 12)(*[float(v) for v in P12]) if P12 is not None else None
    lib = _lib.load()
    check(lib.cut3r_overlap_fwd(_p(pm), N, arr, float(s_align), _p(w2c), B, *[float(v) for v in K4], int(W), int(H),
                                int(clamp_z), _p(counts), _stream()), "cut3r_overlap_fwd")


def overlap_bwd(pms, w2c, K4, W, H, counts, B=None, N=None, grp=0, grp_stride=0):
    """counts[b] = #points of pointmap b inside the ONE camera w2c.  pms: contiguous [B,N,3]-like store; with grp > 0
    pointmap b sits at slot (b//grp)*grp_stride + b%grp of a [slots, N, 3] store (keyframe order of submap_ds)."""
    _cuda(pms, w2c, counts)
    if B is None:
        B = pms.shape[0]
    if N is None:
        N = pms[0].numel() // 3
    _req(pms.dtype == F32 and pms.is_contiguous() and w2c.dtype == F32 and w2c.numel() == 12 and w2c.is_contiguous(), "overlap_bwd inputs")
    nslots = pms.numel() // (3 * N)
    last = (B - 1) if grp == 0 else ((B - 1) // grp) * grp_stride + (B - 1) % grp
    _req(B >= 1 and last < nslots, "overlap_bwd: pointmap store too small for B")
    _req(counts.dtype == torch.int32 and counts.numel() >= B, "counts int32[B]")
    lib = _lib.load()
    check(lib.cut3r_overlap_bwd(_p(pms), B, N, int(grp), int(grp_stride), _p(w2c), *[float(v) for v in K4], int(W), int(H),
                                _p(counts), _stream()), "cut3r_overlap_bwd")


def align_view(pts, conf, P12, s, ds, pm_ds, conf_ds, depth):
    _cuda(pts, conf, pm_ds, conf_ds, depth)
    H, W = conf.shape[-2:]
    _req(pts.dtype == F32 and pts.is_contiguous() and pts.numel() == H * W * 3 and conf.dtype == F32 and conf.is_contiguous(), "align inputs")
    _req(pm_ds.is_contiguous() and pm_ds.numel() == (H // ds) * (W // ds) * 3 and conf_ds.is_contiguous() and conf_ds.numel() == (H // ds) * (W // ds), "align ds outputs")
    _req(depth.is_contiguous() and depth.numel() == H * W and depth.dtype == F32, "depth output")
    arr = (C.c_float * 12)(*[float(v) for v in P12])
    lib = _lib.load()
    check(lib.cut3r_align_view(_p(pts), _p(conf), H, W, arr, float(s), int(ds), _p(pm_ds), _p(conf_ds), _p(depth), _stream()), "cut3r_align_view")


def window_update(pts, conf, P12s, s, ds, pm_ds, conf_ds, depth, store, w2c, t0, first, K4, counts, grp=5, grp_stride=6,
                  w2c_new=None, lsum_reset=None):
    """Whole-window align + overlap counts (cut3r_window_update).  pts [V,H,W,3], conf [V,H,W]; pm_ds [V,h,w,3], conf_ds [V,h,w],
    depth [V,H,W] are the window's consecutive store slots; store = the full [submaps,6,h,w,3] pointmap store; w2c [>=t0+V,12];
    counts int32 [V,2,ldc] receives forward (row 0) / backward (row 1) counts of keyframes t0+v >= first.  w2c_new: V*12 host
    floats written to rows t0.. of w2c by the kernel; lsum_reset: fp64[1] accumulator zeroed for the next window."""
    _cuda(pts, conf, pm_ds, conf_ds, depth, store, w2c, counts)
    V, H, W = conf.shape
    h, w = H // ds, W // ds
    _req(pts.dtype == F32 and pts.is_contiguous() and pts.shape == (V, H, W, 3) and conf.dtype == F32 and conf.is_contiguous(), "window inputs")
    _req(1 <= V <= 6 and len(P12s) == V * 12, "V <= 6 views with 12 floats each")
    _req(pm_ds.is_contiguous() and pm_ds.numel() == V * h * w * 3 and conf_ds.is_contiguous() and conf_ds.numel() == V * h * w, "ds outputs")
    _req(depth.is_contiguous() and depth.numel() == V * H * W and depth.dtype == F32, "depth rows")
    _req(store.dtype == F32 and store.is_contiguous() and w2c.dtype == F32 and w2c.is_contiguous() and w2c.shape[0] >= t0 + V and w2c.shape[1] == 12, "store / w2c")
    last = t0 + V - 1
    nslots = store.numel() // (3 * h * w)
    _req(last < 1 or ((last - 1) // grp) * grp_stride + (last - 1) % grp < nslots, "pointmap store too small")
    ldc = counts.shape[-1]
    _req(counts.dtype == torch.int32 and counts.is_contiguous() and counts.shape == (V, 2, ldc) and ldc >= t0 + V, "counts int32 [V,2,ldc]")
    arr = (C.c_float * (12 * V))(*[float(v) for v in P12s])
    arr_w = None
    if w2c_new is not None:
        _req(len(w2c_new) == 12 * V, "w2c_new: V*12 floats")
        arr_w = (C.c_float * (12 * V))(*[float(v) for v in w2c_new])
    if lsum_reset is not None:
        _cuda(lsum_reset)
        _req(lsum_reset.dtype == torch.float64 and lsum_reset.numel() == 1, "lsum_reset fp64[1]")
    lib = _lib.load()
    check(lib.cut3r_window_update(_p(pts), _p(conf), V, H, W, arr, float(s), int(ds), _p(pm_ds), _p(conf_ds), _p(depth), _p(store),
                                  int(grp), int(grp_stride), _p(w2c), arr_w, int(t0), int(first), *[float(v) for v in K4], _p(counts), ldc,
                                  _p(lsum_reset), _stream()), "cut3r_window_update")


def logdepth_sum(prev_depth, pts, out):
    _cuda(prev_depth, pts, out)
    n = prev_depth.numel()
    _req(prev_depth.dtype == F32 and prev_depth.is_contiguous() and pts.dtype == F32 and pts.is_contiguous() and pts.numel() == 3 * n, "logdepth inputs")
    _req(out.dtype == torch.float64 and out.numel() == 1, "out fp64[1]")
    lib = _lib.load()
    check(lib.cut3r_logdepth_sum(_p(prev_depth), _p(pts), n, _p(out), _stream()), "cut3r_logdepth_sum")


def logdepth_accum(prev_depth, pts, out):
    """out += sum(log prev - log z); out must hold 0 beforehand (window_update's lsum_reset keeps it so)."""
    _cuda(prev_depth, pts, out)
    n = prev_depth.numel()
    _req(prev_depth.dtype == F32 and prev_depth.is_contiguous() and pts.dtype == F32 and pts.is_contiguous() and pts.numel() == 3 * n, "logdepth inputs")
    _req(out.dtype == torch.float64 and out.numel() == 1, "out fp64[1]")
    lib = _lib.load()
    check(lib.cut3r_logdepth_accum(_p(prev_depth), _p(pts), n, _p(out), _stream()), "cut3r_logdepth_accum")


def remap_linear_u8(src_hwc, map_ix, map_iy, out=None):
    """cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) with a fixed-point map (1/32 pixel): src uint8 [H,W,C]; map_ix/map_iy int32 [Ho,Wo]"""
    _cuda(src_hwc, map_ix, map_iy, out)
    _req(src_hwc.dtype == torch.uint8 and src_hwc.dim() == 3 and src_hwc.is_contiguous(), "src uint8 [H,W,C] contiguous")
    H, W, Cc = src_hwc.shape
    _req(map_ix.dtype == torch.int32 and map_iy.dtype == torch.int32 and map_ix.shape == map_iy.shape and map_ix.dim() == 2
         and map_ix.is_contiguous() and map_iy.is_contiguous(), "maps int32 [Ho,Wo]")
    Ho, Wo = map_ix.shape
    if out is None:
        out = torch.empty((Ho, Wo, Cc), dtype=torch.uint8, device=src_hwc.device)
    _req(out.dtype == torch.uint8 and out.shape == (Ho, Wo, Cc) and out.is_contiguous(), "out uint8 [Ho,Wo,C]")
    lib = _lib.load()
    check(lib.cut3r_remap_linear_u8(_p(src_hwc), H, W, Cc, _p(map_ix), _p(map_iy), _p(out), Ho, Wo, _stream()), "cut3r_remap_linear_u8")
    return out


def resize_linear_u8(img_hwc, H1, W1, chw_out=True, out=None):
    """cv2.resize(img, (W1, H1)) (INTER_LINEAR, u8) on the GPU.  img_hwc u8 [H0,W0,C] -> u8 [C,H1,W1] (or [H1,W1,C]); `out`: write
    there (e.g. straight into the keyframe store) instead of a new tensor."""
    _cuda(img_hwc, out)
    _req(img_hwc.dtype == torch.uint8 and img_hwc.dim() == 3 and img_hwc.is_contiguous() and img_hwc.shape[2] <= 4, "u8 [H,W,C<=4] contiguous")
    H0, W0, Cc = img_hwc.shape
    shape = (Cc, int(H1), int(W1)) if chw_out else (int(H1), int(W1), Cc)
    if out is None:
        out = torch.empty(shape, dtype=torch.uint8, device=img_hwc.device)
    _req(out.dtype == torch.uint8 and tuple(out.shape) == shape and out.is_contiguous(), f"out u8 {shape} contiguous")
    lib = _lib.load()
    check(lib.cut3r_resize_linear_u8(_p(img_hwc), H0, W0, Cc, _p(out), int(H1), int(W1), int(bool(chw_out)), _stream()), "cut3r_resize_linear_u8")
    return out


# ------------------------------------------------------------------------------------------------ loop closure
def lc_optimize(submaps, mask, cur, cur_lc, iters, lr=5e-4, return_loss=False):
    """Fused Adam over per-submap se(3) corrections (track_backend.py:256-299).  submaps: [B,6,h,w,3] fp32 contiguous
    (slot 0 = first, slot 5 = last pointmap of each submap); mask: bool/uint8 [B-1,h*w] or None; cur, cur_lc: [h*w,3].
    Returns (xi [B,6], T [B,3,4]) (+ per-iteration loss)."""
    _cuda(submaps, mask, cur, cur_lc)
    _req(submaps.dtype == F32 and submaps.dim() == 5 and submaps.is_contiguous() and submaps.shape[1] == 6 and submaps.shape[4] == 3, "submaps [B,6,h,w,3]")
    B = submaps.shape[0]
    N = submaps.shape[2] * submaps.shape[3]
    _req(B >= 2, "need at least two submaps")
    cur = cur.reshape(-1, 3).contiguous()
    cur_lc = cur_lc.reshape(-1, 3).contiguous()
    _req(cur.shape == (N, 3) and cur_lc.shape == (N, 3) and cur.dtype == F32 and cur_lc.dtype == F32, "cur / cur_lc [N,3] fp32")
    dev = submaps.device
    if mask is not None:
        mask = mask.reshape(B - 1, N).to(torch.uint8).contiguous()
        n_masked = int(mask.sum().item())
    else:
        n_masked = (B - 1) * N
    lib = _lib.load()
    xi = torch.zeros(B, 6, device=dev)
    m, v = torch.zeros_like(xi), torch.zeros_like(xi)
    T = torch.eye(4, device=dev)[:3].reshape(1, 12).repeat(B, 1).contiguous()
    ws = torch.empty(lib.cut3r_lc_workspace_floats(B, N), device=dev)
    loss = torch.zeros(max(iters, 1), device=dev) if return_loss else None
    first = submaps
    last_ptr = C.c_void_p(submaps.data_ptr() + 5 * N * 3 * 4)
    check(lib.cut3r_lc_optimize(_p(first), last_ptr, 6 * N * 3, _p(mask), _p(cur), _p(cur_lc), B, N, n_masked, int(iters), float(lr),
                                _p(xi), _p(m), _p(v), _p(T), _p(ws), _p(loss), _stream()), "cut3r_lc_optimize")
    T = T.view(B, 3, 4)
    return (xi, T, loss) if return_loss else (xi, T)


def lc_optimize_terms(terms, P, N, iters, lr=5e-4, return_loss=False):
    """Fused Adam over P-1 se(3) vectors for a list of L1 terms (track_backend.py:400-461).  terms: list of
    (a [N,3] fp32 cuda, ia, c [N,3] fp32 cuda, ic, weight, mask uint8 [N] | None) with transform indices in [0, P), 0 = the
    fixed identity.  Returns (xi [P,6], T [P,3,4]) (+ per-iteration loss)."""
    _req(len(terms) >= 1 and P >= 2, "at least one term and one free transform")
    keep = []
    arr = (_lib.LcTerm * len(terms))()
    dev = terms[0][0].device
    for t, (a, ia, c, ic, w, mask) in enumerate(terms):
        _cuda(a, c, mask)
        _req(a.dtype == F32 and c.dtype == F32 and a.is_contiguous() and c.is_contiguous() and a.numel() == 3 * N and c.numel() == 3 * N,
             "term operands: contiguous fp32 [N,3]")
        _req(0 <= ia < P and 0 <= ic < P, "transform index out of range")
        if mask is not None:
            _req(mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == N, "mask uint8 [N]")
        arr[t].a, arr[t].c, arr[t].mask = a.data_ptr(), c.data_ptr(), (mask.data_ptr() if mask is not None else None)
        arr[t].ia, arr[t].ic, arr[t].w, arr[t].pad = int(ia), int(ic), float(w), 0
        keep.append((a, c, mask))
    raw = bytes(arr)
    terms_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    lib = _lib.load()
    xi = torch.zeros(P, 6, device=dev)
    m, v = torch.zeros_like(xi), torch.zeros_like(xi)
    T = torch.eye(4, device=dev)[:3].reshape(1, 12).repeat(P, 1).contiguous()
    ws = torch.empty(lib.cut3r_lc_workspace_floats(len(terms), N), device=dev)
    loss = torch.zeros(max(iters, 1), device=dev) if return_loss else None
    check(lib.cut3r_lc_optimize_terms(_p(terms_dev), len(terms), int(P), int(N), int(iters), float(lr), _p(xi), _p(m), _p(v), _p(T), _p(ws),
                                      _p(loss), _stream()), "cut3r_lc_optimize_terms")
    T = T.view(P, 3, 4)
    return (xi, T, loss) if return_loss else (xi, T)


def transform_submaps(submaps, T):
    """in place: every point of submap b <- T[b] (3x4) applied (track_backend.py:301-310)."""
    _cuda(submaps, T)
    B = submaps.shape[0]
    _req(submaps.dtype == F32 and submaps.is_contiguous() and T.dtype == F32 and T.is_contiguous() and T.numel() == 12 * B, "transform_submaps")
    per = submaps[0].numel() // 3
    lib = _lib.load()
    check(lib.cut3r_transform_submaps(_p(submaps), _p(T), B, per, _stream()), "cut3r_transform_submaps")
    return submaps
