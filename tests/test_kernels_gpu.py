"""GPU parity of every HIP kernel (called through the C ABI) against the oracle / a torch fp32 statement of the op.

Tolerances: operands are rounded to fp16 before the MFMA and accumulated in fp32, so against an fp32 evaluation on
the SAME fp16-rounded operands the error is accumulation-order only (<= 2e-3 relative to the output scale for the
fp16-stored outputs, which carry 2^-11 rounding).  Integer/byte results are exact.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cut3r_slam_amd import ops  # noqa: E402
from oracle import cut3r_oracle as O  # noqa: E402
from oracle import geom as G  # noqa: E402

DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _report(name, got, ref, tol):
    err = _rel(got, ref)
    if not err <= tol:
        d = (got.double().cpu() - ref.double().cpu()).abs()
        idx = np.unravel_index(int(d.argmax()), d.shape)
        nbad = int((d > tol * ref.double().abs().max()).sum())
        raise AssertionError(f"{name}: rel err {err:.3e} > {tol:.1e}; worst at {idx}: got {got.cpu()[idx].item():.6f} "
                             f"ref {ref.cpu()[idx].item():.6f}; {nbad}/{d.numel()} elements out of tolerance; "
                             f"nan={bool(torch.isnan(got).any())}")


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,tile", [
    (768, 1024, 1024, 0), (769, 768, 768, 0), (4608, 3072, 1024, 128), (4608, 1024, 4096, 0), (1, 1536, 1536, 0),
    (6, 64, 48, 0), (13, 144, 48, 64), (256, 4608, 1536, 0), (130, 8, 192, 0), (200, 132, 72, 128),
    # 256^2 ping-pong kernel: full tiles, ragged M / N / K (K tail inside a 64-deep tile), one K-tile, two K-tiles
    (3072, 1024, 1024, 256), (769, 768, 768, 256), (1000, 516, 200, 256), (300, 260, 64, 256), (257, 256, 128, 256),
    (5, 12, 8, 256),
    # 192 x 128 tile (two-band epilogue): full tiles, ragged M / N / K
    (6152, 768, 768, 192128), (385, 260, 200, 192128), (191, 128, 64, 192128), (193, 132, 72, 192128),
    (3076, 768, 768, 128192), (260, 388, 200, 128192), (129, 196, 72, 128192),
    # skinny M <= 64 kernel (pose memory / pose MLP rows): K split over 4 and 8 waves, 1/2/4 row blocks, ragged N and K
    (8, 1536, 1536, 16), (8, 1536, 6144, 16), (1, 4608, 1536, 16), (16, 6144, 1536, 16), (3, 20, 40, 16), (7, 3072, 768, 16), (12, 8, 3072, 16),
    (24, 1536, 6144, 16), (33, 1536, 1536, 16), (64, 768, 3072, 16), (50, 36, 72, 16), (150, 1536, 1536, 16)])
def test_gemm_bias_gelu_residual(M, N, K, tile):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = A.float() @ W.float().t() + bias
    for act, use_res, out_dt in ((0, False, torch.float32), (1, False, torch.float16), (0, True, torch.float32), (2, True, torch.float16)):
        r = ref
        if act == 1:
            r = F.gelu(r)
        if act == 2:
            r = F.relu(r)
        if use_res:
            r = r + res
        out = torch.full((M, N), float("nan"), dtype=out_dt, device=DEV)
        ops.linear(A.to(DEV), W.to(DEV), out, bias.to(DEV), act, res.to(DEV) if use_res else None, tile=tile)
        torch.cuda.synchronize()
        _report(f"gemm M{M} N{N} K{K} act{act} res{use_res} {out_dt}", out.float(), r, 2e-3 if out_dt == torch.float16 else 2e-5)


@pytest.mark.parametrize("M,N,K", [(4096, 2048, 1024), (3076, 768, 3072), (2500, 1024, 192)])
def test_gemm256_race_screen(M, N, K):
    """The 256^2 kernel orders its LDS-DMA ring only by counted vmcnt + barriers: repeated launches under load must
    be bit-identical to each other and match the 128^2 kernel (same MFMA, same K order -> same fp32 sums)."""
    g = torch.Generator().manual_seed(K)
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    ref = torch.empty(M, N, device=DEV)
    ops.linear(A, W, ref, None, 0, tile=128)
    outs = [torch.empty(M, N, device=DEV) for _ in range(4)]
    for rep in range(40):
        ops.linear(A, W, outs[rep % 4], None, 0, tile=256)
        if rep % 4 == 3:
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o, ref), f"rep {rep}: {(o != ref).sum().item()} elements differ, max {(o - ref).abs().max().item()}"


@pytest.mark.parametrize("M,N,cols,hd,tile", [
    (1538, 2304, 1536, 64, 64), (1538, 2304, 1536, 64, 128), (1538, 2304, 1536, 64, 256), (1538, 2304, 1536, 64, 192128),
    (769, 768, 768, 64, 128), (1536, 1536, 768, 64, 256), (300, 128, 64, 64, 64), (300, 192, 128, 64, 128192),
    # 48-wide heads (state side): only the 128 x 192 tile holds whole heads; auto (0) must pick it
    (1536, 2304, 1536, 48, 0), (768, 768, 768, 48, 128192), (1537, 1536, 768, 48, 0), (300, 192, 96, 48, 0), (70, 240, 240, 48, 0)])
def test_gemm_fused_rope_equals_gemm_then_rope_kernel(M, N, cols, hd, tile):
    """RoPE in the projection epilogue (head dimension 64 or 48) must reproduce the separate rope_2d kernel bit for bit:
    q|k|v, q-only and k|v layouts, positions including the pose token's -1, ragged last row / column tiles."""
    K = 192
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    pos = torch.randint(-1, 33, (M, 2), generator=g, dtype=torch.int64).to(DEV)
    ref = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.linear(A, W, ref, b, 0, tile=tile if tile else 128)
    heads = cols // hd
    ops.rope_2d(ref.view(1, M, N // hd, hd)[:, :, :heads], pos.view(1, M, 2), 100.0, 1.0)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
    ops.linear(A, W, out, b, 0, tile=tile, rope=(pos, cols, 100.0, hd))
    torch.cuda.synchronize()
    assert torch.equal(out, ref), f"{(out != ref).sum().item()} elements differ, max {(out.float() - ref.float()).abs().max().item()}"
    assert not torch.equal(out[:, :cols], (A.float() @ W.float().t() + b).half()[:, :cols])      # RoPE did something


@pytest.mark.parametrize("hi", [47, 48, 100, 256])
def test_gemm256_fused_rope_beyond_the_table_rows_it_keeps_in_lds(hi):
    """the 256^2 kernel keeps table rows 0..47 (positions -1..46) per wave in LDS and reads the others from global memory: positions on
    both sides of that line, mixed inside one row group, and beyond the table (clamped like the stand-alone kernel) give the same bits"""
    M, N, K, cols = 1100, 1536, 192, 1024
    g = torch.Generator().manual_seed(hi)
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    pos = torch.randint(-1, hi + 1, (M, 2), generator=g, dtype=torch.int64).to(DEV)
    pos[512:768] = torch.randint(-1, 40, (256, 2), generator=g, dtype=torch.int64).to(DEV)       # a whole tile of rows inside the LDS rows
    ref = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.linear(A, W, ref, b, 0, tile=128)
    ops.rope_2d(ref.view(1, M, N // 64, 64)[:, :, :cols // 64], pos.view(1, M, 2), 100.0, 1.0)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
    ops.linear(A, W, out, b, 0, tile=256, rope=(pos, cols, 100.0, 64))
    torch.cuda.synchronize()
    assert torch.equal(out, ref), f"{(out != ref).sum().item()} elements differ, max {(out.float() - ref.float()).abs().max().item()}"


def test_skinny_gemm_rows_do_not_depend_on_the_batch():
    """a row of the M <= 64 kernel must come out bit-identical whatever M is (window-batch invariance of the pose path)"""
    g = torch.Generator().manual_seed(4)
    K, N = 6144, 1536
    A = torch.randn(40, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    A = torch.cat([A, A[:37] * 0.5], 0)                  # 77 rows: more than one 64-row launch
    full = torch.empty(77, N, device=DEV)
    ops.linear(A, W, full, b, 0, tile=16)
    for m in (1, 8, 16, 17, 32, 65):
        part = torch.empty(m, N, device=DEV)
        ops.linear(A[:m].contiguous(), W, part, b, 0, tile=16)
        torch.cuda.synchronize()
        assert torch.equal(part, full[:m]), m


@pytest.mark.parametrize("M,N,K", [(1536, 1024, 1024), (2000, 768, 768), (777, 512, 320)])
def test_every_tile_kernel_gives_the_same_bits(M, N, K):
    """A row's result must not depend on the tile kernel that computed it (the encoder is batch invariant bit for bit and
    windows batched through the decoder equal windows decoded alone only because of this): same MFMA, same K order, and the
    same epilogue arithmetic (gemm.hip is compiled with -ffp-contract=off: the compiler contracted the GELU differently in the
    256^2 kernel once) -- for every epilogue the model uses."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r32 = torch.randn(M, N, generator=g).to(DEV)
    cases = ((0, None, torch.float16), (1, None, torch.float16), (0, r32, torch.float32), (0, r32.half(), torch.float16), (2, None, torch.float32),
             (1, None, torch.float32), (0, None, torch.float32))
    for act, res, odt in cases:
        ref = torch.zeros(M, N, dtype=odt, device=DEV)
        ops.linear(A, W, ref, b, act, res, tile=128)
        for tile in (64, 256, 192128, 128192, 256128, 12864):
            out = torch.zeros(M, N, dtype=odt, device=DEV)
            ops.linear(A, W, out, b, act, res, tile=tile)
            torch.cuda.synchronize()
            assert torch.equal(out, ref), (tile, act, odt, int((out != ref).sum()))


def test_gemm_strided_output_and_inplace_residual():
    g = torch.Generator().manual_seed(1)
    M, N, K = 70, 96, 64
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = torch.randn(N, K, generator=g).half().to(DEV)
    big = torch.zeros(M + 1, 2 * N, device=DEV)
    x = torch.randn(M, N, generator=g).to(DEV)
    big[1:, :N] = x
    ops.linear(A, W, big[1:, :N], None, 0, res1=big[1:, :N])            # in place: out == res1, row stride 2N, row offset 1
    torch.cuda.synchronize()
    _report("gemm strided/in-place", big[1:, :N], x + A.float() @ W.float().t(), 2e-5)
    assert float(big[0].abs().max()) == 0 and float(big[:, N:].abs().max()) == 0


@pytest.mark.parametrize("tile", [0, 256])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,relu_in", [
    (1, 12, 16, 256, 256, 1, True), (2, 5, 7, 96, 256, 1, False), (1, 24, 32, 768, 768, 2, False), (3, 2, 3, 256, 128, 1, False),
    (1, 1, 2, 256, 256, 1, True), (2, 9, 9, 128, 128, 2, False), (2, 24, 32, 256, 256, 1, True)])
def test_conv3x3_implicit_gemm(B, H, W, Cin, Cout, stride, relu_in, tile):
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(B, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).half()
    b = torch.randn(Cout, generator=g)
    xin = F.relu(x.float()) if relu_in else x.float()
    ref = F.conv2d(xin, w.float(), b, stride=stride, padding=1)
    r1 = torch.randn_like(ref).half()
    ref = (ref + r1.float()).permute(0, 2, 3, 1)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    out = torch.full(ref.shape, float("nan"), dtype=torch.float16, device=DEV)
    ops.conv3x3_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), wk.to(DEV), out, b.to(DEV), stride, relu_in, 0,
                     res1=r1.permute(0, 2, 3, 1).contiguous().to(DEV), tile=tile)
    torch.cuda.synchronize()
    _report(f"conv3x3 {B}x{H}x{W} {Cin}->{Cout} s{stride}", out.float(), ref, 2e-3)


@pytest.mark.parametrize("s,Cin,Cout,H,W", [(4, 96, 96, 2, 3), (2, 192, 192, 6, 8), (4, 96, 96, 24, 32)])
def test_conv_transpose_pixel_shuffle(s, Cin, Cout, H, W):
    g = torch.Generator().manual_seed(s)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g).half()
    w = (torch.randn(Cin, Cout, s, s, generator=g) / Cin ** 0.5).half()
    b = torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x.float(), w.float(), b, stride=s).permute(0, 2, 3, 1)
    wt = w.permute(2, 3, 1, 0).reshape(s * s * Cout, Cin).contiguous()
    out = torch.full(ref.shape, float("nan"), dtype=torch.float16, device=DEV)
    ops.conv_transpose_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), wt.to(DEV), out, b.to(DEV), s)
    torch.cuda.synchronize()
    _report(f"convT s{s}", out.float(), ref, 2e-3)


def test_gemv_silu():
    g = torch.Generator().manual_seed(3)
    X = torch.randn(6, 768, generator=g)
    W = (torch.randn(1536, 768, generator=g) / 768 ** 0.5).half()
    b = torch.randn(1536, generator=g)
    out = torch.empty(6, 1536, device=DEV)
    ops.gemv(X.to(DEV), W.to(DEV), out, b.to(DEV), silu_in=True)
    torch.cuda.synchronize()
    ref = F.silu(X).half().float() @ W.float().t() + b
    _report("gemv", out, ref, 1e-4)


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,H,Nq,Nk,D", [
    (1, 16, 768, 768, 64), (2, 16, 768, 768, 64), (1, 12, 769, 769, 64), (1, 16, 768, 769, 48), (1, 12, 769, 768, 64),
    (1, 12, 256, 256, 128), (1, 12, 256, 1, 128), (1, 12, 1, 256, 128), (1, 12, 1, 1, 128), (3, 4, 6, 6, 16),
    (1, 3, 7, 12, 16), (1, 3, 8, 1, 32), (6, 16, 768, 768, 64), (1, 2, 100, 333, 32)])
def test_attention_vs_sdpa(B, H, Nq, Nk, D):
    g = torch.Generator().manual_seed(Nq * 3 + Nk + D)
    q = (torch.randn(B, Nq, H, D, generator=g) * 1.5).half()
    k = (torch.randn(B, Nk, H, D, generator=g) * 1.5).half()
    v = torch.randn(B, Nk, H, D, generator=g).half()
    k[0, 0, 0] *= 6.0                                # a spiked key: exercises the running-max rescale
    if Nk > 70:
        k[0, 69, -1] *= 8.0                          # ... in a later KV tile as well
    scale = D ** -0.5
    ref = F.scaled_dot_product_attention(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3),
                                         v.float().permute(0, 2, 1, 3), scale=scale).permute(0, 2, 1, 3)
    # fused-buffer strides, like the model: q/k/v are slices of one [B,N,3,H,D] buffer when Nq == Nk
    out = torch.full((B, Nq, H, D), float("nan"), dtype=torch.float16, device=DEV)
    if Nq == Nk:
        qkv = torch.stack([q, k, v], dim=2).contiguous().to(DEV)
        ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out, scale)
    else:
        ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), out, scale)
    torch.cuda.synchronize()
    _report(f"attention B{B} H{H} Nq{Nq} Nk{Nk} D{D}", out.float(), ref, 3e-3)


@pytest.mark.parametrize("B,H,Nq,Nk,D", [
    (2, 16, 768, 768, 64), (2, 12, 769, 769, 64), (2, 12, 769, 768, 64), (2, 16, 768, 768, 48), (2, 16, 768, 769, 48),
    (3, 7, 769, 769, 48), (9, 5, 300, 129, 64), (4, 8, 200, 100, 64), (4, 8, 130, 65, 48), (5, 8, 33, 640, 64), (4, 8, 256, 64, 64),
    (4, 8, 257, 63, 48), (4, 8, 300, 193, 48), (1, 3, 7, 12, 48), (1, 3, 1, 1, 64), (28, 12, 769, 769, 64)])
def test_pipelined_attention_kernel_gives_the_bits_of_the_staged_one(B, H, Nq, Nk, D):
    """attn_pipe_kernel (LDS-DMA ring, one barrier per tile, softmax of tile j beside the MFMAs of tiles j-1 and j+1; the default for
    48- and 64-wide heads) against attn_kernel (register-staged, two barriers per tile): the same arithmetic per row in the same order,
    so the SAME BITS -- for whole and ragged key tiles, the 64 m + 1 key counts of the decoder (key 0 folded in first), query counts that
    leave idle waves, (batch x head) counts that do not fill the 8-wide XCD groups, and a spiked key that moves the running maximum late.
    Either kernel is also inside the fp32-SDPA tolerance of test_attention_vs_sdpa."""
    from cut3r_slam_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 7 + Nq + Nk + D)
    q = (torch.randn(B, Nq, H, D, generator=g) * 1.5).half().to(DEV)
    k = (torch.randn(B, Nk, H, D, generator=g) * 1.5).half()
    v = torch.randn(B, Nk, H, D, generator=g).half().to(DEV)
    k[:, min(3, Nk - 1)] *= 6.0
    if Nk > 70:
        k[0, 69, -1] *= 8.0
    k = k.to(DEV)
    outs = []
    prev = lib.cut3r_attention_variant(-1)
    try:
        for variant in (0, 1):
            lib.cut3r_attention_variant(variant)
            o = torch.full((B, Nq, H, D), float("nan"), dtype=torch.float16, device=DEV)
            ops.attention(q, k, v, o, D ** -0.5)
            torch.cuda.synchronize()
            outs.append(o.cpu())
    finally:
        lib.cut3r_attention_variant(prev)
    assert prev == 1, "the pipelined kernel is the default"
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), f"{int((outs[0] != outs[1]).sum())} elements differ between the two attention kernels"
    ref = F.scaled_dot_product_attention(q[:1].float().permute(0, 2, 1, 3), k[:1].float().permute(0, 2, 1, 3),
                                         v[:1].float().permute(0, 2, 1, 3), scale=D ** -0.5).permute(0, 2, 1, 3)
    _report(f"pipelined attention B{B} H{H} Nq{Nq} Nk{Nk} D{D}", outs[1][:1].float(), ref.cpu(), 3e-3)


# ------------------------------------------------------------------------------------------------ RoPE / LN / helpers
def test_rope_matches_reference_golden_and_oracle():
    f = np.load(os.path.join(GOLD, "rope2d.npz"))
    for D in (16, 48, 64):
        tok, pos = torch.from_numpy(f[f"D{D}_tok"]), torch.from_numpy(f[f"D{D}_pos"])
        for F0 in (1, -1):
            t = tok.clone().to(DEV)
            ops.rope_2d(t, pos.to(DEV), 100.0, float(F0))
            torch.cuda.synchronize()
            np.testing.assert_allclose(t.cpu().numpy(), f[f"D{D}_F{F0}_out"], rtol=0, atol=1e-5)   # reference CPU op
            # device sinf/cosf/powf vs libm: a few ulp on angles up to 40 rad
            np.testing.assert_allclose(t.cpu().numpy(), G.rope2d(tok.numpy(), pos.numpy(), 100.0, float(F0)), rtol=0, atol=6e-6)


def test_rope_strided_half_view_like_the_model():
    g = torch.Generator().manual_seed(0)
    B, N, H, D = 2, 769, 12, 64
    qkv = torch.randn(B, N, 3, H, D, generator=g).half()
    pos = torch.randint(-1, 32, (B, N, 2), generator=g)
    dev = qkv.to(DEV)
    ops.rope_2d(dev[:, :, 0], pos.to(DEV), 100.0, 1.0)
    ops.rope_2d(dev[:, :, 1], pos.to(DEV), 100.0, 1.0)
    torch.cuda.synchronize()
    for j in range(2):
        ref = O.rope2d(qkv[:, :, j].permute(0, 2, 1, 3), pos).permute(0, 2, 1, 3)
        _report(f"rope half slice {j}", dev[:, :, j].float().cpu(), ref.float(), 1.5e-3)
    assert torch.equal(dev[:, :, 2].cpu(), qkv[:, :, 2])                  # v untouched
    with pytest.raises(RuntimeError):
        ops.rope_2d(dev[:, :, 0].transpose(1, 2), pos.to(DEV), 100.0, 1.0)    # reference's stride check


@pytest.mark.parametrize("H,D", [(12, 64), (16, 48), (3, 32), (5, 16), (2, 128)])
def test_rope_table_launch_gives_the_bits_of_the_reference_shaped_kernel(H, D):
    """the network's RoPE launches (ops.rope_2d_pair: table-driven, two token ranges per launch) against ops.rope_2d (the curope
    drop-in, one workgroup per token): identical bits, on the strided q / k views of a qkv buffer, for a cross-attention pair with
    different token counts, for the pose token's position -1 and for positions beyond the table (evaluated in place)."""
    g = torch.Generator().manual_seed(H * 100 + D)
    B, N, N2 = 3, 257, 64
    qkv = torch.randn(B, N, 3, H, D, generator=g).half().to(DEV)
    pos = torch.randint(-1, 40, (B, N, 2), generator=g)
    pos[0, 0] = -1
    pos[1, 5, 0], pos[1, 6, 1], pos[2, 7, 0] = 300, 100000, -7             # outside the table [-1, 256]
    pos = pos.to(DEV)
    ref = qkv.clone()
    ops.rope_2d(ref[:, :, 0], pos, 100.0, 1.0)
    ops.rope_2d(ref[:, :, 1], pos, 100.0, 1.0)
    ops.rope_2d_pair(qkv[:, :, 0], pos, qkv[:, :, 1], pos, 100.0, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(qkv, ref)
    assert not torch.equal(qkv[:, :, 0], qkv[:, :, 2])
    q = torch.randn(B, N, H, D, generator=g).half().to(DEV)
    kv = torch.randn(B, N2, 2, H, D, generator=g).half().to(DEV)
    pos2 = torch.randint(0, 16, (B, N2, 2), generator=g).to(DEV)
    rq, rkv = q.clone(), kv.clone()
    ops.rope_2d(rq, pos, 100.0, 1.0)
    ops.rope_2d(rkv[:, :, 0], pos2, 100.0, 1.0)
    ops.rope_2d_pair(q, pos, kv[:, :, 0], pos2, 100.0, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(q, rq) and torch.equal(kv, rkv)
    one = rq.clone()
    ops.rope_2d(rq, pos, 100.0, 1.0)
    ops.rope_2d_pair(one, pos, None, None, 100.0, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(one, rq)
    with pytest.raises(RuntimeError):
        ops.rope_2d_pair(q.transpose(1, 2), pos, None, None, 100.0, 1.0)


@pytest.mark.parametrize("M,C", [(768, 1024), (769, 768), (5, 48), (256, 1536), (1, 1536)])
def test_layernorm_and_adaln(M, C):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(M, C, generator=g) * 3 + 0.5
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    sc, sh = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g)
    ref = F.layer_norm(x, (C,), w, b, 1e-6)
    o16 = torch.empty(M, C, dtype=torch.float16, device=DEV)
    o32 = torch.empty(M, C, device=DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, o16, o32)
    torch.cuda.synchronize()
    _report("ln fp32", o32, ref, 5e-6)
    _report("ln fp16", o16.float(), ref, 1e-3)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, None, o32, sc.to(DEV), sh.to(DEV))
    torch.cuda.synchronize()
    _report("adaLN", o32, ref * (1 + sc) + sh, 5e-6)


@pytest.mark.parametrize("M,C", [(769, 768), (130, 1024), (5, 1536)])
def test_layernorm_dual_equals_two_layernorms(M, C):
    g = torch.Generator().manual_seed(C + M)
    x = (torch.randn(M, C, generator=g) * 3 + 0.5).to(DEV)
    prm = [torch.randn(C, generator=g).to(DEV) for _ in range(4)]
    ref1, ref2 = torch.empty(M, C, dtype=torch.float16, device=DEV), torch.empty(M, C, dtype=torch.float16, device=DEV)
    ops.layernorm(x, prm[0], prm[1], 1e-6, ref1, None)
    ops.layernorm(x, prm[2], prm[3], 1e-6, ref2, None)
    o1, o2 = torch.zeros_like(ref1), torch.zeros_like(ref2)
    ops.layernorm_dual(x, prm[0], prm[1], o1, prm[2], prm[3], o2, 1e-6)
    torch.cuda.synchronize()
    assert torch.equal(o1, ref1) and torch.equal(o2, ref2)


def test_im2col_cast_colmean_upsample():
    g = torch.Generator().manual_seed(5)
    img = torch.randn(2, 3, 32, 48, generator=g)
    out = torch.empty(2 * 2 * 3, 768, dtype=torch.float16, device=DEV)
    ops.im2col_patch(img.to(DEV), 16, out)
    ref = F.unfold(img, 16, stride=16).transpose(1, 2).reshape(-1, 768)
    torch.cuda.synchronize()
    _report("im2col", out.float(), ref.half().float(), 0)
    u8 = torch.randint(0, 256, (1, 3, 16, 32), generator=g, dtype=torch.uint8)
    out2 = torch.empty(2, 768, dtype=torch.float16, device=DEV)
    ops.im2col_patch(u8.to(DEV), 16, out2)
    ref2 = F.unfold((u8.float() / 255.0 - 0.5) / 0.5, 16, stride=16).transpose(1, 2).reshape(-1, 768)
    torch.cuda.synchronize()
    _report("im2col u8", out2.float(), ref2.half().float(), 1e-3)
    x = torch.randn(769, 768, generator=g)
    y = torch.empty(768, 768, dtype=torch.float16, device=DEV)
    ops.cast_f16(x.to(DEV)[1:], y)
    m = torch.empty(768, device=DEV)
    ops.colmean(x.to(DEV), m)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x[1:].half())
    _report("colmean", m, x.mean(0), 1e-5)
    t = torch.randn(2, 16, 5, 7, generator=g).half()
    up = torch.empty(2, 10, 14, 16, dtype=torch.float16, device=DEV)
    ops.upsample2x(t.permute(0, 2, 3, 1).contiguous().to(DEV), up)
    torch.cuda.synchronize()
    refu = F.interpolate(t.float(), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    _report("upsample2x", up.float(), refu, 1.5e-3)


@pytest.mark.parametrize("P,Cin", [(1000, 128), (4099, 256), (777, 48), (130, 64)])
def test_output_activations(P, Cin):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(P, Cin, generator=g).half()
    w = torch.randn(4, Cin, generator=g) * 0.05
    b = torch.randn(4, generator=g) * 0.1
    raw = x.float() @ w.t() + b
    pts, conf = torch.empty(P, 3, device=DEV), torch.empty(P, device=DEV)
    ops.dpt_final(x.to(DEV), w.to(DEV), b.to(DEV), 0, pts, conf)
    torch.cuda.synchronize()
    _report("dpt_final pts", pts, O.reg_dense_depth_exp(raw[:, :3]), 1e-5)
    _report("dpt_final conf", conf, 1 + raw[:, 3].exp(), 1e-5)
    rgb = torch.empty(P, 3, device=DEV)
    ops.dpt_final(x.to(DEV), w[:3].contiguous().to(DEV), b[:3].contiguous().to(DEV), 1, rgb, None)
    torch.cuda.synchronize()
    _report("dpt_final rgb", rgb, (torch.sigmoid(raw[:, :3]) * (1 - 2e-6) + 1e-6 - 0.5) * 2, 1e-5)
    ops.postprocess_pts(raw.contiguous().to(DEV), True, pts, conf)
    torch.cuda.synchronize()
    _report("postprocess pos_z", pts, O.reg_dense_depth_exp(raw[:, :3].clone(), pos_z=True), 1e-5)
    pr = torch.randn(5, 7, generator=g)
    po = torch.empty(5, 7, device=DEV)
    ops.postprocess_pose(pr.to(DEV), po)
    torch.cuda.synchronize()
    _report("postprocess_pose", po, O.postprocess_pose(pr), 1e-6)


# ------------------------------------------------------------------------------------------------ geometry (bit-exact)
def _scene(n, H, W, seed):
    g = np.random.default_rng(seed)
    from scipy.spatial.transform import Rotation
    c2w = np.tile(np.eye(4), (n, 1, 1))
    for i in range(n):
        c2w[i, :3, :3] = Rotation.from_euler("yxz", g.normal(0, 0.2, 3)).as_matrix()
        c2w[i, :3, 3] = g.normal(0, 0.7, 3)
    fx = fy = 0.9 * W
    K4 = [fx, fy, W / 2 - 0.5, H / 2 - 0.5]
    depth = g.uniform(2.0, 3.0, size=(n, H, W))
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cam = np.stack([(xs - K4[2]) / fx * depth, (ys - K4[3]) / fy * depth, depth, np.ones_like(depth)], -1)
    pm = np.einsum("nij,nhwj->nhwi", c2w, cam)[..., :3].astype(np.float32)
    pm[0, 0, 0] = [0, 0, -5]          # behind-camera / z <= 0 cases
    return c2w, pm, K4


@pytest.mark.parametrize("n,H,W", [(9, 24, 32), (50, 192, 256), (3, 5, 7)])
def test_overlap_counts_bit_exact(n, H, W):
    c2w, pm, K4 = _scene(n, H, W, n)
    w2c = G.w2c_rows(c2w)
    i = n - 1
    cnt = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    ops.overlap_fwd(torch.from_numpy(pm[i]).to(DEV), torch.from_numpy(w2c[:i]).to(DEV), K4, W, H, cnt)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(cnt[:i].cpu().numpy(), G.overlap_fwd(pm[i], w2c[:i], K4, W, H))
    if (H * W) % 4 == 0:
        cnt2 = torch.full((n,), -1, dtype=torch.int32, device=DEV)
        ops.overlap_bwd(torch.from_numpy(pm[:i]).to(DEV), torch.from_numpy(w2c[i]).to(DEV), K4, W, H, cnt2)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(cnt2[:i].cpu().numpy(), G.overlap_bwd(pm[:i], w2c[i], K4, W, H))


@pytest.mark.parametrize("W,H,K4", [(512, 384, (256.0, 211.8, 255.8, 191.6)), (64, 48, (60.0, 59.0, 31.5, 23.5)), (333, 77, (410.3, 95.7, 170.21, 33.33))])
def test_overlap_counts_on_the_image_border_bit_exact(W, H, K4):
    """the division-free form of the projection test (geometry.hip proj_valid) must take the reference's literal test for points
    within rounding of a bound: points are placed ON the four borders (u = -0.5, W - 0.5, v = -0.5, H - 0.5: the round-half-even
    ties), a few ulp either side and up to 1e-2 px away, under cameras of depth 0.05 .. 50 and the z <= 1e-5 clamp; counts and
    per-point decisions must equal oracle_geom.c"""
    g = np.random.default_rng(W)
    fx, fy, cx, cy = K4
    n = 4096
    zc = np.exp(g.uniform(np.log(0.05), np.log(50.0), n)).astype(np.float32)
    zc[:64] = g.uniform(-1e-5, 2e-5, 64).astype(np.float32)            # around the clamp
    side = g.integers(0, 4, n)
    off = np.where(g.random(n) < 0.5, 0.0, g.normal(0, 1, n) * np.exp(g.uniform(np.log(1e-7), np.log(1e-2), n)))
    u = np.where(side == 0, -0.5, np.where(side == 1, W - 0.5, g.uniform(-3, W + 3, n))) + np.where(side < 2, off, 0)
    v = np.where(side == 2, -0.5, np.where(side == 3, H - 0.5, g.uniform(-3, H + 3, n))) + np.where(side >= 2, off, 0)
    zd = np.maximum(zc.astype(np.float64), 1e-5)
    pc = np.stack([(u - cx) * zd / fx, (v - cy) * zd / fy, zc.astype(np.float64)], -1)      # camera-space points
    for k in range(1, 9):                                              # nudge the in-plane coordinates by whole ulps
        sl = slice(k * 256, (k + 1) * 256)
        pc32 = pc[sl].astype(np.float32)
        pc[sl, 0] = np.nextafter(pc32[:, 0], np.float32(np.inf if k % 2 else -np.inf)).astype(np.float64) if k < 5 else pc32[:, 0]
        pc[sl, 1] = np.nextafter(pc32[:, 1], np.float32(np.inf if k % 2 else -np.inf)).astype(np.float64) if k >= 5 else pc32[:, 1]
    pm = pc.astype(np.float32)
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)
    cams, sets = [ident], [pm]
    for _ in range(5):                 # general cameras, each with ITS OWN border set (the fp32 world->camera chain scatters it
        c2w, _, _ = _scene(2, 8, 8, int(g.integers(1 << 30)))          # a few 1e-5 px around the bounds: the critical zone)
        cams.append(G.w2c_rows(c2w)[1])
        sets.append((pc @ c2w[1, :3, :3].T + c2w[1, :3, 3]).astype(np.float32))
    w2c = np.stack(cams).astype(np.float32)
    pm_all = np.concatenate(sets)
    for clamp in (True, False):
        cnt = torch.full((len(w2c),), -1, dtype=torch.int32, device=DEV)
        ops.overlap_fwd(torch.from_numpy(pm_all).to(DEV), torch.from_numpy(w2c).to(DEV), K4, W, H, cnt, clamp_z=clamp)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(cnt.cpu().numpy(), G.overlap_fwd(pm_all, w2c, K4, W, H, clamp_z=clamp))
    for clamp in (True, False):
        cnt = torch.full((len(w2c),), -1, dtype=torch.int32, device=DEV)
        ops.overlap_fwd(torch.from_numpy(pm).to(DEV), torch.from_numpy(w2c).to(DEV), K4, W, H, cnt, clamp_z=clamp)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(cnt.cpu().numpy(), G.overlap_fwd(pm, w2c, K4, W, H, clamp_z=clamp))
    # per-point decisions under the identity camera (one point per launch row would be slow: 64-point slices instead)
    ref = np.array([G.overlap_fwd(pm[i:i + 1], w2c[:1], K4, W, H)[0] for i in range(n)])
    assert 0.1 < ref.mean() < 0.9                                      # the borders split the set
    got = torch.zeros(n // 64, dtype=torch.int32, device=DEV)
    pmd = torch.from_numpy(pm).to(DEV)
    one = torch.from_numpy(w2c[:1]).to(DEV)
    for i in range(n // 64):
        ops.overlap_fwd(pmd[i * 64:(i + 1) * 64].contiguous(), one, K4, W, H, got[i:i + 1])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), ref.reshape(-1, 64).sum(1))
    if n % 4 == 0:                                                     # backward form (no clamp), all points as one pointmap
        cb = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.overlap_bwd(pmd.view(1, n, 3), one[0].contiguous(), K4, W, H, cb, B=1, N=n)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(cb.cpu().numpy(), G.overlap_bwd(pm[None], w2c[0], K4, W, H))


def test_overlap_matches_reference_golden_decisions():
    f = np.load(os.path.join(GOLD, "graph.npz"))
    pm, c2w, K = f["pointmaps"], f["c2w"], f["K"]
    n, H, W, _ = pm.shape
    K4 = [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]
    i = n - 1
    w2c = G.w2c_rows(c2w)
    cnt = torch.zeros(n, dtype=torch.int32, device=DEV)
    ops.overlap_fwd(torch.from_numpy(pm[i]).to(DEV), torch.from_numpy(w2c[:i]).to(DEV), K4, W, H, cnt)
    cb = torch.zeros(n, dtype=torch.int32, device=DEV)
    ops.overlap_bwd(torch.from_numpy(pm[:i]).to(DEV), torch.from_numpy(w2c[i]).to(DEV), K4, W, H, cb)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(cnt[:i].cpu().numpy() / (H * W) > 0.3, f["ovl_batch_last"] > 0.3)
    np.testing.assert_array_equal(cb[:i].cpu().numpy() / (H * W) > 0.3, f["ovl_bi_last"].reshape(-1) > 0.3)


def test_align_view_bit_exact_and_logdepth():
    g = np.random.default_rng(2)
    H, W = 48, 64
    pts = g.normal(0, 1, (H, W, 3)).astype(np.float32)
    pts[..., 2] = np.abs(pts[..., 2]) + 0.5
    conf = (1 + np.exp(g.normal(0, 1, (H, W)))).astype(np.float32)
    P = g.normal(0, 1, 12).astype(np.float32)
    s = 1.37
    pm_ds = torch.empty(H // 2, W // 2, 3, device=DEV)
    cf_ds = torch.empty(H // 2, W // 2, device=DEV)
    dp = torch.empty(H, W, device=DEV)
    ops.align_view(torch.from_numpy(pts).to(DEV), torch.from_numpy(conf).to(DEV), P, s, 2, pm_ds, cf_ds, dp)
    torch.cuda.synchronize()
    rpm, rcf, rdp = G.align_view(pts, conf, P, s, 2)
    np.testing.assert_array_equal(pm_ds.cpu().numpy(), rpm)
    np.testing.assert_array_equal(cf_ds.cpu().numpy(), rcf)
    np.testing.assert_array_equal(dp.cpu().numpy(), rdp)
    prev = (np.abs(g.normal(2, 0.3, (H, W))) + 0.1).astype(np.float32)
    out = torch.zeros(1, dtype=torch.float64, device=DEV)
    ops.logdepth_sum(torch.from_numpy(prev).to(DEV), torch.from_numpy(pts).to(DEV), out)
    torch.cuda.synchronize()
    ref = G.logdepth_sum(prev, pts)
    assert abs(out.item() - ref) <= 2e-6 * H * W          # device logf vs libm logf: <= 2 ulp per term


def test_patch_overlap_count_matches_oracle_and_reference():
    f = np.load(os.path.join(GOLD, "graph.npz"))
    f0 = torch.from_numpy(f["feat0"]).to(DEV)
    N, C = f0.shape
    ws = torch.empty(2 * (N - 1) * C + N, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    for k, ref in enumerate(f["patch_ratios"]):
        ops.patch_overlap_count(f0, torch.from_numpy(f[f"feat1_{k}"]).to(DEV), 0.7, ws, cnt)
        torch.cuda.synchronize()
        assert abs(cnt.item() / (N - 1) - ref) < 1e-6, (k, cnt.item(), ref)
    g = torch.Generator().manual_seed(4)
    a = torch.randn(768, 1024, generator=g)
    b = a[torch.randperm(768, generator=g)] + 0.9 * torch.randn(768, 1024, generator=g)
    ws = torch.empty(2 * 767 * 1024 + 768, device=DEV)
    ops.patch_overlap_count(a.to(DEV), b.to(DEV), 0.7, ws, cnt)
    torch.cuda.synchronize()
    ratio, mx = G.patch_overlap_ratio(a.numpy(), b.numpy())
    border = int((np.abs(mx - 0.7) < 1e-5).sum())
    assert abs(cnt.item() - round(ratio * 767)) <= border


def test_window_update_matches_per_keyframe_calls():
    """cut3r_window_update (4 launches per window) == the per-keyframe entry points, bit for bit: stored pointmaps,
    confidences, depths and every forward / backward overlap count."""
    g = torch.Generator().manual_seed(11)
    V, H, W, ds, t0 = 6, 48, 64, 2, 10
    h, w = H // ds, W // ds
    nsub = 4
    store = (torch.randn(nsub, 6, h, w, 3, generator=g) * 0.5 + torch.tensor([0, 0, 2.0])).to(DEV)
    pts = (torch.randn(V, H, W, 3, generator=g) * 0.4 + torch.tensor([0, 0, 1.5])).to(DEV)
    conf = (1.0 + torch.rand(V, H, W, generator=g) * 4).to(DEV)
    w2c = torch.eye(4)[:3].reshape(1, 12).repeat(t0 + V, 1)
    w2c[:, 3] = torch.randn(t0 + V, generator=g) * 0.3
    w2c[:, 7] = torch.randn(t0 + V, generator=g) * 0.3
    w2c = w2c.contiguous().to(DEV)
    P = torch.eye(4)[:3].reshape(1, 12).repeat(V, 1)
    P[:, 3] = torch.randn(V, generator=g) * 0.2
    P[:, 11] = torch.randn(V, generator=g) * 0.2
    s, K4 = 1.37, [40.0, 42.0, 31.5, 23.5]
    sub = t0 // 5
    # per-keyframe path
    ref_store = store.clone()
    ref_conf = torch.zeros(nsub, 6, h, w, device=DEV)
    ref_depth = torch.zeros(V, H, W, device=DEV)
    ref_f, ref_b = [], []
    for v in range(V):
        ops.align_view(pts[v], conf[v], P[v].tolist(), s, ds, ref_store[sub, v], ref_conf[sub, v], ref_depth[v])
        i = t0 + v
        cf = torch.zeros(i, dtype=torch.int32, device=DEV)
        cb = torch.zeros(i, dtype=torch.int32, device=DEV)
        ops.overlap_fwd(pts[v], w2c[:i].contiguous(), K4, W, H, cf, P[v].tolist(), s)
        ops.overlap_bwd(ref_store, w2c[i].contiguous(), K4, w, h, cb, B=i, N=w * h, grp=5, grp_stride=6)
        ref_f.append(cf.cpu())
        ref_b.append(cb.cpu())
    # fused path
    got_store = store.clone()
    got_conf = torch.zeros(nsub, 6, h, w, device=DEV)
    got_depth = torch.zeros(V, H, W, device=DEV)
    counts = torch.full((V, 2, 64), -1, dtype=torch.int32, device=DEV)
    w2c_dev = w2c.clone()
    w2c_dev[t0:] = 7.0                               # rows of the window's keyframes arrive through the kernel arguments
    lsum = torch.full((1,), 3.0, dtype=torch.float64, device=DEV)
    ops.window_update(pts, conf, P.reshape(-1).tolist(), s, ds, got_store[sub, :V], got_conf[sub, :V], got_depth, got_store, w2c_dev, t0, 3,
                      K4, counts, w2c_new=w2c[t0:].reshape(-1).tolist(), lsum_reset=lsum)
    assert torch.equal(w2c_dev, w2c) and float(lsum) == 0.0
    torch.cuda.synchronize()
    assert torch.equal(got_store, ref_store) and torch.equal(got_conf, ref_conf) and torch.equal(got_depth, ref_depth)
    c = counts.cpu()
    for v in range(V):
        i = t0 + v
        assert torch.equal(c[v, 0, :i], ref_f[v]) and torch.equal(c[v, 1, :i], ref_b[v]), f"view {v}"
        assert int(c[v, 0, :i].sum()) > 0 and int(c[v, 1, :i].sum()) > 0          # the case is not vacuous


def test_mfma_probe_reports_a_plausible_clock_and_rate():
    """bench.py's measurement aid: the stamps give a shader clock between 0.8 and 2.6 GHz, the rate stays below the nominal 2.5 PFLOP/s, and
    the accumulators of a launch on random operands are finite (the loop really multiplied)"""
    import ctypes
    from cut3r_slam_amd import _lib
    lib = _lib.load()
    grid, iters = 256, 2000
    data = torch.randn(1 << 16, device=DEV).half()
    sink = torch.empty(grid * 512, device=DEV)
    stamps = torch.zeros(grid, 2, dtype=torch.int64, device=DEV)
    args = (ctypes.c_void_p(data.data_ptr()), data.numel(), iters, grid, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(stamps.data_ptr()),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert lib.cut3r_mfma_probe(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        assert lib.cut3r_mfma_probe(*args) == 0
    e1.record()
    torch.cuda.synchronize()
    s = stamps.cpu().double()
    assert bool((s > 0).all())
    clk = float((s[:, 0] / s[:, 1]).median()) * 0.1
    tf = grid * 8 * iters * 16 * 16384 / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12
    print(f"[mfma probe] in-kernel clock {clk:.2f} GHz, {tf:.0f} TFLOP/s on N(0,1) operands")
    assert 0.8 < clk < 2.6 and 300 < tf < 2500, (clk, tf)
    assert bool(torch.isfinite(sink).all()) and float(sink.abs().max()) > 0
    # argument checks
    assert lib.cut3r_mfma_probe(args[0], 1000, iters, grid, args[4], args[5], args[6]) != 0
