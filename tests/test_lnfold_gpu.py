"""GPU: LayerNorm folded into the GEMMs (cut3r_gemm_desc.ln_stats / ln_colsum / stats_out / out16; replaces the nn.LayerNorm launches in
front of qkv / projq / projk|projv / fc1: /root/reference/src/croco/models/blocks.py:187-190, src/dust3r/blocks.py:292-297).

  producer   the fp32 + fp32-residual GEMM also writes the fp16 copy of its output rows and, per row and 64-column slab, (sum, m2);
  consumer   A = that fp16 copy, B = fp16(gamma . W), bias = W beta + b, the epilogue applies acc * rstd - (rstd mu) c_n first.

Checked: the producer's by-products against torch (the copy bit for bit, the statistics to fp32 rounding); the folded consumer against an
fp64 LayerNorm + Linear and against the UNFUSED HIP path (LayerNorm kernel -> fp16 -> GEMM) -- it must be as close to fp64 as that path --
for every epilogue the network folds into (plain fp16, GELU, fused RoPE with 64- and 48-wide heads) on rows with a large mean and a
massive channel; every tile kernel gives the same bits (batch invariance of the network rests on it); argument errors are refused."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cut3r_slam_amd import ops  # noqa: E402

DEV = "cuda:0"
F16, F32 = torch.float16, torch.float32
EPS = 1e-6


def _rows(M, C, seed, kind="typical"):
    """residual-stream-like rows.  typical: unit noise, a per-row mean of up to ~1 sigma, one massive channel (|x| ~ 300) in a third of the
    rows.  adversarial: a per-row mean of ~3 sigma ON TOP of the massive channel -- the fold rounds x itself to fp16 (error relative to
    |x|), the LayerNorm kernel rounds (x - mu) / sigma (error relative to |x - mu|), so a mean that dwarfs the deviations costs the fold
    precision by about |mu| / |x - mu|."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, C, generator=g)
    x += (3.0 if kind == "adversarial" else 0.5) * torch.randn(M, 1, generator=g)
    x[::3, 7] += 300.0
    return x


def _produce(x, tile):
    """x through the PRODUCER: out = 0 @ W^T + 0 + x (an fp32 + residual GEMM whose product is zero) -> (out, x16, stats)"""
    M, C = x.shape
    A = torch.zeros(M, 64, dtype=F16, device=DEV)
    W = torch.zeros(C, 64, dtype=F16, device=DEV)
    out = torch.empty(M, C, dtype=F32, device=DEV)
    x16 = torch.empty(M, C, dtype=F16, device=DEV)
    st = torch.full((C // 64, M, 2), float("nan"), dtype=F32, device=DEV)
    ops.linear(A, W, out, torch.zeros(C, device=DEV), res1=x.to(DEV), tile=tile, emit=(st, x16))
    torch.cuda.synchronize()
    return out, x16, st


@pytest.mark.parametrize("M,C", [(769, 768), (300, 1024), (130, 192)])
def test_producer_writes_the_fp16_copy_and_the_slab_statistics(M, C):
    x = _rows(M, C, 1)
    ref = x.double().view(M, C // 64, 64)
    s_ref, m2_ref = ref.sum(-1), ((ref - ref.mean(-1, keepdim=True)) ** 2).sum(-1)
    first = None
    for tile in (64, 128, 256):
        out, x16, st = _produce(x, tile)
        assert torch.equal(out.cpu(), x) and torch.equal(x16.cpu(), x.half())
        st = st.cpu().double().permute(1, 0, 2)          # [C/64, M, 2] (slab-major) -> [M, C/64, 2]
        assert torch.isfinite(st).all()
        assert (st[..., 0] - s_ref).abs().max() <= 2e-6 * ref.abs().sum(-1).max()
        assert ((st[..., 1] - m2_ref).abs() / m2_ref).max() < 2e-5
        if first is None:
            first = st
        assert torch.equal(st, first), f"tile {tile}: slab statistics differ from tile 64 (fixed-order reduction broken)"


def _folded_setup(M, K, N, seed, kind="typical"):
    g = torch.Generator().manual_seed(100 + seed)
    x = _rows(M, K, seed, kind)
    gamma = 1.0 + 0.3 * torch.randn(K, generator=g)
    beta = 0.2 * torch.randn(K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = 0.1 * torch.randn(N, generator=g)
    Wf = (W.double() * gamma.double()[None]).float().half()
    d = (W.double() @ beta.double() + b.double()).float()
    c = Wf.double().sum(1).float()
    xd = x.double()
    ln = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + EPS) * gamma.double() + beta.double()
    y64 = ln @ W.double().T + b.double()
    return x, gamma, beta, W, b, Wf, d, c, y64


def _unfused(x, gamma, beta, W, b, act, rope, tile):
    M, K = x.shape
    ln16 = torch.empty(M, K, dtype=F16, device=DEV)
    ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), EPS, ln16, None)
    out = torch.empty(M, W.shape[0], dtype=F16, device=DEV)
    ops.linear(ln16, W.half().to(DEV), out, b.to(DEV), act, tile=tile, rope=rope)
    return out


def _fold_run(x16, st, Wf, d, c, act, rope, tile):
    out = torch.empty(x16.shape[0], Wf.shape[0], dtype=F16, device=DEV)
    ops.linear(x16, Wf.to(DEV), out, d.to(DEV), act, tile=tile, rope=rope, ln=(st, c.to(DEV), EPS))
    return out


@pytest.mark.parametrize("M,K,N", [(769, 768, 2304), (515, 1024, 1024), (260, 192, 384)])
@pytest.mark.parametrize("act,kind", [(0, "typical"), (1, "typical"), (0, "adversarial")])
def test_folded_consumer_matches_layernorm_then_linear(M, K, N, act, kind):
    x, gamma, beta, W, b, Wf, d, c, y64 = _folded_setup(M, K, N, 3, kind)
    if act == 1:
        y64 = torch.nn.functional.gelu(y64)
    _, x16, st = _produce(x, 64)
    scale = float(y64.abs().max())
    results = {}
    for tile in (64, 128, 256):
        got = _fold_run(x16, st, Wf, d, c, act, None, tile)
        ref = _unfused(x, gamma, beta, W, b, act, None, tile)
        torch.cuda.synchronize()
        e_fold = float((got.cpu().double() - y64).abs().max()) / scale
        e_ref = float((ref.cpu().double() - y64).abs().max()) / scale
        print(f"[ln fold {kind} M={M} K={K} N={N} act={act} tile={tile}] folded {e_fold:.2e} | LayerNorm kernel + GEMM {e_ref:.2e} (of the largest |y|, vs fp64)")
        # typical rows: as close to fp64 as the LayerNorm kernel + GEMM path (x2 + a floor); rows whose mean dwarfs their deviations: see _rows
        assert e_fold <= (2.0 if kind == "typical" else 4.0) * e_ref + 1e-4 and e_fold < 3e-3
        results[tile] = got.cpu()
    assert torch.equal(results[64], results[128]) and torch.equal(results[64], results[256]), "folded rows differ between tile kernels"


@pytest.mark.parametrize("hd,tiles", [(64, (64, 128, 256)), (48, (128192,))])
def test_folded_consumer_with_the_fused_rope_epilogue(hd, tiles):
    """q | k | v projection with 2-D RoPE on q and k in the epilogue (EPI 6 for 64-wide heads, the 128 x 192 tile for 48-wide ones):
    folded == LayerNorm kernel + the same fused-RoPE GEMM, to fp16 rounding of the outputs"""
    M, K = 769, 768
    heads = K // hd
    N = 3 * K
    x, gamma, beta, W, b, Wf, d, c, _ = _folded_setup(M, K, N, 5)
    g = torch.Generator().manual_seed(9)
    pos = torch.randint(-1, 32, (M, 2), generator=g, dtype=torch.int64).to(DEV)
    _, x16, st = _produce(x, 64)
    base = None
    for tile in tiles:
        rope = (pos, 2 * K, 100.0, hd)
        got = _fold_run(x16, st, Wf, d, c, 0, rope, tile).cpu().float()
        ref = _unfused(x, gamma, beta, W, b, 0, rope, tile).cpu().float()
        torch.cuda.synchronize()
        err = float((got - ref).abs().max() / ref.abs().max())
        print(f"[ln fold + rope, heads of {hd}, tile {tile}] folded vs unfused {err:.2e}")
        assert err < 4e-3                      # both carry one fp16 rounding of the operand and one of the output
        assert heads * hd == K
        if base is None:
            base = got
        assert torch.equal(got, base)


def test_fold_arguments_are_checked():
    x, gamma, beta, W, b, Wf, d, c, _ = _folded_setup(130, 192, 128, 7)
    _, x16, st = _produce(x, 64)
    out = torch.empty(130, 128, dtype=F16, device=DEV)
    with pytest.raises(Exception):          # the skinny tile has no folded epilogue
        ops.linear(x16[:32], Wf.to(DEV), out[:32], d.to(DEV), tile=16, ln=(st[:, :32].contiguous(), c.to(DEV), EPS))
    with pytest.raises(Exception):          # statistics of the wrong width
        ops.linear(x16, Wf.to(DEV), out, d.to(DEV), ln=(st[:2].contiguous(), c.to(DEV), EPS))
    with pytest.raises(Exception):          # producer side needs an fp32 output with an fp32 residual
        ops.linear(x16, Wf.to(DEV), out, d.to(DEV), emit=(torch.empty(2, 130, 2, device=DEV), torch.empty(130, 128, dtype=F16, device=DEV)))
