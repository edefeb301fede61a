#!/usr/bin/env python3
"""LayerNorm fold, kernel by kernel, on the 28-window shapes of the step: consumer GEMMs with and without the folded epilogue, producer
GEMMs with and without the fp16 copy + slab statistics, and the LayerNorm launches the fold removes (graph replays, interleaved arms)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cut3r_slam_amd import ops
from tools.bench_gemm import timeit

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
REPS = 20


def best(fn):
    return min(timeit(fn, reps=REPS) for _ in range(3))


CONS = [(107520, 3072, 1024, 0, True, "enc qkv + rope"), (107520, 4096, 1024, 1, False, "enc fc1 gelu"), (21532, 2304, 768, 0, True, "dec qkv + rope (img)"),
        (21504, 2304, 768, 0, False, "dec qkv (state, 48-wide)"), (21532, 768, 768, 0, True, "dec projq + rope"), (21504, 1536, 768, 0, True, "dec projkv + rope"),
        (21532, 3072, 768, 1, False, "dec fc1 gelu")]
PROD = [(107520, 1024, 1024, "enc proj+res"), (107520, 1024, 4096, "enc fc2+res"), (21532, 768, 768, "dec proj+res"), (21532, 768, 3072, "dec fc2+res")]
tot_plain = tot_fold = 0.0
for M, N, K, act, rope, label in CONS:
    x = torch.randn(M, K, generator=g)
    A = x.half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    c = W.float().sum(1).contiguous()
    st = torch.stack([x.view(M, K // 64, 64).sum(-1), x.view(M, K // 64, 64).var(-1, unbiased=False) * 64], -1).permute(1, 0, 2).contiguous().to(DEV)
    out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    n = torch.arange(M, device=DEV)
    pos = torch.stack([(n // 32) % 24, n % 32], 1).contiguous()
    rp = (pos, (2 * N // 3 if N % 3 == 0 and N != 768 and N != 1536 else (N if N == 768 else N // 2)), 100.0) if rope else None
    t_plain = best(lambda: ops.linear(A, W, out, b, act, rope=rp))
    t_fold = best(lambda: ops.linear(A, W, out, b, act, rope=rp, ln=(st, c, 1e-6)))
    t_plain2 = best(lambda: ops.linear(A, W, out, b, act, rope=rp))
    xf = x.to(DEV)
    gam, bet = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    ln16 = torch.empty(M, K, dtype=torch.float16, device=DEV)
    t_ln = best(lambda: ops.layernorm(xf, gam, bet, 1e-6, ln16, None))
    tot_plain += min(t_plain, t_plain2) + t_ln
    tot_fold += t_fold
    print(f"consumer {label:26s} {M:6d}x{N:5d}x{K:5d}  plain {min(t_plain, t_plain2):8.1f} us | folded {t_fold:8.1f} us ({t_fold / min(t_plain, t_plain2) - 1:+.1%}) | "
          f"LayerNorm launch it replaces {t_ln:6.1f} us", flush=True)
    del x, A, W, st, out, xf, ln16
for M, N, K, label in PROD:
    A = torch.randn(M, K, generator=g).half().to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    x16 = torch.empty(M, N, dtype=torch.float16, device=DEV)
    st = torch.empty(N // 64, M, 2, device=DEV)
    t_plain = best(lambda: ops.linear(A, W, out, b, 0, out))
    t_emit = best(lambda: ops.linear(A, W, out, b, 0, out, emit=(st, x16)))
    t_plain2 = best(lambda: ops.linear(A, W, out, b, 0, out))
    tot_plain += min(t_plain, t_plain2)
    tot_fold += t_emit
    print(f"producer {label:26s} {M:6d}x{N:5d}x{K:5d}  plain {min(t_plain, t_plain2):8.1f} us | + fp16 copy + statistics {t_emit:8.1f} us ({t_emit / min(t_plain, t_plain2) - 1:+.1%})", flush=True)
print(f"sum over the listed launches (one each): LayerNorm + plain {tot_plain:.0f} us, folded {tot_fold:.0f} us")
