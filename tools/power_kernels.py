#!/usr/bin/env python3
"""Package power and shader clock (amdgpu hwmon, bench.PowerSampler) while ONE kernel class of the step runs in a loop for ~1.5 s each:
where the watts of a step go.  Prints W, MHz, achieved rate and joules per TFLOP (dynamic = above idle).  Run on the GPU box."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cut3r_slam_amd import ops, _lib

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
SEC = float(os.environ.get("POWER_SECONDS", "1.5"))


def run(label, fn, flops=0.0, byts=0.0, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn()
    gr.replay()
    torch.cuda.synchronize()
    s = bench.PowerSampler(0).start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SEC:
        for _ in range(4):
            gr.replay()
        torch.cuda.synchronize()
        n += 4 * reps
    el = time.perf_counter() - t0
    p = s.stop() or {}
    us = el / n * 1e6
    w = p.get("package_w_mean", float("nan"))
    dyn = w - IDLE
    rate = f"{flops / us / 1e6:7.0f} TF/s" if flops else f"{byts / us / 1e6:7.2f} TB/s"
    eff = f"{dyn / (flops / us / 1e6):6.2f} J/PFLOP-dyn... " if False else ""
    jt = f"{dyn / (flops / us / 1e6) :6.3f} J per TFLOP above idle" if flops else f"{dyn / (byts / us / 1e6):6.1f} J per TB above idle"
    print(f"{label:44s} {us:9.1f} us  {rate}  {w:7.0f} W  {p.get('sclk_mhz_mean', float('nan')):6.0f} MHz  {jt}", flush=True)


# idle
s = bench.PowerSampler(0).start()
time.sleep(1.0)
IDLE = (s.stop() or {}).get("package_w_mean", 240.0)
print(f"idle: {IDLE:.0f} W")


def gemm(M, N, K, label, act=0, res=False, f32=False, zeros=False):
    A = (torch.zeros(M, K) if zeros else torch.randn(M, K, generator=g)).half().to(DEV)
    W = (torch.zeros(N, K) if zeros else torch.randn(N, K, generator=g) / K ** 0.5).half().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV) if res else None
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.float16, device=DEV)
    run(label, lambda: ops.linear(A, W, out, b, act, res1=r), flops=2.0 * M * N * K)


lib = _lib.load()
gemm(107520, 3072, 1024, "gemm256 enc qkv (fp16 out)")
gemm(107520, 3072, 1024, "gemm256 enc qkv, all-zero operands", zeros=True)
gemm(107520, 4096, 1024, "gemm256 enc fc1 + GELU", act=1)
gemm(107520, 1024, 4096, "gemm256 enc fc2 + fp32 residual", res=True, f32=True)
gemm(107520, 1024, 1024, "gemm256 enc proj + fp32 residual (HBM-bound)", res=True, f32=True)
gemm(21532, 768, 768, "gemm256 dec proj + fp32 residual (HBM-bound)", res=True, f32=True)
gemm(21532, 3072, 768, "gemm256 dec fc1 + GELU", act=1)
gemm(6152, 3072, 768, "gemm 128x128 dec fc1 + GELU (8 windows)", act=1)
gemm(769, 3072, 768, "gemm 64x64 dec fc1 + GELU (1 window)", act=1)
for B, H, Nq, Nk, D, label in [(140, 16, 768, 768, 64, "attention enc [140,16,768,64]"), (28, 16, 768, 768, 48, "attention dec state [28,16,768,48]"),
                               (28, 12, 256, 256, 128, "attention memory [28,12,256,128]")]:
    q = torch.randn(B, Nq, H, D, generator=g).half().to(DEV)
    k = torch.randn(B, Nk, H, D, generator=g).half().to(DEV)
    v = torch.randn(B, Nk, H, D, generator=g).half().to(DEV)
    o = torch.empty_like(q)
    run(label, lambda: ops.attention(q, k, v, o, D ** -0.5), flops=4.0 * B * H * Nq * Nk * D)
x = torch.randn(107520, 1024, generator=g).to(DEV)
gam, bet = torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)
o16 = torch.empty(107520, 1024, dtype=torch.float16, device=DEV)
run("layernorm 107520 x 1024 (fp32 -> fp16)", lambda: ops.layernorm(x, gam, bet, 1e-6, out16=o16), byts=107520 * 1024 * 6.0)
a = torch.empty(1 << 28, dtype=torch.float16, device=DEV)
b2 = torch.empty_like(a)
run("device copy 512 MB (HBM read + write)", lambda: b2.copy_(a), byts=2.0 * a.numel() * 2)
