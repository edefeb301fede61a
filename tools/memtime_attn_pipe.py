#!/usr/bin/env python3
"""s_memtime instrumentation of attn_pipe_kernel (developer tool; the product is not touched).  Patches a COPY of
cut3r_slam_amd/csrc/attention.hip (build/memtime/): wave 0 of one workgroup in the middle of the grid stamps the shader clock in
steady-state steps 3..8 at: step top | after the counted wait | after the barrier | after the DMA issue | end of the step's compute.
Run on the GPU box:  CUT3R_ATTN_PIPE=1 [CUT3R_ATTN_PIPE_NW=4|8] python tools/memtime_attn_pipe.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "cut3r_slam_amd/csrc/attention.hip")).read()
out_dir = os.path.join(ROOT, "build", "memtime")
os.makedirs(out_dir, exist_ok=True)


def sub(old, new, count=1):
    global src
    assert old in src, "anchor not found (attention.hip changed): " + old[:70]
    src = src.replace(old, new, count)


src = src.replace('#include "common.h"', '#include "../../cut3r_slam_amd/csrc/common.h"')
sub("namespace {\n", '''__device__ unsigned long long g_dbg[16][8];
extern "C" int cut3r_dbg_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 128); }
#define TS(i) do { if (dbg_on) asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(ts[i]) :: "memory"); } while (0)
namespace {
''')
sub("        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n        if (HASA) {\n            wait_tiles_ahead<2 * PPW, AHEAD>(ntiles - 2 - j);",
    "        unsigned long long ts[6] = {0};\n        const bool dbg_on = HASA && HASC && HASB && (int)blockIdx.x == (int)gridDim.x / 2 + 3 && wave == 0 && j >= 3 && j <= 8;\n"
    "        TS(0);\n        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n        if (HASA) {\n            wait_tiles_ahead<2 * PPW, AHEAD>(ntiles - 2 - j);\n            TS(1);")
sub("            asm volatile(\"s_barrier\" ::: \"memory\");\n            issue(j + NST - 2);\n        }\n        half8_t vf[2][2][DP], kf[2][DQ];",
    "            asm volatile(\"s_barrier\" ::: \"memory\");\n            TS(2);\n            issue(j + NST - 2);\n            TS(3);\n        }\n        half8_t vf[2][2][DP], kf[2][DQ];")
sub("            if (__any(gflag)) {                  // wave-uniform: after the first tiles the running maxima rarely move",
    "            TS(4);\n            if (dbg_on && lane == 0) for (int i = 0; i < 5; i++) g_dbg[j][i] = ts[i];\n"
    "            if (__any(gflag)) {                  // wave-uniform: after the first tiles the running maxima rarely move")
# shader clock vs the constant 100 MHz counter over the whole workgroup: the clock this kernel really runs at
sub("    if (bh >= HB) return;                        // (the grid is padded to whole groups of 8 (batch, head) pairs)",
    "    if (bh >= HB) return;\n    unsigned long long c0, r0;\n    asm volatile(\"s_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(c0), \"=s\"(r0) :: \"memory\");")
sub("#pragma unroll\n    for (int qb = 0; qb < QB; qb++) {\n        float l0_, l1_;\n        halves(l_run[qb], l0_, l1_);",
    "    if ((int)blockIdx.x == (int)gridDim.x / 2 + 3 && wave == 0) {\n        unsigned long long c1, r1;\n"
    "        asm volatile(\"s_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(c1), \"=s\"(r1) :: \"memory\");\n"
    "        if (lane == 0) { g_dbg[15][0] = c1 - c0; g_dbg[15][1] = r1 - r0; }\n    }\n"
    "#pragma unroll\n    for (int qb = 0; qb < QB; qb++) {\n        float l0_, l1_;\n        halves(l_run[qb], l0_, l1_);")
path = os.path.join(out_dir, "attention_memtime.hip")
open(path, "w").write(src)
lib = os.path.join(out_dir, "libattn_memtime.so")
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-inline-asm", "-I" + os.path.join(ROOT, "include"), "-shared", "-o", lib, path])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch  # noqa: E402
L = ctypes.CDLL(lib)
B, H, Nq, Nk, D = 140, 16, 768, 768, 64
g = torch.Generator().manual_seed(0)
q = torch.randn(B, Nq, H, D, generator=g).half().cuda()
k = torch.randn(B, Nk, H, D, generator=g).half().cuda()
v = torch.randn(B, Nk, H, D, generator=g).half().cuda()
o = torch.empty_like(q)
P = ctypes.c_void_p
L.cut3r_attention_f16.argtypes = [P, P, P, P] + [ctypes.c_int] * 5 + [ctypes.c_longlong] * 8 + [ctypes.c_float, P]
for _ in range(3):
    rc = L.cut3r_attention_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Nq, Nk, D, q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                               v.stride(0), v.stride(1), o.stride(0), o.stride(1), D ** -0.5, None)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
L.cut3r_dbg_read(buf)
names = ["lgkm + counted vmcnt wait", "barrier", "DMA issue", "compute (reads, 16 MFMAs, softmax)"]
print(f"rc={rc}  env: " + " ".join(f"{k_}={v_}" for k_, v_ in os.environ.items() if k_.startswith("CUT3R_ATTN")))
cyc, ref = buf[15 * 8], buf[15 * 8 + 1]
if ref:
    print(f"workgroup lifetime {cyc} shader cycles = {ref} ticks of the 100 MHz counter -> {cyc / ref * 0.1:.3f} GHz in this kernel; {cyc / 12:.0f} cycles per key tile")
prev_end = None
for j in range(3, 9):
    ts = [buf[j * 8 + i] for i in range(5)]
    if ts[0] == 0:
        continue
    seg = [ts[i + 1] - ts[i] for i in range(4)]
    gap = (ts[0] - prev_end) if prev_end else 0
    prev_end = ts[4]
    print(f"step {j}: " + ", ".join(f"{n} {s}" for n, s in zip(names, seg)) + f" | step total {ts[4] - ts[0]} (+{gap} to the next stamp: rescale branch)")
