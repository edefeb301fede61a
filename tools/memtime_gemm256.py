#!/usr/bin/env python3
"""s_memtime instrumentation of the 256x256 GEMM kernel (developer tool; the product is not touched).

Patches a COPY of cut3r_slam_amd/csrc/gemm.hip (build/memtime/) so that one wave of each of the kernel's two wave groups of
workgroup 0 stamps the shader clock at the boundaries of
    tile    : address set-up + prologue issue | first data + barrier | main loop | re-join barrier | epilogue | store drain
    ktile   : the segments of K-tile 5 of the two-phase main loop (16 reads | 2 DMA | counted waits | barrier | 32 MFMAs | barrier |
              8 reads | 6 DMA)
builds a side library (never the in-tree one) and runs three Linear shapes through it.  Every stamp is `s_memtime` followed by
`s_waitcnt lgkmcnt(0)`: outstanding LDS reads are drained at a stamp, so a segment that ends in a stamp includes the return of
the reads issued in it.  Run on the GPU box:   python tools/memtime_gemm256.py tile|ktile
The numbers quoted in DESIGN.md section 5 / profiles/r02/memtime.md came from this procedure."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "tile"
src = open(os.path.join(ROOT, "cut3r_slam_amd/csrc/gemm.hip")).read()
out_dir = os.path.join(ROOT, "build", "memtime")
os.makedirs(out_dir, exist_ok=True)


def sub(old, new):
    global src
    assert old in src, "anchor not found (gemm.hip changed): " + old[:60]
    src = src.replace(old, new, 1)


src = src.replace('#include "common.h"', '#include "../../cut3r_slam_amd/csrc/common.h"')
sub("namespace {\n", '''__device__ unsigned long long g_dbg[2][16];
extern "C" int cut3r_dbg_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 32); }
#define TS(i) do { if (dbg_on) asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(ts[i]) :: "memory"); } while (0)
namespace {
''')
if mode == "tile":
    names = ["set-up + prologue issue", "first data + barrier", "main loop", "re-join barrier", "epilogue (last store issued)", "store drain"]
    sub("    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    const int wr = wave >> 2, wc = wave & 3;\n    const int z = bz;",
        "    unsigned long long ts[8] = {0};\n    const bool dbg_on = true;\n    TS(0);\n    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n"
        "    const int wr = wave >> 2, wc = wave & 3;\n    const int z = bz;")
    sub("    if (nt > 1) wait_vmcnt<6>(); else wait_vmcnt<0>();\n    CUT3R_BARRIER();\n    if (wr == 1) CUT3R_BARRIER();",
        "    TS(1);\n    if (nt > 1) wait_vmcnt<6>(); else wait_vmcnt<0>();\n    CUT3R_BARRIER();\n    TS(2);\n    if (wr == 1) CUT3R_BARRIER();")
    sub("    if (wr == 0) CUT3R_BARRIER();          // re-join", "    TS(3);\n    if (wr == 0) CUT3R_BARRIER();          // re-join")
    sub("#undef CUT3R_QUADRANT\n", "#undef CUT3R_QUADRANT\n    TS(4);\n")
    sub("#undef CUT3R_BARRIER\n}", "    TS(5);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    TS(6);\n"
        "    if (bx == 0 && (wave == 0 || wave == 4) && lane == 0)\n        for (int i = 0; i < 7; i++) g_dbg[wave >> 2][i] = ts[i];\n#undef CUT3R_BARRIER\n}")
    order = [0, 1, 2, 3, 5, 6]        # ts[3] -> ts[4] is the re-join barrier; ts[4] -> ts[5] the epilogue
    nts = 7
else:
    names = ["16 reads", "2 DMA", "counted waits", "barrier", "32 MFMAs (issue)", "barrier", "8 reads", "6 DMA"]
    sub("    for (int t = 0; t < nt; t++) {\n        const unsigned char* buf = smem + (t & 1) * BUF;\n        // phase A: rows 0..63 of the wave tile\n        read_a(buf);",
        "    unsigned long long ts[9] = {0};\n    for (int t = 0; t < nt; t++) {\n        const unsigned char* buf = smem + (t & 1) * BUF;\n"
        "        const bool dbg_on = (t == 5) && bx == 0 && (wave == 0 || wave == 4);\n        TS(0);\n        // phase A: rows 0..63 of the wave tile\n        read_a(buf);")
    sub("        issue_a(1, t + 1, 3);\n        if (t + 1 < nt) wait_vmcnt<8>(); else wait_vmcnt<0>();", "        TS(1);\n        issue_a(1, t + 1, 3);\n        TS(2);\n        if (t + 1 < nt) wait_vmcnt<8>(); else wait_vmcnt<0>();")
    sub("        CUT3R_BARRIER();\n        CUT3R_QUADRANT(0, 0, fb0);\n        CUT3R_QUADRANT(0, 2, fb1);\n        CUT3R_BARRIER();",
        "        TS(3);\n        CUT3R_BARRIER();\n        TS(4);\n        CUT3R_QUADRANT(0, 0, fb0);\n        CUT3R_QUADRANT(0, 2, fb1);\n        TS(5);\n        CUT3R_BARRIER();\n        TS(6);")
    sub("        read_a(buf + 3 * UNIT);\n        issue_a(0, t + 2, 0); issue_b(0, t + 2, 1); issue_b(1, t + 2, 2);",
        "        read_a(buf + 3 * UNIT);\n        TS(7);\n        issue_a(0, t + 2, 0); issue_b(0, t + 2, 1); issue_b(1, t + 2, 2);\n        TS(8);\n"
        "        if (dbg_on && lane == 0)\n            for (int i = 0; i < 9; i++) g_dbg[wave >> 2][i] = ts[i];")
    nts = 9
open(os.path.join(out_dir, "gemm_memtime.hip"), "w").write(src)
csrc = os.path.join(ROOT, "cut3r_slam_amd", "csrc")
so = os.path.join(out_dir, "libmemtime.so")
if not os.path.isfile(os.path.join(out_dir, "gemm_memtime.o")) or "--no-build" not in sys.argv:
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                           "-ffp-contract=off", "-c", os.path.join(out_dir, "gemm_memtime.hip"), "-o", os.path.join(out_dir, "gemm_memtime.o")])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(out_dir, "gemm_memtime.o")] +
                          [os.path.join(csrc, o) for o in ("attention.o", "elementwise.o", "geometry.o", "lie.o", "lc.o", "ba.o")])
if "--build-only" in sys.argv:
    print("built", so)
    sys.exit(0)

sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cut3r_slam_amd import _lib  # noqa: E402
_lib.LIB_PATH = so
_lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in ("cut3r_abi_version", "cut3r_gemm_f16", "cut3r_gemm_tile_for")}
from cut3r_slam_amd import ops  # noqa: E402

dev = "cuda:0"
for (M, N, K) in ((30720, 3072, 1024), (21532, 768, 768), (21532, 768, 3072)):
    A = torch.randn(M, K, device=dev).half()
    W = torch.randn(N, K, device=dev).half()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3):
        ops.linear(A, W, out, b, tile=256)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.cut3r_dbg_read.argtypes = [ctypes.c_void_p]
    buf = (ctypes.c_ulonglong * 32)()
    rc = lib.cut3r_dbg_read(buf)
    print(f"{M}x{N}x{K} (fp16 out + bias), rc={rc}, cycles:")
    for g in range(2):
        ts = [int(buf[g * 16 + i]) for i in range(nts)]
        if mode == "tile":
            seg = [ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5]]
        else:
            seg = [ts[i + 1] - ts[i] for i in range(8)]
        print(f"  wave group {g}: " + ", ".join(f"{n} {v}" for n, v in zip(names, seg)) + f" | total {ts[nts - 1] - ts[0]}")
