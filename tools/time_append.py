#!/usr/bin/env python3
"""Which host call of KeyFrame.append blocks while a long graph replay is in flight?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.side_latency import make_graph

DEV = "cuda:0"
gr, keep = make_graph(15360, 4096, 1024, 150)     # ~30 ms of GPU work per replay
img = torch.zeros(40, 3, 384, 512, dtype=torch.uint8, device=DEV)
src = torch.ones(1, 3, 384, 512, dtype=torch.uint8, device=DEV)
ts = torch.zeros(40, device=DEV)
intr_host = torch.zeros(40, 4)
intr = torch.tensor([1.0, 2, 3, 4])


def t(fn):
    a = time.perf_counter(); fn(); return 1e3 * (time.perf_counter() - a)


for name, fn in {
    "tstamp[i] = float": lambda: ts.__setitem__(3, 5.0),
    "image[i].copy_(frame[0])": lambda: img[3].copy_(src[0].to(DEV, non_blocking=True)),
    "intrinsic host write": lambda: intr_host.__setitem__(3, torch.as_tensor(intr, dtype=torch.float).reshape(-1)[:4]),
    "as_tensor(list, device)": lambda: torch.as_tensor([1, 2, 3], device=DEV),
    "graph replay (same exec again)": lambda: gr.replay(),
}.items():
    torch.cuda.synchronize()
    gr.replay()
    ms = t(fn)
    torch.cuda.synchronize()
    print(f"{name:34s} {ms:8.3f} ms while a 30 ms replay is in flight")
