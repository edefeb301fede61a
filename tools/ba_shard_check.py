#!/usr/bin/env python3
"""world-size-N check of the edge-sharded dense BA step (run under torch.distributed.run; backend from CUT3R_DIST_BACKEND, default
nccl = RCCL; "gloo" lets several ranks share one GPU in the tests): every rank assembles the reduced normal equations of the source
frames it owns, one all-reduce sums S / vS / diag(H), every rank solves; the result must equal the single-rank step."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from cut3r_slam_amd import ba as B
from cut3r_slam_amd.lietorch import SE3
from tests.test_ba_gpu import _scene

backend = os.environ.get("CUT3R_DIST_BACKEND", "nccl")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = "cuda:0" if backend != "nccl" else f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"
torch.cuda.set_device(dev)
dist.init_process_group(backend)
P, ht, wd, fixedp = 6, 8, 10, 1
poses, disps, intr, ii, jj, tgt, wgt, eta = _scene(P, ht, wd, 7)
f = lambda a: torch.from_numpy(np.asarray(a)).float().to(dev)
args = (f(tgt)[None], f(wgt)[None], f(eta), SE3(f(poses)[None]), f(disps)[None], f(intr)[None], torch.from_numpy(ii), torch.from_numpy(jj))
p1, d1, i1 = B.BA(*args, fixedp=fixedp)
mask = B.shard_edges_by_source(torch.from_numpy(ii), world, rank)
pn, dn, inn = B.BA(*args, fixedp=fixedp, group=dist.group.WORLD, edge_mask=mask)
torch.cuda.synchronize()
sc = float(i1["dx"].abs().max())
assert float((inn["dx"] - i1["dx"]).abs().max()) <= 1e-4 * sc + 1e-7, (inn["dx"], i1["dx"])
assert float((pn.data - p1.data).abs().max()) <= 1e-5
assert float((dn - d1).abs().max()) <= 1e-4 * float(i1["dz"].abs().max()) + 1e-7
# every rank ends with the same poses and disparities
ref = [torch.empty_like(dn.cpu()) for _ in range(world)]
dist.all_gather(ref, dn.cpu()) if backend != "nccl" else None
if backend != "nccl":
    assert all(torch.equal(r, ref[0]) for r in ref)
print(f"rank {rank}/{world}: sharded BA == single-rank BA (|dx| {sc:.3e}, edges here {int(mask.sum())}/{len(ii)}) OK", flush=True)
dist.barrier()
dist.destroy_process_group()
