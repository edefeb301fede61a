"""CPU, world_size 2 and 4, gloo: the window-sharded tracking driver (cut3r_slam_amd/dist.py) -- window assignment, the
all-gather exchange and the replicated in-order replay.  The network and the HIP chaining are replaced by fakes here
(this file runs without a GPU); the GPU path uses the same driver with backend "nccl" (RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cut3r_slam_amd.dist import ShardedTracker, window_ranges


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeKF:
    def __init__(self):
        self._n = 0
        self.log = []

        class C:
            pass
        self.counter = C()
        self.counter.value = 0


class _FakeSlam:
    def __init__(self):
        self.keyframes = _FakeKF()

        class T:
            t1 = 6
        self.tracker = T()


def _worker(rank, world, port, q, wb):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slam = _FakeSlam()
    slam.keyframes.counter.value = 7
    appended, tracked, lag = [], [], []

    def append_fn(k, frame, tstamp, intr, mine):
        assert k == slam.keyframes.counter.value
        slam.keyframes.counter.value += 1
        appended.append((k, int(tstamp), bool(mine)))

    def infer_fn(ranges):
        # "network output" that encodes who computed which window: [wb*6, ...] like the batched decoder returns
        assert len(ranges) == wb
        pts = torch.cat([torch.full((6, 4, 5, 3), float(100 * rank + t0)) for t0, _ in ranges])
        conf = torch.cat([torch.full((6, 4, 5), float(t0)) for t0, _ in ranges])
        pose = torch.full((6 * wb, 7), float(rank))
        lag.append(len(tracked))
        return pts, conf, pose

    def track_fn(t0, t1, outs):
        assert outs[0].shape[0] == 6
        tracked.append((t0, t1, float(outs[0][0, 0, 0, 0]), float(outs[2][0, 0])))

    st = ShardedTracker(slam, world, rank, wb=wb, infer_fn=infer_fn, track_fn=track_fn, append_fn=append_fn)
    frames = torch.zeros(st.frames_needed(2, 10, 5), 1)
    t = 61
    for _ in range(2):
        t = st.step(frames, t, 10, 5, None)
    n_before_flush = len(tracked)
    st.flush()
    # the count exchange of the sharded overlap counting: every rank contributes the rows of the windows it owns
    counts = torch.zeros(world * wb, 6, 2, 64, dtype=torch.int32)
    counts[rank * wb:(rank + 1) * wb] = rank + 1
    total = st._exchange_counts(counts)
    for r in range(world):
        assert bool((total[r * wb:(r + 1) * wb] == r + 1).all())
    q.put((rank, appended, tracked, t, slam.tracker.t1, lag, n_before_flush))
    dist.barrier()
    dist.destroy_process_group()


def test_window_ranges_chain():
    r = window_ranges(5, 4)
    assert r == [(5, 11), (10, 16), (15, 21), (20, 26)]
    assert all(a[1] - 1 == b[0] for a, b in zip(r, r[1:]))         # consecutive windows share one keyframe


def _run_world2(wb):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, wb)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sharded_tracker_world2_gloo():
    (r0, app0, trk0, t0, t1_0, lag0, nb0), (r1, app1, trk1, t1, t1_1, lag1, nb1) = _run_world2(1)
    # both ranks replay the same windows, in sequence order, with the owner's outputs
    assert trk0 == trk1
    assert [(a, b) for a, b, _, _ in trk0] == [(5, 11), (10, 16), (15, 21), (20, 26)]
    assert [owner for *_, owner in trk0] == [0.0, 1.0, 0.0, 1.0]
    assert [v for _, _, v, _ in trk0] == [5.0, 110.0, 15.0, 120.0]
    assert t0 == t1 == 61 + 2 * 2 * 50 and t1_0 == t1_1 == 26
    # every keyframe is registered on every rank, encoded by exactly one
    assert [(k, ts) for k, ts, _ in app0] == [(k, ts) for k, ts, _ in app1] == [(7 + i, 70 + 10 * i) for i in range(20)]
    # `mine` = this rank needs the keyframe's pixels: its own windows' keyframes, plus the keyframe its first window shares with the
    # previous rank's last window (rank 1: k = 10, 20) and, for rank 0, the step's last keyframe = the first of its next step (k = 15, 25)
    for (k, _, m0), (_, _, m1) in zip(app0, app1):
        assert (m0 and m1) if k in (10, 15, 20, 25) else (m0 != m1), (k, m0, m1)
    # software pipeline: the network pass of step 2 is issued BEFORE step 1 is replayed; flush() drains the last step
    assert lag0 == lag1 == [0, 0] and nb0 == nb1 == 2


def test_sharded_tracker_world2_window_batch2():
    (r0, app0, trk0, t0, t1_0, lag0, nb0), (r1, app1, trk1, t1, t1_1, lag1, nb1) = _run_world2(2)
    assert trk0 == trk1
    want = [(5 + 5 * j, 11 + 5 * j) for j in range(8)]
    assert [(a, b) for a, b, _, _ in trk0] == want
    # rank r owns wb=2 consecutive windows of every step
    assert [owner for *_, owner in trk0] == [0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0, 1.0]
    assert [v for _, _, v, _ in trk0] == [a + 100 * o for (a, _), o in zip(want, [0, 0, 1, 1, 0, 0, 1, 1])]
    assert t0 == t1 == 61 + 2 * 4 * 50 and t1_0 == t1_1 == 46
    assert [(k, ts) for k, ts, _ in app0] == [(k, ts) for k, ts, _ in app1] == [(7 + i, 70 + 10 * i) for i in range(40)]
    # `mine` = this rank needs the keyframe's pixels: its own windows' keyframes, plus the keyframe its first window shares with the
    # previous rank's last window (blocks of wb * 5 = 10 keyframes: k = 15, 35 for rank 1) and, for rank 0, the step's last keyframe = the first of its next step (k = 25, 45)
    for (k, _, m0), (_, _, m1) in zip(app0, app1):
        assert (m0 and m1) if k in (15, 25, 35, 45) else (m0 != m1), (k, m0, m1)
    # keyframes 7..16 belong to rank 0's two windows of step 1 (the shared keyframe 5/6 were initialised earlier)
    assert [m for k, _, m in app0 if k <= 15] == [True] * 9
    assert nb0 == 4


def test_memory_plan_of_the_drivers_8_gpu_job_fits_one_mi355x():
    """the replicated stores of the driver's SCALE run (N = 8, --steps 20 --warmup 5, 28 windows per rank and step) per rank, with the
    network's measured workspace (bench.py reports `hbm_peak_gb` at N = 1: weights + graphs' static buffers + activations of a 28-window
    pass), must fit the 288 GB of one MI355X with a margin; the plan grows linearly with the steps, so its slope is checked too"""
    from cut3r_slam_amd.dist import memory_plan
    plan = memory_plan(world=8, steps=25, wb=28, workspace_bytes=60 * 10**9)
    assert plan["keyframes"] == 7 + 5 * 28 * 8 * 25 + 2 + 8
    assert plan["fits"] and plan["total"] < 0.6 * plan["hbm"], plan
    per_kf = (memory_plan(8, 26, 28)["total"] - memory_plan(8, 25, 28)["total"]) / (5 * 28 * 8)
    assert 2.2e6 < per_kf < 2.5e6, per_kf                      # image 0.59 + depth 0.79 + 6/5 x (stride-2 pointmap 0.59 + confidence 0.20) MB
    assert not memory_plan(world=8, steps=120, wb=28, workspace_bytes=60 * 10**9)["fits"]      # (the bound is real: ~95 steps at N = 8)


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [4, 8])
def test_sharded_tracker_more_ranks_window_batch2(world):
    """four and eight gloo ranks (the driver's scaling bench launches N = 2, 4, 8): every rank replays ALL windows in sequence order with
    the owning rank's network outputs, rank r owns the r-th block of wb consecutive windows of every step, every keyframe is registered on
    every rank and its pixels are kept by its owner(s) only, the count exchange returns every owner's rows."""
    wb, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, wb)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    trk = res[0][2]
    nwin = 2 * world * wb                                            # two steps
    want = [(5 + 5 * j, 11 + 5 * j) for j in range(nwin)]
    owners = [float((j // wb) % world) for j in range(nwin)]
    for r, app, tracked, t, t1, lag, nb in res:
        assert tracked == trk and t == 61 + 2 * world * wb * 50 and t1 == want[-1][1]
        assert [(k, ts) for k, ts, _ in app] == [(7 + i, 70 + 10 * i) for i in range(5 * nwin)]
        assert lag == [0, 0] and nb == world * wb
    assert [(a, b) for a, b, _, _ in trk] == want and [o for *_, o in trk] == owners
    assert [v for _, _, v, _ in trk] == [a + 100 * o for (a, _), o in zip(want, owners)]
    # every keyframe's pixels live on one rank, the shared keyframes of neighbouring blocks on two
    mine = np.asarray([[m for _, _, m in app] for _, app, *_ in res])
    per_kf = mine.sum(0)
    assert per_kf.min() >= 1 and per_kf.max() <= 2 and int((per_kf == 2).sum()) == 2 * world
