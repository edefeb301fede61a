"""Throughput driver of the tracking loop: window-sharded across GPUs and software-pipelined on each GPU.

One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

What shards: every `inference()` call re-initialises the recurrent state and the pose memory
(/root/reference/src/dust3r/model.py:819-822), so tracking WINDOWS are independent network evaluations; only the cheap
post-hoc chaining (/root/reference/hislam2/track_frontend.py:216-234: needs the previous window's last depth and pose)
and the graph update are sequential.  Per step every rank pushes `wb` consecutive windows through the network in one
batched pass (encoder for its new keyframes, decoder + heads batched over the windows), the three consumed outputs
(pts3d_in_self_view, conf_self, camera_pose: 19 MB per window at 384x512) are exchanged with ONE all_gather per tensor
-- small messages: latency, not ring bandwidth, matters -- and every rank replays the chaining + graph update of all
world*wb windows in sequence order, keeping the keyframe store and the graph replicated (no second collective; any rank
can serve the trajectory).

Pipelining: the replay of step s runs on a side HIP stream while the network pass of step s+1 is already executing on
the main stream (the replay is host-latency-bound: tiny kernels + small device->host reads), so a step costs
max(network time, replay time) instead of their sum.  `flush()` drains the last replay.
"""
from __future__ import annotations

import contextlib
import os
import time
from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def window_ranges(first_t0: int, count: int, win: int = 5) -> List[Tuple[int, int]]:
    """keyframe ranges [t0, t1) of `count` consecutive windows; consecutive windows share one keyframe."""
    return [(first_t0 + win * j, first_t0 + win * j + win + 1) for j in range(count)]


def all_gather_outputs(outs: Sequence[torch.Tensor], world: int, force_collective: bool = False) -> List[torch.Tensor]:
    """outs: this rank's tensors (same shapes on every rank, leading dim = its windows*views).  Returns each tensor
    concatenated over ranks in rank order.  world == 1: private copies (the network outputs are static graph buffers
    that the next replay overwrites)."""
    if not dist.is_initialized() or (world == 1 and not force_collective):
        return [t.clone() for t in outs]
    res = []
    for t in outs:
        t = t.contiguous()
        if t.is_cuda:
            buf = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(buf.view(-1), t.view(-1))       # one RCCL all-gather per tensor
        else:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)                                   # gloo (CPU tests)
            buf = torch.stack(parts, 0)
        res.append(buf.reshape((world * t.shape[0],) + tuple(t.shape[1:])))
    return res


class ShardedTracker:
    """Drives a replicated `Cut3rSlam` with window-sharded, batched, pipelined network inference."""

    def __init__(self, slam, world: int, rank: int, wb: int = 1, infer_fn: Callable = None, track_fn: Callable = None,
                 append_fn: Callable = None, pipelined: bool = True, views: int = 6, force_collective: bool = False):
        self.slam, self.world, self.rank, self.wb = slam, world, rank, max(1, int(wb))
        self.infer_fn = infer_fn or self._infer
        self.track_fn = track_fn          # None: TrackFrontend.track_many over all windows of the step (one round trip each)
        self.append_fn = append_fn or self._append
        self.pipelined = pipelined
        self.views = views
        self.force_collective = force_collective    # world == 1 rehearsal of the RCCL exchange
        self.emulate_gather = False
        self._pending = None
        self._next_t0 = None            # first keyframe of the next window to be scheduled
        self._side = None
        self._pose_pinned = [None, None]
        self._first_event = None
        self.side_priority = int(os.environ.get("CUT3R_SIDE_PRIORITY", "0"))         # 0: measured best; -1 (high) starves the network pass once the host runs ahead
        self.stats = {"append_s": 0.0, "issue_s": 0.0, "replay_s": 0.0, "replay_wait_s": 0.0, "exchange_s": 0.0, "steps": 0}     # host wall-clock per phase

    def frames_needed(self, total_steps: int, kf_every: int, win: int) -> int:
        """frames consumed by the 7-keyframe initialisation plus `total_steps` steps"""
        return (7 + win * self.world * self.wb * total_steps + 1) * kf_every + 1

    # ---- default callbacks on the real SLAM objects
    def _append(self, kf_index: int, frame, tstamp, intr, mine: bool):
        # fixed cadence: the encoder pass of a keyframe is deferred to the rank that owns its window
        self.slam.keyframes.append(tstamp, frame[0], None, None, None, None, intr, None, None)

    def _infer(self, ranges):
        """batched network pass over this rank's windows -> (pts [wb*V,H,W,3], conf [wb*V,H,W], pose [wb*V,7])"""
        tr, kf = self.slam.tracker, self.slam.keyframes
        tr.window_features(ranges[0][0], ranges[-1][1])                 # encode the not-yet-encoded keyframes, batched
        feats = torch.stack([tr.window_features(a, b) for a, b in ranges], 0)
        res = self.slam.model.decode_windows(feats, kf.ht, kf.wd)
        return tuple(res[k] for k in ("pts3d_in_self_view", "conf_self", "camera_pose"))

    # ---- the pipeline
    def _side_ctx(self, gathered, ev):
        if ev is None or not self.pipelined:
            return contextlib.nullcontext()
        if self._side is None:
            self._side = torch.cuda.Stream(priority=self.side_priority)
        return torch.cuda.stream(self._side)

    def _prefetch(self, pending):
        """queue the log-depth reduction of the pending replay's first window (side stream, behind the exchange event)"""
        self._first_event = None
        if pending is None or self.track_fn is not None or not self.pipelined:
            return
        ranges_all, gathered, ev = pending
        if ev is None:
            return
        with self._side_ctx(gathered, ev):
            torch.cuda.current_stream().wait_event(ev)
            V = self.views
            self._first_event = self.slam.tracker.prefetch_logdepth(ranges_all[0][0], gathered[0][:V], gathered[2][:V])

    def _replay(self, pending):
        ranges_all, gathered, ev = pending
        V = self.views
        tic = time.perf_counter()
        if ev is not None and self.pipelined:
            if self._side is None:
                self._side = torch.cuda.Stream(priority=self.side_priority)
            self._side.wait_event(ev)
            ev.synchronize()                        # pinned pose copy complete (issued a whole network pass ago)
            self.stats["replay_wait_s"] += time.perf_counter() - tic
            for g in gathered:
                if g.is_cuda:
                    g.record_stream(self._side)     # allocated on the main stream, consumed on the side stream
            ctx = torch.cuda.stream(self._side)
        else:
            if ev is not None:
                ev.synchronize()
                self.stats["replay_wait_s"] += time.perf_counter() - tic
            ctx = contextlib.nullcontext()
        with ctx:
            outs = [tuple(g[j * V:(j + 1) * V] for g in gathered) for j in range(len(ranges_all))]
            if self.track_fn is not None:
                for (a, b), o in zip(ranges_all, outs):
                    self.track_fn(a, b, o)
                    self.slam.tracker.t1 = b
            else:
                self.slam.tracker.track_many(ranges_all, outs, first_event=self._first_event)
                self._first_event = None
                self.slam.tracker.t1 = ranges_all[-1][1]
        self.stats["replay_s"] += time.perf_counter() - tic

    def step(self, frames, t, kf_every, win, intr):
        """Advance world*wb windows (= world*wb*win*kf_every frames).  Returns the new frame counter."""
        slam, world, rank, wb = self.slam, self.world, self.rank, self.wb
        if self._next_t0 is None:
            self._next_t0 = slam.tracker.t1 - 1
        first_t0 = self._next_t0
        ranges_all = window_ranges(first_t0, world * wb, win)
        mine = ranges_all[rank * wb:(rank + 1) * wb]
        # 1. keyframe filter in fixed-cadence mode: every kf_every-th frame is a keyframe (motion_filter.py:83,109,124);
        #    every rank registers all of them, only the owner of a window ever encodes them
        n_frames = world * wb * win * kf_every
        tic0 = time.perf_counter()
        for f in range(t, t + n_frames):
            if f % kf_every == 0:
                k = slam.keyframes.counter.value
                owner = min(max((k - first_t0 - 1) // (win * wb), 0), world - 1) if k > first_t0 else 0
                self.append_fn(k, frames[f:f + 1], f, intr, owner == rank)
        # 1b. the first device step of the pending replay goes onto the side stream BEFORE the network pass is queued
        self._prefetch(self._pending)
        # 2. this rank's windows through the network (asynchronous on the main stream)
        tic = time.perf_counter()
        self.stats["append_s"] += tic - tic0
        outs = self.infer_fn(mine)
        self.stats["issue_s"] += time.perf_counter() - tic
        # 3. meanwhile: replay the previous step's chaining + graph update (host-bound) on the side stream
        if self._pending is not None:
            self._replay(self._pending)
            self._pending = None
        # 4. one exchange over xGMI (private copies when world == 1)
        tic = time.perf_counter()
        if self.emulate_gather:      # debug (bench CUT3R_EMULATE_WORLD): this rank's outputs stand in for every other rank's
            gathered = [torch.cat([t] * world, 0) for t in outs]
        else:
            gathered = all_gather_outputs(outs, world, self.force_collective)
        self.stats["exchange_s"] += time.perf_counter() - tic
        self.stats["steps"] += 1
        ev = None
        if gathered[0].is_cuda:
            # the V x 7 camera poses go to pinned host memory now, on the main stream: the replay never waits for them
            slot = self.stats["steps"] & 1          # two pinned buffers: the previous step's poses are still being replayed
            if self._pose_pinned[slot] is None or self._pose_pinned[slot].shape != gathered[2].shape:
                self._pose_pinned[slot] = torch.empty(gathered[2].shape, dtype=gathered[2].dtype).pin_memory()
            pose_host = self._pose_pinned[slot]
            pose_host.copy_(gathered[2], non_blocking=True)
            gathered = [gathered[0], gathered[1], pose_host]
            ev = torch.cuda.Event()
            ev.record()
        self._pending = (ranges_all, gathered, ev)
        self._next_t0 = ranges_all[-1][1] - 1
        if not self.pipelined:
            self.flush()
        return t + n_frames

    def flush(self):
        """drain the pipeline: replay the last step and join the side stream"""
        if self._pending is not None:
            self._replay(self._pending)
            self._pending = None
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
