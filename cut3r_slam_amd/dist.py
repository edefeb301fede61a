"""Throughput driver of the tracking loop: window-sharded across GPUs and software-pipelined on each GPU.

One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

What shards: every `inference()` call re-initialises the recurrent state and the pose memory
(/root/reference/src/dust3r/model.py:819-822), so tracking WINDOWS are independent network evaluations; only the post-hoc
chaining (/root/reference/hislam2/track_frontend.py:216-234: needs the previous window's last depth and pose) and the
graph decisions are sequential.  Per step every rank
  1. pushes `wb` consecutive windows through the network in one batched pass (encoder for its new keyframes, decoder +
     heads batched over the windows),
  2. exchanges 352 bytes per window: the two log-depth sums and the six raw poses the chain needs (all_gather), after which
     every rank knows every scale and every chained pose by a host scan (log-scales add, poses compose: SURVEY 8(e)(2)),
  3. aligns and stores ITS OWN windows (O(wb) device work), completes the stride-2 pointmap / confidence stores of the step
     with two in-place all_gathers (4.7 MB per window instead of the 19 MB of full-resolution outputs; depth rows only on
     request), counts the O(#keyframes) reprojection overlaps for its own windows, sums the owners' counts with one small
     all-reduce, and takes the graph decisions (host only, replicated).
  (world == 1 keeps the sequential replay of TrackFrontend.track_many, which reads the stored depth like the reference;
  CUT3R_SCAN=1 forces the scan form, whose results are bit-identical for every world size.)

Pipelining: the replay of step s runs on a side HIP stream while the encoder graph of step s+1 executes on the main
stream; the decoder graph of step s+1 is launched after the replay has been issued (a graph with parallel branches
delays anything queued behind it on a shared hardware queue).  `flush()` drains the last replay.
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
        if t.is_cuda and dist.get_backend() == "nccl":
            buf = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(buf.view(-1), t.view(-1))       # one RCCL all-gather per tensor
        else:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)                                   # gloo (CPU tests; functional multi-rank runs on one GPU)
            buf = torch.stack(parts, 0)
        res.append(buf.reshape((world * t.shape[0],) + tuple(t.shape[1:])))
    return res


def memory_plan(world: int, steps: int, wb: int = 28, H: int = 384, W: int = 512, win: int = 5, ds: int = 2, feat_dim: int = 1024, patch: int = 16,
                weights_bytes: int = 1_610_000_000, workspace_bytes: int = 0, frames_resident: int = 2000, frame_bytes: int = 480 * 640 * 3,
                hbm_bytes: int = 288 * 10**9):
    """HBM bytes PER RANK of a `world`-GPU run of `steps` (timed + warm-up) steps of `wb` windows per rank, as bench.py sets the job up
    (every rank holds the REPLICATED keyframe store of the whole job: the covisibility tests of a window read every earlier stride-2
    pointmap of its sequence; images and depths are allocated for every keyframe although a rank only fills its own).  Returns a dict
    of the stores, their sum, and whether it fits."""
    n_kf = 7 + win * wb * world * steps + 2 + 8
    nsub = n_kf // win + 1
    h, w = H // ds, W // ds
    tokens = (H // patch) * (W // patch)
    feat_rows = max(64, 2 * (win * wb * world + 1)) + 6
    plan = {
        "keyframes": n_kf,
        "image_u8": n_kf * 3 * H * W,
        "depth_f32": n_kf * H * W * 4,
        "submap_ds_f32": nsub * (win + 1) * h * w * 3 * 4,
        "conf_ds_f32": nsub * (win + 1) * h * w * 4,
        "w2c_f32": n_kf * 12 * 4,
        "encoder_feature_ring_f32": feat_rows * tokens * feat_dim * 4,
        "camera_frames_u8": frames_resident * frame_bytes,
        "weights_f16": int(weights_bytes),
        "network_workspaces": int(workspace_bytes),
    }
    plan["total"] = sum(v for k, v in plan.items() if k != "keyframes")
    plan["hbm"] = int(hbm_bytes)
    plan["fits"] = plan["total"] <= 0.9 * hbm_bytes
    return plan


class PinnedFrames:
    """A camera recording in PINNED HOST memory, delivered to the GPU the way a live system receives it: the reference uploads every
    frame it touches (/root/reference/hislam2/motion_filter.py:78,91 `.to('cuda')`, track_frontend.py:48).  `fetch(f)` issues an
    asynchronous host-to-device copy of frame `f % period` on this object's own HIP stream into a device staging ring and returns the
    staged [1,H0,W0,3] u8 tensor; whatever consumes it must run on `stream` (ShardedTracker does the keyframe's resize there: copy and consumer are then
    in stream order, which is also what makes a small ring safe) or wait for `event()`.  `upload_every_frame=True` also pushes the frames nobody reads (non-keyframes of a fixed-cadence stream) over the
    link: the reference's motion filter uploads every tested frame."""

    def __init__(self, host_frames: torch.Tensor, virtual_len: int, device, ring: int = 16, upload_every_frame: bool = False):
        assert not host_frames.is_cuda and host_frames.dtype == torch.uint8 and host_frames.dim() == 4
        self.base = host_frames if host_frames.is_pinned() else host_frames.pin_memory()
        self.period = self.base.shape[0]
        self.shape = (int(virtual_len),) + tuple(self.base.shape[1:])
        self.device, self.dtype, self.is_cuda = torch.device(device), torch.uint8, False
        self.stream = torch.cuda.Stream(device=self.device)
        self.ring = torch.empty((int(ring),) + tuple(self.base.shape[1:]), dtype=torch.uint8, device=self.device)
        self._slot = 0
        self.upload_every_frame = bool(upload_every_frame)
        self._uploaded_upto = 0
        self._scratch = None
        self.bytes_uploaded = 0

    def fetch(self, f: int) -> torch.Tensor:
        """frame f -> its staging slot, copied on `self.stream` (call inside `torch.cuda.stream(self.stream)` or wait for event())"""
        slot = self._slot
        self._slot = (slot + 1) % self.ring.shape[0]
        dst = self.ring[slot:slot + 1]
        with torch.cuda.stream(self.stream):
            dst.copy_(self.base[int(f) % self.period][None], non_blocking=True)
        self.bytes_uploaded += dst.numel()
        return dst

    def upload_range(self, a: int, b: int, keep: Callable[[int], bool]):
        """the every-frame operating point: frames a..b-1 that `keep` does not claim (they are fetched by their consumer) cross the link
        too, into a scratch slot, and are dropped -- nothing reads a non-keyframe of a fixed-cadence stream"""
        if not self.upload_every_frame:
            return
        # consecutive unclaimed frames are consecutive in the pinned recording: ONE asynchronous copy per run (a fixed-cadence stream has runs of
        # kf_every - 1 frames; 1260 single-frame copies per step cost the issuing thread ~25 ms of a 246-ms step), into a scratch buffer nobody reads
        if self._scratch is None:
            self._scratch = torch.empty((32,) + tuple(self.base.shape[1:]), dtype=torch.uint8, device=self.device)
        cap = self._scratch.shape[0]
        with torch.cuda.stream(self.stream):
            f = max(a, self._uploaded_upto)
            while f < b:
                if keep(f):
                    f += 1
                    continue
                g = f + 1                        # the run [f, g): unclaimed, inside one period of the recording, at most `cap` frames
                while g < b and not keep(g) and g - f < cap and (g % self.period) > (f % self.period):
                    g += 1
                src = self.base[f % self.period:f % self.period + (g - f)]
                self._scratch[:g - f].copy_(src, non_blocking=True)
                self.bytes_uploaded += src.numel()
                f = g
        self._uploaded_upto = max(self._uploaded_upto, b)

    def event(self) -> torch.cuda.Event:
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def __getitem__(self, key):           # untimed paths (prologue): a synchronous slice on the device
        if isinstance(key, slice):
            a = 0 if key.start is None else key.start
            b = self.shape[0] if key.stop is None else key.stop
            return torch.stack([self.base[f % self.period] for f in range(a, b)], 0).to(self.device)
        return self.base[int(key) % self.period].to(self.device)


class ShardedTracker:
    """Drives a replicated `Cut3rSlam` with window-sharded, batched, pipelined network inference."""

    def __init__(self, slam, world: int, rank: int, wb: int = 1, infer_fn: Callable = None, track_fn: Callable = None,
                 append_fn: Callable = None, pipelined: bool = True, views: int = 6, force_collective: bool = False):
        self.slam, self.world, self.rank, self.wb = slam, world, rank, max(1, int(wb))
        self.infer_fn = infer_fn          # None: the real network (encoder graph, replay of the previous step, decoder graph)
        self.track_fn = track_fn          # None: TrackFrontend.track_many over all windows of the step (one round trip each)
        self.append_fn = append_fn or self._append
        self.pipelined = pipelined
        self.views = views
        self.force_collective = force_collective    # world == 1 rehearsal of the RCCL exchange
        self.emulate_gather = False
        self.shard_counting = os.environ.get("CUT3R_SHARD_COUNTING", "1") == "1"
        self._pending = None
        self._next_t0 = None            # first keyframe of the next window to be scheduled
        self._side = None
        self._pose_pinned = [None, None]
        self._appended_upto = 0
        self._enc_stream, self._ahead = None, None        # look-ahead encoder pass: (ranges, features, event)
        self.encode_ahead = os.environ.get("CUT3R_ENC_AHEAD", "1") == "1"
        self.side_priority = int(os.environ.get("CUT3R_SIDE_PRIORITY", "0"))         # 0: measured best; -1 (high) starves the network pass once the host runs ahead
        # scan form of the replay (TrackFrontend.track_sharded): every rank stores and counts only ITS windows; scalars, the
        # stride-2 stores and the counts are exchanged.  Always on with more than one rank; CUT3R_SCAN=1 forces it for one rank
        # (bit-identical to the multi-rank result by construction: same scalars, same host scan)
        self.scan = world > 1 or force_collective or os.environ.get("CUT3R_SCAN", "0") == "1"
        self.replicate_depth = os.environ.get("CUT3R_REPLICATE_DEPTH", "0") == "1"
        self._chain = None
        self._f0 = 0
        self._upload_ev = None            # frames from pinned host memory (PinnedFrames): the event behind the last upload + resize
        slam.tracked_only = True          # trajectory writers stop at the last TRACKED keyframe (the look-ahead registers more)
        self.stats = {"append_s": 0.0, "issue_s": 0.0, "issue_enc_s": 0.0, "replay_s": 0.0, "replay_wait_s": 0.0, "exchange_s": 0.0, "steps": 0}     # host wall-clock per phase

    def frames_needed(self, total_steps: int, kf_every: int, win: int) -> int:
        """frames consumed by the 7-keyframe initialisation plus `total_steps` steps"""
        return (7 + win * self.world * self.wb * total_steps + 1) * kf_every + 1

    # ---- default callbacks on the real SLAM objects
    def _append(self, kf_index: int, frame, tstamp, intr, mine: bool):
        # fixed cadence: the encoder pass of a keyframe is deferred to the rank that owns its window; the other ranks register the
        # keyframe (time stamp, calibration, counter) without copying pixels they never read (at 8 GPUs: 7 of 8 keyframes)
        kf = self.slam.keyframes
        if mine and frame is not None and frame.dim() == 4 and frame.shape[-1] == 3 and frame.shape[1] != 3:
            # a raw camera frame u8 [1,H0,W0,3] (cv2.imread layout, e.g. 480x640): demo_s.py:69-73 resizes it to the tracking resolution
            # with cv2.resize; here the keyframe is resized on the GPU straight into its row of the store (non-keyframes of a
            # fixed-cadence stream are never read)
            from . import ops
            i = kf.counter.value
            kf.append(tstamp, None, None, None, None, None, intr, None, None)
            ops.resize_linear_u8(frame[0], kf.ht, kf.wd, chw_out=True, out=kf.image[i])
            return
        kf.append(tstamp, frame[0] if (mine and frame is not None) else None, None, None, None, None, intr, None, None)

    def _encode(self, ranges):
        """encoder pass over this rank's not-yet-encoded keyframes (batched) -> window features [wb,V,N,E]"""
        tr = self.slam.tracker
        if self.world > 1:
            # keep the encoder batch shape the same in every step (one captured graph): with more than one rank the first
            # keyframe of a rank's range is usually another rank's, so it is (re-)encoded always (bit-identical features)
            self.slam.keyframes.feat_valid[ranges[0][0]] = False
        tr.window_features(ranges[0][0], ranges[-1][1])
        return torch.stack([tr.window_features(a, b) for a, b in ranges], 0)

    def _decode(self, feats):
        """batched decoder + heads over this rank's windows -> (pts [wb*V,H,W,3], conf [wb*V,H,W], pose [wb*V,7])"""
        kf = self.slam.keyframes
        res = self.slam.model.decode_windows(feats, kf.ht, kf.wd)
        return tuple(res[k] for k in ("pts3d_in_self_view", "conf_self", "camera_pose"))

    # ---- the pipeline
    def _side_ctx(self, gathered, ev):
        if ev is None or not self.pipelined:
            return contextlib.nullcontext()
        if self._side is None:
            self._side = self._make_side_stream()
        return torch.cuda.stream(self._side)

    def _make_side_stream(self):
        # HIP deals streams onto the hardware queues round-robin; CUT3R_SIDE_SKIP throwaway streams shift the side stream to
        # another queue (tuning knob: a queue shared with a graph branch delays the first replay operation of every step)
        self._skipped = [torch.cuda.Stream() for _ in range(int(os.environ.get("CUT3R_SIDE_SKIP", "0")))]
        return torch.cuda.Stream(priority=self.side_priority)

    def _exchange_counts(self, counts):
        """sum of the owners' overlap counts over the ranks (host int32 tensor; rows a rank does not own are zero)"""
        if self.emulate_gather or not dist.is_initialized():
            return counts
        if dist.get_backend() == "nccl":
            dev = counts.cuda(non_blocking=True)
            dist.all_reduce(dev, op=dist.ReduceOp.SUM)
            return dev.cpu()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        return counts

    def _replay(self, pending, defer=False):
        """chain + store every window of `pending`; returns the deferred host-only decisions (multi-rank with defer) or None"""
        fin = None
        ranges_all, gathered, ev = pending
        V = self.views
        tic = time.perf_counter()
        if ev is not None and self.pipelined:
            if self._side is None:
                self._side = self._make_side_stream()
            self._side.wait_event(ev)
            ev.synchronize()                        # pinned pose copy complete (issued a whole network pass ago)
            self.stats["replay_wait_s"] += time.perf_counter() - tic
            for g in (gathered if not self.scan or self.track_fn is not None else [t for o in gathered[0] for t in o]):
                if torch.is_tensor(g) and g.is_cuda:
                    g.record_stream(self._side)     # allocated on the main stream, consumed on the side stream
            ctx = torch.cuda.stream(self._side)
        else:
            if ev is not None:
                ev.synchronize()
                self.stats["replay_wait_s"] += time.perf_counter() - tic
            ctx = contextlib.nullcontext()
        with ctx:
            if self.scan and self.track_fn is None:
                fin = self._replay_scan(ranges_all, gathered, defer)
                self.slam.tracker.t1 = ranges_all[-1][1]
                self.stats["replay_s"] += time.perf_counter() - tic
                return fin
            outs = [tuple(g[j * V:(j + 1) * V] for g in gathered) for j in range(len(ranges_all))]
            if self.track_fn is not None:
                for (a, b), o in zip(ranges_all, outs):
                    self.track_fn(a, b, o)
                    self.slam.tracker.t1 = b
            else:
                mask, exch = None, None
                if (self.world > 1 or self.force_collective) and self.shard_counting:
                    # the O(#keyframes) overlap counting of a window runs on its owner only; one small all-reduce per step
                    mask = [self.rank * self.wb <= j < (self.rank + 1) * self.wb for j in range(len(ranges_all))]
                    exch = self._exchange_counts
                fin = self.slam.tracker.track_many(ranges_all, outs, count_mask=mask, exchange=exch, defer_decisions=defer)
                self.slam.tracker.t1 = ranges_all[-1][1]
        self.stats["replay_s"] += time.perf_counter() - tic
        return fin

    # ---- scan-form replay (multi-GPU)
    def _gather_rows(self, block, per_rank):
        """in-place all-gather of `block` [world * per_rank, ...] (this rank's rows already hold its own data)"""
        world, rank = self.world, self.rank
        if self.emulate_gather:
            own = block[rank * per_rank:(rank + 1) * per_rank]
            for r in range(world):                       # debug: stand-in traffic for the other ranks' rows
                if r != rank:
                    block[r * per_rank:(r + 1) * per_rank].copy_(own)
            return
        if not dist.is_initialized() or (world == 1 and not self.force_collective):
            return
        if block.is_cuda and dist.get_backend() == "nccl":
            own = block[rank * per_rank:(rank + 1) * per_rank]
            dist.all_gather_into_tensor(block.view(-1), own.reshape(-1))       # RCCL in-place form: own rows are the rank's chunk
        else:
            own = block[rank * per_rank:(rank + 1) * per_rank].cpu()
            parts = [torch.empty_like(own) for _ in range(world)]
            dist.all_gather(parts, own)                                        # gloo (CPU tests; two ranks on one GPU)
            for r in range(world):
                if r != rank:
                    block[r * per_rank:(r + 1) * per_rank].copy_(parts[r])

    def _gather_store(self, sub0, n, phase):
        kf = self.slam.keyframes
        if phase == 0:
            self._gather_rows(kf.submap_ds[sub0:sub0 + n], self.wb)
            self._gather_rows(kf.conf_ds[sub0:sub0 + n], self.wb)
        elif self.replicate_depth:                       # rows of views 0..4 of every window (a keyframe's final depth)
            self._gather_rows(kf.depth[sub0 * 5:(sub0 + n) * 5], self.wb * 5)

    def _replay_scan(self, ranges_all, payload, defer):
        own_outs, scal = payload
        tr = self.slam.tracker
        if self._chain is None:
            self._chain = tr.chain_state(ranges_all[0][0])
        i0 = self.rank * self.wb
        return tr.track_sharded(ranges_all, (i0, i0 + self.wb), own_outs, scal.numpy(), self._chain, self._gather_store,
                                self._exchange_counts, defer_decisions=defer)

    def _exchange_scalars(self, scal_own):
        """[wb, 44] fp64 device -> [world * wb, 44] (rank order = window order)"""
        world = self.world
        if self.emulate_gather:
            return torch.cat([scal_own] * world, 0)
        if not dist.is_initialized() or (world == 1 and not self.force_collective):
            return scal_own
        if scal_own.is_cuda and dist.get_backend() == "nccl":
            buf = torch.empty((world,) + tuple(scal_own.shape), dtype=scal_own.dtype, device=scal_own.device)
            dist.all_gather_into_tensor(buf.view(-1), scal_own.reshape(-1))
            return buf.view(-1, scal_own.shape[1])
        host = scal_own.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        return torch.cat(parts, 0).to(scal_own.device)

    def _append_range(self, frames, t, n_frames, kf_every, win, intr, first_t0):
        """register the keyframes among frames t..t+n_frames-1 (once: a look-ahead may already have done it)"""
        slam, world, rank, wb = self.slam, self.world, self.rank, self.wb
        host = isinstance(frames, PinnedFrames)
        if host:
            # the recording lives in pinned host memory: a keyframe's pixels cross the link by an asynchronous copy on the upload stream
            # and are resized there, straight into the keyframe store; the encoder pass that reads them waits for `_upload_ev`
            frames.upload_range(max(t, self._appended_upto), t + n_frames, lambda f: f % kf_every == 0)
        ctx = torch.cuda.stream(frames.stream) if host else contextlib.nullcontext()
        with ctx:
            self._append_frames(frames, t, n_frames, kf_every, win, intr, host)
        if host:
            self._upload_ev = frames.event()
        self._appended_upto = max(self._appended_upto, t + n_frames)

    def _append_frames(self, frames, t, n_frames, kf_every, win, intr, host):
        slam, world, rank, wb = self.slam, self.world, self.rank, self.wb
        for f in range(max(t, self._appended_upto), t + n_frames):
            if f % kf_every == 0:
                k = slam.keyframes.counter.value
                # which ranks read this keyframe's pixels: the owner of its window (windows are dealt to the ranks in blocks of wb,
                # step after step: keyframe F0 + o lies in window (o - 1) // win), and -- when it is the last keyframe of a rank's
                # block -- also the next rank (of this or the next step), whose first window starts with it.  Keyframes are appended
                # up to two ahead of the step that tracks them, so the rule is absolute, not relative to the step being issued.
                o = k - self._f0
                blk = (o - 1) // (win * wb) if o >= 1 else 0
                mine = blk % world == rank or (o >= 1 and o % (win * wb) == 0 and (blk + 1) % world == rank)
                self.append_fn(k, (frames.fetch(f) if host else frames[f:f + 1]) if mine else None, f, intr, mine)

    def step(self, frames, t, kf_every, win, intr):
        """Advance world*wb windows (= world*wb*win*kf_every frames).  Returns the new frame counter."""
        slam, world, rank, wb = self.slam, self.world, self.rank, self.wb
        if self._next_t0 is None:
            self._next_t0 = slam.tracker.t1 - 1
            self._f0 = self._next_t0          # first keyframe of the first window this driver schedules
        first_t0 = self._next_t0
        ranges_all = window_ranges(first_t0, world * wb, win)
        mine = ranges_all[rank * wb:(rank + 1) * wb]
        # 1. keyframe filter in fixed-cadence mode: every kf_every-th frame is a keyframe (motion_filter.py:83,109,124);
        #    every rank registers all of them, only the owner of a window ever encodes them
        n_frames = world * wb * win * kf_every
        tic0 = time.perf_counter()
        self._append_range(frames, t, n_frames, kf_every, win, intr, first_t0)
        tic = time.perf_counter()
        self.stats["append_s"] += tic - tic0
        if self.infer_fn is not None:
            # 2'. (test doubles) whole network pass, then the replay of the previous step
            outs = self.infer_fn(mine)
            self.stats["issue_s"] += time.perf_counter() - tic
            if self._pending is not None:
                self._replay(self._pending)
                self._pending = None
        else:
            # 2. encoder features of this rank's windows: from the look-ahead pass issued during the previous step (its own
            #    stream, beside that step's decoder), else encoded now on the main stream
            if self._ahead is not None and self._ahead[0] == mine:
                _, feats, ev_enc = self._ahead
                torch.cuda.current_stream().wait_event(ev_enc)
                feats.record_stream(torch.cuda.current_stream())
            else:
                if self._ahead is not None:      # a look-ahead pass for other windows may still be writing the feature store
                    torch.cuda.current_stream().wait_event(self._ahead[2])
                if self._upload_ev is not None:  # host frames: this step's keyframes were uploaded + resized on the upload stream
                    torch.cuda.current_stream().wait_event(self._upload_ev)
                feats = self._encode(mine)
            self._ahead = None
            self.stats["issue_s"] += time.perf_counter() - tic
            self.stats["issue_enc_s"] += time.perf_counter() - tic
            # 2b. look-ahead: register the next step's keyframes and start THEIR encoder pass on the encoder stream; it runs
            #     beside this step's decoder (large efficient GEMMs filling the gaps of the decoder's mid-size kernels)
            t_next = t + n_frames
            if self.encode_ahead and self.pipelined and t_next + n_frames <= frames.shape[0] and feats.is_cuda:
                nxt_first = ranges_all[-1][1] - 1
                nxt_all = window_ranges(nxt_first, world * wb, win)
                nxt_mine = nxt_all[rank * wb:(rank + 1) * wb]
                self._append_range(frames, t_next, n_frames, kf_every, win, intr, nxt_first)
                if self._enc_stream is None:
                    self._enc_stream = torch.cuda.Stream()      # default priority: raising either side's priority measured 15-20 % slower
                self._enc_stream.wait_stream(torch.cuda.current_stream())      # the keyframe images were copied on this stream
                if self._upload_ev is not None:                                # ... or, from pinned host memory, on the upload stream
                    self._enc_stream.wait_event(self._upload_ev)
                with torch.cuda.stream(self._enc_stream):
                    nf = self._encode(nxt_mine)
                    ev_enc = torch.cuda.Event()
                    ev_enc.record()
                self._ahead = (nxt_mine, nf, ev_enc)
            # 3. replay of the previous step's chaining + graph update (side stream) while the encoder runs.  It is
            #    issued BEFORE the decoder graph on purpose: that graph has parallel branches on several hardware queues and
            #    anything queued behind it on a shared queue waits for the whole branch (measured: ~40 ms per step)
            fin = None
            if self._pending is not None:
                fin = self._replay(self._pending, defer=True)
                self._pending = None
            # 4'. recurrent decoder + heads graph
            tic = time.perf_counter()
            outs = self._decode(feats)
            self.stats["issue_s"] += time.perf_counter() - tic
            if fin is not None:      # multi-rank: the host-only graph decisions of the replayed step, now that the GPU has its work
                tic = time.perf_counter()
                fin()
                self.stats["replay_s"] += time.perf_counter() - tic
        if self.scan and self.infer_fn is None:
            # 4s. scan form: the three outputs stay on this rank (private copies: the graph's buffers are rewritten by the next
            #     pass); only [wb, 44] fp64 scalars per rank cross the links now, the stores follow in the replay
            tic = time.perf_counter()
            V = self.views
            own = [tuple(o[j * V:(j + 1) * V].clone() for o in outs) for j in range(wb)]
            scal = self._exchange_scalars(slam.tracker.window_scalars(own))
            slot = self.stats["steps"] & 1
            if self._pose_pinned[slot] is None or self._pose_pinned[slot].shape != scal.shape or self._pose_pinned[slot].dtype != scal.dtype:
                self._pose_pinned[slot] = torch.empty(tuple(scal.shape), dtype=scal.dtype).pin_memory()
            scal_host = self._pose_pinned[slot]
            scal_host.copy_(scal, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.stats["exchange_s"] += time.perf_counter() - tic
            self.stats["steps"] += 1
            self._pending = (ranges_all, (own, scal_host), ev)
            self._next_t0 = ranges_all[-1][1] - 1
            if not self.pipelined:
                self.flush()
            return t + n_frames
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
