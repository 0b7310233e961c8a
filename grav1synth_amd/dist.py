"""Frame-shard mode: one process per GPU, one exchange, one ordered fold.

The reference's `diff` loop is strictly serial (src/main.rs:432-521), but
everything pixel-sized in it is per-frame independent.  Rank r runs the HIP
kernels over its contiguous chunk of frames and keeps only the per-frame
integer records; ONE all-gather (RCCL over xGMI with the "nccl" backend, gloo
in the CPU tests) moves them to every rank, and rank 0 replays the sequential
noise-model update over the records in global frame order.  Records are exact
integers, so the result does not depend on the number of ranks.
"""
from __future__ import annotations

import os
import queue
import threading
from fractions import Fraction
from typing import List, Optional

import numpy as np
import torch

from .diff import DiffGenerator, GrainTableSegment, RecordFold


def gather_records(records: np.ndarray, dist, device: Optional[torch.device] = None) -> List[np.ndarray]:
    """All-gather each rank's [n_r, record_size] uint8 records.  Returns the
    per-rank arrays in rank order (on every rank).  One collective for the
    payload (+ one tiny one for the frame counts)."""
    world = dist.get_world_size()
    backend = dist.get_backend()
    dev = device if (backend == "nccl" and device is not None) else torch.device("cpu")
    n_local = int(records.shape[0])
    rs = int(records.shape[1]) if records.ndim == 2 else 0
    meta = torch.tensor([n_local, rs], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    counts = [int(m[0].item()) for m in metas]
    rs = max(int(m[1].item()) for m in metas)
    nmax = max(counts)
    pad = torch.zeros((nmax, rs), dtype=torch.uint8)
    if n_local:
        pad[:n_local] = torch.from_numpy(np.ascontiguousarray(records))
    pad = pad.to(dev)
    if backend == "nccl":  # RCCL: one flat all-gather over xGMI
        out = torch.empty((world, nmax, rs), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out, pad)
        host = out.cpu().numpy()
    else:  # gloo (CPU tests)
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        host = torch.stack(parts).numpy()
    return [host[r, : counts[r]] for r in range(world)]


def fold_records(per_rank: List[np.ndarray], fps, ar_coeff_lag: int = 3) -> List[GrainTableSegment]:
    """The ordered fold over all records, rank-major == global frame order for
    contiguous frame chunks."""
    fold = RecordFold(fps, ar_coeff_lag)
    for recs in per_rank:
        fold.push_many(recs)
    segs = fold.finish()
    fold.close()
    return segs


_GATHER_TO_ROOT_OK = True  # falls back to an all-gather if the backend refuses a rooted gather


def gather_msgs(msg: np.ndarray, dist, device: Optional[torch.device] = None) -> Optional[np.ndarray]:
    """The transport of a round: every rank's fixed-size message (g1s_shard_pack) to rank 0 -- [world, bytes] there, None
    elsewhere.  RCCL: a rooted gather (send/recv over xGMI, 1/N of an all-gather's traffic); gloo in the tests."""
    global _GATHER_TO_ROOT_OK
    world, rank, backend = dist.get_world_size(), dist.get_rank(), dist.get_backend()
    dev = device if (backend == "nccl" and device is not None) else torch.device("cpu")
    t = torch.from_numpy(msg).to(dev)
    host = None
    if _GATHER_TO_ROOT_OK:
        try:
            parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
            dist.gather(t, gather_list=parts, dst=0)
            if rank == 0:
                host = torch.stack(parts).cpu().numpy()
        except (RuntimeError, NotImplementedError):  # raised on every rank alike, before any traffic
            _GATHER_TO_ROOT_OK = False
    if not _GATHER_TO_ROOT_OK:
        if backend == "nccl":
            out = torch.empty((world, t.numel()), dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(out, t)
            host = out.cpu().numpy() if rank == 0 else None
        else:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            host = torch.stack(parts).numpy() if rank == 0 else None
    return host


_RCCL_ROUNDS = {}  # (group identity, device, message bytes) -> _RcclRounds; a transport keeps its group alive (self.dist), so an
#                     id() in a key cannot be handed to another group while the entry stands


def release_transports() -> None:
    """frees the pinned rings and streams of every cached RCCL transport (a process that is done with its process groups)"""
    for t in _RCCL_ROUNDS.values():
        t.close()
    _RCCL_ROUNDS.clear()



class _RcclRounds:
    """The transport of the rounds over RCCL, with nothing allocated and nothing waited for on the feeding thread: the rank's
    message is packed into one of a ring of pinned buffers, copied to the device on a copy stream and gathered to rank 0 (send /
    recv over xGMI) on a collective stream; on rank 0 the gathered rows go to one of a ring of pinned host buffers and the MERGER
    thread waits for that copy.  The feeding thread waits only when the collective stream is RING rounds behind.
    (The first form of this class' job -- gather_msgs per round -- allocated, copied through pageable memory and ended in a
    blocking `.cpu()` of world x 1.7 MB on rank 0's feeding thread: comparable to a batch's kernel time at eight ranks.  The
    second had one stream: the next round's message waited behind the last round's gather kernel, which itself waits for a
    free workgroup slot next to the accumulation launches: 0.4 - 0.6 ms a round.)"""

    RING = 4

    def __init__(self, dist, device: torch.device, msg_bytes: int):
        self.dist = dist
        self.dev = device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.copy_stream = torch.cuda.Stream(device=device)
        self.coll_stream = torch.cuda.Stream(device=device)
        self.send_host = [torch.zeros(msg_bytes, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        self.send_dev = [torch.zeros(msg_bytes, dtype=torch.uint8, device=device) for _ in range(self.RING)]
        self.gathered = [None] * self.RING  # event: the gather that read send_dev[i] has run
        self.round = 0
        # gather-to-root or all-gather: decided ONCE, here, by every rank together (a one-byte probe, then a barrier) -- never
        # under an `except` in the round loop, where an error on one rank would leave it issuing a different collective from
        # its peers.  An exception in a later round is fatal.
        self.rooted = True
        try:
            probe = torch.zeros(1, dtype=torch.uint8, device=device)
            parts = [torch.zeros(1, dtype=torch.uint8, device=device) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(probe, gather_list=parts, dst=0)
        except (RuntimeError, NotImplementedError):
            self.rooted = False
        flag = torch.tensor([1 if self.rooted else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # (every rank rooted, or none)
        self.rooted = bool(int(flag.item()))
        self.recv_dev = None
        self.recv_host = []
        self.closed = False
        if self.rank == 0 or not self.rooted:
            self.recv_dev = torch.zeros((self.world, msg_bytes), dtype=torch.uint8, device=device)
            self.recv_host = [torch.zeros((self.world, msg_bytes), dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
            self.free = queue.Queue()
            for k in range(self.RING):
                self.free.put(k)

    def pack_buffer(self) -> np.ndarray:
        """where this round's message is packed (the round RING rounds ago has left it)"""
        if self.closed:
            raise RuntimeError("this RCCL transport was released (release_transports): make a new StreamingShardedDiff")
        i = self.round % self.RING
        if self.gathered[i] is not None:
            self.gathered[i].synchronize()
        return self.send_host[i].numpy()

    def exchange(self):
        """the packed message -> rank 0.  Returns (ring index, event) on rank 0 -- the rows are in recv_host[index] once the
        event has passed; release(index) hands the buffer back -- and None elsewhere."""
        if self.closed:
            raise RuntimeError("this RCCL transport was released (release_transports): make a new StreamingShardedDiff")
        i = self.round % self.RING
        self.round += 1
        with torch.cuda.stream(self.copy_stream):
            self.send_dev[i].copy_(self.send_host[i], non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.copy_stream)
        out = None
        with torch.cuda.stream(self.coll_stream):
            self.coll_stream.wait_event(copied)
            if self.rooted:
                parts = [self.recv_dev[r] for r in range(self.world)] if self.rank == 0 else None
                self.dist.gather(self.send_dev[i], gather_list=parts, dst=0)
            else:
                self.dist.all_gather_into_tensor(self.recv_dev, self.send_dev[i])
            done = torch.cuda.Event()
            done.record(self.coll_stream)
            self.gathered[i] = done
            if self.rank == 0:
                k = self.free.get()  # (blocks only if the merger is RING rounds behind)
                self.recv_host[k].copy_(self.recv_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.coll_stream)
                out = (k, ev)
        return out

    def rows(self, k: int) -> np.ndarray:
        return self.recv_host[k].numpy()

    def release(self, k: int) -> None:
        self.free.put(k)

    def close(self) -> None:
        """frees the pinned rings (the cache below holds one transport per process group and message size)"""
        if self.closed:
            return
        self.closed = True  # (a generator that still holds this transport gets an error, not an IndexError, from its next round)
        for st in (self.copy_stream, self.coll_stream):
            st.synchronize()
        self.send_host, self.send_dev, self.recv_host = [], [], []
        self.gathered = [None] * self.RING
        self.recv_dev = None
        self.copy_stream = self.coll_stream = None  # (torch frees a stream when its last reference goes)


class StreamingShardedDiff:
    """Frame-shard mode with the fold streamed: the video is dealt to the ranks batch by batch
    (global batch j goes to rank j % N), every rank runs the kernels AND the per-frame half of the
    fold on its batches, and after each batch ONE small all-gather (a latest state is ~27 KB a frame)
    brings the round's states to rank 0, which merges them in global frame order while the GPUs are
    already on the next batch.  The ordered merge (3-5 us a frame, on its own thread) is all that stays serial.

    Every rank makes the same number of calls: diff_prepared for a batch of `batch_frames` frames (the video's last
    batch may be short), idle_round when a round has no batch for it."""

    def __init__(self, fps, source_bit_depth: int, denoised_bit_depth: int, *, ar_coeff_lag: int = 3,
                 luma_only: bool = False, device: int = -1, batch_frames: int = 16, group=None):
        self.dist = group
        self.fps = Fraction(fps)
        self.lag = ar_coeff_lag
        self.device = device
        self.batch = batch_frames
        self.generator = DiffGenerator(fps, source_bit_depth, denoised_bit_depth, ar_coeff_lag=ar_coeff_lag,
                                       luma_only=luma_only, device=device, batch_frames=batch_frames,
                                       records_only=2 if group is not None else False)
        self._fold = None
        self._msg_bytes = 0
        if group is not None:
            self._msg_bytes = int(self.generator._L.g1s_shard_msg_size(ar_coeff_lag, batch_frames))
            if group.get_rank() == 0:
                self._fold = RecordFold(fps, ar_coeff_lag)
                self._merge_q = queue.Queue()
                self._merge_err = None
                self._merger = threading.Thread(target=self._merge_main, daemon=True)
                self._merger.start()
        self._dev = None
        if torch.cuda.is_available():
            self._dev = torch.device("cuda", device if device >= 0 else torch.cuda.current_device())
        self._rccl = None
        if group is not None and self._dev is not None and group.get_backend() == "nccl":
            # (the pinned rings and streams are made once per process and message size: pinning memory costs milliseconds)
            key = (id(group), str(self._dev), self._msg_bytes)
            if key not in _RCCL_ROUNDS:
                _RCCL_ROUNDS[key] = _RcclRounds(group, self._dev, self.ROUNDS_PER_GATHER * self._msg_bytes)
            self._rccl = _RCCL_ROUNDS[key]
        self._group_buf, self._group_n = None, 0
        self.exchange_s = 0.0  # seconds the feeding thread spent in the rounds' exchange (bench.py prints it)
        self.exchange_rounds = 0  # rounds this rank took part in: feeds, idle rounds, the flush rounds and the last group's padding

    # Rounds per gather: every rank packs one message a round (the library's protocol, unchanged), the transport moves
    # ROUNDS_PER_GATHER of them at a time.  A collective next to the accumulation launches costs the GPU ~0.2 ms whatever it
    # moves (its kernel waits for a workgroup slot and holds it): one per round took 11 % off a rank's throughput, one per four
    # rounds 3 % (tools/rccl_round_cost.py).  The merge orders by the batch index in the messages, so grouping changes nothing
    # but when the states reach rank 0.
    ROUNDS_PER_GATHER = int(os.environ.get("G1S_ROUNDS_PER_GATHER", "4"))

    def _exchange_one(self, flush: bool = False) -> None:
        """One round: this rank's message (g1s_shard_pack decides which batch goes out) into the group's buffer; every
        ROUNDS_PER_GATHER-th round the buffer goes to rank 0 in one gather and on to g1s_shard_merge there."""
        import time as _time

        t0 = _time.perf_counter()
        L = self.generator._L
        R, mb = self.ROUNDS_PER_GATHER, self._msg_bytes
        if self._group_buf is None:
            self._group_buf = self._rccl.pack_buffer() if self._rccl is not None else np.zeros(R * mb, dtype=np.uint8)
        slot = self._group_buf[self._group_n * mb:(self._group_n + 1) * mb]
        self.generator._check(L.g1s_shard_pack(self.generator._h, int(flush), slot.ctypes.data, mb))
        self._group_n += 1
        if self._group_n == R:
            self._group_n = 0
            buf, self._group_buf = self._group_buf, None
            if os.environ.get("G1S_ROUNDS_LOCAL") and self.dist.get_world_size() == 1:  # measurement: the rounds without a transport
                self._merge_q.put(buf.reshape(1, R * mb))
            elif self._rccl is not None:
                got = self._rccl.exchange()
                if self._fold is not None:
                    self._merge_q.put(got)  # (ring index, event): the merger waits for the copy, merges, hands the buffer back
            else:
                gathered = gather_msgs(buf, self.dist, self._dev)
                if self._fold is not None:
                    # the merge itself runs on a thread of its own (the C call drops the GIL): this thread goes back to feeding its GPU
                    self._merge_q.put(gathered)
        self.exchange_s += _time.perf_counter() - t0
        self.exchange_rounds += 1

    def _merge_main(self) -> None:
        L = self.generator._L
        while True:
            item = self._merge_q.get()
            if item is None:
                return
            ring = None
            if isinstance(item, tuple):  # RCCL rounds: the gathered rows are on their way into a pinned ring buffer
                ring, ev = item
                ev.synchronize()
                item = self._rccl.rows(ring)
            # item: [ranks, ROUNDS_PER_GATHER x message]: the messages of one round sit a row apart (the row index IS the rank: the
            # merge computes a batch's place in the video from it)
            for j in range(item.shape[1] // self._msg_bytes):
                if self._merge_err is not None:
                    break
                rc = L.g1s_shard_merge(self._fold._h, item.ctypes.data + j * self._msg_bytes, item.strides[0], item.shape[0])
                if rc:  # surfaces in finish()
                    from ._lib import G1SError
                    self._merge_err = G1SError(rc, L.g1s_fold_last_error(self._fold._h).decode())
            if ring is not None:
                self._rccl.release(ring)

    def diff_prepared(self, prepared, sync_torch: bool = True) -> None:
        """Feeds ONE batch (this rank's next batch in the global order)."""
        self.generator.diff_prepared(prepared, sync_torch=sync_torch)
        self._fed = getattr(self, "_fed", 0) + int(prepared.n)
        if self.dist is not None:
            # the states of an earlier batch (which one is a function of the call sequence only, so every
            # rank contributes the same batch index; nothing on the first calls)
            self._exchange_one()

    def idle_round(self) -> None:
        """A round in which this rank has no batch to feed (the video's batch count is not a multiple of the rank
        count: the last round is short): it still takes part in the round's exchange, contributing whatever of its
        earlier batches is ready -- every rank makes the same number of diff_prepared + idle_round calls."""
        if self.dist is not None:
            self._exchange_one()

    def finish(self) -> Optional[List[GrainTableSegment]]:
        if self.dist is None:
            return self.generator.finish()
        for _ in range(int(self.generator._L.g1s_shard_flush_rounds())):
            self._exchange_one(flush=True)
        while self._group_n:  # (the last group goes out full: empty messages behind the last states)
            self._exchange_one(flush=True)
        # every frame fed by any rank must have been merged: g1s_shard_flush_rounds() is what a generator's slots can hold, and a
        # change there must fail here, loudly, not drop the video's last batches
        if self.dist.get_world_size() > 1:
            # (a CPU tensor unless the backend only moves device memory: a gloo job without a GPU is checked too)
            on_dev = self._dev is not None and self.dist.get_backend() == "nccl"
            t = torch.tensor([getattr(self, "_fed", 0)], dtype=torch.int64, device=self._dev if on_dev else "cpu")
            self.dist.all_reduce(t)
            total = int(t.item())
        else:
            total = getattr(self, "_fed", 0)
        if self._fold is None:
            return None
        self._stop_merger()
        if self._merge_err is not None:
            raise self._merge_err
        merged = int(self.generator._L.g1s_fold_frames(self._fold._h))
        if total is not None and merged != total:
            raise RuntimeError(f"frame-shard job: {merged} of {total} frames merged after the flush rounds")
        segs = self._fold.finish()
        self._fold.close()
        self._fold = None
        return segs

    def _stop_merger(self) -> None:
        if getattr(self, "_merger", None) is not None:
            self._merge_q.put(None)
            self._merger.join()
            self._merger = None

    def close(self) -> None:
        self._stop_merger()
        self.generator.close()
        if self._fold is not None:
            self._fold.close()
            self._fold = None


class ShardedDiff:
    """DiffGenerator over a frame shard.  With `group=None` it is the plain
    single-GPU generator; with a torch.distributed module/group, each rank feeds
    ITS frames (rank r's frames precede rank r+1's in the video) and `finish()`
    returns the segments on rank 0 (None elsewhere)."""

    def __init__(self, fps, source_bit_depth: int, denoised_bit_depth: int, *, ar_coeff_lag: int = 3,
                 luma_only: bool = False, device: int = -1, batch_frames: int = 0, group=None):
        self.dist = group
        self.fps = Fraction(fps)
        self.lag = ar_coeff_lag
        self.device = device
        self.generator = DiffGenerator(fps, source_bit_depth, denoised_bit_depth, ar_coeff_lag=ar_coeff_lag,
                                       luma_only=luma_only, device=device, batch_frames=batch_frames,
                                       records_only=group is not None)
        self._shape = None
        self._nframes = 0
        self._luma_only = luma_only

    def diff_frame(self, source, denoised, xdec: int = 1, ydec: int = 1, sync_torch: bool = True) -> None:
        if self._shape is None:
            p0 = source[0] if not hasattr(source, "planes") else source.planes[0]
            npl = 1 if self._luma_only else (len(source) if not hasattr(source, "planes") else len(source.planes))
            self._shape = (int(p0.shape[1]), int(p0.shape[0]), npl)
        self.generator.diff_frame(source, denoised, xdec, ydec, sync_torch=sync_torch)
        self._nframes += 1

    def diff_prepared(self, prepared, width: int, height: int, nplanes: int, sync_torch: bool = True) -> None:
        if self._shape is None:
            self._shape = (width, height, 1 if self._luma_only else nplanes)
        self.generator.diff_prepared(prepared, sync_torch=sync_torch)
        self._nframes += prepared.n

    def finish(self) -> Optional[List[GrainTableSegment]]:
        if self.dist is None:
            return self.generator.finish()
        w, h, npl = self._shape if self._shape else (32, 32, 1)
        recs, n = self.generator.take_records(w, h, npl, self._nframes)
        dev = torch.device("cuda", self.device if self.device >= 0 else torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        per_rank = gather_records(recs, self.dist, dev)
        if self.dist.get_rank() != 0:
            return None
        return fold_records(per_rank, self.fps, self.lag)

    def close(self) -> None:
        self.generator.close()
