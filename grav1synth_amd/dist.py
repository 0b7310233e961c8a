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


def gather_latest_round(blobs: np.ndarray, blob_size: int, max_frames: int, dist,
                        device: Optional[torch.device] = None) -> Optional[List[np.ndarray]]:
    """One round of the streaming exchange: every rank contributes the latest states of ONE batch
    ([n_r, blob_size], n_r <= max_frames; fixed-size message: [count | blobs]).  Only rank 0 merges, so
    the round is a gather to rank 0 (RCCL: send/recv over xGMI; 1/N of an all-gather's traffic).
    Returns the per-rank arrays in rank order on rank 0, None on the other ranks."""
    global _GATHER_TO_ROOT_OK
    world = dist.get_world_size()
    rank = dist.get_rank()
    backend = dist.get_backend()
    dev = device if (backend == "nccl" and device is not None) else torch.device("cpu")
    n_local = int(blobs.shape[0])
    msg = torch.zeros(16 + max_frames * blob_size, dtype=torch.uint8)
    msg[:8] = torch.from_numpy(np.array([n_local], dtype=np.int64).view(np.uint8))
    if n_local:
        msg[16 : 16 + n_local * blob_size] = torch.from_numpy(np.ascontiguousarray(blobs).reshape(-1))
    msg = msg.to(dev)
    host = None
    if _GATHER_TO_ROOT_OK:
        try:
            parts = [torch.empty_like(msg) for _ in range(world)] if rank == 0 else None
            dist.gather(msg, gather_list=parts, dst=0)
            if rank == 0:
                host = torch.stack(parts).cpu().numpy()
        except (RuntimeError, NotImplementedError):  # raised on every rank alike, before any traffic
            _GATHER_TO_ROOT_OK = False
    if not _GATHER_TO_ROOT_OK:
        if backend == "nccl":
            out = torch.empty((world, msg.numel()), dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(out, msg)
            host = out.cpu().numpy() if rank == 0 else None
        else:
            parts = [torch.empty_like(msg) for _ in range(world)]
            dist.all_gather(parts, msg)
            host = torch.stack(parts).numpy() if rank == 0 else None
    if host is None:
        return None
    res = []
    for r in range(world):
        n_r = int(host[r, :8].view(np.int64)[0])
        res.append(host[r, 16 : 16 + n_r * blob_size].reshape(n_r, blob_size))
    return res


class StreamingShardedDiff:
    """Frame-shard mode with the fold streamed: the video is dealt to the ranks batch by batch
    (global batch j goes to rank j % N), every rank runs the kernels AND the per-frame half of the
    fold on its batches, and after each batch ONE small all-gather (a latest state is ~27 KB a frame)
    brings the round's states to rank 0, which merges them in global frame order while the GPUs are
    already on the next batch.  The ordered merge (3-5 us a frame, on its own thread) is all that stays serial.

    Every rank must feed the same number of batches of `batch_frames` frames (the last may be short)."""

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
        self._blob = 0
        self._fold = None
        self._queue: List[np.ndarray] = []  # this rank's delivered, not yet exchanged batches
        if group is not None:
            from .diff import latest_size
            self._blob = latest_size(ar_coeff_lag)
            if group.get_rank() == 0:
                self._fold = RecordFold(fps, ar_coeff_lag)
                self._merge_q = queue.Queue()
                self._merge_err = None
                self._merger = threading.Thread(target=self._merge_main, daemon=True)
                self._merger.start()
        self._dev = None
        if torch.cuda.is_available():
            self._dev = torch.device("cuda", device if device >= 0 else torch.cuda.current_device())

    # batches that can still be inside a generator when the last frame has been queued: being filled,
    # pixel pass queued, accumulation queued, draining (csrc/engine.hip, kSlots)
    PIPELINE_BATCHES = 4

    def _exchange_one(self) -> None:
        """One fixed-size round: the next undelivered batch of every rank (possibly none)."""
        mine = self._queue.pop(0) if self._queue else np.zeros((0, self._blob), dtype=np.uint8)
        per_rank = gather_latest_round(mine, self._blob, self.batch, self.dist, self._dev)
        if self._fold is not None:
            # global order: batch by batch, ranks in order within a batch; the merge itself runs on a
            # thread of its own (the C call drops the GIL): this thread goes back to feeding its GPU
            blobs = [b for b in per_rank if len(b)]
            if blobs:
                self._merge_q.put(np.concatenate(blobs) if len(blobs) > 1 else blobs[0])

    def _merge_main(self) -> None:
        while True:
            item = self._merge_q.get()
            if item is None:
                return
            if self._merge_err is None:
                try:
                    self._fold.push_latest_many(item)
                except Exception as e:  # surfaces in finish()
                    self._merge_err = e

    def _collect(self, sync: bool) -> None:
        blobs = self.generator.take_latest(self.PIPELINE_BATCHES * self.batch, sync=sync)
        for k in range(0, len(blobs), self.batch):
            self._queue.append(blobs[k:k + self.batch])

    def diff_prepared(self, prepared, sync_torch: bool = True) -> None:
        """Feeds ONE batch (this rank's next batch in the global order)."""
        self.generator.diff_prepared(prepared, sync_torch=sync_torch)
        if self.dist is not None:
            # the states of an earlier batch (which one is a function of the call sequence only, so every
            # rank contributes the same batch index; nothing on the first calls)
            self._collect(sync=False)
            self._exchange_one()

    def finish(self) -> Optional[List[GrainTableSegment]]:
        if self.dist is None:
            return self.generator.finish()
        self._collect(sync=True)
        for _ in range(self.PIPELINE_BATCHES):
            self._exchange_one()
        assert not self._queue
        if self._fold is None:
            return None
        self._stop_merger()
        if self._merge_err is not None:
            raise self._merge_err
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
