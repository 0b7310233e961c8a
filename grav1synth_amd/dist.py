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

    # batches that can still be inside a generator when the last frame has been queued: being filled,
    # pixel pass queued, accumulation queued, draining (csrc/engine.hip, kSlots)
    PIPELINE_BATCHES = 4

    def _exchange_one(self, flush: bool = False) -> None:
        """One fixed-size round.  The protocol is the library's (g1s_shard_pack: which batch goes out, the message;
        g1s_shard_merge: the root's order); this class only moves the bytes -- one gather to rank 0."""
        L = self.generator._L
        msg = np.zeros(self._msg_bytes, dtype=np.uint8)
        self.generator._check(L.g1s_shard_pack(self.generator._h, int(flush), msg.ctypes.data, msg.nbytes))
        gathered = gather_msgs(msg, self.dist, self._dev)
        if self._fold is not None:
            # the merge itself runs on a thread of its own (the C call drops the GIL): this thread goes back to feeding its GPU
            self._merge_q.put(gathered)

    def _merge_main(self) -> None:
        L = self.generator._L
        while True:
            item = self._merge_q.get()
            if item is None:
                return
            if self._merge_err is None:
                rc = L.g1s_shard_merge(self._fold._h, item.ctypes.data, item.strides[0], item.shape[0])
                if rc:  # surfaces in finish()
                    from ._lib import G1SError
                    self._merge_err = G1SError(rc, L.g1s_fold_last_error(self._fold._h).decode())

    def diff_prepared(self, prepared, sync_torch: bool = True) -> None:
        """Feeds ONE batch (this rank's next batch in the global order)."""
        self.generator.diff_prepared(prepared, sync_torch=sync_torch)
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
        for _ in range(self.PIPELINE_BATCHES):
            self._exchange_one(flush=True)
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
