"""CPU suite, part 3: the frame-shard path (N > 1) with world_size 2 over gloo.
Each rank owns a contiguous chunk of frames' records; one all-gather; rank 0
folds in global frame order and must reproduce the single-process table."""
import os
import sys
from fractions import Fraction

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nframes, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from grav1synth_amd.diff import format_tbl
    from grav1synth_amd.dist import fold_records, gather_records
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import np_pair, record_from_oracle
    from tests.oracle_binding import OracleDiff

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = SynthSpec(256, 160, 8)
    per = (nframes + world - 1) // world
    mine = range(rank * per, min(nframes, (rank + 1) * per))
    recs = []
    for k in mine:
        # stand-in for the GPU kernels: per-frame records from the oracle's exact integers.
        # Records are per-frame independent, so a fresh oracle per frame is equivalent.
        o = OracleDiff(24, 1, 8, 8, 3, True)
        s, d = np_pair(spec, k)
        o.diff_frame(s, d, 1, 1)
        recs.append(record_from_oracle(o, spec, 3, 3).buf)
    local = np.stack(recs) if recs else np.zeros((0, 0), np.uint8)
    per_rank = gather_records(local, dist)
    if rank == 0:
        segs = fold_records(per_rank, Fraction(24, 1), 3)
        with open(out_path, "wb") as f:
            f.write(format_tbl(segs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nframes", [5, 4])
def test_two_rank_shard_equals_single_process(tmp_path, nframes):
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import oracle_run

    out = str(tmp_path / "sharded.tbl")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, nframes, out), nprocs=2, join=True)
    want, _ = oracle_run(SynthSpec(256, 160, 8), range(nframes))
    assert open(out, "rb").read() == want


def _stream_worker(rank, world, port, total_batches, batch, out_path):
    """Streaming frame-shard fold over the library's round protocol: global batch j goes to rank j % world; per round every
    rank packs ONE message (g1s_shard_msg_from_latest here: the latest states come from oracle-made records; a GPU rank
    uses g1s_shard_pack), one gather to rank 0 (grav1synth_amd.dist.gather_msgs, gloo), g1s_shard_merge there.  The batch
    count is odd: in the last round rank 1 has nothing and sends an empty message."""
    sys.path.insert(0, ROOT)
    import ctypes as C

    import torch.distributed as dist

    from grav1synth_amd import _lib
    from grav1synth_amd.diff import RecordFold, format_tbl, latest_from_records, latest_size
    from grav1synth_amd.dist import gather_msgs
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import np_pair, record_from_oracle
    from tests.oracle_binding import OracleDiff

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = _lib.lib()
    spec = SynthSpec(256, 160, 8)
    fold = RecordFold(Fraction(24, 1), 3) if rank == 0 else None
    bs = latest_size(3)
    msg_bytes = L.g1s_shard_msg_size(3, batch)
    assert msg_bytes == 16 + batch * bs
    rounds = (total_batches + world - 1) // world
    for r in range(rounds):
        j = r * world + rank  # this rank's global batch in round r
        blobs = np.zeros((0, bs), dtype=np.uint8)
        if j < total_batches:
            recs = []
            for i in range(batch):
                o = OracleDiff(24, 1, 8, 8, 3, True)
                s, d = np_pair(spec, j * batch + i)
                o.diff_frame(s, d, 1, 1)
                recs.append(record_from_oracle(o, spec, 3, 3).buf)
            blobs = latest_from_records(np.stack(recs), 3)
            assert blobs.shape == (batch, bs)
        msg = np.zeros(msg_bytes, dtype=np.uint8)
        assert L.g1s_shard_msg_from_latest(blobs.ctypes.data if len(blobs) else None, len(blobs), 3, batch, msg.ctypes.data, msg.nbytes) == 0
        gathered = gather_msgs(msg, dist)
        if fold is not None:
            assert gathered.shape == (world, msg_bytes)
            assert L.g1s_shard_merge(fold._h, gathered.ctypes.data, gathered.strides[0], world) == 0
        else:
            assert gathered is None
    if rank == 0:
        bad = np.zeros((world, msg_bytes), dtype=np.uint8)  # no magic: refused, nothing merged
        assert L.g1s_shard_merge(fold._h, bad.ctypes.data, bad.strides[0], world) == -1
        with open(out_path, "wb") as f:
            f.write(format_tbl(fold.finish()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_streaming_fold_equals_single_process(tmp_path):
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import oracle_run

    total_batches, batch = 3, 2
    out = str(tmp_path / "streamed.tbl")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_stream_worker, args=(2, port, total_batches, batch, out), nprocs=2, join=True)
    want, _ = oracle_run(SynthSpec(256, 160, 8), range(total_batches * batch))
    assert open(out, "rb").read() == want
