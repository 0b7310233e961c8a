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


def _stream_worker(rank, world, port, nbatches, batch, out_path):
    """Streaming frame-shard fold: global batch j goes to rank j % world; per round ONE all-gather of
    latest states (the per-frame half of the fold done where the frame lives); rank 0 merges in order."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from grav1synth_amd.diff import RecordFold, format_tbl, latest_from_records, latest_size
    from grav1synth_amd.dist import gather_latest_round
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import np_pair, record_from_oracle
    from tests.oracle_binding import OracleDiff

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = SynthSpec(256, 160, 8)
    fold = RecordFold(Fraction(24, 1), 3) if rank == 0 else None
    bs = latest_size(3)
    for k in range(nbatches):  # this rank's k-th batch is global batch k * world + rank
        recs = []
        for i in range(batch):
            o = OracleDiff(24, 1, 8, 8, 3, True)
            s, d = np_pair(spec, (k * world + rank) * batch + i)
            o.diff_frame(s, d, 1, 1)
            recs.append(record_from_oracle(o, spec, 3, 3).buf)
        blobs = latest_from_records(np.stack(recs), 3)
        assert blobs.shape == (batch, bs)
        per_rank = gather_latest_round(blobs, bs, 2 * batch, dist)
        if fold is not None:
            for b in per_rank:
                fold.push_latest_many(b)
    if rank == 0:
        with open(out_path, "wb") as f:
            f.write(format_tbl(fold.finish()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_streaming_fold_equals_single_process(tmp_path):
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import oracle_run

    nbatches, batch = 2, 2
    out = str(tmp_path / "streamed.tbl")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_stream_worker, args=(2, port, nbatches, batch, out), nprocs=2, join=True)
    want, _ = oracle_run(SynthSpec(256, 160, 8), range(2 * nbatches * batch))
    assert open(out, "rb").read() == want


def test_latest_blobs_equal_record_fold():
    """push_latest(latest_from_record(r)) == push(r), and a failing frame travels inside its blob."""
    from grav1synth_amd.diff import G1SError, Record, RecordFold, format_tbl, latest_from_records
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import np_pair, record_from_oracle
    from tests.oracle_binding import OracleDiff

    spec = SynthSpec(256, 160, 8)
    recs = []
    for k in range(3):
        o = OracleDiff(24, 1, 8, 8, 3, True)
        s, d = np_pair(spec, k)
        o.diff_frame(s, d, 1, 1)
        recs.append(record_from_oracle(o, spec, 3, 3).buf)
    recs = np.stack(recs)
    a, b = RecordFold(Fraction(24, 1), 3), RecordFold(Fraction(24, 1), 3)
    a.push_many(recs)
    b.push_latest_many(latest_from_records(recs, 3))
    assert format_tbl(a.finish()) == format_tbl(b.finish())
    blank = Record.blank(256, 160, 1, 1, 3, 3).buf[None, :]  # no flat blocks
    c = RecordFold(Fraction(24, 1), 3)
    with pytest.raises(G1SError):
        c.push_latest_many(latest_from_records(blank, 3))
