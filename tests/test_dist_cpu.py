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
    assert msg_bytes == 24 + batch * bs
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


def _simulate_rounds(world, total_batches, batch, defer):
    """The round protocol of g1s_shard_pack / g1s_shard_merge with `world` simulated ranks in one process: global batch j is
    fed by rank j % world; in a feeding round a rank may send the oldest batch not sent yet among those it fed before its
    `defer` most recent feeds (3 by default: two in the pipeline and the one whose back half is not queued yet; 2 with
    G1S_NO_DEFER / per-kernel timing / one stream); the video's batch count
    need not be a multiple of `world`, so in the last round some ranks feed nothing and are one feed behind the others --
    they send an OLDER local batch in the same round.  Four flush rounds end the job.  Returns the merged table."""
    import ctypes as C

    from grav1synth_amd import _lib
    from grav1synth_amd.diff import RecordFold, format_tbl, latest_from_records, latest_size
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import np_pair, record_from_oracle
    from tests.oracle_binding import OracleDiff

    L = _lib.lib()
    spec = SynthSpec(256, 160, 8)
    bs = latest_size(3)
    msg_bytes = L.g1s_shard_msg_size(3, batch)
    blobs = {}
    for j in range(total_batches):
        recs = []
        for i in range(batch):
            o = OracleDiff(24, 1, 8, 8, 3, True)
            sp, dp = np_pair(spec, j * batch + i)
            o.diff_frame(sp, dp, 1, 1)
            recs.append(record_from_oracle(o, spec, 3, 3).buf)
        blobs[j] = latest_from_records(np.stack(recs), 3)
    fold = RecordFold(Fraction(24, 1), 3)
    fed = [0] * world
    sent = [0] * world
    order = []
    rounds = (total_batches + world - 1) // world

    def one_round(flush):
        msgs = np.zeros((world, msg_bytes), dtype=np.uint8)
        for r in range(world):
            limit = fed[r] if flush else max(0, fed[r] - defer)
            if sent[r] < limit:
                j = sent[r] * world + r
                b = blobs[j]
                assert L.g1s_shard_msg_from_latest_at(b.ctypes.data, len(b), 3, batch, sent[r], msgs[r].ctypes.data, msg_bytes) == 0
                order.append(j)
                sent[r] += 1
            else:
                assert L.g1s_shard_msg_from_latest_at(None, 0, 3, batch, 2 ** 64 - 1, msgs[r].ctypes.data, msg_bytes) == 0
        assert L.g1s_shard_merge(fold._h, msgs.ctypes.data, msgs.strides[0], world) == 0

    for k in range(rounds):
        for r in range(world):
            if k * world + r < total_batches:
                fed[r] += 1
        one_round(False)
    for _ in range(4):
        one_round(True)
    assert sent == fed and sum(sent) == total_batches
    return format_tbl(fold.finish()), order


@pytest.mark.parametrize("world,total_batches,defer", [(2, 7, 3), (2, 9, 3), (2, 5, 2), (3, 7, 3), (3, 10, 3), (3, 8, 2), (2, 6, 3), (2, 5, 0)])
def test_shard_rounds_merge_in_global_order_when_ranks_are_out_of_step(world, total_batches, defer):
    """with an idle rank in the short last round the ranks send different local batches in the same
    round (arrival order 0,2,1,4,3,... for 2 ranks and 7 batches); the merge goes by the batch index in the message."""
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import oracle_run

    batch = 2
    got, order = _simulate_rounds(world, total_batches, batch, defer)
    if (world, total_batches, defer) in {(2, 7, 3), (2, 9, 3), (2, 5, 2)}:  # (the cases ADVICE r02 lists)
        assert order != sorted(order), "the scenario is supposed to deliver batches out of order"
    want, _ = oracle_run(SynthSpec(256, 160, 8), range(total_batches * batch))
    assert got == want


def test_shard_merge_refuses_a_missing_or_repeated_batch():
    import ctypes as C

    from grav1synth_amd import _lib
    from grav1synth_amd._lib import G1SError
    from grav1synth_amd.diff import RecordFold, latest_from_records, latest_size
    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import np_pair, record_from_oracle
    from tests.oracle_binding import OracleDiff

    L = _lib.lib()
    spec = SynthSpec(256, 160, 8)
    o = OracleDiff(24, 1, 8, 8, 3, True)
    sp, dp = np_pair(spec, 0)
    o.diff_frame(sp, dp, 1, 1)
    blob = latest_from_records(np.stack([record_from_oracle(o, spec, 3, 3).buf]), 3)
    msg_bytes = L.g1s_shard_msg_size(3, 1)
    fold = RecordFold(Fraction(24, 1), 3)
    msgs = np.zeros((2, msg_bytes), dtype=np.uint8)
    # rank 1's batch 0 (global batch 1) arrives, rank 0 sends nothing: it waits
    assert L.g1s_shard_msg_from_latest_at(None, 0, 3, 1, 2 ** 64 - 1, msgs[0].ctypes.data, msg_bytes) == 0
    assert L.g1s_shard_msg_from_latest_at(blob.ctypes.data, 1, 3, 1, 0, msgs[1].ctypes.data, msg_bytes) == 0
    assert L.g1s_shard_merge(fold._h, msgs.ctypes.data, msgs.strides[0], 2) == 0
    # the same batch again: refused
    assert L.g1s_shard_merge(fold._h, msgs.ctypes.data, msgs.strides[0], 2) != 0
    assert b"twice" in L.g1s_fold_last_error(fold._h)
    # finishing while global batch 0 is missing: refused
    with pytest.raises(G1SError) as e:
        fold.finish()
    assert "never arrived" in str(e.value)
    fold.close()


def test_latest_blobs_equal_record_fold():
    """push_latest(latest_from_record(r)) == push(r), and a failing frame travels inside its blob (restored: ADVICE r02)."""
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
