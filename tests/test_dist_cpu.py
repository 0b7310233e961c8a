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
