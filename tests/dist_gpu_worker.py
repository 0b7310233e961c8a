"""python -m tests.dist_gpu_worker OUT.tbl -- one rank of a frame-shard job (tests/test_gpu_parity.py launches two of these on ONE
device with the gloo backend, the way G1S_BENCH_SHARE_GPU=1 runs bench.py --gpus N on a single-GPU box).

The job: G1S_TEST_BATCHES (default 5) batches of 2 frame pairs; with an odd count rank 1 sits out the last round (five batches:
rank 0: batches 0, 2, 4; rank 1: batches 1, 3 and an idle round) and is from then on one feed behind rank 0 -- with 7 or 9
batches the ranks then send different local batches in the same round and the root must order by the index the
messages carry.  A scene cut in the middle of the video, inside a batch.  Rank 0 writes the table."""
import os
import sys
from fractions import Fraction

import torch
import torch.distributed as dist

from grav1synth_amd.diff import DiffGenerator, format_tbl
from grav1synth_amd.dist import StreamingShardedDiff
from grav1synth_amd.synth import SynthSpec, make_pair

FPS = Fraction(30000, 1001)
A = SynthSpec(320, 192, 8)
B = SynthSpec(320, 192, 8, gain_scale=3)
BATCH = 2


def specs(nbatches: int = 5):
    """nbatches batches of BATCH frame pairs; the scene changes in the middle of the video, inside a batch"""
    n = nbatches * BATCH
    return [A] * (n // 2) + [B] * (n - n // 2)


SPECS = specs(int(os.environ.get("G1S_TEST_BATCHES", "5")))


def main(out_path):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = StreamingShardedDiff(FPS, 8, 8, device=0, batch_frames=BATCH, group=dist)
    nbatches = (len(SPECS) + BATCH - 1) // BATCH
    rounds = (nbatches + world - 1) // world
    keep = []
    for r in range(rounds):
        j = r * world + rank  # global batch index of this rank in round r
        if j < nbatches:
            pairs = [make_pair(SPECS[k], k, device="cuda") for k in range(j * BATCH, min((j + 1) * BATCH, len(SPECS)))]
            keep.append(pairs)
            sd.diff_prepared(DiffGenerator.prepare_frames(pairs, 1, 1))
        else:
            sd.idle_round()
    segs = sd.finish()
    if rank == 0:
        with open(out_path, "wb") as f:
            f.write(format_tbl(segs))
    else:
        assert segs is None
    sd.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
