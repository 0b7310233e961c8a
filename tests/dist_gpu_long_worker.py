"""python -m tests.dist_gpu_long_worker OUTDIR -- one rank of BASELINE.json configs[3] at its stated size: 3840x2160 10-bit 4:2:0,
lag 3, chroma, 1000 frames in eight shards (tests/test_gpu_long.py launches eight of these on ONE device over gloo, the way
G1S_BENCH_SHARE_GPU=1 runs bench.py --gpus 8 on a single-GPU box).

Three jobs over the same 1000 frames (tests/golden/make_golden.py LONG: scene cuts at frames 500 and 768), rank 0 writes a table each:

  streaming_host.tbl    StreamingShardedDiff (what bench.py --gpus N runs): 64-frame batches dealt round-robin (16 batches, the last
                        one 40 frames: rank r feeds batches r and r + 8), the per-frame half of the fold on each rank's host pool,
                        one gather of latest states a round, the ordered merge on rank 0.  Frame 768 is a batch boundary, frame 500
                        lies inside batch 7.
  streaming_device.tbl  the same with the per-frame half on the device (k4_latest, G1S_LATEST=device: what a 16-core quota takes).
  contiguous.tbl        ShardedDiff: 125 consecutive frames a rank ("125 per GPU x 8"), ONE all-gather of the integer records at
                        the end, the whole fold on rank 0.  Frame 500 is a shard boundary, frame 768 lies inside shard 6.

G1S_LONG_NAME picks the job (tests/golden/make_golden.py LONG).  The second one is configs[4]'s format in eight shards: 7680x4320 10-bit
4:4:4, 32 frames, batches and shards of 4 (one batch a rank), cuts at frame 16 (a boundary of both) and 22 (inside batch / shard 5).
"""
import os
import sys

import torch
import torch.distributed as dist

from grav1synth_amd.diff import DiffGenerator, format_tbl
from grav1synth_amd.dist import ShardedDiff, StreamingShardedDiff
from grav1synth_amd.synth import make_pair
from tests.golden import make_golden

NAME = os.environ.get("G1S_LONG_NAME", "oracle_full_3840x2160_10b_420_lag3_1000frames.tbl")  # (the 8K 4:4:4 job of configs[4]: 32 frames, 4 a batch and shard)
BATCH = int(make_golden.LONG[NAME].get("batch", 64))


def job():
    g = make_golden.LONG[NAME]
    from fractions import Fraction

    return g, make_golden.frame_specs(g), Fraction(*g["fps"])


def main(outdir):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, specs, fps = job()
    spec, n = g["spec"], g["frames"]
    nbatches = (n + BATCH - 1) // BATCH
    rounds = (nbatches + world - 1) // world

    # ---- the streaming job, twice: the per-frame half on the host pool, then on the device ----
    mine = {}  # global batch -> frame pairs (resident for both runs)
    for r in range(rounds):
        j = r * world + rank
        if j < nbatches:
            mine[j] = [make_pair(specs[k], k, device="cuda") for k in range(j * BATCH, min((j + 1) * BATCH, n))]
    prepared = {j: DiffGenerator.prepare_frames(p, spec.xdec, spec.ydec) for j, p in mine.items()}
    for where in ("host", "device"):
        os.environ["G1S_LATEST"] = where  # (read per generator)
        sd = StreamingShardedDiff(fps, spec.bit_depth, spec.bit_depth, ar_coeff_lag=g["lag"], device=0, batch_frames=BATCH, group=dist)
        for r in range(rounds):
            j = r * world + rank
            if j in prepared:
                sd.diff_prepared(prepared[j])
            else:
                sd.idle_round()
        segs = sd.finish()  # (raises when a fed frame was not merged)
        if rank == 0:
            with open(os.path.join(outdir, f"streaming_{where}.tbl"), "wb") as f:
                f.write(format_tbl(segs))
        else:
            assert segs is None
        sd.close()
        dist.barrier()
    os.environ.pop("G1S_LATEST")
    del prepared, mine
    torch.cuda.empty_cache()

    # ---- contiguous shards, one exchange at the end ----
    per = (n + world - 1) // world
    lo, hi = rank * per, min((rank + 1) * per, n)
    pairs = [make_pair(specs[k], k, device="cuda") for k in range(lo, hi)]
    sh = ShardedDiff(fps, spec.bit_depth, spec.bit_depth, ar_coeff_lag=g["lag"], device=0, batch_frames=BATCH, group=dist)
    for i in range(0, len(pairs), BATCH):
        sh.diff_prepared(DiffGenerator.prepare_frames(pairs[i:i + BATCH], spec.xdec, spec.ydec), spec.width, spec.height, 3)
    segs = sh.finish()
    if rank == 0:
        with open(os.path.join(outdir, "contiguous.tbl"), "wb") as f:
            f.write(format_tbl(segs))
    else:
        assert segs is None
    sh.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
