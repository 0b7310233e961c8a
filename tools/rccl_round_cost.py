#!/usr/bin/env python3
"""tools/rccl_round_cost.py -- what a frame-shard round costs the feeding thread over the real backend ("nccl" = RCCL) with ONE
rank (two ranks on a device are refused): the 4K bench workload in 64-frame batches through StreamingShardedDiff (pack ->
pinned buffer -> device -> rooted gather -> pinned ring -> merger thread) against the plain generator."""
import os, sys, time
from fractions import Fraction
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29537")
import torch, torch.distributed as dist
from grav1synth_amd.diff import DiffGenerator, format_tbl
from grav1synth_amd.dist import StreamingShardedDiff
from grav1synth_amd.synth import SynthSpec, make_pair
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
spec = SynthSpec(3840, 2160, 10)
B, NB = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 200
pairs = [make_pair(spec, k, device="cuda") for k in range(B)]
prep = DiffGenerator.prepare_frames(pairs, 1, 1)
res = {}
for mode in ("rounds over RCCL", "plain generator", "rounds over RCCL", "plain generator"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode.startswith("rounds"):
        sd = StreamingShardedDiff(Fraction(24, 1), 10, 10, device=0, batch_frames=B, group=dist)
        for _ in range(NB):
            sd.diff_prepared(prep, sync_torch=False)
        tbl = format_tbl(sd.finish())
        ex = sd.exchange_s * 1e3 / max(sd.exchange_rounds, 1)
        sd.close()
    else:
        g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=B)
        for _ in range(NB):
            g.diff_prepared(prep, sync_torch=False)
        tbl = format_tbl(g.finish()); g.close(); ex = None
    dt = time.perf_counter() - t0
    print(f"{mode}: {NB * B} frames in {dt * 1e3:.1f} ms = {NB * B * spec.width * spec.height / dt / 1e6:.0f} Mpx/s" + (f", exchange {ex:.3f} ms per round on the feeding thread" if ex is not None else ""))
    res[mode] = tbl
assert len(set(res.values())) == 1
print("tables identical")
dist.destroy_process_group()
