#!/usr/bin/env python3
"""tools/diff_pmc.py [batches] -- the 4K 10-bit bench workload as a lean driver for rocprofv3 --pmc runs: frames made on the CPU
(no torch kernels on the device under the profiler), 4 distinct pairs copied to 64 distinct addresses, 64-frame batches (BATCH=...), one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("G1S_ONE_STREAM", "1")
from fractions import Fraction
import torch
from grav1synth_amd.diff import DiffGenerator
from grav1synth_amd.synth import SynthSpec, make_pair

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(os.environ.get("BATCH", "64"))
spec = SynthSpec(3840, 2160, 10)
t0 = time.time()
pairs = []
for k in range(4):
    s, d = make_pair(spec, k, device="cpu")
    pairs.append(([p.cuda() for p in s], [p.cuda() for p in d]))
torch.cuda.synchronize()
print("frames ready in %.1f s" % (time.time() - t0), flush=True)
# every frame of a launch at its own address (copies of the four pairs): what the caches see is what a video gives them
pairs = [pairs[k] if k < 4 else ([p.clone() for p in pairs[k % 4][0]], [p.clone() for p in pairs[k % 4][1]]) for k in range(B)]
torch.cuda.synchronize()
g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=B)
for k in range(nb * B):
    s, d = pairs[k % B]
    g.diff_frame(s, d, 1, 1, sync_torch=False)
g.sync()
print("done", len(g.finish()), "segments", flush=True)
