#!/usr/bin/env python3
"""tools/job_rate.py [batches] -- the pipelined 4K 10-bit job's batch period under the environment's switches, errors of the fold ignored
(for timing experiments whose switches may leave the results wrong)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from fractions import Fraction
import torch
from grav1synth_amd.diff import DiffGenerator
from grav1synth_amd.synth import SynthSpec, make_pair

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 240
B = 64
spec = SynthSpec(3840, 2160, 10)
pairs = [make_pair(spec, k, device="cuda") for k in range(4 * B)]
prep = [DiffGenerator.prepare_frames(pairs[i:i + B], 1, 1) for i in range(0, 4 * B, B)]
torch.cuda.synchronize()
for rep in range(2):
    g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=B)
    t0 = time.perf_counter()
    try:
        for k in range(nb):
            g.diff_prepared(prep[k & 3], sync_torch=False)
        g.sync()
    except Exception as e:
        print("(fold error ignored:", str(e)[:60], ")")
    dt = time.perf_counter() - t0
    try:
        g.finish()
    except Exception:
        pass
    g.close()
print("%.1f us a batch = %.0f Mpx/s" % (dt / nb * 1e6, nb * B * 3840 * 2160 / dt / 1e6))
