#!/usr/bin/env python3
"""tools/bench_estimate.py [frames] -- N4 on the GPU box: `estimate`'s per-frame noise estimator over HBM-resident 4K 10-bit luma
planes: whole-loop rate (FFI call per frame, one launch + one 16-byte D2H per batch of 32) and the kernel alone (HIP events),
against the 8 TB/s HBM roofline (2 bytes per luma pixel: the plane is read once), with the oracle's scalar rate beside it."""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grav1synth_amd.estimate import NoiseEstimator
from grav1synth_amd.synth import SynthSpec, make_pair
from tests.oracle_binding import estimate_plane_noise

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spec = SynthSpec(3840, 2160, 10)
planes = [make_pair(spec, k, device="cuda")[0][0] for k in range(64)]
torch.cuda.synchronize()
out = {}
for rep in range(3):
    est = NoiseEstimator(10, batch_frames=32)
    est.kernel_time(True)
    t0 = time.perf_counter()
    for k in range(n):
        est.estimate_frame(planes[k % 64])
    got = est.finish()
    dt = time.perf_counter() - t0
    ms, fr = est.kernel_time(True)
    est.close()
px = spec.width * spec.height
out["frames"] = n
out["loop_Mpx_s"] = n * px / dt / 1e6
out["kernel_us_per_frame"] = ms * 1e3 / fr
out["kernel_GB_s"] = 2 * px * fr / (ms * 1e-3) / 1e9
out["kernel_roofline_frac"] = out["kernel_GB_s"] / 8000.0
p0 = planes[0].cpu().numpy()
t0 = time.perf_counter()
want = estimate_plane_noise(p0, 10)
out["oracle_Mpx_s_one_thread"] = px / (time.perf_counter() - t0) / 1e6
assert got[0] == want, (got[0], want)
out["estimate_frame0"] = got[0]
print(json.dumps(out))
