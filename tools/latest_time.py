#!/usr/bin/env python3
"""tools/latest_time.py [WxH [BD [BATCH]]] -- k4_latest (the per-frame half of the fold on the device) next to the other
kernels of a batch: HIP events, one stream (G1S_LATEST=device), and the job rate with the half on the host / on the device."""
import os, sys, time
from fractions import Fraction
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
w, h = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
bd = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
import torch
from grav1synth_amd.diff import DiffGenerator, format_tbl
from grav1synth_amd.synth import SynthSpec, make_pair
spec = SynthSpec(w, h, bd)
pairs = [make_pair(spec, k, device="cuda") for k in range(min(B, 64))]
tables = {}
for where in ("host", "device"):
    os.environ["G1S_LATEST"] = where
    for timing in (True, False):
        g = DiffGenerator(Fraction(24, 1), bd, bd, batch_frames=B)
        g.set_timing(timing)
        n = 2 * B if timing else 16 * B
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            s, d = pairs[k % len(pairs)]
            g.diff_frame(s, d, spec.xdec, spec.ydec)
        tbl = format_tbl(g.finish())
        dt = time.perf_counter() - t0
        if timing:
            kt = g.kernel_times()
            print(where, {k: round(v[0] / v[1] * 1e3, 1) for k, v in kt.items()})
        else:
            print(where, f"{n} frames in {dt * 1e3:.1f} ms = {n / dt:.0f} frames/s = {n * w * h / dt / 1e6:.0f} Mpx/s")
        tables[(where, timing)] = tbl
        g.close()
assert len(set(tables.values())) == 1, "tables differ"
print("tables identical")
