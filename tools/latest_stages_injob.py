#!/usr/bin/env python3
"""tools/latest_stages_injob.py [4k10|1080p8] [batches] -- the per-frame half's stages INSIDE the pipelined job (a library built with
`make -C grav1synth_amd/csrc variant NAME=lprof DEFS=-DG1S_LATEST_PROFILE`): CPU microseconds a frame per stage, summed over the
pool's threads (the timers are plain doubles added by sixteen threads: a few per cent of the additions are lost, evenly).
Stages as tools/latest_stages.py: 0 flat list + means, 1 integer sums -> f64 systems, 2 AR solves, 3 block statistics gathered,
7 noise variances, 4 luma strength + uncorrelated stds, 5 measurements accumulated, 6 strength solves."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "grav1synth_amd", "libg1s_v_lprof.so")
if not os.path.exists(lib):  # (no profiling build: the job's rate and cores only)
    lib = os.path.join(ROOT, "grav1synth_amd", "libg1s_diff.so")
os.environ["G1S_LIB"] = lib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from fractions import Fraction  # noqa: E402

import torch  # noqa: E402

from grav1synth_amd import _lib  # noqa: E402
from grav1synth_amd.diff import DiffGenerator  # noqa: E402
from grav1synth_amd.synth import SynthSpec, make_pair  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "4k10"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 120
W, H, bd, B = {"4k10": (3840, 2160, 10, 64), "1080p8": (1920, 1080, 8, 128)}[wl]
spec = SynthSpec(W, H, bd)
pairs = [make_pair(spec, k, device="cuda") for k in range(2 * B)]
prep = [DiffGenerator.prepare_frames(pairs[i:i + B], 1, 1) for i in (0, B)]
torch.cuda.synchronize()
L = _lib.lib()
syms = [s for s in os.popen(f"nm -D {lib}").read().split() if "latest_stage" in s]
arr = (C.c_double * 8).in_dll(L, syms[0]) if syms else (C.c_double * 8)()
for rep in range(2):
    for i in range(8):
        arr[i] = 0.0
    g = DiffGenerator(Fraction(24, 1), bd, bd, batch_frames=B)
    c0, t0 = sum(os.times()[:2]), time.perf_counter()
    for k in range(nb):
        g.diff_prepared(prep[k & 1], sync_torch=False)
    g.finish()
    dt, cpu = time.perf_counter() - t0, sum(os.times()[:2]) - c0
    g.close()
n = nb * B
print(f"{wl}: {n / dt:.0f} frames/s = {n * W * H / dt / 1e6:.0f} Mpx/s, {cpu / dt:.1f} cores busy = {cpu / n * 1e6:.1f} us of CPU a frame; the per-frame half's stages 0..7 "
      f"(us of CPU a frame):", [round(x / n * 1e6, 1) for x in arr], "sum", round(sum(arr) / n * 1e6, 1))
