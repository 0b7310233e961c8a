#!/usr/bin/env python3
"""tools/latest_stages.py [WxH] -- where the per-frame half of the fold (compute_latest, csrc/fold.cpp) spends its time, on the host
this runs on: a library built with `make -C grav1synth_amd/csrc variant NAME=lprof DEFS=-DG1S_LATEST_PROFILE` (timers between the
stages), records made by the oracle (no GPU needed), one thread.  Stages: 0 flat list + means, 1 integer sums -> f64 systems,
2 AR solves, 3 block statistics gathered, 7 noise variances, 4 luma strength + uncorrelated stds, 5 measurements accumulated,
6 strength solves."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "grav1synth_amd", "libg1s_v_lprof.so")
os.environ["G1S_LIB"] = lib
os.environ["G1S_FOLD_THREADS"] = "1"
import numpy as np  # noqa: E402

from grav1synth_amd import _lib  # noqa: E402
from grav1synth_amd.diff import latest_from_records  # noqa: E402
from grav1synth_amd.synth import SynthSpec  # noqa: E402
from tests.helpers import oracle_run, record_from_oracle  # noqa: E402

w, h = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
bd = 10 if w > 1920 else 8
spec = SynthSpec(w, h, bd)
recs = []
oracle_run(spec, [0, 1], 3, True, collect=lambda o, k: recs.append(record_from_oracle(o, spec, 3, 3).buf.copy()))
many = np.concatenate([np.stack(recs)] * (20 if w > 1920 else 80))
L = _lib.lib()
latest_from_records(many[:2], 3)  # (warm)
sym = [s for s in os.popen(f"nm -D {lib}").read().split() if "latest_stage" in s][0]
arr = (C.c_double * 8).in_dll(L, sym)
for i in range(8):
    arr[i] = 0.0
t0 = time.perf_counter()
latest_from_records(many, 3)
dt = (time.perf_counter() - t0) / len(many)
print(f"{w}x{h}: per-frame half {dt * 1e6:.1f} us a frame, one thread; stages 0..7 (us a frame):", [round(x / len(many) * 1e6, 1) for x in arr])
