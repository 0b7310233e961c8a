#!/usr/bin/env python3
"""tools/half_scaling.py -- the per-frame half of one rank (g1s_latest_from_records) against the size of its pool."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from grav1synth_amd import _lib
L = _lib.lib()
R = np.load("/tmp/g1s_host_budget_records.npy")
src = np.concatenate([R] * 32)
bs = int(L.g1s_latest_size(3))
blobs = np.zeros((64, bs), dtype=np.uint8)
L.g1s_latest_from_records(src.ctypes.data, src.shape[1], 64, 3, blobs.ctypes.data, bs)
t0 = time.perf_counter(); c0 = time.process_time(); n = 0
while time.perf_counter() - t0 < 3.0:
    L.g1s_latest_from_records(src.ctypes.data, src.shape[1], 64, 3, blobs.ctypes.data, bs); n += 64
w = time.perf_counter() - t0
print("%%2s threads: %%7.0f frames/s, %%6.1f us cpu per frame" %% (os.environ["G1S_FOLD_THREADS"], n / w, (time.process_time() - c0) / n * 1e6))
''' % ROOT
if __name__ == "__main__":
    if not os.path.exists("/tmp/g1s_host_budget_records.npy"):
        sys.path.insert(0, ROOT)
        import numpy as np
        from tools.host_budget_8ranks import make_records
        np.save("/tmp/g1s_host_budget_records.npy", make_records())
    for t in (1, 2, 4, 8, 16, 32):
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, G1S_FOLD_THREADS=str(t)))
