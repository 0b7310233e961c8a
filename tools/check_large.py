#!/usr/bin/env python3
"""tools/check_large.py [W H BD XDEC YDEC FRAMES] -- one-off full-size parity run: the test case of
tests/test_gpu_parity.py (records and table against the CPU oracle) at a BASELINE.json size the test
suite cannot afford (the oracle needs ~15 s per 4K frame, ~1.5 min per 8K 4:4:4 frame).
Default: 7680x4320 10-bit 4:4:4, 2 frames (configs[4])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grav1synth_amd.synth import SynthSpec
from tests import test_gpu_parity as T

a = [int(x) for x in sys.argv[1:]]
w, h, bd, xd, yd, n = (a + [7680, 4320, 10, 0, 0, 2][len(a):])[:6]
spec = SynthSpec(w, h, bd, xdec=xd, ydec=yd)
t0 = time.time()
T.test_records_and_table_match_oracle.__wrapped__(((spec, 3, True, n, True))) if hasattr(
    T.test_records_and_table_match_oracle, "__wrapped__") else T.test_records_and_table_match_oracle((spec, 3, True, n, True))
print(f"OK: {w}x{h} {bd}-bit xdec={xd} ydec={yd}, {n} frames: flat mask, score bits, AR sums, block statistics and "
      f".tbl identical to the oracle ({time.time() - t0:.0f} s)")
