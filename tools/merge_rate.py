#!/usr/bin/env python3
"""tools/merge_rate.py [rounds] -- rank 0's ordered merge (g1s_shard_merge, eight messages of one 64-frame batch a round), microseconds
per frame as the video grows; G1S_MERGE_POOL = size of the merge pool (1: serial)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("G1S_MERGE_POOL", "1")
os.environ.setdefault("G1S_FOLD_THREADS", "8")
import numpy as np  # noqa: E402
from fractions import Fraction  # noqa: E402

from grav1synth_amd import _lib  # noqa: E402
from grav1synth_amd.diff import RecordFold, latest_from_records  # noqa: E402

if not os.path.exists("/tmp/g1s_host_budget_records.npy"):
    from tools.host_budget_8ranks import make_records

    np.save("/tmp/g1s_host_budget_records.npy", make_records())
L = _lib.lib()
R = np.load("/tmp/g1s_host_budget_records.npy")
blobs = latest_from_records(np.concatenate([R] * 32), 3)
mb = int(L.g1s_shard_msg_size(3, 64))
msgs = np.zeros((8, mb), dtype=np.uint8)
fold = RecordFold(Fraction(24, 1), 3)
for r in range(8):
    assert L.g1s_shard_msg_from_latest_at(blobs.ctypes.data, 64, 3, 64, 0, msgs[r].ctypes.data, mb) == 0
idx = msgs.view(np.uint32)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 400
step = max(1, total // 10)
rounds = 0
while rounds < total:
    t0 = time.perf_counter()
    c0 = time.process_time()
    for _ in range(step):
        idx[:, 4] = rounds
        assert L.g1s_shard_merge(fold._h, msgs.ctypes.data, msgs.strides[0], 8) == 0
        rounds += 1
    print("pool %s, frames %7d: %6.2f us wall, %6.2f us cpu per frame" % (os.environ["G1S_MERGE_POOL"], rounds * 512, (time.perf_counter() - t0) / (step * 512) * 1e6,
                                                                     (time.process_time() - c0) / (step * 512) * 1e6), flush=True)
