#!/usr/bin/env python3
"""tools/bench_fold.py [WxH] -- the host half of the pass, in CPU-seconds per frame (no GPU needed).

A frame's record (exact integer sums, ~1 MB at 4K) goes through two host stages:
  per-frame half   g1s_latest_from_record: symmetric mirror, AR solve, block measurements, strength solve -> a ~27 KB
                   latest state.  Independent across frames: runs on the generator's thread pool, on EVERY rank of a
                   frame-shard job, for that rank's frames only.
  ordered merge    g1s_fold_push_latest: the sequential noise-model update.  Serial, rank 0 only.
Printed: single-thread time of each per frame.  The budget they are held against (multi-GPU): a rank that
diffs F frames/s needs F x per-frame CPU-seconds of cores; rank 0 of an N-rank job needs N x F x merge seconds < 1."""
import json
import os
import sys
import time
from fractions import Fraction

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grav1synth_amd.diff import RecordFold, latest_from_records  # noqa: E402
from grav1synth_amd.synth import SynthSpec  # noqa: E402
from tests.helpers import oracle_run, record_from_oracle  # noqa: E402

w, h = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
spec = SynthSpec(w, h, 10)
recs = []
oracle_run(spec, [0, 1], 3, True, collect=lambda o, k: recs.append(record_from_oracle(o, spec, 3, 3).buf.copy()))
R = np.stack(recs)
os.environ.setdefault("G1S_FOLD_THREADS", "1")  # single-thread figures
reps = 20
many = np.concatenate([R] * reps)
t0 = time.perf_counter()
blobs = latest_from_records(many, 3)
per_frame = (time.perf_counter() - t0) / len(many)
big = np.concatenate([blobs] * 50)
f = RecordFold(Fraction(24, 1), 3)
t0 = time.perf_counter()
f.push_latest_many(big)
merge = (time.perf_counter() - t0) / len(big)
f.finish()
f.close()
out = {"frame": f"{w}x{h} 10-bit 4:2:0 lag 3", "record_bytes": int(R.shape[1]), "latest_bytes": int(blobs.shape[1]),
       "per_frame_half_cpu_us": round(per_frame * 1e6, 1), "ordered_merge_cpu_us": round(merge * 1e6, 2), "hw_threads": os.cpu_count()}
# the ordered merge with its pool (the solves of a window of frames run on the merge pool, the rest is serial): wall time per
# frame, what rank 0 of an N-rank job has to stay under (1 / (N x frames/s of a rank)).  A fresh process per pool size
# (the pool is made once per process).
if len(sys.argv) <= 2:
    import subprocess

    out["ordered_merge_wall_us_by_threads"] = {}
    for t in (1, 4, 8, 16, 32, 64):
        if t > (os.cpu_count() or 1):
            break
        r = subprocess.run([sys.executable, __file__, f"{w}x{h}", "merge-only"], env=dict(os.environ, G1S_FOLD_THREADS=str(t)),
                           capture_output=True, text=True)
        try:
            out["ordered_merge_wall_us_by_threads"][str(t)] = json.loads(r.stdout.strip().splitlines()[-1])["ordered_merge_cpu_us"]
        except Exception:
            out["ordered_merge_wall_us_by_threads"][str(t)] = None
print(json.dumps(out))
