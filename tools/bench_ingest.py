#!/usr/bin/env python3
"""tools/bench_ingest.py [frames] -- rates of the paths that do NOT start from HBM-resident frames
(never bench.py's `value`): (a) pageable host frames through g1s_diff_frame (copied before the call returns),
(a2) PINNED host frames (on_device = 2: copies queued on the upload stream, g1s_diff_frames_copied), (b) two .y4m
files through g1s_diff_y4m_files (file read + PCIe inclusive; the readers' pinned rings feed the generator
asynchronously; G1S_INGEST_SYNC=1 = the synchronous path of round 1).  4K 10-bit 4:2:0, lag 3, chroma."""
import json, os, sys, tempfile, time
from fractions import Fraction

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grav1synth_amd.diff import DiffGenerator, Frame
from grav1synth_amd.ingest import diff_y4m_files, write_y4m
from grav1synth_amd.synth import SynthSpec, make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
spec = SynthSpec(3840, 2160, 10)
pairs = [make_pair(spec, k, device="cuda") for k in range(8)]
host = [([p.cpu().contiguous() for p in s], [p.cpu().contiguous() for p in d]) for s, d in pairs]
mpx = spec.width * spec.height / 1e6
out = {}
# (a) pageable host frames, the `&Frame` borrow of the reference: copied before diff_frame returns
for rep in range(2):
    g = DiffGenerator(Fraction(24, 1), 10, 10)
    t0 = time.perf_counter()
    for k in range(n):
        s, d = host[k % 8]
        g.diff_frame(s, d, 1, 1)
    g.finish()
    dt = time.perf_counter() - t0
    g.close()
out["host_frames_Mpx_s"] = n * mpx / dt
out["bytes_per_pair"] = 2 * spec.width * spec.height * 3  # 2 B x 1.5 samples x 2 sides
out["host_frames_GB_s"] = n * out["bytes_per_pair"] / dt / 1e9
# (a2) pinned host frames: queued copies
pinned = [([p.pin_memory() for p in s], [p.pin_memory() for p in d]) for s, d in host]
for rep in range(2):
    g = DiffGenerator(Fraction(24, 1), 10, 10)
    t0 = time.perf_counter()
    for k in range(n):
        s, d = pinned[k % 8]
        g.diff_frame(Frame(s, 1, 1, async_host=True), Frame(d, 1, 1, async_host=True))
    g.finish()
    dt = time.perf_counter() - t0
    g.close()
out["pinned_frames_Mpx_s"] = n * mpx / dt
out["pinned_frames_GB_s"] = n * out["bytes_per_pair"] / dt / 1e9
# (b) files
d = tempfile.mkdtemp(dir="/tmp")
write_y4m(d + "/src.y4m", (host[k % 8][0] for k in range(n)), 10, 1, 1)
write_y4m(d + "/den.y4m", (host[k % 8][1] for k in range(n)), 10, 1, 1)
for rep in range(2):
    t0 = time.perf_counter()
    frames, unequal = diff_y4m_files(d + "/src.y4m", d + "/den.y4m", d + "/out.tbl")
    dt = time.perf_counter() - t0
out["y4m_files_Mpx_s"] = frames * mpx / dt
out["frames"] = n
out["y4m_GB_s"] = frames * out["bytes_per_pair"] / dt / 1e9
for f in ("src.y4m", "den.y4m", "out.tbl"):
    os.remove(os.path.join(d, f))
print(json.dumps(out))
