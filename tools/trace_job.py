#!/usr/bin/env python3
"""tools/trace_job.py [batches] -- G1S_TRACE timeline of the PIPELINED 4K 10-bit job (default streams, no profiler): per kernel
the time its stream reached it and the time the stream reached the next mark (= its end when nothing else gates the stream), and
the host's submit times; prints the batches in the middle of the job."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
path = os.environ.setdefault("G1S_TRACE", "/tmp/g1s_trace.txt")
if os.path.exists(path):
    os.remove(path)
from fractions import Fraction
import torch
from grav1synth_amd.diff import DiffGenerator
from grav1synth_amd.synth import SynthSpec, make_pair

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = 64
spec = SynthSpec(3840, 2160, 10, xdec=1, ydec=1)
pairs = [make_pair(spec, k, device="cuda") for k in range(128)]
torch.cuda.synchronize()
for rep in range(2):  # (the first job warms the box; the second is the one traced)
    if os.path.exists(path):
        os.remove(path)
    g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=B)
    t0 = time.perf_counter()
    for k in range(nb * B):
        s, d = pairs[k % len(pairs)]
        g.diff_frame(s, d, 1, 1, sync_torch=False)
    g.finish()
    dt = time.perf_counter() - t0
    g.close() if hasattr(g, "close") else None
    del g
print("job: %.1f us per batch" % (dt / nb * 1e6))
ev = []
streams = {}
for ln in open(path):
    f = ln.split()
    if f[0] == "G":
        t, slot, st, name = float(f[1]), int(f[3]), f[5], " ".join(f[6:])
        streams.setdefault(st, len(streams))
        ev.append((t, "G", streams[st], slot, name))
    else:
        ev.append((float(f[1]), "H", -1, int(f[-1]), " ".join(f[2:-2])))
ev.sort()
# end of a G mark = the next mark on the same stream
nxt = {}
out = []
for i in range(len(ev) - 1, -1, -1):
    t, kind, st, slot, name = ev[i]
    if kind == "G":
        end = nxt.get(st)
        nxt[st] = t
        out.append((t, end, kind, st, slot, name))
    else:
        out.append((t, None, kind, st, slot, name))
out.reverse()
g_only = [o for o in out if o[2] == "G"]
tmid = g_only[len(g_only) // 2][0]
for t, end, kind, st, slot, name in out:
    if t < tmid - 1500 or t > tmid + 2500:
        continue
    if kind == "H":
        print("%10.1f            host  %-16s slot %d" % (t - tmid, name, slot))
    elif name != "-":
        print("%10.1f %9.1f  s%d  slot %d  %s" % (t - tmid, (end - t) if end else -1, st, slot, name))
