#!/usr/bin/env python3
"""tools/ktime.py [batches] -- per-kernel HIP-event times of the 4K 10-bit bench workload (64-frame batches, one stream), errors of
the fold ignored: for timing experiments that leave parts of a kernel out (G1S_S_DBG, G1S_DBG_SKIP)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractions import Fraction
import torch
from grav1synth_amd.diff import DiffGenerator
from grav1synth_amd.synth import SynthSpec, make_pair

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(os.environ.get("BATCH", "64"))
flat = os.environ.get("FLAT") is not None
wl = os.environ.get("WL", "4k10")
W, H, bd, xd, yd = {"4k10": (3840, 2160, 10, 1, 1), "1080p8": (1920, 1080, 8, 1, 1), "8k10_444": (7680, 4320, 10, 0, 0)}[wl]
spec = SynthSpec(W, H, bd, xdec=xd, ydec=yd, textured=not flat)
nd = int(os.environ.get("DISTINCT", "64"))  # (distinct frame pairs: 64 = none shared inside a launch, what a video gives the caches)
dbd = int(os.environ.get("DEN_BD", str(bd)))  # (a denoised video of another depth: the wide chain's general residual form)
if dbd == bd:
    pairs = [make_pair(spec, k, device="cuda") for k in range(nd)]
else:
    dspec = SynthSpec(W, H, dbd, xdec=xd, ydec=yd, textured=not flat)
    pairs = [(make_pair(spec, k, device="cuda")[0], make_pair(dspec, k, device="cuda")[1]) for k in range(nd)]
torch.cuda.synchronize()
g = DiffGenerator(Fraction(24, 1), bd, dbd, batch_frames=B)
try:
    # one untimed batch first: the first launch of a kernel in a process loads its code object (milliseconds, on whichever
    # kernel it lands)
    for k in range(B):
        s, d = pairs[k % nd]
        g.diff_frame(s, d, xd, yd, sync_torch=False)
    g.sync()
except Exception as e:
    print("error (ignored):", str(e)[:100], file=sys.stderr)
g.set_timing(True)
try:
    for k in range(nb * B):
        s, d = pairs[k % nd]
        g.diff_frame(s, d, xd, yd, sync_torch=False)
    g.sync()
except Exception as e:
    print("error (ignored):", str(e)[:100], file=sys.stderr)
kt = g.kernel_times()
out = {k: round(v[0] / v[1] * 1e3, 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}
# the same batches once more with ONE pair of events around each batch's chain of kernels (set_timing(2))
chain_us = None
try:
    g.set_timing(2)
    for k in range(nb * B):
        s, d = pairs[k % nd]
        g.diff_frame(s, d, xd, yd, sync_torch=False)
    g.sync()
    st = g.stats()
    chain_us = round(st.ms_chain / st.chain_batches * 1e3, 1) if st.chain_batches else None
except Exception as e:
    print("error (ignored):", str(e)[:100], file=sys.stderr)
print(json.dumps({"tag": os.environ.get("TAG", ""), "sum_us": round(sum(out.values()), 1), "chain_us": chain_us, "kernels_us": out}))
