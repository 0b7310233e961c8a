#!/usr/bin/env python3
"""tools/soak.py [N] -- create / run / free many generators (thread, stream-set and slot-cache lifecycle), some
freed without finish, some with a partial last batch; prints the table digest (must be one value) and the time."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractions import Fraction
import torch
from grav1synth_amd.diff import DiffGenerator, format_tbl
from grav1synth_amd.synth import SynthSpec, make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
spec = SynthSpec(352, 208, 10)
pairs = [make_pair(spec, k, device="cuda") for k in range(7)]
digests = set()
t0 = time.time()
for i in range(n):
    g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=1 + i % 4)
    for s, d in pairs[: 3 + i % 5]:
        g.diff_frame(s, d, 1, 1)
    if i % 7 == 3:
        g.close()          # freed with work queued and no finish
        continue
    if i % 5 == 0:
        g.sync()
    for s, d in pairs[3 + i % 5:]:
        g.diff_frame(s, d, 1, 1)
    digests.add(hashlib.sha256(format_tbl(g.finish())).hexdigest())
    g.close()
print(f"{n} generators, {len(digests)} distinct table(s), {time.time() - t0:.1f} s, "
      f"{torch.cuda.memory_allocated() >> 20} MiB held by torch")
sys.exit(0 if len(digests) == 1 else 1)
