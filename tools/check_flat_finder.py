#!/usr/bin/env python3
"""tools/check_flat_finder.py [frames] -- certified fast path vs literal kernel of the flat-block finder on
4K 10-bit frames (8160 blocks each): mask bytes and f32 score bits must agree for every block."""
import sys
from fractions import Fraction

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from grav1synth_amd.diff import DiffGenerator  # noqa: E402
from grav1synth_amd.synth import SynthSpec, make_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bad = tot = lit = 0
for textured, gain in ((True, 1), (False, 1), (True, 3), (True, 7)):
    spec = SynthSpec(3840, 2160, 10, textured=textured, gain_scale=gain)
    out = []
    for literal in (False, True):
        g = DiffGenerator(Fraction(24, 1), 10, 10, batch_frames=1)
        g.set_flat_finder(literal)
        g.set_timing(True)
        rs = []
        for k in range(n):
            s, d = make_pair(spec, 1000 + k, device="cuda")
            g.diff_frame(s, d, 1, 1)
            g.sync()
            r = g.last_record()
            rs.append((r.flat_mask().copy(), r.scores().view(np.uint32).copy()))
        if not literal:
            lit += g.stats().literal_blocks
        out.append(rs)
        g.close()
    for (m0, s0), (m1, s1) in zip(*out):
        bad += int((m0 != m1).sum()) + int((s0 != s1).sum())
        tot += m0.size
print(f"blocks {tot}  mismatches {bad}  sent to the literal kernel by the fast path {lit} ({100.0 * lit / tot:.3f} %)")
sys.exit(1 if bad else 0)
