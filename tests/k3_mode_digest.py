"""python -m tests.k3_mode_digest -- one JSON line: for each case, the SHA-256 of what a frame's record means (flat mask,
score bits, AR sums, statistics of the flat blocks) and of the final table.  G1S_K3 is read once per process, so
tests/test_gpu_selfcheck.py::test_wide_and_stream_chains_agree runs this once per chain and compares the lines."""
import hashlib
import json
from fractions import Fraction

import numpy as np

from grav1synth_amd.diff import DiffGenerator, format_tbl
from grav1synth_amd.synth import SynthSpec, make_pair

CASES = [
    (SynthSpec(320, 200, 8), 3, True, 3),
    (SynthSpec(326, 198, 10), 2, True, 2),
    (SynthSpec(256, 160, 10, xdec=0, ydec=0), 3, True, 2),
    (SynthSpec(320, 192, 10, xdec=1, ydec=0), 1, True, 2),
    (SynthSpec(320, 192, 8, xdec=0, ydec=1), 3, True, 2),
    (SynthSpec(320, 192, 8), 2, False, 2),
    (SynthSpec(1280, 720, 12), 3, True, 2),                   # 12-bit: residuals outside int8, the deferred blocks
    (SynthSpec(960, 544, 10, gain_scale=8), 3, True, 2),
    (SynthSpec(3840, 2160, 10), 3, True, 3),
]


def main():
    out = []
    for spec, lag, chroma, n in CASES:
        h = hashlib.sha256()
        g = DiffGenerator(Fraction(24, 1), spec.bit_depth, spec.bit_depth, ar_coeff_lag=lag, luma_only=not chroma, batch_frames=2)
        pairs = [make_pair(spec, 7 + k, device="cuda") for k in range(n)]
        for s, d in pairs:
            g.diff_frame(s, d, spec.xdec, spec.ydec)
            g.sync()
            r = g.last_record()
            flat = r.flat_mask().ravel() != 0
            h.update(r.flat_mask().tobytes())
            h.update(r.scores().view(np.uint32).tobytes())
            for c in range(3 if chroma else 1):
                S, Sb, nobs = r.ar_sums(c)
                h.update(np.ascontiguousarray(S).tobytes())
                h.update(np.ascontiguousarray(Sb).tobytes())
                h.update(str(int(nobs)).encode())
                for a in r.block_stats(c):
                    h.update(np.ascontiguousarray(a.ravel()[flat]).tobytes())
        h.update(format_tbl(g.finish()))
        out.append(h.hexdigest())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
